"""TEST INFRASTRUCTURE: numpy restatement of the y-slab kernels (same method surface
as qgcm_hip.slab.HipSlab) so that the distributed orchestration `SlabOcean` and the
transports can be exercised on CPU (LocalComm in-process, DistComm over gloo).
Each method follows the same reference lines as the HIP kernel it stands in for
(k_tend.h, k_dst*.h, k_thomas.h, k_misc.h)."""
import numpy as np
import scipy.fft
import torch

from qgcm_hip.slab import local_rows, slab_slice


class NumpySlab:
    def __init__(self, cfg, consts, g0, g1, rank, nranks):
        self.cfg, self.rank, self.nranks, self.g0, self.g1 = cfg, rank, nranks, g0, g1
        self.nyg = cfg.nypo
        self.nyl, self.joff, self.jlo, self.jhi = local_rows(self.nyg, g0, g1)
        sl = slab_slice(self.nyg, g0, g1)
        self.c = consts
        self.yporel = consts["yporel"][sl].copy()
        self.ddynoc = consts["ddynoc"][:, sl].copy()
        self.ochom = consts["ochom"][:, sl, :].copy()
        nx, nl = cfg.nxpo, cfg.nlo
        self.nk = nx - 2
        z3 = lambda: np.zeros((nx, self.nyl, nl))
        self.po, self.pom, self.qo, self.qom = z3(), z3(), z3(), z3()
        self.wekpo = np.zeros((nx, self.nyl))
        self.entoc = np.zeros((nx, self.nyl))
        self.xon = np.zeros(nl - 1)
        self.scal = np.zeros(2 * (nl - 1) + 4 * nl)
        self.wrk = np.zeros((self.nk, self.nyl, nl))
        # owned rows that are interior to the global domain (0-based local indices)
        self.r0 = self.jlo - 1 + (1 if g0 == 1 else 0)
        self.r1 = self.jhi - 1 - (1 if g1 == self.nyg else 0)
        self.th_len = 3 * nl * self.nk
        self.cst_len = 4 * nl * self.nk
        self.cgath = None
        # weights of the spectral area integral: sum_i 2 sin(k i pi/n) = 2 cot(k pi/2n) for odd k (k_thomas.h)
        n = nx - 1
        kk = np.arange(1, self.nk + 1)
        self.wcot = np.where(kk % 2 == 1, 2.0 / np.tan(kk * np.pi / (2.0 * n)), 0.0)
        self.ksum = np.zeros((self.nk, nl))
        self.halo_len = 4 * nl * nx
        # Thomas pivots of the slab rows (src/ocisubs.F:472-477), global recurrence
        rg0 = (self.r0 + 1 + self.joff) - 2
        nr = self.r1 - self.r0 + 1
        self.bet = np.zeros((self.nk, nr, nl))
        for m in range(nl):
            boc = consts["bd2oc"][:self.nk] - consts["rdm2oc"][m]
            betinv = 1.0 / boc
            for rg in range(rg0 + nr):
                if rg > 0:
                    gam = consts["aoc"] * betinv
                    betinv = 1.0 / (boc - consts["aoc"] * gam)
                if rg >= rg0:
                    self.bet[:, rg - rg0, m] = betinv
        self.hclco = np.zeros(nl - 1)

    # -- buffers ---------------------------------------------------------------
    def new_buffer(self, n):
        return torch.zeros(int(n), dtype=torch.float64)

    def sync(self):
        pass

    def close(self):
        pass

    # -- state -------------------------------------------------------------------
    def set_state(self, po, pom, qo, qom):
        self.po, self.pom, self.qo, self.qom = [np.array(x, dtype=np.float64) for x in (po, pom, qo, qom)]

    def get_state(self):
        return [self.po.copy(), self.pom.copy(), self.qo.copy(), self.qom.copy()]

    def set_forcing(self, wekpo, entoc, xon):
        self.wekpo, self.entoc, self.xon = np.array(wekpo), np.array(entoc), np.array(xon, dtype=np.float64)

    def set_scalars(self, s):
        self.scal = np.array(s, dtype=np.float64)

    def get_scalars(self):
        return self.scal.copy()

    # -- helpers -----------------------------------------------------------------
    def _G(self):
        return np.arange(1, self.nyl + 1) + self.joff  # global row of each local row

    def _lap_bc(self, f, bcf, dxom2):
        """5-point Laplacian with the mixed-BC rules of qgosubs.F:94-127 on every local
        row that can be formed (first/last local rows of interior slabs stay 0)."""
        nx, ny = f.shape
        out = np.zeros_like(f)
        G = self._G()
        out[1:-1, 1:-1] = (f[1:-1, :-2] + f[:-2, 1:-1] + f[2:, 1:-1] + f[1:-1, 2:] - 4.0 * f[1:-1, 1:-1]) * dxom2
        out[0, 1:-1] = bcf * (f[1, 1:-1] - f[0, 1:-1])
        out[-1, 1:-1] = bcf * (f[-2, 1:-1] - f[-1, 1:-1])
        if G[0] == 1:
            out[:, 0] = bcf * (f[:, 1] - f[:, 0])
        if G[-1] == self.nyg:
            out[:, -1] = bcf * (f[:, -2] - f[:, -1])
        return out

    # -- kernels -------------------------------------------------------------------
    def qgostep(self):
        cfg, c = self.cfg, self.c
        nl = cfg.nlo
        dxom2 = 1.0 / (cfg.dxo * cfg.dxo)
        adfaco = 1.0 / (12.0 * cfg.dxo * cfg.dyo * cfg.fnot)
        bcf = cfg.bccooc * dxom2 / (0.5 * cfg.bccooc + 1.0)
        bdrfac = 0.5 * np.sign(cfg.fnot) * cfg.delek / cfg.hoc[-1]
        G = self._G()
        own = np.zeros(self.nyl, dtype=bool)
        own[self.jlo - 1:self.jhi] = True
        step = own & (G >= 2) & (G <= self.nyg - 1)   # rows that are time-stepped
        qnew = self.qom.copy()
        dq = np.zeros_like(self.qo)
        d2bot = None
        for k in range(nl):
            d2 = self._lap_bc(self.pom[:, :, k], bcf, dxom2)
            d4 = self._lap_bc(d2, bcf, dxom2)
            d6 = np.zeros_like(d4)
            d6[1:-1, 1:-1] = dxom2 * (d4[1:-1, :-2] + d4[:-2, 1:-1] + d4[2:, 1:-1] + d4[1:-1, 2:] - 4.0 * d4[1:-1, 1:-1])
            p, q = self.po[:, :, k], self.qo[:, :, k]
            P = lambda di, dj: p[1 + di:p.shape[0] - 1 + di, 1 + dj:p.shape[1] - 1 + dj]
            Q = lambda di, dj: q[1 + di:q.shape[0] - 1 + di, 1 + dj:q.shape[1] - 1 + dj]
            jac = ((Q(1, 0) - Q(-1, 0)) * (P(0, 1) - P(0, -1)) + (Q(0, -1) - Q(0, 1)) * (P(1, 0) - P(-1, 0))
                   + Q(1, 0) * (P(1, 1) - P(1, -1)) - Q(-1, 0) * (P(-1, 1) - P(-1, -1))
                   - Q(0, 1) * (P(1, 1) - P(-1, 1)) + Q(0, -1) * (P(1, -1) - P(-1, -1))
                   + P(0, 1) * (Q(1, 1) - Q(-1, 1)) - P(0, -1) * (Q(1, -1) - Q(-1, -1))
                   - P(1, 0) * (Q(1, 1) - Q(1, -1)) + P(-1, 0) * (Q(-1, 1) - Q(-1, -1)))
            diffus = (cfg.ah2oc[k] / cfg.fnot) * d4[1:-1, 1:-1] - (cfg.ah4oc[k] / cfg.fnot) * d6[1:-1, 1:-1]
            dq[1:-1, 1:-1, k] = adfaco * jac + diffus
            if k == nl - 1:
                d2bot = d2
        qdot = dq.copy()
        qdot[:, :, 0] = dq[:, :, 0] + (cfg.fnot / cfg.hoc[0]) * (self.wekpo - self.entoc)
        qdot[:, :, 1] = dq[:, :, 1] + (cfg.fnot / cfg.hoc[1]) * self.entoc
        qdot[:, :, nl - 1] = qdot[:, :, nl - 1] - bdrfac * d2bot
        for k in range(nl):
            qnew[:, step, k] = self.qom[:, step, k] + cfg.tdto * qdot[:, step, k]
        edge = own & ((G == 1) | (G == self.nyg))
        qnew[:, edge, :] = self.qo[:, edge, :]
        # projection (ocisubs.F:117-139) on the stepped rows, interior columns
        betay = cfg.beta * self.yporel
        ql = qnew - betay[None, :, None]
        ql[:, :, nl - 1] = ql[:, :, nl - 1] - self.ddynoc
        for m in range(nl):
            qm = np.zeros((self.cfg.nxpo, self.nyl))
            for k in range(nl):
                qm = qm + c["ctl2moc"][k, m] * ql[:, :, k]
            self.wrk[:, step, m] = cfg.fnot * qm[1:-1, :][:, step]
        self.qom = self.qo       # rotation: old qo becomes qom
        self.qo = qnew

    def row_transform(self, inverse):
        r0, r1 = self.r0, self.r1
        self.wrk[:, r0:r1 + 1, :] = scipy.fft.dst(self.wrk[:, r0:r1 + 1, :], type=1, axis=0)

    def _fwd(self, w, uin):
        """forward sweep over the slab rows from inflow uin (nk, nl)."""
        a = self.c["aoc"]
        u = np.zeros_like(w)
        prev = uin
        for r in range(w.shape[1]):
            prev = (w[:, r, :] - a * prev) * self.bet[:, r, :]
            u[:, r, :] = prev
        return u

    def _bwd(self, u, vin):
        a = self.c["aoc"]
        v = np.zeros_like(u)
        nxt = vin
        for r in range(u.shape[1] - 1, -1, -1):
            nxt = u[:, r, :] - a * self.bet[:, r, :] * nxt
            v[:, r, :] = nxt
        return v

    def thomas_phase(self, phase, gath, send):
        """Same single-exchange protocol as k_thomas.h: a slab is summarised per wavenumber
        and mode by (Cf, D, Cb, E, S0, SP, SQ); phase 2 also leaves the basin-wide spectral
        column sums (ksum) behind, from which constr() takes the area integrals."""
        a = self.c["aoc"]
        nl, nk = self.cfg.nlo, self.nk
        z, one = np.zeros((nk, nl)), np.ones((nk, nl))
        w = self.wrk[:, self.r0:self.r1 + 1, :]
        ft = 0.5 / (self.cfg.nxpo - 1)
        if phase == 1:
            u0 = self._fwd(w, z)
            v0 = self._bwd(u0, z)
            t = send.numpy().reshape(nl, nk, 3)
            for i, x in enumerate((u0[:, -1, :], v0[:, 0, :], v0.sum(axis=1))):
                t[:, :, i] = x.T
            return
        g = gath.numpy().reshape(self.nranks, nl, nk, 3).transpose(0, 2, 1, 3)  # (P, nk, nl, 3): Cf, Cb, S0
        if self.nranks == 1:
            k = self._consts().reshape(1, nl, nk, 4).transpose(0, 2, 1, 3)
        else:
            k = self.cgath.reshape(self.nranks, nl, nk, 4).transpose(0, 2, 1, 3)   # (P, nk, nl, 4): D, E, SP, SQ
        us = []
        u = z
        for r in range(self.nranks):
            us.append(u)
            u = g[r, :, :, 0] + k[r, :, :, 0] * u
        vs = [None] * self.nranks
        v = z
        for r in range(self.nranks - 1, -1, -1):
            vs[r] = v
            v = g[r, :, :, 1] + k[r, :, :, 1] * us[r] + k[r, :, :, 0] * v
        tot = z
        for r in range(self.nranks):
            tot = tot + (g[r, :, :, 2] + us[r] * k[r, :, :, 2] + vs[r] * k[r, :, :, 3])
        self.ksum = ft * tot
        uf = self._fwd(w, us[self.rank])
        self.wrk[:, self.r0:self.r1 + 1, :] = ft * self._bwd(uf, vs[self.rank])

    def _consts(self):
        """(nl, nk, 4) = D, E, SP, SQ of this slab: right-hand-side independent."""
        a = self.c["aoc"]
        nl, nk = self.cfg.nlo, self.nk
        z, one = np.zeros((nk, nl)), np.ones((nk, nl))
        w0 = np.zeros((nk, self.r1 - self.r0 + 1, nl))
        bp = self._bwd(self._fwd(w0, one), z)   # backward image of the unit forward response
        q = self._bwd(w0, one)                  # unit backward response
        D = np.prod(-a * self.bet, axis=1)
        out = np.zeros((nl, nk, 4))
        for i, x in enumerate((D, bp[:, 0, :], bp.sum(axis=1), q.sum(axis=1))):
            out[:, :, i] = x.T
        return out

    def thomas_consts(self, dst):
        dst.numpy()[:] = self._consts().reshape(-1)

    def set_thomas_consts(self, gath):
        self.cgath = gath.numpy().copy()

    def constr(self):
        cfg, c = self.cfg, self.c
        nl = cfg.nlo
        xin = (self.wcot[:, None] * self.ksum).sum(axis=0) * cfg.dxo * cfg.dyo
        dpioc, dpiocp = self.scal[:nl - 1].copy(), self.scal[nl - 1:2 * (nl - 1)].copy()
        aient = np.zeros(nl - 1)
        aient[0] = self.xon[0]
        new = dpiocp - cfg.tdto * np.asarray(cfg.gpoc) * aient
        rhs = new - c["cdiffo"].T @ xin
        self.hclco = np.linalg.solve(c["cdhoc"], rhs)
        self.scal[:nl - 1] = new
        self.scal[nl - 1:2 * (nl - 1)] = dpioc

    def unpack(self, fuse_ocqbdy=True):
        cfg, c = self.cfg, self.c
        nl, nx = cfg.nlo, cfg.nxpo
        pm = np.zeros((nx, self.nyl, nl))
        pm[1:-1, :, :] = self.wrk
        G = self._G()
        notin = (G < 2) | (G > self.nyg - 1)
        pm[:, notin, :] = 0.0
        for m in range(1, nl):
            pm[:, :, m] = pm[:, :, m] + self.hclco[m - 1] * self.ochom[:, :, m - 1]
        pnew = self.pom.copy()
        own = slice(self.jlo - 1, self.jhi)
        for k in range(nl):
            acc = np.zeros((nx, self.nyl))
            for m in range(nl):
                acc = acc + c["ctm2loc"][m, k] * pm[:, :, m]
            pnew[:, own, k] = acc[:, own]
        self.pom = self.po
        self.po = pnew
        if fuse_ocqbdy:
            self._ocqbdy()

    def _ocqbdy(self):
        cfg, c = self.cfg, self.c
        nl = cfg.nlo
        dxom2 = 1.0 / (cfg.dxo * cfg.dxo)
        bcf = cfg.bccooc * dxom2 / (0.5 * cfg.bccooc + 1.0) / cfg.fnot
        F = cfg.fnot * c["amatoc"]
        G = self._G()
        p, q = self.po, self.qo
        for k in range(nl):
            if k == 0:
                ap = F[0, 0] * p[:, :, 0] + F[0, 1] * p[:, :, 1]
            elif k == nl - 1:
                ap = F[k, k - 1] * p[:, :, k - 1] + F[k, k] * p[:, :, k]
            else:
                ap = F[k, k - 1] * p[:, :, k - 1] + F[k, k] * p[:, :, k] + F[k, k + 1] * p[:, :, k + 1]
            dd = self.ddynoc if k == nl - 1 else 0.0
            for j in range(self.jlo - 1, self.jhi):
                by = cfg.beta * self.yporel[j]
                ddj = dd[:, j] if k == nl - 1 else 0.0
                if G[j] == 1:
                    q[:, j, k] = bcf * (p[:, j + 1, k] - p[:, j, k]) - ap[:, j] + by + ddj
                elif G[j] == self.nyg:
                    q[:, j, k] = bcf * (p[:, j - 1, k] - p[:, j, k]) - ap[:, j] + by + ddj
                else:
                    d0 = dd[0, j] if k == nl - 1 else 0.0
                    d1 = dd[-1, j] if k == nl - 1 else 0.0
                    q[0, j, k] = bcf * (p[1, j, k] - p[0, j, k]) - ap[0, j] + by + d0
                    q[-1, j, k] = bcf * (p[-2, j, k] - p[-1, j, k]) - ap[-1, j] + by + d1

    def halo_pack(self, to_lo, to_hi):
        nl, nx = self.cfg.nlo, self.cfg.nxpo
        for buf, prow, qrow in ((to_lo, self.jlo - 1, self.jlo - 1), (to_hi, self.jhi - 3, self.jhi - 1)):
            if buf is None:
                continue
            t = buf.numpy()
            t[:3 * nl * nx] = self.po[:, prow:prow + 3, :].transpose(2, 1, 0).reshape(-1)
            t[3 * nl * nx:] = self.qo[:, qrow, :].T.reshape(-1)

    def halo_unpack(self, from_lo, from_hi):
        nl, nx = self.cfg.nlo, self.cfg.nxpo
        for buf, prow, qrow in ((from_lo, self.jlo - 4, self.jlo - 2), (from_hi, self.jhi, self.jhi)):
            if buf is None:
                continue
            t = buf.numpy()
            self.po[:, prow:prow + 3, :] = t[:3 * nl * nx].reshape(nl, 3, nx).transpose(2, 1, 0)
            self.qo[:, qrow, :] = t[3 * nl * nx:].reshape(nl, nx).T

    def stage(self, n, a=None, b=None, c=None, flags=0):
        """Same stage split as qgcm_hip_slab_stage."""
        if n == 1:
            self.qgostep(); self.row_transform(0); self.thomas_phase(1, None, a)
        elif n == 2:
            self.thomas_phase(2, a, None); self.constr(); self.row_transform(1); self.unpack(True)
            if self.nranks > 1:
                self.halo_pack(b, c)
        else:
            if self.nranks > 1:
                self.halo_unpack(a, b)
            if flags & 1:
                self.lf_average()

    def lf_average(self):
        nl = self.cfg.nlo
        self.qo = 0.5 * (self.qo + self.qom)
        self.po = 0.5 * (self.po + self.pom)
        self.scal[:nl - 1] = 0.5 * (self.scal[:nl - 1] + self.scal[nl - 1:2 * (nl - 1)])
