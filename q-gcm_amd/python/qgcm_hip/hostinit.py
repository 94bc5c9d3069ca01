"""Start-up arithmetic the reference main program performs before the time loop,
restated with numpy for the host side (init-only, not on the per-step path):

  eigmod   src/eigmode.f:41-538      (A matrix, modes, Flierl normalisation)
  bd2oc    src/q-gcm.F:932-954       (tridiagonal diagonal per wavenumber)
  xintp    src/intsubs.f:78-133      (trapezoid area integral)
  constr   src/conhoms.F:93-193      (dpioc / dpiocp [+ cyclic line integrals])
  qcomp    src/vorsubs.F:49-138      (interior q from p)
  merqcy   src/vorsubs.F:142-239     (cyclic W/E edge q)
  ocqbdy   src/vorsubs.F:245-388     (boundary q; numpy twin used only at init)
  homsol   src/conhoms.F:376-641     (homogeneous solutions; the Helmholtz solves
                                      go through the HIP solver, see OceanModel)

The atmospheric channel (cfg.atmos) shares all of it with the cyclic ocean; where the
reference's atmosphere differs the functions branch on ``cfg.atmos``: eigmod without the Flierl normalisation
(eigmode.f:309), topography under layer 1 (last argument of qcomp, src/q-gcm.F:738-749), atqzbd
(src/vorsubs.F:396-480), dpiat = integral of pa(k)-pa(k+1) (src/conhoms.F:205-216), homsol on ypa
(src/conhoms.F:644-810).
"""
import numpy as np

PI = 3.14159265358979324
TWOPI = 6.28318530717958648


def eigmod(gpoc, hoc, fnot, atmos=False):
    """Returns amatoc, rdm2oc, ctl2moc(k,m), ctm2loc(m,k) as Fortran-ordered arrays.
    atmos: case 'Atmosphere' of the reference - no Flierl normalisation; the right eigenvectors keep the scaling
    LAPACK's DTREVC gives them (largest component of magnitude 1).  Their sign comes out of LAPACK's Schur vectors and
    is not restatable; positive at k = 1 matches the reference for the example parameters, and pa, qa depend on
    neither scale nor sign of a mode."""
    h = np.asarray(hoc, dtype=np.float64)
    g = np.asarray(gpoc, dtype=np.float64)
    nl = len(h)
    A = np.zeros((nl, nl), order="F")
    # eigmode.f:131-144
    A[0, 1] = -1.0 / (g[0] * h[0])
    A[0, 0] = -A[0, 1]
    for k in range(1, nl - 1):
        A[k, k - 1] = -1.0 / (g[k - 1] * h[k])
        A[k, k + 1] = -1.0 / (g[k] * h[k])
        A[k, k] = -A[k, k - 1] - A[k, k + 1]
    A[nl - 1, nl - 2] = -1.0 / (g[nl - 2] * h[nl - 1])
    A[nl - 1, nl - 1] = -A[nl - 1, nl - 2]
    # A = H^-1 T, T symmetric  =>  symmetric problem for H^1/2 A H^-1/2
    S = np.zeros((nl, nl))
    for k in range(nl):
        S[k, k] = A[k, k]
    for k in range(nl - 1):
        S[k, k + 1] = S[k + 1, k] = -1.0 / (g[k] * np.sqrt(h[k] * h[k + 1]))
    lam, V = np.linalg.eigh(S)
    order = np.argsort(np.abs(lam), kind="stable")  # eigmode.f:386-402
    htot = h.sum()
    ctl2m = np.zeros((nl, nl), order="F")
    ctm2l = np.zeros((nl, nl), order="F")
    rdm2 = np.zeros(nl)
    for m, im in enumerate(order):
        R = V[:, im] / np.sqrt(h)
        fl = np.sqrt(htot / np.sum(h * R * R))  # Flierl normalisation, eigmode.f:310-328
        if R[0] < 0:
            fl = -fl
        if atmos:
            fl = (-1.0 if R[0] < 0 else 1.0) / np.abs(R).max()
        R = fl * R
        LR = np.sum(h * R * R)
        ctl2m[:, m] = h * R / LR   # ctl2m(k,m) = cl2m(m,k)
        ctm2l[m, :] = R            # ctm2l(m,k) = cm2l(k,m)
        rdm2[m] = 0.0 if m == 0 else fnot * fnot * abs(lam[im])
    return A, rdm2, ctl2m, ctm2l


def bd2oc(cfg):
    nxt = cfg.nxto
    aoc = 1.0 / (cfg.dyo * cfg.dyo)
    dxom2 = 1.0 / (cfg.dxo * cfg.dxo)
    b = np.zeros(nxt)
    if cfg.cyclic:
        for i in range(2, nxt // 2 + 1):
            i1 = 2 * i - 1
            b[i1 - 2] = -2.0 * aoc + 2.0 * dxom2 * (np.cos((i - 1) * TWOPI / nxt) - 1.0)
            b[i1 - 1] = b[i1 - 2]
        b[0] = -2.0 * aoc
        b[nxt - 1] = -2.0 * aoc - 4.0 * dxom2
    else:
        for i in range(2, nxt + 1):
            b[i - 2] = -2.0 * aoc + 2.0 * dxom2 * (np.cos((i - 1) * PI / nxt) - 1.0)
        b[nxt - 1] = 0.0
    return aoc, b


def sponge_ramp(cfg):
    """r_spl(nxpo, nypo) of a -Dsponge_layer_k247 build (src/q-gcm.F:1154-1168): exp(-2 pi (d_y / l_spl)^2)
    [+ exp(-2 pi (d_x / l_spl)^2) unless nospl_in_ewbdy_k247] with d = half the (grid-point count x spacing) minus the
    distance of the 1-based index from it."""
    pi = 3.14159265358979324  # src/q-gcm.F:89

    def ramp(n, dx):
        idx = np.arange(1, n + 1, dtype=np.float64)
        d = 0.5 * dx * float(n) - np.abs(dx * idx - 0.5 * dx * float(n))
        return np.exp(-2.0 * pi * (d / cfg.l_spl) ** 2.0)
    rx = np.zeros(cfg.nxpo) if cfg.spl_ns_only else ramp(cfg.nxpo, cfg.dxo)  # nospl_in_ewbdy_k247: N-S boundaries only
    return np.asfortranarray(ramp(cfg.nypo, cfg.dyo)[None, :] + rx[:, None])


def xintp(v):
    """Area integral with weights 1 / 0.5 (edges) / 0.25 (corners), intsubs.f:78-133."""
    v = np.asarray(v)
    rows = 0.5 * v[0, 1:-1] + v[1:-1, 1:-1].sum(axis=0) + 0.5 * v[-1, 1:-1]
    xxs = 0.5 * v[0, 0] + v[1:-1, 0].sum() + 0.5 * v[-1, 0]
    xxn = 0.5 * v[0, -1] + v[1:-1, -1].sum() + 0.5 * v[-1, -1]
    return rows.sum() + 0.5 * (xxs + xxn)


def _topo_layer(cfg):
    """0-based layer that feels the topography: ocean nlo, atmosphere 1 (last argument of qcomp / merqcy)."""
    return 0 if getattr(cfg, "atmos", False) else cfg.nlo - 1


def _ap(A, p, k, fnot):
    nl = p.shape[2]
    if k == 0:
        return A[0, 0] * p[:, :, 0] + A[0, 1] * p[:, :, 1]
    if k == nl - 1:
        return A[k, k - 1] * p[:, :, k - 1] + A[k, k] * p[:, :, k]
    return A[k, k - 1] * p[:, :, k - 1] + A[k, k] * p[:, :, k] + A[k, k + 1] * p[:, :, k + 1]


def qcomp(cfg, A, yporel, ddynoc, p):
    """Interior q from p (vorsubs.F:49-138); boundaries left zero."""
    nl = cfg.nlo
    q = np.zeros_like(p, order="F")
    dx2fac = (1.0 / (cfg.dxo * cfg.dxo)) / cfg.fnot
    betay = cfg.beta * yporel[None, 1:-1]
    for k in range(nl):
        pk = p[:, :, k]
        lap = dx2fac * (pk[1:-1, :-2] + pk[:-2, 1:-1] + pk[2:, 1:-1] + pk[1:-1, 2:] - 4.0 * pk[1:-1, 1:-1]) + betay
        q[1:-1, 1:-1, k] = lap - cfg.fnot * _ap(A, p, k, cfg.fnot)[1:-1, 1:-1]
    q[1:-1, 1:-1, _topo_layer(cfg)] += ddynoc[1:-1, 1:-1]
    return q


def merqcy(cfg, A, yporel, ddynoc, p, q):
    """Periodic W/E edge q (vorsubs.F:142-239), in place."""
    nl = cfg.nlo
    dx2fac = (1.0 / (cfg.dxo * cfg.dxo)) / cfg.fnot
    betay = cfg.beta * yporel[1:-1]
    for k in range(nl):
        pk = p[:, :, k]
        lap = dx2fac * (pk[0, :-2] + pk[-2, 1:-1] + pk[1, 1:-1] + pk[0, 2:] - 4.0 * pk[0, 1:-1]) + betay
        q[0, 1:-1, k] = lap - cfg.fnot * _ap(A, p, k, cfg.fnot)[0, 1:-1]
    q[0, 1:-1, _topo_layer(cfg)] += ddynoc[0, 1:-1]
    q[-1, 1:-1, :] = q[0, 1:-1, :]


def ocqbdy(cfg, A, yporel, ddynoc, p, q):
    """Boundary q from p (vorsubs.F:245-388), in place; init-time numpy twin of k_ocqbdy."""
    nl = cfg.nlo
    dxom2 = 1.0 / (cfg.dxo * cfg.dxo)
    bcf = cfg.bccooc * dxom2 / (0.5 * cfg.bccooc + 1.0) / cfg.fnot
    F = cfg.fnot * np.asarray(A)  # f0Am/f0Ac/f0Ap are formed first, vorsubs.F:281-282
    atm = getattr(cfg, "atmos", False)
    for k in range(nl):
        ap = _ap(F, p, k, cfg.fnot)
        pk = p[:, :, k]
        aps = ap[:, 0]
        if atm and k == nl - 1:  # atqzbd, src/vorsubs.F:470: the southern value of the top layer reads row 2
            aps = F[k, k - 1] * p[:, 0, k - 1] + F[k, k] * p[:, 1, k]
        q[:, 0, k] = bcf * (pk[:, 1] - pk[:, 0]) - aps + cfg.beta * yporel[0]
        q[:, -1, k] = bcf * (pk[:, -2] - pk[:, -1]) - ap[:, -1] + cfg.beta * yporel[-1]
        if k == _topo_layer(cfg):
            q[:, 0, k] += ddynoc[:, 0]
            q[:, -1, k] += ddynoc[:, -1]
        if not cfg.cyclic:
            by = cfg.beta * yporel[1:-1]
            q[0, 1:-1, k] = bcf * (pk[1, 1:-1] - pk[0, 1:-1]) - ap[0, 1:-1] + by
            q[-1, 1:-1, k] = bcf * (pk[-2, 1:-1] - pk[-1, 1:-1]) - ap[-1, 1:-1] + by
            if k == nl - 1:
                q[0, 1:-1, k] += ddynoc[0, 1:-1]
                q[-1, 1:-1, k] += ddynoc[-1, 1:-1]


def q_from_p(cfg, A, yporel, ddynoc, p):
    """qcomp + ocqbdy (+ merqcy), the start-up sequence of src/q-gcm.F:719-731."""
    q = qcomp(cfg, A, yporel, ddynoc, p)
    ocqbdy(cfg, A, yporel, ddynoc, p, q)
    if cfg.cyclic:
        merqcy(cfg, A, yporel, ddynoc, p, q)
    return q


def constr(cfg, A, po, pom):
    """Constraint scalars in the layout of qgcm_hip_set_scalars (conhoms.F:93-193)."""
    nl = cfg.nlo
    s = np.zeros(2 * (nl - 1) + 4 * nl)
    dA = cfg.dxo * cfg.dyo
    sgn = -1.0 if getattr(cfg, "atmos", False) else 1.0  # dpiat integrates pa(k) - pa(k+1), conhoms.F:205-216
    for k in range(nl - 1):
        s[k] = xintp(sgn * (po[:, :, k + 1] - po[:, :, k])) * dA
        s[nl - 1 + k] = xintp(sgn * (pom[:, :, k + 1] - pom[:, :, k])) * dA
    if cfg.cyclic:
        o = 2 * (nl - 1)

        def line(v):  # trapezoid along x
            return 0.5 * v[0] + v[1:-1].sum() + 0.5 * v[-1]
        f = 0.5 * cfg.dyo * cfg.fnot * cfg.fnot
        for name, p, off_s, off_n in (("cur", po, o, o + nl), ("prev", pom, o + 2 * nl, o + 3 * nl)):
            pins = np.array([cfg.dxo * line(p[:, 0, k]) for k in range(nl)])
            pinn = np.array([cfg.dxo * line(p[:, -1, k]) for k in range(nl)])
            for k in range(nl):
                cs = line(p[:, 1, k] - p[:, 0, k]) * (cfg.dxo / cfg.dyo)
                cn = line(p[:, -1, k] - p[:, -2, k]) * (cfg.dxo / cfg.dyo)
                s[off_s + k] = -cs + f * float(A[k, :] @ pins)
                s[off_n + k] = cn + f * float(A[k, :] @ pinn)
    return s


def homsol_box(cfg, rdm2, ctm2l, bd2, helmholtz):
    """conhoms.F:544-641; `helmholtz(wrk, boc)` is the (HIP) Helmholtz solver."""
    nl, nx, ny = cfg.nlo, cfg.nxpo, cfg.nypo
    ochom = np.zeros((nx, ny, nl - 1), order="F")
    aipohs = np.zeros(nl - 1)
    for m in range(nl - 1):
        boc = bd2 - rdm2[m + 1]
        sol0 = helmholtz(np.ones((nx, ny), order="F"), boc)
        ochom[:, :, m] = 1.0 + rdm2[m + 1] * sol0
        aipohs[m] = xintp(ochom[:, :, m]) * cfg.dxo * cfg.dyo
    cdiffo = np.zeros((nl, nl - 1), order="F")
    cdhoc = np.zeros((nl - 1, nl - 1), order="F")
    for k in range(nl - 1):
        for m in range(nl):
            cdiffo[m, k] = ctm2l[m, k + 1] - ctm2l[m, k]
        for m in range(nl - 1):
            cdhoc[k, m] = (ctm2l[m + 1, k + 1] - ctm2l[m + 1, k]) * aipohs[m]
    return dict(ochom=ochom, aipohs=aipohs, cdiffo=cdiffo, cdhoc=cdhoc)


def homsol_cyc(cfg, rdm2, bd2, yporel, helmholtz):
    """conhoms.F:376-543."""
    nl, nx, ny = cfg.nlo, cfg.nxpo, cfg.nypo
    pbhoc = (ny - np.arange(1, ny + 1)) / float(ny - 1)
    out = dict(pbhoc=pbhoc, hbsioc=cfg.ylo / cfg.xlo, aipbho=0.5 * cfg.xlo * cfg.ylo,
               pch1oc=np.zeros((ny, nl - 1), order="F"), pch2oc=np.zeros((ny, nl - 1), order="F"),
               aipcho=np.zeros(nl - 1), hc1soc=np.zeros(nl - 1), hc2soc=np.zeros(nl - 1),
               hc1noc=np.zeros(nl - 1), hc2noc=np.zeros(nl - 1))
    dyo = cfg.dyo
    for m in range(nl - 1):
        rd = rdm2[m + 1]
        boc = bd2 - rd
        yy = cfg.ypa() if getattr(cfg, "atmos", False) else yporel  # conhoms.F:664-665 uses ypa itself
        l1 = (yy[-1] - yy) / cfg.ylo
        l2 = (yy - yy[0]) / cfg.ylo
        w1 = helmholtz(np.asfortranarray(np.broadcast_to(l1, (nx, ny))), boc)
        w2 = helmholtz(np.asfortranarray(np.broadcast_to(l2, (nx, ny))), boc)
        f1 = l1[None, :] + rd * w1
        f2 = l2[None, :] + rd * w2
        p1, p2 = f1[0, :].copy(), f2[0, :].copy()
        out["pch1oc"][:, m], out["pch2oc"][:, m] = p1, p2
        out["aipcho"][m] = 0.5 * (xintp(f1) + xintp(f2)) * cfg.dxo * cfg.dyo
        p1ys = cfg.xlo * (-(p1[1] - p1[0]) / dyo + 0.5 * dyo * rd * p1[0])
        p2ys = cfg.xlo * (-(p2[1] - p2[0]) / dyo + 0.5 * dyo * rd * p2[0])
        p1yn = cfg.xlo * ((p1[-1] - p1[-2]) / dyo + 0.5 * dyo * rd * p1[-1])
        p2yn = cfg.xlo * ((p2[-1] - p2[-2]) / dyo + 0.5 * dyo * rd * p2[-1])
        det = p1ys * p2yn - p2ys * p1yn
        out["hc1soc"][m], out["hc2soc"][m] = p1ys / det, p2ys / det
        out["hc1noc"][m], out["hc2noc"][m] = p1yn / det, p2yn / det
    return out


def helmholtz_cyc_column(cfg, rhs, boc):
    """hscyoc (src/ocisubs.F:520-618) for a right-hand side that does not depend on x - all homsol asks of the cyclic
    solver (conhoms.F:376-543): only the zonal-mean coefficient of every row is non-zero, so the solve is ONE
    tridiagonal system in y with the diagonal boc(1) (same recurrence as the reference's sweep; the row transforms
    reduce to the identity up to the rounding of nxto*c/nxto).  Used where no whole-domain handle exists (y-slabs)."""
    r = np.asarray(rhs, dtype=np.float64)
    col = r[0, :] if r.ndim == 2 else r
    ny = cfg.nypo
    aoc = 1.0 / (cfg.dyo * cfg.dyo)
    b = float(np.asarray(boc)[0])
    nr = ny - 2
    u = np.zeros(nr)
    gam = np.zeros(nr)
    betinv = 1.0 / b
    u[0] = col[1] * betinv
    for j in range(1, nr):
        gam[j] = aoc * betinv
        betinv = 1.0 / (b - aoc * gam[j])
        u[j] = (col[1 + j] - aoc * u[j - 1]) * betinv
    for j in range(nr - 2, -1, -1):
        u[j] = u[j] - gam[j + 1] * u[j + 1]
    out = np.zeros((cfg.nxpo, ny), order="F")
    out[:, 1:-1] = u[None, :]
    return out


def helmholtz_box_host(cfg, rhs, boc):
    """Init-only host solve of the box Helmholtz problem (same algorithm as hsbxoc,
    src/ocisubs.F:415-512: row DST-I, Thomas along y per wavenumber, row DST-I) with
    scipy's DST-I (identical definition to FFTPACK dsint).  Used to build the
    homogeneous solutions of the y-extended basins of the multi-GPU weak-scaling runs,
    whose global row count exceeds what one handle solves; never on the per-step path."""
    import scipy.fft
    nx, ny = cfg.nxpo, cfg.nypo
    aoc = 1.0 / (cfg.dyo * cfg.dyo)
    ftnorm = 0.5 / cfg.nxto
    w = scipy.fft.dst(np.asarray(rhs, dtype=np.float64)[1:-1, 1:-1], type=1, axis=0)  # (nk, ny-2)
    b = np.asarray(boc, dtype=np.float64)[:nx - 2]
    nr = ny - 2
    gam = np.zeros((nx - 2, nr))
    u = np.zeros((nx - 2, nr))
    betinv = 1.0 / b
    u[:, 0] = w[:, 0] * betinv
    for r in range(1, nr):
        gam[:, r] = aoc * betinv
        betinv = 1.0 / (b - aoc * gam[:, r])
        u[:, r] = (w[:, r] - aoc * u[:, r - 1]) * betinv
    for r in range(nr - 2, -1, -1):
        u[:, r] = u[:, r] - gam[:, r + 1] * u[:, r + 1]
    out = np.zeros((nx, ny), order="F")
    out[1:-1, 1:-1] = scipy.fft.dst(ftnorm * u, type=1, axis=0)
    return out
