"""TEST INFRASTRUCTURE ONLY - ctypes view of oracle/libqgcm_oracle.so (the C
restatement of the reference algorithm, oracle/qgcm_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (q-gcm_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libqgcm_oracle.so")


def build(force=False):
    src = os.path.join(HERE, "qgcm_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "libqgcm_oracle.so"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        L.qgo_create.restype = C.c_void_p
        L.qgo_create.argtypes = [C.c_int] * 4 + [C.c_double] * 6 + [dp] * 6
        L.qgo_create_atmos.restype = C.c_void_p
        L.qgo_create_atmos.argtypes = [C.c_int] * 3 + [C.c_double] * 5 + [dp] * 5
        L.qgo_get_bsums.argtypes = [C.c_void_p, dp]
        L.qgo_xintp.restype = C.c_double
        L.qgo_xintp.argtypes = [dp, C.c_int, C.c_int]
        for name in ("qgo_destroy", "qgo_qgostep", "qgo_ocinvq", "qgo_ocqbdy", "qgo_lf_average"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.qgo_steps.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.qgo_set_p.argtypes = [C.c_void_p, dp, dp]
        L.qgo_set_state.argtypes = [C.c_void_p, dp, dp, dp, dp]
        L.qgo_get_state.argtypes = [C.c_void_p, dp, dp, dp, dp]
        L.qgo_set_forcing.argtypes = [C.c_void_p, dp, dp, dp]
        L.qgo_set_cyc_forcing.argtypes = [C.c_void_p, C.c_double, C.c_double, dp, dp]
        L.qgo_set_sponge.argtypes = [C.c_void_p, dp, C.c_double]
        L.qgo_get_scalars.argtypes = [C.c_void_p, dp]
        L.qgo_set_scalars.argtypes = [C.c_void_p, dp]
        L.qgo_get_inv_diag.argtypes = [C.c_void_p, dp, dp]
        L.qgo_get_consts.argtypes = [C.c_void_p] + [dp] * 6
        L.qgo_get_homog.argtypes = [C.c_void_p, dp, dp]
        L.qgo_project.argtypes = [C.c_void_p, dp]
        L.qgo_helmholtz.argtypes = [C.c_void_p, dp, dp]
        L.qgo_dsint.argtypes = [C.c_int, dp]
        L.qgo_rfftf.argtypes = [C.c_int, dp]
        L.qgo_rfftb.argtypes = [C.c_int, dp]
        L.qgo_eigmod.argtypes = [C.c_int, dp, dp, C.c_double, dp, dp, dp, dp]
        L.qgo_eigmod_atmos.argtypes = [C.c_int, dp, dp, C.c_double, dp, dp, dp, dp]
        L.qgo_wekpo_from_tau.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, dp, dp]
        L.qgo_set_threads.argtypes = [C.c_int]
        L.qgo_oml_init.argtypes = [C.c_void_p] + [C.c_double] * 7 + [C.c_int, C.c_double, C.c_int, C.c_double]
        L.qgo_oml_set.argtypes = [C.c_void_p] + [dp] * 6
        L.qgo_oml_get.argtypes = [C.c_void_p] + [dp] * 4
        L.qgo_oml.argtypes = [C.c_void_p]
        L.qgo_steps_oml.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.qgo_valids.argtypes = [C.c_void_p, dp, dp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def set_threads(n):
    lib().qgo_set_threads(int(n))


def dsint(x):
    n = len(x)
    buf = np.zeros(n + 1)
    buf[:n] = x
    lib().qgo_dsint(n, _dp(buf))
    return buf[:n].copy()


def rfftf(x):
    buf = np.array(x, dtype=np.float64)
    lib().qgo_rfftf(len(buf), _dp(buf))
    return buf


def rfftb(x):
    buf = np.array(x, dtype=np.float64)
    lib().qgo_rfftb(len(buf), _dp(buf))
    return buf


def xintp(val):
    v = np.asfortranarray(val, dtype=np.float64)
    return lib().qgo_xintp(_dp(v), v.shape[0], v.shape[1])


def eigmod(gpr, h, fnot, atmos=False):
    nl = len(h)
    g = np.ascontiguousarray(gpr, dtype=np.float64)
    hh = np.ascontiguousarray(h, dtype=np.float64)
    amat = np.zeros((nl, nl), order="F")
    cl2m = np.zeros((nl, nl), order="F")
    cm2l = np.zeros((nl, nl), order="F")
    rdm2 = np.zeros(nl)
    (lib().qgo_eigmod_atmos if atmos else lib().qgo_eigmod)(nl, _dp(g), _dp(hh), fnot, _dp(amat), _dp(rdm2), _dp(cl2m), _dp(cm2l))
    return dict(amatoc=amat, rdm2oc=rdm2, ctl2moc=cl2m, ctm2loc=cm2l)


def wekpo_from_tau(tauxo, tauyo, cyclic, dxo, fnot):
    tx = np.asfortranarray(tauxo, dtype=np.float64)
    ty = np.asfortranarray(tauyo, dtype=np.float64)
    nx, ny = tx.shape
    wt = np.zeros((nx - 1, ny - 1), order="F")
    wp = np.zeros((nx, ny), order="F")
    lib().qgo_wekpo_from_tau(nx, ny, int(cyclic), dxo, fnot, _dp(tx), _dp(ty), _dp(wt), _dp(wp))
    return wt, wp


class Oracle:
    """One ocean configuration (same call surface as ref_binding.RefLib)."""

    def __init__(self, nx, ny, nl, cyclic, fnot, beta, dxo, dto, delek, bccooc,
                 ah2oc, ah4oc, hoc, gpoc, yporel, ddynoc=None):
        self.nx, self.ny, self.nl, self.cyclic = nx, ny, nl, int(cyclic)
        self.fnot, self.beta = fnot, beta
        a2 = np.ascontiguousarray(ah2oc, dtype=np.float64)
        a4 = np.ascontiguousarray(ah4oc, dtype=np.float64)
        h = np.ascontiguousarray(hoc, dtype=np.float64)
        g = np.ascontiguousarray(gpoc, dtype=np.float64)
        yp = np.ascontiguousarray(yporel, dtype=np.float64)
        dd = np.zeros((nx, ny), order="F") if ddynoc is None else np.asfortranarray(ddynoc, dtype=np.float64)
        self.L = lib()
        self.h = self.L.qgo_create(nx, ny, nl, int(cyclic), fnot, beta, dxo, dto, delek, bccooc,
                                   _dp(a2), _dp(a4), _dp(h), _dp(g), _dp(yp), _dp(dd))
        self.nscal = 2 * (nl - 1) + 4 * nl

    def close(self):
        if self.h:
            self.L.qgo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _f3(self):
        return np.zeros((self.nx, self.ny, self.nl), order="F")

    def set_p(self, po, pom):
        a = np.asfortranarray(po, dtype=np.float64)
        b = np.asfortranarray(pom, dtype=np.float64)
        self.L.qgo_set_p(self.h, _dp(a), _dp(b))

    def set_state(self, po, pom, qo, qom):
        a = [np.asfortranarray(x, dtype=np.float64) for x in (po, pom, qo, qom)]
        self.L.qgo_set_state(self.h, *[_dp(x) for x in a])

    def get_state(self):
        a = [self._f3() for _ in range(4)]
        self.L.qgo_get_state(self.h, *[_dp(x) for x in a])
        return a

    def set_forcing(self, wekpo, entoc=None, xon=None):
        w = np.asfortranarray(wekpo, dtype=np.float64)
        e = np.zeros((self.nx, self.ny), order="F") if entoc is None else np.asfortranarray(entoc, dtype=np.float64)
        x = np.zeros(self.nl - 1) if xon is None else np.ascontiguousarray(xon, dtype=np.float64)
        self.L.qgo_set_forcing(self.h, _dp(w), _dp(e), _dp(x))

    def set_cyc_forcing(self, txis, txin, enis=None, enin=None):
        es = np.zeros(self.nl - 1) if enis is None else np.ascontiguousarray(enis, dtype=np.float64)
        en = np.zeros(self.nl - 1) if enin is None else np.ascontiguousarray(enin, dtype=np.float64)
        self.L.qgo_set_cyc_forcing(self.h, txis, txin, _dp(es), _dp(en))

    def set_sponge(self, r_spl, c1_spl):
        if r_spl is None:
            self.L.qgo_set_sponge(self.h, None, 0.0)
            return
        r = np.asfortranarray(r_spl, dtype=np.float64)
        self.L.qgo_set_sponge(self.h, _dp(r), float(c1_spl))

    def get_scalars(self):
        s = np.zeros(self.nscal)
        self.L.qgo_get_scalars(self.h, _dp(s))
        return s

    def set_scalars(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64)
        self.L.qgo_set_scalars(self.h, _dp(s))

    def get_inv_diag(self):
        x = np.zeros(self.nl)
        c = np.zeros(2 * self.nl + 1)
        self.L.qgo_get_inv_diag(self.h, _dp(x), _dp(c))
        n = 2 * (self.nl - 1) + 1 if self.cyclic else self.nl - 1
        return x, c[:n].copy()

    def get_consts(self):
        nl = self.nl
        amat = np.zeros((nl, nl), order="F")
        cl2m = np.zeros((nl, nl), order="F")
        cm2l = np.zeros((nl, nl), order="F")
        rdm2 = np.zeros(nl)
        bd2 = np.zeros(self.nx - 1)
        aoc = C.c_double()
        self.L.qgo_get_consts(self.h, _dp(amat), _dp(cl2m), _dp(cm2l), _dp(rdm2), _dp(bd2), C.cast(C.byref(aoc), C.POINTER(C.c_double)))
        return dict(amatoc=amat, ctl2moc=cl2m, ctm2loc=cm2l, rdm2oc=rdm2, bd2oc=bd2, aoc=aoc.value)

    def get_homog(self):
        nl, nx, ny = self.nl, self.nx, self.ny
        if self.cyclic:
            hom = np.zeros(ny * (2 * (nl - 1) + 1))
            aux = np.zeros(5 * (nl - 1) + 2)
            self.L.qgo_get_homog(self.h, _dp(hom), _dp(aux))
            n1 = ny * (nl - 1)
            o = nl - 1
            return dict(pch1oc=hom[:n1].reshape((ny, o), order="F").copy(order="F"),
                        pch2oc=hom[n1:2 * n1].reshape((ny, o), order="F").copy(order="F"),
                        pbhoc=hom[2 * n1:].copy(), aipcho=aux[0:o].copy(), hc1soc=aux[o:2 * o].copy(),
                        hc2soc=aux[2 * o:3 * o].copy(), hc1noc=aux[3 * o:4 * o].copy(),
                        hc2noc=aux[4 * o:5 * o].copy(), hbsioc=aux[5 * o], aipbho=aux[5 * o + 1])
        hom = np.zeros(nx * ny * (nl - 1))
        aux = np.zeros((nl - 1) + nl * (nl - 1) + (nl - 1) ** 2)
        self.L.qgo_get_homog(self.h, _dp(hom), _dp(aux))
        o = nl - 1
        return dict(ochom=hom.reshape((nx, ny, o), order="F").copy(order="F"), aipohs=aux[:o].copy(),
                    cdiffo=aux[o:o + nl * o].reshape((nl, o), order="F").copy(order="F"),
                    cdhoc=aux[o + nl * o:].reshape((o, o), order="F").copy(order="F"))

    def qgostep(self):
        self.L.qgo_qgostep(self.h)

    def ocinvq(self):
        self.L.qgo_ocinvq(self.h)

    def ocqbdy(self):
        self.L.qgo_ocqbdy(self.h)

    def lf_average(self):
        self.L.qgo_lf_average(self.h)

    def steps(self, s0, n):
        self.L.qgo_steps(self.h, int(s0), int(n))

    # -- ocean mixed layer (src/omlsubs.F), SURVEY 8 row f1 -------------------
    def oml_init(self, hmoc, toc1, toc2, st2d, st4d, ycexp, rrcpoc, sb_hflux=0, tsbdy=0.0, nb_hflux=0, tnbdy=0.0):
        self.L.qgo_oml_init(self.h, hmoc, toc1, toc2, st2d, st4d, ycexp, rrcpoc, int(sb_hflux), tsbdy, int(nb_hflux), tnbdy)

    def oml_set(self, sst, sstm, fnetoc, wekto, tauxo, tauyo):
        a = [np.asfortranarray(x, dtype=np.float64) for x in (sst, sstm, fnetoc, wekto, tauxo, tauyo)]
        assert a[0].shape == (self.nx - 1, self.ny - 1) and a[4].shape == (self.nx, self.ny)
        self.L.qgo_oml_set(self.h, *[_dp(x) for x in a])

    def oml_get(self):
        sst = np.zeros((self.nx - 1, self.ny - 1), order="F")
        sstm = np.zeros_like(sst)
        ent = np.zeros((self.nx, self.ny), order="F")
        scal = np.zeros(5)
        self.L.qgo_oml_get(self.h, _dp(sst), _dp(sstm), _dp(ent), _dp(scal))
        return sst, sstm, ent, scal

    def oml(self):
        self.L.qgo_oml(self.h)

    def steps_oml(self, s0, n):
        self.L.qgo_steps_oml(self.h, int(s0), int(n))

    def valids(self, dtopoc=None):
        """(solnok, out) of the ocean part of valids (src/valsubs.F:272-527)."""
        out = np.zeros(14 + self.nl)
        d = None if dtopoc is None else np.asfortranarray(dtopoc, dtype=np.float64)
        ok = self.L.qgo_valids(self.h, None if d is None else _dp(d), _dp(out))
        return bool(ok), out

    def project(self):
        w = self._f3()
        self.L.qgo_project(self.h, _dp(w))
        return w

    def helmholtz(self, wrk, boc):
        w = np.asfortranarray(wrk, dtype=np.float64).copy(order="F")
        b = np.ascontiguousarray(boc, dtype=np.float64)
        self.L.qgo_helmholtz(self.h, _dp(w), _dp(b))
        return w


class AtmosOracle(Oracle):
    """The atmospheric channel qgastep / atinvq / atqzbd - same call surface as
    ref_binding.RefAtmos; the inherited names map as qgostep = qgastep, ocinvq = atinvq, ocqbdy = atqzbd."""

    def __init__(self, nx, ny, nl, fnot, beta, dxa, dta, bccoat, ah4at, hat, gpat, yparel, ddynat=None):
        self.nx, self.ny, self.nl, self.cyclic = nx, ny, nl, 1
        self.fnot, self.beta = fnot, beta
        a4 = np.ascontiguousarray(ah4at, dtype=np.float64)
        h = np.ascontiguousarray(hat, dtype=np.float64)
        g = np.ascontiguousarray(gpat, dtype=np.float64)
        yp = np.ascontiguousarray(yparel, dtype=np.float64)
        dd = np.zeros((nx, ny), order="F") if ddynat is None else np.asfortranarray(ddynat, dtype=np.float64)
        self.L = lib()
        self.h = self.L.qgo_create_atmos(nx, ny, nl, fnot, beta, dxa, dta, bccoat, _dp(a4), _dp(h), _dp(g), _dp(yp), _dp(dd))
        self.nscal = 2 * (nl - 1) + 4 * nl

    def set_forcing(self, wekpa, entat=None, xan=None, txis=0.0, txin=0.0, enis=None, enin=None):
        Oracle.set_forcing(self, wekpa, entat, xan)
        self.set_cyc_forcing(txis, txin, enis, enin)

    def get_bsums(self):
        b = np.zeros(4 * self.nl)
        self.L.qgo_get_bsums(self.h, _dp(b))
        return b

    qgastep = Oracle.qgostep
    atinvq = Oracle.ocinvq
    atqzbd = Oracle.ocqbdy
