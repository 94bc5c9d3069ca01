"""TEST INFRASTRUCTURE ONLY (oracle) - ctypes view of oracle/_ref/libqgcm_ref_<cfg>.so.

The library is the *true reference* (jinkakei/q-gcm Fortran + FFTPACK) compiled
by oracle/build_ref.sh plus our harness oracle/ref/qgcm_ref_harness.F90.  It is
used (a) in this container to generate tests/golden/ and to validate the C
restatement oracle/qgcm_oracle.c, and (b) by bench.py's ``cpu_baseline`` leg
(kind "reference").  Nothing in the product path may import this module.

All arrays are Fortran ordered float64, shapes (nxpo, nypo, nlo) etc.
"""
import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFDIR = os.path.join(HERE, "_ref")

# name -> (nxta, nyta, nxaooc, nyaooc, ndxr, nlo, fnot, beta, cyclic)
# box_natl5 / cyc_socn5 are exactly examples/double_gyre_ocean_only/parameters_data.F.dg_oo
# and examples/southern_ocean_ocean_only/parameters_data.F.so_oo.
CONFIGS = {
    "box_tiny": (8, 8, 4, 3, 12, 3, "9.37456D-05", "1.75360D-11", 0),
    "box_small": (12, 10, 6, 5, 16, 3, "9.37456D-05", "1.75360D-11", 0),
    "box_tiny2": (8, 8, 5, 4, 6, 2, "5.92D-05", "2.08D-11", 0),
    # more layers than the examples carry (nlo is a compile-time PARAMETER of the reference, src/parameters_data.F:41,54)
    "box_tiny5": (8, 8, 4, 3, 12, 5, "9.37456D-05", "1.75360D-11", 0),
    "cyc_tiny6": (4, 8, "nxta", 3, 12, 6, "-1.19467D-04", "1.31301D-11", 1),
    # box_tiny compiled with the specified-temperature southern boundary of the mixed layer (-Dsb_hflux,
    # as examples/double_gyre_coupled; src/omlsubs.F:405-422)
    "box_tiny_sb": (8, 8, 4, 3, 12, 3, "9.37456D-05", "1.75360D-11", 0, "-Dsb_hflux"),
    # the tiny grids compiled with the fork's sponge layer (-Dsponge_layer_k247: src/qgosubs.F:203-205, ramp
    # src/q-gcm.F:1154-1168, constants src/parameters_data.F:140-144)
    "box_tiny_spl": (8, 8, 4, 3, 12, 3, "9.37456D-05", "1.75360D-11", 0, "-Dsponge_layer_k247"),
    # (the channel with nospl_in_ewbdy_k247, "N-S boundary only": the full ramp is not periodic in x - its values at
    #  i = 1 and i = nxpo differ - and the reference, which steps the duplicate column nxpo by itself, would leave
    #  qo(nxpo,j) != qo(1,j))
    "cyc_tiny_spl": (4, 8, "nxta", 3, 12, 3, "-1.19467D-04", "1.31301D-11", 1, "-Dsponge_layer_k247 -Dnospl_in_ewbdy_k247"),
    "cyc_tiny": (4, 8, "nxta", 3, 12, 3, "-1.19467D-04", "1.31301D-11", 1),
    "cyc_small": (6, 10, "nxta", 4, 16, 3, "-1.19467D-04", "1.31301D-11", 1),
    "box_natl5": (384, 96, 60, 60, 16, 3, "9.37456D-05", "1.75360D-11", 0),
    "cyc_socn5": (288, 108, "nxta", 36, 16, 3, "-1.19467D-04", "1.31301D-11", 1),
    # NAtl 1 km (BASELINE configs[4]): the grid lines of src/parameters_data.F.NAtl.1km:45,50 (40 km atmosphere,
    # nxaooc = nyaooc = 120, ndxr = 40 -> 4801 x 4801 x 3 p-grid); used by tests/golden/make_golden_fullsize.py only
    "box_natl1": (768, 192, 120, 120, 40, 3, "9.37456D-05", "1.75360D-11", 0),
    # coupled builds (no -Docean_only; -Dsb_hflux as examples/double_gyre_coupled/make.config.coupled): the ocean
    # path as above + the atmosphere path qgastep/atinvq/atqzbd.  Atmosphere (nxta+1, nyta+1, 3):
    # cpl_tiny 17x13 over the box_tiny ocean, cpl_small 33x21, cpl_natl5 = examples/double_gyre_coupled (385x97).
    "cpl_tiny": (16, 12, 4, 3, 12, 3, "9.37456D-05", "1.75360D-11", 0, "-Dsb_hflux", 1),
    "cpl_small": (32, 20, 6, 5, 16, 3, "9.37456D-05", "1.75360D-11", 0, "-Dsb_hflux", 1),
    "cpl_natl5": (384, 96, 60, 60, 16, 3, "9.37456D-05", "1.75360D-11", 0, "-Dsb_hflux", 1),
}


_LOADED_CFG = None
_OMP_THREADS = None  # set_threads(): thread count for the reference's OpenMP regions (None = runtime default)


def set_threads(n):
    global _OMP_THREADS
    _OMP_THREADS = n


def lib_path(cfg):
    return os.path.join(REFDIR, "libqgcm_ref_%s.so" % cfg)


def build(cfg, force=False):
    """Compile the reference for one config (no-op when /root/reference is absent)."""
    import subprocess
    if os.path.exists(lib_path(cfg)) and not force:
        return lib_path(cfg)
    p = CONFIGS[cfg]
    env = dict(os.environ)
    if cfg == "box_natl1":
        # > 2 GB of static module arrays (po, pom, qo, qom alone are 4 x 553 MB): the medium code model keeps them in
        # .lbss, out of the +-2 GB reach the small-model flang run-time objects need for their own data
        env["FC"] = env.get("FC", "/opt/rocm/bin/amdflang") + " -mcmodel=medium"
    subprocess.check_call([os.path.join(HERE, "build_ref.sh"), cfg] + [str(x) for x in p], env=env)
    return lib_path(cfg)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def run_big_stack(fn, *args):
    """The reference keeps 22 MB automatic arrays (src/qgosubs.F:64-66): call it on
    a thread with a large stack instead of requiring ``ulimit -s unlimited``."""
    out = {}

    def tgt():
        try:
            if _OMP_THREADS is not None:
                # OpenMP's num-threads setting is per initial thread: it has to be made on THIS thread
                for name in ("libomp.so", "/opt/rocm/lib/llvm/lib/libomp.so"):
                    try:
                        C.CDLL(name).omp_set_num_threads(int(_OMP_THREADS))
                        break
                    except OSError:
                        continue
            out["r"] = fn(*args)
        except BaseException as e:  # pragma: no cover
            out["e"] = e
    old = threading.stack_size(1 << 30)
    try:
        t = threading.Thread(target=tgt)
        t.start()
        t.join()
    finally:
        threading.stack_size(old)
    if "e" in out:
        raise out["e"]
    return out.get("r")


class RefLib:
    def __init__(self, cfg):
        os.environ.setdefault("OMP_STACKSIZE", "1G")
        path = lib_path(cfg)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.cfg = cfg
        # RTLD_GLOBAL: MKL dlopens its CPU-specific kernels against libmkl_core symbols.
        # Consequence: every config exports the same Fortran module symbols (with
        # different array sizes), so only ONE reference config may live in a process.
        global _LOADED_CFG
        if _LOADED_CFG not in (None, cfg):
            raise RuntimeError("reference config %s already loaded in this process; use a subprocess for %s"
                               % (_LOADED_CFG, cfg))
        _LOADED_CFG = cfg
        self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        nx, ny, nl, cyc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.lib.ref_dims(C.byref(nx), C.byref(ny), C.byref(nl), C.byref(cyc))
        self.nx, self.ny, self.nl, self.cyclic = nx.value, ny.value, nl.value, cyc.value
        f0, b = C.c_double(), C.c_double()
        self.lib.ref_params(C.byref(f0), C.byref(b))
        self.fnot, self.beta = f0.value, b.value
        self.nscal = 2 * (self.nl - 1) + 4 * self.nl

    # -- helpers -------------------------------------------------------
    def _f3(self):
        return np.zeros((self.nx, self.ny, self.nl), order="F")

    def _f2(self):
        return np.zeros((self.nx, self.ny), order="F")

    def init(self, dxo, dto, delek, bccooc, ah2oc, ah4oc, hoc, gpoc, ddynoc=None):
        ah2 = np.ascontiguousarray(ah2oc, dtype=np.float64)
        ah4 = np.ascontiguousarray(ah4oc, dtype=np.float64)
        h = np.ascontiguousarray(hoc, dtype=np.float64)
        g = np.ascontiguousarray(gpoc, dtype=np.float64)
        dd = self._f2() if ddynoc is None else np.asfortranarray(ddynoc, dtype=np.float64)
        self.lib.ref_init.argtypes = [C.c_double] * 4 + [C.POINTER(C.c_double)] * 5
        run_big_stack(self.lib.ref_init, dxo, dto, delek, bccooc, _dp(ah2), _dp(ah4), _dp(h), _dp(g), _dp(dd))

    def set_p(self, po, pom):
        po = np.asfortranarray(po, dtype=np.float64)
        pom = np.asfortranarray(pom, dtype=np.float64)
        run_big_stack(self.lib.ref_set_p, _dp(po), _dp(pom))

    def set_state(self, po, pom, qo, qom):
        a = [np.asfortranarray(x, dtype=np.float64) for x in (po, pom, qo, qom)]
        self.lib.ref_set_state(*[_dp(x) for x in a])

    def get_state(self):
        a = [self._f3() for _ in range(4)]
        self.lib.ref_get_state(*[_dp(x) for x in a])
        return a

    def set_forcing(self, wekpo, entoc=None, xon=None):
        w = np.asfortranarray(wekpo, dtype=np.float64)
        e = self._f2() if entoc is None else np.asfortranarray(entoc, dtype=np.float64)
        x = np.zeros(self.nl - 1) if xon is None else np.ascontiguousarray(xon, dtype=np.float64)
        self.lib.ref_set_forcing(_dp(w), _dp(e), _dp(x))

    # -- ocean mixed layer (src/omlsubs.F), SURVEY 8 row f1 -------------------
    def oml_flags(self):
        sb, nb = C.c_int(), C.c_int()
        self.lib.ref_oml_flags(C.byref(sb), C.byref(nb))
        return sb.value, nb.value

    def oml_init(self, hmoc, toc1, toc2, st2d, st4d, ycexp, rrcpoc, tsbdy, tnbdy):
        self.lib.ref_oml_init.argtypes = [C.c_double] * 9
        self.lib.ref_oml_init(hmoc, toc1, toc2, st2d, st4d, ycexp, rrcpoc, tsbdy, tnbdy)

    def oml_set(self, sst, sstm, fnetoc, wekto, tauxo, tauyo):
        a = [np.asfortranarray(x, dtype=np.float64) for x in (sst, sstm, fnetoc, wekto, tauxo, tauyo)]
        assert a[0].shape == (self.nx - 1, self.ny - 1) and a[4].shape == (self.nx, self.ny)
        self.lib.ref_oml_set(*[_dp(x) for x in a])

    def oml_get(self):
        """sst, sstm (T grid), entoc (p grid), (xon(1), cfraoc, centoc, enisoc(1), eninoc(1))."""
        sst = np.zeros((self.nx - 1, self.ny - 1), order="F")
        sstm = np.zeros_like(sst)
        ent = self._f2()
        scal = np.zeros(5)
        self.lib.ref_oml_get(_dp(sst), _dp(sstm), _dp(ent), _dp(scal))
        return sst, sstm, ent, scal

    def oml(self):
        run_big_stack(self.lib.ref_oml)

    def steps_oml(self, s0, n):
        self.lib.ref_steps_oml.argtypes = [C.c_int, C.c_int]
        run_big_stack(self.lib.ref_steps_oml, int(s0), int(n))

    def write_restart(self, path, tyrs):
        """Unformatted restart dump in the reference's record sequence (src/q-gcm.F:3076-3086)."""
        b = path.encode()
        self.lib.ref_write_restart.argtypes = [C.c_char_p, C.c_int, C.c_double]
        self.lib.ref_write_restart(b, len(b), float(tyrs))

    def atmos_dims(self):
        a, b = C.c_int(), C.c_int()
        self.lib.ref_atmos_dims(C.byref(a), C.byref(b))
        return a.value, b.value

    def valids(self, dtopoc=None):
        """The reference's verdict solnok (src/valsubs.F:43); its prints go to stdout."""
        d = self._f2() if dtopoc is None else np.asfortranarray(dtopoc, dtype=np.float64)
        ok = C.c_int()
        run_big_stack(self.lib.ref_valids, C.byref(ok), _dp(d))
        return bool(ok.value)

    def get_sponge(self):
        """(on, r_spl(nxpo, nypo), c1_spl, l_spl) of the build (on = False without -Dsponge_layer_k247)."""
        on, c1, ls = C.c_int(), C.c_double(), C.c_double()
        r = self._f2()
        self.lib.ref_get_sponge(C.byref(on), _dp(r), C.byref(c1), C.byref(ls))
        return bool(on.value), r, c1.value, ls.value

    def set_cyc_forcing(self, txis, txin, enis=None, enin=None):
        es = np.zeros(self.nl - 1) if enis is None else np.ascontiguousarray(enis, dtype=np.float64)
        en = np.zeros(self.nl - 1) if enin is None else np.ascontiguousarray(enin, dtype=np.float64)
        self.lib.ref_set_cyc_forcing.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        self.lib.ref_set_cyc_forcing(txis, txin, _dp(es), _dp(en))

    def get_scalars(self):
        s = np.zeros(self.nscal)
        self.lib.ref_get_scalars(_dp(s))
        return s

    def set_scalars(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64)
        self.lib.ref_set_scalars(_dp(s))

    def get_monitors(self):
        """(ermaso, emfroc) of the last ocinvq (src/ocisubs.F:268-283; zeros in a box build)."""
        e, f = np.zeros(self.nl - 1), np.zeros(self.nl - 1)
        self.lib.ref_get_monitors(_dp(e), _dp(f))
        return e, f

    def get_consts(self):
        nl = self.nl
        amat = np.zeros((nl, nl), order="F")
        cl2m = np.zeros((nl, nl), order="F")
        cm2l = np.zeros((nl, nl), order="F")
        rdm2 = np.zeros(nl)
        bd2 = np.zeros(self.nx - 1)
        ypr = np.zeros(self.ny)
        aoc = C.c_double()
        self.lib.ref_get_consts(_dp(amat), _dp(cl2m), _dp(cm2l), _dp(rdm2), _dp(bd2), _dp(ypr), C.byref(aoc))
        return dict(amatoc=amat, ctl2moc=cl2m, ctm2loc=cm2l, rdm2oc=rdm2, bd2oc=bd2, yporel=ypr, aoc=aoc.value)

    def get_homog(self):
        nl, nx, ny = self.nl, self.nx, self.ny
        if self.cyclic:
            hom = np.zeros(ny * (2 * (nl - 1) + 1))
            aux = np.zeros(5 * (nl - 1) + 2)
            self.lib.ref_get_homog(_dp(hom), _dp(aux))
            n1 = ny * (nl - 1)
            return dict(pch1oc=hom[:n1].reshape((ny, nl - 1), order="F").copy(order="F"),
                        pch2oc=hom[n1:2 * n1].reshape((ny, nl - 1), order="F").copy(order="F"),
                        pbhoc=hom[2 * n1:].copy(),
                        aipcho=aux[0:nl - 1].copy(), hc1soc=aux[nl - 1:2 * (nl - 1)].copy(),
                        hc2soc=aux[2 * (nl - 1):3 * (nl - 1)].copy(), hc1noc=aux[3 * (nl - 1):4 * (nl - 1)].copy(),
                        hc2noc=aux[4 * (nl - 1):5 * (nl - 1)].copy(), hbsioc=aux[5 * (nl - 1)], aipbho=aux[5 * (nl - 1) + 1])
        hom = np.zeros(nx * ny * (nl - 1))
        aux = np.zeros((nl - 1) + nl * (nl - 1) + (nl - 1) ** 2)
        self.lib.ref_get_homog(_dp(hom), _dp(aux))
        o = nl - 1
        return dict(ochom=hom.reshape((nx, ny, nl - 1), order="F").copy(order="F"),
                    aipohs=aux[:o].copy(),
                    cdiffo=aux[o:o + nl * (nl - 1)].reshape((nl, nl - 1), order="F").copy(order="F"),
                    cdhoc=aux[o + nl * (nl - 1):].reshape((nl - 1, nl - 1), order="F").copy(order="F"))

    def qgostep(self):
        run_big_stack(self.lib.ref_qgostep)

    def ocinvq(self):
        run_big_stack(self.lib.ref_ocinvq)

    def ocqbdy(self):
        run_big_stack(self.lib.ref_ocqbdy)

    def lf_average(self):
        self.lib.ref_lf_average()

    def steps(self, s0, n):
        self.lib.ref_steps.argtypes = [C.c_int, C.c_int]
        run_big_stack(self.lib.ref_steps, int(s0), int(n))

    def helmholtz(self, wrk, boc):
        w = np.asfortranarray(wrk, dtype=np.float64).copy(order="F")
        b = np.ascontiguousarray(boc, dtype=np.float64)
        run_big_stack(self.lib.ref_helmholtz, _dp(w), _dp(b))
        return w

    def xintp(self, val):
        v = np.asfortranarray(val, dtype=np.float64)
        r = C.c_double()
        self.lib.ref_xintp(_dp(v), C.byref(r))
        return r.value

    def dsint(self, x):
        n = len(x)
        buf = np.zeros(n + 1)
        buf[:n] = x
        self.lib.ref_dsint.argtypes = [C.c_int, C.POINTER(C.c_double)]
        self.lib.ref_dsint(n, _dp(buf))
        return buf[:n].copy()

    def drfft(self, x, dirn=+1):
        buf = np.array(x, dtype=np.float64)
        self.lib.ref_drfft.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_int]
        self.lib.ref_drfft(len(buf), _dp(buf), dirn)
        return buf

    def eigmod(self, gpr, h):
        nl = len(h)
        g = np.ascontiguousarray(gpr, dtype=np.float64)
        hh = np.ascontiguousarray(h, dtype=np.float64)
        amat = np.zeros((nl, nl), order="F")
        cl2m = np.zeros((nl, nl), order="F")
        cm2l = np.zeros((nl, nl), order="F")
        rdm2 = np.zeros(nl)
        self.lib.ref_eigmod.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 6
        self.lib.ref_eigmod(nl, _dp(g), _dp(hh), _dp(amat), _dp(rdm2), _dp(cl2m), _dp(cm2l))
        return dict(amatoc=amat, rdm2oc=rdm2, ctl2moc=cl2m, ctm2loc=cm2l)


class RefAtmos:
    """Atmosphere half (qgastep / atinvq / atqzbd) of a *coupled* reference build
    (CONFIGS "cpl_*"); shares the process-wide library with RefLib(cfg).  Arrays (nxpa, nypa, nla)."""

    def __init__(self, ref):
        self.ref = ref
        self.lib = ref.lib
        nx, ny, nl = C.c_int(), C.c_int(), C.c_int()
        self.lib.ref_atm_dims(C.byref(nx), C.byref(ny), C.byref(nl))
        self.nx, self.ny, self.nl = nx.value, ny.value, nl.value
        self.fnot, self.beta = ref.fnot, ref.beta
        self.nscal = 2 * (self.nl - 1) + 4 * self.nl

    def _f3(self):
        return np.zeros((self.nx, self.ny, self.nl), order="F")

    def _f2(self):
        return np.zeros((self.nx, self.ny), order="F")

    def init(self, dxa, dta, bccoat, ah4at, hat, gpat, ddynat=None):
        a4 = np.ascontiguousarray(ah4at, dtype=np.float64)
        h = np.ascontiguousarray(hat, dtype=np.float64)
        g = np.ascontiguousarray(gpat, dtype=np.float64)
        dd = self._f2() if ddynat is None else np.asfortranarray(ddynat, dtype=np.float64)
        self.lib.ref_atm_init.argtypes = [C.c_double] * 3 + [C.POINTER(C.c_double)] * 4
        run_big_stack(self.lib.ref_atm_init, dxa, dta, bccoat, _dp(a4), _dp(h), _dp(g), _dp(dd))

    def homsol(self):
        """homsol of the coupled build: ocean AND atmosphere (both must be initialised)."""
        run_big_stack(self.lib.ref_homsol)

    def set_p(self, pa, pam):
        pa = np.asfortranarray(pa, dtype=np.float64)
        pam = np.asfortranarray(pam, dtype=np.float64)
        run_big_stack(self.lib.ref_atm_set_p, _dp(pa), _dp(pam))

    def set_state(self, pa, pam, qa, qam):
        a = [np.asfortranarray(x, dtype=np.float64) for x in (pa, pam, qa, qam)]
        self.lib.ref_atm_set_state(*[_dp(x) for x in a])

    def get_state(self):
        a = [self._f3() for _ in range(4)]
        self.lib.ref_atm_get_state(*[_dp(x) for x in a])
        return a

    def set_forcing(self, wekpa, entat=None, xan=None, txis=0.0, txin=0.0, enis=None, enin=None):
        w = np.asfortranarray(wekpa, dtype=np.float64)
        e = self._f2() if entat is None else np.asfortranarray(entat, dtype=np.float64)

        def z(v):
            return np.zeros(self.nl - 1) if v is None else np.ascontiguousarray(v, dtype=np.float64)
        x, es, en = z(xan), z(enis), z(enin)
        self.lib.ref_atm_set_forcing.argtypes = ([C.POINTER(C.c_double)] * 3 + [C.c_double] * 2 +
                                                 [C.POINTER(C.c_double)] * 2)
        self.lib.ref_atm_set_forcing(_dp(w), _dp(e), _dp(x), float(txis), float(txin), _dp(es), _dp(en))

    def get_scalars(self):
        s = np.zeros(self.nscal)
        self.lib.ref_atm_get_scalars(_dp(s))
        return s

    def set_scalars(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64)
        self.lib.ref_atm_set_scalars(_dp(s))

    def get_monitors(self):
        """(ermasa, emfrat) of the last atinvq (MODULE monitor; src/atisubs.F:236-248)."""
        e, f = np.zeros(self.nl - 1), np.zeros(self.nl - 1)
        self.lib.ref_atm_get_monitors(_dp(e), _dp(f))
        return e, f

    def get_bsums(self):
        """ajisat, ajinat, ap5sat, ap5nat (nla each) of the last qgastep."""
        b = np.zeros(4 * self.nl)
        self.lib.ref_atm_get_bsums(_dp(b))
        return b

    def get_consts(self):
        nl = self.nl
        amat = np.zeros((nl, nl), order="F")
        cl2m = np.zeros((nl, nl), order="F")
        cm2l = np.zeros((nl, nl), order="F")
        rdm2 = np.zeros(nl)
        bd2 = np.zeros(self.nx - 1)
        ypr = np.zeros(self.ny)
        aat = C.c_double()
        self.lib.ref_atm_get_consts(_dp(amat), _dp(cl2m), _dp(cm2l), _dp(rdm2), _dp(bd2), _dp(ypr), C.byref(aat))
        return dict(amatat=amat, ctl2mat=cl2m, ctm2lat=cm2l, rdm2at=rdm2, bd2at=bd2, yparel=ypr, aat=aat.value)

    def get_homog(self):
        nl, ny = self.nl, self.ny
        hom = np.zeros(ny * (2 * (nl - 1) + 1))
        aux = np.zeros(5 * (nl - 1) + 2)
        self.lib.ref_atm_get_homog(_dp(hom), _dp(aux))
        n1 = ny * (nl - 1)
        return dict(pch1at=hom[:n1].reshape((ny, nl - 1), order="F").copy(order="F"),
                    pch2at=hom[n1:2 * n1].reshape((ny, nl - 1), order="F").copy(order="F"),
                    pbhat=hom[2 * n1:].copy(),
                    aipcha=aux[0:nl - 1].copy(), hc1sat=aux[nl - 1:2 * (nl - 1)].copy(),
                    hc2sat=aux[2 * (nl - 1):3 * (nl - 1)].copy(), hc1nat=aux[3 * (nl - 1):4 * (nl - 1)].copy(),
                    hc2nat=aux[4 * (nl - 1):5 * (nl - 1)].copy(), hbsiat=aux[5 * (nl - 1)], aipbha=aux[5 * (nl - 1) + 1])

    def qgastep(self):
        run_big_stack(self.lib.ref_qgastep)

    def atinvq(self):
        run_big_stack(self.lib.ref_atinvq)

    def atqzbd(self):
        run_big_stack(self.lib.ref_atqzbd)

    def lf_average(self):
        self.lib.ref_atm_lf_average()

    def steps(self, nt0, n):
        self.lib.ref_atm_steps.argtypes = [C.c_int, C.c_int]
        run_big_stack(self.lib.ref_atm_steps, int(nt0), int(n))

    def coupled_steps(self, nt0, n, nstr):
        self.lib.ref_coupled_steps.argtypes = [C.c_int, C.c_int, C.c_int]
        run_big_stack(self.lib.ref_coupled_steps, int(nt0), int(n), int(nstr))

    def helmholtz(self, wrk, bat):
        w = np.asfortranarray(wrk, dtype=np.float64).copy(order="F")
        b = np.ascontiguousarray(bat, dtype=np.float64)
        run_big_stack(self.lib.ref_atm_helmholtz, _dp(w), _dp(b))
        return w
