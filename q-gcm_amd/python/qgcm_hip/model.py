"""OceanModel - Python mirror of the reference's operator interface for the ocean
hot path.  The reference exposes three argument-less module procedures acting on
MODULE ocstate / ochomog arrays (src/q-gcm.F:1243-1249):

    call qgostep ; call ocinvq ; call ocqbdy (qo, po)

Here the same three names are methods, the module arrays are device resident,
and ``po / pom / qo / qom`` are fetched on demand (properties).  Everything that
computes goes through the C ABI of include/qgcm_hip.h; nothing here falls back
to numpy for the per-step path.
"""
import ctypes as C

import numpy as np

from . import hostinit
from .lib import MAXL, Params, QgcmHipError, check, load_library


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _f(a):
    return None if a is None else np.asfortranarray(a, dtype=np.float64)


class OceanModel:
    """One ocean configuration on one MI355X.

    Start-up follows src/q-gcm.F:380-976 for the ocean: grid, eigmod, tridiagonal
    coefficients, homsol (its Helmholtz solves run on the GPU through
    qgcm_hip_helmholtz).  A caller that already has the reference's constants
    (the Fortran host) can pass them in ``consts`` to skip that arithmetic.
    """

    def __init__(self, cfg, ddynoc=None, device=-1, consts=None):
        self.cfg = cfg
        self.L = load_library()
        self.h = C.c_void_p()
        nl = cfg.nlo
        if nl > MAXL:
            raise QgcmHipError("nlo=%d exceeds QGCM_HIP_MAXL" % nl)
        self.yporel = cfg.yporel()
        self.ddynoc = np.zeros((cfg.nxpo, cfg.nypo), order="F") if ddynoc is None else _f(ddynoc)
        atmos = bool(getattr(cfg, "atmos", False))
        if consts is None:
            A, rdm2, cl2m, cm2l = hostinit.eigmod(cfg.gpoc, cfg.hoc, cfg.fnot, atmos=atmos)
        else:
            A, rdm2, cl2m, cm2l = (consts[k] for k in ("amatoc", "rdm2oc", "ctl2moc", "ctm2loc"))
        self.amatoc, self.rdm2oc, self.ctl2moc, self.ctm2loc = A, rdm2, cl2m, cm2l
        self.aoc, self.bd2oc = hostinit.bd2oc(cfg)
        p = Params()
        p.nxpo, p.nypo, p.nlo, p.cyclic, p.atmos = cfg.nxpo, cfg.nypo, nl, int(cfg.cyclic), int(atmos)
        p.fnot, p.beta, p.dxo, p.dyo = cfg.fnot, cfg.beta, cfg.dxo, cfg.dyo
        p.tdto, p.delek, p.bccooc, p.aoc = cfg.tdto, cfg.delek, cfg.bccooc, self.aoc
        for k in range(nl):
            p.ah2oc[k], p.ah4oc[k], p.hoc[k], p.rdm2oc[k] = cfg.ah2oc[k], cfg.ah4oc[k], cfg.hoc[k], rdm2[k]
        for k in range(nl - 1):
            p.gpoc[k] = cfg.gpoc[k]
        for name, M in (("amatoc", A), ("ctl2moc", cl2m), ("ctm2loc", cm2l)):
            flat = np.asarray(M).ravel(order="F")
            arr = getattr(p, name)
            for i, v in enumerate(flat):
                arr[i] = v
        self.params = p
        check(self.L.qgcm_hip_create(C.byref(self.h), C.byref(p), int(device)))
        check(self.L.qgcm_hip_set_grid(self.h, _dp(self.yporel), _dp(self.bd2oc), _dp(self.ddynoc)))
        # homsol
        if consts is not None and ("ochom" in consts or "pch1oc" in consts):
            self.homog = consts
        elif cfg.cyclic:
            self.homog = hostinit.homsol_cyc(cfg, rdm2, self.bd2oc, self.yporel, self.helmholtz)
        else:
            self.homog = hostinit.homsol_box(cfg, rdm2, cm2l, self.bd2oc, self.helmholtz)
        hg = self.homog
        if cfg.cyclic:
            args = [_f(hg["pch1oc"]), _f(hg["pch2oc"]), np.ascontiguousarray(hg["pbhoc"]),
                    *[np.ascontiguousarray(hg[k], dtype=np.float64) for k in ("aipcho", "hc1soc", "hc2soc", "hc1noc", "hc2noc")]]
            self._keep = args
            check(self.L.qgcm_hip_set_homog_cyc(self.h, *[_dp(a) for a in args], float(hg["hbsioc"]), float(hg["aipbho"])))
        else:
            args = [_f(hg["ochom"]), _f(hg["cdiffo"]), _f(hg["cdhoc"])]
            check(self.L.qgcm_hip_set_homog_box(self.h, *[_dp(a) for a in args]))
        self.nscal = 2 * (nl - 1) + 4 * nl
        self.step_index = 1  # next 1-based ocean step
        if getattr(cfg, "l_spl", 0.0) > 0.0:  # a -Dsponge_layer_k247 configuration
            self.set_sponge(hostinit.sponge_ramp(cfg), cfg.c1_spl)

    # -- life cycle ----------------------------------------------------------
    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.L.qgcm_hip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _f3(self):
        c = self.cfg
        return np.zeros((c.nxpo, c.nypo, c.nlo), order="F")

    # -- state ----------------------------------------------------------------
    def set_p(self, po, pom=None):
        """Load pressures and derive q and the constraint scalars as the reference
        does at start-up (src/q-gcm.F:711-731: constr, qcomp, ocqbdy, merqcy)."""
        c = self.cfg
        po = _f(po)
        pom = po.copy(order="F") if pom is None else _f(pom)
        qo = hostinit.q_from_p(c, self.amatoc, self.yporel, self.ddynoc, po)
        qom = hostinit.q_from_p(c, self.amatoc, self.yporel, self.ddynoc, pom)
        self.set_state(po, pom, qo, qom)
        self.set_scalars(hostinit.constr(c, self.amatoc, po, pom))

    def init_from_p(self, po, pom=None):
        """The same start-up sequence ON THE DEVICE (qgcm_hip_init_from_p): only po, pom cross PCIe - the restart
        path of a host that keeps no q (restart dumps carry po, pom only, src/q-gcm.F:3076-3086)."""
        po = _f(po)
        pom = po if pom is None else _f(pom)
        self.set_state(po, pom, None, None)
        check(self.L.qgcm_hip_init_from_p(self.h))

    def wekpo_from_tau(self, tauxo, tauyo):
        """Ocean-only Ekman pumping from the wind stress on the device (src/xfosubs.F:566-645)."""
        check(self.L.qgcm_hip_wekpo_from_tau(self.h, _dp(_f(tauxo)), _dp(_f(tauyo))))

    def prsamp(self):
        """The ocean numbers of the reference's progress print-out (src/q-gcm.F:1933-2066) without pulling the state:
        dict(po_centre, qo_centre, pavgoc, qavgoc: nlo each; sstmin, sstmax)."""
        nl = self.cfg.nlo
        out = np.zeros(4 * nl + 2)
        check(self.L.qgcm_hip_prsamp(self.h, _dp(out)))
        return dict(po_centre=out[:nl].copy(), qo_centre=out[nl:2 * nl].copy(), pavgoc=out[2 * nl:3 * nl].copy(),
                    qavgoc=out[3 * nl:4 * nl].copy(), sstmin=out[4 * nl], sstmax=out[4 * nl + 1])

    def set_state(self, po=None, pom=None, qo=None, qom=None):
        a = [_f(x) for x in (po, pom, qo, qom)]
        check(self.L.qgcm_hip_set_state(self.h, *[_dp(x) for x in a]))

    def get_state(self):
        a = [self._f3() for _ in range(4)]
        check(self.L.qgcm_hip_get_state(self.h, *[_dp(x) for x in a]))
        return a

    po = property(lambda self: self.get_state()[0])
    pom = property(lambda self: self.get_state()[1])
    qo = property(lambda self: self.get_state()[2])
    qom = property(lambda self: self.get_state()[3])

    def set_forcing(self, wekpo=None, entoc=None, xon=None):
        w, e = _f(wekpo), _f(entoc)
        x = None if xon is None else np.ascontiguousarray(xon, dtype=np.float64)
        check(self.L.qgcm_hip_set_forcing(self.h, _dp(w), _dp(e), _dp(x)))

    def set_cyc_forcing(self, txisoc, txinoc, enisoc=None, eninoc=None):
        nl = self.cfg.nlo
        es = np.zeros(nl - 1) if enisoc is None else np.ascontiguousarray(enisoc, dtype=np.float64)
        en = np.zeros(nl - 1) if eninoc is None else np.ascontiguousarray(eninoc, dtype=np.float64)
        check(self.L.qgcm_hip_set_cyc_forcing(self.h, float(txisoc), float(txinoc), _dp(es), _dp(en)))

    def set_sponge(self, r_spl, c1_spl):
        """The fork's sponge layer (src/qgosubs.F:203-205): ramp r_spl(nxpo, nypo) and the constant c1_spl;
        r_spl = None switches the term off."""
        if r_spl is None:
            check(self.L.qgcm_hip_set_sponge(self.h, None, 0.0))
            return
        r = _f(r_spl)
        assert r.shape == (self.cfg.nxpo, self.cfg.nypo)
        check(self.L.qgcm_hip_set_sponge(self.h, _dp(r), float(c1_spl)))

    def set_scalars(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64)
        assert s.size == self.nscal
        check(self.L.qgcm_hip_set_scalars(self.h, _dp(s)))

    def get_scalars(self):
        s = np.zeros(self.nscal)
        check(self.L.qgcm_hip_get_scalars(self.h, _dp(s)))
        return s

    def get_monitors(self):
        """(ermaso, emfroc) of the last ocinvq of a zonally cyclic ocean (src/ocisubs.F:268-283; atmosphere: ermasa,
        emfrat): nlo-1 values each."""
        e, f = np.zeros(self.cfg.nlo - 1), np.zeros(self.cfg.nlo - 1)
        check(self.L.qgcm_hip_get_monitors(self.h, _dp(e), _dp(f)))
        return e, f

    def get_inv_diag(self):
        nl = self.cfg.nlo
        x = np.zeros(nl)
        cf = np.zeros(2 * nl + 1)
        check(self.L.qgcm_hip_get_inv_diag(self.h, _dp(x), _dp(cf)))
        n = 2 * (nl - 1) + 1 if self.cfg.cyclic else nl - 1
        return x, cf[:n].copy()

    # -- the path (same names as the reference's module procedures) ------------
    def qgostep(self):
        check(self.L.qgcm_hip_qgostep(self.h))

    def ocinvq(self):
        check(self.L.qgcm_hip_ocinvq(self.h))

    def ocqbdy(self):
        check(self.L.qgcm_hip_ocqbdy(self.h))

    def lf_average(self):
        check(self.L.qgcm_hip_lf_average(self.h))

    def ocqbdy_host(self, q, p):
        """`call ocqbdy (q, p)` / `call atqzbd (q, p)` on host arrays (start-up use, src/q-gcm.F:724-725, 743-744):
        returns q with its boundary ring recomputed from p; the device-resident state is not touched."""
        q = np.array(q, dtype=np.float64, order="F", copy=True)
        check(self.L.qgcm_hip_ocqbdy_host(self.h, _dp(q), _dp(_f(p))))
        return q

    def steps(self, n, s0=None):
        """n whole ocean steps (q-gcm.F:1243-1249 + the averaging of :1328)."""
        s0 = self.step_index if s0 is None else int(s0)
        check(self.L.qgcm_hip_steps(self.h, s0, int(n)))
        self.step_index = s0 + int(n)

    def sync(self):
        check(self.L.qgcm_hip_sync(self.h))

    # -- validity scan (`call valids (solnok)`, src/q-gcm.F:1278) --
    def set_dtopoc(self, dtopoc):
        check(self.L.qgcm_hip_set_dtopoc(self.h, _dp(_f(dtopoc))))

    def valids(self):
        """(solnok, out): out = min/max of po, qo, sst, wekto, layer thickness top/intermediate/bottom,
        then hfbad(1..nlo) in per cent (src/valsubs.F:272-527)."""
        out = np.zeros(14 + self.cfg.nlo)
        ok = C.c_int()
        check(self.L.qgcm_hip_valids(self.h, _dp(out), C.byref(ok)))
        return bool(ok.value), out

    # -- ocean mixed layer (`call oml`, src/q-gcm.F:1232) -------
    def oml_init(self, om):
        """Switch the mixed layer on (om: qgcm_hip.config.OmlConfig): steps() then runs oml before
        qgostep in every step and averages sst with the other fields."""
        from .lib import OmlParams
        p = OmlParams()
        p.hmoc, p.toc1, p.toc2, p.st2d, p.st4d = om.hmoc, om.toc[0], om.toc[1], om.st2d, om.st4d
        p.ycexp, p.rrcpoc, p.tsbdy, p.tnbdy = om.ycexp, om.rrcpoc, om.tsbdy, om.tnbdy
        p.sb_hflux, p.nb_hflux = int(om.sb_hflux), int(om.nb_hflux)
        check(self.L.qgcm_hip_oml_init(self.h, C.byref(p)))

    def oml_set_state(self, sst=None, sstm=None):
        a = [_f(x) for x in (sst, sstm)]
        for x in a:
            assert x is None or x.shape == (self.cfg.nxto, self.cfg.nyto)
        check(self.L.qgcm_hip_oml_set_state(self.h, *[_dp(x) for x in a]))

    def oml_get_state(self):
        a = [np.zeros((self.cfg.nxto, self.cfg.nyto), order="F") for _ in range(2)]
        check(self.L.qgcm_hip_oml_get_state(self.h, *[_dp(x) for x in a]))
        return a

    def oml_set_forcing(self, fnetoc=None, wekto=None, tauxo=None, tauyo=None):
        a = [_f(x) for x in (fnetoc, wekto, tauxo, tauyo)]
        check(self.L.qgcm_hip_oml_set_forcing(self.h, *[_dp(x) for x in a]))

    def oml(self):
        check(self.L.qgcm_hip_oml(self.h))

    def oml_get_diag(self):
        """entoc(nxpo,nypo) and (xon(1), cfraoc, centoc, enisoc(1), eninoc(1))."""
        e = np.zeros((self.cfg.nxpo, self.cfg.nypo), order="F")
        d = np.zeros(5)
        check(self.L.qgcm_hip_oml_get_diag(self.h, _dp(e), _dp(d)))
        return e, d

    # -- the row transforms by themselves (replacements of FFTPACK's dsint / drfftf / drfftb) -------------------
    def wrk_set(self, wrk):
        w = _f(wrk)
        assert w.shape == (self.cfg.nxpo, self.cfg.nypo, self.cfg.nlo)
        check(self.L.qgcm_hip_wrk_set(self.h, _dp(w)))

    def wrk_get(self):
        w = np.zeros((self.cfg.nxpo, self.cfg.nypo, self.cfg.nlo), order="F")
        check(self.L.qgcm_hip_wrk_get(self.h, _dp(w)))
        return w

    def row_transform(self, inverse):
        check(self.L.qgcm_hip_row_transform(self.h, int(inverse)))

    def helmholtz(self, wrk, boc):
        """hsbxoc / hscyoc replacement (src/ocisubs.F:415-618); returns the solution."""
        w = np.array(wrk, dtype=np.float64, order="F", copy=True)
        b = np.ascontiguousarray(boc, dtype=np.float64)
        check(self.L.qgcm_hip_helmholtz(self.h, _dp(w), _dp(b)))
        return w

    # -- measurement ------------------------------------------------------------
    def time_steps(self, n, s0=None):
        """HIP-event time (ms) of n steps on the handle's stream."""
        s0 = self.step_index if s0 is None else int(s0)
        ms = C.c_float()
        check(self.L.qgcm_hip_time_steps(self.h, s0, int(n), C.byref(ms)))
        self.step_index = s0 + int(n)
        return ms.value

    def prepare_steps(self, n, s0=None):
        """Build the HIP graphs steps(n, s0) will replay (nothing runs): keeps graph capture / instantiation out of a
        window the caller times with its own clock."""
        s0 = self.step_index if s0 is None else int(s0)
        check(self.L.qgcm_hip_prepare_steps(self.h, s0, int(n)))

    def profile_steps(self, n, s0=None):
        """Per-kernel HIP-event totals over n eagerly launched steps:
        {name: (total_ms, launches)}."""
        s0 = self.step_index if s0 is None else int(s0)
        cap = 32
        ms = (C.c_double * cap)()
        ln = (C.c_int * cap)()
        names = (C.c_char_p * cap)()
        nk = C.c_int(cap)
        check(self.L.qgcm_hip_profile_steps(self.h, s0, int(n), ms, ln, names, C.byref(nk)))
        self.step_index = s0 + int(n)
        return {names[i].decode(): (ms[i], ln[i]) for i in range(nk.value)}

    def stream_mix_bandwidth(self, nr, nw, field_bytes, reps=20):
        """GB/s of a pure streaming kernel reading nr and writing nw fields of field_bytes (the practical ceiling for a
        kernel of that mix at that size)."""
        g = C.c_double()
        check(self.L.qgcm_hip_stream_mix_bandwidth(self.h, int(nr), int(nw), C.c_size_t(int(field_bytes)), int(reps), C.byref(g)))
        return g.value

    def copy_bandwidth(self, nbytes=1 << 30, reps=10):
        g = C.c_double()
        check(self.L.qgcm_hip_copy_bandwidth(self.h, C.c_size_t(nbytes), int(reps), C.byref(g)))
        return g.value


_ATM_NAMES = {"amatat": "amatoc", "rdm2at": "rdm2oc", "ctl2mat": "ctl2moc", "ctm2lat": "ctm2loc",
              "pch1at": "pch1oc", "pch2at": "pch2oc", "pbhat": "pbhoc", "aipcha": "aipcho", "hc1sat": "hc1soc",
              "hc2sat": "hc2soc", "hc1nat": "hc1noc", "hc2nat": "hc2noc", "hbsiat": "hbsioc", "aipbha": "aipbho"}


class AtmosModel(OceanModel):
    """The atmospheric channel of a coupled run on the GPU.  The reference steps it with

        call qgastep ; call atinvq ; call atqzbd (qa, pa)          (src/q-gcm.F:1262-1268)

    on MODULE atstate / athomog arrays; the same three names are methods here.  ``cfg`` is an AtmosConfig; the
    inherited accessors carry the atmosphere's arrays: get_state() = pa, pam, qa, qam; get_scalars() = dpiat,
    dpiatp, atmcs, atmcn, atmcsp, atmcnp; steps(n) averages the time levels when mod(nt-1,100) == 0
    (src/q-gcm.F:1370).  ``consts`` may carry the reference's own eigmod / homsol products under their atmosphere
    names (amatat, ctl2mat, ..., pch1at, ...), as a drop-in host would pass them."""

    def __init__(self, cfg, ddynat=None, device=-1, consts=None):
        if consts is not None:
            consts = {_ATM_NAMES.get(k, k): v for k, v in consts.items()}
        OceanModel.__init__(self, cfg, ddynoc=ddynat, device=device, consts=consts)

    def set_forcing(self, wekpa=None, entat=None, xan=None, txis=None, txin=None, enis=None, enin=None):
        """wekpa, entat (p grid), xan(nla-1) and the line integrals txisat, txinat, enisat, eninat that
        xforc / aml leave in MODULE atstate / athomog."""
        OceanModel.set_forcing(self, wekpa, entat, xan)
        if txis is not None or txin is not None or enis is not None or enin is not None:
            self.set_cyc_forcing(0.0 if txis is None else txis, 0.0 if txin is None else txin, enis, enin)

    def qgastep(self):
        check(self.L.qgcm_hip_qgastep(self.h))

    def atinvq(self):
        check(self.L.qgcm_hip_atinvq(self.h))

    def atqzbd(self):
        check(self.L.qgcm_hip_atqzbd(self.h))

    def get_bsums(self):
        """ajisat, ajinat, ap5sat, ap5nat (nla each) of the last qgastep (valid after the following atinvq)."""
        b = np.zeros(4 * self.cfg.nlo)
        check(self.L.qgcm_hip_get_bsums(self.h, _dp(b)))
        return b

    pa = OceanModel.po
    pam = OceanModel.pom
    qa = OceanModel.qo
    qam = OceanModel.qom


def share_gpu(ocean, atmos, atmos_cus=None):
    """Give the two halves of a coupled run disjoint CU ranges of the GPU they share (qgcm_hip_set_cu_range): the
    atmosphere the first atmos_cus compute units (default 2/8 of the device: 64 of an MI355X's 256 = two XCDs), the ocean the rest;
    atmos_cus = 0 returns both to unrestricted streams.  Call before coupled_steps."""
    import torch
    ncu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    acu = ncu // 4 if atmos_cus is None else int(atmos_cus)
    if acu == 0:
        check(ocean.L.qgcm_hip_set_cu_range(ocean.h, 0, 0))
        check(atmos.L.qgcm_hip_set_cu_range(atmos.h, 0, 0))
    else:
        check(atmos.L.qgcm_hip_set_cu_range(atmos.h, 0, acu))
        check(ocean.L.qgcm_hip_set_cu_range(ocean.h, acu, ncu - acu))
    return acu


def coupled_steps(ocean, atmos, nt0, n, nstr):
    """n atmospheric steps nt = nt0.. with one ocean step before every one with mod(nt,nstr) == 1
    (src/q-gcm.F:1220-1268), forcing held; either model may be None."""
    L = (ocean or atmos).L
    check(L.qgcm_hip_coupled_steps(ocean.h if ocean is not None else None, atmos.h if atmos is not None else None,
                                   int(nt0), int(n), int(nstr)))
    if atmos is not None:
        atmos.step_index = int(nt0) + int(n)
