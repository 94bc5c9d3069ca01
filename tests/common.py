"""Shared helpers for the parity tests (tests may use oracle/, the product may not)."""
import os

import numpy as np

import oracle_binding as ob
from qgcm_hip import config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIELDS = ("po", "pom", "qo", "qom")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / den) if den > 0 else float(np.abs(a).max())


def make_oracle(cfg, yporel=None):
    o = ob.Oracle(cfg.nxpo, cfg.nypo, cfg.nlo, cfg.cyclic, cfg.fnot, cfg.beta, cfg.dxo, cfg.dto,
                  cfg.delek, cfg.bccooc, cfg.ah2oc, cfg.ah4oc, cfg.hoc, cfg.gpoc,
                  cfg.yporel() if yporel is None else yporel)
    if getattr(cfg, "l_spl", 0.0) > 0.0:  # a -Dsponge_layer_k247 configuration: as OceanModel does by itself
        from qgcm_hip import hostinit
        o.set_sponge(hostinit.sponge_ramp(cfg), cfg.c1_spl)
    return o


def apply_inputs(model, g, cfg):
    """Load the fixture's inputs into an Oracle or an OceanModel (same method names)."""
    model.set_p(g["in_po"], g["in_pom"])
    model.set_forcing(g["in_wekpo"], g["in_entoc"], g["in_xon"])
    if cfg.cyclic:
        model.set_cyc_forcing(float(g["in_txis"]), float(g["in_txin"]), g["in_enis"], g["in_enin"])
    if "in_rspl" in g:  # the reference build's own ramp (its exp() need not round like numpy's)
        model.set_sponge(g["in_rspl"], float(g["in_c1spl"]))


def load_snapshot(model, g, tag):
    model.set_state(*[g["%s_%s" % (tag, f)] for f in FIELDS])
    model.set_scalars(g[tag + "_scal"])


def state_errs(model, g, tag):
    st = model.get_state()
    return {f: relerr(st[i], g["%s_%s" % (tag, f)]) for i, f in enumerate(FIELDS)}


def scal_err(model, g, tag, cfg):
    """Constraint scalars are cancelling area integrals: compare relative to
    xlo*ylo*max|po|, not to their own magnitude."""
    s, r = model.get_scalars(), g[tag + "_scal"]
    scale = cfg.xlo * cfg.ylo * np.abs(g[tag + "_po"]).max()
    nl = cfg.nlo
    e = np.abs(s[:2 * (nl - 1)] - r[:2 * (nl - 1)]).max() / scale
    if cfg.cyclic:
        den = np.abs(r[2 * (nl - 1):]).max()
        e = max(e, np.abs(s[2 * (nl - 1):] - r[2 * (nl - 1):]).max() / den)
    return float(e)


# *_ah2: the tiny grids with ah2oc != 0 (the Del-4th-of-p viscosity term of src/qgosubs.F:375-377 and, cyclic, the
# ap3soc / ap3noc boundary sums) - tests/golden/make_golden.py
# *_spl: the tiny grids of reference builds with -Dsponge_layer_k247 (src/qgosubs.F:203-205)
CONFIG_NAMES = ("box_tiny", "box_tiny2", "box_small", "cyc_tiny", "cyc_small", "box_tiny_ah2", "cyc_tiny_ah2",
                "box_tiny_spl", "cyc_tiny_spl", "box_tiny5", "cyc_tiny6")
BOX_NAMES = ("box_tiny", "box_tiny2", "box_small", "box_tiny_ah2", "box_tiny_spl", "box_tiny5")
SNAPS = {"box_tiny": (1, 2, 25, 26, 60), "box_tiny2": (1, 26), "box_tiny5": (1, 2, 26), "cyc_tiny6": (1, 2, 26), "box_small": (1, 30),
         "cyc_tiny": (1, 2, 25, 26, 60), "cyc_small": (1, 30), "box_tiny_ah2": (1, 2, 26), "cyc_tiny_ah2": (1, 2, 26),
         "box_tiny_spl": (1, 2, 26), "cyc_tiny_spl": (1, 2, 26)}


def preset(name):
    return config.preset(name)


# ---- ocean mixed layer fixtures (tests/golden/make_golden_oml.py) -----------------------------
OML_CASES = (("oml_box_tiny", "box_tiny"), ("oml_box_tiny_sb", "box_tiny"), ("oml_cyc_tiny", "cyc_tiny"))
OML_SNAPS = (1, 2, 26, 40)


def oml_config(g):
    """OmlConfig of a mixed-layer fixture."""
    p = g["oml_params"]
    return config.OmlConfig(hmoc=p[0], toc=(p[1], p[2]), st2d=p[3], st4d=p[4], ycexp=p[5], rhooc=1.0, cpoc=1.0 / p[6],
                            sb_hflux=bool(p[7]), tsbdy=p[8], nb_hflux=bool(p[9]), tnbdy=p[10])


def oml_load(model, g, cfg, is_oracle):
    """Inputs of a mixed-layer fixture into an Oracle or an OceanModel."""
    nl = cfg.nlo
    model.set_p(g["in_po"], g["in_pom"])
    model.set_forcing(g["in_wekpo"], np.zeros((cfg.nxpo, cfg.nypo), order="F"), np.zeros(nl - 1))
    if cfg.cyclic:
        model.set_cyc_forcing(float(g["in_txis"]), float(g["in_txin"]), np.zeros(nl - 1), np.zeros(nl - 1))
    if is_oracle:
        model.oml_set(g["in_sst"], g["in_sstm"], g["in_fnetoc"], g["in_wekto"], g["in_tauxo"], g["in_tauyo"])
    else:
        model.oml_set_state(g["in_sst"], g["in_sstm"])
        model.oml_set_forcing(g["in_fnetoc"], g["in_wekto"], g["in_tauxo"], g["in_tauyo"])


# ---- atmosphere fixtures (tests/golden/make_golden_atmos.py), SURVEY 8 row f3 ------------------------------
ATM_CASES = (("atm_tiny", "cpl_tiny"), ("atm_small", "cpl_small"), ("atm_natl5", "cpl_natl5"))
ATM_SNAPS = {"atm_tiny": (1, 2, 100, 101, 130), "atm_small": (1, 40), "atm_natl5": (1, 6, 101)}
ATM_FIELDS = ("pa", "pam", "qa", "qam")


def atm_inputs(g, acfg):
    """Inputs of an atmosphere fixture. atm_natl5 stores every 4th row / column: the full fields are re-generated
    from qgcm_hip.synth.atmos_fields and must reproduce the stored sample bit for bit."""
    st = int(g["stride"]) if "stride" in g else 1
    if st == 1:
        return {k: g["in_" + k] for k in ("pa", "pam", "wekpa", "entat", "ddynat", "xan", "txis", "txin", "enis", "enin")}
    from qgcm_hip import synth
    f = synth.atmos_fields(acfg)
    for k in ("pa", "pam", "wekpa", "entat", "ddynat"):
        assert np.array_equal(f[k][::st, ::st], g["in_" + k]), "synthetic atmosphere input %s drifted from the fixture" % k
    for k in ("xan", "txis", "txin", "enis", "enin"):
        assert np.array_equal(np.asarray(f[k]), g["in_" + k]), k
    return f


def make_atm_oracle(acfg, g, f):
    return ob.AtmosOracle(acfg.nxpa, acfg.nypa, acfg.nla, acfg.fnot, acfg.beta, acfg.dxa, acfg.dta, acfg.bccoat,
                          acfg.ah4at, acfg.hat, acfg.gpat, g["c_yparel"], f["ddynat"])


def atm_apply(model, f):
    """Start-up sequence + forcing into an AtmosOracle or an AtmosModel (same method names)."""
    model.set_p(f["pa"], f["pam"])
    model.set_forcing(f["wekpa"], f["entat"], f["xan"], float(f["txis"]), float(f["txin"]), f["enis"], f["enin"])


def atm_state_errs(model, g, tag):
    st = int(g["stride"]) if "stride" in g else 1
    s = model.get_state()
    return {n: relerr(s[i][::st, ::st], g["%s_%s" % (tag, n)]) for i, n in enumerate(ATM_FIELDS)}


def atm_load_snapshot(model, g, tag):
    model.set_state(*[g["%s_%s" % (tag, n)] for n in ATM_FIELDS])
    model.set_scalars(g[tag + "_scal"])


def atm_scal_err(model, g, tag, acfg):
    """dpiat relative to xla*yla*max|pa| (cancelling area integrals); atmc* relative to their maximum."""
    s, r = model.get_scalars(), g[tag + "_scal"]
    nl = acfg.nla
    scale = acfg.xla * acfg.yla * np.abs(g[tag + "_pa"]).max()
    e = np.abs(s[:2 * (nl - 1)] - r[:2 * (nl - 1)]).max() / scale
    den = np.abs(r[2 * (nl - 1):]).max()
    return float(max(e, np.abs(s[2 * (nl - 1):] - r[2 * (nl - 1):]).max() / den))


def cpl_fullsize_inputs(g, oc, at):
    """Inputs of tests/golden/cpl_natl5_sample.npz (make_golden_atmos.py coupled_fullsize): re-generated from
    qgcm_hip.synth; the stored strided samples must be reproduced bit for bit."""
    from qgcm_hip import synth
    so, sa = int(g["stride_oc"]), int(g["stride"])
    po = synth.gaussian_eddy(oc, noise=1.0e-3)
    pom = np.asfortranarray(0.98 * po)
    tx, ty = synth.wind_stress(oc)
    _, wekpo = synth.wekpo_from_tau(oc, tx, ty)
    for k, v in (("po", po), ("pom", pom), ("wekpo", wekpo)):
        assert np.array_equal(v[::so, ::so], g["in_" + k]), "synthetic ocean input %s drifted from the fixture" % k
    f = synth.atmos_fields(at)
    for k in ("pa", "pam", "wekpa", "entat", "ddynat"):
        assert np.array_equal(f[k][::sa, ::sa], g["in_" + k]), "synthetic atmosphere input %s drifted from the fixture" % k
    for k in ("xan", "txis", "txin", "enis", "enin"):
        assert np.array_equal(np.asarray(f[k]), g["in_" + k]), k
    return po, pom, wekpo, f


def cpl_fullsize_errs(ocean, atmos, g, nt):
    so = int(g["stride_oc"])
    e = {n: float(np.abs(ocean.get_state()[i][::so, ::so] - g["nt%d_%s" % (nt, n)]).max() / float(g["nt%d_%s_max" % (nt, n)]))
         for i, n in enumerate(FIELDS)}
    e.update(atm_state_errs(atmos, g, "nt%d" % nt))
    return e
