"""Host-side logic of the product (numpy start-up arithmetic, configs, synthetic
inputs) against the golden vectors from the true reference.  CPU only."""
import numpy as np
import pytest

import oracle_binding as ob
from common import CONFIG_NAMES, load_golden, make_oracle, preset, relerr
from qgcm_hip import config, hostinit, synth


@pytest.fixture(scope="module", params=CONFIG_NAMES)
def case(request):
    return preset(request.param), load_golden(request.param)


def test_eigmod_and_bd2oc(case):
    cfg, g = case
    A, rd, cl, cm = hostinit.eigmod(cfg.gpoc, cfg.hoc, cfg.fnot)
    assert np.array_equal(A, g["c_amatoc"])
    for got, key in ((rd, "c_rdm2oc"), (cl, "c_ctl2moc"), (cm, "c_ctm2loc")):
        assert relerr(got, g[key]) < 5e-15, key
    aoc, bd2 = hostinit.bd2oc(cfg)
    assert aoc == float(g["c_aoc"])
    assert np.array_equal(bd2, g["c_bd2oc"])


def test_init_q_and_constraints(case):
    cfg, g = case
    dd = np.zeros((cfg.nxpo, cfg.nypo), order="F")
    yp = cfg.yporel()
    qo = hostinit.q_from_p(cfg, g["c_amatoc"], yp, dd, g["in_po"])
    qom = hostinit.q_from_p(cfg, g["c_amatoc"], yp, dd, g["in_pom"])
    assert relerr(qo, g["init_qo"]) < 1e-15
    assert relerr(qom, g["init_qom"]) < 1e-15
    s = hostinit.constr(cfg, g["c_amatoc"], g["in_po"], g["in_pom"])
    assert relerr(s, g["init_scal"]) < 1e-14


def test_homsol_with_oracle_solver(case):
    """hostinit.homsol_* with the CPU oracle standing in for the HIP Helmholtz solver."""
    cfg, g = case
    o = make_oracle(cfg, g["c_yporel"])
    try:
        if cfg.cyclic:
            h = hostinit.homsol_cyc(cfg, g["c_rdm2oc"], g["c_bd2oc"], g["c_yporel"], o.helmholtz)
            big = max(np.abs(g["h_hc1soc"]).max(), np.abs(g["h_hc2noc"]).max())
            for k in ("pch1oc", "pch2oc", "pbhoc", "aipcho", "hbsioc", "aipbho"):
                assert relerr(h[k], g["h_" + k]) < 1e-13, k
            for k in ("hc1soc", "hc2soc", "hc1noc", "hc2noc"):
                assert np.abs(h[k] - g["h_" + k]).max() / big < 1e-13, k
        else:
            h = hostinit.homsol_box(cfg, g["c_rdm2oc"], g["c_ctm2loc"], g["c_bd2oc"], o.helmholtz)
            for k in ("ochom", "aipohs", "cdiffo", "cdhoc"):
                assert relerr(h[k], g["h_" + k]) < 1e-13, k
    finally:
        o.close()


def test_channel_homsol_from_the_column_solver(case):
    """The homogeneous solutions of a channel depend on y only, so y-slab set-up solves ONE tridiagonal system per mode
    on the host (hostinit.helmholtz_cyc_column) instead of calling a whole-domain Helmholtz solver: against the golden
    vectors of the reference's homsol (conhoms.F:376-543)."""
    cfg, g = case
    if not cfg.cyclic:
        pytest.skip("box ocean: homsol runs on the slabs themselves (SlabOcean.homsol)")
    h = hostinit.homsol_cyc(cfg, g["c_rdm2oc"], g["c_bd2oc"], g["c_yporel"], lambda rhs, boc: hostinit.helmholtz_cyc_column(cfg, rhs, boc))
    big = max(np.abs(g["h_hc1soc"]).max(), np.abs(g["h_hc2noc"]).max())
    for k in ("pch1oc", "pch2oc", "pbhoc", "aipcho", "hbsioc", "aipbho"):
        assert relerr(h[k], g["h_" + k]) < 1e-12, k
    for k in ("hc1soc", "hc2soc", "hc1noc", "hc2noc"):
        assert np.abs(h[k] - g["h_" + k]).max() / big < 1e-12, k


def test_wekpo_matches_oracle_restatement(case):
    cfg, g = case
    tx, ty = synth.wind_stress(cfg)
    ty = np.asfortranarray(1e-5 * np.sin(np.arange(cfg.nxpo) / 3.0)[:, None] * np.ones(cfg.nypo)[None, :])
    if cfg.cyclic:
        ty[-1, :] = ty[0, :]
    wt, wp = synth.wekpo_from_tau(cfg, tx, ty)
    wt2, wp2 = ob.wekpo_from_tau(tx, ty, cfg.cyclic, cfg.dxo, cfg.fnot)
    assert relerr(wt, wt2) < 1e-15 and relerr(wp, wp2) < 1e-15
    # the p-point average conserves the area integral (xfosubs.F comment "to conserve area integral")
    if not cfg.cyclic:
        assert abs(hostinit.xintp(wp) - wt.sum()) < 1e-12 * np.abs(wt).sum()


@pytest.mark.parametrize("name", ["box_small", "cyc_small"])
def test_wekpo_from_tau_is_the_reference_xforc(name):
    """The ocean-only Ekman pumping of qgcm_hip.synth (the restatement the device kernels k_wekto / k_wekpo are held
    to, bitwise, by tests/test_gpu_setup.py) against the REFERENCE's own `call xforc` on the same wind stress
    (src/xfosubs.F:566-683, compiled into oracle/_ref by build_ref.sh; tests/golden/setup_ref.npz): bit for bit."""
    import importlib.util
    import os
    from common import GOLDEN
    spec = importlib.util.spec_from_file_location("make_golden_setup", os.path.join(GOLDEN, "make_golden_setup.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    cfg, g = preset(name), load_golden("setup_ref")
    tx, ty = mg.stress(cfg)
    wekto, wekpo = synth.wekpo_from_tau(cfg, tx, ty)
    assert np.array_equal(wekto, g[name + "_wekto"]) and np.array_equal(wekpo, g[name + "_wekpo"])
    assert np.abs(wekpo).max() > 1e-7


def test_xintp():
    rng = np.random.default_rng(5)
    v = np.asfortranarray(rng.standard_normal((23, 17)))
    assert abs(hostinit.xintp(v) - ob.xintp(v)) < 1e-13


def test_presets_match_reference_examples():
    n5 = config.preset("natl5")
    assert (n5.nxpo, n5.nypo, n5.nlo, n5.dto) == (961, 961, 3, 540.0)
    s5 = config.preset("socn5")
    assert (s5.nxpo, s5.nypo, s5.cyclic) == (4609, 577, True)
    n1 = config.preset("natl1")
    assert (n1.nxpo, n1.nypo, n1.dto) == (4801, 4801, 180.0)
    # yporel spans -2400..+2400 km for NAtl 5 km
    yp = n5.yporel()
    assert yp[0] == -2.4e6 and yp[-1] == 2.4e6
    assert abs(n5.model_years_per_day(62.9) - 93.06) < 0.05


# ---- atmosphere: the numpy start-up arithmetic of AtmosModel vs the coupled reference build ----
@pytest.fixture(scope="module", params=["atm_tiny", "atm_small"])
def atm_case(request):
    from common import ATM_CASES
    return config.atmos_preset(dict(ATM_CASES)[request.param]), load_golden(request.param)


def test_atmos_eigmod_and_bd2at(atm_case):
    acfg, g = atm_case
    A, rd, cl, cm = hostinit.eigmod(acfg.gpat, acfg.hat, acfg.fnot, atmos=True)
    assert np.array_equal(A, g["c_amatat"])
    for got, key in ((rd, "c_rdm2at"), (cl, "c_ctl2mat"), (cm, "c_ctm2lat")):
        assert relerr(got, g[key]) < 5e-15, key
    aat, bd2 = hostinit.bd2oc(acfg)
    assert aat == float(g["c_aat"]) and np.array_equal(bd2, g["c_bd2at"])
    assert np.array_equal(acfg.yporel(), g["c_yparel"])


def test_atmos_init_q_and_constraints(atm_case):
    acfg, g = atm_case
    yp = acfg.yporel()
    qa = hostinit.q_from_p(acfg, g["c_amatat"], yp, g["in_ddynat"], g["in_pa"])
    qam = hostinit.q_from_p(acfg, g["c_amatat"], yp, g["in_ddynat"], g["in_pam"])
    assert relerr(qa, g["init_qa"]) < 1e-15 and relerr(qam, g["init_qam"]) < 1e-15
    s = hostinit.constr(acfg, g["c_amatat"], g["in_pa"], g["in_pam"])
    nl = acfg.nla
    scale = acfg.xla * acfg.yla * np.abs(g["in_pa"]).max()
    assert np.abs(s[:2 * (nl - 1)] - g["init_scal"][:2 * (nl - 1)]).max() < 1e-14 * scale
    assert relerr(s[2 * (nl - 1):], g["init_scal"][2 * (nl - 1):]) < 1e-12


def test_atmos_homsol_with_oracle_solver(atm_case):
    from common import atm_inputs, make_atm_oracle
    acfg, g = atm_case
    o = make_atm_oracle(acfg, g, atm_inputs(g, acfg))
    try:
        h = hostinit.homsol_cyc(acfg, g["c_rdm2at"], g["c_bd2at"], g["c_yparel"], o.helmholtz)
        big = max(np.abs(g["h_hc1sat"]).max(), np.abs(g["h_hc2nat"]).max())
        for k, kk in (("pch1oc", "pch1at"), ("pch2oc", "pch2at"), ("pbhoc", "pbhat"), ("aipcho", "aipcha"),
                      ("hbsioc", "hbsiat"), ("aipbho", "aipbha")):
            assert relerr(h[k], g["h_" + kk]) < 1e-13, k
        for k, kk in (("hc1soc", "hc1sat"), ("hc2soc", "hc2sat"), ("hc1noc", "hc1nat"), ("hc2noc", "hc2nat")):
            assert np.abs(h[k] - g["h_" + kk]).max() / big < 1e-13, k
    finally:
        o.close()


def test_coupled_presets_match_the_reference_example():
    oc, at = config.preset("cpl_natl5"), config.atmos_preset("cpl_natl5")
    assert (oc.nxpo, oc.nypo, oc.nstr) == (961, 961, 3)
    assert (at.nxpa, at.nypa, at.nla, at.dxa, at.dta) == (385, 97, 3, 8.0e4, 180.0)
    assert at.ah4at == (1.5e14,) * 3 and at.hat == (2000.0, 3000.0, 4000.0) and at.gpat == (1.2, 0.4)
