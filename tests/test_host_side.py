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
