"""Atmosphere path on the GPU (BASELINE configs[3] double_gyre_coupled): qgcm_hip_qgastep /
_atinvq / _atqzbd through the C ABI against golden vectors from a coupled build of the true reference
(tests/golden/make_golden_atmos.py) and against the CPU oracle.  Needs an MI355X."""
import numpy as np
import pytest

from common import (ATM_CASES, ATM_SNAPS, atm_apply, atm_inputs, atm_load_snapshot, atm_scal_err, atm_state_errs,
                    load_golden, make_atm_oracle, make_oracle, relerr)
from qgcm_hip import config

pytestmark = pytest.mark.gpu

TOL_CALL = 1e-12   # one call from identical state, relative to the field max-norm
TOL_RUN = 1e-10    # free-running, up to 130 steps


def ref_consts(g):
    """eigmod / homsol products as the reference computed them - what a drop-in host hands over."""
    c = {k: g["c_" + k] for k in ("amatat", "rdm2at", "ctl2mat", "ctm2lat")}
    c.update({k[2:]: g[k] for k in g if k.startswith("h_")})
    return c


@pytest.fixture(scope="module", params=[c[0] for c in ATM_CASES])
def case(request):
    from qgcm_hip import AtmosModel
    name = request.param
    acfg = config.atmos_preset(dict(ATM_CASES)[name])
    g = load_golden(name)
    f = atm_inputs(g, acfg)
    m = AtmosModel(acfg, ddynat=f["ddynat"])  # own start-up arithmetic; homsol through qgcm_hip_helmholtz
    yield name, acfg, g, f, m
    m.close()


def test_helmholtz_vs_reference(case):
    name, acfg, g, f, m = case
    if "helm_rhs" in g:
        rhs, st = g["helm_rhs"], 1
    else:
        rng = np.random.default_rng(int(g["helm_seed"]))
        rhs = np.asfortranarray(rng.standard_normal((acfg.nxpa, acfg.nypa)))
        rhs[-1, :] = rhs[0, :]
        st = int(g["stride"])
    b = g["c_bd2at"]
    assert relerr(m.helmholtz(rhs, b - g["c_rdm2at"][1])[::st, ::st], g["helm_sol"]) < TOL_CALL
    assert relerr(m.helmholtz(rhs, b - g["c_rdm2at"][0])[::st, ::st], g["helm_sol0"]) < TOL_CALL


def test_start_up_products(case):
    """eigmod (host) and homsol (hscyat on the GPU) against the reference's."""
    name, acfg, g, f, m = case
    assert np.array_equal(m.amatoc, g["c_amatat"]) and np.array_equal(m.bd2oc, g["c_bd2at"])
    for k, v in (("rdm2at", m.rdm2oc), ("ctl2mat", m.ctl2moc), ("ctm2lat", m.ctm2loc)):
        assert relerr(v, g["c_" + k]) < 1e-14, k
    big = max(np.abs(g["h_hc1sat"]).max(), np.abs(g["h_hc2nat"]).max())
    for k, kk in (("pch1oc", "pch1at"), ("pch2oc", "pch2at"), ("pbhoc", "pbhat"), ("aipcho", "aipcha"),
                  ("hbsioc", "hbsiat"), ("aipbho", "aipbha")):
        assert relerr(m.homog[k], g["h_" + kk]) < 1e-12, k
    for k, kk in (("hc1soc", "hc1sat"), ("hc2soc", "hc2sat"), ("hc1noc", "hc1nat"), ("hc2noc", "hc2nat")):
        assert np.abs(m.homog[k] - g["h_" + kk]).max() / big < 1e-12, k


def test_init_q_and_scalars(case):
    name, acfg, g, f, m = case
    atm_apply(m, f)
    e = atm_state_errs(m, g, "init")
    assert all(v < 1e-14 for v in e.values()), e
    assert atm_scal_err(m, g, "init", acfg) < 1e-13


def test_qgastep_bit_exact(case):
    name, acfg, g, f, m = case
    if "qgastep_pa" not in g:
        pytest.skip("per-call snapshots only stored for the small grids")
    atm_apply(m, f)
    atm_load_snapshot(m, g, "init")
    m.qgastep()
    e = atm_state_errs(m, g, "qgastep")
    assert all(v == 0.0 for v in e.values()), e


def test_atinvq_and_atqzbd(case):
    name, acfg, g, f, m = case
    if "qgastep_pa" not in g:
        pytest.skip("per-call snapshots only stored for the small grids")
    atm_apply(m, f)
    atm_load_snapshot(m, g, "init")
    m.qgastep()
    m.atinvq()
    e = atm_state_errs(m, g, "atinvq")
    assert e["pam"] == 0.0 and e["qa"] == 0.0 and e["qam"] == 0.0 and e["pa"] < TOL_CALL, e
    assert atm_scal_err(m, g, "atinvq", acfg) < 1e-13
    b, r = m.get_bsums(), g["bsums1"]   # ajisat, ajinat, ap5sat, ap5nat: parallel sums, compared to rounding
    nl = acfg.nla
    for q in range(4):
        assert np.abs(b[q * nl:(q + 1) * nl] - r[q * nl:(q + 1) * nl]).max() <= 1e-11 * np.abs(r[q * nl:(q + 1) * nl]).max() + 1e-300, q
    m.atqzbd()
    e = atm_state_errs(m, g, "atqzbd")
    assert e["qa"] < TOL_CALL and e["pa"] < TOL_CALL, e


def test_atqzbd_bit_exact(case):
    name, acfg, g, f, m = case
    if "atinvq_pa" not in g:
        pytest.skip("per-call snapshots only stored for the small grids")
    atm_apply(m, f)
    atm_load_snapshot(m, g, "atinvq")
    m.atqzbd()
    e = atm_state_errs(m, g, "atqzbd")
    assert all(v == 0.0 for v in e.values()), e


def test_whole_steps_vs_reference(case):
    name, acfg, g, f, m = case
    atm_apply(m, f)
    done = 0
    for s in ATM_SNAPS[name]:
        m.steps(s - done, s0=done + 1)
        done = s
        e = atm_state_errs(m, g, "steps%d" % s)
        tol = TOL_CALL if s <= 2 else TOL_RUN
        assert all(v < tol for v in e.values()), (s, e)
        assert atm_scal_err(m, g, "steps%d" % s, acfg) < 1e-11


def test_reference_constants_as_a_drop_in_host_passes_them(case):
    """The same steps with the reference's own eigmod / homsol products (MODULE atconst / athomog contents)."""
    from qgcm_hip import AtmosModel
    name, acfg, g, f, _ = case
    m = AtmosModel(acfg, ddynat=f["ddynat"], consts=ref_consts(g))
    atm_apply(m, f)
    s = ATM_SNAPS[name][1]
    m.steps(s, s0=1)
    e = atm_state_errs(m, g, "steps%d" % s)
    assert all(v < TOL_RUN for v in e.values()), e
    m.close()


def test_graph_replay_equals_eager(case):
    """steps() replays captured 50- and 10-step graphs (averaging every 100 steps): bitwise the eager launches."""
    name, acfg, g, f, m = case
    atm_apply(m, f)
    m.steps(230, s0=1)
    a, sa = m.get_state(), m.get_scalars()
    atm_apply(m, f)
    for s in range(1, 231):
        m.qgastep()
        m.atinvq()
        m.atqzbd()
        if (s - 1) % 100 == 0:
            m.lf_average()
    b, sb = m.get_state(), m.get_scalars()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert np.array_equal(sa, sb)
    assert all(np.isfinite(x).all() for x in a)


def test_full_size_steps_vs_oracle():
    """385 x 97 x 3 (examples/double_gyre_coupled): 300 steps against the CPU oracle, incl. three averagings."""
    from qgcm_hip import AtmosModel
    acfg = config.atmos_preset("cpl_natl5")
    g = load_golden("atm_natl5")
    f = atm_inputs(g, acfg)
    m = AtmosModel(acfg, ddynat=f["ddynat"])
    o = make_atm_oracle(acfg, g, f)
    atm_apply(m, f)
    atm_apply(o, f)
    m.steps(300, s0=1)
    o.steps(1, 300)
    for x, y in zip(m.get_state(), o.get_state()):
        assert relerr(x, y) < 1e-9
    m.close()
    o.close()


def test_coupled_main_loop_vs_reference():
    """Ocean and atmosphere handles stepped by qgcm_hip_coupled_steps in the reference's loop order
    (src/q-gcm.F:1220-1268, forcing held) against the coupled reference run."""
    from qgcm_hip import AtmosModel, OceanModel, coupled_steps
    g = load_golden("cpl_tiny")
    oc, at = config.preset("cpl_tiny"), config.atmos_preset("cpl_tiny")
    f = {k: g["in_" + k] for k in ("pa", "pam", "wekpa", "entat", "ddynat", "xan", "txis", "txin", "enis", "enin")}
    o = OceanModel(oc)
    a = AtmosModel(at, ddynat=f["ddynat"])
    o.set_p(g["in_po"], g["in_pom"])
    o.set_forcing(g["in_wekpo"], np.zeros_like(g["in_wekpo"]), np.zeros(oc.nlo - 1))
    atm_apply(a, f)
    nstr = int(g["nstr"])
    done = 0
    for upto in (12, 101):
        coupled_steps(o, a, done + 1, upto - done, nstr)
        done = upto
        for i, n in enumerate(("po", "pom", "qo", "qom")):
            assert relerr(o.get_state()[i], g["nt%d_%s" % (upto, n)]) < TOL_RUN, (upto, n)
        e = atm_state_errs(a, g, "nt%d" % upto)
        assert all(v < TOL_RUN for v in e.values()), (upto, e)
    o.close()
    a.close()


def test_atmosphere_continuity_monitors_vs_reference():
    """ermasa / emfrat of atinvq (MODULE monitor, src/atisubs.F:236-248: est1 = aiplay(k) - aiplay(k+1)), which the
    drop-in shim copies back with every pull: after each of six atmospheric steps against the coupled reference build
    (tests/golden/atm_tiny_monitors.npz, make_golden_monitors.py atmos), stand-alone calls and qgcm_hip_steps."""
    from qgcm_hip import AtmosModel
    acfg = config.atmos_preset("cpl_tiny")
    g, gm = load_golden("atm_tiny"), load_golden("atm_tiny_monitors")
    f = atm_inputs(g, acfg)
    for fused in (False, True):
        m = AtmosModel(acfg, ddynat=f["ddynat"])
        try:
            atm_apply(m, f)
            for s in range(1, 7):
                if fused:
                    m.steps(1, s0=s)
                else:
                    m.qgastep()
                    m.atinvq()
                    m.atqzbd()
                    if (s - 1) % 100 == 0:
                        m.lf_average()
                e, fr = m.get_monitors()
                sr = gm["step%d_scal" % s]
                scale = np.abs(sr[:acfg.nla - 1]) + np.abs(sr[acfg.nla - 1:2 * (acfg.nla - 1)])  # |dpiat| + |dpiatp| ~ esum
                # (est1, est2 are cancelling area integrals themselves: measured 1.7e-12 of |dpiat| + |dpiatp|)
                assert (np.abs(e - gm["step%d_ermasa" % s]) / scale).max() < 1e-11, (fused, s, e, gm["step%d_ermasa" % s])
                assert np.abs(fr - gm["step%d_emfrat" % s]).max() < 1e-11, (fused, s, fr, gm["step%d_emfrat" % s])
                assert np.abs(gm["step%d_emfrat" % s]).min() > 1e-5   # the fixture is not rounding noise
        finally:
            m.close()


@pytest.mark.parametrize("partition", [False, True])
def test_coupled_main_loop_full_size_vs_reference_sample(partition):
    """BASELINE configs[3] at FULL size through qgcm_hip_coupled_steps: NAtl 5 km ocean (961 x 961 x 3) and the
    385 x 97 x 3 atmosphere on this GPU, forcing held, against samples of the coupled reference build itself after
    nt = 9 (three ocean steps) and nt = 30 (ten) - tests/golden/make_golden_atmos.py coupled_fullsize.  Second run: the
    two handles on disjoint ranges of compute units (share_gpu -> qgcm_hip_set_cu_range), as bench.py's coupled figure
    runs them: where a workgroup runs changes no result."""
    from common import cpl_fullsize_errs, cpl_fullsize_inputs
    from qgcm_hip import AtmosModel, OceanModel, coupled_steps, share_gpu
    g = load_golden("cpl_natl5_sample")
    oc, at = config.preset("cpl_natl5"), config.atmos_preset("cpl_natl5")
    po, pom, wekpo, f = cpl_fullsize_inputs(g, oc, at)
    o = OceanModel(oc)
    a = AtmosModel(at, ddynat=f["ddynat"])
    o.set_p(po, pom)
    o.set_forcing(wekpo, np.zeros_like(wekpo), np.zeros(oc.nlo - 1))
    atm_apply(a, f)
    if partition:
        assert share_gpu(o, a) > 0
    done = 0
    for upto, tol in ((9, 1e-12), (30, 1e-11)):
        coupled_steps(o, a, done + 1, upto - done, int(g["nstr"]))
        done = upto
        e = cpl_fullsize_errs(o, a, g, upto)
        assert all(v < tol for v in e.values()), (upto, e)
    o.close()
    a.close()


def test_cu_range_arguments_and_bitwise_results():
    """qgcm_hip_set_cu_range: bad ranges are refused; 30 steps on 64 of the CUs, and back on all of them, give the bits
    of an unrestricted handle (graphs are rebuilt for the new stream)."""
    import torch
    from qgcm_hip import OceanModel, QgcmHipError, check, synth
    cfg = config.preset("box_small")
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    m, r = OceanModel(cfg), OceanModel(cfg)
    try:
        for bad in ((-1, 8), (0, ncu + 1), (ncu - 4, 8), (0, -2)):
            with pytest.raises(QgcmHipError, match="qgcm_hip_set_cu_range"):
                check(m.L.qgcm_hip_set_cu_range(m.h, *bad))
        po = synth.gaussian_eddy(cfg, noise=1e-2)
        tx, ty = synth.wind_stress(cfg)
        _, wek = synth.wekpo_from_tau(cfg, tx, ty)
        for mod in (m, r):
            mod.set_p(po, 0.99 * po)
            mod.set_forcing(wek, np.zeros_like(wek), np.zeros(cfg.nlo - 1))
        r.steps(60, s0=1)
        check(m.L.qgcm_hip_set_cu_range(m.h, ncu // 4, ncu // 4))
        m.steps(30, s0=1)
        check(m.L.qgcm_hip_set_cu_range(m.h, 0, 0))
        m.steps(30, s0=31)
        for x, y in zip(m.get_state(), r.get_state()):
            assert np.array_equal(x, y)
    finally:
        m.close()
        r.close()


def test_atmosphere_entry_points_reject_an_ocean_handle():
    from qgcm_hip import OceanModel, QgcmHipError, check
    m = OceanModel(config.preset("cyc_tiny"))
    with pytest.raises(QgcmHipError, match="atmos = 1"):
        check(m.L.qgcm_hip_qgastep(m.h))
    m.close()
