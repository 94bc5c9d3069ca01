"""Pins the C restatement (oracle/qgcm_oracle.c) to the golden vectors that were
generated from the true reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import oracle_binding as ob
from common import (CONFIG_NAMES, FIELDS, SNAPS, apply_inputs, load_golden, load_snapshot, make_oracle,
                    preset, relerr, scal_err, state_errs)

TOL_POINT = 0.0      # qgostep / ocqbdy / init q: same association order, no FMA -> bit exact
TOL_SOLVE = 2e-13    # anything through the FFT-based Helmholtz solve (different FFT factorisation)


@pytest.fixture(scope="module", params=CONFIG_NAMES)
def case(request):
    cfg = preset(request.param)
    g = load_golden(request.param)
    o = make_oracle(cfg, g["c_yporel"])
    yield cfg, g, o
    o.close()


def test_yporel_formula(case):
    cfg, g, _ = case
    assert np.array_equal(cfg.yporel(), g["c_yporel"])


def test_constants(case):
    cfg, g, o = case
    c = o.get_consts()
    assert np.array_equal(c["amatoc"], g["c_amatoc"])
    assert np.array_equal(c["bd2oc"], g["c_bd2oc"])
    assert c["aoc"] == float(g["c_aoc"])
    for k in ("ctl2moc", "ctm2loc", "rdm2oc"):
        # (two eigen-solvers of the nlo x nlo stratification matrix: the rounding grows with its size - 6.2e-15 at nlo = 6)
        assert relerr(c[k], g["c_" + k]) < 5e-15 * max(1.0, cfg.nlo / 3.0), k


def test_homog(case):
    cfg, g, o = case
    h = o.get_homog()
    if cfg.cyclic:
        big = max(np.abs(g["h_hc1soc"]).max(), np.abs(g["h_hc2noc"]).max())
        for k in ("pch1oc", "pch2oc", "pbhoc", "aipcho", "hbsioc", "aipbho"):
            assert relerr(h[k], g["h_" + k]) < 1e-13, k
        for k in ("hc1soc", "hc2soc", "hc1noc", "hc2noc"):  # two of them are ~exp(-ylo/Rd): compare absolutely
            assert np.abs(h[k] - g["h_" + k]).max() / big < 1e-13, k
    else:
        for k in ("ochom", "aipohs", "cdiffo", "cdhoc"):
            assert relerr(h[k], g["h_" + k]) < 1e-13, k


def test_init_q_and_scalars(case):
    cfg, g, o = case
    apply_inputs(o, g, cfg)
    e = state_errs(o, g, "init")
    assert all(v <= TOL_POINT for v in e.values()), e
    assert scal_err(o, g, "init", cfg) < 1e-15


def test_single_calls(case):
    cfg, g, o = case
    if "qgostep_po" not in g:
        pytest.skip("per-call snapshots only stored for the tiny grids")
    apply_inputs(o, g, cfg)
    o.qgostep()
    e = state_errs(o, g, "qgostep")
    assert all(v <= TOL_POINT for v in e.values()), e
    load_snapshot(o, g, "qgostep")
    o.ocinvq()
    e = state_errs(o, g, "ocinvq")
    assert e["pom"] == 0.0 and e["qo"] == 0.0 and e["qom"] == 0.0 and e["po"] < TOL_SOLVE, e
    assert scal_err(o, g, "ocinvq", cfg) < 1e-13
    load_snapshot(o, g, "ocinvq")
    o.ocqbdy()
    e = state_errs(o, g, "ocqbdy")
    assert all(v <= TOL_POINT for v in e.values()), e


def test_whole_steps(case):
    cfg, g, o = case
    apply_inputs(o, g, cfg)
    done = 0
    for s in SNAPS[cfg.name]:
        o.steps(done + 1, s - done)
        done = s
        e = state_errs(o, g, "steps%d" % s)
        # free-running comparison: rounding differences of the solver grow slowly
        assert all(v < 5e-12 for v in e.values()), (s, e)
        assert scal_err(o, g, "steps%d" % s, cfg) < 1e-12


def test_helmholtz(case):
    cfg, g, o = case
    if "helm_rhs" not in g:
        pytest.skip("the *_ah2 fixtures carry no Helmholtz vectors (the solver does not see ah2oc)")
    assert relerr(o.helmholtz(g["helm_rhs"], g["helm_boc"]), g["helm_sol"]) < TOL_SOLVE
    assert relerr(o.helmholtz(g["helm_rhs"], g["helm_boc0"]), g["helm_sol0"]) < TOL_SOLVE


def test_fftpack_vectors():
    g = load_golden("fftpack_eigmod")
    for n in (3, 4, 5, 14, 47, 59, 95, 959):
        assert relerr(ob.dsint(g["dsint_in_%d" % n]), g["dsint_out_%d" % n]) < 1e-14, n
    for n in (4, 6, 9, 10, 48, 96, 384, 385):
        f = ob.rfftf(g["drfft_in_%d" % n])
        assert relerr(f, g["drfftf_out_%d" % n]) < 1e-14, n
        assert relerr(ob.rfftb(g["drfftf_out_%d" % n]), g["drfftb_out_%d" % n]) < 1e-14, n


def test_fftpack_vectors_at_the_long_row_lengths():
    """dsint at n = 4799 (NAtl 1 km rows) and drfftf / drfftb at n = 960, 4608 (SOcn 5 km), 4800 - FFTPACK itself,
    tests/golden/make_golden.py fft_long."""
    g = load_golden("fftpack_long")   # (two different factorisations of a length-4800 transform: 1.6e-14 measured)
    assert relerr(ob.dsint(g["dsint_in_4799"]), g["dsint_out_4799"]) < 5e-14
    for n in (960, 4608, 4800):
        assert relerr(ob.rfftf(g["drfft_in_%d" % n]), g["drfftf_out_%d" % n]) < 5e-14, n
        assert relerr(ob.rfftb(g["drfftf_out_%d" % n]), g["drfftb_out_%d" % n]) < 5e-14, n


def test_sponge_ramp_formula():
    """hostinit.sponge_ramp (what a host without the reference main program uses) against the ramp the reference
    build set itself (src/q-gcm.F:1154-1168); its exp() need not round like numpy's."""
    from qgcm_hip import hostinit
    for name in ("box_tiny_spl", "cyc_tiny_spl"):
        g = load_golden(name)
        cfg = preset(name)
        assert (float(g["in_c1spl"]), float(g["in_lspl"])) == (cfg.c1_spl, cfg.l_spl)
        assert np.abs(hostinit.sponge_ramp(cfg) - g["in_rspl"]).max() < 4e-16 * np.abs(g["in_rspl"]).max()


def test_dsint_is_self_inverse():
    # FFTPACK: dsint(dsint(x)) = 2(n+1) x  (fft.doc:342-344)
    rng = np.random.default_rng(3)
    for n in (7, 48, 959):
        x = rng.standard_normal(n)
        assert relerr(ob.dsint(ob.dsint(x)) / (2.0 * (n + 1)), x) < 1e-13


def test_eigmod_vectors():
    g = load_golden("fftpack_eigmod")
    f0 = float(g["eig_fnot"])
    for nl in (2, 3, 4):
        tag = "eig%d" % nl
        e = ob.eigmod(g[tag + "_gp"], g[tag + "_h"], f0)
        assert np.array_equal(e["amatoc"], g[tag + "_amatoc"])
        for k in ("rdm2oc", "ctl2moc", "ctm2loc"):
            assert relerr(e[k], g[tag + "_" + k]) < 1e-14, (nl, k)
        # cm2l * cl2m = identity (the reference prints this check, eigmode.f:478-505)
        prod = e["ctm2loc"].T @ e["ctl2moc"].T
        assert np.abs(prod - np.eye(nl)).max() < 1e-14


@pytest.mark.parametrize("name", ["natl5", "socn5"])
def test_full_size_sample_from_the_reference(name):
    """The C restatement at the FULL size of BASELINE configs[1] / [2] against the reference itself
    (tests/golden/make_golden_fullsize.py: every 16th / 32nd row and column after steps 1 and 4)."""
    from qgcm_hip import synth
    cfg = preset(name)
    g = load_golden(name + "_sample")
    st = int(g["stride"])
    po = synth.gaussian_eddy(cfg, noise=1e-3)
    tx, ty = synth.wind_stress(cfg)
    _, wek = synth.wekpo_from_tau(cfg, tx, ty)
    assert np.array_equal(po[::st, ::st], g["in_po"]) and np.array_equal(wek[::st, ::st], g["in_wekpo"])
    o = make_oracle(cfg)
    try:
        o.set_p(po, po)
        o.set_forcing(wek, np.zeros_like(wek), np.zeros(cfg.nlo - 1))
        if cfg.cyclic:
            o.set_cyc_forcing(float(g["in_txis"]), float(g["in_txin"]))
        done = 0
        for s in (1, 4):
            o.steps(done + 1, s - done)
            done = s
            for i, n in enumerate(FIELDS):
                x = o.get_state()[i]
                assert np.abs(x[::st, ::st] - g["steps%d_%s" % (s, n)]).max() < 1e-12 * float(g["steps%d_%s_max" % (s, n)]), (s, n)
    finally:
        o.close()


def test_natl5_long_run_160_steps_vs_reference_sample():
    """The C restatement at NAtl 5 km after 160 ocean steps against the reference itself (tests/golden/
    natl5_long_sample.npz): SURVEY 8(d)'s 1e-9, or ten times the reference's own 8- vs 1-thread spread."""
    from qgcm_hip import synth
    cfg = preset("natl5")
    g = load_golden("natl5_long_sample")
    st = int(g["stride"])
    po = synth.gaussian_eddy(cfg, noise=1e-3)
    tx, ty = synth.wind_stress(cfg)
    _, wek = synth.wekpo_from_tau(cfg, tx, ty)
    assert np.array_equal(po[::st, ::st], g["in_po"])
    o = make_oracle(cfg)
    try:
        o.set_p(po, po)
        o.set_forcing(wek, np.zeros_like(wek), np.zeros(cfg.nlo - 1))
        o.steps(1, 160)
        for i, n in enumerate(FIELDS):
            err = float(np.abs(o.get_state()[i][::st, ::st] - g["steps160_" + n]).max() / float(g["steps160_%s_max" % n]))
            assert err < max(1e-9, 10.0 * float(g["spread160_" + n])), (n, err, float(g["spread160_" + n]))
    finally:
        o.close()
