"""Pins the atmosphere mode of the C restatement (oracle/qgcm_oracle.c: qgastep / atinvq / atqzbd)
to golden vectors from a coupled build of the true reference (tests/golden/make_golden_atmos.py).  CPU only."""
import numpy as np
import pytest

import oracle_binding as ob
from common import (ATM_CASES, ATM_SNAPS, atm_apply, atm_inputs, atm_load_snapshot, atm_scal_err, atm_state_errs,
                    load_golden, make_atm_oracle, make_oracle, relerr)
from qgcm_hip import config

TOL_POINT = 0.0      # qgastep / atqzbd / init q: same association order, no FMA -> bit exact
TOL_SOLVE = 2e-13    # anything through the FFT-based Helmholtz solve (different FFT factorisation)


@pytest.fixture(scope="module", params=[c[0] for c in ATM_CASES])
def case(request):
    name = request.param
    acfg = config.atmos_preset(dict(ATM_CASES)[name])
    g = load_golden(name)
    f = atm_inputs(g, acfg)
    o = make_atm_oracle(acfg, g, f)
    yield name, acfg, g, f, o
    o.close()


def test_grid_formulas(case):
    _, acfg, g, _, _ = case
    assert np.array_equal(acfg.yporel(), g["c_yparel"])  # src/q-gcm.F:401-403


def test_constants(case):
    _, acfg, g, _, o = case
    c = o.get_consts()
    assert np.array_equal(c["amatoc"], g["c_amatat"])
    assert np.array_equal(c["bd2oc"], g["c_bd2at"])
    assert c["aoc"] == float(g["c_aat"])
    assert relerr(c["rdm2oc"], g["c_rdm2at"]) < 5e-15
    # mode matrices: the reference's come from LAPACK (case 'Atmosphere': DTREVC scaling, sign from the Schur
    # vectors); the restatement's sign rule reproduces them for these parameters
    for k, kk in (("ctl2moc", "ctl2mat"), ("ctm2loc", "ctm2lat")):
        assert relerr(c[k], g["c_" + kk]) < 5e-15, k


def test_homog(case):
    _, acfg, g, _, o = case
    h = o.get_homog()
    big = max(np.abs(g["h_hc1sat"]).max(), np.abs(g["h_hc2nat"]).max())
    for k, kk in (("pch1oc", "pch1at"), ("pch2oc", "pch2at"), ("pbhoc", "pbhat"), ("aipcho", "aipcha"),
                  ("hbsioc", "hbsiat"), ("aipbho", "aipbha")):
        assert relerr(h[k], g["h_" + kk]) < 1e-13, k
    for k, kk in (("hc1soc", "hc1sat"), ("hc2soc", "hc2sat"), ("hc1noc", "hc1nat"), ("hc2noc", "hc2nat")):
        assert np.abs(h[k] - g["h_" + kk]).max() / big < 1e-13, k


def test_init_q_and_scalars(case):
    _, acfg, g, f, o = case
    atm_apply(o, f)
    e = atm_state_errs(o, g, "init")
    assert all(v <= TOL_POINT for v in e.values()), e
    assert atm_scal_err(o, g, "init", acfg) < 1e-14


def test_single_calls(case):
    name, acfg, g, f, o = case
    if "qgastep_pa" not in g:
        pytest.skip("per-call snapshots only stored for the small grids")
    atm_apply(o, f)
    o.qgastep()
    e = atm_state_errs(o, g, "qgastep")
    assert all(v <= TOL_POINT for v in e.values()), e
    atm_load_snapshot(o, g, "qgastep")
    o.atinvq()
    assert relerr(o.get_bsums(), g["bsums1"]) == 0.0  # ajisat, ajinat, ap5sat, ap5nat: same summation order
    e = atm_state_errs(o, g, "atinvq")
    assert e["pam"] == 0.0 and e["qa"] == 0.0 and e["qam"] == 0.0 and e["pa"] < TOL_SOLVE, e
    assert atm_scal_err(o, g, "atinvq", acfg) < 1e-13
    atm_load_snapshot(o, g, "atinvq")
    o.atqzbd()
    e = atm_state_errs(o, g, "atqzbd")
    assert all(v <= TOL_POINT for v in e.values()), e


def test_atqzbd_reads_row_two_for_the_top_layer():
    """src/vorsubs.F:470 uses pa(i,2,nla) in the southern value of the top layer where every other layer uses the
    boundary point: the golden vectors only agree with that form."""
    acfg = config.atmos_preset("cpl_tiny")
    g = load_golden("atm_tiny")
    f = atm_inputs(g, acfg)
    o = make_atm_oracle(acfg, g, f)
    atm_load_snapshot(o, g, "atinvq")
    o.atqzbd()
    qa, pa = o.get_state()[2], g["atinvq_pa"]
    nl = acfg.nla
    f0Ac = acfg.fnot * g["c_amatat"][nl - 1, nl - 1]
    as_written = g["atqzbd_qa"][:, 0, nl - 1]
    assert np.array_equal(qa[:, 0, nl - 1], as_written)
    symmetric = as_written - (f0Ac * pa[:, 0, nl - 1] - f0Ac * pa[:, 1, nl - 1])
    assert np.abs(symmetric - as_written).max() > 1e-6 * np.abs(as_written).max()
    o.close()


def test_whole_steps(case):
    name, acfg, g, f, o = case
    atm_apply(o, f)
    done = 0
    for s in ATM_SNAPS[name]:
        o.steps(done + 1, s - done)
        done = s
        e = atm_state_errs(o, g, "steps%d" % s)
        assert all(v < 5e-12 for v in e.values()), (s, e)
        assert atm_scal_err(o, g, "steps%d" % s, acfg) < 1e-12


def test_helmholtz(case):
    name, acfg, g, f, o = case
    if "helm_rhs" in g:
        rhs = g["helm_rhs"]
        st = 1
    else:
        rng = np.random.default_rng(int(g["helm_seed"]))
        rhs = np.asfortranarray(rng.standard_normal((acfg.nxpa, acfg.nypa)))
        rhs[-1, :] = rhs[0, :]
        st = int(g["stride"])
    b = g["c_bd2at"]
    assert relerr(o.helmholtz(rhs, b - g["c_rdm2at"][1])[::st, ::st], g["helm_sol"]) < TOL_SOLVE
    assert relerr(o.helmholtz(rhs, b - g["c_rdm2at"][0])[::st, ::st], g["helm_sol0"]) < TOL_SOLVE


def test_coupled_main_loop():
    """ocean + atmosphere in the reference's loop order (src/q-gcm.F:1220-1268) with the forcing held."""
    g = load_golden("cpl_tiny")
    oc, at = config.preset("cpl_tiny"), config.atmos_preset("cpl_tiny")
    f = {k: g["in_" + k] for k in ("pa", "pam", "wekpa", "entat", "ddynat", "xan", "txis", "txin", "enis", "enin")}
    o = make_oracle(oc)
    a = make_atm_oracle(at, g, f)
    o.set_p(g["in_po"], g["in_pom"])
    o.set_forcing(g["in_wekpo"])
    atm_apply(a, f)
    nstr = int(g["nstr"])
    nt = 0
    for upto in (12, 101):
        while nt < upto:
            nt += 1
            if nt % nstr == 1:
                o.steps((nt - 1) // nstr + 1, 1)
            a.steps(nt, 1)
        for i, n in enumerate(("po", "pom", "qo", "qom")):
            assert relerr(o.get_state()[i], g["nt%d_%s" % (upto, n)]) < 5e-12, (upto, n)
        e = atm_state_errs(a, g, "nt%d" % upto)
        assert all(v < 5e-12 for v in e.values()), (upto, e)
    o.close()
    a.close()


def test_coupled_main_loop_full_size_sample():
    """BASELINE configs[3] at full size: NAtl 5 km ocean (961 x 961 x 3) under the 385 x 97 x 3 atmosphere, the coupled
    loop with the forcing held after three ocean steps (nt = 9) against samples of the coupled reference build itself."""
    from common import cpl_fullsize_errs, cpl_fullsize_inputs
    g = load_golden("cpl_natl5_sample")
    oc, at = config.preset("cpl_natl5"), config.atmos_preset("cpl_natl5")
    po, pom, wekpo, f = cpl_fullsize_inputs(g, oc, at)
    o = make_oracle(oc)
    a = ob.AtmosOracle(at.nxpa, at.nypa, at.nla, at.fnot, at.beta, at.dxa, at.dta, at.bccoat, at.ah4at, at.hat, at.gpat,
                       at.yporel(), f["ddynat"])
    o.set_p(po, pom)
    o.set_forcing(wekpo)
    atm_apply(a, f)
    nstr = int(g["nstr"])
    for nt in range(1, 10):
        if nt % nstr == 1:
            o.steps((nt - 1) // nstr + 1, 1)
        a.steps(nt, 1)
    e = cpl_fullsize_errs(o, a, g, 9)
    assert all(v < 1e-11 for v in e.values()), e
    o.close()
    a.close()
