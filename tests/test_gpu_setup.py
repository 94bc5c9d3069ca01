"""Start-up / restart arithmetic and the progress sample on the device: qgcm_hip_init_from_p
(constr + qcomp + ocqbdy / atqzbd + merqcy), qgcm_hip_wekpo_from_tau, qgcm_hip_prsamp - against golden vectors of the
true reference.  Needs an MI355X."""
import numpy as np
import pytest

from common import (ATM_CASES, CONFIG_NAMES, FIELDS, atm_inputs, atm_scal_err, atm_state_errs, load_golden, preset, relerr,
                    scal_err, state_errs)
from qgcm_hip import config, hostinit, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CONFIG_NAMES)
def test_init_from_p_on_the_device(name):
    """src/q-gcm.F:711-731 on the device: q bitwise the reference's, constraint scalars to rounding."""
    from qgcm_hip import OceanModel
    cfg, g = preset(name), load_golden(name)
    m = OceanModel(cfg)
    try:
        m.init_from_p(g["in_po"], g["in_pom"])
        e = state_errs(m, g, "init")
        assert all(v == 0.0 for v in e.values()), e
        assert scal_err(m, g, "init", cfg) < 1e-13
    finally:
        m.close()


@pytest.mark.parametrize("name", [c[0] for c in ATM_CASES])
def test_atmosphere_init_from_p_on_the_device(name):
    """src/q-gcm.F:711, 738-749: constr, qcomp with the topography under layer 1, atqzbd (line 470 as written), merqcy."""
    from qgcm_hip import AtmosModel
    acfg = config.atmos_preset(dict(ATM_CASES)[name])
    g = load_golden(name)
    f = atm_inputs(g, acfg)
    m = AtmosModel(acfg, ddynat=f["ddynat"])
    try:
        m.init_from_p(f["pa"], f["pam"])
        e = atm_state_errs(m, g, "init")
        assert all(v == 0.0 for v in e.values()), e
        assert atm_scal_err(m, g, "init", acfg) < 1e-13
    finally:
        m.close()


@pytest.mark.parametrize("name", ["box_small", "cyc_small"])
def test_wekpo_from_tau_on_the_device(name):
    """Ekman pumping from the stress on the device (src/xfosubs.F:566-645): the steps it drives are bitwise those
    driven by the host-side restatement - which is bitwise the REFERENCE's own `call xforc` on this very stress
    (tests/golden/setup_ref.npz, tests/test_host_side.py::test_wekpo_from_tau_is_the_reference_xforc)."""
    from qgcm_hip import OceanModel
    cfg, g = preset(name), load_golden(name)
    tx, ty = synth.wind_stress(cfg)
    ty = np.asfortranarray(1e-5 * np.sin(np.arange(cfg.nxpo) / 3.0)[:, None] * np.cos(np.arange(cfg.nypo) / 5.0)[None, :])
    if cfg.cyclic:
        ty[-1, :] = ty[0, :]
    _, wek = synth.wekpo_from_tau(cfg, tx, ty)
    m = OceanModel(cfg)
    try:
        outs = []
        for dev in (False, True):
            m.set_p(g["in_po"], g["in_pom"])
            m.set_forcing(None if dev else wek, np.zeros_like(wek), np.zeros(cfg.nlo - 1))
            if dev:
                m.set_forcing(np.full_like(wek, 7.0), None, None)  # poison: the device result has to replace it
                m.wekpo_from_tau(tx, ty)
            if cfg.cyclic:
                m.set_cyc_forcing(*synth.tau_line_integrals(cfg, tx))
            m.steps(3, s0=1)
            outs.append(m.get_state())
        for x, y in zip(*outs):
            assert np.array_equal(x, y)
    finally:
        m.close()


@pytest.mark.parametrize("name", ["box_small", "cyc_tiny"])
def test_prsamp_numbers(name):
    """The ocean numbers of prsamp (src/q-gcm.F:1933-2066): spot values bitwise the device state, layer averages =
    xintp * ocnorm - and both against the TRUE reference after the same 5 steps (tests/golden/setup_ref.npz:
    the reference's xintp on its own po, qo; make_golden_setup.py)."""
    from qgcm_hip import OceanModel
    cfg, g = preset(name), load_golden(name)
    gs = load_golden("setup_ref")
    m = OceanModel(cfg)
    try:
        m.set_p(g["in_po"], g["in_pom"])
        m.set_forcing(g["in_wekpo"], g["in_entoc"], g["in_xon"])
        m.steps(5, s0=1)
        po, _, qo, _ = m.get_state()
        s = m.prsamp()
        ic, jc = (cfg.nxpo + 1) // 2 - 1, (cfg.nypo + 1) // 2 - 1
        assert np.array_equal(s["po_centre"], po[ic, jc, :]) and np.array_equal(s["qo_centre"], qo[ic, jc, :])
        ocnorm = 1.0 / (cfg.nxto * cfg.nyto)
        for k in range(cfg.nlo):
            assert abs(s["pavgoc"][k] - hostinit.xintp(po[:, :, k]) * ocnorm) < 1e-13 * np.abs(po).max()
            assert abs(s["qavgoc"][k] - hostinit.xintp(qo[:, :, k]) * ocnorm) < 1e-13 * np.abs(qo).max()
        assert s["sstmin"] == 1e30 and s["sstmax"] == -1e30   # no device mixed layer
        pmx, qmx = float(gs[name + "_pomax"]), float(gs[name + "_qomax"])
        assert np.abs(s["pavgoc"] - gs[name + "_pavg"]).max() < 1e-12 * pmx
        assert np.abs(s["qavgoc"] - gs[name + "_qavg"]).max() < 1e-12 * qmx
        assert np.abs(s["po_centre"] - gs[name + "_po_centre"]).max() < 1e-11 * pmx
        assert np.abs(s["qo_centre"] - gs[name + "_qo_centre"]).max() < 1e-11 * qmx
    finally:
        m.close()
