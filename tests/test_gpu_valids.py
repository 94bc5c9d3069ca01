"""MI355X parity of the validity scan through the C ABI: qgcm_hip_valids against the verdicts of
the TRUE reference (crafted states either side of every criterion) and, number for number, against the CPU oracle.
Bar: bit exact - min, max and the quarter-integer weighted counts do not depend on the order of evaluation."""
import numpy as np
import pytest

from common import make_oracle
from qgcm_hip import OceanModel, oml_preset, preset, synth
from test_valids_oracle import G, NAMES, load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", NAMES)
def test_verdict_and_numbers(name):
    cfg = preset("box_tiny")
    om = oml_preset(cfg)
    o = make_oracle(cfg)
    m = OceanModel(cfg)
    try:
        o.oml_init(om.hmoc, om.toc[0], om.toc[1], om.st2d, om.st4d, om.ycexp, om.rrcpoc)
        m.oml_init(om)
        load_case(o, name, True)
        load_case(m, name, False)
        m.set_dtopoc(G[name + "_dtopoc"])
        ok, out = m.valids()
        ok_o, out_o = o.valids(G[name + "_dtopoc"])
        assert ok == bool(G[name + "_solnok"])      # the reference's verdict
        assert np.array_equal(out, out_o)           # bit exact against the restatement
    finally:
        m.close()
        o.close()


def test_full_size_after_steps_and_without_mixed_layer():
    """NAtl 5 km after 60 steps: numbers equal numpy's own min/max of the pulled state; without the mixed
    layer the sst / wekto entries keep the reference's initial +/-1e30 and do not enter the verdict."""
    cfg = preset("natl5")
    m = OceanModel(cfg)
    try:
        po = synth.gaussian_eddy(cfg)
        tx, ty = synth.wind_stress(cfg)
        _, wek = synth.wekpo_from_tau(cfg, tx, ty)
        m.set_p(po, po)
        m.set_forcing(wek, np.zeros_like(wek), np.zeros(cfg.nlo - 1))
        m.steps(60, s0=1)
        ok, out = m.valids()
        p, _, q, _ = m.get_state()
        assert ok
        assert out[0] == p.min() and out[1] == p.max() and out[2] == q.min() and out[3] == q.max()
        assert out[4] == 1e30 and out[5] == -1e30 and out[6] == 1e30 and out[7] == -1e30
        rg = 1.0 / np.asarray(cfg.gpoc)
        eta = [rg[k] * (p[:, :, k + 1] - p[:, :, k]) for k in range(cfg.nlo - 1)]
        ht = cfg.hoc[0] - eta[0]
        hi = cfg.hoc[1] - eta[1] + eta[0]
        hb = cfg.hoc[2] + eta[1] - 0.0
        assert (out[8], out[9]) == (ht.min(), ht.max())
        assert (out[10], out[11]) == (hi.min(), hi.max())
        assert (out[12], out[13]) == (hb.min(), hb.max())
        assert np.all(out[14:] == 0.0)
    finally:
        m.close()


@pytest.mark.parametrize("name", ["box_tiny5", "cyc_tiny6"])
def test_more_than_four_layers(name):
    """valids at five and six layers (k_valids_scan / k_valids_final are templates on nlo = 2 .. 8) after ten steps:
    extrema equal numpy's own of the pulled state, the thickness extrema of the top / inner / bottom layers
    (src/valsubs.F:390-430) from the same expressions, no mixed layer."""
    cfg = preset(name)
    m = OceanModel(cfg)
    try:
        po = synth.gaussian_eddy(cfg, noise=1e-3)
        tx, ty = synth.wind_stress(cfg)
        _, wek = synth.wekpo_from_tau(cfg, tx, ty)
        m.set_p(po, po)
        m.set_forcing(wek, np.zeros_like(wek), np.zeros(cfg.nlo - 1))
        if cfg.cyclic:
            m.set_cyc_forcing(*synth.tau_line_integrals(cfg, tx))
        m.steps(10, s0=1)
        ok, out = m.valids()
        p, _, q, _ = m.get_state()
        assert ok
        assert out[0] == p.min() and out[1] == p.max() and out[2] == q.min() and out[3] == q.max()
        rg = 1.0 / np.asarray(cfg.gpoc)
        eta = [rg[k] * (p[:, :, k + 1] - p[:, :, k]) for k in range(cfg.nlo - 1)]
        nl = cfg.nlo
        h = [cfg.hoc[0] - eta[0]] + [cfg.hoc[k] - eta[k] + eta[k - 1] for k in range(1, nl - 1)] + [cfg.hoc[nl - 1] + eta[nl - 2] - 0.0]
        assert (out[8], out[9]) == (h[0].min(), h[0].max())
        assert (out[10], out[11]) == (min(x.min() for x in h[1:-1]), max(x.max() for x in h[1:-1]))   # all inner layers together
        assert (out[12], out[13]) == (h[-1].min(), h[-1].max())
        assert np.all(out[14:] == 0.0)
    finally:
        m.close()
