"""Validity scan `valids` (src/valsubs.F:43-627, ocean part): the CPU restatement against the
verdicts of the TRUE reference on crafted states either side of every criterion
(tests/golden/make_golden_valids.py)."""
import numpy as np
import pytest

from common import load_golden, make_oracle
from qgcm_hip import oml_preset, preset

G = load_golden("valids_box_tiny")
NAMES = [str(n) for n in G["names"]]


def load_case(model, name, is_oracle):
    zT = np.zeros_like(G[name + "_sst"])
    po, qo = G[name + "_po"], G[name + "_qo"]
    model.set_state(po, po, qo, qo)
    if is_oracle:
        zP = np.zeros((po.shape[0], po.shape[1]), order="F")
        model.oml_set(G[name + "_sst"], G[name + "_sst"], zT, G[name + "_wekto"], zP, zP)
    else:
        model.oml_set_state(G[name + "_sst"], G[name + "_sst"])
        model.oml_set_forcing(None, G[name + "_wekto"], None, None)


@pytest.mark.parametrize("name", NAMES)
def test_verdict_matches_reference(name):
    cfg = preset("box_tiny")
    om = oml_preset(cfg)
    o = make_oracle(cfg)
    try:
        o.oml_init(om.hmoc, om.toc[0], om.toc[1], om.st2d, om.st4d, om.ycexp, om.rrcpoc)
        load_case(o, name, True)
        ok, out = o.valids(G[name + "_dtopoc"])
        assert ok == bool(G[name + "_solnok"]), out
        assert out[0] == G[name + "_po"].min() and out[1] == G[name + "_po"].max()
        assert out[4] == G[name + "_sst"].min() and out[7] == G[name + "_wekto"].max()
    finally:
        o.close()
