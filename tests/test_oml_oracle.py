"""Ocean mixed layer: the CPU restatement of oml / omladf (oracle/qgcm_oracle.c) against the
golden vectors of the TRUE reference (tests/golden/make_golden_oml.py; three reference builds: box with no-flux
walls, box with -Dsb_hflux, cyclic with -Dnb_hflux)."""
import numpy as np
import pytest

from common import OML_CASES, OML_SNAPS, FIELDS, load_golden, make_oracle, oml_config, oml_load, relerr
from qgcm_hip import preset


@pytest.mark.parametrize("case,cfgname", OML_CASES)
def test_one_call_is_bitwise_the_reference(case, cfgname):
    g, cfg = load_golden(case), preset(cfgname)
    om = oml_config(g)
    o = make_oracle(cfg)
    try:
        o.oml_init(om.hmoc, om.toc[0], om.toc[1], om.st2d, om.st4d, om.ycexp, om.rrcpoc, om.sb_hflux, om.tsbdy,
                   om.nb_hflux, om.tnbdy)
        oml_load(o, g, cfg, True)
        o.oml()
        sst, sstm, ent, scal = o.oml_get()
        # the reference's omlsubs.F is compiled without OpenMP (flang rejects its REDUCTION(-:)), so even the
        # global sums run in the order restated here
        assert np.array_equal(sst, g["call_sst"])
        assert np.array_equal(sstm, g["call_sstm"])
        assert np.array_equal(ent, g["call_entoc"])
        assert scal[1] == g["call_scal"][1]                       # convecting fraction
        assert abs(scal[2] - g["call_scal"][2]) <= 1e-14 * abs(g["call_scal"][2])
        # xon(1) is the area integral of a field whose mean was removed: compare to area * max|entoc|
        area = cfg.xlo * cfg.ylo * np.abs(ent).max()
        assert abs(scal[0] - g["call_scal"][0]) <= 1e-14 * area
        if cfg.cyclic:
            assert np.allclose(scal[3:], g["call_scal"][3:], rtol=1e-13, atol=0.0)
    finally:
        o.close()


@pytest.mark.parametrize("case,cfgname", OML_CASES)
def test_coupled_steps(case, cfgname):
    """oml, qgostep, ocinvq, ocqbdy (+ averaging incl. sst), src/q-gcm.F:1232-1249,1328-1366."""
    g, cfg = load_golden(case), preset(cfgname)
    om = oml_config(g)
    o = make_oracle(cfg)
    try:
        o.oml_init(om.hmoc, om.toc[0], om.toc[1], om.st2d, om.st4d, om.ycexp, om.rrcpoc, om.sb_hflux, om.tsbdy,
                   om.nb_hflux, om.tnbdy)
        oml_load(o, g, cfg, True)
        done = 0
        for n in OML_SNAPS:
            o.steps_oml(done + 1, n - done)
            done = n
            sst, sstm, ent, _ = o.oml_get()
            assert relerr(sst, g["steps%d_sst" % n]) < 1e-14
            assert relerr(ent, g["steps%d_entoc" % n]) < 1e-11
            for f, x in zip(FIELDS, o.get_state()):
                assert relerr(x, g["steps%d_%s" % (n, f)]) < 1e-12, (f, n)
    finally:
        o.close()
