"""MI355X parity of the ocean mixed layer through the C ABI: qgcm_hip_oml against the golden
vectors of the TRUE reference and against the CPU oracle.

Bars: everything point-wise is bit exact (same expressions, contraction off); the only reordered arithmetic is
the global sum behind the mean entrainment (reference: sequential over i then j; here a fixed tree), so
entoc / xon(1) agree to rounding of that mean: 1e-13 of max|xfo|; sst itself is bitwise after one call."""
import numpy as np
import pytest

from common import OML_CASES, OML_SNAPS, FIELDS, load_golden, make_oracle, oml_config, oml_load, relerr
from qgcm_hip import OceanModel, oml_preset, preset, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case,cfgname", OML_CASES)
def test_one_call_vs_reference(case, cfgname):
    g, cfg = load_golden(case), preset(cfgname)
    om = oml_config(g)
    m = OceanModel(cfg)
    try:
        m.oml_init(om)
        oml_load(m, g, cfg, False)
        m.oml()
        sst, sstm = m.oml_get_state()
        ent, d = m.oml_get_diag()
        assert np.array_equal(sst, g["call_sst"])      # bit exact
        assert np.array_equal(sstm, g["call_sstm"])
        assert relerr(ent, g["call_entoc"]) < 1e-13
        assert d[1] == g["call_scal"][1]               # convecting fraction: an exact count
        assert abs(d[2] - g["call_scal"][2]) <= 1e-13 * abs(g["call_scal"][2])
        area = cfg.xlo * cfg.ylo * np.abs(ent).max()
        assert abs(d[0] - g["call_scal"][0]) <= 1e-13 * area
        if cfg.cyclic:
            assert np.allclose(d[3:], g["call_scal"][3:], rtol=1e-11, atol=0.0)
    finally:
        m.close()


@pytest.mark.parametrize("case,cfgname", OML_CASES)
def test_coupled_steps_vs_reference(case, cfgname):
    """qgcm_hip_steps with the mixed layer on = oml, qgostep, ocinvq, ocqbdy + averaging incl. sst
    (src/q-gcm.F:1232-1249, 1328-1366); 40 steps cross the averaging twice and the three sst buffers rotate."""
    g, cfg = load_golden(case), preset(cfgname)
    om = oml_config(g)
    m = OceanModel(cfg)
    try:
        m.oml_init(om)
        oml_load(m, g, cfg, False)
        done = 0
        for n in OML_SNAPS:
            m.steps(n - done, s0=done + 1)
            done = n
            sst, sstm = m.oml_get_state()
            ent, _ = m.oml_get_diag()
            assert relerr(sst, g["steps%d_sst" % n]) < 1e-13, n
            assert relerr(sstm, g["steps%d_sstm" % n]) < 1e-13, n
            assert relerr(ent, g["steps%d_entoc" % n]) < 1e-10, n
            for f, x in zip(FIELDS, m.get_state()):
                assert relerr(x, g["steps%d_%s" % (n, f)]) < 1e-10, (f, n)
    finally:
        m.close()


@pytest.mark.parametrize("cfgname,sb,nb", [("box_med", False, False), ("box_med", True, True), ("cyc_small", True, True)])
def test_graph_replay_and_oracle(cfgname, sb, nb):
    """120 steps (two 50-step HIP graphs with different sst-buffer rotations + eager remainder) against the CPU
    oracle, on grids that run the fused kernels; every boundary-option combination incl. those the reference
    example builds do not carry (the oracle restates them from src/omlsubs.F)."""
    cfg = preset(cfgname)
    om = oml_preset(cfg, sb_hflux=sb, nb_hflux=nb)
    o = make_oracle(cfg)
    m = OceanModel(cfg)
    try:
        po = synth.gaussian_eddy(cfg, noise=1e-3)
        sst, sstm, fnet, tx, ty = synth.mixed_layer_fields(cfg, om, seed=5)
        wekto, wekpo = synth.wekpo_from_tau(cfg, tx, ty)
        nl = cfg.nlo
        o.oml_init(om.hmoc, om.toc[0], om.toc[1], om.st2d, om.st4d, om.ycexp, om.rrcpoc, om.sb_hflux, om.tsbdy,
                   om.nb_hflux, om.tnbdy)
        m.oml_init(om)
        for mod in (o, m):
            mod.set_p(po, po)
            mod.set_forcing(wekpo, np.zeros_like(wekpo), np.zeros(nl - 1))
            if cfg.cyclic:
                txis, txin = synth.tau_line_integrals(cfg, tx)
                mod.set_cyc_forcing(txis, txin, np.zeros(nl - 1), np.zeros(nl - 1))
        o.oml_set(sst, sstm, fnet, wekto, tx, ty)
        m.oml_set_state(sst, sstm)
        m.oml_set_forcing(fnet, wekto, tx, ty)
        o.steps_oml(1, 120)
        m.steps(120, s0=1)
        a, b, e, _ = o.oml_get()
        sa, sb_ = m.oml_get_state()
        ent, _ = m.oml_get_diag()
        assert relerr(sa, a) < 1e-12 and relerr(sb_, b) < 1e-12
        assert relerr(ent, e) < 1e-9
        for f, x, y in zip(FIELDS, m.get_state(), o.get_state()):
            assert relerr(x, y) < 1e-9, f
    finally:
        m.close()
        o.close()


def test_one_for_one_calls_equal_steps():
    """The call sequence of the reference main program issued call by call (oml, qgostep, ocinvq, ocqbdy and the
    averaging block when mod(s-1,25) == 0 - as the Fortran shim does) is bitwise qgcm_hip_steps."""
    g, cfg = load_golden("oml_box_tiny"), preset("box_tiny")
    om = oml_config(g)
    ma, mb = OceanModel(cfg), OceanModel(cfg)
    try:
        for m in (ma, mb):
            m.oml_init(om)
            oml_load(m, g, cfg, False)
        ma.steps(30, s0=1)
        for s in range(1, 31):
            mb.oml(); mb.qgostep(); mb.ocinvq(); mb.ocqbdy()
            if (s - 1) % 25 == 0:
                mb.lf_average()
        for x, y in zip(ma.get_state() + ma.oml_get_state(), mb.get_state() + mb.oml_get_state()):
            assert np.array_equal(x, y)
        assert np.array_equal(ma.get_scalars(), mb.get_scalars())
    finally:
        ma.close()
        mb.close()


def test_oml_needs_init_and_whole_domain():
    cfg = preset("box_small")
    m = OceanModel(cfg)
    try:
        with pytest.raises(Exception, match="qgcm_hip_oml_init has not been called"):
            m.oml()
    finally:
        m.close()


@pytest.mark.parametrize("cfgname,nranks,sb,nb", [("box_small", 1, False, False), ("box_small", 2, True, True), ("box_med", 3, False, True),
                                                  ("cyc_small", 2, True, True), ("cyc_med", 3, False, False)])
def test_mixed_layer_on_y_slabs(cfgname, nranks, sb, nb):
    """The mixed layer on y-slabs (virtual ranks on this one GPU): `oml` in two halves around one all-gather of three
    numbers per rank (the mean entrainment), the T row below every slab recomputed locally (entoc averages two T rows
    onto a p row), xon(1) and the line integrals in the step message, edge rows of sst in the halo messages.  60 steps
    (two averagings, every sst buffer rotation) against the whole-domain handle: sst bitwise-close, fields <= 1e-10."""
    import torch
    from qgcm_hip import hostinit
    from qgcm_hip.slab import HipSlab, LocalComm, SlabOcean, global_consts, partition
    cfg = preset(cfgname)
    om = oml_preset(cfg, sb_hflux=sb, nb_hflux=nb)
    m = OceanModel(cfg)
    slabs = []
    try:
        consts = global_consts(cfg, None if cfg.cyclic else m.helmholtz)
        po = synth.gaussian_eddy(cfg, noise=1e-3)
        pom = np.asfortranarray(0.995 * po)
        sst, sstm, fnet, tx, ty = synth.mixed_layer_fields(cfg, om, seed=7)
        wekto, wekpo = synth.wekpo_from_tau(cfg, tx, ty)
        nl = cfg.nlo
        zero2, xon = np.zeros_like(wekpo), np.zeros(nl - 1)
        m.oml_init(om)
        m.set_p(po, pom)
        m.set_forcing(wekpo, zero2, xon)
        cyc = (synth.tau_line_integrals(cfg, tx) + (np.zeros(nl - 1), np.zeros(nl - 1))) if cfg.cyclic else None
        if cyc:
            m.set_cyc_forcing(*cyc)
        m.oml_set_state(sst, sstm)
        m.oml_set_forcing(fnet, wekto, tx, ty)
        st = m.get_state()
        scal = m.get_scalars()
        parts = partition(cfg.nypo, nranks)
        slabs = [HipSlab(cfg, consts, g0, g1, r, nranks, sync_each_call=True) for r, (g0, g1) in enumerate(parts)]
        for sl in slabs:
            sl.oml_init(om)
        so = SlabOcean(cfg, slabs, LocalComm(nranks, after=torch.cuda.synchronize))
        so.scatter_state(st[0], st[1], st[2], st[3], wekpo, zero2, xon, scal)
        for sl in slabs:
            if cyc:
                sl.set_cyc_forcing(*cyc)
            sl.oml_set_state(sst, sstm)
            sl.oml_set_forcing(fnet, wekto, tx, ty)
        for nst in (1, 59):
            so.steps(nst)
            m.steps(nst)
            got = [np.zeros((cfg.nxpo, cfg.nypo, cfg.nlo)) for _ in range(4)]
            for g0, g1, fields in so.gather_local():
                for dst, src in zip(got, fields):
                    dst[:, g0 - 1:g1, :] = src
            for f, x, y in zip(FIELDS, got, m.get_state()):
                assert relerr(x, y) < 1e-10, (f, nst)
            ga, gb = m.oml_get_state()
            for sl in slabs:
                la, lb = sl.oml_get_state()
                t1 = min(sl.g1, cfg.nypo - 1)  # owned T rows g0..t1 (global, 1-based)
                loc = slice(sl.g0 - 1 - sl.joff, t1 - sl.joff)
                assert relerr(la[:, loc], ga[:, sl.g0 - 1:t1]) < 1e-12, (sl.rank, nst)
                assert relerr(lb[:, loc], gb[:, sl.g0 - 1:t1]) < 1e-12, (sl.rank, nst)
        for sl in slabs:
            assert np.array_equal(sl.get_scalars(), slabs[0].get_scalars())
    finally:
        for sl in slabs:
            sl.close()
        m.close()
