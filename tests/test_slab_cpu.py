"""The y-slab decomposition (qgcm_hip.slab.SlabOcean + transports) on CPU: numpy slab
kernels stand in for the HIP ones; results are compared with the single-domain
oracle.  Covers LocalComm (virtual ranks) and torch.distributed/gloo world_size 2."""
import os
import sys

import numpy as np
import pytest
import torch

from common import FIELDS, make_oracle, preset, relerr
from numpy_slab import NumpySlab
from qgcm_hip import hostinit, synth
from qgcm_hip.slab import DistComm, LocalComm, SlabOcean, global_consts, partition


def problem(name):
    cfg = preset(name)
    o = make_oracle(cfg)
    consts = global_consts(cfg, o.helmholtz)
    po = synth.gaussian_eddy(cfg, noise=1e-2)
    pom = np.asfortranarray(0.99 * po)
    tx, ty = synth.wind_stress(cfg)
    _, wek = synth.wekpo_from_tau(cfg, tx, ty)
    ent = np.asfortranarray(1e-7 * np.cos(np.arange(cfg.nxpo) / 5.0)[:, None] * np.ones(cfg.nypo)[None, :])
    xon = np.zeros(cfg.nlo - 1)
    xon[0] = 5e2
    qo = hostinit.q_from_p(cfg, consts["amatoc"], consts["yporel"], consts["ddynoc"], po)
    qom = hostinit.q_from_p(cfg, consts["amatoc"], consts["yporel"], consts["ddynoc"], pom)
    scal = hostinit.constr(cfg, consts["amatoc"], po, pom)
    o.set_p(po, pom)
    o.set_forcing(wek, ent, xon)
    return cfg, o, consts, (po, pom, qo, qom, wek, ent, xon, scal)


def assemble(cfg, pieces):
    out = [np.zeros((cfg.nxpo, cfg.nypo, cfg.nlo)) for _ in range(4)]
    for g0, g1, fields in pieces:
        for dst, src in zip(out, fields):
            dst[:, g0 - 1:g1, :] = src
    return out


@pytest.mark.parametrize("name,nranks", [("box_tiny", 1), ("box_tiny", 2), ("box_tiny", 3), ("box_small", 4)])
def test_virtual_ranks_match_oracle(name, nranks):
    cfg, o, consts, init = problem(name)
    try:
        parts = partition(cfg.nypo, nranks)
        slabs = [NumpySlab(cfg, consts, g0, g1, r, nranks) for r, (g0, g1) in enumerate(parts)]
        so = SlabOcean(cfg, slabs, LocalComm(nranks))
        so.scatter_state(*init)
        nsteps = 30
        so.steps(nsteps, s0=1)
        o.steps(1, nsteps)
        got = assemble(cfg, so.gather_local())
        for f, x, y in zip(FIELDS, got, o.get_state()):
            assert relerr(x, y) < 1e-10, (f, nranks)
        s_ref = o.get_scalars()
        for sl in slabs:  # every rank carries identical constraint scalars
            assert np.abs(sl.get_scalars() - s_ref).max() / (cfg.xlo * cfg.ylo * np.abs(init[0]).max()) < 1e-12
            assert np.array_equal(sl.get_scalars(), slabs[0].get_scalars())
    finally:
        o.close()


def test_partition_covers_rows():
    for nyg, p in ((37, 3), (961, 8), (4801, 8), (97, 4)):
        parts = partition(nyg, p)
        assert parts[0][0] == 1 and parts[-1][1] == nyg
        assert all(parts[i][1] + 1 == parts[i + 1][0] for i in range(p - 1))
        sizes = [b - a + 1 for a, b in parts]
        assert max(sizes) - min(sizes) <= 1


def _gloo_worker(rank, world, port, name, nsteps, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, o, consts, init = problem(name)
    g0, g1 = partition(cfg.nypo, world)[rank]
    so = SlabOcean(cfg, [NumpySlab(cfg, consts, g0, g1, rank, world)], DistComm())
    so.scatter_state(*init)
    so.steps(nsteps, s0=1)
    (a, b, fields), = so.gather_local()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), g0=a, g1=b, po=fields[0], pom=fields[1], qo=fields[2], qom=fields[3],
             scal=so.slabs[0].get_scalars())
    if rank == 0:
        o.steps(1, nsteps)
        ref = o.get_state()
        np.savez(os.path.join(outdir, "ref.npz"), po=ref[0], pom=ref[1], qo=ref[2], qom=ref[3], scal=o.get_scalars())
    o.close()
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    world, nsteps, name = 2, 27, "box_tiny"
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker, args=(world, port, name, nsteps, str(tmp_path)), nprocs=world, join=True)
    cfg = preset(name)
    ref = np.load(tmp_path / "ref.npz")
    pieces = []
    for r in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % r))
        pieces.append((int(d["g0"]), int(d["g1"]), [d[f] for f in FIELDS]))
        assert np.abs(d["scal"] - ref["scal"]).max() / (cfg.xlo * cfg.ylo * np.abs(ref["po"]).max()) < 1e-12
    got = assemble(cfg, pieces)
    for f, x in zip(FIELDS, got):
        assert relerr(x, ref[f]) < 1e-10, f


def test_step_choreography_with_mixed_layer():
    """SlabOcean.step with the mixed layer on the slabs: `oml` in two halves around its own small all-gather BEFORE
    the tendency stage, then the usual stage / exchange sequence - checked on recording stand-ins (the kernels behind
    the stages are covered on the GPU; the order and the message plumbing are host logic)."""
    log = []

    class Rec:
        th_len, cst_len, halo_len, oml_len, oml_on = 7, 4, 5, 3, True

        def __init__(self, rank, n):
            self.rank, self.nranks = rank, n

        def new_buffer(self, n):
            return torch.zeros(int(n), dtype=torch.float64)

        def thomas_consts(self, dst):
            dst.fill_(self.rank + 1.0)

        def set_thomas_consts(self, gath):
            assert gath.numel() == self.cst_len * self.nranks and gath[self.cst_len].item() == 2.0

        def sync(self):
            pass

        def stage(self, n, a=None, b=None, c=None, flags=0):
            log.append((self.rank, n, None if a is None else a.numel(), flags))
            if n == 10:
                a.fill_(10.0 + self.rank)          # this slab's three sums
            if n == 11:
                assert a.numel() == 3 * self.nranks and a[3].item() == 11.0   # rank 1's sums arrived
            if n == 1:
                a.fill_(100.0 + self.rank)
            if n == 2:
                assert a.numel() == self.th_len * self.nranks and a[self.th_len].item() == 101.0
                for buf, val in ((b, 200.0 + self.rank), (c, 300.0 + self.rank)):
                    if buf is not None:
                        buf.fill_(val)
            if n == 3 and self.nranks > 1:
                if self.rank == 0:
                    assert a is None and b[0].item() == 201.0   # what rank 1 sent downwards
                else:
                    assert b is None and a[0].item() == 300.0   # what rank 0 sent upwards

    slabs = [Rec(0, 2), Rec(1, 2)]
    so = SlabOcean(preset("box_tiny"), slabs, LocalComm(2))
    assert so.oml_on
    so.step(1)   # (1 - 1) % 25 == 0: averaging flag in stage 3
    assert [e[1] for e in log] == [10, 10, 11, 11, 1, 1, 2, 2, 3, 3]
    assert [e[3] for e in log if e[1] == 3] == [1, 1]
    del log[:]
    so.step(2)
    assert [e[1] for e in log] == [10, 10, 11, 11, 1, 1, 2, 2, 3, 3] and [e[3] for e in log if e[1] == 3] == [0, 0]
