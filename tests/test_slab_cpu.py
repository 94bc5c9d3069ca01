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
