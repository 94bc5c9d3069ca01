"""Worker of tests/test_gpu_multiproc.py: ONE PROCESS PER SLAB on the same GPU (torch.distributed, gloo - RCCL refuses
duplicate devices), each compared bitwise with the same decomposition run as virtual ranks inside this process.
What this covers and the virtual-rank tests cannot: asynchronous stage calls (no synchronisation after every call),
set-up exchanges between separate handles in separate processes, the DistComm transports.
usage (under torch.distributed.run): mp_slab_worker.py <preset> <halo: allgather|p2p> [oml|early]
(early: stage 4 of the next step before stage 3 - the order of the overlapped halo exchange - in this process's run)"""
import os
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "q-gcm_amd", "python"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from qgcm_hip import hostinit, oml_preset, preset, synth  # noqa: E402
from qgcm_hip.slab import DistComm, HipSlab, LocalComm, SlabOcean, global_consts, partition  # noqa: E402


def main():
    rank, P = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=P)
    cfg = preset(sys.argv[1])
    po = synth.gaussian_eddy(cfg, noise=1e-2)
    pom = np.asfortranarray(0.99 * po)
    tx, ty = synth.wind_stress(cfg)
    _, wek = synth.wekpo_from_tau(cfg, tx, ty)
    consts = global_consts(cfg)  # homogeneous solutions come from SlabOcean.homsol()
    qo = hostinit.q_from_p(cfg, consts["amatoc"], consts["yporel"], consts["ddynoc"], po)
    qom = hostinit.q_from_p(cfg, consts["amatoc"], consts["yporel"], consts["ddynoc"], pom)
    scal = hostinit.constr(cfg, consts["amatoc"], po, pom)
    ent, xon = np.zeros_like(wek), np.zeros(cfg.nlo - 1)
    parts = partition(cfg.nypo, P)
    # the reference: all P slabs as virtual ranks in this process
    vs = [HipSlab(cfg, consts, g0, g1, r, P, sync_each_call=True) for r, (g0, g1) in enumerate(parts)]
    with_oml = "oml" in sys.argv[3:]  # mixed layer on: one more small all-gather per step
    if with_oml:
        om = oml_preset(cfg, sb_hflux=True, nb_hflux=False)
        sst, sstm, fnet, tx, ty = synth.mixed_layer_fields(cfg, om, seed=7)
        wekto, wek = synth.wekpo_from_tau(cfg, tx, ty)
        for v in vs:
            v.oml_init(om)
    vo = SlabOcean(cfg, vs, LocalComm(P, after=torch.cuda.synchronize))
    cyc = bool(cfg.cyclic)  # channel: the homogeneous solutions (functions of y) came with global_consts
    txis, txin = synth.tau_line_integrals(cfg, tx) if cyc else (0.0, 0.0)
    vh = None if cyc else vo.homsol()
    vo.scatter_state(po, pom, qo, qom, wek, ent, xon, scal)
    for v in vs:
        if cyc:
            v.set_cyc_forcing(txis, txin)
        if with_oml:
            v.oml_set_state(sst, sstm)
            v.oml_set_forcing(fnet, wekto, tx, ty)
    # this process's own slab, exchanges through torch.distributed
    g0, g1 = parts[rank]
    slab = HipSlab(cfg, consts, g0, g1, rank, P, device=0)
    if with_oml:
        slab.oml_init(om)
    torch.cuda.set_stream(torch.cuda.ExternalStream(slab.stream_ptr, device=slab.device))
    so = SlabOcean(cfg, [slab], DistComm(halo_via_all_gather=(sys.argv[2] != "p2p")))
    so.early_tend = "early" in sys.argv[3:]
    ok = True
    if not cyc:
        dh = so.homsol()
        ok = np.array_equal(dh["aipohs"], vh["aipohs"]) and np.array_equal(slab.ochom_local, vs[rank].ochom_local)
    so.scatter_state(po, pom, qo, qom, wek, ent, xon, scal)
    if cyc:
        slab.set_cyc_forcing(txis, txin)
    if with_oml:
        slab.oml_set_state(sst, sstm)
        slab.oml_set_forcing(fnet, wekto, tx, ty)
    for nst in (1, 1, 28):  # crosses the averaging after step 26
        so.steps(nst)
        vo.steps(nst)
        ok = ok and all(np.array_equal(x, y) for x, y in zip(slab.get_state(), vs[rank].get_state()))
        ok = ok and np.array_equal(slab.get_scalars(), vs[rank].get_scalars())
        if with_oml:
            ok = ok and all(np.array_equal(x, y) for x, y in zip(slab.oml_get_state(), vs[rank].oml_get_state()))
    fin = all(np.isfinite(x).all() for x in slab.get_state())
    t = torch.tensor([1.0 if (ok and fin) else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("MP_SLAB_RESULT", "OK" if t.item() > 0.5 else "MISMATCH", flush=True)
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream())
    slab.close()
    for v in vs:
        v.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
