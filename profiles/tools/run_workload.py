"""Steps one workload on cuda:0 for rocprofv3 (profiles/collect.sh): `python3 profiles/tools/run_workload.py <what> [steps]`
  socn5 / natl5 / <any ocean preset>   whole-domain handle, Gaussian-eddy IC + synthetic wind (+ k_copy calibration launches)
  natl5_oml / <preset>_oml             the same with the ocean mixed layer on the device (k_oml_step, k_oml_entoc)
  atmos                                385 x 97 x 3 atmospheric channel of double_gyre_coupled
  natl1_slabs                          NAtl 1 km as eight y-slabs (virtual ranks on this one GPU, every kernel HBM-cold)
  natl1_one_slab                       the same set-up, then ONE middle slab stepped alone: its fields stay in the
                                       Infinity Cache between its kernels, as on its own GPU of eight (exchange buffers
                                       held at the last real step's values; no collective is timed)
  natl5_one_slab_of_8 / _of_4 / _of_2  the headline basin (961 x 961 x 3) cut into 8 / 4 / 2 slabs, one middle slab stepped
                                       alone the same way: what one GPU of an N-GPU run of BASELINE's metric computes per
                                       step (the two exchanges not included)
Eager launches (no graphs) so that every kernel shows in the trace with its own name."""
import os
import sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "q-gcm_amd", "python"))
import numpy as np  # noqa: E402

what = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
if what == "atmos":
    from qgcm_hip import AtmosModel, atmos_preset, synth
    at = atmos_preset("cpl_natl5")
    f = synth.atmos_fields(at)
    a = AtmosModel(at, ddynat=f["ddynat"])
    a.set_p(f["pa"], f["pam"])
    a.set_forcing(f["wekpa"], f["entat"], f["xan"], f["txis"], f["txin"], f["enis"], f["enin"])
    if os.environ.get("ATM_CUS"):   # the atmosphere's share of a coupled run (qgcm_hip_set_cu_range)
        from qgcm_hip.lib import check
        check(a.L.qgcm_hip_set_cu_range(a.h, 0, int(os.environ["ATM_CUS"])))
    a.steps(200, s0=1)
    ms = a.time_steps(1000, s0=201)
    print("atmosphere 385x97x3: %.2f us per step (graph replay)" % (1e3 * ms / 1000))
    for s in range(1201, 1201 + nsteps):   # eager, named kernels
        a.qgastep(); a.atinvq(); a.atqzbd()
    a.sync()
elif what in ("natl1_slabs", "natl1_one_slab", "natl5_slabs", "natl5_one_slab_of_8", "natl5_one_slab_of_4", "natl5_one_slab_of_2"):
    import torch
    from qgcm_hip import hostinit, preset, synth
    from qgcm_hip.slab import HipSlab, LocalComm, SlabOcean, global_consts, partition
    cfg, P = preset(what[:5]), (int(what.rsplit("_", 1)[1]) if "_of_" in what else 8)
    consts = global_consts(cfg)
    po = synth.gaussian_eddy(cfg)
    tx, ty = synth.wind_stress(cfg)
    _, wek = synth.wekpo_from_tau(cfg, tx, ty)
    z2 = np.zeros((cfg.nxpo, cfg.nypo), order="F")
    qo = hostinit.q_from_p(cfg, consts["amatoc"], consts["yporel"], consts["ddynoc"], po)
    scal = hostinit.constr(cfg, consts["amatoc"], po, po)
    slabs = [HipSlab(cfg, consts, g0, g1, r, P, sync_each_call=True) for r, (g0, g1) in enumerate(partition(cfg.nypo, P))]
    so = SlabOcean(cfg, slabs, LocalComm(P, after=torch.cuda.synchronize))
    so.homsol()
    so.scatter_state(po, po, qo, qo, wek, z2, np.zeros(cfg.nlo - 1), scal)
    if what.endswith("_slabs"):
        so.steps(min(nsteps, 45), s0=1)
    else:
        import time
        so.steps(4, s0=1)          # real steps of all eight slabs: the exchange buffers hold consistent values
        torch.cuda.synchronize()
        r = P // 2
        x = slabs[r]
        x.sync_each_call = False

        st = torch.cuda.ExternalStream(x.stream_ptr)
        mine = so.th_gath[r][r * x.th_len:(r + 1) * x.th_len]
        state0, scal0 = x.get_state(), x.get_scalars()

        def one(n):
            with torch.cuda.stream(st):
                for _ in range(n):     # the stages of SlabOcean.step for this slab alone (s with (s - 1) % 25 != 0)
                    x.stage(1, so.th_send[r])
                    mine.copy_(so.th_send[r])   # its own summary is current, the other ranks' stay at the last real step
                    x.stage(2, so.th_gath[r], so.h_to_lo[r], so.h_to_hi[r])
                    x.stage(3, so.h_from_lo[r], so.h_from_hi[r], None, 0)
            x.sync()
        # A slab coupled to neighbours that stand still is not a stable system (it leaves the neighbourhood of the real
        # solution after 5-6 steps): time 3 steps at a time, one untimed step after every restore of the state.
        nch, per, tot = max(5, min(nsteps, 200) // 3), 3, 0.0
        for _ in range(nch):
            x.set_state(*state0); x.set_scalars(scal0)
            one(1)
            t0 = time.perf_counter()
            one(per)
            tot += time.perf_counter() - t0
        print("%s, slab %d of %d (rows %d..%d) alone: %.1f us per step over %d x %d steps (wall clock, launches queued ahead); slab finite: %s"
              % (cfg.name, r, P, x.g0, x.g1, 1e6 * tot / (nch * per), nch, per, all(np.isfinite(f).all() for f in x.get_state())))
        # the same four steps as ONE captured graph (no host in the loop: what qgcm_hip_slab_steps' graph mode replays,
        # minus the two collectives), HIP events around each replay
        try:
            x.set_state(*state0); x.set_scalars(scal0)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(4):
                    x.stage(1, so.th_send[r])
                    mine.copy_(so.th_send[r])
                    x.stage(2, so.th_gath[r], so.h_to_lo[r], so.h_to_hi[r])
                    x.stage(3, so.h_from_lo[r], so.h_from_hi[r], None, 0)
            e0, e1, tg = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), 0.0
            for _ in range(nch):
                x.set_state(*state0); x.set_scalars(scal0)
                with torch.cuda.stream(st):
                    e0.record(); gr.replay(); e1.record()
                e1.synchronize()
                tg += e0.elapsed_time(e1)
            print("%s, slab %d of %d alone, four steps as one captured graph: %.1f us per step (HIP events, %d replays)"
                  % (cfg.name, r, P, 1e3 * tg / (4 * nch), nch))
        except Exception as e:  # noqa: BLE001 - the eager figure above stands
            print("graph replay of the slab stages not available: %r" % (e,))
    torch.cuda.synchronize()
    print("finite", all(np.isfinite(f).all() for _, _, fs in so.gather_local() for f in fs))
else:
    from qgcm_hip import OceanModel, preset, synth
    with_oml = what.endswith("_oml")       # e.g. natl5_oml: the same with the ocean mixed layer on the device (`call oml`)
    what = what[:-4] if with_oml else what
    cfg = preset(what)
    m = OceanModel(cfg)
    if with_oml:
        from qgcm_hip import oml_preset
        om = oml_preset(cfg)
        sst, sstm, fnet, txo, tyo = synth.mixed_layer_fields(cfg, om)
        wekto, _ = synth.wekpo_from_tau(cfg, txo, tyo)
        m.oml_init(om)
        m.oml_set_state(sst, sstm)
        m.oml_set_forcing(fnet, wekto, txo, tyo)
    po = synth.gaussian_eddy(cfg)
    tx, ty = synth.wind_stress(cfg)
    _, wek = synth.wekpo_from_tau(cfg, tx, ty)
    m.set_p(po, po)
    m.set_forcing(wek, np.zeros_like(wek), np.zeros(cfg.nlo - 1))
    if cfg.cyclic:
        m.set_cyc_forcing(*synth.tau_line_integrals(cfg, tx))
    m.steps(150, s0=1)
    ms = m.time_steps(400, s0=151)
    print("%s%s: %.2f us per step (graph replay)" % (what, " with the mixed layer" if with_oml else "", 1e3 * ms / 400))
    m.profile_steps(nsteps, s0=551)   # eager launches with named kernels (brackets between them)
    m.copy_bandwidth(1 << 30, 3)      # 1 GiB k_copy launches: the calibration of the FETCH_SIZE correction
    print("finite", bool(np.isfinite(m.get_state()[0]).all()))
