"""Phase timeline of the hot kernels from in-kernel stamps (lib built with -DQG_STAMPS).
build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DQG_STAMPS -shared -o /tmp/lib_stamps.so q-gcm_amd/csrc/qgcm_hip.hip
run:   QGCM_HIP_LIB=/tmp/lib_stamps.so python profiles/tools/stamps.py [preset]   (from the repo root, on the GPU box)"""
import ctypes as C, os, sys
sys.path.insert(0, "q-gcm_amd/python"); sys.path.insert(0, ".")
import numpy as np
from qgcm_hip import preset
from qgcm_hip.model import OceanModel
import bench
cfg = preset(sys.argv[1] if len(sys.argv) > 1 else "natl5")
po, wek = bench.synthetic_inputs(cfg)
m = OceanModel(cfg, device=0)
m.set_p(po, po); m.set_forcing(wek, np.zeros_like(wek), np.zeros(cfg.nlo - 1))
m.steps(60, s0=1)
m.sync()
NK, NB, NS = 4, 4096, 10
names = {0: ("k_dst64 fwd", ["entry", "front done (rows in, M-DFT, LDS)", "back done (transform)", "stores issued", "stores drained"]),
         1: ("k_thomas", ["entry", "rows in + local fwd", "barrier1", "scan+barrier2", "fwd rerun + local bwd + barrier3", "scan+barrier4", "bwd rerun", "stores issued", "stores drained"]),
         2: ("k_dst64_unpack", ["entry", "front done", "back done", "barrier", "combine + stores issued", "stores drained"]),
         3: ("k_tend", ["entry", "s1", "s2", "s3", "s4", "s5", "s6", "s7 (stores issued)", "s8 drained"])}
buf = np.zeros((NK, NB, NS), dtype=np.int64)
for rep in range(3):
    # eager single step (odd count: eager path)
    if os.environ.get("STAMPS_EAGER") == "1":
        m.L.qgcm_hip_qgostep(m.h); m.L.qgcm_hip_ocinvq(m.h); m.L.qgcm_hip_ocqbdy(m.h)
    else:
        m.steps(20)   # one captured 20-step block: the stamps of its last step remain
    m.sync()
    m.L.qgcm_hip_debug_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
for k in range(NK):
    b = buf[k]
    live = b[:, 0] > 0
    if not live.any():
        continue
    t0 = b[live, 0].min()
    nm, labels = names[k]
    print("== %s: %d workgroups stamped" % (nm, live.sum()))
    for i, lab in enumerate(labels):
        v = b[live, i]
        ok = v > 0
        if not ok.any():
            continue
        us = (v[ok] - t0) / 100.0
        print("  %-40s min %6.2f  p10 %6.2f  med %6.2f  p90 %6.2f  max %6.2f us" % (lab, us.min(), np.percentile(us, 10), np.median(us), np.percentile(us, 90), us.max()))
# gaps between consecutive kernels of the step (absolute 100 MHz clock): last drained stamp -> next kernel's first entry
order = [3, 0, 1, 2]
last = {3: 8, 0: 4, 1: 8, 2: 5}
ends = {}
for k in order:
    b = buf[k]; live = b[:, 0] > 0
    ends[k] = (b[live, 0].min(), b[live, last[k]].max())
for a, b in zip(order, order[1:]):
    print("gap %s -> %s: %.2f us   (kernel %s spans %.2f us first entry to last drain)" % (names[a][0], names[b][0], (ends[b][0] - ends[a][1]) / 100.0, names[a][0], (ends[a][1] - ends[a][0]) / 100.0))
print("step span (tend entry -> unpack drained): %.2f us" % ((ends[2][1] - ends[3][0]) / 100.0))
# the slowest workgroups of k_thomas at barrier1 and of the others at their last stamp
for k, idx in ((1, 2), (0, 4), (2, 5)):
    b = buf[k]; live = np.where(b[:, 0] > 0)[0]
    t0 = b[live, 0].min()
    order_ = live[np.argsort(-(b[live, idx]))][:8]
    print(names[k][0], "slowest WGs at stamp", idx, [(int(w), round(float(b[w, idx] - t0) / 100.0, 2), round(float(b[w, 0] - t0) / 100.0, 2)) for w in order_])
