"""Phase durations of one stamped row kernel (slot 0 of the in-kernel stamps, lib built with -DQG_STAMPS; see stamps.py).
run: QGCM_HIP_LIB=/tmp/lib_stamps.so python profiles/tools/stamps_row.py [preset] [nstamps]"""
import ctypes as C, os, sys
sys.path.insert(0, "q-gcm_amd/python"); sys.path.insert(0, ".")
import numpy as np
from qgcm_hip import preset, synth
from qgcm_hip.model import OceanModel
name = sys.argv[1] if len(sys.argv) > 1 else "socn5"
if name == "box4800":   # a box basin with the rows of NAtl 1 km (nxto = 4800) and the 600 rows of one of its eight slabs
    from qgcm_hip.config import OceanConfig
    cfg = OceanConfig("box4800", 602, 77, 600, 75, 8, 3, dxo=1.0e3, dta=60.0, bccooc=0.1, ah4oc=(5.0e7,) * 3,
                      fnot=9.37456e-05, beta=1.7536e-11, cyclic=False)
else:
    cfg = preset(name)
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 7
slot = int(sys.argv[3]) if len(sys.argv) > 3 else 0
m = OceanModel(cfg, device=0)
po = synth.gaussian_eddy(cfg)
tx, ty = synth.wind_stress(cfg)
_, wek = synth.wekpo_from_tau(cfg, tx, ty)
m.set_p(po, po); m.set_forcing(wek, np.zeros_like(wek), np.zeros(cfg.nlo - 1))
if cfg.cyclic:
    m.set_cyc_forcing(*synth.tau_line_integrals(cfg, tx))
m.steps(60, s0=1); m.sync()
buf = np.zeros((4, 4096, 10), dtype=np.int64)
for rep in range(3):
    m.steps(20); m.sync()
    m.L.qgcm_hip_debug_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
b = buf[slot]; live = b[:, 0] > 0
b = b[live]; t0 = b[:, 0].min()
print("%d workgroups stamped (of the launch's first 4096); entry spread: p10 %.2f med %.2f p90 %.2f max %.2f us; last stamp max %.2f us" % (
    len(b), *[np.percentile((b[:, 0] - t0) / 100.0, q) for q in (10, 50, 90, 100)], (b[:, 1:ns].max() - t0) / 100.0))
for i in range(1, ns):
    ok = (b[:, i] > 0) & (b[:, i - 1] > 0)
    if not ok.any():
        continue
    d = (b[ok, i] - b[ok, i - 1]) / 100.0
    print("  phase %d: min %6.2f  p10 %6.2f  med %6.2f  p90 %6.2f  max %6.2f us" % (i, d.min(), np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max()))
last = b[:, 1:ns].max(axis=1)
d = (last - b[:, 0]) / 100.0
print("  whole workgroup: min %.2f med %.2f p90 %.2f max %.2f us" % (d.min(), np.median(d), np.percentile(d, 90), d.max()))
