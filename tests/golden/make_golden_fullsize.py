#!/usr/bin/env python3
"""Pins the FULL-SIZE configurations to the reference itself (not only to the C restatement): a few steps of the
true reference (oracle/_ref/libqgcm_ref_<cfg>.so) from the deterministic synthetic inputs of qgcm_hip.synth at
  natl5   961 x 961 x 3   examples/double_gyre_ocean_only  (BASELINE configs[1]; the bench workload's grid)
  socn5   4609 x 577 x 3  examples/southern_ocean_ocean_only (BASELINE configs[2])
  natl1   4801 x 4801 x 3 src/parameters_data.F.NAtl.1km:45,50 + src/input.params.NAtl.1km with dta = 60 s, nstr = 3
                          (BASELINE configs[4]; SURVEY 8d: as shipped, nstr = 1 never steps the ocean); steps 1 and 2,
                          every 64th row / column; `python make_golden_fullsize.py natl1` (10 GB of host memory)
stored as every 16th (socn5: 32nd) row / column of po, pom, qo, qom + the constraint scalars after steps 1 and 4
(tests/golden/<cfg>_sample.npz, ~100 KB each).  The tests re-generate the inputs from the same synth code; a
strided sample of the inputs is stored too and must match bit for bit.  Build container only."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "q-gcm_amd", "python"))

import ref_binding  # noqa: E402
from qgcm_hip import config, synth  # noqa: E402

STRIDE = {"natl5": 16, "socn5": 32, "natl1": 64}
REFCFG = {"natl5": "box_natl5", "socn5": "cyc_socn5", "natl1": "box_natl1"}
STEPS = {"natl1": (1, 2)}


def make(name):
    cfg = config.preset(name)
    ST = STRIDE[name]
    ref_binding.build(REFCFG[name])
    r = ref_binding.RefLib(REFCFG[name])
    assert (r.nx, r.ny, r.nl, bool(r.cyclic)) == (cfg.nxpo, cfg.nypo, cfg.nlo, cfg.cyclic)
    ref_binding.set_threads(8)
    r.init(cfg.dxo, cfg.dto, cfg.delek, cfg.bccooc, cfg.ah2oc, cfg.ah4oc, cfg.hoc, cfg.gpoc)
    po = synth.gaussian_eddy(cfg, noise=1e-3)
    tx, ty = synth.wind_stress(cfg)
    _, wek = synth.wekpo_from_tau(cfg, tx, ty)
    r.set_p(po, po)
    r.set_forcing(wek, np.zeros_like(wek), np.zeros(cfg.nlo - 1))
    out = {"stride": np.array(ST), "in_po": po[::ST, ::ST].copy(), "in_wekpo": wek[::ST, ::ST].copy()}
    if cfg.cyclic:
        txis, txin = synth.tau_line_integrals(cfg, tx)
        r.set_cyc_forcing(txis, txin)
        out["in_txis"], out["in_txin"] = np.array(txis), np.array(txin)
    done = 0
    for s in STEPS.get(name, (1, 4)):
        r.steps(done + 1, s - done)
        done = s
        for n, v in zip(("po", "pom", "qo", "qom"), r.get_state()):
            out["steps%d_%s" % (s, n)] = v[::ST, ::ST].copy()
            out["steps%d_%s_max" % (s, n)] = np.array(np.abs(v).max())
        out["steps%d_scal" % s] = r.get_scalars()
    np.savez_compressed(os.path.join(HERE, name + "_sample.npz"), **out)
    print("wrote %s_sample.npz" % name)


if __name__ == "__main__":
    if len(sys.argv) == 2:
        make(sys.argv[1])
    else:  # one process per config: the reference libraries export identical symbols (natl1: on request only)
        for n in ("natl5", "socn5"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), n])
