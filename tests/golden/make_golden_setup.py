#!/usr/bin/env python3
"""Reference values for the set-up / monitoring arithmetic done on the device, from the TRUE
reference (oracle/build_ref.sh builds; the harness calls the reference's own routines):
  <cfg>_wekto, <cfg>_wekpo   `call xforc` of an ocean-only build (src/xfosubs.F:566-683) on the wind stress of
                             tests/test_gpu_setup.py::test_wekpo_from_tau_on_the_device (box_small, cyc_small)
  <cfg>_pavg, <cfg>_qavg     the layer averages pavgoc / qavgoc of prsamp / monnc_comp (xintp * ocnorm) after 5 steps
                             from the golden inputs of the configuration (box_small, cyc_tiny), + the centre values
-> tests/golden/setup_ref.npz.  Build container only:   python tests/golden/make_golden_setup.py > /dev/null"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "q-gcm_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def stress(cfg):
    """the stress of the device test: double-gyre / channel taux + a wavy tauy"""
    from qgcm_hip import synth
    tx, _ = synth.wind_stress(cfg)
    ty = np.asfortranarray(1e-5 * np.sin(np.arange(cfg.nxpo) / 3.0)[:, None] * np.cos(np.arange(cfg.nypo) / 5.0)[None, :])
    if cfg.cyclic:
        ty[-1, :] = ty[0, :]
    return tx, ty


def one(name):
    import ref_binding
    from qgcm_hip import config
    cfg = config.preset(name)
    ref_binding.build(name, force=True)   # (the harness gained ref_xforc / ref_layer_avgs in round 3)
    r = ref_binding.RefLib(name)
    r.init(cfg.dxo, cfg.dto, cfg.delek, cfg.bccooc, cfg.ah2oc, cfg.ah4oc, cfg.hoc, cfg.gpoc)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    out = {}
    if name in ("box_small", "cyc_small"):
        tx, ty = stress(cfg)
        wt, wp = np.zeros((cfg.nxto, cfg.nyto), order="F"), np.zeros((cfg.nxpo, cfg.nypo), order="F")
        ref_binding.run_big_stack(r.lib.ref_xforc, dp(np.asfortranarray(tx)), dp(np.asfortranarray(ty)), dp(wt), dp(wp))
        out[name + "_wekto"], out[name + "_wekpo"] = wt, wp
    if name in ("box_small", "cyc_tiny"):
        g = np.load(os.path.join(HERE, name + ".npz"))
        r.set_p(g["in_po"], g["in_pom"])
        r.set_forcing(g["in_wekpo"], g["in_entoc"], g["in_xon"])
        r.steps(1, 5)
        pa, qa = np.zeros(cfg.nlo), np.zeros(cfg.nlo)
        r.lib.ref_layer_avgs(dp(pa), dp(qa))
        po, _, qo, _ = r.get_state()
        ic, jc = (cfg.nxpo + 1) // 2 - 1, (cfg.nypo + 1) // 2 - 1
        out[name + "_pavg"], out[name + "_qavg"] = pa, qa
        out[name + "_po_centre"], out[name + "_qo_centre"] = po[ic, jc, :].copy(), qo[ic, jc, :].copy()
        out[name + "_pomax"], out[name + "_qomax"] = np.array(np.abs(po).max()), np.array(np.abs(qo).max())
    np.savez(os.path.join(HERE, "_setup_%s.npz" % name), **out)


if __name__ == "__main__":
    if len(sys.argv) == 2:
        one(sys.argv[1])
    else:  # one process per configuration: the reference libraries export identical symbols
        allv = {}
        for n in ("box_small", "cyc_small", "cyc_tiny"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), n], stdout=subprocess.DEVNULL)
            f = os.path.join(HERE, "_setup_%s.npz" % n)
            allv.update(dict(np.load(f)))
            os.remove(f)
        np.savez_compressed(os.path.join(HERE, "setup_ref.npz"), **allv)
        sys.stderr.write("wrote setup_ref.npz: %s\n" % sorted(allv))
