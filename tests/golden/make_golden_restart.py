#!/usr/bin/env python3
"""Restart-file fixture written by the reference TOOLCHAIN (amdflang, the reference's modules) in the record
sequence of SUBROUTINE resave (src/q-gcm.F:3053-3088, ocean_only build):  tyrs | po,pom | sst,sstm | ast,astm |
hmixa,hmixam.  resave itself lives in the main program file and cannot be linked into the harness library, so
oracle/ref/qgcm_ref_oml.F90 restates its six WRITE statements; what the fixture pins is the wire format (4-byte
record markers, fp64, Fortran order).  Run in the build container only:

    python tests/golden/make_golden_restart.py > /dev/null
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "q-gcm_amd", "python"))


def main():
    import ref_binding as rb
    from qgcm_hip import config, synth
    rb.build("box_tiny")
    cfg = config.preset("box_tiny")
    r = rb.RefLib("box_tiny")
    assert r.atmos_dims() == (cfg.nxta, cfg.nyta)
    r.init(cfg.dxo, cfg.dto, cfg.delek, cfg.bccooc, cfg.ah2oc, cfg.ah4oc, cfg.hoc, cfg.gpoc)
    om = config.oml_preset(cfg)
    po = synth.gaussian_eddy(cfg, noise=1e-3)
    pom = np.asfortranarray(0.97 * po)
    sst, sstm, fnet, tx, ty = synth.mixed_layer_fields(cfg, om, seed=2)
    wekto, _ = synth.wekpo_from_tau(cfg, tx, ty)
    r.set_p(po, pom)
    r.oml_set(sst, sstm, fnet, wekto, tx, ty)
    r.write_restart(os.path.join(HERE, "restart_box_tiny.bin"), 12.5)
    np.savez_compressed(os.path.join(HERE, "restart_box_tiny_fields.npz"), po=po, pom=pom, sst=sst, sstm=sstm)


if __name__ == "__main__":
    main()
