#!/usr/bin/env python3
"""Golden verdicts of the validity scan `valids` (src/valsubs.F:43-627, ocean part) from the
TRUE reference (oracle/build_ref.sh, box_tiny build).  Run in the build container only:

    python tests/golden/make_golden_valids.py > /dev/null     # the reference prints its diagnostics

The routine's extremes are local variables, so only its verdict `solnok` is observable: the fixture holds
crafted states on both sides of every criterion (|po| >= 1e4, |qo| >= 0.05, |sst| >= 75, |wekto| >= 1e-3,
more than 20 % of the basin with a layer thinner than 100 m) together with the reference's verdict."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "q-gcm_amd", "python"))


def cases(cfg):
    from qgcm_hip import config, synth
    om = config.oml_preset(cfg)
    nx, ny, nl = cfg.nxpo, cfg.nypo, cfg.nlo
    po0 = synth.gaussian_eddy(cfg, noise=1.0e-3)
    qo0 = np.asfortranarray(1.0e-3 * np.sin(np.arange(nx)[:, None, None] / 7.0) * np.ones((1, ny, nl)))
    sst0, _, _, tx, ty = synth.mixed_layer_fields(cfg, om, seed=3)
    wek0, _ = synth.wekpo_from_tau(cfg, tx, ty)
    flat = np.zeros((nx, ny), order="F")

    def base():
        return dict(po=po0.copy(order="F"), qo=qo0.copy(order="F"), sst=sst0.copy(order="F"),
                    wekto=wek0.copy(order="F"), dtopoc=flat.copy(order="F"))

    out = {}
    out["ok"] = base()
    c = base(); c["po"][10, 5, 1] = 1.0e4; out["po_at_limit"] = c
    c = base(); c["po"][10, 5, 1] = -0.9999e4; out["po_below_limit"] = c
    c = base(); c["qo"][3, 30, 2] = -0.05; out["qo_at_limit"] = c
    c = base(); c["qo"][3, 30, 2] = 0.0499; out["qo_below_limit"] = c
    c = base(); c["sst"][0, 0] = 75.0; out["sst_at_limit"] = c
    c = base(); c["sst"][5, 7] = -74.9; out["sst_below_limit"] = c
    c = base(); c["wekto"][47, 35] = 1.0e-3; out["wek_at_limit"] = c
    # thin top layer: hfull = hoc(1) - (po2 - po1)/gpoc(1) = 90 m on n interior points; 100 n / (48*36) per cent
    for tag, n in (("thin_top_19p97", 345), ("thin_top_20p02", 346)):
        c = base()
        c["po"][:] = 0.0
        idx = np.argwhere(np.ones((nx - 2, ny - 2), dtype=bool))[:n] + 1
        c["po"][idx[:, 0], idx[:, 1], 1] = cfg.gpoc[0] * 260.0
        c["po"][idx[:, 0], idx[:, 1], 2] = cfg.gpoc[0] * 260.0   # keeps the interface 2 flat
        out[tag] = c
    # boundary points weigh 1/2, corners 1/4: the whole S row + part of the W column
    c = base(); c["po"][:] = 0.0
    c["po"][:, 0:8, 1:] = cfg.gpoc[0] * 251.0
    out["thin_top_rows_incl_wall"] = c      # 100*(48*7.5)/1728 = 20.83 % -> fails
    c = base(); c["po"][:] = 0.0
    c["po"][:, 0:7, 1:] = cfg.gpoc[0] * 251.0
    out["thin_top_rows_incl_wall_ok"] = c   # 100*(48*6.5)/1728 = 18.06 % -> passes
    # thickness exactly thkmin: triggers the detailed scan (<=) but is not counted (<)
    c = base(); c["po"][:] = 0.0
    c["po"][:, :, 1:] = cfg.gpoc[0] * 250.0
    out["top_exactly_thkmin"] = c
    # intermediate layer: hoc(2) - eta2 + eta1 with eta1 = -(700) -> 750 - 0 - 700 = 50 m everywhere
    c = base(); c["po"][:] = 0.0
    c["po"][:, :, 0] = cfg.gpoc[0] * 700.0
    out["thin_intermediate"] = c
    # bottom layer against topography: 2900 + eta2 - dtopoc
    c = base(); c["po"][:] = 0.0
    c["dtopoc"][:, :] = 2750.0
    c["dtopoc"][:, : ny // 2] = 2850.0
    out["thin_bottom_topography_half"] = c  # ~49 % of the basin at 50 m -> fails
    c = base(); c["po"][:] = 0.0
    c["dtopoc"][5:12, 5:12] = 2850.0
    out["thin_bottom_seamount_ok"] = c
    return om, out


def main():
    import ref_binding
    from qgcm_hip import config
    cfg = config.preset("box_tiny")
    ref_binding.build("box_tiny")
    r = ref_binding.RefLib("box_tiny")
    r.init(cfg.dxo, cfg.dto, cfg.delek, cfg.bccooc, cfg.ah2oc, cfg.ah4oc, cfg.hoc, cfg.gpoc)
    om, cs = cases(cfg)
    r.oml_init(om.hmoc, om.toc[0], om.toc[1], om.st2d, om.st4d, om.ycexp, om.rrcpoc, om.tsbdy, om.tnbdy)
    out = {"names": np.array(sorted(cs))}
    zT = np.zeros_like(cs["ok"]["sst"])
    zP = np.zeros((cfg.nxpo, cfg.nypo), order="F")
    for name in sorted(cs):
        c = cs[name]
        r.set_state(c["po"], c["po"], c["qo"], c["qo"])
        r.oml_set(c["sst"], c["sst"], zT, c["wekto"], zP, zP)
        ok = r.valids(c["dtopoc"])
        for k, v in c.items():
            out["%s_%s" % (name, k)] = v
        out[name + "_solnok"] = np.array(int(ok))
        print(name, ok, file=sys.stderr)
    np.savez_compressed(os.path.join(HERE, "valids_box_tiny.npz"), **out)


if __name__ == "__main__":
    main()
