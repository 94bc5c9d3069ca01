#!/usr/bin/env python3
"""Golden vectors of the ocean mixed layer (oml / omladf, src/omlsubs.F) from the TRUE
reference compiled by oracle/build_ref.sh.  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_oml.py

One reference build per boundary variant (they are compile-time options there):
  oml_box_tiny      box ocean, no-flux walls                      (-Docean_only)
  oml_box_tiny_sb   box ocean, specified southern temperature     (-Docean_only -Dsb_hflux)
  oml_cyc_tiny      zonally cyclic ocean, specified northern T    (-Docean_only -Dcyclic_ocean -Dnb_hflux)
Each fixture holds the inputs, the result of ONE `call oml` from them, and coupled runs
(oml, qgostep, ocinvq, ocqbdy + averaging, src/q-gcm.F:1232-1249,1328-1366) after 1, 2, 26 and 40 steps.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "q-gcm_amd", "python"))

VARIANTS = {"oml_box_tiny": ("box_tiny", "box_tiny"), "oml_box_tiny_sb": ("box_tiny_sb", "box_tiny"),
            "oml_cyc_tiny": ("cyc_tiny", "cyc_tiny")}
SNAPS = (1, 2, 26, 40)


def make(name):
    import ref_binding
    from qgcm_hip import config, synth
    refcfg, preset = VARIANTS[name]
    cfg = config.preset(preset)
    ref_binding.build(refcfg)
    r = ref_binding.RefLib(refcfg)
    sb, nb = r.oml_flags()
    om = config.oml_preset(cfg, sb_hflux=bool(sb), nb_hflux=bool(nb))
    r.init(cfg.dxo, cfg.dto, cfg.delek, cfg.bccooc, cfg.ah2oc, cfg.ah4oc, cfg.hoc, cfg.gpoc)
    r.oml_init(om.hmoc, om.toc[0], om.toc[1], om.st2d, om.st4d, om.ycexp, om.rrcpoc, om.tsbdy, om.tnbdy)
    nl = cfg.nlo
    po = synth.gaussian_eddy(cfg, noise=1.0e-3)
    pom = np.asfortranarray(0.98 * po)
    sst, sstm, fnet, tx, ty = synth.mixed_layer_fields(cfg, om, seed=11)
    wekto, wekpo = synth.wekpo_from_tau(cfg, tx, ty)
    out = dict(in_po=po, in_pom=pom, in_sst=sst, in_sstm=sstm, in_fnetoc=fnet, in_tauxo=tx, in_tauyo=ty,
               in_wekto=wekto, in_wekpo=wekpo,
               oml_params=np.array([om.hmoc, om.toc[0], om.toc[1], om.st2d, om.st4d, om.ycexp, om.rrcpoc,
                                    float(sb), om.tsbdy, float(nb), om.tnbdy]))

    def load():
        r.set_p(po, pom)
        r.set_forcing(wekpo, None, np.zeros(nl - 1))
        if cfg.cyclic:
            txis, txin = synth.tau_line_integrals(cfg, tx)
            r.set_cyc_forcing(txis, txin, np.zeros(nl - 1), np.zeros(nl - 1))
            out.update(in_txis=txis, in_txin=txin)
        r.oml_set(sst, sstm, fnet, wekto, tx, ty)

    load()
    r.oml()
    a, b, e, s = r.oml_get()
    out.update(call_sst=a, call_sstm=b, call_entoc=e, call_scal=s)
    load()
    done = 0
    for n in SNAPS:
        r.steps_oml(done + 1, n - done)
        done = n
        st = r.get_state()
        a, b, e, s = r.oml_get()
        for f, x in zip(("po", "pom", "qo", "qom"), st):
            out["steps%d_%s" % (n, f)] = x
        out["steps%d_sst" % n], out["steps%d_sstm" % n], out["steps%d_entoc" % n] = a, b, e
        out["steps%d_omlscal" % n] = s
        out["steps%d_scal" % n] = r.get_scalars()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "written; convecting fraction after one call:", out["call_scal"][1])


if __name__ == "__main__":
    if len(sys.argv) > 1:
        make(sys.argv[1])
    else:
        for v in VARIANTS:  # one reference configuration per process (oracle/ref_binding.py)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), v])
