#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the TRUE reference
(jinkakei/q-gcm Fortran + FFTPACK compiled by oracle/build_ref.sh).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Inputs are the deterministic synthetic fields of qgcm_hip.synth (stored in the
fixture, so the tests do not depend on that code) and the parameter presets of
qgcm_hip.config; outputs are whatever the reference routines return:

  consts   eigmod + tridiagonal diagonal           (src/eigmode.f, src/q-gcm.F:932-954)
  homog    homsol products                          (src/conhoms.F:376-641)
  init     q and constraint scalars from p          (src/q-gcm.F:711-731)
  qgostep / ocinvq / ocqbdy                         one call each, from the init state
  stepsN   whole steps incl. the LF averaging        (src/q-gcm.F:1243-1249,1328-1366)
  helm     hsbxoc / hscyoc on a random RHS          (src/ocisubs.F:415-618)
  dsint / drfft  FFTPACK known-answer vectors       (src/fftpack/newbihar)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "q-gcm_amd", "python"))

import ref_binding  # noqa: E402
from qgcm_hip import config, synth  # noqa: E402

SNAPS = {"box_tiny": (1, 2, 25, 26, 60), "box_tiny2": (1, 26), "box_tiny5": (1, 2, 26), "cyc_tiny6": (1, 2, 26), "box_small": (1, 30),
         "cyc_tiny": (1, 2, 25, 26, 60), "cyc_small": (1, 30),
         "box_tiny_ah2": (1, 2, 26), "cyc_tiny_ah2": (1, 2, 26),
         "box_tiny_spl": (1, 2, 26), "cyc_tiny_spl": (1, 2, 26)}
# fixtures that differ from another one only in run-time parameters share its reference build
REFCFG = {"box_tiny_ah2": "box_tiny", "cyc_tiny_ah2": "cyc_tiny"}


def state_dict(r, tag, out):
    po, pom, qo, qom = r.get_state()
    out[tag + "_po"], out[tag + "_pom"], out[tag + "_qo"], out[tag + "_qom"] = po, pom, qo, qom
    out[tag + "_scal"] = r.get_scalars()


def make(name):
    cfg = config.preset(name)
    ref_binding.build(REFCFG.get(name, name))
    r = ref_binding.RefLib(REFCFG.get(name, name))
    assert (r.nx, r.ny, r.nl, bool(r.cyclic)) == (cfg.nxpo, cfg.nypo, cfg.nlo, cfg.cyclic)
    assert r.fnot == cfg.fnot and r.beta == cfg.beta
    r.init(cfg.dxo, cfg.dto, cfg.delek, cfg.bccooc, cfg.ah2oc, cfg.ah4oc, cfg.hoc, cfg.gpoc)
    out = {}
    for k, v in r.get_consts().items():
        out["c_" + k] = np.asarray(v)
    for k, v in r.get_homog().items():
        out["h_" + k] = np.asarray(v)
    rng = np.random.default_rng(247)
    nx, ny, nl = cfg.nxpo, cfg.nypo, cfg.nlo
    # ---- inputs -------------------------------------------------------
    po = synth.gaussian_eddy(cfg, noise=1.0e-3)
    pom = np.asfortranarray(0.98 * po)
    tx, ty = synth.wind_stress(cfg)
    _, wekpo = synth.wekpo_from_tau(cfg, tx, ty)
    i = np.arange(nx)[:, None] / (nx - 1.0)
    j = np.arange(ny)[None, :] / (ny - 1.0)
    entoc = np.asfortranarray(1.0e-7 * np.cos(2 * np.pi * i) * np.sin(np.pi * j))
    xon = np.zeros(nl - 1)
    xon[0] = 1.0e3
    out.update(in_po=po, in_pom=pom, in_wekpo=wekpo, in_entoc=entoc, in_xon=xon)
    spl_on, rspl, c1, lspl = r.get_sponge()
    assert spl_on == (cfg.l_spl > 0.0)
    if spl_on:  # -Dsponge_layer_k247 build: its ramp (set as src/q-gcm.F:1154-1168 does) and constants are inputs
        assert (c1, lspl) == (cfg.c1_spl, cfg.l_spl)
        out.update(in_rspl=rspl, in_c1spl=np.array(c1), in_lspl=np.array(lspl))
    r.set_p(po, pom)
    r.set_forcing(wekpo, entoc, xon)
    if cfg.cyclic:
        txis, txin = synth.tau_line_integrals(cfg, tx)
        enis = np.full(nl - 1, 1.0e-3)
        enin = np.full(nl - 1, 2.0e-3)
        r.set_cyc_forcing(txis, txin, enis, enin)
        out.update(in_txis=txis, in_txin=txin, in_enis=enis, in_enin=enin)
    state_dict(r, "init", out)
    # ---- one call each ---------------------------------------------------
    full = "tiny" in name  # per-call snapshots only for the tiny grids (fixture size)
    r.qgostep()
    if full:
        state_dict(r, "qgostep", out)
    r.ocinvq()
    if full:
        state_dict(r, "ocinvq", out)
    r.ocqbdy()
    state_dict(r, "ocqbdy", out)
    r.lf_average()  # ocean step 1 is followed by the averaging (mod(nt-1,25*nstr)==0)
    done = 1
    for s in SNAPS[name]:
        if s > done:
            r.steps(done + 1, s - done)
            done = s
        state_dict(r, "steps%d" % s, out)
    # ---- Helmholtz solver ------------------------------------------------
    if name in REFCFG or spl_on:  # (the Helmholtz solver sees neither ah2oc nor the sponge: its vectors stay with the base fixture)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name, "%d arrays" % len(out))
        return r
    rhs = np.asfortranarray(rng.standard_normal((nx, ny)))
    boc = out["c_bd2oc"] - out["c_rdm2oc"][1]
    out["helm_rhs"], out["helm_boc"] = rhs, boc
    out["helm_sol"] = r.helmholtz(rhs, boc)
    boc0 = out["c_bd2oc"] - out["c_rdm2oc"][0]
    if cfg.cyclic:
        boc0 = boc0.copy()  # barotropic cyclic mode: boc(1) = -2 aoc is fine (non-singular)
    out["helm_boc0"] = boc0
    out["helm_sol0"] = r.helmholtz(rhs, boc0)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "%d arrays" % len(out))
    return r


def make_fft(r):
    rng = np.random.default_rng(11)
    out = {}
    for n in (3, 4, 5, 14, 47, 59, 95, 959):
        x = rng.standard_normal(n)
        out["dsint_in_%d" % n] = x
        out["dsint_out_%d" % n] = r.dsint(x)
    for n in (4, 6, 9, 10, 48, 96, 384, 385):
        x = rng.standard_normal(n)
        f = r.drfft(x, +1)
        out["drfft_in_%d" % n] = x
        out["drfftf_out_%d" % n] = f
        out["drfftb_out_%d" % n] = r.drfft(f, -1)
    for gp, h in (((0.015, 0.0075), (350.0, 750.0, 2900.0)), ((0.02,), (500.0, 3500.0)),
                  ((0.02, 0.01, 0.005), (300.0, 500.0, 1200.0, 2000.0))):
        e = r.eigmod(gp, h)
        tag = "eig%d" % len(h)
        out[tag + "_gp"], out[tag + "_h"] = np.array(gp), np.array(h)
        for k, v in e.items():
            out[tag + "_" + k] = v
    out["eig_fnot"] = np.array(r.fnot)
    np.savez_compressed(os.path.join(HERE, "fftpack_eigmod.npz"), **out)
    print("wrote fftpack_eigmod")


def make_fft_long(r):
    """FFTPACK known-answer vectors at the row lengths of NAtl 1 km (dsint, n = 4799) and SOcn 5 km (drfftf / drfftb,
    n = 4608) and at 960 / 4800 for the cyclic kernels; any reference build serves (FFTPACK does not depend on the grid)."""
    rng = np.random.default_rng(12)
    out = {}
    for n in (4799,):
        x = rng.standard_normal(n)
        out["dsint_in_%d" % n] = x
        out["dsint_out_%d" % n] = r.dsint(x)
    for n in (960, 4608, 4800):
        x = rng.standard_normal(n)
        f = r.drfft(x, +1)
        out["drfft_in_%d" % n] = x
        out["drfftf_out_%d" % n] = f
        out["drfftb_out_%d" % n] = r.drfft(f, -1)
    np.savez_compressed(os.path.join(HERE, "fftpack_long.npz"), **out)
    print("wrote fftpack_long")


if __name__ == "__main__":
    # one process per config: the reference libraries export identical symbols
    if len(sys.argv) == 2 and sys.argv[1] == "fft_long":
        ref_binding.build("box_tiny")
        make_fft_long(ref_binding.RefLib("box_tiny"))
    elif len(sys.argv) == 2:
        r = make(sys.argv[1])
        if sys.argv[1] == "box_tiny":
            make_fft(r)
    else:
        import subprocess
        for n in (sys.argv[1:] or list(SNAPS)):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), n])
