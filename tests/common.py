"""Shared helpers for the parity tests (tests may use oracle/, the product may not)."""
import os

import numpy as np

import oracle_binding as ob
from qgcm_hip import config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIELDS = ("po", "pom", "qo", "qom")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / den) if den > 0 else float(np.abs(a).max())


def make_oracle(cfg, yporel=None):
    return ob.Oracle(cfg.nxpo, cfg.nypo, cfg.nlo, cfg.cyclic, cfg.fnot, cfg.beta, cfg.dxo, cfg.dto,
                     cfg.delek, cfg.bccooc, cfg.ah2oc, cfg.ah4oc, cfg.hoc, cfg.gpoc,
                     cfg.yporel() if yporel is None else yporel)


def apply_inputs(model, g, cfg):
    """Load the fixture's inputs into an Oracle or an OceanModel (same method names)."""
    model.set_p(g["in_po"], g["in_pom"])
    model.set_forcing(g["in_wekpo"], g["in_entoc"], g["in_xon"])
    if cfg.cyclic:
        model.set_cyc_forcing(float(g["in_txis"]), float(g["in_txin"]), g["in_enis"], g["in_enin"])


def load_snapshot(model, g, tag):
    model.set_state(*[g["%s_%s" % (tag, f)] for f in FIELDS])
    model.set_scalars(g[tag + "_scal"])


def state_errs(model, g, tag):
    st = model.get_state()
    return {f: relerr(st[i], g["%s_%s" % (tag, f)]) for i, f in enumerate(FIELDS)}


def scal_err(model, g, tag, cfg):
    """Constraint scalars are cancelling area integrals: compare relative to
    xlo*ylo*max|po|, not to their own magnitude."""
    s, r = model.get_scalars(), g[tag + "_scal"]
    scale = cfg.xlo * cfg.ylo * np.abs(g[tag + "_po"]).max()
    nl = cfg.nlo
    e = np.abs(s[:2 * (nl - 1)] - r[:2 * (nl - 1)]).max() / scale
    if cfg.cyclic:
        den = np.abs(r[2 * (nl - 1):]).max()
        e = max(e, np.abs(s[2 * (nl - 1):] - r[2 * (nl - 1):]).max() / den)
    return float(e)


CONFIG_NAMES = ("box_tiny", "box_tiny2", "box_small", "cyc_tiny", "cyc_small")
BOX_NAMES = ("box_tiny", "box_tiny2", "box_small")
SNAPS = {"box_tiny": (1, 2, 25, 26, 60), "box_tiny2": (1, 26), "box_small": (1, 30),
         "cyc_tiny": (1, 2, 25, 26, 60), "cyc_small": (1, 30)}


def preset(name):
    return config.preset(name)
