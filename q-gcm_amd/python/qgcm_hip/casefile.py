"""Case files for the stand-alone Fortran host (q-gcm_amd/fortran/qgcm_ocean_host.F90).

Layout (little-endian stream, Fortran order), matching the READ statements there:
  int32   nxpo, nypo, nlo, nstr
  float64 fnot, beta, dxo, dto, delek, bccooc, aoc
  float64 ah2oc(nlo), ah4oc(nlo), hoc(nlo), gpoc(nlo-1), amatoc(nlo,nlo), rdm2oc(nlo),
          ctl2moc(nlo,nlo), ctm2loc(nlo,nlo)
  float64 yporel(nypo), bd2oc(nxto), ddynoc(nxpo,nypo)
  float64 po, pom, qo, qom (nxpo,nypo,nlo each), wekpo, entoc (nxpo,nypo), xon(nlo-1),
          dpioc(nlo-1), dpiocp(nlo-1)
Output file: po, pom, qo, qom, dpioc, dpiocp.
"""
import numpy as np

from . import hostinit


def write_case(path, cfg, po, pom, wekpo, entoc=None, xon=None, ddynoc=None):
    if cfg.cyclic:
        raise ValueError("the Fortran host covers the box ocean")
    nl = cfg.nlo
    A, rdm2, cl2m, cm2l = hostinit.eigmod(cfg.gpoc, cfg.hoc, cfg.fnot)
    aoc, bd2 = hostinit.bd2oc(cfg)
    yp = cfg.yporel()
    dd = np.zeros((cfg.nxpo, cfg.nypo)) if ddynoc is None else np.asarray(ddynoc)
    po = np.asfortranarray(po, dtype=np.float64)
    pom = np.asfortranarray(pom, dtype=np.float64)
    qo = hostinit.q_from_p(cfg, A, yp, dd, po)
    qom = hostinit.q_from_p(cfg, A, yp, dd, pom)
    scal = hostinit.constr(cfg, A, po, pom)
    ent = np.zeros((cfg.nxpo, cfg.nypo)) if entoc is None else np.asarray(entoc)
    x = np.zeros(nl - 1) if xon is None else np.asarray(xon, dtype=np.float64)
    with open(path, "wb") as f:
        np.array([cfg.nxpo, cfg.nypo, nl, cfg.nstr], dtype="<i4").tofile(f)
        np.array([cfg.fnot, cfg.beta, cfg.dxo, cfg.dto, cfg.delek, cfg.bccooc, aoc], dtype="<f8").tofile(f)
        for a in (cfg.ah2oc, cfg.ah4oc, cfg.hoc, cfg.gpoc):
            np.asarray(a, dtype="<f8").tofile(f)
        for a in (A, rdm2, cl2m, cm2l, yp, bd2, dd, po, pom, qo, qom, wekpo, ent, x, scal[:nl - 1], scal[nl - 1:2 * (nl - 1)]):
            np.asarray(a, dtype="<f8").ravel(order="F").tofile(f)


def read_output(path, cfg):
    nl, N = cfg.nlo, cfg.nxpo * cfg.nypo * cfg.nlo
    raw = np.fromfile(path, dtype="<f8")
    assert raw.size == 4 * N + 2 * (nl - 1), raw.size
    f = [raw[i * N:(i + 1) * N].reshape((cfg.nxpo, cfg.nypo, nl), order="F") for i in range(4)]
    return f, raw[4 * N:4 * N + nl - 1], raw[4 * N + nl - 1:]
