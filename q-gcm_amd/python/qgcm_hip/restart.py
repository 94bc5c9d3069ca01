"""Restart dumps in the reference's wire format (SUBROUTINE resave, src/q-gcm.F:3053-3088; read back at
src/q-gcm.F:612-640): Fortran sequential unformatted records with 4-byte length markers,

    tyrs | po, pom | [pa, pam] | sst, sstm | ast, astm | hmixa, hmixam

`[pa, pam]` is absent in an ocean_only build; the atmospheric mixed-layer arrays are (nxta, nyta) and are written
even then (MODULE intrfac declares them unconditionally).  All fields are fp64, Fortran order.
This is the on-disk contract either side of the device path: a host reads a dump, pushes po/pom (+ sst/sstm) with
set_p / oml_set_state, and writes one from get_state / oml_get_state."""
import struct

import numpy as np


def _rec(f):
    head = f.read(4)
    if len(head) != 4:
        raise EOFError("restart file ends inside a record marker")
    (n,) = struct.unpack("<i", head)
    data = f.read(n)
    (m,) = struct.unpack("<i", f.read(4))
    if len(data) != n or m != n:
        raise ValueError("corrupt Fortran record (markers %d / %d, %d bytes read)" % (n, m, len(data)))
    return np.frombuffer(data, dtype="<f8")


def _put(f, *arrays):
    body = b"".join(np.asfortranarray(a, dtype="<f8").tobytes(order="F") for a in arrays)
    f.write(struct.pack("<i", len(body)))
    f.write(body)
    f.write(struct.pack("<i", len(body)))


def read_restart(path, cfg, coupled=False):
    """-> dict(tyrs, po, pom, sst, sstm, ast, astm, hmixa, hmixam) of an ocean_only dump for grid `cfg`;
    coupled=True: the dump of a coupled build, which also carries pa, pam (nxta+1, nyta+1, 3)."""
    np3, nT, nA = cfg.nxpo * cfg.nypo * cfg.nlo, cfg.nxto * cfg.nyto, cfg.nxta * cfg.nyta
    with open(path, "rb") as f:
        t = _rec(f)
        p = _rec(f)
        pa = _rec(f) if coupled else None
        s = _rec(f)
        a = _rec(f)
        h = _rec(f)
    if t.size != 1 or p.size != 2 * np3 or s.size != 2 * nT or a.size != 2 * nA or h.size != 2 * nA:
        raise ValueError("restart file %s does not match grid %s" % (path, cfg.name))
    sh3, shT, shA = (cfg.nxpo, cfg.nypo, cfg.nlo), (cfg.nxto, cfg.nyto), (cfg.nxta, cfg.nyta)
    r = lambda v, sh: np.asfortranarray(v.reshape(sh, order="F"))
    out = dict(tyrs=float(t[0]), po=r(p[:np3], sh3), pom=r(p[np3:], sh3), sst=r(s[:nT], shT), sstm=r(s[nT:], shT),
               ast=r(a[:nA], shA), astm=r(a[nA:], shA), hmixa=r(h[:nA], shA), hmixam=r(h[nA:], shA))
    if coupled:
        sha = (cfg.nxta + 1, cfg.nyta + 1, 3)
        na3 = sha[0] * sha[1] * sha[2]
        if pa.size != 2 * na3:
            raise ValueError("restart file %s: atmospheric record does not match grid %s" % (path, cfg.name))
        out.update(pa=r(pa[:na3], sha), pam=r(pa[na3:], sha))
    return out


def write_restart(path, cfg, tyrs, po, pom, sst=None, sstm=None, ast=None, astm=None, hmixa=None, hmixam=None,
                  pa=None, pam=None):
    """pa, pam given: the dump of a coupled build (record [pa, pam] after [po, pom])."""
    zT, zA = np.zeros((cfg.nxto, cfg.nyto)), np.zeros((cfg.nxta, cfg.nyta))
    d = lambda x, z: z if x is None else x
    for x, sh in ((po, (cfg.nxpo, cfg.nypo, cfg.nlo)), (pom, (cfg.nxpo, cfg.nypo, cfg.nlo))):
        if np.shape(x) != sh:
            raise ValueError("po / pom must be %s" % (sh,))
    with open(path, "wb") as f:
        _put(f, np.array([tyrs], dtype=np.float64))
        _put(f, po, pom)
        if pa is not None:
            _put(f, pa, pam)
        _put(f, d(sst, zT), d(sstm, zT))
        _put(f, d(ast, zA), d(astm, zA))
        _put(f, d(hmixa, zA), d(hmixam, zA))
