"""The stand-alone Fortran host (ISO_C_BINDING over the C ABI, reference call
sequence) against the golden vectors of the true reference.  Needs an MI355X and
amdflang (part of ROCm)."""
import os
import subprocess

import numpy as np
import pytest

from common import FIELDS, load_golden, preset, relerr

pytestmark = pytest.mark.gpu


def test_fortran_host_reproduces_reference(tmp_path, repo_root):
    from qgcm_hip import casefile
    fdir = os.path.join(repo_root, "q-gcm_amd", "fortran")
    exe = os.path.join(fdir, "qgcm_ocean_host")
    if not os.path.exists(exe):
        if not os.path.exists("/opt/rocm/bin/amdflang"):
            pytest.skip("amdflang not available")
        subprocess.check_call(["make", "-C", fdir])
    cfg = preset("box_small")
    g = load_golden("box_small")
    case, out = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    casefile.write_case(case, cfg, g["in_po"], g["in_pom"], g["in_wekpo"], g["in_entoc"], g["in_xon"])
    res = subprocess.run([exe, case, out, "30"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    fields, dpioc, dpiocp = casefile.read_output(out, cfg)
    for f, x in zip(FIELDS, fields):
        assert relerr(x, g["steps30_" + f]) < 1e-10, f
    scale = cfg.xlo * cfg.ylo * np.abs(g["steps30_po"]).max()
    assert np.abs(dpioc - g["steps30_scal"][:cfg.nlo - 1]).max() / scale < 1e-11
