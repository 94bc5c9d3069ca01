"""The ACTUAL drop-in (SURVEY 8 row b): the reference main program src/q-gcm.F - patched copy with the host <-> device
synchronisation edits of INTEGRATION.md section 3 - and the rest of the reference model compiled from
/root/reference/src, with src/qgosubs.F, src/ocisubs.F, src/omlsubs.F (coupled: + src/qgasubs.F, src/atisubs.F)
replaced by q-gcm_amd/fortran/qgcm_hip_shim.F90, linked against libqgcm_hip.so.

CPU part (build container): the three configurations compile and LINK (box, -Dcyclic_ocean, coupled).
GPU part: the executables run the case of dropin/make_case.py; their final restart dumps are compared with the
ones the UNMODIFIED reference executable wrote here (tests/golden/dropin_*_lastday.bin)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from common import GOLDEN, relerr
from dropin_cases import CASES, DROP, RUNS, RUNS_RESTART, THREADS, exe_path, golden_name, prepare_case
from qgcm_hip import config, restart

HAVE_REF = os.path.isdir("/root/reference/src")


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference sources (build container only)")
@pytest.mark.parametrize("cfg", list(CASES))
def test_dropin_compiles_and_links(cfg):
    dims, mode = CASES[cfg]
    r = subprocess.run([os.path.join(DROP, "build_dropin.sh"), cfg] + [str(x) for x in dims] + [mode],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    exe = exe_path(cfg, "hip")
    assert os.path.exists(exe)
    ldd = subprocess.run(["ldd", exe], stdout=subprocess.PIPE, text=True).stdout
    assert "libqgcm_hip.so" in ldd and "not found" not in ldd, ldd
    # nothing of the reference's source text stays behind in the build directory
    for sub in ("hip", "ref"):
        left = [f for f in os.listdir(os.path.join(os.path.dirname(exe), sub)) if f.endswith((".F", ".f", ".F90"))]
        assert left == [], (sub, left)
    # the replaced routines really come from the shim: the executable binds the C ABI, not FFTPACK-based solvers
    syms = subprocess.run(["nm", "-D", "--undefined-only", exe], stdout=subprocess.PIPE, text=True).stdout
    for s in ("qgcm_hip_qgostep", "qgcm_hip_ocinvq", "qgcm_hip_ocqbdy", "qgcm_hip_helmholtz", "qgcm_hip_lf_average"):
        assert s in syms, s
    if mode == "coupled":
        for s in ("qgcm_hip_qgastep", "qgcm_hip_atinvq", "qgcm_hip_atqzbd"):
            assert s in syms, s


def test_patch_script_finds_every_edit_point():
    if not HAVE_REF:
        pytest.skip("needs the reference sources (build container only)")
    sys.path.insert(0, DROP)
    import patch_main
    src = open("/root/reference/src/q-gcm.F").read().split("\n")
    out = patch_main.patch(src)   # raises if an edit point of INTEGRATION.md section 3 is not found
    added = [l for l in out if "qgcm_hip" in l]
    assert any("call qgcm_hip_push" in l for l in added) and any("qgcm_hip_lf_average(qgcm_hip_handle)" in l for l in added)
    assert any("qgcm_hip_lf_average(qgcm_hip_atm_handle)" in l for l in added)
    # the three per-step calls of the hot path are untouched (src/q-gcm.F:1243-1249, 1262-1268)
    for call in ("call qgostep", "call ocinvq", "call ocqbdy (qo, po)", "call qgastep", "call atinvq", "call atqzbd (qa, pa)"):
        assert sum(1 for l in out if l.strip() == call) >= 1, call


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,nsteps,tol,rst", [(c, n, t, False) for c in CASES for n, t in RUNS[c]] +
                         [(c, n, t, True) for c in RUNS_RESTART for n, t in RUNS_RESTART[c]])
def test_dropin_executable_matches_the_reference_restart(cfg, nsteps, tol, rst, tmp_path):
    exe = exe_path(cfg, "hip")
    if not os.path.exists(exe):
        # Row (b) of SURVEY 8 rests on this test: a GPU box whose snapshot lacks the executables must not show green.
        # They are built by __graft_entry__.build() in the build container (make check_dropin needs /root/reference)
        # and travel with the snapshot; only an explicit opt-out turns their absence into a skip.
        if os.environ.get("QGCM_SKIP_DROPIN") == "1":
            pytest.skip("drop-in executable not built and QGCM_SKIP_DROPIN=1")
        pytest.fail("drop-in executable %s is missing: run __graft_entry__.build() where /root/reference is mounted "
                    "(q-gcm_amd/fortran/dropin/build_dropin.sh), or set QGCM_SKIP_DROPIN=1 to skip knowingly" % exe)
    mode = CASES[cfg][1]
    d = str(tmp_path)
    prepare_case(cfg, d, nsteps, rst)
    env = dict(os.environ, OMP_NUM_THREADS=str(THREADS[cfg]), OMP_STACKSIZE="512M")
    r = subprocess.run("ulimit -s unlimited 2>/dev/null; exec %s" % exe, shell=True, cwd=d, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "End of run" in r.stdout, r.stdout[-3000:]
    oc = config.preset(cfg)
    got = restart.read_restart(os.path.join(d, "out", "last.day"), oc, coupled=(mode == "coupled"))
    ref = restart.read_restart(os.path.join(GOLDEN, golden_name(cfg, nsteps, rst)), oc, coupled=(mode == "coupled"))
    assert got["tyrs"] == ref["tyrs"]
    # ocean steps from rest under wind (zero IC) / from radiative balance (coupled): free-running comparison
    # (tolerances: dropin_cases.RUNS)
    keys = ("po", "pom", "sst", "sstm") + (("pa", "pam", "ast", "astm", "hmixa", "hmixam") if mode == "coupled" else ())
    errs = {k: relerr(got[k], ref[k]) for k in keys}
    try:  # the measured differences, for the record (tolerances in dropin_cases.RUNS are calibrated against them)
        with open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "dropin_errors.log"), "a") as f:
            f.write("%s %d steps%s: %s\n" % (cfg, nsteps, " (eddy restart)" if rst else "", " ".join("%s=%.2e" % kv for kv in errs.items())))
    except OSError:
        pass
    for k in keys:
        if nsteps > 1:   # (after one step from radiative balance the ocean is still at rest: po = pom = 0 in both)
            assert np.abs(ref[k]).max() > 0, k
        assert errs[k] < tol, (k, errs)
