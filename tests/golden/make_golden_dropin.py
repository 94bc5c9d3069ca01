#!/usr/bin/env python3
"""Golden restart dumps for the drop-in test: the UNMODIFIED reference main program (built from
/root/reference/src by q-gcm_amd/fortran/dropin/build_dropin.sh -> _dropin/<cfg>/q-gcm_ref) is run on the case
written by dropin/make_case.py, and its final restart dump `out/last.day` (SUBROUTINE resave, src/q-gcm.F:3053-3088)
is stored as tests/golden/dropin_<cfg>_lastday.bin.  Build container only.

    python tests/golden/make_golden_dropin.py [cfg ...] [--force]
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
DROP = os.path.join(ROOT, "q-gcm_amd", "fortran", "dropin")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dropin_cases import CASES, RUNS, RUNS_RESTART, THREADS, golden_name, prepare_case  # noqa: E402

only = sys.argv[1:]   # optional: configurations to (re)generate; existing vectors of the others are kept
for cfg, (dims, mode) in CASES.items():
    if only and cfg not in only:
        continue
    subprocess.check_call([os.path.join(DROP, "build_dropin.sh"), cfg] + [str(x) for x in dims] + [mode],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for nsteps, rst in [(n, False) for n, _ in RUNS[cfg]] + [(n, True) for n, _ in RUNS_RESTART.get(cfg, ())]:
      if only and os.path.exists(os.path.join(HERE, golden_name(cfg, nsteps, rst))) and "--force" not in only:
          continue
      with tempfile.TemporaryDirectory() as d:
        prepare_case(cfg, d, nsteps, rst)
        exe = os.path.join(ROOT, "q-gcm_amd", "fortran", "_dropin", cfg, "q-gcm_ref")
        env = dict(os.environ, OMP_NUM_THREADS=str(THREADS[cfg]), OMP_STACKSIZE="512M")
        log = subprocess.run("ulimit -s unlimited; exec %s" % exe, shell=True, cwd=d, env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True)
        assert log.returncode == 0 and "End of run" in log.stdout, log.stdout[-2000:]
        shutil.copy(os.path.join(d, "out", "last.day"), os.path.join(HERE, golden_name(cfg, nsteps, rst)))
        print("wrote %s (%d ocean steps%s)" % (golden_name(cfg, nsteps, rst), nsteps, ", from the eddy restart" if rst else ""))
