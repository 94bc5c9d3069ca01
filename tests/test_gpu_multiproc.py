"""One process per y-slab (the production layout: one process per GPU) rehearsed on ONE GPU: three processes share the
card, torch.distributed/gloo carries the exchanges (RCCL refuses duplicate devices), and every process checks its slab
BITWISE against the same decomposition run as virtual ranks.  Found in round 2: allocation zero-fills on the NULL
stream overtaking copies queued on the handle's non-blocking stream (wrong slab constants with three ranks) - invisible
to the virtual-rank tests, which synchronise after every call.  Needs an MI355X; 3 worker processes + this one."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


MP_CASES = [("box_small", "allgather", ""), ("box_small", "p2p", ""), ("box_med", "allgather", ""),
            ("cyc_med", "allgather", ""), ("cyc_small", "p2p", ""),
            ("box_small", "p2p", "oml"), ("cyc_small", "allgather", "oml"),
            ("box_med", "p2p", "early"), ("box_med", "allgather", "early")]


@pytest.mark.parametrize("name,halo,extra", MP_CASES)
def test_three_processes_one_slab_each(name, halo, extra):
    # one rendezvous port per case, fixed by the case's position (Python's hash() of a str is randomised per process:
    # two cases could land on one port)
    port = 29600 + 7 * MP_CASES.index((name, halo, extra))
    nproc = "2" if extra == "early" else "3"  # (early: slabs of at least three 16-row tile rows - 97 rows make two)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", nproc, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "mp_slab_worker.py"), name, halo] + ([extra] if extra else [])
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "MP_SLAB_RESULT OK" in r.stdout, r.stdout[-3000:]
