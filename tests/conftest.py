import os
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "q-gcm_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def _usable_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (a GPU box shows all
    256 hardware threads of the host but grants the job 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


# The CPU oracle is OpenMP code with many small parallel regions; with one thread per visible hardware
# thread on a quota-limited box every barrier spins against descheduled threads (a 0.3 s call took > 40 s).
# Must be set before the oracle library (libgomp) is loaded.
os.environ.setdefault("OMP_NUM_THREADS", str(min(_usable_cores(), 8)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


# torch bundles its own HIP runtime: when torch is used at all (the slab transports, torch.distributed) it has to
# be imported before libqgcm_hip.so pulls in /opt/rocm's copy, or torch finds "no HIP GPUs" afterwards.
try:
    import torch  # noqa: F401,E402
except Exception:  # pragma: no cover - torch is optional for the product
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
