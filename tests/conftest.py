"""pytest wiring: path set-up, the `gpu` marker and shared fixture loaders."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "licv-vqa_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def _oracle_threads():
    """Threads for the CPU oracle: torch's default is one per HOST core (128 on the GPU boxes) while the job's cgroup grants 16 CPUs -
    measured there on a 800 x 4096 x 11008 linear: 128 threads 166 ms fp32 / 34 ms bf16, 32 threads 47 / 7.5 ms, 16 threads 55 / 11 ms.
    Twice the quota, never more than the affinity mask; None where no quota is set (torch's default stands)."""
    n = None
    try:                                                    # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        try:                                                # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = max(1, quota // period)
        except (OSError, ValueError):
            pass
    if n is None:
        return None
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(avail, 2 * n))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    n = _oracle_threads()
    if n is not None:
        import torch
        torch.set_num_threads(n)


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    return load
