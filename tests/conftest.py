import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def _host_cores() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup quota (a GPU box shows 256 CPUs and
    grants 16: torch's default of one thread per visible CPU makes the CPU oracle's network ~8x slower there)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(n, 32))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch

        torch.set_num_threads(_host_cores())
    except Exception:  # noqa: BLE001
        pass


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(GOLDEN / name, allow_pickle=False)
    return load


def seeded_images(seed: int, n: int) -> np.ndarray:
    """Same generator as tools/make_golden.py: u8-quantised [n,256,256,4] f32 in [0,1]."""
    rs = np.random.RandomState(seed)
    base = rs.randint(0, 256, size=(n, 32, 32, 4)).astype(np.float32)
    img = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)
    img += rs.randint(-20, 21, size=img.shape)
    return (np.clip(img, 0, 255) / 255).astype(np.float32)
