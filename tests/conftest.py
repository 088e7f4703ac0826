import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")
    # Multi-process GPU tests start their ranks from a fork server that is launched HERE, before anything in this process has
    # touched the GPU: the ranks are then forks of a GPU-free process (no exec from, and no fork of, a process holding the GPU).
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        import multiprocessing as mp
        from multiprocessing import forkserver
        try:
            mp.set_start_method("forkserver", force=True)
            forkserver.set_forkserver_preload(["numpy"])
            forkserver.ensure_running()
        except Exception as err:  # pragma: no cover
            print(f"conftest: fork server not started ({err}); multi-process GPU tests will skip")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="session")
def golden():
    return gold
