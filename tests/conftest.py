import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ptrt-game-engine_amd"), os.path.join(ROOT, "oracle"), ROOT,
          os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def P():
    # torch first: its wheel carries its own libamdhip64.so.7 / libhsa-runtime64.so.1.  Loaded first, that copy
    # also satisfies libptrt_amd.so's NEEDED entry (same SONAME), so the process has ONE HIP runtime; the other
    # way round torch loads a second runtime beside /opt/rocm's, and on some boxes the second one then reports
    # "No HIP GPUs are available" (seen in test_refit / test_build_gpu).  bench.py imports torch first as well.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    import ptrt_amd
    return ptrt_amd


@pytest.fixture(scope="session")
def O():
    import oracle
    return oracle


@pytest.fixture(scope="session")
def blue_noise(P):
    return P.blue_noise_table()
