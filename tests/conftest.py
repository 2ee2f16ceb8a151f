import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle's torch ops peak at ~32 intra-op threads for these shapes and collapse beyond (bench.py: 309 rays/s at 32
    # threads, 10.7 at the GPU host's 256): the suite is bound by the oracle, so cap the pool (SNR_TEST_THREADS overrides).
    import torch
    torch.set_num_threads(int(os.environ.get("SNR_TEST_THREADS", min(32, os.cpu_count() or 1))))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
