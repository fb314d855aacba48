import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch sizes its intra-op pool by the machine's cores (128 on an MI355X host) whatever share of them this process may use: the
    # CPU checkers of the suite (the oracle's torch modules) then run oversubscribed.  Sixteen threads = a one-GPU box's share.
    import torch
    if torch.get_num_threads() > 16:
        torch.set_num_threads(16)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle():
    from oracle import mdx_oracle
    mdx_oracle.build()
    return mdx_oracle


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def torus_rel_l2(x, ref):
    """rel-L2 of the coordinate difference taken on the torus (difference wrapped to [-1/2, 1/2))."""
    diff = np.asarray(x, np.float64) - np.asarray(ref, np.float64)
    diff = diff - np.round(diff)
    return float(np.linalg.norm(diff) / max(np.linalg.norm(np.asarray(ref, np.float64)), 1e-30))
