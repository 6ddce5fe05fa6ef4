import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dgl-0.5-benchmark_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def random_graph(n_src, n_dst, nnz, seed, skew=True):
    """Seeded multigraph with a heavy-tailed in-degree (plus isolated nodes, self loops, duplicates)."""
    rng = np.random.default_rng(seed)
    if skew and n_dst > 4:
        w = 1.0 / np.arange(1, n_dst + 1) ** 0.9
        w[rng.integers(0, n_dst, size=max(1, n_dst // 8))] = 0.0  # isolated destinations
        w /= w.sum()
        dst = rng.choice(n_dst, size=nnz, p=w)
    else:
        dst = rng.integers(0, max(n_dst, 1), size=nnz)
    src = rng.integers(0, max(n_src, 1), size=nnz)
    return src.astype(np.int64), dst.astype(np.int64)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
