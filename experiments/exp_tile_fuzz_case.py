"""The fuzz case of tests/test_tile_spmm.py that first exceeded 1e-4 against the CPU oracle (trial 5: 17 nodes, 258 k edges), taken apart:
tile kernel, row kernel, fp32 oracle, each against a float64 sum.   python experiments/exp_tile_fuzz_case.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mi355x_graph as mg  # noqa: E402
from mi355x_graph import ops  # noqa: E402
import oracle  # noqa: E402

dev = "cuda:0"
rng = np.random.default_rng(2024)
widths = [1, 2, 3, 4, 8, 12, 16, 20, 24, 32, 36, 41, 48, 64, 100, 128, 130]
for trial in range(6):
    n_src = int(rng.choice([3, 17, 64, 129, 700, 2500, 4000]))
    n_dst = n_src if trial % 3 else int(rng.choice([5, 90, 1100]))
    nnz = int(rng.choice([0, 1, 50, 3000, 40000, 250000])) if trial else 0
    src = rng.integers(0, n_src, nnz)
    dst = rng.integers(0, max(1, n_dst - n_dst // 4), nnz)
    if trial % 4 == 1 and nnz:
        src = np.concatenate([src, rng.integers(0, n_src, 5000), np.full(3000, n_src - 1)])
        dst = np.concatenate([dst, np.full(5000, 1 % n_dst), rng.integers(0, n_dst, 3000)])
    src, dst = src.astype(np.int64), dst.astype(np.int64)
    Ds = [int(D) for D in rng.choice(widths, 4, replace=False)]
    xs = [rng.random((n_src, D), dtype=np.float32) for D in Ds]
    if trial < 5:
        continue
    print("trial", trial, n_src, n_dst, src.shape[0], Ds)
    ip, ix, _ = oracle.coo_to_csr(n_dst, dst, src)
    for D, x_np in zip(Ds, xs):
        ref = np.zeros((n_dst, D))
        np.add.at(ref, dst, x_np.astype(np.float64)[src])
        want = oracle.spmm(ip, ix, None, "copy_lhs", "sum", x_np, None).astype(np.float64)
        outs = {}
        for mode in ("1", "0"):
            os.environ["MGX_TILE"] = mode
            g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=torch.int32, device=dev)
            outs[mode] = ops.gspmm(g, "copy_lhs", "sum", torch.from_numpy(x_np).to(dev), None).cpu().numpy().astype(np.float64)
        rel = lambda a, b: float((np.abs(a - b) / np.maximum(np.abs(b), 1e-9)).max())
        print("  D=%-3d vs float64: tile %.2e  row %.2e  oracle (fp32, sequential) %.2e | tile vs oracle %.2e  row vs oracle %.2e"
              % (D, rel(outs["1"], ref), rel(outs["0"], ref), rel(want, ref), rel(outs["1"], want), rel(outs["0"], want)), flush=True)
