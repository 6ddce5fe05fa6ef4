"""Wide odd widths (reddit's 602 input features): the row-per-wave g-SpMM at D = 602 (8-byte lanes, rows not line-aligned)
against the same aggregation on rows padded to 608 floats (19 whole 128-byte lines, 16-byte lanes).
  python experiments/exp_wide_ragged.py [reddit] [602,608,640,300,304,320]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
from mi355x_graph import sparse  # noqa: E402
from mi355x_graph.datasets import SHAPES, synthetic_edges  # noqa: E402


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


name = sys.argv[1] if len(sys.argv) > 1 else "reddit"
widths = [int(w) for w in (sys.argv[2] if len(sys.argv) > 2 else "602,608,640,300,304,320").split(",")]
dev = torch.device("cuda")
spec = SHAPES[name]
n, m = spec["n"], spec["m"]
src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], dev, symmetric=spec["symmetric"])
csr = sparse.coo_to_csr(n, n, dst.int(), src.int())
E = csr.nnz
for D in widths:
    x = torch.rand(n, D, device=dev)
    t = timeit(lambda: sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None))
    tp = timeit(lambda: torch.nn.functional.pad(x, (0, (-D) % 32)))
    print("D = %4d: %.3f ms (gather %.1f TB/s); pad-to-32 copy %.3f ms" % (D, t, E * D * 4 / t / 1e9, tp), flush=True)
