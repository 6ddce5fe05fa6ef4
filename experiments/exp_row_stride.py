"""g-SpMM copy_u/sum at D = 64 on the products-shaped graph with the gathered rows (and the output rows) at different row
strides: does the distance between rows in memory matter once every row is two whole 128-byte lines?
  python experiments/exp_row_stride.py [products] [64]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
from mi355x_graph import sparse  # noqa: E402
from mi355x_graph.datasets import SHAPES, synthetic_edges  # noqa: E402


def timeit(fn, reps=6):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


name = sys.argv[1] if len(sys.argv) > 1 else "products"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda")
spec = SHAPES[name]
n, m = spec["n"], spec["m"]
src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], dev, symmetric=spec["symmetric"])
csr = sparse.coo_to_csr(n, n, dst.int(), src.int())
be = sparse.backend_for(csr.indptr)
x = torch.rand(n, D, device=dev)
t0 = timeit(lambda: sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None))
print("dense rows (stride %d): %.3f ms" % (D, t0), flush=True)
strides = ((2 * D, D), (D, 2 * D), (2 * D, 2 * D), (3 * D, 3 * D), (4 * D, 4 * D), (D + 32, D + 32))
if D % 32:  # rows that are not whole lines (D = 100): strides that are
    strides = ((128, D), (160, D), (192, D), (200, D), (256, D), (384, D))
for su, so in strides:
    if n * max(su, so) * 4 >= (1 << 32):
        continue
    U = torch.zeros(n, su, device=dev)
    U[:, :D] = x
    O = torch.zeros(n, so, device=dev)
    t = timeit(lambda: be.spmm_copy_u_strided(csr, "sum", U[:, :D], O[:, :D]))
    print("gathered rows %4d floats apart, output rows %4d apart: %.3f ms" % (su, so, t), flush=True)
    del U, O
# the backward form: accumulate, gathered rows = right half, output rows = left half of ONE [n, 2D] matrix
acc = torch.zeros(n, D, device=dev)
t = timeit(lambda: sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None, accumulate_into=acc))
print("accumulate, dense rows: %.3f ms" % t, flush=True)
if n * 2 * D * 4 < (1 << 32):
    cat = torch.zeros(n, 2 * D, device=dev)
    cat[:, D:] = x
    t = timeit(lambda: be.spmm_copy_u_strided(csr, "sum", cat[:, D:], cat[:, :D], accumulate=True))
    print("accumulate, right half -> left half of one [n, %d] matrix: %.3f ms" % (2 * D, t), flush=True)
    U = torch.zeros(n, 2 * D, device=dev)
    U[:, :D] = x
    t = timeit(lambda: be.spmm_copy_u_strided(csr, "sum", U[:, :D], acc, accumulate=True))
    print("accumulate, gathered rows %d apart (own matrix), dense output: %.3f ms" % (2 * D, t), flush=True)
    t = timeit(lambda: be.spmm_copy_u_strided(csr, "sum", U[:, :D], cat[:, :D], accumulate=True))
    print("accumulate, gathered rows %d apart (own matrix), output rows %d apart: %.3f ms" % (2 * D, 2 * D, t), flush=True)
