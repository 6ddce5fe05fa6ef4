"""Weight gradient dW = dY^T X for tall operands: mgx_xty (tiled 64 x 128 outputs, operands re-read per tile) against the GEMM
library, at the shapes of the secondary configs (GAT on reddit-small: 232,965 x 128 / 602; arxiv SAGE: 169,343 x 256 / 512)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
from mi355x_graph import sparse  # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


dev = torch.device("cuda")
be = sparse.backend_for(torch.zeros(1, device=dev))
for n, M, K in ((232965, 128, 602), (232965, 41, 128), (169343, 256, 512), (169343, 256, 256), (169343, 40, 512),
                (2449029, 64, 128), (2449029, 64, 200), (2449029, 47, 128), (300000, 64, 128), (7680, 256, 256), (7680, 256, 512), (16128, 256, 256),
                (30000, 256, 256)):
    dy, x = torch.randn(n, M, device=dev), torch.randn(n, K, device=dev)
    t_x = timeit(lambda: be.xty(dy, x))
    t_g = timeit(lambda: dy.t() @ x)
    err = float((be.xty(dy, x) - dy.t() @ x).abs().max())
    print("n = %8d  dY %4d  X %4d: mgx_xty %.3f ms | library GEMM %.3f ms   (max diff %.2e)" % (n, M, K, t_x, t_g, err), flush=True)
