"""Round 4: mgx_rows_gemm against the library GEMM (+ the separate 1 / deg pass) on the products layer shapes, N = 2,449,029."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dgl-0.5-benchmark_amd"))
import tunable  # noqa: E402
tunable.setup()
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from mi355x_graph import sparse  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2449029


def timed(fn, reps=12):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]


be = sparse.backend_for(torch.zeros(1, device=dev))
gen = torch.Generator(device=dev).manual_seed(0)
inv = torch.rand(n, device=dev, generator=gen)
print("n = %d" % n)
for name, K, M in (("forward layer 1  [x | agg] 200 -> 64", 200, 64), ("forward layer 2  128 -> 64", 128, 64), ("forward layer 3  128 -> 47", 128, 47)):
    a = torch.rand(n, K, device=dev, generator=gen)
    w = torch.rand(M, K, device=dev, generator=gen)
    b = torch.rand(M, device=dev, generator=gen)
    t_lib = timed(lambda: F.linear(a, w, b))
    t_own = timed(lambda: be.rows_gemm(a, w, b_transposed=True, bias=b))
    gb = (n * (K + M) * 4) / 1e9
    print("%-40s library %.4f ms (%.2f TB/s)   mgx_rows_gemm %.4f ms (%.2f TB/s)" % (name, t_lib, gb / t_lib, t_own, gb / t_own))
    del a
for name, K, M in (("backward layer 2  dY 64 -> d[h | neigh] 128, right half / deg", 64, 128), ("backward layer 3  dY 47 -> 128", 47, 128)):
    dy = torch.rand(n, K, device=dev, generator=gen)
    w = torch.rand(K, M, device=dev, generator=gen)

    def lib():
        d = dy @ w
        d[:, M // 2:].mul_(inv.view(-1, 1))
        return d
    t_lib = timed(lib)
    t_gemm = timed(lambda: dy @ w)
    t_own = timed(lambda: be.rows_gemm(dy, w, row_scale=inv, scale_from=M // 2))
    gb = (n * (K + M) * 4) / 1e9
    print("%-62s library %.4f ms (GEMM alone %.4f)   mgx_rows_gemm %.4f ms (%.2f TB/s)" % (name, t_lib, t_gemm, t_own, gb / t_own))
    del dy
