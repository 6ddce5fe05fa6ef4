"""Round 4: row-per-lane-group (spmm_rowgroup32_kernel) against row-per-wave (spmm_rowwave32_kernel) by average row length and
width.  One process per setting (the switch is read once): MGX_ROWGROUP=0 | 1 python exp_rowgroup.py."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dgl-0.5-benchmark_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mi355x_graph as mg  # noqa: E402
from mi355x_graph import _lib, sparse  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(4):
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


print("MGX_ROWGROUP=%s" % os.environ.get("MGX_ROWGROUP", "(policy)"))
gen = torch.Generator(device=dev).manual_seed(1)
for n_dst, n_src, avg in ((980000, 310000, 3.4), (310000, 980000, 11.0), (310000, 997000, 3.2), (169343, 169343, 6.9), (169343, 169343, 13.7),
                          (600000, 600000, 20.0), (2449029, 2449029, 50.5)):
    nnz = int(n_dst * avg)
    dst = torch.randint(0, n_dst, (nnz,), device=dev, generator=gen)
    src = torch.randint(0, n_src, (nnz,), device=dev, generator=gen)
    g = mg.create_block((src, dst), n_src, n_dst, idtype=torch.int32, device=dev)
    csc = g._index.csc()
    for D in (16, 64, 100, 128):
        x = torch.rand(n_src, D, device=dev, generator=gen)
        acc = torch.zeros(n_dst, D, device=dev)
        t = timed(lambda: sparse.gspmm_raw(csc, "copy_lhs", "sum", x, None, accumulate_into=acc))
        print("rows %8d cols %8d avg %5.1f D %4d  %-10s %8.4f ms" % (n_dst, n_src, avg, D, _lib.lib().mgx_last_spmm_kernel().decode(), t))
    del g, csc
