"""Round 4: does the two-part plan (short items on the lane-group kernel, the rest on the wave-per-item kernel) pay on the HEADLINE graph,
where the policy refuses it (products: 78 % of the rows but 22 % of the edges in rows of <= 32 edges, 14.5 edges per short row)?
Forces the split for D = 64 / 100 / 16 and times it against the one-launch schedule."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dgl-0.5-benchmark_amd"))
import torch  # noqa: E402
import kernel_bench  # noqa: E402
from mi355x_graph import _lib, schedule, sparse  # noqa: E402

dev = torch.device("cuda:0")


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


scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
g = kernel_bench.get_graph("products", dev, scale).int()
csc = g._index.csc()
n = g.num_nodes()
gen = torch.Generator(device=dev).manual_seed(1)
for D in (64, 100, 16):
    x = torch.rand(n, D, device=dev, generator=gen)
    acc = torch.zeros(n, D, device=dev)
    for form in ("accumulate", "mean"):
        if form == "accumulate":
            run = lambda: sparse.gspmm_raw(csc, "copy_lhs", "sum", x, None, accumulate_into=acc)
        else:
            run = lambda: sparse.gspmm_raw(csc, "copy_lhs", "mean", x, None)
        csc._short = {}
        t0 = timed(run)
        k0 = _lib.lib().mgx_last_spmm_kernel().decode()
        print("products x%.2f D %3d %-10s one launch %-10s %7.4f ms" % (scale, D, form, k0, t0))
        for limit in (8, 16, 24, 32):
            plan, lens, edges = schedule.split_short_items(csc, csc.plan(), any_share=True, limit=limit)
            csc._short = {nb: plan for nb in (2, 4, 8, 16, 32, 64)}
            t1 = timed(run)
            k1 = _lib.lib().mgx_last_spmm_kernel().decode()
            print("      two-part, short = at most %2d edges (%d items, %.1f %% of the edges, %.1f per item; %d others) %-10s %7.4f ms"
                  % (limit, plan.num_items, 100.0 * edges / csc.nnz, edges / plan.num_items, plan.rest.num_items, k1, t1))
            del plan
