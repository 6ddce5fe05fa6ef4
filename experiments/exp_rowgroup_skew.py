"""Round 4: the row-per-lane-group kernel on SKEWED short-row graphs (the arxiv / reddit-small / molhiv stand-ins of
kernel_bench.py, power-law in-degrees) against the row-per-wave kernel, by width, with the batch imbalance CsrView.short_rows()
decides on.  One process per setting (MGX_ROWGROUP is read once): python exp_rowgroup_skew.py runs itself three times."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dgl-0.5-benchmark_amd"))

if len(sys.argv) == 1:
    for forced in ("0", "1", None):   # never / whenever eligible / the host layer's policy (CsrView.short_rows)
        env = dict(os.environ)
        env.pop("MGX_ROWGROUP", None)
        if forced is not None:
            env["MGX_ROWGROUP"] = forced
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
    sys.exit(0)

import torch  # noqa: E402
import kernel_bench  # noqa: E402
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


def imbalance(csc, nb, clip=32):
    plan = csc.plan()
    lens = (plan.item_end - plan.item_beg) if plan is not None else (csc.indptr[1:] - csc.indptr[:-1])
    lens = lens.clamp(max=clip)
    pad = (-int(lens.shape[0])) % nb
    if pad:
        lens = torch.cat([lens, lens.new_zeros(pad)])
    return float(lens.view(-1, nb).max(dim=1)[0].sum()) * nb / float(lens.sum()), csc.nnz / int(lens.shape[0])


print("MGX_ROWGROUP=%s" % os.environ.get("MGX_ROWGROUP", "(policy)"))
gen = torch.Generator(device=dev).manual_seed(1)
for name, scale in (("arxiv", 1.0), ("reddit-small", 0.1), ("arxiv", 4.0)):
    g = kernel_bench.get_graph(name, dev, scale).int()
    csc = g._index.csc()
    n = g.num_nodes()
    for D in (4, 8, 16, 32, 64, 128):
        G = 1
        while G < D // 4:
            G *= 2
        nb = 64 // G
        x = torch.rand(n, D, device=dev, generator=gen)
        acc = torch.zeros(n, D, device=dev)
        t = timed(lambda: sparse.gspmm_raw(csc, "copy_lhs", "sum", x, None, accumulate_into=acc))
        imb, avg = imbalance(csc, max(nb, 2))
        plan, short = csc.spmm_plan_for(D)
        form = "-" if not short else ("two-part plan (%d short + %d other items)" % (plan.num_items, plan.rest.num_items)
                                      if plan is not None and plan.rest is not None else "whole schedule")
        print("%-12s x%.1f n %8d avg item %5.1f D %4d  batch %2d  imbalance(clip 32) %5.2f  %-10s %8.4f ms  policy: %s"
              % (name, scale, n, avg, D, nb, imb, _lib.lib().mgx_last_spmm_kernel().decode(), t, form))
    del g, csc
