"""Experiment: rows-per-workgroup x schedule on the products-shaped graph (copy_u/sum)."""
# NOTE (round 4): the C library reads its A/B switches ONCE per process (common.h MGX_ENV_FLAG): the in-process sweeps below
# recorded the round-1/2 numbers; to repeat them now, run one process per setting.
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import torch
import dgl
from mi355x_graph import schedule, sparse
from mi355x_graph.datasets import SHAPES, synthetic_edges

name = sys.argv[1] if len(sys.argv) > 1 else "products"
Ds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "64").split(",")]
spec = SHAPES[name]
dev = torch.device("cuda:0")
src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], dev, symmetric=spec["symmetric"])
g = dgl.graph((src, dst), num_nodes=spec["n"]).int()
csc = g._index.csc()
n, nnz = csc.num_rows, csc.nnz
print(name, n, nnz)

def run(D, reps=6):
    x = torch.rand(n, D, device=dev)
    ts = []
    for i in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); sparse.gspmm_raw(csc, "copy_lhs", "sum", x, None); e.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(s.elapsed_time(e))
    return sum(ts) / len(ts)

t0 = time.time(); order = schedule.locality_order(csc); torch.cuda.synchronize(); print("locality_order: %.2fs" % (time.time() - t0))
plans = {"none": None,
         "natural+split": schedule.build_plan(csc, None, 1024, "natural"),
         "cluster+split": schedule.build_plan(csc, order, 1024, "cluster"),
         "cluster+split256": schedule.build_plan(csc, order, 256, "cluster")}
for D in Ds:
    for pname, plan in plans.items():
        csc._plan = plan
        row = []
        for rpb in (4, 8, 16, 32, 64, 256):
            os.environ["MGX_ROWS_PER_BLOCK"] = str(rpb)
            row.append("%d:%.3f" % (rpb, run(D)))
        print("D=%d %-18s ms by rows/block  %s" % (D, pname, "  ".join(row)), flush=True)
