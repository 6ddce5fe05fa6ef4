"""Does relabelling the nodes by locality order (cluster members contiguous in memory, what dgl.reorder_graph does) help the
g-SpMM beyond scheduling the rows in that order?  products shape, copy_u/sum."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import torch
import dgl
from mi355x_graph import schedule, sparse
from mi355x_graph.datasets import SHAPES, synthetic_edges

name = sys.argv[1] if len(sys.argv) > 1 else "products"
Ds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "64,100").split(",")]
spec = SHAPES[name]
dev = torch.device("cuda:0")
src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], dev, symmetric=spec["symmetric"])
n = spec["n"]


def timeit(csc, D, reps=12):
    x = torch.rand(csc.num_cols, D, device=dev)
    ts = []
    for i in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        sparse.gspmm_raw(csc, "copy_lhs", "sum", x, None)
        e.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(s.elapsed_time(e))
    return sum(ts) / len(ts)


g = dgl.graph((src, dst), num_nodes=n).int()
csc = g._index.csc()
csc.plan()
order = csc._row_order[0]
for D in Ds:
    print("original ids, cluster schedule      D=%3d  %.3f ms" % (D, timeit(csc, D)), flush=True)
rank = torch.empty(n, dtype=torch.int64, device=dev)
rank[order.long()] = torch.arange(n, device=dev)
g2 = dgl.graph((rank[src], rank[dst]), num_nodes=n).int()
csc2 = g2._index.csc()
for mode in ("natural", "auto"):
    os.environ["MGX_SCHEDULE"] = mode
    csc2._plan = False
    csc2._row_order = (None, None)
    for D in Ds:
        print("relabelled ids, %-8s schedule    D=%3d  %.3f ms" % (mode, D, timeit(csc2, D)), flush=True)
