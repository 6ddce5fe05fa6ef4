"""Profile target: copy_u/sum on a named graph with a chosen schedule; run under rocprofv3."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import torch
import dgl
from mi355x_graph import schedule, sparse
from mi355x_graph.datasets import SHAPES, synthetic_edges

name, D, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
split = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda:0")
if name == "banded":
    n, half = 2449029, 25
    base = torch.arange(n, device=dev)
    src = torch.cat([(base + k) % n for k in range(-half, half + 1) if k != 0])
    dst = base.repeat(2 * half)
else:
    spec = SHAPES[name]
    n = spec["n"]
    src, dst = synthetic_edges(n, spec["m"], spec["max_deg"], spec["seed"], dev, symmetric=spec["symmetric"])
g = dgl.graph((src, dst), num_nodes=n).int()
csc = g._index.csc()
if mode == "none":
    csc._plan = None
elif mode == "natural":
    csc._plan = schedule.build_plan(csc, None, split, "natural")
else:
    csc._plan = schedule.build_plan(csc, schedule.locality_order(csc), split, "cluster")
x = torch.rand(n, D, device=dev)
ts = []
for i in range(6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); sparse.gspmm_raw(csc, "copy_lhs", "sum", x, None); e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e))
print(name, D, mode, "ms:", " ".join("%.3f" % t for t in ts))
