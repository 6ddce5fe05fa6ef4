import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import torch, dgl
from mi355x_graph.datasets import NodeData
from mi355x_graph import sparse
import kernel_controls as kc
dev = torch.device("cuda:0")
d = NodeData("proteins", device=dev)
g = d.graph.int().formats(["csc"]).to(dev)
csc = g._index.csc()
plan = csc.plan()
print("items", plan.num_items, "hubs", plan.num_hubs, "slots", plan.num_slots, "order", plan.order_kind)
for D in (32, 64, 96, 128, 256):
    X = torch.rand(g.number_of_nodes(), D, device=dev)
    print("D", D, "ms", round(kc.time_spmm(csc, X, reps=8), 3), flush=True)
