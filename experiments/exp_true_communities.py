"""How much of the L2 locality that EXISTS in the benchmark graph does the label-propagation schedule find?
Ground truth: the generator's planted communities.  Compares the g-SpMM (D = 64) under (a) the shipped schedule (label
propagation), (b) rows ordered by their TRUE planted community, (c) natural order; reports how well the LP clusters match."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import torch, dgl
from mi355x_graph import schedule, sparse
from mi355x_graph.datasets import SHAPES, synthetic_edges
import kernel_controls as kc
dev = torch.device("cuda:0")
spec = SHAPES["products"]
n, m = spec["n"], spec["m"]
src, dst, comm = synthetic_edges(n, m, spec["max_deg"], spec["seed"], dev, symmetric=True, return_communities=True)
print(json.dumps({"planted_communities": int(comm.max()) + 1, "intra_planted_edge_frac": float((comm[src] == comm[dst]).float().mean())}))
g = dgl.graph((src, dst), num_nodes=n).int().formats(["csc"]).to(dev)
csc = g._index.csc()
X = torch.rand(n, 64, device=dev)
ms = kc.time_spmm(csc, X); kc._marker(dev)
hist = schedule.label_propagation(csc.indptr, csc.indices, n, 5)
lab = hist[-1]
print(json.dumps({"variant": "label propagation (shipped)", "ms": round(ms, 3), "clusters": int(torch.unique(lab).numel()),
                  "intra_cluster_edge_frac": float((lab[src] == lab[dst]).float().mean())}), flush=True)
def run(order, name):
    c2 = sparse.CsrView(n, n, csc.indptr, csc.indices, None)
    c2._row_order = (order, "cluster")
    c2._plan = schedule.build_plan(c2, order, 256, "cluster")
    ms = kc.time_spmm(c2, X); kc._marker(dev)
    print(json.dumps({"variant": name, "ms": round(ms, 3)}), flush=True)
run(torch.sort(comm, stable=True)[1], "rows ordered by TRUE planted community")
# true community, then by degree inside the community (hubs first)
deg = (csc.indptr[1:] - csc.indptr[:-1]).long()
key = comm * (int(deg.max()) + 1) + (int(deg.max()) - deg)
run(torch.sort(key, stable=True)[1], "true community, high degree first")
run(torch.arange(n, device=dev), "natural order (hub split only)")
for rounds in (8, 12):
    h = schedule.label_propagation(csc.indptr, csc.indices, n, rounds)
    order = torch.arange(n, device=dev)
    for labels in h[-3:]:
        order = order[torch.sort(labels[order], stable=True)[1]]
    l = h[-1]
    print(json.dumps({"lp_rounds": rounds, "clusters": int(torch.unique(l).numel()), "intra_cluster_edge_frac": float((l[src] == l[dst]).float().mean())}))
    run(order, "label propagation, %d rounds" % rounds)
