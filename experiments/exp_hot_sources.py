"""How much of the gather traffic of a cluster goes to its K hottest source rows?  (Upper bound of what an LDS-resident
software cache of K rows per workgroup could take away from L2.)  products shape."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import torch
import dgl
from mi355x_graph import schedule
from mi355x_graph.datasets import SHAPES, synthetic_edges

spec = SHAPES["products"]
dev = torch.device("cuda:0")
src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], dev, symmetric=spec["symmetric"])
n = spec["n"]
g = dgl.graph((src, dst), num_nodes=n).int()
csc = g._index.csc()
labels = schedule.label_propagation(csc.indptr, csc.indices, n, 5)[-1]
deg = (csc.indptr[1:] - csc.indptr[:-1]).long()
rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
cols = csc.indices.long()
lab_dst = labels[rows]
E = cols.shape[0]
# count per (cluster of the destination, source row)
key = lab_dst * n + cols
ukey, cnt = torch.unique(key, return_counts=True)
ulab = torch.div(ukey, n, rounding_mode="floor")
# rank inside the cluster by count (descending)
o = torch.sort(cnt, descending=True, stable=True)[1]
o = o[torch.sort(ulab[o], stable=True)[1]]
lab_s, cnt_s = ulab[o], cnt[o]
first = torch.ones_like(lab_s, dtype=torch.bool)
first[1:] = lab_s[1:] != lab_s[:-1]
start = torch.cummax(torch.where(first, torch.arange(lab_s.shape[0], device=dev), torch.zeros_like(lab_s)), 0)[0]
rank = torch.arange(lab_s.shape[0], device=dev) - start
print("clusters %d, distinct (cluster, source) pairs %.1f M of %.1f M edges" % (int(first.sum()), ukey.shape[0] / 1e6, E / 1e6))
for K in (64, 128, 256, 512, 1024):
    hot = cnt_s[rank < K].sum().item()
    print("K = %4d hottest sources per cluster receive %.1f %% of all gathers (LDS %d KB at D = 64)" % (K, 100.0 * hot / E, K * 256 // 1024))
