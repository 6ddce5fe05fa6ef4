"""How many times would a source row be reused if R consecutive rows of the schedule were aggregated together (an
LDS / register tile)?  reuse(R) = edges of the tile / distinct sources of the tile, averaged over tiles and weighted by
edges.  Runs on the CPU (graph structure only).   python experiments/exp_tile_reuse.py reddit|proteins|products [scale]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
from mi355x_graph.datasets import SHAPES, synthetic_edges  # noqa: E402
from mi355x_graph.schedule import label_propagation  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "reddit"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
spec = SHAPES[name]
n, m = int(spec["n"] * scale), int(spec["m"] * scale)
t0 = time.time()
src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], "cpu", symmetric=spec["symmetric"])
print("graph", n, src.numel(), "%.1fs" % (time.time() - t0), flush=True)
order = torch.argsort(dst, stable=True)
indices = src[order]
deg = torch.bincount(dst, minlength=n)
indptr = torch.zeros(n + 1, dtype=torch.int64)
indptr[1:] = torch.cumsum(deg, 0)
t0 = time.time()
hist = label_propagation(indptr, indices, n, 8)
perm = torch.arange(n)
for labels in hist[-3:]:
    perm = perm[torch.sort(labels[perm], stable=True)[1]]
print("schedule %.1fs, clusters %d" % (time.time() - t0, int(torch.unique(hist[-1]).numel())), flush=True)
pos = torch.empty(n, dtype=torch.int64)
pos[perm] = torch.arange(n)
row_of_edge = torch.repeat_interleave(torch.arange(n), deg)
E = indices.numel()
for label, p in (("natural", torch.arange(n)), ("schedule", pos)):
    for R in (16, 64, 128, 256, 512, 1024):
        tile = (p[row_of_edge] // R).numpy()
        key = tile * np.int64(n) + indices.numpy()
        distinct = np.unique(key).shape[0]
        print("%-9s R = %3d: edges / distinct sources per tile = %.2f" % (label, R, E / distinct), flush=True)
