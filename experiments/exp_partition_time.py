"""How long the edge-cut partitioner takes on the products-shaped graph, and what it yields (P = 2, 4, 8)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import torch
from mi355x_graph import dist as mdist
from mi355x_graph.datasets import SHAPES, synthetic_edges

dev = torch.device("cuda:0")
spec = SHAPES["products"]
scale = float(os.environ.get("SCALE", "1"))
n, m = int(spec["n"] * scale), int(spec["m"] * scale)
src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], dev, symmetric=True)
torch.cuda.synchronize()
for P in [int(p) for p in os.environ.get("PARTS", "2,4,8").split(",")]:
    t0 = time.perf_counter()
    assign, stats = mdist.partition_nodes(src, dst, n, P)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    load = torch.bincount(assign[dst], minlength=P).float()
    own = torch.bincount(assign, minlength=P)
    t2 = time.perf_counter()
    blk, plan, _ = mdist.build_local_partition(src, dst, n, assign, 0, P)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(json.dumps({"P": P, "partition_s": round(t1 - t0, 2), "build_local_s": round(t3 - t2, 2), **stats,
                      "edge_load_max_over_mean": round(float(load.max() / load.mean()), 3),
                      "owned_min_max": [int(own.min()), int(own.max())], "rank0_halo_rows": plan.n_halo}), flush=True)
