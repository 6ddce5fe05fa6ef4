"""How much of the remaining SpMM time is community size?  Same generator, smaller planted communities."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import torch
import dgl
from mi355x_graph import schedule, sparse
from mi355x_graph.datasets import SHAPES, synthetic_edges
dev = torch.device("cuda:0")
spec = SHAPES["products"]
for avg in (12520, 6000, 3000, 1500, 500):
    for mixing in (0.25, 0.10):
        src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], dev, symmetric=True, avg_comm=avg, mixing=mixing)
        g = dgl.graph((src, dst), num_nodes=spec["n"]).int()
        csc = g._index.csc()
        csc._plan = schedule.build_plan(csc, schedule.locality_order(csc), 256, "cluster")
        x = torch.rand(csc.num_rows, 64, device=dev)
        ts = []
        for i in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); sparse.gspmm_raw(csc, "copy_lhs", "sum", x, None); e.record(); torch.cuda.synchronize()
            if i >= 2: ts.append(s.elapsed_time(e))
        print("avg community %6d nodes (%.1f MB at D=64), mixing %.2f: %.3f ms" % (avg, avg * 256 / 1e6, mixing, sum(ts) / len(ts)), flush=True)
        del g, csc, x, src, dst
        torch.cuda.empty_cache()
