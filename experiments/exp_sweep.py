"""Sweep split threshold x items-per-workgroup for the clustered schedule (products shape)."""
# NOTE (round 4): the C library reads its A/B switches ONCE per process (common.h MGX_ENV_FLAG): the in-process sweeps below
# recorded the round-1/2 numbers; to repeat them now, run one process per setting.
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import torch
import dgl
from mi355x_graph import schedule, sparse
from mi355x_graph.datasets import SHAPES, synthetic_edges
dev = torch.device("cuda:0")
spec = SHAPES["products"]
src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], dev, symmetric=True)
g = dgl.graph((src, dst), num_nodes=spec["n"]).int()
csc = g._index.csc()
n = csc.num_rows
def run(x, reps=7):
    ts = []
    for i in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); sparse.gspmm_raw(csc, "copy_lhs", "sum", x, None); e.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(s.elapsed_time(e))
    return sum(ts) / len(ts)
for rounds in (3, 5, 8):
    order = schedule.locality_order(csc, rounds=rounds)
    for D in (64, 100):
        x = torch.rand(n, D, device=dev)
        for split in (128, 256, 512):
            csc._plan = schedule.build_plan(csc, order, split, "cluster")
            cells = []
            for rpb in (8, 16, 32):
                os.environ["MGX_ROWS_PER_BLOCK"] = str(rpb)
                cells.append("rpb%d:%.3f" % (rpb, run(x)))
            print("LP rounds=%d D=%d split=%d  %s" % (rounds, D, split, "  ".join(cells)), flush=True)
