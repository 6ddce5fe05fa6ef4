"""A/B of kernel variants selected by environment variables (default: 64-bit row-per-wave kernel vs the lean 32-bit one),
by graph / D / schedule."""
# NOTE (round 4): the C library reads its A/B switches ONCE per process (common.h MGX_ENV_FLAG): the in-process sweeps below
# recorded the round-1/2 numbers; to repeat them now, run one process per setting.
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import torch
import dgl
from mi355x_graph import schedule, sparse
from mi355x_graph.datasets import SHAPES, synthetic_edges

dev = torch.device("cuda:0")
names = sys.argv[1].split(",")
Ds = [int(x) for x in sys.argv[2].split(",")]
envs = [e for e in (sys.argv[3] if len(sys.argv) > 3 else "MGX_SPMM_V1=1;").split(";")]

def make(name):
    if name == "banded":
        n, half = 2449029, 25
        base = torch.arange(n, device=dev)
        return n, torch.cat([(base + k) % n for k in range(-half, half + 1) if k != 0]), base.repeat(2 * half)
    spec = SHAPES[name]
    s, d = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], dev, symmetric=spec["symmetric"])
    return spec["n"], s, d

def run(csc, x, reps=6):
    ts = []
    for i in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); sparse.gspmm_raw(csc, "copy_lhs", "sum", x, None); e.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(s.elapsed_time(e))
    return sum(ts) / len(ts)

for name in names:
    n, src, dst = make(name)
    g = dgl.graph((src, dst), num_nodes=n).int()
    csc = g._index.csc()
    order = schedule.locality_order(csc) if name != "banded" else None
    plans = {"natural+split256": schedule.build_plan(csc, None, 256, "natural")}
    if order is not None:
        plans["cluster+split256"] = schedule.build_plan(csc, order, 256, "cluster")
    for D in Ds:
        x = torch.rand(n, D, device=dev)
        for pname, plan in plans.items():
            csc._plan = plan
            cells = []
            for env in envs:
                saved = dict(os.environ)
                for kv in env.split(","):
                    if "=" in kv:
                        k, v = kv.split("="); os.environ[k] = v
                cells.append("[%s] %.3f" % (env or "default", run(csc, x)))
                os.environ.clear(); os.environ.update(saved)
            print("%s nnz=%d D=%d %-18s %s" % (name, csc.nnz, D, pname, "  ".join(cells)), flush=True)
        del x
