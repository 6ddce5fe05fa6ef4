"""Host-side cost of one operator call on a small (molhiv-batch-sized) graph: wall time per call with the GPU kept busy
(launch-bound regime) and a cProfile breakdown of the Python layers above the C ABI."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import torch
import dgl
import dgl.function as fn
from mi355x_graph import ops, sparse

dev = torch.device("cuda:0")
n, e, D = 6500, 14000, 256
src = torch.randint(0, n, (e,), device=dev)
dst = torch.randint(0, n, (e,), device=dev)
g = dgl.graph((src, dst), num_nodes=n).int()
x = torch.rand(n, D, device=dev)
w = torch.rand(e, D, device=dev)
g.ndata["h"] = x
g.edata["w"] = w


def loop(f, reps=2000):
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6


cases = {
    "raw gspmm copy_u/sum": lambda: sparse.gspmm_raw(g._index.csc(), "copy_lhs", "sum", x, None),
    "ops.gspmm copy_u/sum": lambda: ops.gspmm(g, "copy_lhs", "sum", x, None),
    "update_all(copy_u,sum)": lambda: g.update_all(fn.copy_u("h", "m"), fn.sum("m", "o")),
    "update_all(u_add_e,sum)": lambda: g.update_all(fn.u_add_e("h", "w", "m"), fn.sum("m", "o")),
    "torch x+x (reference)": lambda: x + x,
}
for name, f in cases.items():
    host, total = loop(f)
    print("%-28s host %6.1f us/call   end-to-end %6.1f us/call" % (name, host, total))

pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    g.update_all(fn.copy_u("h", "m"), fn.sum("m", "o"))
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

# ---- the same raw call through the dispatcher op torch.ops.mi355x_graph.gspmm: C++ (csrc/torch_bind.cpp) unless
# (a tree without libmi355x_graph_torch.so falls back to the Python registration)
from mi355x_graph import torch_ops
args = torch_ops.csr_args(g._index.csc())
ph = torch_ops.plan_handle(g._index.csc()) if torch_ops.NATIVE else 0
host, total = loop(lambda: torch.ops.mi355x_graph.gspmm(*args, "copy_lhs", "sum", x, None, ph))
print("%-28s host %6.1f us/call   end-to-end %6.1f us/call" % ("torch.ops gspmm (%s)" % ("C++" if torch_ops.NATIVE else "Python"), host, total))
host, total = loop(lambda: torch_ops.raw_gspmm(g._index.csc(), "copy_lhs", "sum", x, None))
print("%-28s host %6.1f us/call   end-to-end %6.1f us/call" % ("torch_ops.raw_gspmm", host, total))
os.environ["MGX_TORCH_OPS"] = "1"
host, total = loop(lambda: ops.gspmm(g, "copy_lhs", "sum", x, None))
print("%-28s host %6.1f us/call   end-to-end %6.1f us/call" % ("ops.gspmm via torch.ops", host, total))
