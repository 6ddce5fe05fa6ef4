"""The slot g-SpMM (csrc/spmm_slots.inc) on the products graph, D = 64, relu + dropout-like input: whole call and by part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import torch
import dgl  # noqa
from mi355x_graph import _lib, sparse
from mi355x_graph.datasets import SHAPES, synthetic_edges
dev = torch.device("cuda:0")
spec = SHAPES["products"]
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n, m = int(spec["n"] * scale), int(spec["m"] * scale)
print("# products shape x %.3f: N = %d, E = %d directed" % (scale, n, 2 * m))
src, dst = synthetic_edges(n, m, spec["max_deg"], spec["seed"], dev, symmetric=True)
g = dgl.graph((src, dst), num_nodes=n).int().formats(["csr", "csc"]).to(dev)
del src, dst
csc = g._index.csc()
be = sparse.backend_for(csc.indptr)
x = torch.rand(n, 128, device=dev)[:, :64]
x.mul_((torch.rand(n, 64, device=dev) < 0.21).float())
out = torch.empty(n, 128, device=dev)[:, 64:]

def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps

slots, ovf = be.rows_slots_pack(x)
print("overflow rows %.3f %%" % (100.0 * int(ovf) / n))
print("dense call            %.3f ms" % timed(lambda: be.spmm_copy_u_strided(csc, "mean", x, out)))
print("pack pass             %.3f ms" % timed(lambda: be.rows_slots_pack(x)))
print("slot call             %.3f ms   (U = 2, two steps in flight; the sweep over U / ping-pong is recorded in csrc/spmm_slots.inc)"
      % timed(lambda: be.spmm_copy_u_strided(csc, "mean", x, out, slots=slots)))
