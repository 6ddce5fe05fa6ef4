"""u_add_v g-SDDMM on a dense-neighbourhood graph: the COO walk (edge-id order, two random gathers per edge) against the CSR
walk over the locality schedule (row operand loaded once per row, neighbours gathered with the g-SpMM's L2 hit rate, output
rows scattered by edge id).   python experiments/exp_sddmm_walks.py [reddit] [16,32,64,128]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import mi355x_graph as mg  # noqa: E402
from mi355x_graph import ops  # noqa: E402
from mi355x_graph.datasets import SHAPES, synthetic_edges  # noqa: E402


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


name = sys.argv[1] if len(sys.argv) > 1 else "reddit"
widths = [int(w) for w in (sys.argv[2] if len(sys.argv) > 2 else "16,32,64,128").split(",")]
dev = torch.device("cuda")
spec = SHAPES[name]
n, m = spec["n"], spec["m"]
src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], dev, symmetric=spec["symmetric"])
g = mg.graph((src, dst), num_nodes=n).int().to(dev)
gc = g.formats(["csr", "csc"])
E = g.number_of_edges()
for D in widths:
    u, v = torch.rand(n, D, device=dev), torch.rand(n, D, device=dev)
    ref = ops.gsddmm(g, "add", u, v)
    out = ops.gsddmm(gc, "add", u, v)
    assert torch.equal(ref, out)
    t0 = timeit(lambda: ops.gsddmm(g, "add", u, v))
    t1 = timeit(lambda: ops.gsddmm(gc, "add", u, v))
    print("D = %4d: COO walk %.3f ms (%.2f TB/s written) | CSR walk %.3f ms (%.2f TB/s)" %
          (D, t0, E * D * 4 / t0 / 1e9, t1, E * D * 4 / t1 / 1e9), flush=True)
