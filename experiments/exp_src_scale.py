"""Round 4: what folding the backward's 1/deg row scaling into the reversed aggregation would cost.  The lean g-SpMM has a
per-edge scalar slot (WMODE 1: lane j loads src_scale[id_j] with the 64 ids of a chunk and hands it on through ds_bpermute);
here the products-shaped reversed aggregation runs with and without it, against the stand-alone N x 64 scaling pass it would
replace (ops.SageMeanCatFn.backward: dcat[:, K:].mul_(inv_deg))."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dgl-0.5-benchmark_amd"))
import torch  # noqa: E402
import mi355x_graph as mg  # noqa: E402
from mi355x_graph import sparse  # noqa: E402
from mi355x_graph.datasets import SHAPES, synthetic_edges  # noqa: E402

dev = torch.device("cuda:0")
spec = SHAPES["products"]
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n, m = int(spec["n"] * scale), int(spec["m"] * scale)
src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], dev, symmetric=True)
g = mg.graph((src, dst), num_nodes=n).int().formats(["csr", "csc"]).to(dev)
csr = g._index.csr()
inv = g._index.csc().inv_degrees()
x = torch.rand(n, 64, device=dev)
acc = torch.rand(n, 64, device=dev)


def timed(fn, reps=12):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]


print("reversed aggregation, accumulate, D=64            %.4f ms" % timed(lambda: sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None, accumulate_into=acc)))
print("same with src_scale = 1/deg (WMODE 1)             %.4f ms" % timed(lambda: sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None, src_scale=inv, accumulate_into=acc)))
print("stand-alone row scaling x.mul_(inv[:, None])      %.4f ms" % timed(lambda: x.mul_(inv.view(-1, 1))))
a = sparse.gspmm_raw(csr, "copy_lhs", "sum", x * inv.view(-1, 1), None)[0]
b = sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None, src_scale=inv)[0]
print("max |difference| of the two forms                 %.3g (of %.3g)" % (float((a - b).abs().max()), float(a.abs().max())))
