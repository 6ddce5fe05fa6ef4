"""The D = 100 layer-1 aggregation of the (constant) products features as a compact [N, 96] block + the last four columns laid out along the
edge list (mgx_spmm_copy_u_edge_tail) against the one-matrix forms: 400-byte rows at a 400-byte stride, at the 800-byte stride of the layer's [x | neigh] buffer, padded to 512 bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import torch
import dgl  # noqa
import kernel_controls as kc
from mi355x_graph import _lib, sparse
dev = torch.device("cuda:0")
n, (src, dst) = kc.control_edges("products", dev, float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
g = dgl.graph((src, dst), num_nodes=n).int().formats(["csc"]).to(dev)
del src, dst
csc = g._index.csc()
be = sparse.backend_for(csc.indptr)

def timed(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps

x = torch.rand(n, 100, device=dev)
wide = torch.zeros(n, 200, device=dev)
wide[:, :100] = x
ref = torch.empty(n, 100, device=dev)
print("one matrix, 400-byte stride            %.3f ms" % timed(lambda: be.spmm_copy_u_strided(csc, "mean", x, ref)))
print("one matrix, 800-byte stride (the layer) %.3f ms" % timed(lambda: be.spmm_copy_u_strided(csc, "mean", wide[:, :100], wide[:, 100:])))
print("# (two compact blocks, 96 + 4 columns, the second gathered by one lane: 4.68 ms -- a fourth REQUEST per edge costs the same whether it\n"
      "#  goes to HBM or to a 39 MB array in the Infinity Cache; variant removed)")
x96 = x[:, :96].contiguous()
o96 = torch.empty(n, 96, device=dev)
print("[N, 96] compact alone (3 lines per edge)  %.3f ms" % timed(lambda: be.spmm_copy_u_strided(csc, "mean", x96, o96)))
ops = be.edge_tail_of(csc, x)
assert ops is not None
a, tail = ops
out = torch.empty(n, 200, device=dev)[:, 100:]
be.spmm_copy_u_edge_tail(csc, "mean", a, tail, out)
torch.cuda.synchronize()
print("[N, 96] + last 4 columns along the edges  %.3f ms   max |diff| / max |ref| = %.1e; columns 0 .. 95 bit-identical: %s"
      % (timed(lambda: be.spmm_copy_u_edge_tail(csc, "mean", a, tail, out)), float((out - ref).abs().max() / ref.abs().max()),
         bool(torch.equal(out[:, :96], ref[:, :96]))))
print("laying the operands out (once)            %.3f ms" % timed(lambda: be.edge_tail_of(csc, x), reps=3))
