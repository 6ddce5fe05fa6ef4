"""Round 4: the tile kernel's direct part (sources used < tau times in a tile) inside the kernel (MGX_TILE_DIRECT=kernel) against a second,
accumulating launch of the row-per-wave kernel over those edges (MGX_TILE_DIRECT=split, default).  One process per setting.
  python exp_tile_direct.py reddit|proteins [widths]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dgl-0.5-benchmark_amd"))
import torch  # noqa: E402
import mi355x_graph as mg  # noqa: E402
from mi355x_graph import sparse  # noqa: E402
from mi355x_graph.datasets import SHAPES, synthetic_edges  # noqa: E402

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "reddit"
widths = [int(w) for w in (sys.argv[2] if len(sys.argv) > 2 else "64,128,256").split(",")]
spec = SHAPES[name]
src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], dev, symmetric=spec["symmetric"])
g = mg.graph((src, dst), num_nodes=spec["n"]).int().formats(["csr", "csc"]).to(dev)
csc = g._index.csc()
print("%s: N %d E %d  MGX_TILE_DIRECT=%s MGX_TILE=%s" % (name, spec["n"], src.shape[0], os.environ.get("MGX_TILE_DIRECT", "split"), os.environ.get("MGX_TILE", "auto")))
for D in widths:
    x = torch.rand(spec["n"], D, device=dev)
    out = torch.empty(spec["n"], D, device=dev)
    be = sparse.backend_for(x)
    fn = lambda: be.spmm_copy_u_strided(csc, "sum", x, out)
    for _ in range(3):
        fn()
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    tp = csc.tile_plan(D)
    extra = ""
    if tp is not None:
        dv = getattr(tp, "direct_csr", None)
        extra = "staged %.1f %% of the edges, direct CSR %s" % (100.0 * tp.stats["staged_edges"] / tp.stats["edges"], "none" if dv is None else "%d edges" % dv.nnz)
    ref = sparse.gspmm_raw(csc, "copy_lhs", "sum", x[:, :4].contiguous(), None)[0] if D >= 4 else None
    print("D %4d  median %.4f ms  best %.4f ms   %s" % (D, ts[len(ts) // 2], ts[0], extra))
