import os, sys
sys.path.insert(0, "/root/repo/dgl-0.5-benchmark_amd")
import torch, dgl
from mi355x_graph import sparse
from mi355x_graph.datasets import SHAPES, synthetic_edges
spec = SHAPES["reddit"]; dev = torch.device("cuda:0")
src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], dev, symmetric=spec["symmetric"])
g = dgl.add_self_loop(dgl.graph((src, dst), num_nodes=spec["n"])).int()
cidx, perm = g._index.canonical()
def t(f, reps=10):
    ts=[]
    for i in range(reps):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        if i>=2: ts.append(s.elapsed_time(e))
    return sum(ts)/len(ts)
for H in (1, 8):
    de = torch.rand(cidx.num_edges(), H, device=dev)
    print("H=%d  d_er (csc, contiguous) %.3f ms   d_el (csr, eids gather) %.3f ms   torch segment view: %.3f ms" % (
        H, t(lambda: sparse.gspmm_raw(cidx.csc(), "copy_rhs", "sum", None, de)), t(lambda: sparse.gspmm_raw(cidx.csr(), "copy_rhs", "sum", None, de)),
        t(lambda: torch.segment_reduce(de, "sum", offsets=cidx.csc().indptr.long(), axis=0))))
