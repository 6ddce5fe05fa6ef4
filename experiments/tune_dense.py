"""Records PyTorch TunableOp GEMM selections for the dense layers of the products SAGE model at every row
count bench.py can present: N (1 GPU) and the per-rank owned-node counts of the P = 2, 4, 8 partitions
(deterministic: seeded generator + deterministic partitioner).  The message-passing library is not involved:
the model runs on a self-loop graph so that only the GEMM shapes matter.

  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=/tmp/tune.csv \
  PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=10 python experiments/tune_dense.py
"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import torch
import torch.nn.functional as F
import dgl
import full_graph
from mi355x_graph import dist as mdist
from mi355x_graph.datasets import SHAPES, synthetic_edges

dev = torch.device("cuda:0")
spec = SHAPES["products"]
n = spec["n"]
counts = {n}
if "--only-full" not in sys.argv:
    src, dst = synthetic_edges(n, spec["m"], spec["max_deg"], spec["seed"], dev, symmetric=True)
    for P in (2, 4, 8):
        assign, stats = mdist.partition_nodes(src, dst, n, P)
        c = torch.bincount(assign, minlength=P).tolist()
        print("P=%d owned rows per rank: %s  (edge cut %.3f)" % (P, c, stats["edge_cut"]), flush=True)
        counts.update(c)
    del src, dst
cfg = full_graph.SAGE_CONFIGS["products"]
for rows in sorted(counts):
    ids = torch.arange(rows, device=dev)
    g = dgl.graph((ids, ids), num_nodes=rows).int()
    torch.manual_seed(0)
    model = full_graph.GraphSAGE(spec["feat"], cfg["hidden"], spec["classes"], cfg["num_layers"], cfg["dropout"]).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    x = torch.rand(rows, spec["feat"], device=dev)
    y = torch.randint(0, spec["classes"], (rows,), device=dev)
    idx = torch.nonzero(torch.rand(rows, device=dev) < 0.08).flatten()
    for it in range(3):
        opt.zero_grad()
        loss = F.nll_loss(model(g, x)[idx], y[idx])
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    print("tuned rows=%d" % rows, flush=True)
    del g, model, x, y
    torch.cuda.empty_cache()
try:
    torch.cuda.tunable.write_file()
except Exception as e:
    print("write_file:", e)
