"""Per-step host timeline of the graphed loop on the full shuffled dataset."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "dgl-0.5-benchmark_amd"))
import torch, torch.nn as nn
import graph_classification as gc
from mi355x_graph.datasets import molhiv_like
from dgl.dataloading import GraphDataLoader
dev = torch.device("cuda:0")
data = molhiv_like(int(os.environ.get("NG", "32901")))
loader = GraphDataLoader(data, batch_size=256, shuffle=True)
torch.manual_seed(0)
model = gc.convert_masked_batchnorm(gc.GCN(256, 1, 5, 0.5).to(dev))
opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
n_pad, e_pad = gc.GraphedBatchTrainer.static_shape(data, 256)
tr = gc.GraphedBatchTrainer(model, opt, nn.BCEWithLogitsLoss(), dev, 256, n_pad, e_pad)
model.train()
T = {"load": 0.0, "pad": 0.0, "wait": 0.0, "copy": 0.0, "replay": 0.0}
it = iter(loader)
n = 0
t_epoch = time.perf_counter()
while True:
    t0 = time.perf_counter()
    try:
        bg, lab = next(it)
    except StopIteration:
        break
    t1 = time.perf_counter()
    if tr.graph is None:
        tr.step(bg, lab); torch.cuda.synchronize(); continue
    pad = tr._pad(bg, lab)
    t2 = time.perf_counter()
    tr.done.synchronize()
    t3 = time.perf_counter()
    for k, v in pad.items():
        tr.buf[k].copy_(v)
    t4 = time.perf_counter()
    tr.graph.replay(); tr.done.record()
    t5 = time.perf_counter()
    T["load"] += t1 - t0; T["pad"] += t2 - t1; T["wait"] += t3 - t2; T["copy"] += t4 - t3; T["replay"] += t5 - t4
    n += 1
torch.cuda.synchronize()
print("steps", n, "epoch %.3f s" % (time.perf_counter() - t_epoch), {k: round(v / n * 1e3, 3) for k, v in T.items()}, "threads", torch.get_num_threads())
