"""Where a GraphedBatchTrainer step spends its time: host padding, input copies, graph replay (device)."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "dgl-0.5-benchmark_amd"))
import torch, torch.nn as nn
import graph_classification as gc
from mi355x_graph.datasets import molhiv_like
from dgl.dataloading import GraphDataLoader
dev = torch.device("cuda:0")
data = molhiv_like(4096)
loader = GraphDataLoader(data, batch_size=256, shuffle=False)
torch.manual_seed(0)
name = sys.argv[1] if len(sys.argv) > 1 else "gcn"
model = gc.convert_masked_batchnorm((gc.GCN if name == "gcn" else gc.GIN)(256, 1, 5, 0.5).to(dev))
opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
n_pad, e_pad = gc.GraphedBatchTrainer.static_shape(data, 256)
tr = gc.GraphedBatchTrainer(model, opt, nn.BCEWithLogitsLoss(), dev, 256, n_pad, e_pad)
model.train()
batches = list(loader)
t0 = time.perf_counter(); batches2 = list(loader); t_collate = (time.perf_counter() - t0) / len(batches2)
tr.step(*batches[0]); torch.cuda.synchronize()
t0 = time.perf_counter()
pads = [tr._pad(bg, lab) for bg, lab in batches]
t_pad = (time.perf_counter() - t0) / len(batches)
t0 = time.perf_counter()
for pad in pads:
    for k, v in pad.items():
        tr.buf[k].copy_(v)
torch.cuda.synchronize()
t_copy = (time.perf_counter() - t0) / len(batches)
t0 = time.perf_counter()
for _ in range(20):
    tr.graph.replay()
torch.cuda.synchronize()
t_replay = (time.perf_counter() - t0) / 20
# eager padded step for comparison
t0 = time.perf_counter()
for _ in range(5):
    opt.zero_grad(set_to_none=True); l = tr._forward_loss(tr.buf); l.backward(); opt.step()
torch.cuda.synchronize()
t_eager = (time.perf_counter() - t0) / 5
print("model %s  n_pad %d e_pad %d | collate %.2f ms  pad %.2f ms  copies %.2f ms  replay %.2f ms  eager padded step %.2f ms  loss %s" %
      (name, n_pad, e_pad, t_collate * 1e3, t_pad * 1e3, t_copy * 1e3, t_replay * 1e3, t_eager * 1e3, float(tr.loss)))
