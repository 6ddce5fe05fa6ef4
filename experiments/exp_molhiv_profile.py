"""Host-side profile of the batched small-graph loop (launch/CPU bound): where do the 10 ms per iteration go?"""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import torch
torch.set_num_threads(4)
import torch.nn as nn
import graph_classification as gc
from mi355x_graph.datasets import molhiv_like
from dgl.dataloading import GraphDataLoader
dev = torch.device("cuda:0")
data = molhiv_like(4096)
loader = GraphDataLoader(data, batch_size=256, shuffle=False)
model = gc.GCN(256, 1, 5, 0.5).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
loss_fn = nn.BCEWithLogitsLoss()
gc.train_epoch(model, dev, loader, opt, loss_fn)
torch.cuda.synchronize()
t0 = time.time(); gc.train_epoch(model, dev, loader, opt, loss_fn); torch.cuda.synchronize()
print("epoch of 16 iterations: %.1f ms / iteration" % ((time.time() - t0) / 16 * 1e3))
# collate alone
t0 = time.time()
for _ in loader: pass
print("collate only: %.1f ms / iteration" % ((time.time() - t0) / 16 * 1e3))
pr = cProfile.Profile(); pr.enable()
gc.train_epoch(model, dev, loader, opt, loss_fn); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(40)
print("== by cumulative time")
st.sort_stats("cumulative").print_stats(70)
