"""Where does a neighbor-sampling iteration spend its time?  (reddit shape, fan-out 10,25, batch 1000)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import torch
import dgl
from mi355x_graph.datasets import NodeData
from mi355x_graph import sampling
dev = torch.device("cuda:0")
data = NodeData("reddit", device=dev)
g = dgl.add_self_loop(data.graph).int(); g.create_formats_()
nid = torch.nonzero(torch.rand(g.number_of_nodes()) < 0.66).flatten()
sampler = dgl.dataloading.MultiLayerNeighborSampler([10, 25])
loader = dgl.dataloading.NodeDataLoader(g, nid, sampler, batch_size=1000, shuffle=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time(); n = 0
    for inp, out, blocks in loader:
        n += 1
        if n == 60: break
    torch.cuda.synchronize(); print("sampling only: %.2f ms / iteration" % ((time.time() - t0) / n * 1e3))
seeds = nid[:1000].to(dev)
for name, fn in [("sample_neighbors(25)", lambda: sampling.sample_neighbors(g, seeds, 25)),
                 ("sample+to_block", lambda: sampler.sample_blocks(g, seeds))]:
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): fn()
        torch.cuda.synchronize(); print("%s: %.2f ms" % (name, (time.time() - t0) / 20 * 1e3))
