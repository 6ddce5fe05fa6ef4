"""Does running fc_self(h) on a second HIP stream next to the g-SpMM shorten the products SAGE epoch?"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "0")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(ROOT, "dgl-0.5-benchmark_amd", "tunableop_products.csv"))
import torch
import torch.nn.functional as F
import dgl.function as fn
import full_graph

dev = torch.device("cuda:0")
cfg, data, g, model, train_idx, opt = full_graph.build_sage("products", dev)
side = torch.cuda.Stream()


def forward_two_streams(self, graph, feat):
    graph = graph.local_var()
    feat_src, feat_dst = feat if isinstance(feat, tuple) else (feat, feat)
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        self_out = self.fc_self(feat_dst)
    graph.srcdata["h"] = feat_src
    graph.update_all(fn.copy_src("h", "m"), fn.mean("m", "neigh"))
    cur.wait_stream(side)
    self_out.record_stream(cur)
    return self_out + self.fc_neigh(graph.dstdata["neigh"])


def epochs(n=12):
    ts = []
    for i in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        full_graph.sage_train_step(model, g, data.features, data.labels, train_idx, opt)
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(time.perf_counter() - t0)
    return 1e3 * sum(ts) / len(ts)


print("one stream  : %.2f ms / epoch" % epochs())
orig = full_graph.SAGEConv.forward
full_graph.SAGEConv.forward = forward_two_streams
print("two streams : %.2f ms / epoch" % epochs())
full_graph.SAGEConv.forward = orig
print("one stream  : %.2f ms / epoch" % epochs())
