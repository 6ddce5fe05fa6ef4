"""TEST-ONLY stand-in for ogb.linkproppred (cluster_gcn_dgl.py:15): an ogbl-citation-shaped toy (directed citations,
128-wide node features, the source/target/target_neg split layout)."""
import torch

from mi355x_graph.datasets import synthetic_edges
import dgl


class DglLinkPropPredDataset(object):
    def __init__(self, name, root="dataset"):
        n, m = 2500, 15000
        src, dst = synthetic_edges(n, m, 100, 17, symmetric=False)
        g = dgl.graph((src, dst), num_nodes=n)
        gen = torch.Generator().manual_seed(0)
        g.ndata["feat"] = torch.rand(n, 128, generator=gen)
        g.ndata["year"] = torch.randint(1990, 2020, (n, 1), generator=gen)
        self.graph = g
        k = 64
        pick = torch.randperm(g.number_of_edges(), generator=gen)[:3 * k]
        self._split = {}
        for i, part in enumerate(("train", "valid", "test")):
            e = pick[i * k:(i + 1) * k]
            self._split[part] = {"source_node": src[e], "target_node": dst[e],
                                 "target_node_neg": torch.randint(0, n, (k, 1000), generator=gen)}

    def get_edge_split(self):
        return self._split

    def __getitem__(self, i):
        return self.graph

    def __len__(self):
        return 1


class Evaluator(object):
    def __init__(self, name):
        self.name = name

    def eval(self, d):
        pos, neg = d["y_pred_pos"].view(-1, 1), d["y_pred_neg"]
        rank = (neg >= pos).sum(1) + 1
        return {"mrr_list": 1.0 / rank.float()}
