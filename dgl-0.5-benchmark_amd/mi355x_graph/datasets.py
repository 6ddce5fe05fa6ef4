"""Synthetic stand-ins for the benchmark datasets + the dgl.data surface the scripts use.

The real datasets (reddit, ogbn-*, cora, ...) cannot be downloaded here (no network), so every
loader generates a seeded graph with the published shape (N, E, feature width,
class count) and a heavy-tailed degree distribution with planted communities.  Node ids are randomly
permuted so the natural numbering carries no locality.  Nothing here comes from reference code.

  load_data(args) / RedditDataset   main_dgl_reddit_sage.py:15,151-164,188
  LegacyTUDataset / Subset           main_dgl_enzymes_gcn.py:11,155-163
"""
import math
import os

import numpy as np
import torch

from .graph import graph as make_graph
from . import transform

# name -> (num_nodes, num_undirected_or_directed_edges, symmetric, feat_dim, num_classes, max_degree, seed)
SHAPES = {
    "cora": dict(n=2708, m=5278, symmetric=True, feat=1433, classes=7, max_deg=168, seed=1),
    "citeseer": dict(n=3327, m=4552, symmetric=True, feat=3703, classes=6, max_deg=99, seed=6),
    "pubmed": dict(n=19717, m=44324, symmetric=True, feat=500, classes=3, max_deg=171, seed=7),
    "arxiv": dict(n=169343, m=1166243, symmetric=False, feat=128, classes=40, max_deg=13155, seed=2),
    "reddit": dict(n=232965, m=57307946, symmetric=True, feat=602, classes=41, max_deg=21657, seed=3),
    "reddit-small": dict(n=232965, m=5803459, symmetric=True, feat=602, classes=41, max_deg=21657, seed=3),
    "products": dict(n=2449029, m=61859140, symmetric=True, feat=100, classes=47, max_deg=17481, seed=4),
    "proteins": dict(n=132534, m=39561252, symmetric=True, feat=8, classes=112, max_deg=7750, seed=8),
}


def _power_law_degrees(n, total, max_deg, gen, device, alpha=2.1):
    """Degrees ~ truncated power law, rescaled so they sum to ~`total`."""
    u = torch.rand(n, generator=gen, device=device, dtype=torch.float64)
    lo, hi = 1.0, float(max_deg)
    a = 1.0 - alpha
    d = (lo ** a + u * (hi ** a - lo ** a)) ** (1.0 / a)
    d = d * (total / d.sum())
    return d.clamp(min=0.5, max=float(max_deg))


def synthetic_edges(n, m, max_deg, seed, device="cpu", mixing=0.25, symmetric=True, permute=True, avg_comm=None,
                    return_communities=False):
    """m edges over n nodes: endpoints drawn proportionally to a power-law weight; a fraction
    (1 - mixing) of the edges stays inside the source's planted community.  Returns int64 (src, dst);
    when `symmetric`, both directions are stored (2m directed edges, like OGB's DGL graphs)."""
    device = torch.device(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    w = _power_law_degrees(n, 2.0 * m, max_deg, gen, device)
    # planted communities: contiguous ranges of power-law sizes
    avg_c = avg_comm or max(16, min(n // 8, int(8 * math.sqrt(n))))
    n_comm = max(1, n // avg_c)
    cuts = torch.sort(torch.randint(0, n, (n_comm - 1,), generator=gen, device=device))[0] if n_comm > 1 else \
        torch.zeros(0, dtype=torch.int64, device=device)
    starts = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), cuts])
    ends = torch.cat([cuts, torch.full((1,), n, dtype=torch.int64, device=device)])
    cdf = torch.cumsum(w, 0)
    cdf0 = torch.cat([torch.zeros(1, dtype=cdf.dtype, device=device), cdf])
    total = float(cdf[-1])
    chunks_s, chunks_d = [], []
    step = 1 << 24
    for off in range(0, m, step):
        k = min(step, m - off)
        r = torch.rand(k, generator=gen, device=device, dtype=torch.float64) * total
        s = torch.searchsorted(cdf, r).clamp(max=n - 1)
        comm = (torch.searchsorted(ends, s, right=True)).clamp(max=n_comm - 1)
        lo, hi = cdf0[starts[comm]], cdf0[ends[comm]]
        intra = torch.rand(k, generator=gen, device=device) >= mixing
        r2 = torch.rand(k, generator=gen, device=device, dtype=torch.float64)
        r2 = torch.where(intra, lo + r2 * (hi - lo), r2 * total)
        d = torch.searchsorted(cdf, r2).clamp(max=n - 1)
        chunks_s.append(s)
        chunks_d.append(d)
    src, dst = torch.cat(chunks_s), torch.cat(chunks_d)
    if permute:
        perm = torch.randperm(n, generator=gen, device=device)
        src, dst = perm[src], perm[dst]
    if symmetric:
        src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    if return_communities:  # planted community of every (relabelled) node -- experiments only (ground truth for the schedules)
        comm_of = torch.searchsorted(ends, torch.arange(n, device=device), right=True).clamp(max=n_comm - 1)
        if permute:
            out = torch.empty_like(comm_of)
            out[perm] = comm_of
            comm_of = out
        return src, dst, comm_of
    return src, dst


class NodeData(object):
    """A full-graph node-classification dataset (features / labels / masks / graph)."""

    def __init__(self, name, device="cpu", feat_dim=None, scale=1.0, gen_device=None):
        spec = dict(SHAPES[name])
        scale = scale * float(os.environ.get("MGX_DATASET_SCALE", "1"))  # tests shrink the stand-ins
        n = max(16, int(spec["n"] * scale))
        m = max(16, int(spec["m"] * scale))
        feat = feat_dim or spec["feat"]
        gdev = gen_device or device
        src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], gdev, symmetric=spec["symmetric"])
        self.name, self.num_nodes, self.num_classes, self.num_labels = name, n, spec["classes"], spec["classes"]
        self.graph = make_graph((src.to(device), dst.to(device)), num_nodes=n)
        g = torch.Generator(device="cpu")
        g.manual_seed(spec["seed"] + 100)
        self.features = torch.rand(n, feat, generator=g).to(device)
        self.labels = torch.randint(0, spec["classes"], (n,), generator=g).to(device)
        r = torch.rand(n, generator=g)
        self.train_mask = (r < 0.08).to(device)
        self.val_mask = ((r >= 0.08) & (r < 0.10)).to(device)
        self.test_mask = (r >= 0.10).to(device)
        self.graph.ndata["feat"] = self.features
        self.graph.ndata["label"] = self.labels
        self.graph.ndata["train_mask"] = self.train_mask
        self.graph.ndata["val_mask"] = self.val_mask
        self.graph.ndata["test_mask"] = self.test_mask

    def __getitem__(self, idx):
        assert idx == 0
        return self.graph

    def __len__(self):
        return 1


def RedditDataset(self_loop=False, **kw):
    from . import diskio
    d = diskio.find_dataset("reddit")  # the real files when MGX_DATA_ROOT holds them
    if d is None:
        d = NodeData("reddit-small" if kw.pop("small", False) else "reddit", **kw)
    if self_loop:
        d.graph = transform.add_self_loop(d.graph)
    return d


def CoraGraphDataset(**kw):
    return NodeData("cora", **kw)


def CiteseerGraphDataset(**kw):
    return NodeData("citeseer", **kw)


def PubmedGraphDataset(**kw):
    return NodeData("pubmed", **kw)


def load_data(args):
    """dgl.data.load_data(args): args.dataset in {cora, citeseer, pubmed, reddit*}."""
    name = args.dataset
    if name in ("cora", "citeseer", "pubmed"):
        # the citation scripts expect the legacy layout: `.graph` is a networkx DiGraph that they pass to
        # dgl.from_networkx (main_dgl_citation_sage.py:190)
        import networkx as nx
        d = NodeData(name)
        s, t = d.graph.edges()
        nxg = nx.DiGraph()
        nxg.add_nodes_from(range(d.num_nodes))
        nxg.add_edges_from(zip(s.tolist(), t.tolist()))
        d.graph = nxg
        return d
    if name is not None and name.startswith("reddit"):
        return RedditDataset(self_loop=("self-loop" in name))
    raise ValueError("Unknown dataset: {}".format(name))


# ----------------------------------------------------------------------------- small-graph datasets
def molecule_like_graph(num_nodes, rng):
    """Random tree plus a few ring-closing edges, both directions stored (molhiv-like: E ~ 2.16 N)."""
    n = int(num_nodes)
    if n < 2:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    parent = (rng.random(n - 1) * np.arange(1, n)).astype(np.int64)
    child = np.arange(1, n, dtype=np.int64)
    n_ring = int(round(0.08 * n))
    a = rng.integers(0, n, n_ring)
    b = rng.integers(0, n, n_ring)
    keep = a != b
    u = np.concatenate([parent, a[keep]])
    v = np.concatenate([child, b[keep]])
    return np.concatenate([u, v]), np.concatenate([v, u])


class SmallGraphDataset(object):
    """List-style dataset of (graph, label) pairs for batched graph classification."""

    def __init__(self, num_graphs, mean_nodes, std_nodes, min_nodes, max_nodes, node_feat, edge_feat,
                 num_classes, seed, int_features=True):
        rng = np.random.default_rng(seed)
        sizes = np.clip(np.round(rng.normal(mean_nodes, std_nodes, num_graphs)), min_nodes, max_nodes).astype(np.int64)
        self.graphs, self.labels = [], []
        for n in sizes:
            s, d = molecule_like_graph(n, rng)
            g = make_graph((torch.from_numpy(s), torch.from_numpy(d)), num_nodes=int(n))
            if int_features:
                g.ndata["feat"] = torch.from_numpy(rng.integers(0, 2, (int(n), node_feat)))
                g.edata["feat"] = torch.from_numpy(rng.integers(0, 2, (len(s), edge_feat)))
            else:
                g.ndata["feat"] = torch.from_numpy(rng.random((int(n), node_feat), dtype=np.float32))
            self.graphs.append(g)
            self.labels.append(int(rng.integers(0, num_classes)))
        self.labels = torch.tensor(self.labels, dtype=torch.int64)
        self.num_classes = self.num_labels = num_classes
        self.max_num_node = int(sizes.max())
        # one storage for all graphs: batch() of any subset is then a dozen index operations (transform.GraphPool)
        self.pool = transform.GraphPool.adopt(self.graphs)

    def __getitem__(self, i):
        return self.graphs[i], self.labels[i]

    def __len__(self):
        return len(self.graphs)

    def statistics(self):
        return self.graphs[0].ndata["feat"].shape[1], self.num_classes, self.max_num_node


def LegacyTUDataset(name, **kw):
    if name.upper() != "ENZYMES":
        raise ValueError("only ENZYMES has a synthetic stand-in")
    return SmallGraphDataset(600, 32.63, 15.0, 2, 126, 18, 0, 6, seed=9, int_features=False)


def molhiv_like(num_graphs=32901, seed=5):
    """ogbg-molhiv-shaped training set: 9 int atom features, 3 int bond features."""
    return SmallGraphDataset(num_graphs, 25.5, 12.0, 2, 222, 9, 3, 2, seed=seed, int_features=True)


class Subset(object):
    """dgl.data.utils.Subset"""

    def __init__(self, dataset, indices):
        self.dataset, self.indices = dataset, indices

    def __getitem__(self, item):
        return self.dataset[self.indices[item]]

    def __len__(self):
        return len(self.indices)
