"""Neighbor sampling and message-flow blocks (the callers on the other side of the hot
path -- bipartite "block" graphs on which the same g-SpMM / g-SDDMM kernels run with N_src != N_dst).

  dgl.dataloading.MultiLayerNeighborSampler(fanouts), NodeDataLoader(g, nids, sampler, batch_size=...)
      end_to_end/sampling/node-classification/reddit/ns-sage-dgl.py:132-141,159-165 (training),
      :67-75 (full-neighbour inference with fanout None)
  dgl.to_block, dgl.sampling.sample_neighbors, g.subgraph(mask)   reddit/load_graph.py:45-51

Sampling is index work done with torch integer ops on whatever device holds the graph (the reference samples in CPU
worker processes; sampling next to the features on the GPU removes the host round trip).  Semantics follow
DGL: uniform WITHOUT replacement over the in-edges of each seed, rows with in-degree <= fanout keep all
their edges, destination nodes of a block are a prefix of its source nodes (block.srcdata[NID][:num_dst] ==
block.dstdata[NID]).
"""
import math

import torch

from ._lib import DGLError
from .graph import DGLGraph, GraphIndex

NID = "_ID"
EID = "_ID"


def sample_neighbors(g, nodes, fanout, edge_dir="in", prob=None, replace=False, generator=None):
    """Returns (src, dst, eid) of the sampled in-edges of `nodes` in GLOBAL ids."""
    if edge_dir != "in" or prob is not None or replace:
        raise DGLError("sample_neighbors: only uniform in-edge sampling without replacement is supported")
    idx = g._index if isinstance(g, DGLGraph) else g
    csc = idx.csc()
    dev = csc.device
    nodes = torch.as_tensor(nodes, device=dev).long()
    if nodes.is_cuda and fanout is not None and 1 <= fanout <= 64:
        # device path: mgx_sample_neighbors (one thread per seed, Floyd's algorithm); the torch formulation below
        # sorts every candidate edge (12 M for a 25 k-node reddit frontier) and is kept for CPU graphs
        from . import sparse
        rng_seed = int(torch.randint(0, 2 ** 62, (1,), generator=generator).item()) if (generator is None or generator.device.type == "cpu") \
            else int(torch.randint(0, 2 ** 62, (1,), generator=generator, device=generator.device).item())
        src, eid, counts = sparse.backend_for(csc.indptr).sample_neighbors(csc, nodes.to(csc.indptr.dtype), fanout, rng_seed)
        dst = torch.repeat_interleave(nodes, counts)
        return src.long(), dst, eid.long()
    indptr = csc.indptr.long()
    beg = indptr[nodes]
    deg = indptr[nodes + 1] - beg
    total = int(deg.sum().item())
    if total == 0:
        z = torch.zeros(0, dtype=torch.int64, device=dev)
        return z, z, z
    seg = torch.repeat_interleave(torch.arange(nodes.shape[0], device=dev), deg)
    first = torch.cumsum(deg, 0) - deg
    pos = torch.arange(total, device=dev) - first[seg] + beg[seg]  # CSR positions of every candidate edge
    if fanout is not None and fanout >= 0 and bool((deg > fanout).any()):
        key = torch.rand(total, device=dev, generator=generator, dtype=torch.float64)
        order = torch.argsort(seg.double() + key * 0.999999)  # random order inside each seed's segment
        rank = torch.arange(total, device=dev) - first[seg[order]]
        keep = order[rank < fanout]
        keep, _ = torch.sort(keep)  # keep CSR order: deterministic block layout for a given draw
        pos, seg = pos[keep], seg[keep]
    src = csc.indices[pos].long()
    dst = nodes[seg]
    eid = pos if csc.eids is None else csc.eids[pos].long()
    return src, dst, eid


def to_block(g_or_edges, dst_nodes, num_nodes=None, idtype=torch.int64, dst_sorted=False, max_in_degree=None):
    """Bipartite block: dst = `dst_nodes` (kept in order), src = dst_nodes followed by the other sources.
    `g_or_edges` is (src, dst[, eid]) in global ids."""
    src, dst = g_or_edges[0].long(), g_or_edges[1].long()
    eid = g_or_edges[2] if len(g_or_edges) > 2 else None
    dev = src.device
    dst_nodes = torch.as_tensor(dst_nodes, device=dev).long()
    if num_nodes is None:
        num_nodes = int(max(src.max().item() if src.numel() else 0, dst_nodes.max().item() if dst_nodes.numel() else 0)) + 1
    lut = torch.full((num_nodes,), -1, dtype=torch.int64, device=dev)
    lut[dst_nodes] = torch.arange(dst_nodes.shape[0], device=dev)
    if int((lut[dst] < 0).sum().item()):
        raise DGLError("to_block: every edge destination must be one of dst_nodes")
    uniq = torch.unique(src)
    extra = uniq[lut[uniq] < 0]
    lut[extra] = dst_nodes.shape[0] + torch.arange(extra.shape[0], device=dev)
    src_nodes = torch.cat([dst_nodes, extra])
    l_src, l_dst = lut[src].to(idtype).contiguous(), lut[dst].to(idtype).contiguous()
    csc = None
    if dst_sorted:  # edges arrive grouped by destination in dst_nodes order: the in-CSR is a cumsum away (no sort)
        from .sparse import CsrView
        indptr = torch.zeros(dst_nodes.shape[0] + 1, dtype=torch.int64, device=dev)
        torch.cumsum(torch.bincount(l_dst.long(), minlength=dst_nodes.shape[0]), 0, out=indptr[1:])
        csc = CsrView(dst_nodes.shape[0], src_nodes.shape[0], indptr.to(idtype), l_src, None)
        csc.dst_is_src_prefix = True
        if max_in_degree is not None and max_in_degree <= 256:
            csc._plan = None  # no row can need splitting and the block is tiny: skip the schedule (and its host syncs)
    block = DGLGraph(GraphIndex(src_nodes.shape[0], dst_nodes.shape[0], coo=(l_src, l_dst), csc=csc), is_block=True)
    block._index.dst_is_src_prefix = True
    block._index.max_in_degree_hint = max_in_degree
    block._index.ephemeral = True  # one training step: kernel forms from host-known numbers, no analysis of the row lengths
    if csc is not None:
        block._index._mark_ephemeral(csc)
    block.srcdata[NID] = src_nodes
    block.dstdata[NID] = dst_nodes
    if eid is not None:
        block.edata[EID] = eid
    return block


class MultiLayerNeighborSampler(object):
    """fanouts[i] = neighbours sampled for layer i (None / -1: all neighbours)."""

    def __init__(self, fanouts, replace=False, return_eids=False):
        if replace:
            raise DGLError("sampling with replacement is not supported")
        self.fanouts = list(fanouts)

    def sample_blocks(self, g, seed_nodes, generator=None):
        blocks = []
        seeds = torch.as_tensor(seed_nodes, device=g.device).long()
        n = g.number_of_nodes()
        for fanout in reversed(self.fanouts):
            frontier = sample_neighbors(g, seeds, fanout, generator=generator)
            # sample_neighbors returns edges grouped by seed in seed order (CSR positions sorted inside a seed)
            block = to_block(frontier, seeds, num_nodes=n, idtype=g.idtype, dst_sorted=True,
                             max_in_degree=fanout if (fanout is not None and fanout >= 0) else None)
            seeds = block.srcdata[NID]
            blocks.insert(0, block)
        return blocks


class MultiLayerFullNeighborSampler(MultiLayerNeighborSampler):
    def __init__(self, n_layers, return_eids=False):
        super(MultiLayerFullNeighborSampler, self).__init__([None] * n_layers)


class NodeDataLoader(object):
    """Iterates (input_nodes, output_nodes, blocks) over mini-batches of seed nodes."""

    def __init__(self, g, nids, block_sampler, device=None, batch_size=1, shuffle=False, drop_last=False,
                 num_workers=0, **kwargs):
        self.g, self.sampler = g, block_sampler
        self.nids = torch.as_tensor(nids).long()
        if self.nids.dtype == torch.bool:
            self.nids = torch.nonzero(self.nids).flatten()
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), shuffle, drop_last
        self.device = device  # num_workers is accepted and ignored: sampling runs where the graph lives

    def __len__(self):
        n = self.nids.shape[0]
        return n // self.batch_size if self.drop_last else math.ceil(n / self.batch_size)

    def __iter__(self):
        nids = self.nids
        if self.shuffle:
            nids = nids[torch.randperm(nids.shape[0])]
        nids = nids.to(self.g.device)
        for i in range(len(self)):
            seeds = nids[i * self.batch_size:(i + 1) * self.batch_size]
            blocks = self.sampler.sample_blocks(self.g, seeds)
            if self.device is not None:
                blocks = [b.to(self.device) for b in blocks]
            yield blocks[0].srcdata[NID], blocks[-1].dstdata[NID], blocks


def node_subgraph(g, nodes):
    """g.subgraph(nodes | mask): induced subgraph with relabelled nodes; node features are sliced."""
    dev = g.device
    if isinstance(nodes, dict):  # {ntype: ids} of a graph with one node type (dgl_cluster_sampler.py:99)
        if len(nodes) != 1:
            raise DGLError("subgraph: a homogeneous graph takes the nodes of exactly one type")
        nodes = next(iter(nodes.values()))
    nodes = torch.as_tensor(nodes, device=dev)
    if nodes.dtype == torch.bool:
        nodes = torch.nonzero(nodes).flatten()
    nodes = nodes.long()
    n = g.number_of_nodes()
    lut = torch.full((n,), -1, dtype=torch.int64, device=dev)
    lut[nodes] = torch.arange(nodes.shape[0], device=dev)
    src, dst = g.edges()
    ls, ld = lut[src.long()], lut[dst.long()]
    keep = (ls >= 0) & (ld >= 0)
    sub = DGLGraph(GraphIndex(nodes.shape[0], nodes.shape[0], coo=(ls[keep].to(g.idtype).contiguous(), ld[keep].to(g.idtype).contiguous())))
    sub._index.ephemeral = True  # a cluster / mini-batch subgraph: see GraphIndex.ephemeral
    for k, v in g.ndata.items():
        sub.ndata[k] = v[nodes.to(v.device)]
    for k, v in g.edata.items():
        sub.edata[k] = v[keep.to(v.device)]
    sub.ndata[NID] = nodes
    sub.edata[EID] = torch.nonzero(keep).flatten()
    return sub
