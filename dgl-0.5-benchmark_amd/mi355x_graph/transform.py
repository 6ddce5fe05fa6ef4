"""Integer graph transforms used by the benchmark scripts (bit-exact index work, torch integer ops):
  dgl.to_bidirected   main_dgl_arxiv_sage.py:162      dgl.add_self_loop  main_dgl_reddit_gat.py:136
  dgl.from_networkx   main_dgl_citation_sage.py:190   dgl.batch          GraphDataLoader, main_dgl_molhiv_gcn.py:163
They run on whatever device holds the index tensors (the scripts call them on the CPU before .to()).
"""
import torch

from ._lib import DGLError
from .graph import DGLGraph, GraphIndex, graph


def to_bidirected(g, copy_ndata=False, readonly=None):
    """Union of the edges and their reverses with duplicates removed; result edges sorted by
    (src, dst).  Edge features are dropped (as DGL does)."""
    if g.is_block:
        raise DGLError("to_bidirected expects a homogeneous graph")
    src, dst = g.edges()
    n = g.number_of_nodes()
    s = torch.cat([src, dst]).long()
    d = torch.cat([dst, src]).long()
    key = torch.unique(s * n + d)  # sorted
    ns = torch.div(key, n, rounding_mode="floor").to(g.idtype)
    nd = (key % n).to(g.idtype)
    out = DGLGraph(GraphIndex(n, n, coo=(ns, nd)))
    if copy_ndata:
        for k, v in g.ndata.items():
            out.ndata[k] = v
    return out


def add_reverse_edges(g, copy_ndata=True, copy_edata=False):
    src, dst = g.edges()
    n = g.number_of_nodes()
    out = DGLGraph(GraphIndex(n, n, coo=(torch.cat([src, dst]), torch.cat([dst, src]))))
    if copy_ndata:
        for k, v in g.ndata.items():
            out.ndata[k] = v
    return out


def add_self_loop(g, etype=None):
    """Appends (i, i) for every node AFTER the existing edges; existing loops are kept.
    Node features are kept; new edges get zero-filled edge features."""
    if g.is_block:
        raise DGLError("add_self_loop expects a homogeneous graph")
    src, dst = g.edges()
    n = g.number_of_nodes()
    loop = torch.arange(n, dtype=g.idtype, device=g.device)
    out = DGLGraph(GraphIndex(n, n, coo=(torch.cat([src, loop]), torch.cat([dst, loop]))))
    for k, v in g.ndata.items():
        out.ndata[k] = v
    for k, v in g.edata.items():
        pad = torch.zeros((n,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        out.edata[k] = torch.cat([v, pad])
    return out


def remove_self_loop(g, etype=None):
    src, dst = g.edges()
    keep = src != dst
    n = g.number_of_nodes()
    out = DGLGraph(GraphIndex(n, n, coo=(src[keep], dst[keep])))
    for k, v in g.ndata.items():
        out.ndata[k] = v
    for k, v in g.edata.items():
        out.edata[k] = v[keep]
    return out


def reverse(g, copy_ndata=True, copy_edata=False):
    src, dst = g.edges()
    out = DGLGraph(GraphIndex(g.number_of_dst_nodes(), g.number_of_src_nodes(), coo=(dst, src)), is_block=g.is_block)
    if copy_ndata and not g.is_block:
        for k, v in g.ndata.items():
            out.ndata[k] = v
    if copy_edata:
        for k, v in g.edata.items():
            out.edata[k] = v
    return out


def from_networkx(nx_graph, node_attrs=None, edge_attrs=None, idtype=None, device=None):
    """Nodes are relabelled 0..N-1 in sorted order when they are not already consecutive integers;
    undirected graphs yield both directions (networkx' to_directed order)."""
    import networkx as nx
    nodes = list(nx_graph.nodes())
    if not all(isinstance(v, int) for v in nodes) or sorted(nodes) != list(range(len(nodes))):
        nx_graph = nx.convert_node_labels_to_integers(nx_graph, ordering="sorted")
    if not nx_graph.is_directed():
        nx_graph = nx_graph.to_directed()
    n = nx_graph.number_of_nodes()
    edges = list(nx_graph.edges())
    src = torch.tensor([e[0] for e in edges], dtype=torch.int64)
    dst = torch.tensor([e[1] for e in edges], dtype=torch.int64)
    g = graph((src, dst), num_nodes=n, idtype=idtype, device=device)
    for attr in (node_attrs or []):
        g.ndata[attr] = torch.as_tensor([nx_graph.nodes[i][attr] for i in range(n)])
    return g


def from_scipy(sp_mat, idtype=None, device=None):
    coo = sp_mat.tocoo()
    return graph((torch.from_numpy(coo.row.astype("int64")), torch.from_numpy(coo.col.astype("int64"))),
                 num_nodes=max(sp_mat.shape), idtype=idtype, device=device)


def batch(graphs, ndata="__ALL__", edata="__ALL__"):
    """Block-diagonal union with node / edge id offsets; records batch_num_nodes / batch_num_edges.
    A handful of tensor ops per batch (not per graph): collation of 256 molecules must not cost more than the
    GPU work of the iteration it feeds."""
    if len(graphs) == 0:
        raise DGLError("The input list of graphs cannot be empty.")
    idtype, device = graphs[0].idtype, graphs[0].device
    srcs, dsts, n_nodes, n_edges, bn, be = [], [], [], [], [], []
    for g in graphs:
        if g.idtype != idtype or g.device != device:
            raise DGLError("all graphs in a batch must share idtype and device")
        s, d = g._index.coo()
        srcs.append(s)
        dsts.append(d)
        n_nodes.append(g.number_of_nodes())
        n_edges.append(int(s.shape[0]))
        if g._batch_num_nodes is None:
            bn.append(n_nodes[-1])
            be.append(n_edges[-1])
        else:  # batching already-batched graphs keeps the finest granularity
            bn.extend(g._batch_num_nodes.tolist())
            be.extend(g._batch_num_edges.tolist())
    total = sum(n_nodes)
    import numpy as np  # offsets on the host with numpy: torch's repeat_interleave fans tiny inputs out over every core
    node_off = np.concatenate([[0], np.cumsum(n_nodes[:-1], dtype=np.int64)]) if len(n_nodes) > 1 else np.zeros(1, np.int64)
    edge_off = torch.from_numpy(np.repeat(node_off, n_edges)).to(device=device, dtype=idtype)
    out = DGLGraph(GraphIndex(total, total, coo=(torch.cat(srcs) + edge_off, torch.cat(dsts) + edge_off)))
    out._index.ephemeral = True  # a batch of small graphs lives for one step: see GraphIndex.ephemeral
    out._batch_num_nodes = torch.tensor(bn, dtype=torch.int64, device=device)
    out._batch_num_edges = torch.tensor(be, dtype=torch.int64, device=device)
    for frames, target in ((lambda g: g._src_frame, out.ndata), (lambda g: g._edge_frame, out.edata)):
        for k in list(frames(graphs[0]).keys()):
            target[k] = torch.cat([frames(g)[k] for g in graphs], dim=0)
    return out


def unbatch(g):
    bn, be = g.batch_num_nodes().tolist(), g.batch_num_edges().tolist()
    src, dst = g.edges()
    out, no, eo = [], 0, 0
    for n, e in zip(bn, be):
        sub = DGLGraph(GraphIndex(n, n, coo=(src[eo:eo + e] - no, dst[eo:eo + e] - no)))
        for k, v in g.ndata.items():
            sub.ndata[k] = v[no:no + n]
        for k, v in g.edata.items():
            sub.edata[k] = v[eo:eo + e]
        out.append(sub)
        no, eo = no + n, eo + e
    return out


def metis_partition_assignment(g, k, balance_ntypes=None, balance_edges=False):
    """k-way edge-cut node partition -> [N] part ids.  The role of dgl.transform.metis_partition_assignment
    (end_to_end/sampling/link-prediction/dgl_cluster_sampler.py:24); METIS itself is not available offline, the
    partitioner is the label-propagation + greedy placement + refinement of mi355x_graph/dist.py."""
    from .dist import partition_nodes
    src, dst = g.edges()
    assign, _ = partition_nodes(src.long(), dst.long(), g.number_of_nodes(), int(k))
    return assign


def metis_partition(g, k, extra_cached_hops=0, reshuffle=False, balance_ntypes=None, balance_edges=False):
    """dict part_id -> induced subgraph carrying ndata[dgl.NID]
    (cluster-sage/dgl/partition_utils.py:9-16: `for k, val in metis_partition(g, psize).items(): val.ndata[dgl.NID]`)."""
    if extra_cached_hops:
        raise DGLError("metis_partition: extra_cached_hops is not supported")
    assign = metis_partition_assignment(g, k)
    order = torch.sort(assign, stable=True)[1]
    counts = torch.bincount(assign, minlength=int(k)).tolist()
    parts, off = {}, 0
    for i, c in enumerate(counts):
        if c:
            parts[i] = g.subgraph(order[off:off + c])
        off += c
    return parts
