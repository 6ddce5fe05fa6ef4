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


class GraphPool(object):
    """The small graphs of a dataset in ONE storage: all edge lists (local node ids) and all feature rows concatenated, with offsets per
    graph.  `GraphPool.adopt(graphs)` builds it once and rewires every graph to VIEWS of it (same DGLGraph objects, same values; in-place
    feature writes land in the pool), so that `batch()` of any subset -- a shuffled mini-batch of 256 molecules, main_dgl_molhiv_gcn.py:163 --
    is a dozen index operations on the pooled arrays instead of four `torch.cat`s over 256 tensors and a Python loop over the graphs:
    3.6 ms -> 0.3 ms per batch on the host; the eager molhiv loop 0.91 -> 0.70 s per epoch (docs/LOG_r05.md section 9a).
    A graph whose fields were reassigned after adoption (Frame._stamp), or a list that mixes pools, takes the general path."""

    def __init__(self, idtype, node_off, edge_off, src, dst, ndata, edata):
        self.idtype, self.node_off, self.edge_off = idtype, node_off, edge_off
        self.src, self.dst, self.ndata, self.edata = src, dst, ndata, edata
        self.members = None  # weak references to the adopted graphs (members_clean)

    def members_clean(self):
        """True when no adopted graph had a field reassigned or deleted since adoption (in-place writes go to the pool and are fine)."""
        for ref in self.members or ():
            g = ref()
            if g is not None:
                p = getattr(g, "_pool", None)
                if p is None or p[2] != g._src_frame._stamp or p[3] != g._edge_frame._stamp:
                    return False
        return True

    @staticmethod
    def adopt(graphs):
        """Pool `graphs` (homogeneous, CPU, not batched, COO available, equal idtype and feature keys / dtypes / trailing shapes) and
        rewire them to views; returns the pool, or None when the list does not qualify (nothing is changed then)."""
        import numpy as np
        if len(graphs) < 2:
            return None
        g0 = graphs[0]
        nkeys, ekeys = list(g0._src_frame.keys()), list(g0._edge_frame.keys())
        for g in graphs:
            if (type(g) is not DGLGraph or g.is_block or g._batch_num_nodes is not None or g.device.type != "cpu" or g.idtype != g0.idtype
                    or not g._index.has_format("coo") or list(g._src_frame.keys()) != nkeys or list(g._edge_frame.keys()) != ekeys
                    or getattr(g, "_pool", None) is not None):
                return None
            for k in nkeys:
                if g._src_frame[k].dtype != g0._src_frame[k].dtype or g._src_frame[k].shape[1:] != g0._src_frame[k].shape[1:]:
                    return None
            for k in ekeys:
                if g._edge_frame[k].dtype != g0._edge_frame[k].dtype or g._edge_frame[k].shape[1:] != g0._edge_frame[k].shape[1:]:
                    return None
        n_nodes = np.fromiter((g.number_of_nodes() for g in graphs), np.int64, len(graphs))
        coos = [g._index.coo() for g in graphs]
        n_edges = np.fromiter((int(s.shape[0]) for s, _ in coos), np.int64, len(graphs))
        node_off = np.concatenate([[0], np.cumsum(n_nodes)])
        edge_off = np.concatenate([[0], np.cumsum(n_edges)])
        src, dst = torch.cat([s for s, _ in coos]), torch.cat([d for _, d in coos])
        ndata = {k: torch.cat([g._src_frame[k] for g in graphs], dim=0) for k in nkeys}
        edata = {k: torch.cat([g._edge_frame[k] for g in graphs], dim=0) for k in ekeys}
        pool = GraphPool(g0.idtype, node_off, edge_off, src, dst, ndata, edata)
        for i, g in enumerate(graphs):
            a, b, ea, eb = int(node_off[i]), int(node_off[i + 1]), int(edge_off[i]), int(edge_off[i + 1])
            g._index = GraphIndex(b - a, b - a, coo=(src[ea:eb], dst[ea:eb]))
            for k in nkeys:
                g._src_frame._d[k] = ndata[k][a:b]
            for k in ekeys:
                g._edge_frame._d[k] = edata[k][ea:eb]
            g._pool = (pool, i, g._src_frame._stamp, g._edge_frame._stamp)
        import weakref
        pool.members = [weakref.ref(g) for g in graphs]
        return pool

    @staticmethod
    def of(graphs):
        """(pool, ids) when every graph of the list is an unmodified member of one pool, else None."""
        import numpy as np
        first = getattr(graphs[0], "_pool", None)
        if first is None:
            return None
        pool = first[0]
        ids = np.empty(len(graphs), np.int64)
        for j, g in enumerate(graphs):
            p = getattr(g, "_pool", None)
            if p is None or p[0] is not pool or p[2] != g._src_frame._stamp or p[3] != g._edge_frame._stamp or g._batch_num_nodes is not None:
                return None
            ids[j] = p[1]
        return pool, ids

    def batch(self, ids):
        import numpy as np

        def ragged(starts, lens):
            total = int(lens.sum())
            if total == 0:
                return torch.zeros(0, dtype=torch.int64)
            excl = np.cumsum(lens) - lens
            return torch.from_numpy(np.arange(total, dtype=np.int64) + np.repeat(starts - excl, lens))

        n_nodes = self.node_off[ids + 1] - self.node_off[ids]
        n_edges = self.edge_off[ids + 1] - self.edge_off[ids]
        e_idx, n_idx = ragged(self.edge_off[ids], n_edges), ragged(self.node_off[ids], n_nodes)
        shift = torch.from_numpy(np.repeat(np.cumsum(n_nodes) - n_nodes, n_edges)).to(self.idtype)
        total = int(n_nodes.sum())
        out = DGLGraph(GraphIndex(total, total, coo=(self.src.index_select(0, e_idx) + shift, self.dst.index_select(0, e_idx) + shift)))
        out._index.ephemeral = True
        out._batch_num_nodes = torch.from_numpy(n_nodes.copy())
        out._batch_num_edges = torch.from_numpy(n_edges.copy())
        for k, v in self.ndata.items():
            out.ndata[k] = v.index_select(0, n_idx)
        for k, v in self.edata.items():
            out.edata[k] = v.index_select(0, e_idx)
        return out


def batch(graphs, ndata="__ALL__", edata="__ALL__"):
    """Block-diagonal union with node / edge id offsets; records batch_num_nodes / batch_num_edges.
    A handful of tensor ops per batch (not per graph): collation of 256 molecules must not cost more than the
    GPU work of the iteration it feeds.  Members of one GraphPool are batched from the pooled arrays."""
    if len(graphs) == 0:
        raise DGLError("The input list of graphs cannot be empty.")
    pooled = GraphPool.of(graphs)
    if pooled is not None:
        return pooled[0].batch(pooled[1])
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
