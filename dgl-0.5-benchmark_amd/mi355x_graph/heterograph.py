"""Heterogeneous graphs: typed node sets + one relation (its own in-CSR/CSR) per canonical edge type.

SURVEY section 8f rank 4 -- the multi-relation dispatch behind GCMC (gcmc_dgl/model.py:205 `dglnn.HeteroGraphConv`,
gcmc_dgl/data.py:245-306 `dgl.bipartite` / `dgl.hetero_from_relations`, `graph[etype]`, `graph.nodes[ntype].data`).

A relation is stored exactly like a homogeneous graph or a block: a GraphIndex with N_src x N_dst and its own execution
plan, so per-relation `update_all` / `apply_edges` run on the same g-SpMM / g-SDDMM kernels; `g[etype]` is a DGLGraph VIEW
whose source / destination frames are the node frames of the relation's endpoint types (shared, not copied).
"""
from collections import OrderedDict
from contextlib import contextmanager

import numpy as np
import torch

from ._lib import DGLError
from .graph import ALL, DGLGraph, Frame, GraphIndex, _to_index_tensor


class _DataHolder(object):
    """`g.nodes['user']` / `g.edges['rates']`: carries `.data`."""

    def __init__(self, frame):
        self.data = frame


class _TypedAccessor(object):
    """`g.nodes` / `g.edges`: indexable by type name (-> .data) and callable (-> ids)."""

    def __init__(self, g, kind):
        self._g, self._kind = g, kind

    def __getitem__(self, key):
        if self._kind == "node":
            return _DataHolder(self._g._node_frame(key))
        return _DataHolder(self._g._edge_frames[self._g.to_canonical_etype(key)])

    def __call__(self, *args, **kwargs):
        if self._kind == "node":
            ntype = args[0] if args else kwargs.get("ntype")
            return torch.arange(self._g.number_of_nodes(self._g._one_ntype(ntype)), dtype=self._g.idtype, device=self._g.device)
        return self._g.all_edges(*args, **kwargs)


class _MultiTypeData(object):
    """ndata / edata of a graph with several types: values are {type: tensor} dicts (DGL's convention)."""

    def __init__(self, frames):
        self._frames = frames  # OrderedDict type -> Frame

    def __getitem__(self, key):
        return {t: f[key] for t, f in self._frames.items() if key in f}

    def __setitem__(self, key, val):
        if not isinstance(val, dict):
            raise DGLError("The graph has several node/edge types: assign a {type: tensor} dict")
        for t, v in val.items():
            self._frames[t][key] = v

    def __contains__(self, key):
        return any(key in f for f in self._frames.values())

    def pop(self, key):
        return {t: f.pop(key) for t, f in self._frames.items() if key in f}


_CROSS = {
    "sum": lambda xs: torch.stack(xs, 0).sum(0),
    "mean": lambda xs: torch.stack(xs, 0).mean(0),
    "max": lambda xs: torch.stack(xs, 0).max(0)[0],
    "min": lambda xs: torch.stack(xs, 0).min(0)[0],
    "stack": lambda xs: torch.stack(xs, 1),
}


class DGLHeteroGraph(object):
    def __init__(self, num_nodes, relations, node_frames=None, edge_frames=None):
        """num_nodes: OrderedDict ntype -> count; relations: OrderedDict (stype, etype, dtype) -> GraphIndex."""
        self._num_nodes = OrderedDict(num_nodes)
        self._rels = OrderedDict(relations)
        for (s, _, d), idx in self._rels.items():
            if idx.num_src != self._num_nodes[s] or idx.num_dst != self._num_nodes[d]:
                raise DGLError("relation (%s -> %s) has %d x %d nodes, the node sets have %d x %d"
                               % (s, d, idx.num_src, idx.num_dst, self._num_nodes[s], self._num_nodes[d]))
        self._node_frames = node_frames or OrderedDict((t, Frame(n, kind="node")) for t, n in self._num_nodes.items())
        self._edge_frames = edge_frames or OrderedDict((c, Frame(i.num_edges(), kind="edge")) for c, i in self._rels.items())

    # ------------------------------------------------------------------ types
    @property
    def ntypes(self):
        return list(self._num_nodes)

    @property
    def etypes(self):
        return [c[1] for c in self._rels]

    @property
    def canonical_etypes(self):
        return list(self._rels)

    @property
    def srctypes(self):
        return [t for t in self._num_nodes if any(c[0] == t for c in self._rels)] or self.ntypes

    @property
    def dsttypes(self):
        return [t for t in self._num_nodes if any(c[2] == t for c in self._rels)] or self.ntypes

    @property
    def is_block(self):
        return False

    @property
    def is_homogeneous(self):
        return len(self._num_nodes) == 1 and len(self._rels) == 1

    def to_canonical_etype(self, etype):
        if etype is None:
            if len(self._rels) != 1:
                raise DGLError("Edge type name must be specified if there are more than one edge types.")
            return next(iter(self._rels))
        if isinstance(etype, tuple):
            if etype not in self._rels:
                raise DGLError('Edge type "%s" does not exist.' % (etype,))
            return etype
        hits = [c for c in self._rels if c[1] == etype]
        if not hits:
            raise DGLError('Edge type "%s" does not exist.' % etype)
        if len(hits) > 1:
            raise DGLError('Edge type "%s" is ambiguous. Please use canonical edge type in the form of (srctype, etype, dsttype)' % etype)
        return hits[0]

    def _one_ntype(self, ntype):
        if ntype is None:
            if len(self._num_nodes) != 1:
                raise DGLError("Node type name must be specified if there are more than one node types.")
            return next(iter(self._num_nodes))
        if ntype not in self._num_nodes:
            raise DGLError('Node type "%s" does not exist.' % ntype)
        return ntype

    def _node_frame(self, ntype):
        return self._node_frames[self._one_ntype(ntype)]

    # ------------------------------------------------------------------ structure queries
    @property
    def idtype(self):
        return next(iter(self._rels.values())).idtype if self._rels else torch.int64

    @property
    def device(self):
        return next(iter(self._rels.values())).device if self._rels else torch.device("cpu")

    def number_of_nodes(self, ntype=None):
        if ntype is None:
            return sum(self._num_nodes.values())
        return self._num_nodes[self._one_ntype(ntype)]

    num_nodes = number_of_nodes

    def number_of_src_nodes(self, ntype=None):
        return self.number_of_nodes(ntype)

    def number_of_dst_nodes(self, ntype=None):
        return self.number_of_nodes(ntype)

    def number_of_edges(self, etype=None):
        if etype is None:
            return sum(i.num_edges() for i in self._rels.values())
        return self._rels[self.to_canonical_etype(etype)].num_edges()

    num_edges = number_of_edges

    def all_edges(self, form="uv", order="eid", etype=None):
        return self[self.to_canonical_etype(etype)].edges(form, order)

    def in_degrees(self, v=ALL, etype=None):
        return self[self.to_canonical_etype(etype)].in_degrees(v)

    def out_degrees(self, u=ALL, etype=None):
        return self[self.to_canonical_etype(etype)].out_degrees(u)

    def find_edges(self, eid, etype=None):
        return self[self.to_canonical_etype(etype)].find_edges(eid)

    def has_edges_between(self, u, v, etype=None):
        return self[self.to_canonical_etype(etype)].has_edges_between(u, v)

    # ------------------------------------------------------------------ relation views
    def __getitem__(self, key):
        """g[etype] / g[stype, etype, dtype]: the relation as a graph sharing this graph's feature storage."""
        cet = self.to_canonical_etype(key)
        s, _, d = cet
        idx = self._rels[cet]
        if s == d:
            return DGLGraph(idx, node_frame=self._node_frames[s], edge_frame=self._edge_frames[cet])
        g = DGLGraph(idx, node_frame=self._node_frames[s], edge_frame=self._edge_frames[cet], dst_frame=self._node_frames[d],
                     is_block=True)
        g._unibipartite = True  # two node sets, but not a message-flow block (dst is NOT a prefix of src)
        g._ntypes = (s, d)
        return g

    def edge_type_subgraph(self, etypes):
        cets = [self.to_canonical_etype(e) for e in etypes]
        keep = [t for t in self._num_nodes if any(t in (c[0], c[2]) for c in cets)]
        return DGLHeteroGraph(OrderedDict((t, self._num_nodes[t]) for t in keep), OrderedDict((c, self._rels[c]) for c in cets),
                              OrderedDict((t, self._node_frames[t]) for t in keep),
                              OrderedDict((c, self._edge_frames[c]) for c in cets))

    # ------------------------------------------------------------------ feature storage
    @property
    def nodes(self):
        return _TypedAccessor(self, "node")

    @property
    def edges(self):
        return _TypedAccessor(self, "edge")

    @property
    def ndata(self):
        if len(self._node_frames) == 1:
            return next(iter(self._node_frames.values()))
        return _MultiTypeData(self._node_frames)

    @property
    def edata(self):
        if len(self._edge_frames) == 1:
            return next(iter(self._edge_frames.values()))
        return _MultiTypeData(self._edge_frames)

    @property
    def srcdata(self):
        types = self.srctypes
        if len(types) == 1:
            return self._node_frames[types[0]]
        return _MultiTypeData(OrderedDict((t, self._node_frames[t]) for t in types))

    @property
    def dstdata(self):
        types = self.dsttypes
        if len(types) == 1:
            return self._node_frames[types[0]]
        return _MultiTypeData(OrderedDict((t, self._node_frames[t]) for t in types))

    def _with(self, rels=None, node_frames=None, edge_frames=None):
        return DGLHeteroGraph(self._num_nodes, self._rels if rels is None else rels,
                              node_frames if node_frames is not None else OrderedDict((t, f.clone()) for t, f in self._node_frames.items()),
                              edge_frames if edge_frames is not None else OrderedDict((c, f.clone()) for c, f in self._edge_frames.items()))

    def local_var(self):
        return self._with()

    @contextmanager
    def local_scope(self):
        saved = (self._node_frames, self._edge_frames)
        self._node_frames = OrderedDict((t, f.clone()) for t, f in saved[0].items())
        self._edge_frames = OrderedDict((c, f.clone()) for c, f in saved[1].items())
        try:
            yield
        finally:
            self._node_frames, self._edge_frames = saved

    # ------------------------------------------------------------------ dtype / device / formats
    def int(self):
        return self._with(rels=OrderedDict((c, i.astype(torch.int32)) for c, i in self._rels.items()))

    def long(self):
        return self._with(rels=OrderedDict((c, i.astype(torch.int64)) for c, i in self._rels.items()))

    def to(self, device, **kwargs):
        device = torch.device(device)
        if device == self.device:
            return self
        return self._with(rels=OrderedDict((c, i.to(device)) for c, i in self._rels.items()),
                          node_frames=OrderedDict((t, f.to(device)) for t, f in self._node_frames.items()),
                          edge_frames=OrderedDict((c, f.to(device)) for c, f in self._edge_frames.items()))

    def cpu(self):
        return self.to("cpu")

    def formats(self, formats=None):
        if formats is None:
            return next(iter(self._rels.values())).format_status()
        return self._with(rels=OrderedDict((c, i.with_formats(formats)) for c, i in self._rels.items()))

    def create_formats_(self):
        for i in self._rels.values():
            i.create_formats_()

    # ------------------------------------------------------------------ message passing
    def update_all(self, message_func, reduce_func, apply_node_func=None, etype=None):
        from . import core
        core.update_all(self[self.to_canonical_etype(etype)], message_func, reduce_func, apply_node_func)

    def apply_edges(self, func, edges=ALL, etype=None):
        from . import core
        if not isinstance(edges, str):
            raise DGLError("apply_edges on an edge subset is not supported by this backend")
        core.apply_edges(self[self.to_canonical_etype(etype)], func)

    def multi_update_all(self, etype_dict, cross_reducer, apply_node_func=None):
        """Per-relation update_all, then `cross_reducer` ('sum' | 'mean' | 'max' | 'min' | 'stack') over the relations that
        write the same field of the same destination type."""
        from . import core
        if cross_reducer not in _CROSS:
            raise DGLError("Invalid cross type reducer. Must be one of 'sum', 'min', 'max', 'mean' or 'stack'.")
        collected = OrderedDict()  # (dtype, field) -> [tensor]
        for etype, funcs in etype_dict.items():
            cet = self.to_canonical_etype(etype)
            mfunc, rfunc = funcs[0], funcs[1]
            afunc = funcs[2] if len(funcs) > 2 else None
            view = self[cet]
            if view.number_of_edges() == 0:
                continue
            # results land in a scratch destination frame so that relations do not overwrite each other
            scratch = view._dst_frame.clone()
            if view._dst_frame is view._src_frame:
                view._src_frame = scratch
            view._dst_frame = scratch
            core.update_all(view, mfunc, rfunc, afunc)
            collected.setdefault((cet[2], rfunc.out_field), []).append(scratch[rfunc.out_field])
        for (dtype, field), parts in collected.items():
            self._node_frames[dtype][field] = _CROSS[cross_reducer](parts)
        if apply_node_func is not None:
            for dtype in set(d for d, _ in collected):
                ret = apply_node_func(core.NodeBatch(self, self._node_frames[dtype]))
                for k, v in ret.items():
                    self._node_frames[dtype][k] = v

    def __repr__(self):
        return "Graph(num_nodes=%r,\n      num_edges=%r,\n      metagraph=%r)" % (
            dict(self._num_nodes), {c: i.num_edges() for c, i in self._rels.items()}, [(c[0], c[2], c[1]) for c in self._rels])


# ---------------------------------------------------------------------- constructors
def _pair_from(data, idtype):
    """(src, dst) index tensors from a pair of arrays / tensors / lists or a scipy sparse matrix."""
    if hasattr(data, "tocoo"):
        coo = data.tocoo()
        return _to_index_tensor(np.asarray(coo.row, dtype=np.int64), idtype), _to_index_tensor(np.asarray(coo.col, dtype=np.int64), idtype), coo.shape
    if not (isinstance(data, (tuple, list)) and len(data) == 2):
        raise DGLError("expected a (src, dst) pair or a scipy sparse matrix")
    src = _to_index_tensor(data[0], idtype)
    dst = _to_index_tensor(data[1], idtype).to(src.dtype)
    if src.shape != dst.shape or src.dim() != 1:
        raise DGLError("src and dst must be 1-D of equal length, got %s and %s" % (tuple(src.shape), tuple(dst.shape)))
    return src.contiguous(), dst.contiguous(), None


def _max_id(t):
    return int(t.max().item()) + 1 if t.numel() else 0


def heterograph(data_dict, num_nodes_dict=None, idtype=None, device=None):
    """dgl.heterograph({(stype, etype, dtype): (src, dst)}, num_nodes_dict)."""
    pairs = OrderedDict()
    need = {}
    for cet in sorted(data_dict, key=lambda c: (c[1], c[0], c[2])):
        if not (isinstance(cet, tuple) and len(cet) == 3):
            raise DGLError("heterograph keys must be (srctype, etype, dsttype) triples, got %r" % (cet,))
        src, dst, shape = _pair_from(data_dict[cet], idtype)
        if device is not None:
            src, dst = src.to(device), dst.to(device)
        if src.numel() and (int(src.min()) < 0 or int(dst.min()) < 0):
            raise DGLError("node IDs must be non-negative")
        pairs[cet] = (src, dst)
        need[cet[0]] = max(need.get(cet[0], 0), _max_id(src), shape[0] if shape else 0)
        need[cet[2]] = max(need.get(cet[2], 0), _max_id(dst), shape[1] if shape else 0)
    counts = OrderedDict()
    for t in sorted(need):
        n = need[t] if num_nodes_dict is None or t not in num_nodes_dict else int(num_nodes_dict[t])
        if n < need[t]:
            raise DGLError("The given number of nodes of node type %s must be larger than the max ID in the data, but got %d and %d."
                           % (t, n, need[t] - 1))
        counts[t] = n
    for t in (num_nodes_dict or {}):
        counts.setdefault(t, int(num_nodes_dict[t]))
    rels = OrderedDict((c, GraphIndex(counts[c[0]], counts[c[2]], coo=p)) for c, p in pairs.items())
    return DGLHeteroGraph(counts, rels)


def bipartite(data, utype="_U", etype="_E", vtype="_V", num_nodes=None, card=None, validate=True, restrict_format="any",
              idtype=None, device=None, **kwargs):
    """dgl.bipartite (DGL <= 0.5 API, gcmc_dgl/data.py:257,306): one relation utype -etype-> vtype."""
    if utype == vtype:
        raise DGLError("utype should not be equal to vtype. Use ``dgl.graph`` instead.")
    num_nodes = num_nodes if num_nodes is not None else card
    nd = None if num_nodes is None else {utype: int(num_nodes[0]), vtype: int(num_nodes[1])}
    return heterograph({(utype, etype, vtype): data}, nd, idtype=idtype, device=device)


def hetero_from_relations(rel_graphs, num_nodes_per_type=None):
    """dgl.hetero_from_relations (DGL <= 0.5 API, gcmc_dgl/data.py:263): merge single-relation graphs."""
    counts, rels, eframes = OrderedDict(), OrderedDict(), OrderedDict()
    for rg in rel_graphs:
        if not isinstance(rg, DGLHeteroGraph) or len(rg.canonical_etypes) != 1:
            raise DGLError("hetero_from_relations expects graphs with exactly one relation each")
        cet = rg.canonical_etypes[0]
        if cet in rels:
            raise DGLError("relation %r appears twice" % (cet,))
        for t in (cet[0], cet[2]):
            n = rg.number_of_nodes(t)
            if counts.setdefault(t, n) != n:
                raise DGLError("node type %s has %d nodes in one relation and %d in another" % (t, counts[t], n))
        rels[cet] = rg._rels[cet]
        eframes[cet] = rg._edge_frames[cet]
    counts = OrderedDict((t, counts[t]) for t in sorted(counts))
    if num_nodes_per_type is not None:
        for t, n in zip(counts, num_nodes_per_type):
            counts[t] = int(n)
    return DGLHeteroGraph(counts, rels, None, eframes)
