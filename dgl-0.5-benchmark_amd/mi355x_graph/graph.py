"""DGLGraph-compatible graph object over torch index tensors.

Mirrors the part of the DGLGraph surface the benchmark scripts touch:
  g.int()/.long()/.to()/.formats()          main_dgl_product_sage.py:158, main_dgl_molhiv_gcn.py:101
  g.local_var()/.local_scope()              main_dgl_product_sage.py:52, main_dgl_proteins_rgcn_for.py:47
  g.srcdata/.dstdata/.ndata/.edata          main_dgl_product_sage.py:61-63
  g.update_all()/.apply_edges()             main_dgl_product_sage.py:62, gcmc_dgl/model.py:342
  g.in_degrees(), number_of_*()             main_dgl_molhiv_gcn.py:41, kernel/dgl-new.py:15-16
Sparse formats are built lazily and cached on a shared GraphIndex: the in-CSR ("csc", rows =
destination nodes) feeds the forward g-SpMM, the out-CSR ("csr") the backward pass.
"""
from collections import namedtuple
from collections.abc import MutableMapping
from contextlib import contextmanager

import torch

from ._lib import DGLError
from . import sparse

ALL = "__ALL__"
_FORMAT_ORDER = ("coo", "csr", "csc")


class GraphIndex(object):
    """Immutable structure of a (possibly bipartite) graph; formats are materialised on demand."""

    def __init__(self, num_src, num_dst, coo=None, csr=None, csc=None, formats=_FORMAT_ORDER):
        self.num_src, self.num_dst = int(num_src), int(num_dst)
        self._coo, self._csr, self._csc = coo, csr, csc
        self._formats = tuple(f for f in _FORMAT_ORDER if f in formats)
        self._num_edges = None
        self._hidden_csc = None  # in-CSR built for SpMM on graphs restricted to formats('coo')
        self._hidden_csr = None
        self._canonical = None
        self.dst_is_src_prefix = False  # blocks: destination nodes are the first source nodes
        self.max_in_degree_hint = None
        self.ephemeral = False  # a structure used for ONE step (a sampled block): its views decide kernel forms without host reads

    # -- basic facts
    def _any(self):
        for f in (self._coo, self._csr, self._csc):
            if f is not None:
                return f
        raise DGLError("graph has no sparse format")

    def num_edges(self):
        if self._num_edges is None:
            a = self._any()
            self._num_edges = int(a[0].shape[0]) if isinstance(a, tuple) else a.nnz
        return self._num_edges

    @property
    def device(self):
        a = self._any()
        return a[0].device if isinstance(a, tuple) else a.device

    @property
    def idtype(self):
        a = self._any()
        return a[0].dtype if isinstance(a, tuple) else a.indptr.dtype

    def has_format(self, f):
        return {"coo": self._coo, "csr": self._csr, "csc": self._csc}[f] is not None

    def allowed(self, f):
        return f in self._formats

    def format_status(self):
        return {"created": [f for f in _FORMAT_ORDER if self.has_format(f)],
                "not created": [f for f in self._formats if not self.has_format(f)]}

    # -- conversions
    def _coo_from_csr(self, view, rows_are_src):
        n = view.num_rows
        counts = (view.indptr[1:] - view.indptr[:-1]).long()
        rows = torch.repeat_interleave(torch.arange(n, device=view.device, dtype=view.indptr.dtype), counts)
        cols = view.indices
        if view.eids is not None:  # back to edge-id order
            inv = torch.empty_like(view.eids, dtype=torch.long)
            inv[view.eids.long()] = torch.arange(view.nnz, device=view.device)
            rows, cols = rows[inv], cols[inv]
        return (rows, cols) if rows_are_src else (cols, rows)

    def coo(self):
        """(src, dst) in edge-id order."""
        if self._coo is None:
            if self._csc is not None:
                coo = self._coo_from_csr(self._csc, rows_are_src=False)
            else:
                coo = self._coo_from_csr(self._csr, rows_are_src=True)
            if not self.allowed("coo"):
                return coo
            self._coo = coo
        return self._coo

    def csc(self):
        """in-CSR: rows = dst, indices = src, eids -> edge id."""
        if self._csc is not None:
            return self._csc
        if self._hidden_csc is not None:
            return self._hidden_csc
        other = self._csr if self._csr is not None else self._hidden_csr
        if self._coo is None and other is not None and other.indptr.is_cuda:
            view = sparse.csr_transpose(other)  # no COO round trip: transpose the out-CSR on the device
        else:
            src, dst = self.coo()
            view = sparse.coo_to_csr(self.num_dst, self.num_src, dst, src)
        view.dst_is_src_prefix = self.dst_is_src_prefix
        self._mark_ephemeral(view)
        if self.allowed("csc"):
            self._csc = view
        else:
            self._hidden_csc = view
        return view

    def _mark_ephemeral(self, view):
        if self.ephemeral and view.short_hint is None:
            view.short_hint = view.nnz / float(max(view.num_rows, 1))  # average row length: CsrView._short_choice decides on it, sync-free

    def csr(self):
        """out-CSR: rows = src, indices = dst, eids -> edge id (the reversed graph's in-CSR)."""
        if self._csr is not None:
            return self._csr
        if self._hidden_csr is not None:
            return self._hidden_csr
        other = self._csc if self._csc is not None else self._hidden_csc
        if self._coo is None and other is not None and other.indptr.is_cuda:
            view = sparse.csr_transpose(other)
        else:
            src, dst = self.coo()
            view = sparse.coo_to_csr(self.num_src, self.num_dst, src, dst)
        self._mark_ephemeral(view)
        if self.allowed("csr"):
            self._csr = view
        else:
            self._hidden_csr = view
        return view

    def create_formats_(self):
        for f in self._formats:
            getattr(self, f)()

    def canonical(self):
        """The same graph with its edges RENUMBERED in in-CSR (destination-major) order.

        Returns (index, perm) with perm[new_edge_id] = old_edge_id (perm is None when the graph already
        is in that order).  Edge tensors that are internal to a module (GATConv's logits / attention)
        can live in this order: the in-CSR then needs no edge-id indirection, so edge_softmax, the
        weighted g-SpMM and the CSR-walk g-SDDMM stream E-sized tensors instead of gathering 4-byte
        elements at random edge ids."""
        if self._canonical is None:
            csc = self.csc()
            if csc.eids is None:
                self._canonical = (self, None)
            else:
                perm = csc.eids
                inv = torch.empty_like(perm)
                inv[perm.long()] = torch.arange(perm.shape[0], dtype=perm.dtype, device=perm.device)
                c_csc = sparse.CsrView(csc.num_rows, csc.num_cols, csc.indptr, csc.indices, None)
                c_csc._plan = csc.plan()  # same rows and edge ranges: the schedule carries over
                c_csc._row_order = csc._row_order
                c_csc.dst_is_src_prefix = csc.dst_is_src_prefix
                csr = self.csr()
                c_csr = sparse.CsrView(csr.num_rows, csr.num_cols, csr.indptr, csr.indices, inv[csr.eids.long()])
                c_csr._plan = csr.plan()
                idx = GraphIndex(self.num_src, self.num_dst, coo=None, csr=c_csr, csc=c_csc, formats=("csr", "csc"))
                self._canonical = (idx, perm)
        return self._canonical

    # -- derived graphs
    def with_formats(self, formats):
        if isinstance(formats, str):
            formats = [formats]
        for f in formats:
            if f not in _FORMAT_ORDER:
                raise DGLError("Unknown sparse format %r; expected one of %s" % (f, list(_FORMAT_ORDER)))
        formats = tuple(f for f in _FORMAT_ORDER if f in formats)
        if not formats:
            raise DGLError("formats() needs at least one sparse format")
        keep = [f for f in formats if self.has_format(f)]
        if not keep:  # materialise the first allowed format before dropping the rest
            tmp = GraphIndex(self.num_src, self.num_dst, self._coo, self._csr, self._csc)
            built = getattr(tmp, formats[0])()
            kw = {formats[0]: built}
            return GraphIndex(self.num_src, self.num_dst, formats=formats, **kw)
        return GraphIndex(self.num_src, self.num_dst,
                          coo=self._coo if "coo" in formats else None,
                          csr=self._csr if "csr" in formats else None,
                          csc=self._csc if "csc" in formats else None, formats=formats)

    def to(self, device):
        device = torch.device(device)
        if device == self.device:
            return self
        mv = lambda v: None if v is None else v.to(device)
        coo = None if self._coo is None else (self._coo[0].to(device), self._coo[1].to(device))
        g = GraphIndex(self.num_src, self.num_dst, coo, mv(self._csr), mv(self._csc), self._formats)
        g._hidden_csc, g._hidden_csr = mv(self._hidden_csc), mv(self._hidden_csr)
        g.dst_is_src_prefix = self.dst_is_src_prefix
        g.ephemeral = self.ephemeral
        return g

    def astype(self, dtype):
        if dtype == self.idtype:
            return self
        if dtype == torch.int32 and (self.num_edges() >= 2 ** 31 or max(self.num_src, self.num_dst) >= 2 ** 31):
            raise DGLError("graph too large for int32 ids")
        cv = lambda v: None if v is None else v.astype(dtype)
        coo = None if self._coo is None else (self._coo[0].to(dtype), self._coo[1].to(dtype))
        g = GraphIndex(self.num_src, self.num_dst, coo, cv(self._csr), cv(self._csc), self._formats)
        g._hidden_csc, g._hidden_csr = cv(self._hidden_csc), cv(self._hidden_csr)
        g.dst_is_src_prefix = self.dst_is_src_prefix
        g.ephemeral = self.ephemeral
        return g

    def in_degrees(self):
        if self._csc is not None or self._hidden_csc is not None or self.device.type != "cpu":
            return self.csc().degrees()
        _, dst = self.coo()
        return torch.bincount(dst.long(), minlength=self.num_dst).to(self.idtype)

    def has_zero_in_degree(self):
        """Cached (the structure is immutable): GATConv / GraphConv ask this on every forward; answering from the cache
        keeps a host synchronisation out of the training step (and lets the step be captured in a HIP graph)."""
        if getattr(self, "_zero_in_deg", None) is None:
            self._zero_in_deg = bool((self.in_degrees() == 0).any()) if self.num_dst else False
        return self._zero_in_deg

    def out_degrees(self):
        if self._csr is not None or self._hidden_csr is not None or self.device.type != "cpu":
            return self.csr().degrees()
        src, _ = self.coo()
        return torch.bincount(src.long(), minlength=self.num_src).to(self.idtype)


class Frame(MutableMapping):
    """Feature storage of one node/edge set: name -> tensor with a fixed number of rows."""

    def __init__(self, num_rows, data=None, kind="node"):
        self._n = num_rows
        self._d = dict(data) if data else {}
        self._kind = kind
        self._stamp = 0  # bumped by every assignment / deletion of a field (transform.GraphPool checks it)

    def __getitem__(self, k):
        return self._d[k]

    def __setitem__(self, k, v):
        if not isinstance(v, torch.Tensor):
            raise DGLError("Feature data must be a tensor, got %s" % type(v))
        if v.dim() == 0 or v.shape[0] != self._n:
            raise DGLError("Expect number of features to match number of %ss. Got %s and %d instead."
                           % (self._kind, v.shape[0] if v.dim() else "a scalar", self._n))
        self._d[k] = v
        self._stamp += 1

    def __delitem__(self, k):
        del self._d[k]
        self._stamp += 1

    def __iter__(self):
        return iter(self._d)

    def __len__(self):
        return len(self._d)

    def __repr__(self):
        return repr({k: "Scheme(shape=%s, dtype=%s)" % (tuple(v.shape[1:]), v.dtype) for k, v in self._d.items()})

    def clone(self):
        return Frame(self._n, self._d, self._kind)

    def to(self, device):
        return Frame(self._n, {k: v.to(device) for k, v in self._d.items()}, self._kind)


Scheme = namedtuple("Scheme", ["shape", "dtype"])


class DGLGraph(object):
    """Homogeneous graph (or bipartite block) with the DGLGraph message-passing surface."""

    def __init__(self, index, node_frame=None, edge_frame=None, dst_frame=None, is_block=False):
        self._index = index
        self._is_block = is_block
        self._src_frame = node_frame if node_frame is not None else Frame(index.num_src, kind="node")
        if is_block:
            self._dst_frame = dst_frame if dst_frame is not None else Frame(index.num_dst, kind="node")
        else:
            self._dst_frame = self._src_frame
        self._edge_frame = edge_frame if edge_frame is not None else Frame(index.num_edges(), kind="edge")
        self._batch_num_nodes = None
        self._batch_num_edges = None
        self._unibipartite = False  # relation view of a heterograph: two node sets, not a message-flow block

    # -- structure queries
    @property
    def is_block(self):
        return self._is_block and not getattr(self, "_unibipartite", False)

    @property
    def idtype(self):
        return self._index.idtype

    @property
    def device(self):
        return self._index.device

    def number_of_nodes(self, ntype=None):
        if self._is_block:
            types = getattr(self, "_ntypes", None)
            if ntype is not None and types is not None:
                if ntype not in types:
                    raise DGLError('Node type "%s" does not exist.' % ntype)
                return self._index.num_src if ntype == types[0] else self._index.num_dst
            return self._index.num_src + self._index.num_dst
        return self._index.num_src

    num_nodes = number_of_nodes

    def number_of_src_nodes(self, ntype=None):
        return self._index.num_src

    num_src_nodes = number_of_src_nodes

    def number_of_dst_nodes(self, ntype=None):
        return self._index.num_dst

    num_dst_nodes = number_of_dst_nodes

    def number_of_edges(self, etype=None):
        return self._index.num_edges()

    num_edges = number_of_edges

    def nodes(self):
        return torch.arange(self.number_of_nodes(), dtype=self.idtype, device=self.device)

    def srcnodes(self):
        return torch.arange(self._index.num_src, dtype=self.idtype, device=self.device)

    def dstnodes(self):
        return torch.arange(self._index.num_dst, dtype=self.idtype, device=self.device)

    def edges(self, form="uv", order="eid"):
        src, dst = self._index.coo()
        if order == "srcdst":
            key = src.long() * max(self._index.num_dst, 1) + dst.long()
            perm = torch.argsort(key, stable=True)
            src, dst = src[perm], dst[perm]
            eid = perm.to(self.idtype)
        else:
            eid = torch.arange(src.shape[0], dtype=self.idtype, device=self.device)
        if form == "uv":
            return src, dst
        if form == "eid":
            return eid
        if form == "all":
            return src, dst, eid
        raise DGLError("Invalid form: %r. Must be one of 'uv', 'eid', 'all'." % form)

    all_edges = edges

    def in_degrees(self, v=ALL):
        deg = self._index.in_degrees()
        return deg if isinstance(v, str) else deg[torch.as_tensor(v, device=deg.device).long()]

    def out_degrees(self, u=ALL):
        deg = self._index.out_degrees()
        return deg if isinstance(u, str) else deg[torch.as_tensor(u, device=deg.device).long()]

    # -- frames
    @property
    def ndata(self):
        if self._is_block:
            raise DGLError("ndata is ambiguous on a graph with two node sets; use srcdata/dstdata")
        return self._src_frame

    @property
    def srcdata(self):
        return self._src_frame

    @property
    def dstdata(self):
        return self._dst_frame

    @property
    def edata(self):
        return self._edge_frame

    def _clone(self, index=None, frames=None):
        g = DGLGraph.__new__(DGLGraph)
        g._index = self._index if index is None else index
        g._is_block = self._is_block
        if frames is None:
            g._src_frame = self._src_frame.clone()
            g._dst_frame = self._dst_frame.clone() if self._is_block else g._src_frame
            g._edge_frame = self._edge_frame.clone()
        else:
            g._src_frame, g._dst_frame, g._edge_frame = frames
        g._batch_num_nodes = self._batch_num_nodes
        g._batch_num_edges = self._batch_num_edges
        g._unibipartite = getattr(self, "_unibipartite", False)
        return g

    def local_var(self):
        """A graph sharing the structure whose feature assignments do not escape to the caller."""
        return self._clone()

    @contextmanager
    def local_scope(self):
        saved = (self._src_frame, self._dst_frame, self._edge_frame)
        self._src_frame = saved[0].clone()
        self._dst_frame = saved[1].clone() if self._is_block else self._src_frame
        self._edge_frame = saved[2].clone()
        try:
            yield
        finally:
            self._src_frame, self._dst_frame, self._edge_frame = saved

    # -- dtype / device / formats
    def int(self):
        return self._clone(index=self._index.astype(torch.int32))

    def long(self):
        return self._clone(index=self._index.astype(torch.int64))

    def to(self, device, **kwargs):
        device = torch.device(device)
        if device == self.device:
            return self
        src = self._src_frame.to(device)
        dst = self._dst_frame.to(device) if self._is_block else src
        g = self._clone(index=self._index.to(device), frames=(src, dst, self._edge_frame.to(device)))
        if g._batch_num_nodes is not None:
            g._batch_num_nodes = g._batch_num_nodes.to(device)
            g._batch_num_edges = g._batch_num_edges.to(device)
        return g

    def cpu(self):
        return self.to("cpu")

    def formats(self, formats=None):
        if formats is None:
            return self._index.format_status()
        return self._clone(index=self._index.with_formats(formats))

    def create_formats_(self):
        self._index.create_formats_()

    # -- batching
    @property
    def batch_size(self):
        return 1 if self._batch_num_nodes is None else int(self._batch_num_nodes.shape[0])

    def batch_num_nodes(self, ntype=None):
        if self._batch_num_nodes is None:
            return torch.tensor([self.number_of_nodes()], dtype=torch.int64, device=self.device)
        return self._batch_num_nodes

    def batch_num_edges(self, etype=None):
        if self._batch_num_edges is None:
            return torch.tensor([self.number_of_edges()], dtype=torch.int64, device=self.device)
        return self._batch_num_edges

    def set_batch_num_nodes(self, val):
        self._batch_num_nodes = torch.as_tensor(val, dtype=torch.int64, device=self.device)

    def set_batch_num_edges(self, val):
        self._batch_num_edges = torch.as_tensor(val, dtype=torch.int64, device=self.device)

    def node_attr_schemes(self, ntype=None):
        """{feature name: (shape of one row, dtype)} of the node frame (cluster_gcn_dgl.py:242 iterates its keys)."""
        return {k: Scheme(tuple(v.shape[1:]), v.dtype) for k, v in self.ndata.items()}

    def edge_attr_schemes(self, etype=None):
        return {k: Scheme(tuple(v.shape[1:]), v.dtype) for k, v in self.edata.items()}

    def in_degree(self, v):
        return int(self.in_degrees(torch.as_tensor([v]))[0])

    def out_degree(self, u):
        return int(self.out_degrees(torch.as_tensor([u]))[0])

    def find_edges(self, eid, etype=None):
        src, dst = self._index.coo()
        e = torch.as_tensor(eid, device=src.device).long().view(-1)
        if e.numel() and (int(e.max()) >= src.shape[0] or int(e.min()) < 0):
            raise DGLError("find_edges: edge id out of range")
        return src[e], dst[e]

    def has_edges_between(self, u, v):
        src, dst = self._index.coo()
        n = max(self._index.num_dst, 1)
        key = torch.sort(src.long() * n + dst.long())[0]
        q = torch.as_tensor(u, device=src.device).long() * n + torch.as_tensor(v, device=src.device).long()
        pos = torch.searchsorted(key, q).clamp(max=max(key.shape[0] - 1, 0))
        return key[pos] == q if key.numel() else torch.zeros_like(q, dtype=torch.bool)

    def subgraph(self, nodes, **kwargs):
        from .sampling import node_subgraph
        return node_subgraph(self, nodes)

    def add_self_loop(self, etype=None):
        from .transform import add_self_loop
        return add_self_loop(self)

    def remove_self_loop(self, etype=None):
        from .transform import remove_self_loop
        return remove_self_loop(self)

    def reverse(self, copy_ndata=True, copy_edata=False):
        from .transform import reverse
        return reverse(self, copy_ndata=copy_ndata, copy_edata=copy_edata)

    # -- message passing (implemented in core.py to keep this file structural)
    def update_all(self, message_func, reduce_func, apply_node_func=None, etype=None):
        from . import core
        core.update_all(self, message_func, reduce_func, apply_node_func)

    def apply_edges(self, func, edges=ALL, etype=None):
        from . import core
        if not isinstance(edges, str):
            raise DGLError("apply_edges on an edge subset is not supported by this backend")
        core.apply_edges(self, func)

    def __repr__(self):
        if self._is_block:
            return "Block(num_src_nodes=%d, num_dst_nodes=%d, num_edges=%d)" % (
                self._index.num_src, self._index.num_dst, self.number_of_edges())
        return "Graph(num_nodes=%d, num_edges=%d,\n      ndata_schemes=%r\n      edata_schemes=%r)" % (
            self.number_of_nodes(), self.number_of_edges(), self._src_frame, self._edge_frame)



def _to_index_tensor(x, idtype):
    if isinstance(x, torch.Tensor):
        if x.dtype not in (torch.int32, torch.int64):
            raise DGLError("Expect the node ID tensor to be int32 or int64, got %s" % x.dtype)
        return x.to(idtype) if idtype is not None else x
    t = torch.as_tensor(x)
    if t.numel() == 0:
        t = t.to(torch.int64)
    if t.dtype not in (torch.int32, torch.int64):
        t = t.to(torch.int64)
    return t.to(idtype) if idtype is not None else t


def graph(data, num_nodes=None, idtype=None, device=None, **kwargs):
    """dgl.graph((src, dst)[, num_nodes=]) -- kernel/utils.py:39."""
    if not (isinstance(data, (tuple, list)) and len(data) == 2):
        raise DGLError("dgl.graph expects a (src, dst) pair on this backend")
    src = _to_index_tensor(data[0], idtype)
    dst = _to_index_tensor(data[1], idtype)
    if src.dtype != dst.dtype:
        dst = dst.to(src.dtype)
    if src.shape != dst.shape or src.dim() != 1:
        raise DGLError("src and dst must be 1-D tensors of equal length, got %s and %s" % (tuple(src.shape), tuple(dst.shape)))
    if device is not None:
        src, dst = src.to(device), dst.to(device)
    if src.numel():
        need = int(torch.max(torch.max(src), torch.max(dst)).item()) + 1
        if int(torch.min(torch.min(src), torch.min(dst)).item()) < 0:
            raise DGLError("node IDs must be non-negative")
    else:
        need = 0
    if num_nodes is None:
        num_nodes = need
    elif num_nodes < need:
        raise DGLError("The num_nodes argument must be larger than the max ID in the data, but got %d and %d."
                       % (num_nodes, need - 1))
    return DGLGraph(GraphIndex(num_nodes, num_nodes, coo=(src.contiguous(), dst.contiguous())))


def create_block(data, num_src_nodes, num_dst_nodes, idtype=None, device=None):
    """Bipartite message-flow graph: same kernels with N_src != N_dst."""
    src = _to_index_tensor(data[0], idtype)
    dst = _to_index_tensor(data[1], idtype).to(src.dtype)
    if device is not None:
        src, dst = src.to(device), dst.to(device)
    return DGLGraph(GraphIndex(num_src_nodes, num_dst_nodes, coo=(src.contiguous(), dst.contiguous())), is_block=True)
