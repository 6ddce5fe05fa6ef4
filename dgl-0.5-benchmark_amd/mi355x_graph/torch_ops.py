"""torch.library registration of the hot-path primitives: namespace `mi355x_graph`.

SURVEY 8b ("What a C-ABI replacement must export") / north_star ("PyTorch-ROCm custom ops"): the seam DGL crosses at
_CAPI_DGLKernelSpMM / _CAPI_DGLKernelSDDMM (python/dgl/sparse.py::_gspmm/_gsddmm, UPSTREAM; reached from
kernel/dgl-new.py:20,39) is exposed here as dispatcher ops over plain tensors,

    torch.ops.mi355x_graph.gspmm(indptr, indices, eids?, num_cols, op, reduce, ufeat?, efeat?) -> (out, arg_u, arg_e)
    torch.ops.mi355x_graph.gsddmm(indptr, indices, eids?, num_cols, op, lhs?, rhs?, lhs_target, rhs_target) -> out
    torch.ops.mi355x_graph.edge_softmax_fwd(indptr, indices, eids?, num_cols, z) -> a
    torch.ops.mi355x_graph.edge_softmax_bwd(indptr, indices, eids?, num_cols, a, da) -> dz
    torch.ops.mi355x_graph.segment_reduce(offsets, x, reduce) -> out
    torch.ops.mi355x_graph.coo_to_csr(row, col, num_rows, num_cols) -> (indptr, indices, eids)
    torch.ops.mi355x_graph.csr_transpose(indptr, indices, eids?, num_cols) -> (indptr_t, indices_t, eids_t)
    torch.ops.mi355x_graph.in_degrees(indptr) -> deg

each with a fake (meta) implementation, so FakeTensor / torch.compile can trace programs that call them, and each a thin
call into the same C ABI (csrc/libmi355x_graph.so through sparse.HipBackend) that the ctypes path uses -- the kernels
do not change.  The CSR travels as its tensors; the execution schedule (mgx_spmm_plan) that belongs to a CSR is found
again through a weak registry keyed by the tensors' storage, so a registered graph keeps its cached plan.

The autograd Functions of ops.py call these ops instead of the direct ctypes wrappers when MGX_TORCH_OPS=1; the default
stays the direct path because a Python-registered custom op costs more host time per call than ctypes (measured by
experiments/exp_host_overhead.py, numbers in DESIGN.md) and the small-graph loops are host-bound.
"""
import weakref
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import sparse
from ._lib import DGLError

NS = "mi355x_graph"
_views = weakref.WeakValueDictionary()


def _key(indptr, indices):
    return (indptr.data_ptr(), indices.data_ptr() if indices.numel() else 0, int(indptr.shape[0]), int(indices.shape[0]))


def register_view(csr):
    """Remember a CsrView so that ops called with its tensors reuse its cached schedule / degrees."""
    _views[_key(csr.indptr, csr.indices)] = csr
    return csr


def _view(indptr, indices, eids, num_cols):
    v = _views.get(_key(indptr, indices))
    if v is None or v.num_cols != num_cols or (v.eids is None) != (eids is None):
        v = sparse.CsrView(indptr.shape[0] - 1, num_cols, indptr, indices, eids)
        _views[_key(indptr, indices)] = v
        _keep.append(v)
        if len(_keep) > 64:
            del _keep[0]
    return v


_keep = []  # a few views built on the fly stay alive so that their plans are not rebuilt on every call


def csr_args(csr):
    register_view(csr)
    return csr.indptr, csr.indices, csr.eids, csr.num_cols


def _feat_shape(U, E):
    ushape = tuple(U.shape[1:]) if U is not None else ()
    eshape = tuple(E.shape[1:]) if E is not None else ()
    if U is not None and E is not None:
        return tuple(torch.broadcast_shapes(ushape, eshape))
    return ushape if U is not None else eshape


# ----------------------------------------------------------------------------- gspmm
@torch.library.custom_op(NS + "::gspmm", mutates_args=())
def gspmm(indptr: Tensor, indices: Tensor, eids: Optional[Tensor], num_cols: int, op: str, reduce: str,
          ufeat: Optional[Tensor], efeat: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    csr = _view(indptr, indices, eids, num_cols)
    want_arg = reduce in ("max", "min")
    out, arg_u, arg_e = sparse.gspmm_raw(csr, op, reduce, ufeat, efeat, want_arg=want_arg)
    out = out.contiguous()  # the wide-row path returns a column slice of a padded result; the fake impl promises dense strides
    return out, (arg_u if arg_u is not None else indptr.new_empty(0)), (arg_e if arg_e is not None else indptr.new_empty(0))


@gspmm.register_fake
def _(indptr, indices, eids, num_cols, op, reduce, ufeat, efeat):
    U = None if op == "copy_rhs" else ufeat
    E = None if op == "copy_lhs" else efeat
    ref = U if U is not None else E
    shape = (indptr.shape[0] - 1,) + _feat_shape(U, E)
    out = ref.new_empty(shape)
    want = reduce in ("max", "min")
    arg_u = indptr.new_empty(shape if (want and op != "copy_rhs") else (0,))
    arg_e = indptr.new_empty(shape if (want and op != "copy_lhs") else (0,))
    return out, arg_u, arg_e


# ----------------------------------------------------------------------------- gsddmm
class _IndexOverCsr(object):
    """The little of GraphIndex that HipBackend.sddmm needs, over an in-CSR only (CSR-walk g-SDDMM)."""

    def __init__(self, csc):
        self._csc = csc
        self.num_src, self.num_dst = csc.num_cols, csc.num_rows

    def num_edges(self):
        return self._csc.nnz

    def has_format(self, f):
        return f == "csc"

    def csc(self):
        return self._csc


@torch.library.custom_op(NS + "::gsddmm", mutates_args=())
def gsddmm(indptr: Tensor, indices: Tensor, eids: Optional[Tensor], num_cols: int, op: str, lhs: Optional[Tensor],
           rhs: Optional[Tensor], lhs_target: str, rhs_target: str) -> Tensor:
    """out[e] = op(lhs[t_l(e)], rhs[t_r(e)]) by edge id, walking the in-CSR (indptr over destination nodes)."""
    return sparse.gsddmm_raw(_IndexOverCsr(_view(indptr, indices, eids, num_cols)), op, lhs, rhs, lhs_target, rhs_target)


@gsddmm.register_fake
def _(indptr, indices, eids, num_cols, op, lhs, rhs, lhs_target, rhs_target):
    L = None if op == "copy_rhs" else lhs
    R = None if op == "copy_lhs" else rhs
    ref = L if L is not None else R
    nnz = indices.shape[0]
    if op == "dot":
        shape = tuple(torch.broadcast_shapes(tuple(L.shape[1:-1]), tuple(R.shape[1:-1]))) + (1,)
    else:
        shape = _feat_shape(L, R)
    return ref.new_empty((nnz,) + shape)


# ----------------------------------------------------------------------------- edge softmax
@torch.library.custom_op(NS + "::edge_softmax_fwd", mutates_args=())
def edge_softmax_fwd(indptr: Tensor, indices: Tensor, eids: Optional[Tensor], num_cols: int, z: Tensor) -> Tensor:
    return sparse.edge_softmax_fwd_raw(_view(indptr, indices, eids, num_cols), z)


@edge_softmax_fwd.register_fake
def _(indptr, indices, eids, num_cols, z):
    return torch.empty_like(z)


@torch.library.custom_op(NS + "::edge_softmax_bwd", mutates_args=())
def edge_softmax_bwd(indptr: Tensor, indices: Tensor, eids: Optional[Tensor], num_cols: int, a: Tensor, da: Tensor) -> Tensor:
    return sparse.edge_softmax_bwd_raw(_view(indptr, indices, eids, num_cols), a, da.contiguous())


@edge_softmax_bwd.register_fake
def _(indptr, indices, eids, num_cols, a, da):
    return torch.empty_like(a)


def _softmax_setup(ctx, inputs, output):
    indptr, indices, eids, num_cols, z = inputs
    ctx.save_for_backward(indptr, indices, eids, output)
    ctx.num_cols = num_cols


def _softmax_backward(ctx, grad):
    indptr, indices, eids, a = ctx.saved_tensors
    return None, None, None, None, torch.ops.mi355x_graph.edge_softmax_bwd(indptr, indices, eids, ctx.num_cols, a, grad)


edge_softmax_fwd.register_autograd(_softmax_backward, setup_context=_softmax_setup)


# ----------------------------------------------------------------------------- segment reduce
@torch.library.custom_op(NS + "::segment_reduce", mutates_args=())
def segment_reduce(offsets: Tensor, x: Tensor, reduce: str) -> Tensor:
    out, _ = sparse.segment_reduce_raw(offsets, x, reduce, want_arg=False, total=int(x.shape[0]))
    return out


@segment_reduce.register_fake
def _(offsets, x, reduce):
    return x.new_empty((offsets.shape[0] - 1,) + tuple(x.shape[1:]))


# ----------------------------------------------------------------------------- formats (integer work)
@torch.library.custom_op(NS + "::coo_to_csr", mutates_args=())
def coo_to_csr(row: Tensor, col: Tensor, num_rows: int, num_cols: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Stable COO -> CSR: (indptr [num_rows + 1], indices = col sorted by row, eids = original positions)."""
    v = sparse.coo_to_csr(num_rows, num_cols, row, col)
    return v.indptr, v.indices, v.eids


@coo_to_csr.register_fake
def _(row, col, num_rows, num_cols):
    return row.new_empty(num_rows + 1), torch.empty_like(col), torch.empty_like(col)


@torch.library.custom_op(NS + "::csr_transpose", mutates_args=())
def csr_transpose(indptr: Tensor, indices: Tensor, eids: Optional[Tensor], num_cols: int) -> Tuple[Tensor, Tensor, Tensor]:
    src = sparse.CsrView(indptr.shape[0] - 1, num_cols, indptr, indices,
                         eids if eids is not None else torch.arange(indices.shape[0], dtype=indices.dtype, device=indices.device))
    t = sparse.csr_transpose(src)
    return t.indptr, t.indices, t.eids


@csr_transpose.register_fake
def _(indptr, indices, eids, num_cols):
    return indptr.new_empty(num_cols + 1), torch.empty_like(indices), torch.empty_like(indices)


@torch.library.custom_op(NS + "::in_degrees", mutates_args=())
def in_degrees(indptr: Tensor) -> Tensor:
    if not indptr.is_cuda:
        raise DGLError("mi355x_graph::in_degrees runs on MI355X (HIP) tensors")
    n = indptr.shape[0] - 1
    return sparse.backend_for(indptr).degrees(sparse.CsrView(n, 0, indptr, indptr.new_empty(0), None)).clone()


@in_degrees.register_fake
def _(indptr):
    return indptr.new_empty(indptr.shape[0] - 1)
