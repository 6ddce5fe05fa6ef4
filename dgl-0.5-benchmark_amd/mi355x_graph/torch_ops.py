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
call into the same C ABI that the ctypes path uses -- the kernels do not change.

Round 4: the ops are DEFINED AND IMPLEMENTED IN C++ (csrc/torch_bind.cpp: TORCH_LIBRARY(mi355x_graph) + TORCH_LIBRARY_IMPL(...,
CUDA, ...) in libmi355x_graph_torch.so, a host translation unit linked against libmi355x_graph.so; the C header stays free of
torch types).  This module loads that library and adds what belongs to Python: the fake implementations and the autograd
formula of edge_softmax.  Every op takes a trailing `plan: int = 0` -- the address of the CSR's live mgx_spmm_plan
(`plan_handle(csr)`), 0 = natural row order.  Without the library (a tree where it was not built)
the same schemas are registered from Python over sparse.HipBackend, where a CSR finds its cached schedule again through a
weak registry keyed by its tensors' storage.

The autograd Functions of ops.py call these ops instead of the direct ctypes wrappers when MGX_TORCH_OPS=1 (per-call host
time of the three routes: experiments/exp_host_overhead.py, numbers in DESIGN.md).
"""
import ctypes
import os
import weakref
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import sparse
from ._lib import CSRC_DIR, DGLError

NS = "mi355x_graph"
_views = weakref.WeakValueDictionary()

EXT_PATH = os.path.join(CSRC_DIR, "libmi355x_graph_torch.so")
NATIVE = False
if os.path.exists(EXT_PATH):
    torch.ops.load_library(EXT_PATH)  # TORCH_LIBRARY(mi355x_graph): schemas + HIP implementations
    NATIVE = True


def _op(name, fn):
    """Register `fn` (type-annotated, the op's Python implementation) as mi355x_graph::<name> -- unless the C++ library already
    defines and implements it; returns an object with .register_fake / .register_autograd either way."""
    if not NATIVE:
        return torch.library.custom_op(NS + "::" + name, mutates_args=())(fn)

    class _Native(object):
        @staticmethod
        def register_fake(f):
            return torch.library.register_fake(NS + "::" + name)(f)

        @staticmethod
        def register_autograd(backward, setup_context=None):
            torch.library.register_autograd(NS + "::" + name, backward, setup_context=setup_context)

    return _Native


_plan_owners = weakref.WeakValueDictionary()  # handle -> the plan object whose c_struct() lives at that address


def _handle_of(plan):
    """A plan's handle = the address of its mgx_spmm_plan struct.  The struct (and the device tables it points to) is owned by the plan
    object, the plan by its CsrView: whoever keeps a handle across calls must keep the owner too -- `plan_owner(handle)` gives it back
    (an autograd node saves it on its ctx, see _softmax_setup), so a graph that lives for one step (a sampled block, a batch) cannot be
    collected between the forward that took the handle and the backward that uses it."""
    if plan is None:
        return 0
    handle = ctypes.addressof(plan.c_struct())
    _plan_owners[handle] = plan
    return handle


def plan_owner(handle):
    """The live plan object behind a handle handed out by plan_handle() / softmax_plan_handle() (None for 0); raises when the
    owner is gone -- a handle that outlived its graph must never reach the C++ op, which would dereference freed host memory."""
    if not handle:
        return None
    plan = _plan_owners.get(handle)
    if plan is None:
        raise DGLError("execution-plan handle %#x is stale: the graph (CsrView) that owned it has been released" % handle)
    return plan


def plan_handle(csr):
    """Address of the CSR's mgx_spmm_plan struct (kept alive by the CsrView), 0 when it has none or the ops are Python's."""
    if not NATIVE or not csr.indptr.is_cuda:
        return 0
    return _handle_of(csr.plan())


def softmax_plan_handle(csr):
    if not NATIVE or not csr.indptr.is_cuda:
        return 0
    return _handle_of(csr.softmax_plan())


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
    if not NATIVE:  # the Python registrations find the CSR's cached schedule again through the registry
        register_view(csr)
    return csr.indptr, csr.indices, csr.eids, csr.num_cols


def _feat_shape(U, E):
    ushape = tuple(U.shape[1:]) if U is not None else ()
    eshape = tuple(E.shape[1:]) if E is not None else ()
    if U is not None and E is not None:
        return tuple(torch.broadcast_shapes(ushape, eshape))
    return ushape if U is not None else eshape


# ----------------------------------------------------------------------------- gspmm
def _gspmm_py(indptr: Tensor, indices: Tensor, eids: Optional[Tensor], num_cols: int, op: str, reduce: str,
              ufeat: Optional[Tensor], efeat: Optional[Tensor], plan: int = 0, flags: int = 0) -> Tuple[Tensor, Tensor, Tensor]:
    csr = _view(indptr, indices, eids, num_cols)
    want_arg = reduce in ("max", "min")
    out, arg_u, arg_e = sparse.gspmm_raw(csr, op, reduce, ufeat, efeat, want_arg=want_arg)
    out = out.contiguous()  # the wide-row path returns a column slice of a padded result; the fake impl promises dense strides
    return out, (arg_u if arg_u is not None else indptr.new_empty(0)), (arg_e if arg_e is not None else indptr.new_empty(0))


gspmm = _op("gspmm", _gspmm_py)


@gspmm.register_fake
def _(indptr, indices, eids, num_cols, op, reduce, ufeat, efeat, plan=0, flags=0):
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


def _gsddmm_py(indptr: Tensor, indices: Tensor, eids: Optional[Tensor], num_cols: int, op: str, lhs: Optional[Tensor],
               rhs: Optional[Tensor], lhs_target: str, rhs_target: str, plan: int = 0) -> Tensor:
    """out[e] = op(lhs[t_l(e)], rhs[t_r(e)]) by edge id, walking the in-CSR (indptr over destination nodes)."""
    return sparse.gsddmm_raw(_IndexOverCsr(_view(indptr, indices, eids, num_cols)), op, lhs, rhs, lhs_target, rhs_target)


gsddmm = _op("gsddmm", _gsddmm_py)


@gsddmm.register_fake
def _(indptr, indices, eids, num_cols, op, lhs, rhs, lhs_target, rhs_target, plan=0):
    L = None if op == "copy_rhs" else lhs
    R = None if op == "copy_lhs" else rhs
    ref = L if L is not None else R
    nnz = indices.shape[0]
    if op == "dot":
        shape = tuple(torch.broadcast_shapes(tuple(L.shape[1:-1]), tuple(R.shape[1:-1]))) + (1,)
    else:
        shape = _feat_shape(L, R)
    return ref.new_empty((nnz,) + shape)


def _gsddmm_coo_py(src: Tensor, dst: Tensor, num_src: int, num_dst: int, op: str, lhs: Optional[Tensor], rhs: Optional[Tensor],
                   lhs_target: str, rhs_target: str) -> Tensor:
    """The COO walk (src / dst in edge-id order) of the same operator."""
    from .graph import GraphIndex
    return sparse.gsddmm_raw(GraphIndex(num_src, num_dst, coo=(src, dst)), op, lhs, rhs, lhs_target, rhs_target)


gsddmm_coo = _op("gsddmm_coo", _gsddmm_coo_py)


@gsddmm_coo.register_fake
def _(src, dst, num_src, num_dst, op, lhs, rhs, lhs_target, rhs_target):
    L = None if op == "copy_rhs" else lhs
    R = None if op == "copy_lhs" else rhs
    ref = L if L is not None else R
    if op == "dot":
        shape = tuple(torch.broadcast_shapes(tuple(L.shape[1:-1]), tuple(R.shape[1:-1]))) + (1,)
    else:
        shape = _feat_shape(L, R)
    return ref.new_empty((src.shape[0],) + shape)


# ----------------------------------------------------------------------------- edge softmax
def _edge_softmax_fwd_py(indptr: Tensor, indices: Tensor, eids: Optional[Tensor], num_cols: int, z: Tensor, plan: int = 0) -> Tensor:
    return sparse.edge_softmax_fwd_raw(_view(indptr, indices, eids, num_cols), z)


edge_softmax_fwd = _op("edge_softmax_fwd", _edge_softmax_fwd_py)


@edge_softmax_fwd.register_fake
def _(indptr, indices, eids, num_cols, z, plan=0):
    return torch.empty_like(z, memory_format=torch.contiguous_format)


def _edge_softmax_bwd_py(indptr: Tensor, indices: Tensor, eids: Optional[Tensor], num_cols: int, a: Tensor, da: Tensor,
                         plan: int = 0) -> Tensor:
    return sparse.edge_softmax_bwd_raw(_view(indptr, indices, eids, num_cols), a, da.contiguous())


edge_softmax_bwd = _op("edge_softmax_bwd", _edge_softmax_bwd_py)


@edge_softmax_bwd.register_fake
def _(indptr, indices, eids, num_cols, a, da, plan=0):
    return torch.empty_like(a, memory_format=torch.contiguous_format)


def _softmax_setup(ctx, inputs, output):
    indptr, indices, eids, num_cols, z, plan = inputs
    ctx.save_for_backward(indptr, indices, eids, output)
    ctx.num_cols, ctx.plan = num_cols, plan
    ctx.plan_owner = plan_owner(plan) if NATIVE else None  # the handle is an address: its owner lives as long as this node


def _softmax_backward(ctx, grad):
    indptr, indices, eids, a = ctx.saved_tensors
    return None, None, None, None, torch.ops.mi355x_graph.edge_softmax_bwd(indptr, indices, eids, ctx.num_cols, a, grad, ctx.plan), None


edge_softmax_fwd.register_autograd(_softmax_backward, setup_context=_softmax_setup)


# ----------------------------------------------------------------------------- segment reduce
def _segment_reduce_py(offsets: Tensor, x: Tensor, reduce: str) -> Tensor:
    out, _ = sparse.segment_reduce_raw(offsets, x, reduce, want_arg=False, total=int(x.shape[0]))
    return out


segment_reduce = _op("segment_reduce", _segment_reduce_py)


@segment_reduce.register_fake
def _(offsets, x, reduce):
    return x.new_empty((offsets.shape[0] - 1,) + tuple(x.shape[1:]))


# ----------------------------------------------------------------------------- formats (integer work)
def _coo_to_csr_py(row: Tensor, col: Tensor, num_rows: int, num_cols: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Stable COO -> CSR: (indptr [num_rows + 1], indices = col sorted by row, eids = original positions)."""
    v = sparse.coo_to_csr(num_rows, num_cols, row, col)
    return v.indptr, v.indices, v.eids


coo_to_csr = _op("coo_to_csr", _coo_to_csr_py)


@coo_to_csr.register_fake
def _(row, col, num_rows, num_cols):
    return row.new_empty(num_rows + 1), torch.empty_like(col), torch.empty_like(col)


def _csr_transpose_py(indptr: Tensor, indices: Tensor, eids: Optional[Tensor], num_cols: int) -> Tuple[Tensor, Tensor, Tensor]:
    src = sparse.CsrView(indptr.shape[0] - 1, num_cols, indptr, indices,
                         eids if eids is not None else torch.arange(indices.shape[0], dtype=indices.dtype, device=indices.device))
    t = sparse.csr_transpose(src)
    return t.indptr, t.indices, t.eids


csr_transpose = _op("csr_transpose", _csr_transpose_py)


@csr_transpose.register_fake
def _(indptr, indices, eids, num_cols):
    return indptr.new_empty(num_cols + 1), torch.empty_like(indices), torch.empty_like(indices)


def _in_degrees_py(indptr: Tensor) -> Tensor:
    if not indptr.is_cuda:
        raise DGLError("mi355x_graph::in_degrees runs on MI355X (HIP) tensors")
    n = indptr.shape[0] - 1
    return sparse.backend_for(indptr).degrees(sparse.CsrView(n, 0, indptr, indptr.new_empty(0), None)).clone()


in_degrees = _op("in_degrees", _in_degrees_py)


@in_degrees.register_fake
def _(indptr):
    return indptr.new_empty(indptr.shape[0] - 1)


# ----------------------------------------------------------------------------- the route ops.py takes under MGX_TORCH_OPS=1
def _native_shapes_ok(U, E):
    """What the C++ op takes without offset tables: equal feature shapes, or one weight per head ((N, H, F) x (E, H, 1))."""
    if U is None or E is None:
        return True
    us, es = tuple(U.shape[1:]), tuple(E.shape[1:])
    return us == es or (len(us) == len(es) and len(us) >= 1 and us[:-1] == es[:-1] and es[-1] == 1)


def raw_gspmm(csr, op, reduce, X, Y, want_arg=False):
    """sparse.gspmm_raw's contract through torch.ops.mi355x_graph.gspmm.  What only the Python layer knows how to route -- general
    broadcasting (offset tables), the LDS-tile kernel and the line-padded wide rows of dense graphs -- stays on the direct path."""
    if X is not None and Y is not None and not _native_shapes_ok(X, Y):
        return sparse.gspmm_raw(csr, op, reduce, X, Y, want_arg=want_arg)
    if op == "copy_lhs" and (reduce == "sum" or reduce == "mean"):
        if csr._tile_plan is False:  # decided once per CSR, as CsrView.tile_plan() does
            from . import tileplan
            csr._tile_plan = {} if tileplan.tile_plan_wanted(csr) else None
        width = X.numel() // max(int(X.shape[0]), 1)
        if csr._tile_plan is not None or (width > 256 and width % 32):
            return sparse.gspmm_raw(csr, op, reduce, X, Y, want_arg=want_arg)
    if not NATIVE:
        register_view(csr)
    flags, handle = 0, None
    if (op == "copy_lhs" or op == "copy_rhs") and (reduce == "sum" or reduce == "mean") and NATIVE and csr.indptr.is_cuda:
        ref = X if op == "copy_lhs" else Y
        plan, short = csr.spmm_plan_for(ref.numel() // max(int(ref.shape[0]), 1))
        if short:  # MGX_SPMM_SHORT_ROWS: one work item per lane group, over the schedule or its two-part form (CsrView.spmm_plan_for)
            flags, handle = 2, _handle_of(plan)
    if handle is None:
        handle = plan_handle(csr)
    try:
        if sparse.PROFILE is not None and csr.indptr.is_cuda:  # bench.py: the same record sparse.gspmm_raw's route appends
            width = max(t.numel() // max(int(t.shape[0]), 1) for t in (X, Y) if t is not None)
            with sparse.timed_call(csr.indptr.device, op=op, reduce=reduce, out_len=width, n_rows=csr.num_rows, n_cols=csr.num_cols,
                                   nnz=csr.nnz, accumulate=False, route="torch.ops", short_rows=bool(flags)):
                out, au, ae = torch.ops.mi355x_graph.gspmm(csr.indptr, csr.indices, csr.eids, csr.num_cols, op, reduce, X, Y, handle, flags)
        else:
            out, au, ae = torch.ops.mi355x_graph.gspmm(csr.indptr, csr.indices, csr.eids, csr.num_cols, op, reduce, X, Y, handle, flags)
    except DGLError:
        raise
    except RuntimeError as err:  # TORCH_CHECK in csrc/torch_bind.cpp: the operator surface raises DGLError (SURVEY 8b "Errors")
        raise DGLError(str(err).split("\n")[0]) from None
    return out, (au if au.numel() else None), (ae if ae.numel() else None)


def raw_gsddmm(gidx, op, L, R, lhs_target="u", rhs_target="v"):
    """sparse.gsddmm_raw's contract through the dispatcher ops: the COO walk when the graph keeps its edge list, else the CSR walk;
    operands that need broadcasting (offset tables) stay on the direct path."""
    if not NATIVE or (L is not None and R is not None and L.shape[1:] != R.shape[1:]) or (L if L is not None else R).dtype != torch.float32:
        return sparse.gsddmm_raw(gidx, op, L, R, lhs_target, rhs_target)
    ref = L if L is not None else R
    width = lambda x: 0 if x is None else x.numel() // max(int(x.shape[0]), 1)  # noqa: E731
    if ref.is_cuda and sparse.HipBackend._sddmm_in_csr_order(gidx, op, L, R, lhs_target, rhs_target, width(L), width(R), width(ref), None, None):
        return sparse.gsddmm_raw(gidx, op, L, R, lhs_target, rhs_target)  # the walk in the in-CSR's order (mgx_sddmm_coo_perm): direct route
    try:
        with sparse.timed_call(ref.device, kernel="sddmm", op=op, out_len=(width(ref) // int(ref.shape[-1]) if op == "dot" else width(ref)),
                               l_len=width(L), r_len=width(R), nnz=gidx.num_edges(), n_src=gidx.num_src, n_dst=gidx.num_dst,
                               targets=lhs_target + rhs_target, route="torch.ops"):
            if gidx.has_format("coo") or not gidx.has_format("csc"):  # the same choice HipBackend.sddmm makes
                src, dst = gidx.coo()
                return torch.ops.mi355x_graph.gsddmm_coo(src, dst, gidx.num_src, gidx.num_dst, op, L, R, lhs_target, rhs_target)
            csc = gidx.csc()
            return torch.ops.mi355x_graph.gsddmm(csc.indptr, csc.indices, csc.eids, csc.num_cols, op, L, R, lhs_target, rhs_target, plan_handle(csc))
    except DGLError:
        raise
    except RuntimeError as err:
        raise DGLError(str(err).split("\n")[0]) from None
