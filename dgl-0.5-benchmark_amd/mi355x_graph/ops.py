"""Autograd-aware g-SpMM / g-SDDMM / edge_softmax / segment_reduce -- the dgl.ops surface.

Signatures follow the reference call sites:
  dgl.ops.gspmm(g, op, reduce_op, lhs_data, rhs_data)            kernel/dgl-new.py:20
  dgl.ops.gsddmm(g, op, lhs_data, rhs_data, lhs_target, rhs_target)   kernel/dgl-new.py:39
  edge_softmax(graph, logits, eids=ALL, norm_by='dst')           via GATConv, main_dgl_reddit_gat.py:10
Backward formulas are DGL's: dU of a sum-SpMM is the same SpMM on the reversed
graph (out-CSR), dE is an SDDMM, etc.  `mean` is fused: the forward kernel divides by
max(in_degree, 1) and the backward SpMM scales the gathered rows by the same factor, instead of
DGL's separate elementwise divide.
"""
import os

import torch
from torch.autograd.function import once_differentiable

from ._lib import DGLError
from . import sparse
from . import config
from .graph import DGLGraph, GraphIndex

__all__ = ["gspmm", "gsddmm", "edge_softmax", "gat_attention", "gat_fused", "segment_reduce", "copy_u_sum", "copy_u_mean", "u_mul_e_sum",
           "copy_e_sum", "u_add_v", "u_dot_v"]


def _torch_ops():
    """MGX_TORCH_OPS=1 (tests): EVERY raw primitive goes through torch.ops.mi355x_graph.* and the fused layer forms of this module
    are switched off, so that a model exercises exactly the registered operator surface."""
    if os.environ.get("MGX_TORCH_OPS", "auto") != "1":
        return None
    from . import torch_ops
    return torch_ops


_NATIVE = None


def _native_ops():
    """Default route of the generic operators behind update_all() / apply_edges() on HIP tensors: the C++ dispatcher ops
    (csrc/torch_bind.cpp, TORCH_LIBRARY(mi355x_graph)) -- 5.6 us of host time per call against 13.2 us through ctypes and 22.8 us
    for a Python-registered op (experiments/exp_host_overhead.py; eager molhiv epoch 0.92 -> 0.80 s).  MGX_TORCH_OPS=0 restores
    the ctypes route; a tree without libmi355x_graph_torch.so uses it as well."""
    global _NATIVE
    if os.environ.get("MGX_TORCH_OPS", "auto") == "0":
        return None
    if _NATIVE is None:
        from . import torch_ops
        _NATIVE = torch_ops if torch_ops.NATIVE else False
    return _NATIVE or _torch_ops()


def _gspmm_over_slots(csr, op, reduce_op, X):
    """copy_u / sum|mean of a [N, 64] float32 operand over a big CSR as 128-byte slots (csrc/spmm_slots.inc) when X MAY be mostly zero:
    a relu + dropout output that reaches update_all() through plain torch modules (the reference's own model code,
    main_dgl_product_sage.py:61-64, 93-96) carries no tag, so X is packed and the overflow count decides -- this CSR's gate stops the
    probing for 64 calls whenever more than a tenth of the rows turn out to hold more than 24 non-zeros.  Exact either way.  None: not
    applicable (the caller takes the dense kernels)."""
    if (not config.PACKED_GATHER or not config.PACKED_GATHER_PROBE or op != "copy_lhs" or reduce_op not in ("sum", "mean") or X is None
            or X.dim() != 2 or X.shape[1] != 64 or not X.is_cuda or X.dtype != torch.float32 or csr.nnz < config.PACKED_GATHER_MIN_NNZ
            or X.shape[0] != csr.num_cols or X.device.type not in sparse._BACKENDS):
        return None
    be = sparse.backend_for(X)
    if not hasattr(be, "rows_slots_pack") or capture_path():
        return None
    plan, short = csr.spmm_plan_for(64)
    if short and (plan is None or plan.rest is None):
        return None
    if csr._gate is None:
        csr._gate = _SlotGate()
    gate = csr._gate
    gate.calls += 1
    if gate.calls <= gate.dense_until:
        return None
    if not X.is_contiguous():
        X = X.contiguous()
    if not be.rows_slots_supported(X, csr):
        return None
    # the decision belongs to THIS operand, so its count is read back here (one host read per probed call): which kernels a call takes
    # -- and with them the order of additions, i.e. the bits -- depends on the operands and the sequence of calls only, never on timing
    slots, overflow = be.rows_slots_pack(X)
    gate.last_fraction = float(int(overflow)) / max(int(X.shape[0]), 1)
    if gate.last_fraction > config.PACKED_GATHER_MAX_OVERFLOW:
        gate.dense_until = gate.calls + 64
        return None
    out = torch.empty((csr.num_rows, 64), dtype=torch.float32, device=X.device)
    return be.spmm_copy_u_strided(csr, reduce_op, X, out, slots=slots)


def _raw_gspmm(csr, op, reduce_op, X, Y, want_arg=False, probe=True):
    if probe and not want_arg and Y is None:
        out = _gspmm_over_slots(csr, op, reduce_op, X)
        if out is not None:
            return out, None, None
    t = _native_ops()
    if t is None or not (X if X is not None else Y).is_cuda:
        return sparse.gspmm_raw(csr, op, reduce_op, X, Y, want_arg=want_arg)
    return t.raw_gspmm(csr, op, reduce_op, X, Y, want_arg)


def _raw_gsddmm(gidx, op, X, Y, lhs_target="u", rhs_target="v"):
    t = _native_ops()
    ref = X if X is not None else Y
    if t is None or ref is None or not ref.is_cuda:
        return sparse.gsddmm_raw(gidx, op, X, Y, lhs_target, rhs_target)
    return t.raw_gsddmm(gidx, op, X, Y, lhs_target, rhs_target)


def _gidx(g):
    if isinstance(g, DGLGraph):
        return g._index
    if isinstance(g, GraphIndex):
        return g
    raise DGLError("expected a DGLGraph, got %s" % type(g))


def _reduce_grad(grad, shape):
    """Sum `grad` over the dims that were broadcast so that it has `shape` (rows excluded)."""
    grad_shape = tuple(grad.shape[1:])
    in_shape = tuple(shape[1:])
    if in_shape == grad_shape:
        return grad
    num_to_squeeze = len(grad_shape) - len(in_shape)
    in_shape = (1,) * num_to_squeeze + in_shape
    reduce_idx = [i + 1 for i, (a, b) in enumerate(zip(grad_shape, in_shape)) if a != b]
    if reduce_idx:
        grad = grad.sum(dim=tuple(reduce_idx), keepdim=True)
    return grad.view((-1,) + tuple(shape[1:]))


def _need_reduce_last_dim(ushp, eshp):
    """(N,H,F) x (E,H,1): the edge gradient is a per-head dot product."""
    return ushp[1:-1] == eshp[1:-1] and eshp[-1] == 1 and ushp[-1] > 1


def _expand(x, shape):
    return x.expand(-1, *shape)


class GSpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gidx, op, reduce_op, X, Y):
        csc = gidx.csc()
        want_arg = reduce_op in ("max", "min")
        out, argX, argY = _raw_gspmm(csc, op, reduce_op, X, Y, want_arg=want_arg)
        ctx.backward_cache = gidx, op, reduce_op
        ctx.x_shape = None if X is None else X.shape
        ctx.y_shape = None if Y is None else Y.shape
        req_x = X is not None and X.requires_grad
        req_y = Y is not None and Y.requires_grad
        save_x = X if (req_y and op == "mul") else None
        save_y = Y if (req_x and op == "mul") or (want_arg and op == "mul") else None
        if want_arg and op == "mul":
            save_x = X
        ctx.save_for_backward(save_x, save_y, argX, argY)
        return out

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dZ):
        gidx, op, reduce_op = ctx.backward_cache
        X, Y, argX, argY = ctx.saved_tensors
        dZ = dZ.contiguous()
        dX = dY = None
        summing = reduce_op in ("sum", "mean")
        if summing:
            # mean: d(sum/deg) -> scale dZ rows by 1/max(deg,1) once (one streaming pass) instead of a
            # per-edge 4-byte gather of the factor inside the reversed SpMM (measured +20 % there)
            dZs = dZ
            if reduce_op == "mean":
                inv = gidx.csc().inv_degrees()
                dZs = dZ * inv.view((-1,) + (1,) * (dZ.dim() - 1))
            if op != "copy_rhs" and ctx.needs_input_grad[3]:
                rev = gidx.csr()  # rows = src: the reversed graph's in-CSR
                if op == "mul":
                    dX, _, _ = _raw_gspmm(rev, "mul", "sum", dZs, Y)
                elif config.SPARSE_GRAD and _torch_ops() is None:
                    dX = sparse.gspmm_grad_raw(rev, dZs)  # measured variant: flags the gradient's all-zero rows and skips them
                else:  # add, copy_lhs: aggregation of the gradient over the reversed graph (a gradient is dense: not probed for slots)
                    dX = _raw_gspmm(rev, "copy_lhs", "sum", dZs, None, probe=False)[0]
                dX = _reduce_grad(dX, ctx.x_shape)
            if op != "copy_lhs" and ctx.needs_input_grad[4]:
                if op == "mul":
                    if _need_reduce_last_dim(ctx.x_shape, ctx.y_shape):
                        dY = _raw_gsddmm(gidx, "dot", X, dZs, "u", "v")
                    else:
                        dY = _raw_gsddmm(gidx, "mul", X, dZs, "u", "v")
                else:  # add, copy_rhs
                    dY = _raw_gsddmm(gidx, "copy_rhs", None, dZs, "u", "v")
                dY = _reduce_grad(dY, ctx.y_shape)
        else:  # max / min: route dZ through the arg indices; empty rows (arg = -1) contribute nothing
            if op != "copy_rhs" and ctx.needs_input_grad[3]:
                valid = (argX >= 0)
                idx = argX.clamp(min=0).long()
                g = dZ
                if op == "mul":
                    yexp = _expand(Y, dZ.shape[1:]) if Y.shape[1:] != dZ.shape[1:] else Y
                    g = dZ * yexp.gather(0, argY.clamp(min=0).long())
                g = g * valid
                full = torch.zeros((ctx.x_shape[0],) + tuple(dZ.shape[1:]), dtype=dZ.dtype, device=dZ.device)
                full.scatter_add_(0, idx, g)
                dX = _reduce_grad(full, ctx.x_shape)
            if op != "copy_lhs" and ctx.needs_input_grad[4]:
                valid = (argY >= 0)
                idx = argY.clamp(min=0).long()
                g = dZ
                if op == "mul":
                    xexp = _expand(X, dZ.shape[1:]) if X.shape[1:] != dZ.shape[1:] else X
                    g = dZ * xexp.gather(0, argX.clamp(min=0).long())
                g = g * valid
                full = torch.zeros((ctx.y_shape[0],) + tuple(dZ.shape[1:]), dtype=dZ.dtype, device=dZ.device)
                full.scatter_add_(0, idx, g)
                dY = _reduce_grad(full, ctx.y_shape)
        return None, None, None, dX, dY


class GSDDMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gidx, op, X, Y, lhs_target, rhs_target):
        out = _raw_gsddmm(gidx, op, X, Y, lhs_target, rhs_target)
        ctx.backward_cache = gidx, op, lhs_target, rhs_target
        ctx.x_shape = None if X is None else X.shape
        ctx.y_shape = None if Y is None else Y.shape
        req_x = X is not None and X.requires_grad
        req_y = Y is not None and Y.requires_grad
        needs_other = op in ("mul", "dot", "div")
        ctx.save_for_backward(X if (needs_other and (req_y or op == "div")) else None,
                              Y if (needs_other and (req_x or req_y)) else None)
        return out

    @staticmethod
    def _to_target(gidx, target, edge_grad):
        """Sum per-edge gradients back onto their target set."""
        if target == "e":
            return edge_grad
        view = gidx.csr() if target == "u" else gidx.csc()
        out, _, _ = _raw_gspmm(view, "copy_rhs", "sum", None, edge_grad)
        return out

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dZ):
        gidx, op, lt, rt = ctx.backward_cache
        X, Y = ctx.saved_tensors
        dZ = dZ.contiguous()
        dX = dY = None
        if op != "copy_rhs" and ctx.needs_input_grad[2]:
            if op in ("add", "sub", "copy_lhs"):
                dX = GSDDMM._to_target(gidx, lt, dZ)
            elif op in ("mul", "dot", "div"):
                if lt == "u" and rt == "v" and op in ("mul", "dot"):
                    # dX[u] = sum_{e: u->v} Y[v] * dZ[e]: a mul-SpMM on the reversed graph
                    dX, _, _ = _raw_gspmm(gidx.csr(), "mul", "sum", Y, dZ)
                elif lt == "v" and rt == "u" and op in ("mul", "dot"):
                    dX, _, _ = _raw_gspmm(gidx.csc(), "mul", "sum", Y, dZ)
                else:
                    y_e = _raw_gsddmm(gidx, "copy_rhs", None, Y, lt, rt) if rt != "e" else Y
                    ge = dZ * y_e if op in ("mul", "dot") else dZ / y_e
                    dX = GSDDMM._to_target(gidx, lt, ge.contiguous())
            dX = _reduce_grad(dX, ctx.x_shape)
        if op != "copy_lhs" and ctx.needs_input_grad[3]:
            if op in ("add", "copy_rhs"):
                dY = GSDDMM._to_target(gidx, rt, dZ)
            elif op == "sub":
                dY = GSDDMM._to_target(gidx, rt, (-dZ).contiguous())
            elif op in ("mul", "dot"):
                if lt == "u" and rt == "v":
                    dY, _, _ = sparse.gspmm_raw(gidx.csc(), "mul", "sum", X, dZ)
                elif lt == "v" and rt == "u":
                    dY, _, _ = sparse.gspmm_raw(gidx.csr(), "mul", "sum", X, dZ)
                else:
                    x_e = _raw_gsddmm(gidx, "copy_lhs", X, None, lt, rt) if lt != "e" else X
                    dY = GSDDMM._to_target(gidx, rt, (dZ * x_e).contiguous())
            else:  # div: d(x/y)/dy = -x / y^2
                x_e = _raw_gsddmm(gidx, "copy_lhs", X, None, lt, rt) if lt != "e" else X
                y_e = _raw_gsddmm(gidx, "copy_rhs", None, Y, lt, rt) if rt != "e" else Y
                dY = GSDDMM._to_target(gidx, rt, (-dZ * x_e / (y_e * y_e)).contiguous())
            dY = _reduce_grad(dY, ctx.y_shape)
        return None, None, dX, dY, None, None


class EdgeSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gidx, score, norm_by):
        view = gidx.csc() if norm_by == "dst" else gidx.csr()
        t = _native_ops()
        if t is not None and score.is_cuda:
            if score.dtype != torch.float32 or score.shape[0] != view.nnz:
                raise DGLError("edge_softmax: expected float32 logits with %d rows, got %s %s" % (view.nnz, score.dtype, tuple(score.shape)))
            with torch.no_grad():
                out = torch.ops.mi355x_graph.edge_softmax_fwd(*t.csr_args(view), score.contiguous(), t.softmax_plan_handle(view))
        else:
            out = sparse.edge_softmax_fwd_raw(view, score)
        ctx.backward_cache = view
        ctx.save_for_backward(out)
        return out

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, grad_out):
        view = ctx.backward_cache
        out, = ctx.saved_tensors
        return None, sparse.edge_softmax_bwd_raw(view, out, grad_out.contiguous()), None


class GATAttention(torch.autograd.Function):
    """a = edge_softmax(leaky_relu(el[u] + er[v])) in one launch; d el / d er by two copy_e g-SpMMs of the fused de."""

    @staticmethod
    def forward(ctx, gidx, el, er, slope):
        csc = gidx.csc()
        if el.shape[0] != csc.num_cols or er.shape[0] != csc.num_rows:  # the kernel indexes el by source id, er by row
            raise DGLError("gat_attention: expected el with %d source rows and er with %d destination rows, got %d and %d"
                           % (csc.num_cols, csc.num_rows, el.shape[0], er.shape[0]))
        shape = el.shape[1:]
        el2, er2 = el.contiguous().view(el.shape[0], -1), er.contiguous().view(er.shape[0], -1)
        a = sparse.backend_for(el2).gat_attention_fwd(csc, el2, er2, float(slope))
        ctx.backward_cache = gidx, float(slope), shape
        ctx.save_for_backward(a, el2, er2)
        return a.view((a.shape[0],) + tuple(shape))

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, da):
        gidx, slope, shape = ctx.backward_cache
        a, el2, er2 = ctx.saved_tensors
        de = sparse.backend_for(a).gat_attention_bwd(gidx.csc(), el2, er2, slope, a, da.contiguous().view(a.shape))
        d_el = d_er = None
        if ctx.needs_input_grad[1]:
            d_el, _, _ = sparse.gspmm_raw(gidx.csr(), "copy_rhs", "sum", None, de)
            d_el = d_el.view((el2.shape[0],) + tuple(shape))
        if ctx.needs_input_grad[2]:
            d_er, _, _ = sparse.gspmm_raw(gidx.csc(), "copy_rhs", "sum", None, de)
            d_er = d_er.view((er2.shape[0],) + tuple(shape))
        return None, d_el, d_er, None


def gat_attention(graph, el, er, negative_slope=0.2):
    """Fused apply_edges(u_add_v) -> leaky_relu -> edge_softmax (norm_by='dst').  el: (N_src, H[, 1]), er: (N_dst, H[, 1])."""
    if el.dtype != torch.float32 or er.dtype != torch.float32:
        raise DGLError("gat_attention expects float32 attention terms")
    if el.shape[1:] != er.shape[1:]:
        raise DGLError("gat_attention: el %s and er %s disagree" % (tuple(el.shape[1:]), tuple(er.shape[1:])))
    return GATAttention.apply(_gidx(graph), el, er, negative_slope)


class GATFused(torch.autograd.Function):
    """out[v,h,:] = sum_e dropout(softmax_{e->v}(leaky_relu(el[u] + er[v])))[e,h] * feat[u,h,:] -- GATConv's whole
    message-passing block (main_dgl_reddit_gat.py:31-55) in two launches forward and two backward, with no E-sized
    tensor: the attention of an edge is rebuilt from per-node statistics where it is needed (csrc/gatfused.hip)."""

    _calls = 0  # advances the counter-based generator: a new mask per call, reproducible after torch.manual_seed

    @staticmethod
    def forward(ctx, gidx, feat, el, er, slope, p, attn_l=None):
        csc = gidx.csc()
        H, F = int(feat.shape[1]), int(feat.shape[2])
        if attn_l is not None:  # el IS (feat * attn_l).sum(-1): the kernels may form it from the gathered row (no gradient flows
            attn_l = attn_l.detach().contiguous().view(H, F)  # through this argument: d_el goes back through el's own producer)
        if el.shape[0] != csc.num_cols or er.shape[0] != csc.num_rows or feat.shape[0] != csc.num_cols:
            raise DGLError("gat_fused: expected feat / el with %d source rows and er with %d destination rows, got %d / %d / %d"
                           % (csc.num_cols, csc.num_rows, feat.shape[0], el.shape[0], er.shape[0]))
        feat = feat.contiguous()
        el2, er2 = el.contiguous().view(el.shape[0], H), er.contiguous().view(er.shape[0], H)
        seed = 0
        if p > 0.0:
            seed = ((torch.initial_seed() & (2 ** 64 - 1)) ^ ((GATFused._calls * 0x9E3779B97F4A7C15) & (2 ** 64 - 1)))
            GATFused._calls += 1
        be = sparse.backend_for(feat)
        if p > 0.0 and be.name == "hip":  # training with attn_drop: the backward needs the out-CSR anyway; the forward's choice of kernel too
            out, nstat, form = be.gat_fused_fwd(csc, feat, el2, er2, float(slope), float(p), seed, attn_l, csr=gidx.csr())
        else:
            out, nstat, form = be.gat_fused_fwd(csc, feat, el2, er2, float(slope), float(p), seed, attn_l)
        ctx.backward_cache = gidx, float(slope), float(p), seed, el.shape, er.shape, attn_l, form
        ctx.save_for_backward(feat, el2, out, nstat)
        return out

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, d_out):
        gidx, slope, p, seed, el_shape, er_shape, attn_l, form = ctx.backward_cache
        feat, el2, out, nstat = ctx.saved_tensors
        need_src = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        d_feat, d_el, d_er = sparse.backend_for(feat).gat_fused_bwd(gidx.csc(), gidx.csr(), feat, el2, slope, p, seed, out,
                                                                     d_out.contiguous(), nstat, need_src, attn_l, form=form)
        return (None, d_feat if ctx.needs_input_grad[1] else None,
                d_el.view(el_shape) if (d_el is not None and ctx.needs_input_grad[2]) else None,
                d_er.view(er_shape) if ctx.needs_input_grad[3] else None, None, None, None)


def gat_fused_supported(graph, feat):
    gidx = _gidx(graph)
    return (feat.dim() == 3 and feat.dtype == torch.float32 and feat.is_cuda and os.environ.get("MGX_GAT_FUSED", "1") == "1"
            and sparse.backend_for(feat).name == "hip"
            and sparse.backend_for(feat).gat_fused_supported(gidx.csc(), int(feat.shape[1]), int(feat.shape[2])))


def gat_fused(graph, feat, el, er, negative_slope=0.2, attn_drop=0.0, training=True, attn_l=None):
    """GATConv's u_add_v -> leaky_relu -> edge_softmax -> attn_drop -> u_mul_e/sum block.  feat (N_src, H, F),
    el (N_src, H[, 1]), er (N_dst, H[, 1]) -> (N_dst, H, F).  Check gat_fused_supported() first.
    `attn_l` (1 | -, H, F): pass it when el is exactly (feat * attn_l).sum(-1) -- multi-head layers then form el from the gathered
    feature row inside the kernels instead of gathering it (one L2 request per edge less)."""
    if feat.dtype != torch.float32 or el.dtype != torch.float32 or er.dtype != torch.float32:
        raise DGLError("gat_fused expects float32 inputs")
    p = float(attn_drop) if training else 0.0
    if not 0.0 <= p < 1.0:
        raise DGLError("gat_fused: attn_drop must be in [0, 1), got %g" % p)
    if p > 0.0 and feat.is_cuda and capture_path():
        raise DGLError("gat_fused: attn_drop > 0 under HIP-graph capture would replay ONE dropout mask (the seed is a launch "
                       "argument); use the unfused operators there, as GATConv does")
    return GATFused.apply(_gidx(graph), feat, el, er, negative_slope, p, attn_l)


class HeadDot(torch.autograd.Function):
    """(out_a, out_b)[n, h] = <feat[n, h, :], attn_{a,b}[h, :]>: one pass over feat forward, one pass backward."""

    @staticmethod
    def forward(ctx, feat, attn_a, attn_b):
        feat = feat.contiguous()
        H, F = feat.shape[1], feat.shape[2]
        a2 = attn_a.contiguous().view(H, F)
        b2 = attn_b.contiguous().view(H, F) if attn_b is not None else None
        out_a, out_b = sparse.backend_for(feat).head_dot_fwd(feat, a2, b2)
        ctx.save_for_backward(feat, a2, b2)
        ctx.attn_shape = attn_a.shape
        if out_b is None:  # autograd Functions return tensors only: an empty placeholder that carries no gradient
            out_b = feat.new_empty(0)
            ctx.mark_non_differentiable(out_b)
        return out_a, out_b

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, d_a, d_b):
        feat, a2, b2 = ctx.saved_tensors
        if d_a is None:
            d_a = torch.zeros(feat.shape[:2], dtype=feat.dtype, device=feat.device)
        if b2 is not None and d_b is None:
            d_b = torch.zeros(feat.shape[:2], dtype=feat.dtype, device=feat.device)
        d_feat, g_a, g_b = sparse.backend_for(feat).head_dot_bwd(
            feat, a2, b2, d_a.contiguous(), d_b.contiguous() if b2 is not None else None, ctx.needs_input_grad[0])
        return d_feat, g_a.view(ctx.attn_shape), (g_b.view(ctx.attn_shape) if g_b is not None else None)


def head_dot_supported(feat):
    return feat.dim() == 3 and feat.dtype == torch.float32 and sparse.backend_for(feat).head_dot_supported(
        int(feat.shape[1]), int(feat.shape[2]))


def head_dot(feat, attn_a, attn_b=None):
    """GATConv's `(feat * attn).sum(-1)`: feat (n, H, F), attn (1, H, F) -> (n, H); with attn_b both terms in one pass."""
    if feat.dtype != torch.float32:
        raise DGLError("head_dot expects float32 features")
    out_a, out_b = HeadDot.apply(feat, attn_a, attn_b)
    return out_a if attn_b is None else (out_a, out_b)


class LinearFn(torch.autograd.Function):
    """y = x W^T + b with the library's GEMMs; the bias gradient (column sum of dY over all N rows) runs in
    mgx_column_sum -- PyTorch's generic reduction needs 19 ms for a [2.4M, 47] column sum on MI355X, 40 % of a products epoch."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = (dy2 @ weight).view(x.shape)
        if ctx.needs_input_grad[1]:
            dw = _weight_grad(dy2, x.reshape(-1, x.shape[-1]))
        if ctx.needs_input_grad[2]:
            db = sparse.backend_for(dy2).column_sum(dy2.contiguous())
        return dx, dw, db


class BiasAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias):
        ctx.bias_shape = bias.shape
        return x + bias

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dy):
        db = None
        if ctx.needs_input_grad[1]:
            C = 1
            for d in ctx.bias_shape:
                C *= int(d)
            db = sparse.backend_for(dy).column_sum(dy.contiguous().view(-1, C)).view(ctx.bias_shape)
        return dy, db


def bias_add(x, bias):
    """x + bias (bias broadcast over the leading dimension of x) with the bias gradient computed by mgx_column_sum."""
    C = bias.numel()
    if x.dim() >= 2 and x[0].numel() == C:
        bias = bias.view((1,) + tuple(x.shape[1:]))
    if (not bias.requires_grad or x.dtype != torch.float32 or x.device.type not in sparse._BACKENDS or x.dim() < 2
            or x[0].numel() != C or C > sparse.backend_for(x).COLUMN_SUM_MAX):
        return x + bias
    return BiasAdd.apply(x, bias)


class ReluDropout(torch.autograd.Function):
    _calls = 0  # advances the counter-based generator: a new mask per call, reproducible after torch.manual_seed

    @staticmethod
    def forward(ctx, x, p, into=None):
        # `into` wraps the destination view in a plain object: handed over as a Tensor argument it would be an autograd INPUT
        # modified in place; created here it is simply this node's output (a view of a buffer autograd does not track)
        out = None if into is None else into.t
        be = sparse.backend_for(x)
        if out is None or not be._row_strided(x):
            x = x.contiguous()
        seed = torch.initial_seed() & (2 ** 64 - 1)
        if capture_path():
            # a HIP graph freezes launch arguments: the position in the random stream comes from a device counter that the
            # captured step itself advances, so every replay draws a new mask
            ctr = _capture_counter(x.device)
            y, mask = be.relu_dropout_fwd(x, float(p), seed, 0, out=out, counter=ctr)
            ctr.add_(1)
        else:
            offset = (ReluDropout._calls * 0x9E3779B97F4A7C15) & (2 ** 63 - 1)
            ReluDropout._calls += 1
            y, mask = be.relu_dropout_fwd(x, float(p), seed, offset, out=out)
        ctx.save_for_backward(mask)
        ctx.p = float(p)
        return y

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        be = sparse.backend_for(dy)
        if not dy.is_contiguous() and not be._row_strided(dy):
            dy = dy.contiguous()
        return be.relu_dropout_bwd(dy, mask, ctx.p), None, None


_WARMING_UP_FOR_CAPTURE = 0


def capture_path():
    """True while the current stream is being captured into a HIP graph AND during the eager warm-up steps that precede a
    capture (utils.GraphedStep, graph_classification.GraphedBatchTrainer): operators that choose a different form under
    capture must choose it in the warm-up too, so that everything the captured form needs -- BLAS handles, workspaces,
    lazily built plans -- exists before the capture starts."""
    return _WARMING_UP_FOR_CAPTURE > 0 or torch.cuda.is_current_stream_capturing()


_CAPTURE_COUNTERS = {}


def _capture_counter(device):
    """One int64 counter per device, advanced by every captured relu_dropout call (created outside any capture: warm-up)."""
    c = _CAPTURE_COUNTERS.get(device)
    if c is None:
        c = _CAPTURE_COUNTERS[device] = torch.full((1,), 1 << 20, dtype=torch.int64, device=device)
    return c


class warming_up_for_capture(object):
    def __enter__(self):
        global _WARMING_UP_FOR_CAPTURE
        _WARMING_UP_FOR_CAPTURE += 1

    def __exit__(self, *exc):
        global _WARMING_UP_FOR_CAPTURE
        _WARMING_UP_FOR_CAPTURE -= 1
        return False


def relu_dropout(x, p=0.5, training=True, out=None):
    """dropout(relu(x), p) in one pass each way (float32 HIP tensors with numel % 4 == 0; anything else and evaluation mode
    take the two PyTorch ops).  Under HIP-graph capture the mask's position in the random stream is read from a device
    counter that the captured step advances, so replays draw new masks.  `out`: a [rows, cols] view with
    unit column stride to write the result into (a column block of a wider matrix); ignored on the PyTorch path."""
    if (not training or p <= 0.0 or p >= 1.0 or x.dtype != torch.float32 or x.device.type != "cuda" or x.numel() % 4
            or x.device.type not in sparse._BACKENDS
            or (capture_path() and (x.dim() != 2 or x.shape[1] % 4 or x.stride(1) != 1 or x.stride(0) % 4))):
        return _structural_zeros(torch.nn.functional.dropout(torch.relu(x), p, training))
    if out is not None and (x.dim() != 2 or x.shape[1] % 4 or out.shape != x.shape or out.stride(1) != 1 or out.stride(0) % 4
                            or out.requires_grad):
        out = None
    return _structural_zeros(ReluDropout.apply(x, p, None if out is None else _Into(out)))


def _structural_zeros(y):
    """Tag a relu (+ dropout) output: its zeros are structural -- this node's backward multiplies whatever gradient arrives at a zero
    position by zero -- which lets a partition's halo exchange send the row as bitmap + non-zeros and take back only the gradient
    entries under that bitmap (dist.SparseHalo).  A plain attribute on the tensor object: it does not survive further ops."""
    y._mgx_structural_zeros = int(y._version)  # (an in-place write afterwards bumps the version and voids the tag: dist.structural_zeros)
    return y


def has_structural_zeros(t):
    """True when `t` carries _structural_zeros' tag and has not been written in place since."""
    tagged = getattr(t, "_mgx_structural_zeros", None)
    return tagged is not None and tagged is not False and int(tagged) == int(t._version)


class _SlotGate(object):
    """Whether a graph's forward aggregations of relu + dropout outputs take the 128-byte-slot form (csrc/spmm_slots.inc).  The form is
    exact at any density -- a row with more than 24 non-zeros is read from the dense matrix -- but only pays while such rows are rare, so
    every producer leaves its overflow count in pinned host memory (asynchronously) and the NEXT call reads it -- waiting for it if it
    should not have arrived yet, so that the decision is a function of the operands and the call sequence, not of timing: above
    config.PACKED_GATHER_MAX_OVERFLOW of the rows the dense kernels take the following 64 calls."""
    __slots__ = ("pending", "host", "dense_until", "calls", "last_fraction")

    def __init__(self):
        self.pending, self.host, self.dense_until, self.calls, self.last_fraction = None, None, 0, 0, None

    def allow(self):
        self.calls += 1
        if self.pending is not None:
            # WAIT for the previous producer's count (it was recorded right behind that kernel, at least one aggregation ago: arrived in
            # practice) rather than ask whether it happens to be there: the decision must not depend on timing, or reruns would differ
            self.pending[0].synchronize()
            self.last_fraction = float(self.host[0]) / max(self.pending[1], 1)
            self.pending = None
            if self.last_fraction > config.PACKED_GATHER_MAX_OVERFLOW:
                self.dense_until = self.calls + 64
        return self.calls > self.dense_until

    def open(self):
        """allow() without counting a call: asked by the PRODUCER of the rows (should its epilogue write the slots as well?)."""
        return self.calls >= self.dense_until

    def watch(self, overflow, n):
        if self.pending is not None:
            return
        if self.host is None:
            self.host = torch.empty(1, dtype=torch.int64).pin_memory()
        self.host.copy_(overflow, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending = (ev, int(n))


def _slot_gate(be, gidx, csc, n):
    """The gate of `gidx` when the forward aggregation of an [n, 64] relu + dropout output over `csc` may take the slot form at all."""
    if (not config.PACKED_GATHER or csc.nnz < config.PACKED_GATHER_MIN_NNZ or not hasattr(be, "rows_slots_pack") or capture_path()
            or csc.idx_bits != 32 or csc.num_cols != n or n >= (1 << 25) or csc.tile_plan(64) is not None):
        return None
    plan, short = csc.spmm_plan_for(64)
    if short and (plan is None or plan.rest is None):
        return None  # every work item is short: the lane-group kernel walks them all, nothing would read the slots
    gate = getattr(gidx, "_slot_gate", None)
    if gate is None:
        gate = gidx._slot_gate = _SlotGate()
    return gate


def _edge_tail_operands(be, csc, cat, h):
    """(block_a, edge_tail) of the layer's CONSTANT input h [N, 100] for mgx_spmm_copy_u_edge_tail, laid out on first use and kept on the
    layer's CatBuffer while h is the same unmodified tensor (the key CatBuffer.static_key already holds); None: the one-matrix form."""
    if (not config.EDGE_TAIL or cat.K != 100 or h.requires_grad or cat.static_key is None or cat.static_key[0] is not h
            or cat.static_key[1] != h._version or csc.nnz < config.EDGE_TAIL_MIN_NNZ or not hasattr(be, "edge_tail_of") or capture_path()):
        return None
    held = cat.static_tail
    if held is not None and held[0][0] is h and held[0][1] == h._version and held[0][2] is csc:
        return held[1]
    ops_ = be.edge_tail_of(csc, h)
    cat.static_tail = ((h, h._version, csc), ops_) if ops_ is not None else None
    return ops_


def _packed_rows(be, gidx, csc, h, left):
    """The 128-byte slots of `left` (== h, a [N, 64] relu + dropout output) for the forward aggregation over `csc`, or None: the dense
    rows.  The layer that produced h may have written them already (h._mgx_slots, the GEMM epilogue of sage_mean_layer_act)."""
    if left.shape[1] != 64 or not has_structural_zeros(h):
        return None
    gate = _slot_gate(be, gidx, csc, left.shape[0])
    if gate is None or not gate.allow():
        return None
    made = getattr(h, "_mgx_slots", None)
    if made is not None and made[2] == int(h._version) and made[0].shape[0] == left.shape[0]:
        slots, overflow = made[0], made[1]
    elif be.rows_slots_supported(left, csc):
        slots, overflow = be.rows_slots_pack(left)
    else:
        return None
    gate.watch(overflow, left.shape[0])
    return slots


class _Into(object):
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class BatchNormFn(torch.autograd.Function):
    """Training-mode BatchNorm1d over the rows of x [N, C]: two column reductions + one per-column affine map each way
    (mgx_column_pair_sums / mgx_column_affine).  Returns (y, batch mean, biased batch variance)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        be = sparse.backend_for(x)
        n = x.shape[0]
        # sums relative to the first row p: E[(x-p)^2] - E[x-p]^2 has O(std) terms, so |mean| >> std does not cancel
        s, ss = be.column_pair_sums(x, shifted=True)
        m1 = s / n
        mean = x[0] + m1
        var = (ss / n - m1 * m1).clamp_(min=0.0)
        invstd = torch.rsqrt(var + eps)
        A = invstd if weight is None else weight * invstd
        Cc = -mean * A if bias is None else bias - mean * A
        y = be.column_affine(x, A.contiguous(), Cc.contiguous())
        ctx.save_for_backward(x, weight, mean, invstd)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _dmean, _dvar):
        x, weight, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous()
        be = sparse.backend_for(dy)
        n = x.shape[0]
        sdy, sdyx = be.column_pair_sums(dy, x, shifted=True)  # sum dy * (x - x[0])
        sdyxhat = (sdyx - (mean - x[0]) * sdy) * invstd
        A = invstd if weight is None else weight * invstd
        B = -A * invstd * sdyxhat / n
        Cc = -A * sdy / n - B * mean
        dx = be.column_affine(dy, A.contiguous(), Cc.contiguous(), x, B.contiguous()) if ctx.needs_input_grad[0] else None
        dw = sdyxhat if weight is not None and ctx.needs_input_grad[1] else None
        db = sdy if ctx.needs_input_grad[2] else None
        return dx, dw, db, None


def batch_norm_supported(x):
    return (x.dim() == 2 and x.dtype == torch.float32 and x.device.type in sparse._BACKENDS and x.shape[1] % 4 == 0
            and 4 <= x.shape[1] <= sparse.backend_for(x).COLUMN_SUM_MAX and x.shape[0] > 1)


def _weight_grad(dy2, x2):
    be = sparse.backend_for(dy2)
    if (x2.shape[0] >= be.XTY_MIN_ROWS and dy2.shape[1] <= be.XTY_MAX[0] and x2.shape[1] <= be.XTY_MAX[1]
            and config.LINEAR_XTY):
        # millions of rows, <= 64 x 128 outputs: streamed once; mgx_xty takes row strides, so a column slice (the [:, :D] view of
        # a line-padded aggregation) is read in place
        return be.xty(dy2 if dy2.stride(1) == 1 else dy2.contiguous(), x2 if x2.stride(1) == 1 else x2.contiguous())
    return dy2.t() @ x2


def _weight_bias_grad(dy2, x2):
    """(dy^T x, column sums of dy): the weight and the bias gradient of a dense layer from ONE pass over dy where mgx_xty_colsum applies."""
    be = sparse.backend_for(dy2)
    if (x2.shape[0] >= be.XTY_MIN_ROWS and dy2.shape[1] <= be.XTY_MAX[0] and x2.shape[1] <= be.XTY_MAX[1]
            and config.LINEAR_XTY and config.XTY_COLSUM):
        return be.xty(dy2 if dy2.stride(1) == 1 else dy2.contiguous(), x2 if x2.stride(1) == 1 else x2.contiguous(), colsum=True)
    return _weight_grad(dy2, x2), be.column_sum(dy2 if dy2.is_contiguous() else dy2.contiguous())


class LinearSumFn(torch.autograd.Function):
    """y = x1 W1^T + x2 W2^T + b in two GEMMs, the second accumulating into the first's output (no separate add pass);
    gradients as LinearFn.  SAGEConv's `fc_self(h) + fc_neigh(h_neigh)` (main_dgl_product_sage.py:64)."""

    @staticmethod
    def forward(ctx, x1, w1, x2, w2, bias):
        ctx.save_for_backward(x1, w1, x2, w2)
        y = torch.nn.functional.linear(x1, w1, bias)
        return y.addmm_(x2, w2.t())

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dy):
        x1, w1, x2, w2 = ctx.saved_tensors
        dy = dy.contiguous()
        need = ctx.needs_input_grad
        dx1 = dy @ w1 if need[0] else None
        dw1 = _weight_grad(dy, x1) if need[1] else None
        dx2 = dy @ w2 if need[2] else None
        dw2 = _weight_grad(dy, x2) if need[3] else None
        db = sparse.backend_for(dy).column_sum(dy) if need[4] else None
        return dx1, dw1, dx2, dw2, db


def linear_sum(x1, weight1, x2, weight2, bias=None):
    """x1 @ weight1.T + x2 @ weight2.T + bias for 2-D float32 HIP tensors (falls back to two F.linear calls otherwise)."""
    if (x1.dim() != 2 or x2.dim() != 2 or x1.dtype != torch.float32 or x1.device.type not in sparse._BACKENDS
            or not torch.is_grad_enabled() or (bias is not None and weight1.shape[0] > sparse.backend_for(x1).COLUMN_SUM_MAX)):
        return torch.nn.functional.linear(x1, weight1) + torch.nn.functional.linear(x2, weight2, bias)
    return LinearSumFn.apply(x1, weight1, x2, weight2, bias)


class SelectRowsFn(torch.autograd.Function):
    """x[rows] for DISTINCT rows (a training-node index): the backward writes the incoming rows into a zero matrix with
    index_copy_ -- torch's x[rows] backward sorts the index to accumulate duplicates (a radix sort, merges and a serial
    accumulation per epoch: ~0.17 ms for the 196 k training nodes of the products shape) although there are none."""

    @staticmethod
    def forward(ctx, x, rows):
        ctx.save_for_backward(rows)
        ctx.n = x.shape[0]
        return x.index_select(0, rows)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (rows,) = ctx.saved_tensors
        dx = torch.zeros((ctx.n,) + tuple(dy.shape[1:]), dtype=dy.dtype, device=dy.device)
        dx.index_copy_(0, rows, dy)
        return dx, None


def select_distinct_rows(x, rows):
    """x[rows] where `rows` holds no duplicates (not checked: duplicates would lose gradient)."""
    return SelectRowsFn.apply(x, rows)


def nll_sum(logp, target):
    """-sum_i logp[i, target[i]] == F.nll_loss(logp, target, reduction='sum'): one gather and a parallel sum instead of
    torch's single-workgroup reduction kernels (0.22 + 0.19 ms forward + backward for 196 k rows on MI355X)."""
    return -logp.gather(1, target.view(-1, 1)).sum()


class SageMeanLayerFn(torch.autograd.Function):
    """One mean-aggregator GraphSAGE layer on a square graph as ONE autograd node:
        y = h W_self^T + mean_{u->v}(h[u]) W_neigh^T + b          (main_dgl_product_sage.py:52-64)
    Same kernels as update_all(copy_src, mean) followed by linear_sum; what the single node adds is the backward: h feeds
    both the self GEMM and the aggregation, and autograd would add their two gradients in a separate N x D pass -- here the
    reversed aggregation ACCUMULATES into the self GEMM's gradient (mgx_spmm_csr, MGX_SPMM_ACCUMULATE)."""

    @staticmethod
    def forward(ctx, gidx, h, w_self, w_neigh, bias):
        neigh, _, _ = _raw_gspmm(gidx.csc(), "copy_lhs", "mean", h, None)
        ctx.gidx = gidx
        ctx.save_for_backward(h, w_self, neigh, w_neigh)
        y = torch.nn.functional.linear(h, w_self, bias)
        return y.addmm_(neigh, w_neigh.t())

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dy):
        h, w_self, neigh, w_neigh = ctx.saved_tensors
        dy = dy.contiguous()
        need = ctx.needs_input_grad
        dh = None
        if need[1]:
            dh = dy @ w_self
            dn = dy @ w_neigh
            dn.mul_(ctx.gidx.csc().inv_degrees().view(-1, 1))  # d(sum / deg): one streaming pass, not a per-edge factor
            sparse.gspmm_raw(ctx.gidx.csr(), "copy_lhs", "sum", dn, None, accumulate_into=dh)
        dws = _weight_grad(dy, h) if need[2] else None
        dwn = _weight_grad(dy, neigh) if need[3] else None
        db = sparse.backend_for(dy).column_sum(dy) if need[4] else None
        return None, dh, dws, dwn, db


class CatBuffer(object):
    """An [N, 2 K] matrix whose left half holds a layer's input h and whose right half receives mean_{u->v} h[u]: the operand of
    the ONE GEMM `[h | neigh] [W_self | W_neigh]^T` that replaces SAGEConv's two.  `generation` counts the forward passes
    that wrote the right half; a backward pass checks that no later forward overwrote what it saved."""
    __slots__ = ("buf", "K", "generation", "static_key", "static_agg", "static_tail")

    def __init__(self, n, K, device):
        self.buf = torch.empty((n, 2 * K), dtype=torch.float32, device=device)
        self.K = K
        self.generation = 0
        self.static_key = None
        self.static_agg = None  # the static_key whose aggregation the right half holds (SageMeanStaticInputProjectFn)
        self.static_tail = None  # ((tensor, version, csc), block_a, edge_tail): the constant 100-column input laid out for mgx_spmm_copy_u_edge_tail

    @property
    def left(self):
        return self.buf[:, :self.K]

    @property
    def right(self):
        return self.buf[:, self.K:]

    def holds(self, h):
        return (h.dim() == 2 and h.shape == (self.buf.shape[0], self.K) and h.data_ptr() == self.buf.data_ptr()
                and h.stride(0) == self.buf.stride(0) and h.stride(1) == 1)


# mgx_rows_gemm (csrc/rowsgemm.hip) for the projections of a layer over a CatBuffer: tall inputs only (the library GEMM wins below,
# and the staging of B is per workgroup).  config.ROWS_GEMM = False: the library GEMM (+ the separate 1 / deg pass) everywhere.
_ROWS_GEMM_MIN = 1 << 16


def _rows_dgrad(be, dy, wcat, inv_deg, K):
    """(d h, d neigh / deg) = the two halves of dy @ wcat, the second times 1 / deg: mgx_rows_gemm with the factor in its epilogue and
    each half a compact matrix of its own (the reversed aggregation then gathers from one and accumulates into the other), else the
    GEMM and a streaming pass over that half, the halves two column blocks of one matrix."""
    if config.ROWS_GEMM and dy.is_cuda and dy.shape[0] >= _ROWS_GEMM_MIN and K % 4 == 0:
        pair = be.rows_gemm(dy, wcat, row_scale=inv_deg, scale_from=K, split_col=K)
        if pair is not None:
            return pair
    dcat = dy @ wcat
    dcat[:, K:].mul_(inv_deg.view(-1, 1))
    return dcat[:, :K], dcat[:, K:]


def _rows_linear(be, x2d, weight, bias):
    """F.linear(x2d, weight, bias) for a tall x2d through mgx_rows_gemm when it has the shape (weight: [out, in] as in nn.Linear).
    Outputs that are not a multiple of four columns wide (the 47 classes) keep the library GEMM: scalar stores, 0.49 against 0.47 ms."""
    if config.ROWS_GEMM and x2d.is_cuda and x2d.shape[0] >= _ROWS_GEMM_MIN and weight.shape[0] % 4 == 0:
        y = be.rows_gemm(x2d, weight, b_transposed=True, bias=bias)
        if y is not None:
            return y
    return torch.nn.functional.linear(x2d, weight, bias)


class SageMeanCatFn(torch.autograd.Function):
    """SageMeanLayerFn over a CatBuffer: the aggregation reads the left half and writes the right half in place
    (mgx_spmm_copy_u_strided), forward and weight gradients are ONE GEMM / ONE mgx_xty against the stacked weights, and the
    backward aggregation reads the right half of d[h | neigh] and accumulates into its left half."""

    @staticmethod
    def forward(ctx, gidx, cat, h, w_self, w_neigh, bias, act=None):
        csc = gidx.csc()
        be = sparse.backend_for(h)
        if not cat.holds(h):
            # a layer input that lives elsewhere -- the model's input features -- is copied into the left half; when it is the
            # SAME tensor object, unmodified (version counter), as in the previous pass the copy is skipped.  The buffer keeps
            # a reference to that tensor: a per-step temporary at a recycled address is a different object and is copied.
            same = cat.static_key is not None and cat.static_key[0] is h and cat.static_key[1] == h._version
            if not same or h.requires_grad or not config.SAGE_STATIC_CAT:
                cat.left.copy_(h)
                cat.static_key = None if h.requires_grad else (h, h._version)
        cat.generation += 1
        tail_ops = _edge_tail_operands(be, csc, cat, h)
        if tail_ops is not None:
            # the constant 100-column input: columns 0 .. 95 as a compact block (three cache lines per row instead of four and an eighth),
            # the last four laid out along the edge list -- once, for as long as the tensor is not written to
            be.spmm_copy_u_edge_tail(csc, "mean", tail_ops[0], tail_ops[1], cat.right)
        else:
            # a relu + dropout output (a hidden layer's input) is gathered as 128-byte slots, one cache line per edge instead of two
            be.spmm_copy_u_strided(csc, "mean", cat.left, cat.right, slots=_packed_rows(be, gidx, csc, h, cat.left))
        ctx.gidx, ctx.cat, ctx.generation = gidx, cat, cat.generation
        if act is not None:
            # relu + dropout in the GEMM's epilogue (mgx_rows_gemm_relu_dropout): the activation lands in the next layer's buffer and
            # the N x out pre-activation is never stored.  Same position in the random stream as ops.relu_dropout would take.
            p, into = act[0], act[1]
            made = act[2] if len(act) > 2 else None  # [slots, overflow, written?]: the activation's rows as 128-byte slots too
            seed = torch.initial_seed() & (2 ** 64 - 1)
            offset = (ReluDropout._calls * 0x9E3779B97F4A7C15) & (2 ** 63 - 1)
            ReluDropout._calls += 1
            wcat = torch.cat([w_self, w_neigh], dim=1)
            fused = be.rows_gemm_relu_dropout(cat.buf, wcat, True, bias, float(p), seed, offset, out=None if into is None else into.t,
                                              slots=None if made is None else made[0], overflow=None if made is None else made[1])
            if fused is not None and made is not None:
                made[2] = True
            if fused is None:  # no fused kernel for this operand after all: the composition it stands for, same seed and offset (same bits)
                y, mask = be.relu_dropout_fwd(_rows_linear(be, cat.buf, wcat, bias), float(p), seed, offset,
                                              out=None if into is None else into.t)
            else:
                y, mask = fused
            ctx.save_for_backward(w_self, w_neigh, mask)
            ctx.p = float(p)
            return y
        ctx.save_for_backward(w_self, w_neigh)
        ctx.p = None
        return _rows_linear(be, cat.buf, torch.cat([w_self, w_neigh], dim=1), bias)

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dy):
        cat = ctx.cat
        if cat.generation != ctx.generation:
            raise DGLError("SAGEConv: a later forward pass overwrote the [h | neigh] buffer this backward pass needs "
                           "(two forwards before one backward); set MGX_SAGE_CAT=0")
        need = ctx.needs_input_grad
        be = sparse.backend_for(dy)
        K = cat.K
        dh = None
        aggregated_first = False
        if ctx.p is not None:
            w_self, w_neigh, mask = ctx.saved_tensors
            if not dy.is_contiguous() and not be._row_strided(dy):
                dy = dy.contiguous()
            gate = None
            if need[2] and dy.dim() == 2 and dy.shape[1] == 64 and K % 4 == 0 and config.ROWS_GEMM and be.rows_gemm_supported(128, K, 128):
                gate = _slot_gate(be, ctx.gidx, ctx.gidx.csr(), dy.shape[0])
                if gate is not None and not gate.allow():
                    gate = None
            if gate is not None:
                # d h = d y W_self + A^T (D^-1 d y) W_neigh: the gradient behind relu + dropout is as sparse as the activation, so the
                # reversed aggregation runs on IT -- as 128-byte slots, 1 / deg folded into the packed values -- and the projection
                # follows as one GEMM on [d y | A^T D^-1 d y], instead of projecting first and aggregating 64 dense columns
                dcat = torch.empty((dy.shape[0], 128), dtype=torch.float32, device=dy.device)
                inv = ctx.gidx.csc().inv_degrees()
                if dy.data_ptr() % 16 == 0:
                    dy, slots, overflow = be.relu_dropout_bwd_slots(dy, mask, ctx.p, dcat[:, :64], row_scale=inv)
                else:
                    dy = be.relu_dropout_bwd(dy, mask, ctx.p, out=dcat[:, :64])
                    slots, overflow = be.rows_slots_pack(dy, row_scale=inv)
                gate.watch(overflow, dy.shape[0])
                be.spmm_copy_u_strided(ctx.gidx.csr(), "sum", dy, dcat[:, 64:], slots=slots, src_scale=inv)
                dh = be.rows_gemm(dcat, torch.cat([w_self, w_neigh], dim=0))  # [N, 128] x [128, K]
                if dh is None:
                    dh = dcat @ torch.cat([w_self, w_neigh], dim=0)
                aggregated_first = True
            else:
                dy = be.relu_dropout_bwd(dy, mask, ctx.p)  # the gradient of the pre-activation, as ReluDropout.backward forms it
        else:
            w_self, w_neigh = ctx.saved_tensors
        if not aggregated_first:
            dy = dy.contiguous()
        if need[2] and not aggregated_first:
            # [N, 2K] = d[h | neigh], the `neigh` half times 1 / deg (d(sum / deg)): one mgx_rows_gemm with the factor in its epilogue,
            # else the GEMM and a streaming pass over that half -- never a per-edge factor
            dh, dn = _rows_dgrad(be, dy, torch.cat([w_self, w_neigh], dim=1), ctx.gidx.csc().inv_degrees(), K)
            be.spmm_copy_u_strided(ctx.gidx.csr(), "sum", dn, dh, accumulate=True)
        dws = dwn = db = None
        if (need[3] or need[4]) and need[5]:
            dw, db = _weight_bias_grad(dy, cat.buf)               # [out, 2K], [out]: one pass over dy
            dws, dwn = dw[:, :K].contiguous(), dw[:, K:].contiguous()
        elif need[3] or need[4]:
            dw = _weight_grad(dy, cat.buf)                        # [out, 2K]
            dws, dwn = dw[:, :K].contiguous(), dw[:, K:].contiguous()
        elif need[5]:
            db = be.column_sum(dy if dy.is_contiguous() else dy.contiguous())
        return None, None, dh, dws, dwn, db, None


_ROW_BITS = {}  # (rows.data_ptr(), len, N) -> (rows, bitmap): the loss rows of a training run are one tensor, reused every step


def _row_bits(rows, n, be):
    key = (rows.data_ptr(), int(rows.shape[0]), int(n))
    held = _ROW_BITS.get(key)
    if held is None or held[0] is not rows:
        flag = torch.zeros((n, 4), dtype=torch.float32, device=rows.device)
        flag[rows] = 1.0
        if len(_ROW_BITS) > 8:
            _ROW_BITS.clear()
        held = _ROW_BITS[key] = (rows, be.row_nonzero_bits(flag))
    return held[1]


class SageMeanCatRowsFn(torch.autograd.Function):
    """SageMeanCatFn followed by the selection of DISTINCT output rows -- the loss rows, `model(g, feats)[train_idx]` at
    main_dgl_product_sage.py:105 -- as one node (MGX_SAGE_SPARSE_LAST=1; off by default).  The forward is the whole layer, every row
    aggregated and projected as the reference does; the backward uses what the selection implies: d y is zero outside `rows` (92 % of
    the rows of ogbn-products), so the dense gradients are formed on those rows only and the reversed aggregation skips the gathers
    of rows that are zero by construction (mgx_spmm_copy_u_masked with a bitmap of `rows`; they are never even written).  Exact."""

    @staticmethod
    def forward(ctx, gidx, cat, h, w_self, w_neigh, bias, rows):
        csc = gidx.csc()
        be = sparse.backend_for(h)
        if not cat.holds(h):
            cat.left.copy_(h)
            cat.static_key = None
        cat.generation += 1
        be.spmm_copy_u_strided(csc, "mean", cat.left, cat.right)
        ctx.gidx, ctx.cat, ctx.generation, ctx.rows = gidx, cat, cat.generation, rows
        ctx.save_for_backward(w_self, w_neigh)
        y = torch.nn.functional.linear(cat.buf, torch.cat([w_self, w_neigh], dim=1), bias)
        return y.index_select(0, rows)

    @staticmethod
    @once_differentiable
    def backward(ctx, dyr):
        w_self, w_neigh = ctx.saved_tensors
        cat, rows = ctx.cat, ctx.rows
        if cat.generation != ctx.generation:
            raise DGLError("SAGEConv: a later forward pass overwrote the [h | neigh] buffer this backward pass needs")
        dyr = dyr.contiguous()
        need = ctx.needs_input_grad
        be = sparse.backend_for(dyr)
        K, n = cat.K, cat.buf.shape[0]
        dh = None
        if need[2]:
            dcat = dyr @ torch.cat([w_self, w_neigh], dim=1)                     # [R, 2K]: d[h | neigh] on the loss rows
            dn = torch.empty((n, K), dtype=dyr.dtype, device=dyr.device)          # rows outside `rows` stay unwritten: never gathered
            dn.index_copy_(0, rows, dcat[:, K:] * ctx.gidx.csc().inv_degrees()[rows].view(-1, 1))
            dh = be.spmm_copy_u_masked(ctx.gidx.csr(), "sum", dn, _row_bits(rows, n, be))
            dh.index_add_(0, rows, dcat[:, :K])
        dws = dwn = None
        if need[3] or need[4]:
            dw = _weight_grad(dyr, cat.buf.index_select(0, rows))                  # [out, 2K]
            dws, dwn = dw[:, :K].contiguous(), dw[:, K:].contiguous()
        db = be.column_sum(dyr) if need[5] else None
        return None, None, dh, dws, dwn, db, None


def sage_mean_layer_rows(g, h, w_self, w_neigh, bias, cat, rows):
    """The last SAGE layer of a model whose loss reads `rows` only (distinct): SageMeanCatRowsFn when it applies, else None."""
    if (os.environ.get("MGX_SAGE_SPARSE_LAST", "0") != "1" or cat is None or hasattr(g, "sage_mean_layer") or rows is None
            or sage_mean_layer.__globals__["_cat_eligible"](g, h, cat) is False or type(g) is not DGLGraph or not h.is_cuda
            or not torch.is_grad_enabled() or h.dtype != torch.float32 or g.idtype != torch.int32 or _torch_ops() is not None
            or (bias is not None and w_self.shape[0] > sparse.backend_for(h).COLUMN_SUM_MAX)):
        return None
    return SageMeanCatRowsFn.apply(g._index, cat, h, w_self, w_neigh, bias, rows)


class SageMeanProjectFirstFn(torch.autograd.Function):
    """y = h W_self^T + mean_{u->v}(h[u] W_neigh^T) + b: the projection BEFORE the aggregation -- mean and the linear map commute,
    so this is SAGEConv's `fc_self(h) + fc_neigh(mean_agg(h))` (main_dgl_reddit_sage.py:73-80) with the aggregation running at
    the OUTPUT width (upstream dgl.nn.SAGEConv does the same when in_feats > out_feats: `lin_before_mp`).  reddit's first layer
    aggregates 16 columns instead of 602.  One GEMM gives [s | z] = h [W_self | W_neigh]^T, the aggregation reads the z half and
    accumulates into the s half in place; backward: dz = A_mean^T dy (one more aggregation at the output width), then
    d[W_self | W_neigh] = [dy | dz]^T h and dh = [dy | dz] [W_self ; W_neigh]."""

    @staticmethod
    def forward(ctx, gidx, h, w_self, w_neigh, bias):
        K = w_self.shape[0]
        w = torch.cat([w_self, w_neigh], dim=0)            # [2K, in]
        sz = torch.nn.functional.linear(h, w)               # [N, 2K] = [s | z]
        sparse.backend_for(h).spmm_copy_u_strided(gidx.csc(), "mean", sz[:, K:], sz[:, :K], accumulate=True)
        y = sz[:, :K].contiguous()
        if bias is not None:
            y.add_(bias)
        ctx.gidx = gidx
        ctx.save_for_backward(h, w)
        return y

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dy):
        h, w = ctx.saved_tensors
        K = w.shape[0] // 2
        need = ctx.needs_input_grad
        be = sparse.backend_for(dy)
        dcat = torch.empty((dy.shape[0], 2 * K), dtype=dy.dtype, device=dy.device)
        dcat[:, :K] = dy
        dn = dy * ctx.gidx.csc().inv_degrees().view(-1, 1)  # d(sum / deg): one streaming pass, not a per-edge factor
        be.spmm_copy_u_strided(ctx.gidx.csr(), "sum", dn, dcat[:, K:])
        dh = dcat @ w if need[1] else None
        dws = dwn = None
        if need[2] or need[3]:
            dw = _weight_grad(dcat, h)                       # [2K, in]
            dws, dwn = dw[:K], dw[K:]
        db = be.column_sum(dy.contiguous()) if need[4] else None
        return None, dh, dws, dwn, db


STATIC_AGGREGATIONS_BUILT = [0]  # how often SageMeanStaticInputProjectFn aggregated its constant input (bench.py reports it: once per model)


class SageMeanStaticInputProjectFn(torch.autograd.Function):
    """The FIRST SAGE layer of a full-graph model, whose input x is a constant (the node features: no gradient), with the
    projection before the aggregation (MGX_SAGE_L1_PROJECT_FIRST=1; off by default, reported beside the headline):

        y = x W_self^T + mean_agg(x W_neigh^T) + b        ==  fc_self(x) + fc_neigh(mean_agg(x))   (main_dgl_product_sage.py:61-64)

    so the products layer aggregates 64 columns instead of 100 (whole 256-byte rows instead of 400-byte rows that straddle 4.125
    cache lines).  SageMeanProjectFirstFn pays for that with a second aggregation in the backward (dz = A_mean^T dy) because its
    weight gradient is [dy | dz]^T h; here x is constant, so  dW_neigh = dy^T (A_mean x)  is taken against a CONSTANT matrix that is
    aggregated once and kept in the right half of the layer's CatBuffer for as long as the left half holds the same unmodified
    tensor object (identity + version counter, exactly like the resident copy of x itself) -- the forward aggregation still runs
    every pass, over every edge, and no aggregation is added to the backward."""

    @staticmethod
    def forward(ctx, gidx, cat, x, w_self, w_neigh, bias):
        be = sparse.backend_for(x)
        K = w_self.shape[0]
        same = cat.static_key is not None and cat.static_key[0] is x and cat.static_key[1] == x._version
        if not same or cat.static_agg is not cat.static_key:
            cat.left.copy_(x)
            cat.static_key = (x, x._version)
            be.spmm_copy_u_strided(gidx.csc(), "mean", cat.left, cat.right)      # A_mean x: once per (tensor, version)
            STATIC_AGGREGATIONS_BUILT[0] += 1
            cat.static_agg = cat.static_key
            cat.generation += 1
        w = torch.cat([w_self, w_neigh], dim=0)                                   # [2K, in]
        b2 = None if bias is None else torch.cat([bias, torch.zeros_like(bias)])
        sz = torch.nn.functional.linear(x, w, b2)                                 # [N, 2K] = [x W_self^T + b | x W_neigh^T]
        be.spmm_copy_u_strided(gidx.csc(), "mean", sz[:, K:], sz[:, :K], accumulate=True)
        ctx.cat, ctx.generation = cat, cat.generation
        return sz[:, :K]                                                           # row-strided view: relu_dropout reads it in place

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dy):
        cat = ctx.cat
        if cat.generation != ctx.generation:
            raise DGLError("SAGEConv: the cached [x | mean_agg(x)] buffer was rewritten before this backward pass")
        need = ctx.needs_input_grad
        be = sparse.backend_for(dy)
        if dy.stride(1) != 1:
            dy = dy.contiguous()
        D = cat.K
        dws = dwn = None
        if need[3] or need[4]:
            dw = _weight_grad(dy, cat.buf)                                        # [K, 2 in] = dy^T [x | A_mean x]
            dws, dwn = dw[:, :D].contiguous(), dw[:, D:].contiguous()
        db = be.column_sum(dy if dy.is_contiguous() else dy.contiguous()) if need[5] else None
        return None, None, None, dws, dwn, db


def sage_static_input_project(g, x, w_self, w_neigh, bias, cat):
    """SageMeanStaticInputProjectFn when it applies (opt-in switch, constant input, out_feats < in_feats), else None."""
    K, D = w_self.shape
    if (os.environ.get("MGX_SAGE_L1_PROJECT_FIRST", "0") != "1" or cat is None or x.requires_grad or type(g) is not DGLGraph
            or g.is_block or x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda or x.device.type not in sparse._BACKENDS
            or not torch.is_grad_enabled() or g.idtype != torch.int32 or cat.K != D or cat.buf.shape[0] != x.shape[0]
            or g.number_of_src_nodes() != g.number_of_dst_nodes() or x.shape[0] != g.number_of_src_nodes()
            or K % 4 or D % 4 or K >= D or _torch_ops() is not None or not _cat_eligible(g, x, cat)
            or (bias is not None and K > sparse.backend_for(x).COLUMN_SUM_MAX) or x.shape[0] * 2 * K * 4 >= (1 << 32)):
        return None
    return SageMeanStaticInputProjectFn.apply(g._index, cat, x, w_self, w_neigh, bias)


def sage_project_first(g, h, w_self, w_neigh, bias=None):
    """SAGEConv with the projection before the aggregation, or None when that form does not apply / does not pay: a square int32
    DGLGraph, 2-D float32 HIP features, out_feats a multiple of 4, and TWO aggregations at the output width (forward and
    backward) cheaper than ONE at the input width (the input needs no gradient) resp. two (it does)."""
    K, D = w_self.shape
    if hasattr(g, "sage_project_first"):  # dist.DistGraph: the same form with the halo exchange at the output width
        return g.sage_project_first(h, w_self, w_neigh, bias)
    if (type(g) is not DGLGraph or g.is_block or h.dim() != 2 or h.dtype != torch.float32 or not h.is_cuda
            or h.device.type not in sparse._BACKENDS or not torch.is_grad_enabled() or g.idtype != torch.int32
            or g.number_of_src_nodes() != g.number_of_dst_nodes() or h.shape[0] != g.number_of_src_nodes()
            or K % 4 or K > 128 or _torch_ops() is not None or not config.SAGE_PROJECT_FIRST
            or (bias is not None and K > sparse.backend_for(h).COLUMN_SUM_MAX)
            or g.number_of_src_nodes() * 2 * K * 4 >= (1 << 32)):
        return None
    aggs_now = 2 if h.requires_grad else 1
    if 2 * max(K, 16) * 1.1 > aggs_now * D:  # rows narrower than 64 bytes cost as much as 64-byte rows (request bound)
        return None
    return SageMeanProjectFirstFn.apply(g._index, h, w_self, w_neigh, bias)


def sage_mean_layer(g, h, w_self, w_neigh, bias=None, cat=None):
    """Fused form of SAGEConv on a homogeneous square DGLGraph with 2-D float32 HIP features; None when the fused node does
    not apply (the caller then composes update_all + linear_sum)."""
    if hasattr(g, "sage_mean_layer"):  # dist.DistGraph: the same one-GEMM layer with the halo exchange inside
        return g.sage_mean_layer(h, w_self, w_neigh, bias, cat)
    if (type(g) is not DGLGraph or g.is_block or h.dim() != 2 or h.dtype != torch.float32
            or not h.is_cuda or h.device.type not in sparse._BACKENDS or not torch.is_grad_enabled()
            or g.number_of_src_nodes() != g.number_of_dst_nodes() or h.shape[0] != g.number_of_src_nodes()
            or g.idtype != torch.int32  # the accumulating aggregation is exercised on the int32 kernels only
            or (bias is not None and w_self.shape[0] > sparse.backend_for(h).COLUMN_SUM_MAX)
            or _torch_ops() is not None or not config.SAGE_FUSED_LAYER):
        return None
    if cat is not None and _cat_eligible(g, h, cat):
        return SageMeanCatFn.apply(g._index, cat, h, w_self, w_neigh, bias)
    return SageMeanLayerFn.apply(g._index, h, w_self, w_neigh, bias)


def sage_mean_layer_act(g, h, w_self, w_neigh, bias, cat, p, out):
    """dropout(relu(SAGEConv(g, h)), p) as ONE node whose GEMM applies the activation in its epilogue and writes `out` (the left half of
    the next layer's CatBuffer, or None for a new matrix): bit for bit ops.relu_dropout(ops.sage_mean_layer(...), out=out), one pass
    over the N x out pre-activation less each way.  None when that form does not apply (the caller composes the two)."""
    if hasattr(g, "sage_mean_layer_act"):  # dist.DistGraph: the same fused layer with the halo exchange inside
        return g.sage_mean_layer_act(h, w_self, w_neigh, bias, cat, p, out)
    if (not config.SAGE_FUSED_ACT or not config.ROWS_GEMM or type(g) is not DGLGraph or cat is None or capture_path()
            or not (0.0 < p < 1.0) or h.dim() != 2 or h.dtype != torch.float32 or not h.is_cuda or h.device.type not in sparse._BACKENDS
            or not torch.is_grad_enabled() or g.is_block or g.number_of_src_nodes() != g.number_of_dst_nodes()
            or h.shape[0] != g.number_of_src_nodes() or h.shape[0] < _ROWS_GEMM_MIN or g.idtype != torch.int32
            or _torch_ops() is not None or not config.SAGE_FUSED_LAYER or not _cat_eligible(g, h, cat)
            or (bias is not None and w_self.shape[0] > sparse.backend_for(h).COLUMN_SUM_MAX) or w_self.shape[0] % 4
            or not sparse.backend_for(h).rows_gemm_supported(2 * cat.K, w_self.shape[0], cat.buf.stride(0))
            or (out is not None and (out.shape != (h.shape[0], w_self.shape[0]) or out.stride(1) != 1 or out.stride(0) % 4
                                     or out.data_ptr() % 16 or out.requires_grad))):
        return None
    K, D = w_self.shape
    if config.SAGE_PROJECT_FIRST and K % 4 == 0 and K <= 128 and 2 * max(K, 16) * 1.1 <= (2 if h.requires_grad else 1) * D:
        return None  # sage_project_first's rule: this layer aggregates fewer columns projected first (reddit: 602 -> 16)
    if os.environ.get("MGX_SAGE_L1_PROJECT_FIRST", "0") == "1" and not h.requires_grad and K < D:
        return None  # the opt-in layer-1 form (sage_static_input_project) takes this layer
    # the NEXT layer aggregates this layer's output over the same graph: when that aggregation will gather 128-byte slots (64 columns,
    # _slot_gate), the GEMM's epilogue writes them beside the rows and the pack pass over the N x 64 activation is saved
    made = None
    be = sparse.backend_for(h)
    if K == 64 and out is not None:
        gate = _slot_gate(be, g._index, g._index.csc(), h.shape[0])
        if gate is not None and gate.open():
            made = [torch.empty((h.shape[0], 32), dtype=torch.int32, device=h.device), torch.zeros(1, dtype=torch.int64, device=h.device), False]
    y = _structural_zeros(SageMeanCatFn.apply(g._index, cat, h, w_self, w_neigh, bias, (float(p), None if out is None else _Into(out), made)))
    if made is not None and made[2]:
        y._mgx_slots = (made[0], made[1], int(y._version))
    return y


def _cat_eligible(g, h, cat):
    K = h.shape[1]
    idx = g._index
    return (config.SAGE_CAT and cat.K == K and K % 4 == 0 and cat.buf.shape[0] == h.shape[0]
            and idx.csc().indptr.dtype == torch.int32 and h.shape[0] * 2 * K * 4 < (1 << 32))


def cat_buffer_for(g, x, K):
    """A CatBuffer for a layer whose input has K columns, or None when the one-GEMM form does not apply to this graph / width."""
    if ((type(g) is not DGLGraph and not hasattr(g, "sage_mean_layer")) or g.is_block or x.dtype != torch.float32 or not x.is_cuda or x.device.type not in sparse._BACKENDS
            or not torch.is_grad_enabled() or K % 4 or g.number_of_src_nodes() != g.number_of_dst_nodes()
            or g.number_of_src_nodes() * 2 * K * 4 >= (1 << 32) or g.idtype != torch.int32
            or not config.SAGE_CAT or not config.SAGE_FUSED_LAYER):
        return None
    return CatBuffer(g.number_of_src_nodes(), K, x.device)


def linear(x, weight, bias=None):
    """torch.nn.functional.linear whose bias gradient (column sum) and tall-skinny weight gradient (dY^T X) are computed
    by the library (float32 HIP tensors; <= 256 outputs with a bias)."""
    if (x.dtype != torch.float32 or x.device.type not in sparse._BACKENDS or not torch.is_grad_enabled()
            or (bias is not None and weight.shape[0] > sparse.backend_for(x).COLUMN_SUM_MAX)):
        return torch.nn.functional.linear(x, weight, bias)
    return LinearFn.apply(x, weight, bias)


class SegmentReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, op, x, offsets, total):
        out, arg = sparse.segment_reduce_raw(offsets, x, op, want_arg=op in ("max", "min"), total=total)
        ctx.backward_cache = op, x.shape[0]
        ctx.save_for_backward(arg, offsets)
        return out

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dy):
        op, n = ctx.backward_cache
        arg, offsets = ctx.saved_tensors
        dy = dy.contiguous()
        if op in ("sum", "mean"):
            lens = offsets[1:] - offsets[:-1]
            if op == "mean":
                dy = dy / lens.clamp(min=1).to(dy.dtype).view((-1,) + (1,) * (dy.dim() - 1))
            seg = torch.repeat_interleave(torch.arange(lens.shape[0], device=dy.device), lens, output_size=n)
            dx = sparse.gather_rows_raw(dy, seg) if dy.is_cuda else dy[seg]
        else:
            dx = torch.zeros((n,) + tuple(dy.shape[1:]), dtype=dy.dtype, device=dy.device)
            valid = arg >= 0
            dx.scatter_add_(0, arg.clamp(min=0), dy * valid)
        return None, dx, None, None


# ----------------------------------------------------------------------------- public API
def gspmm(g, op, reduce_op, lhs_data, rhs_data):
    """Generalized SpMM: out[v] = reduce_{(u->v)} op(lhs[u], rhs[e]).  (kernel/dgl-new.py:20)"""
    if op not in ("add", "sub", "mul", "div", "copy_lhs", "copy_rhs"):
        raise DGLError("Unsupported binary op %r for gspmm" % (op,))
    if reduce_op not in ("sum", "max", "min", "mean"):
        raise DGLError("Unsupported reduce op %r for gspmm" % (reduce_op,))
    gidx = _gidx(g)
    if op == "sub":  # same rewrites as DGL's ops/spmm.py
        op, rhs_data = "add", -rhs_data
    elif op == "div":
        op, rhs_data = "mul", 1.0 / rhs_data
    expand_l = expand_r = False
    if op != "copy_rhs":
        if lhs_data is None:
            raise DGLError("gspmm: op %r needs lhs_data" % op)
        if lhs_data.dim() == 1:
            lhs_data, expand_l = lhs_data.unsqueeze(-1), True
    if op != "copy_lhs":
        if rhs_data is None:
            raise DGLError("gspmm: op %r needs rhs_data" % op)
        if rhs_data.dim() == 1:
            rhs_data, expand_r = rhs_data.unsqueeze(-1), True
    X = None if op == "copy_rhs" else lhs_data
    Y = None if op == "copy_lhs" else rhs_data
    out = GSpMM.apply(gidx, op, reduce_op, X, Y)
    squeeze = (expand_l or X is None) and (expand_r or Y is None) and (expand_l or expand_r)
    return out.squeeze(-1) if squeeze else out


def gsddmm(g, op, lhs_data, rhs_data, lhs_target="u", rhs_target="v"):
    """Generalized SDDMM: out[e] = op(lhs[t_l(e)], rhs[t_r(e)]).  (kernel/dgl-new.py:39)"""
    if op not in ("add", "sub", "mul", "div", "dot", "copy_lhs", "copy_rhs"):
        raise DGLError("Unsupported binary op %r for gsddmm" % (op,))
    for t in (lhs_target, rhs_target):
        if t not in ("u", "e", "v"):
            raise DGLError("Unsupported target %r; expected 'u', 'e' or 'v'" % (t,))
    gidx = _gidx(g)
    expand_l = expand_r = False
    if op != "copy_rhs" and lhs_data is not None and lhs_data.dim() == 1:
        lhs_data, expand_l = lhs_data.unsqueeze(-1), True
    if op != "copy_lhs" and rhs_data is not None and rhs_data.dim() == 1:
        rhs_data, expand_r = rhs_data.unsqueeze(-1), True
    X = None if op == "copy_rhs" else lhs_data
    Y = None if op == "copy_lhs" else rhs_data
    out = GSDDMM.apply(gidx, op, X, Y, lhs_target, rhs_target)
    squeeze = (expand_l or X is None) and (expand_r or Y is None) and (expand_l or expand_r) and op != "dot"
    return out.squeeze(-1) if squeeze else out


def edge_softmax(graph, logits, eids="__ALL__", norm_by="dst"):
    """Softmax of edge logits over the in-edges of every destination node (fused kernel)."""
    if not isinstance(eids, str):
        raise DGLError("edge_softmax on an edge subset is not supported by this backend")
    if norm_by not in ("dst", "src"):
        raise DGLError("norm_by must be 'dst' or 'src'")
    return EdgeSoftmax.apply(_gidx(graph), logits, norm_by)


def segment_reduce(seglen, value, reducer="sum", total=None):
    """dgl.ops.segment_reduce: reduce consecutive row segments of `value` (AvgPooling readout)."""
    if reducer not in ("sum", "mean", "max", "min"):
        raise DGLError("Unsupported segment reducer %r" % (reducer,))
    # value must have sum(seglen) rows: known on the host for host lengths, or passed by a caller that knows it
    # (a batched graph's node count); otherwise read back once
    if total is None and not seglen.is_cuda:
        total = int(seglen.sum())
    offsets = torch.zeros(seglen.shape[0] + 1, dtype=torch.int64, device=value.device)
    torch.cumsum(seglen.to(value.device).long(), 0, out=offsets[1:])
    return SegmentReduce.apply(reducer, value, offsets, total)


# named shortcuts (dgl.ops.copy_u_sum, ...)
def copy_u_sum(g, x):
    return gspmm(g, "copy_lhs", "sum", x, None)


def copy_u_mean(g, x):
    return gspmm(g, "copy_lhs", "mean", x, None)


def copy_e_sum(g, x):
    return gspmm(g, "copy_rhs", "sum", None, x)


def u_mul_e_sum(g, x, y):
    return gspmm(g, "mul", "sum", x, y)


def u_add_v(g, x, y):
    return gsddmm(g, "add", x, y, "u", "v")


def u_dot_v(g, x, y):
    return gsddmm(g, "dot", x, y, "u", "v")
