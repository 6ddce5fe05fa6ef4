"""CPU (OpenMP) backend: the message-passing primitives on HOST tensors through libmi355x_graph_cpu.so (csrc/cpu_ops.cpp,
include/mi355x_graph_cpu.h) -- SURVEY 8b's "CPU (OpenMP) variants of each with the same signatures", what the reference reaches
with `--gpu -1` (kernel/dgl-new.py:55-58) or on a machine without a GPU (main_dgl_product_sage.py:149; BASELINE configs[0]).

OPT-IN.  By default message passing on CPU tensors raises DGLError (this library is the MI355X backend); after
`mi355x_graph.enable_cpu_backend()` -- or with MGX_CPU_BACKEND=1 in the environment -- CPU tensors compute here.  There is no
fallback in either direction: HIP tensors never come here, and a missing libmi355x_graph.so still fails at the first HIP op.

g-SpMM, g-SDDMM, edge softmax, segment reduce and the format conversions are the C++ kernels; what the layers around them need on
dense matrices (column sums, X^T Y, BatchNorm sums, the per-head dot products of GATConv, row gathers, the packed halo rows of
dist.SparseHalo) is PyTorch's own CPU arithmetic -- dense algebra on the host is PyTorch's job, not this library's.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import DGLError, MgxCsr, OP, REDUCE, TARGET

CPU_LIB_PATH = os.path.join(_lib.CSRC_DIR, "libmi355x_graph_cpu.so")
_i32, _i64, _vp, _fp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p
_csr_p = ctypes.POINTER(MgxCsr)

SIGNATURES = {  # must list every function of include/mi355x_graph_cpu.h (tests/test_cpu_backend.py checks it)
    "mgx_cpu_last_error": (ctypes.c_char_p, []),
    "mgx_cpu_num_threads": (_i32, []),
    "mgx_cpu_set_num_threads": (None, [_i32]),
    "mgx_cpu_spmm_csr": (_i32, [_csr_p, _vp, _i32, _i32, _fp, _fp, _i64, _i64, _i64, _vp, _vp, _fp, _fp, _fp, _vp, _vp, _fp, _i32, _vp]),
    "mgx_cpu_sddmm_coo": (_i32, [_i64, _i64, _i64, _vp, _vp, _i32, _i32, _fp, _fp, _i32, _i32, _i64, _i64, _i64, _i64, _vp, _vp, _fp, _vp]),
    "mgx_cpu_sddmm_csr": (_i32, [_csr_p, _vp, _i32, _fp, _fp, _i32, _i32, _i64, _i64, _i64, _i64, _vp, _vp, _fp, _vp]),
    "mgx_cpu_edge_softmax_fwd": (_i32, [_csr_p, _vp, _i64, _fp, _fp, _fp, _vp]),
    "mgx_cpu_edge_softmax_bwd": (_i32, [_csr_p, _vp, _i64, _fp, _fp, _fp, _fp, _vp]),
    "mgx_cpu_segment_reduce": (_i32, [_i64, _vp, _i64, _i32, _fp, _fp, _vp, _vp]),
    "mgx_cpu_coo_to_csr": (_i32, [_i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mgx_cpu_csr_transpose": (_i32, [_csr_p, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mgx_cpu_csr_degrees": (_i32, [_i64, _vp, _i32, _vp, _vp]),
}

_handle = None


def lib():
    global _handle
    if _handle is None:
        if not os.path.exists(CPU_LIB_PATH):
            raise DGLError("CPU backend: %s is missing -- build it with `make -C %s`" % (CPU_LIB_PATH, _lib.CSRC_DIR))
        h = ctypes.CDLL(CPU_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _handle = h
    return _handle


def _check(status):
    if status != 0:
        msg = lib().mgx_cpu_last_error()
        raise DGLError("%s (mgx status %d)" % (msg.decode() if msg else "unknown error", status))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _host(*tensors):
    for t in tensors:
        if t is not None and t.device.type != "cpu":
            raise DGLError("CPU backend: a %s tensor reached the CPU kernels (graph and features must live on one device)" % t.device.type)


def _f32(t):
    return None if t is None else t.contiguous()


class CpuBackend(object):
    name = "cpu"
    COLUMN_SUM_MAX = 1 << 30
    XTY_MAX = (1 << 30, 1 << 30)
    XTY_MIN_ROWS = 1 << 16

    # ---- structure
    @staticmethod
    def _csr(csr):
        _host(csr.indptr, csr.indices, csr.eids)
        return csr.c_struct()

    def degrees(self, csr):
        deg = torch.empty(csr.num_rows, dtype=csr.indptr.dtype)
        _check(lib().mgx_cpu_csr_degrees(csr.num_rows, _ptr(csr.indptr), csr.idx_bits, _ptr(deg), None))
        return deg

    def inv_degrees(self, csr):
        return 1.0 / self.degrees(csr).clamp(min=1).to(torch.float32)

    def coo_to_csr(self, num_rows, num_cols, row, col):
        from . import sparse
        row, col = row.contiguous(), col.contiguous()
        nnz = int(row.shape[0])
        indptr = torch.empty(num_rows + 1, dtype=row.dtype)
        indices, eids = torch.empty(nnz, dtype=row.dtype), torch.empty(nnz, dtype=row.dtype)
        _check(lib().mgx_cpu_coo_to_csr(num_rows, nnz, _ptr(row), _ptr(col), 32 if row.dtype == torch.int32 else 64, _ptr(indptr), _ptr(indices),
                                        _ptr(eids), None, 0, None))
        return sparse.CsrView(num_rows, num_cols, indptr, indices, eids)

    def csr_transpose(self, csr):
        from . import sparse
        ip = torch.empty(csr.num_cols + 1, dtype=csr.indptr.dtype)
        ix, ei = torch.empty(csr.nnz, dtype=csr.indptr.dtype), torch.empty(csr.nnz, dtype=csr.indptr.dtype)
        _check(lib().mgx_cpu_csr_transpose(ctypes.byref(self._csr(csr)), _ptr(ip), _ptr(ix), _ptr(ei), None, 0, None))
        return sparse.CsrView(csr.num_cols, csr.num_rows, ip, ix, ei)

    # ---- the primitives
    def spmm(self, csr, op, reduce, U, E, u_len, e_len, out_len, u_off, e_off, src_scale, dst_scale, want_arg, accumulate_into=None):
        U, E = _f32(U), _f32(E)
        _host(U, E, u_off, e_off, src_scale, dst_scale, accumulate_into)
        n = csr.num_rows
        if accumulate_into is not None:
            if not accumulate_into.is_contiguous() or accumulate_into.numel() != n * out_len:
                raise DGLError("CPU backend: accumulate_into must be a dense [rows, %d] matrix" % out_len)
            out = accumulate_into.view(n, out_len)
        else:
            out = torch.empty((n, out_len), dtype=torch.float32)
        arg = reduce in ("max", "min") and want_arg
        idt = csr.indptr.dtype
        arg_u = torch.empty((n, out_len), dtype=idt) if arg and op != "copy_rhs" else None
        arg_e = torch.empty((n, out_len), dtype=idt) if arg and op != "copy_lhs" else None
        _check(lib().mgx_cpu_spmm_csr(ctypes.byref(self._csr(csr)), None, OP[op], REDUCE[reduce], _ptr(U), _ptr(E), u_len, e_len, out_len,
                                      _ptr(u_off), _ptr(e_off), _ptr(src_scale), _ptr(dst_scale), _ptr(out), _ptr(arg_u), _ptr(arg_e), None,
                                      1 if accumulate_into is not None else 0, None))
        return out, arg_u, arg_e

    def sddmm(self, gidx, op, L, R, lt, rt, l_len, r_len, out_len, reduce_size, l_off, r_off):
        L, R = _f32(L), _f32(R)
        _host(L, R, l_off, r_off)
        nnz = gidx.num_edges()
        out = torch.empty((nnz, out_len), dtype=torch.float32)
        if gidx.has_format("coo") or not gidx.has_format("csc"):
            src, dst = gidx.coo()
            src, dst = src.contiguous(), dst.contiguous()
            _host(src, dst)
            _check(lib().mgx_cpu_sddmm_coo(gidx.num_src, gidx.num_dst, nnz, _ptr(src), _ptr(dst), 32 if src.dtype == torch.int32 else 64,
                                           OP[op], _ptr(L), _ptr(R), TARGET[lt], TARGET[rt], l_len, r_len, out_len, reduce_size,
                                           _ptr(l_off), _ptr(r_off), _ptr(out), None))
        else:
            _check(lib().mgx_cpu_sddmm_csr(ctypes.byref(self._csr(gidx.csc())), None, OP[op], _ptr(L), _ptr(R), TARGET[lt], TARGET[rt],
                                           l_len, r_len, out_len, reduce_size, _ptr(l_off), _ptr(r_off), _ptr(out), None))
        return out

    def edge_softmax_fwd(self, csr, z2d):
        z2d = _f32(z2d)
        _host(z2d)
        a = torch.empty_like(z2d)
        _check(lib().mgx_cpu_edge_softmax_fwd(ctypes.byref(self._csr(csr)), None, z2d.shape[1], _ptr(z2d), _ptr(a), None, None))
        return a

    def edge_softmax_bwd(self, csr, a2d, da2d):
        a2d, da2d = _f32(a2d), _f32(da2d)
        _host(a2d, da2d)
        dz = torch.empty_like(a2d)
        _check(lib().mgx_cpu_edge_softmax_bwd(ctypes.byref(self._csr(csr)), None, a2d.shape[1], _ptr(a2d), _ptr(da2d), _ptr(dz), None, None))
        return dz

    def segment_reduce(self, offsets, x2d, reduce, want_arg):
        offsets, x2d = offsets.contiguous(), _f32(x2d)
        _host(offsets, x2d)
        n = offsets.shape[0] - 1
        out = torch.empty((n, x2d.shape[1]), dtype=torch.float32)
        arg = torch.empty((n, x2d.shape[1]), dtype=torch.int64) if want_arg and reduce in ("max", "min") else None
        _check(lib().mgx_cpu_segment_reduce(n, _ptr(offsets), x2d.shape[1], REDUCE[reduce], _ptr(x2d), _ptr(out), _ptr(arg), None))
        return out, arg

    # ---- GAT attention as its composition: logits by edge, then the softmax kernels (the fused HIP kernels have no CPU twin)
    @staticmethod
    def _edge_rows(csr):
        return torch.repeat_interleave(torch.arange(csr.num_rows), (csr.indptr[1:] - csr.indptr[:-1]).long())

    def _gat_logits(self, csr, el2d, er2d, slope):
        t = el2d[csr.indices.long()] + er2d[self._edge_rows(csr)]  # CSR-position order
        if csr.eids is not None:                                    # -> edge-id order
            by_id = torch.empty_like(t)
            by_id[csr.eids.long()] = t
            t = by_id
        return torch.where(t > 0, t, t * slope).contiguous(), t

    def gat_attention_fwd(self, csr, el2d, er2d, slope):
        return self.edge_softmax_fwd(csr, self._gat_logits(csr, el2d, er2d, slope)[0])

    def gat_attention_bwd(self, csr, el2d, er2d, slope, a2d, da2d):
        t = self._gat_logits(csr, el2d, er2d, slope)[1]
        return self.edge_softmax_bwd(csr, a2d, da2d) * torch.where(t > 0, torch.ones_like(t), torch.full_like(t, slope))

    @staticmethod
    def gat_fused_supported(csr, H, F):
        return False

    # ---- dense-side helpers: PyTorch's CPU arithmetic
    @staticmethod
    def head_dot_supported(H, F):
        return True

    def head_dot_fwd(self, feat3d, attn_a, attn_b):
        return (feat3d * attn_a).sum(-1), ((feat3d * attn_b).sum(-1) if attn_b is not None else None)

    def head_dot_bwd(self, feat3d, attn_a, attn_b, d_a, d_b, need_feat_grad):
        d_feat = d_a.unsqueeze(-1) * attn_a
        g_a = (d_a.unsqueeze(-1) * feat3d).sum(0)
        g_b = None
        if attn_b is not None:
            d_feat = d_feat + d_b.unsqueeze(-1) * attn_b
            g_b = (d_b.unsqueeze(-1) * feat3d).sum(0)
        return (d_feat if need_feat_grad else None), g_a, g_b

    def column_pair_sums(self, a2d, b2d=None, shifted=False):
        a = a2d
        if b2d is None:
            if shifted:
                a = a - a[0]
            return a.sum(0), (a * a).sum(0)
        b = b2d - b2d[0] if shifted else b2d
        return a.sum(0), (a * b).sum(0)

    def column_affine(self, a2d, A, Cc, b2d=None, B=None):
        out = a2d * A + Cc
        return out if b2d is None else out + b2d * B

    def xty(self, a2d, b2d, colsum=False):
        out = a2d.t() @ b2d
        return (out, a2d.sum(0)) if colsum else out

    def column_sum(self, x2d):
        return x2d.sum(0)

    def gather_rows(self, x2d, idx):
        return x2d.index_select(0, idx.long())

    def scatter_add_rows(self, x2d, idx, rows2d):
        x2d.index_add_(0, idx.long(), rows2d)
        return x2d

    # ---- halo rows as bitmaps + packed values (dist.SparseHalo over gloo): the contract of csrc/rowpack.hip in tensor ops; any bit order
    # that pack and unpack share is valid -- here column order
    @staticmethod
    def rows_pack_supported(x2d):
        return x2d.dim() == 2 and x2d.dtype == torch.float32 and x2d.shape[1] % 4 == 0 and 4 <= x2d.shape[1] <= 256

    @staticmethod
    def _bits(D):
        W = (D + 63) // 64
        return W, (torch.ones((), dtype=torch.int64) << (torch.arange(W * 64) % 64)).view(W, 64)

    def _flags(self, masks, D):
        W, bit = self._bits(D)
        return ((masks.view(-1, W, 1) & bit) != 0).view(masks.shape[0], W * 64)[:, :D]

    def rows_pack_count(self, x2d, idx):
        x = x2d if idx is None else x2d[idx.long()]
        D = x.shape[1]
        W, bit = self._bits(D)
        nz = torch.zeros((x.shape[0], W * 64), dtype=torch.bool)
        nz[:, :D] = x != 0
        return (nz.view(-1, W, 64).to(torch.int64) * bit).sum(-1), nz.sum(1).to(torch.int32)

    def rows_mask_count(self, masks, D):
        return self._flags(masks, D).sum(1).to(torch.int32)

    def rows_pack_values(self, x2d, idx, masks, offsets, total):
        x = x2d if idx is None else x2d[idx.long()]
        return x[self._flags(masks, x.shape[1])].contiguous()

    def rows_unpack(self, masks, offsets, values, D, out=None):
        dense = torch.zeros((masks.shape[0], D), dtype=torch.float32)
        dense[self._flags(masks, D)] = values
        if out is None:
            return dense
        out.copy_(dense)
        return out


_ENABLED = False


def enable_cpu_backend(on=True):
    """Route message passing on CPU tensors through libmi355x_graph_cpu.so (on=False: back to raising DGLError).  Returns the
    previous state.  Explicit by design: see the module docstring."""
    global _ENABLED
    from . import sparse
    was = _ENABLED
    if on:
        lib()  # fail now, loudly, if the library was not built
        sparse.register_backend("cpu", CpuBackend())
    else:
        if isinstance(sparse._BACKENDS.get("cpu"), CpuBackend):
            sparse._BACKENDS.pop("cpu", None)
    _ENABLED = bool(on)
    return was


def cpu_backend_enabled():
    return _ENABLED
