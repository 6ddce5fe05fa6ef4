"""ctypes binding of the C ABI declared in include/mi355x_graph.h.

The library is the product: if libmi355x_graph.so is missing this module raises at import of the
first op -- there is no CPU or eager-PyTorch fallback for the message-passing arithmetic.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
LIB_PATH = os.environ.get("MGX_LIB_PATH") or os.path.join(CSRC_DIR, "libmi355x_graph.so")  # MGX_LIB_PATH: A/B against another build
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "mi355x_graph.h"))


class DGLError(RuntimeError):
    """Error type of the DGL operator surface (re-exported as dgl.DGLError)."""


class MgxCsr(ctypes.Structure):
    _fields_ = [
        ("num_rows", ctypes.c_int64),
        ("num_cols", ctypes.c_int64),
        ("nnz", ctypes.c_int64),
        ("indptr", ctypes.c_void_p),
        ("indices", ctypes.c_void_p),
        ("eids", ctypes.c_void_p),
        ("idx_bits", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


OP = {"add": 0, "sub": 1, "mul": 2, "div": 3, "copy_lhs": 4, "copy_rhs": 5, "dot": 6}
REDUCE = {"sum": 0, "max": 1, "min": 2, "mean": 3}
TARGET = {"u": 0, "e": 1, "v": 2}

_i32, _i64, _vp, _fp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p
_csr_p = ctypes.POINTER(MgxCsr)

# name -> (restype, argtypes); must list every symbol of include/mi355x_graph.h
SIGNATURES = {
    "mgx_last_error": (ctypes.c_char_p, []),
    "mgx_abi_version": (_i32, []),
    "mgx_last_spmm_kernel": (ctypes.c_char_p, []),
    "mgx_device_info": (_i32, [ctypes.POINTER(_i32), ctypes.POINTER(_i32), ctypes.c_char_p, _i32]),
    "mgx_spmm_plan_workspace": (_i64, [_i64]),
    "mgx_spmm_plan_count": (_i32, [_csr_p, _i64, _vp, _vp, _vp, _i64, _vp]),
    "mgx_spmm_plan_fill": (_i32, [_csr_p, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mgx_spmm_csr": (_i32, [_csr_p, _vp, _i32, _i32, _fp, _fp, _i64, _i64, _i64, _vp, _vp, _fp, _fp, _fp, _vp, _vp,
                            _fp, _i32, _vp]),
    "mgx_row_nonzero_bits": (_i32, [_i64, _i64, _fp, _vp, _vp]),
    "mgx_spmm_copy_u_masked": (_i32, [_csr_p, _vp, _i32, _fp, _i64, _vp, _fp, _fp, _fp, _i32, _vp]),
    "mgx_spmm_copy_u_strided": (_i32, [_csr_p, _vp, _i32, _fp, _i64, _i64, _fp, _fp, _i64, _fp, _i32, _vp]),
    "mgx_edge_tail_fill": (_i32, [_csr_p, _fp, _i64, _fp, _vp]),
    "mgx_spmm_copy_u_edge_tail": (_i32, [_csr_p, _vp, _i32, _fp, _fp, _fp, _fp, _i64, _fp, _i32, _vp]),
    "mgx_rows_slots_pack": (_i32, [_i64, _i64, _fp, _i64, _fp, _vp, _vp, _vp]),
    "mgx_spmm_copy_u_slots": (_i32, [_csr_p, _vp, _i32, _fp, _i64, _i64, _vp, _fp, _fp, _fp, _i64, _fp, _i32, _vp]),
    "mgx_spmm_tile_copy_u": (_i32, [_csr_p, _vp, _vp, _i32, _fp, _i64, _i64, _fp, _fp, _i64, _fp, _i32, _vp]),
    "mgx_rows_mask_words": (_i64, [_i64]),
    "mgx_rows_pack_count": (_i32, [_i64, _vp, _i32, _i64, _fp, _i64, _vp, _vp, _vp]),
    "mgx_rows_mask_count": (_i32, [_i64, _i64, _vp, _vp, _vp]),
    "mgx_rows_pack_values": (_i32, [_i64, _vp, _i32, _i64, _fp, _i64, _vp, _vp, _fp, _vp]),
    "mgx_rows_unpack": (_i32, [_i64, _i64, _vp, _vp, _fp, _fp, _i64, _vp]),
    "mgx_rows_unpack_add_csr": (_i32, [_i64, _vp, _vp, _i64, _vp, _vp, _fp, _fp, _i64, _vp]),
    "mgx_sddmm_coo": (_i32, [_i64, _i64, _i64, _vp, _vp, _i32, _i32, _fp, _fp, _i32, _i32, _i64, _i64, _i64, _i64,
                             _vp, _vp, _fp, _vp]),
    "mgx_sddmm_coo_perm": (_i32, [_i64, _i64, _i64, _vp, _vp, _vp, _i32, _i32, _fp, _fp, _i32, _i32, _i64, _fp, _vp]),
    "mgx_sddmm_csr": (_i32, [_csr_p, _vp, _i32, _fp, _fp, _i32, _i32, _i64, _i64, _i64, _i64, _vp, _vp, _fp, _vp]),
    "mgx_edge_softmax_fwd": (_i32, [_csr_p, _vp, _i64, _fp, _fp, _fp, _vp]),
    "mgx_edge_softmax_bwd": (_i32, [_csr_p, _vp, _i64, _fp, _fp, _fp, _fp, _vp]),
    "mgx_gat_attention_fwd": (_i32, [_csr_p, _vp, _i64, _fp, _fp, ctypes.c_float, _fp, _fp, _vp]),
    "mgx_gat_attention_bwd": (_i32, [_csr_p, _vp, _i64, _fp, _fp, ctypes.c_float, _fp, _fp, _fp, _fp, _vp]),
    "mgx_gat_fused_workspace": (_i64, [_vp, _i64, _i64]),
    "mgx_gat_fused_pack_workspace": (_i64, [_i64, _i64, _i64, _i64]),
    "mgx_gat_fused_fwd": (_i32, [_csr_p, _vp, _i64, _i64, _fp, _fp, _fp, _fp, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, _fp, _fp,
                                 _vp, _vp, _vp]),
    "mgx_gat_fused_bwd": (_i32, [_csr_p, _vp, _csr_p, _vp, _i64, _i64, _fp, _fp, _fp, ctypes.c_float, ctypes.c_float, ctypes.c_uint64,
                                 _fp, _fp, _fp, _fp, _fp, _fp, _vp, _vp, _vp]),
    "mgx_gat_tile_fwd": (_i32, [_csr_p, _vp, _vp, _i64, _i64, _fp, _fp, _fp, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, _fp, _fp,
                                _vp, _vp, _vp]),
    "mgx_gat_tile_bwd": (_i32, [_csr_p, _vp, _vp, _csr_p, _vp, _vp, _i64, _i64, _fp, _fp, ctypes.c_float, ctypes.c_float, ctypes.c_uint64,
                                _fp, _fp, _fp, _fp, _fp, _fp, _vp, _vp, _vp]),
    "mgx_head_dot_fwd": (_i32, [_i64, _i64, _i64, _fp, _fp, _fp, _fp, _fp, _vp]),
    "mgx_head_dot_bwd_workspace": (_i64, [_i64, _i64]),
    "mgx_head_dot_bwd": (_i32, [_i64, _i64, _i64, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _vp, _vp]),
    "mgx_segment_reduce": (_i32, [_i64, _vp, _i64, _i32, _fp, _fp, _vp, _vp]),
    "mgx_relu_dropout_fwd": (_i32, [_i64, _fp, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _fp, _vp, _vp]),
    "mgx_relu_dropout_bwd": (_i32, [_i64, _fp, _vp, ctypes.c_float, _fp, _vp]),
    "mgx_relu_dropout_fwd_strided": (_i32, [_i64, _i64, _fp, _i64, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _fp, _i64, _vp, _vp]),
    "mgx_relu_dropout_bwd_strided": (_i32, [_i64, _i64, _fp, _i64, _vp, ctypes.c_float, _fp, _i64, _vp]),
    "mgx_relu_dropout_bwd_slots": (_i32, [_i64, _fp, _i64, _vp, ctypes.c_float, _fp, _i64, _fp, _vp, _vp, _vp]),
    "mgx_relu_dropout_fwd_counter": (_i32, [_i64, _i64, _fp, _i64, ctypes.c_float, ctypes.c_uint64, _vp, _fp, _i64, _vp, _vp]),
    "mgx_column_pair_sums": (_i32, [_i64, _i64, _i32, _fp, _fp, _fp, _fp, _vp, _vp]),
    "mgx_column_affine": (_i32, [_i64, _i64, _fp, _fp, _fp, _fp, _fp, _fp, _vp]),
    "mgx_xty_workspace": (_i64, [_i64, _i64]),
    "mgx_xty": (_i32, [_i64, _i64, _i64, _fp, _i64, _fp, _i64, _fp, _i64, _vp, _vp]),
    "mgx_xty_colsum": (_i32, [_i64, _i64, _i64, _fp, _i64, _fp, _i64, _fp, _i64, _fp, _vp, _vp]),
    "mgx_rows_gemm": (_i32, [_i64, _i64, _i64, _fp, _i64, _fp, _i64, _i32, _fp, _fp, _i64, _fp, _i64, _fp, _i64, _i64, _vp]),
    "mgx_rows_gemm_supported": (_i32, [_i64, _i64, _i64]),
    "mgx_rows_gemm_relu_dropout": (_i32, [_i64, _i64, _i64, _fp, _i64, _fp, _i64, _i32, _fp, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64,
                                          _fp, _i64, _vp, _vp, _vp, _vp]),
    "mgx_column_sum_workspace": (_i64, [_i64]),
    "mgx_column_sum": (_i32, [_i64, _i64, _fp, _fp, _vp, _vp]),
    "mgx_coo_to_csr_workspace": (_i64, [_i64, _i64, _i32]),
    "mgx_coo_to_csr": (_i32, [_i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mgx_csr_transpose_workspace": (_i64, [_i64, _i64, _i32]),
    "mgx_csr_transpose": (_i32, [_csr_p, _vp, _vp, _vp, _vp, _i64, _vp]),
    "mgx_csr_degrees": (_i32, [_i64, _vp, _i32, _vp, _vp]),
    "mgx_csr_inv_degrees": (_i32, [_i64, _vp, _i32, _fp, _vp]),
    "mgx_coo_to_csr_host": (_i32, [_i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp]),
    "mgx_sample_neighbors": (_i32, [_csr_p, _i64, _vp, _i32, ctypes.c_uint64, _vp, _vp, _vp, _vp]),
    "mgx_gather_rows": (_i32, [_i64, _vp, _i32, _i64, _fp, _fp, _vp]),
    "mgx_gather_rows_strided": (_i32, [_i64, _vp, _i32, _i64, _fp, _i64, _fp, _i64, _vp]),
    "mgx_scatter_add_rows": (_i32, [_i64, _vp, _i32, _i64, _fp, _fp, _vp]),
}

_lib = None


def lib():
    """Load libmi355x_graph.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DGLError(
                "MI355X message-passing library not found at %s -- build it with "
                "`make -C %s` (hipcc --offload-arch=gfx950). There is no CPU fallback." % (LIB_PATH, CSRC_DIR))
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


ERR_UNSUPPORTED = 2  # MGX_ERR_UNSUPPORTED: the entry point has no kernel for these operands; the caller takes its fallback


def check(status):
    if status != 0:
        msg = lib().mgx_last_error()
        raise DGLError("%s (mgx status %d)" % (msg.decode() if msg else "unknown error", status))
