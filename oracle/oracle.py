"""ctypes/numpy front-end of the CPU oracle (oracle.c) + numpy restatements of the
integer graph transforms.

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package never imports it.
PARITY UNPINNED against the reference (see oracle.c header): the reference ships
no tests or golden vectors and its arithmetic lives in the un-vendored DGL wheel.

Every function takes / returns numpy arrays (int32 indices, float32 features).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

OPS = {"add": 0, "sub": 1, "mul": 2, "div": 3, "copy_lhs": 4, "copy_rhs": 5, "dot": 6,
       "copy_u": 4, "copy_e": 5}
REDUCES = {"sum": 0, "max": 1, "min": 2, "mean": 3}
TARGETS = {"u": 0, "e": 1, "v": 2}


def build(force=False):
    """Compile oracle.c with gcc (no GPU needed)."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def use_native_build():
    """bench.py's cpu_baseline leg: recompile for the ISA of the host it is timed on (the shipped library targets
    x86-64-v3 so that it loads anywhere).  Falls back to the shipped build when no compiler is available."""
    global _lib
    import tempfile
    out = os.path.join(tempfile.gettempdir(), "liboracle_native_%d.so" % os.getuid())
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c99", "-fno-fast-math", "-ffp-contract=off",
                               "-fopenmp", "-shared", "-o", out, os.path.join(_HERE, "oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _lib = ctypes.CDLL(out)
        _lib.orc_num_threads.restype = ctypes.c_int
        return True
    except (OSError, subprocess.CalledProcessError):
        return False


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int64)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


_c64 = ctypes.c_int64
_ci = ctypes.c_int


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


# --------------------------------------------------------------------------- formats
def coo_to_csr(n_rows, row, col):
    """Stable COO->CSR on `row`; returns (indptr, indices, eids).  In-CSR: row=dst, col=src."""
    row, col = _i32(row), _i32(col)
    nnz = row.shape[0]
    indptr = np.empty(n_rows + 1, np.int32)
    indices = np.empty(nnz, np.int32)
    eids = np.empty(nnz, np.int32)
    lib().orc_coo_to_csr(_c64(n_rows), _c64(nnz), _p(row), _p(col), _p(indptr), _p(indices), _p(eids))
    return indptr, indices, eids


def in_degrees(indptr):
    indptr = _i32(indptr)
    deg = np.empty(indptr.shape[0] - 1, np.int32)
    lib().orc_in_degrees(_c64(deg.shape[0]), _p(indptr), _p(deg))
    return deg


# --------------------------------------------------------------------------- broadcasting
def bcast_offsets(lhs_shape, rhs_shape):
    """numpy-style broadcast of two per-row feature shapes -> (out_shape, lhs_off, rhs_off).

    Offsets are None when the operand is used as-is (no broadcast).  Mirrors DGL's
    CalcBcastOff (UPSTREAM src/array/kernel.cc semantics)."""
    lhs_shape, rhs_shape = tuple(lhs_shape), tuple(rhs_shape)
    nd = max(len(lhs_shape), len(rhs_shape))
    ls = (1,) * (nd - len(lhs_shape)) + lhs_shape
    rs = (1,) * (nd - len(rhs_shape)) + rhs_shape
    out = []
    for a, b in zip(ls, rs):
        if a != b and a != 1 and b != 1:
            raise ValueError("shapes %s and %s do not broadcast" % (lhs_shape, rhs_shape))
        out.append(max(a, b))
    out = tuple(out)
    if ls == rs:
        return out, None, None
    idx = np.indices(out).reshape(nd, -1) if nd else np.zeros((0, 1), np.int64)

    def off(shape):
        strides = np.ones(nd, np.int64)
        for d in range(nd - 2, -1, -1):
            strides[d] = strides[d + 1] * shape[d + 1]
        o = np.zeros(idx.shape[1], np.int64)
        for d in range(nd):
            if shape[d] != 1:
                o += idx[d] * strides[d]
        return o

    return out, off(ls), off(rs)


# --------------------------------------------------------------------------- g-SpMM
def spmm(indptr, indices, eids, op, reduce, U, E, want_arg=False):
    """out[v] = reduce_{(u->v)} op(U[u], E[eid]); mean = sum / clamp(deg,1) (DGL ops/spmm.py)."""
    indptr, indices, eids = _i32(indptr), _i32(indices), _i32(eids)
    n_rows = indptr.shape[0] - 1
    U, E = _f32(U), _f32(E)
    if op == "sub":  # DGL rewrites sub -> add(-rhs), div -> mul(1/rhs) before the kernel
        op, E = "add", -E
    elif op == "div":
        op, E = "mul", (np.float32(1.0) / E).astype(np.float32)
    if op in ("copy_lhs", "copy_u"):
        E = None
    if op in ("copy_rhs", "copy_e"):
        U = None
    ushape = U.shape[1:] if U is not None else ()
    eshape = E.shape[1:] if E is not None else ()
    if U is not None and E is not None:
        oshape, uoff, eoff = bcast_offsets(ushape, eshape)
    else:
        oshape, uoff, eoff = (ushape if U is not None else eshape), None, None
    out_len = int(np.prod(oshape)) if len(oshape) else 1
    u_len = int(np.prod(ushape)) if len(ushape) else 1
    e_len = int(np.prod(eshape)) if len(eshape) else 1
    red = "sum" if reduce == "mean" else reduce
    out = np.empty((n_rows, out_len), np.float32)
    arg_u = np.full((n_rows, out_len), -1, np.int32) if red != "sum" else None
    arg_e = np.full((n_rows, out_len), -1, np.int32) if red != "sum" else None
    lib().orc_spmm(_c64(n_rows), _p(indptr), _p(indices), _p(eids), _ci(OPS[op]), _ci(REDUCES[red]),
                   _p(U), _p(E), _c64(u_len), _c64(e_len), _c64(out_len), _p(uoff), _p(eoff),
                   _p(out), _p(arg_u), _p(arg_e))
    if reduce == "mean":
        deg = np.maximum(np.diff(indptr), 1).astype(np.float32)
        out = out / deg[:, None]
    out = out.reshape((n_rows,) + tuple(oshape))
    if want_arg:
        return out, (None if arg_u is None else arg_u.reshape(out.shape)), \
            (None if arg_e is None else arg_e.reshape(out.shape))
    return out


# --------------------------------------------------------------------------- g-SDDMM
def sddmm(src, dst, op, L, R, lhs_target="u", rhs_target="v"):
    """out[e] = op(L[t_l(e)], R[t_r(e)]) indexed by edge id; dot reduces the last dim to 1."""
    src, dst = _i32(src), _i32(dst)
    nnz = src.shape[0]
    L, R = _f32(L), _f32(R)
    if op in ("copy_lhs", "copy_u"):
        R = None
    if op in ("copy_rhs", "copy_e"):
        L = None
    lshape = L.shape[1:] if L is not None else ()
    rshape = R.shape[1:] if R is not None else ()
    reduce_size = 1
    if op == "dot":
        reduce_size = lshape[-1]
        assert rshape[-1] == reduce_size
        oshape, loff, roff = bcast_offsets(lshape[:-1], rshape[:-1])
        out_len = int(np.prod(oshape)) if len(oshape) else 1
        oshape = tuple(oshape) + (1,)
    elif L is not None and R is not None:
        oshape, loff, roff = bcast_offsets(lshape, rshape)
        out_len = int(np.prod(oshape)) if len(oshape) else 1
    else:
        oshape, loff, roff = (lshape if L is not None else rshape), None, None
        out_len = int(np.prod(oshape)) if len(oshape) else 1
    l_len = int(np.prod(lshape)) if len(lshape) else 1
    r_len = int(np.prod(rshape)) if len(rshape) else 1
    out = np.empty((nnz, out_len), np.float32)
    lib().orc_sddmm(_c64(nnz), _p(src), _p(dst), _ci(OPS[op]), _p(L), _p(R),
                    _ci(TARGETS[lhs_target]), _ci(TARGETS[rhs_target]),
                    _c64(l_len), _c64(r_len), _c64(out_len), _c64(reduce_size),
                    _p(loff), _p(roff), _p(out))
    return out.reshape((nnz,) + tuple(oshape))


# --------------------------------------------------------------------------- edge softmax
def edge_softmax_fwd(indptr, eids, z):
    indptr, eids, z = _i32(indptr), _i32(eids), _f32(z)
    H = int(np.prod(z.shape[1:])) if z.ndim > 1 else 1
    a = np.empty_like(z)
    lib().orc_edge_softmax_fwd(_c64(indptr.shape[0] - 1), _p(indptr), _p(eids), _c64(H), _p(z), _p(a))
    return a


def edge_softmax_bwd(indptr, eids, a, da):
    indptr, eids, a, da = _i32(indptr), _i32(eids), _f32(a), _f32(da)
    H = int(np.prod(a.shape[1:])) if a.ndim > 1 else 1
    dz = np.empty_like(a)
    lib().orc_edge_softmax_bwd(_c64(indptr.shape[0] - 1), _p(indptr), _p(eids), _c64(H), _p(a), _p(da), _p(dz))
    return dz


# --------------------------------------------------------------------------- segment reduce
def segment_reduce(offsets, x, reduce="sum"):
    offsets, x = _i64(offsets), _f32(x)
    n_seg = offsets.shape[0] - 1
    D = int(np.prod(x.shape[1:])) if x.ndim > 1 else 1
    out = np.empty((n_seg,) + x.shape[1:], np.float32)
    lib().orc_segment_reduce(_c64(n_seg), _p(offsets), _c64(D), _ci(REDUCES[reduce]), _p(x), _p(out), None)
    return out


# --------------------------------------------------------------------------- integer transforms (numpy)
def to_bidirected(src, dst, num_nodes):
    """dgl.to_bidirected (main_dgl_arxiv_sage.py:162): union of edges and reverses, duplicates
    removed, result sorted by (src, dst)."""
    s = np.concatenate([src, dst]).astype(np.int64)
    d = np.concatenate([dst, src]).astype(np.int64)
    key = np.unique(s * np.int64(num_nodes) + d)
    return (key // num_nodes), (key % num_nodes)


def add_self_loop(src, dst, num_nodes):
    """dgl.add_self_loop (main_dgl_reddit_gat.py:136): appends (i,i) for every node AFTER the
    existing edges; existing loops are kept."""
    loop = np.arange(num_nodes, dtype=np.asarray(src).dtype)
    return np.concatenate([src, loop]), np.concatenate([dst, loop])


def batch(graphs):
    """dgl.batch (GraphDataLoader, main_dgl_molhiv_gcn.py:163): block-diagonal union.
    graphs: list of (num_nodes, src, dst).  Returns (N, src, dst, batch_num_nodes, batch_num_edges)."""
    off = 0
    ss, dd, bn, be = [], [], [], []
    for n, s, d in graphs:
        ss.append(np.asarray(s, np.int64) + off)
        dd.append(np.asarray(d, np.int64) + off)
        bn.append(n)
        be.append(len(s))
        off += n
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int64)
    return off, cat(ss), cat(dd), np.asarray(bn, np.int64), np.asarray(be, np.int64)
