"""GPU suite (-m gpu): the HIP library, called through its C ABI, against the CPU oracle and the
committed golden vectors.  Bar (north_star): index/degree ops bit-exact; fp32 aggregations within
1e-4 relative of the reference CPU algorithm.  All sizes finish in seconds on the oracle."""
import os

import numpy as np
import pytest
import torch

from mi355x_graph import config as mgx_config

import mi355x_graph as mg
from mi355x_graph import ops, sparse
from conftest import random_graph

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-4  # north_star: fp32 aggregations within 1e-4 relative
DEV = "cuda:0"


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-5))) if a.size else 0.0


def mk(n_src, n_dst, src, dst, idtype=torch.int32, formats=None):
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=idtype, device=DEV)
    return g.formats(formats) if formats else g


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_library_is_loaded_and_sees_gfx950():
    import ctypes
    L = mg._lib.lib()
    cus, lds = ctypes.c_int32(), ctypes.c_int32()
    name = ctypes.create_string_buffer(64)
    mg._lib.check(L.mgx_device_info(ctypes.byref(cus), ctypes.byref(lds), name, 64))
    assert name.value.decode().startswith("gfx950"), name.value
    assert cus.value == 256


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("shape", [(1, 1, 0), (7, 5, 3), (1000, 1000, 30000), (5000, 300, 200000), (300, 5000, 20000)])
def test_device_coo_to_csr_and_degrees_bit_exact(oracle, idtype, shape):
    n_src, n_dst, nnz = shape
    src, dst = random_graph(n_src, n_dst, nnz, seed=nnz + 1)
    g = mk(n_src, n_dst, src, dst, idtype)
    csc, csr = g._index.csc(), g._index.csr()
    ip, ix, ei = oracle.coo_to_csr(n_dst, dst, src)
    assert np.array_equal(csc.indptr.cpu().numpy(), ip)
    assert np.array_equal(csc.indices.cpu().numpy(), ix)
    assert np.array_equal(csc.eids.cpu().numpy(), ei)
    rp, rx, re = oracle.coo_to_csr(n_src, src, dst)
    assert np.array_equal(csr.indptr.cpu().numpy(), rp) and np.array_equal(csr.indices.cpu().numpy(), rx)
    assert np.array_equal(csr.eids.cpu().numpy(), re)
    deg = g.in_degrees()
    assert deg.dtype == idtype and np.array_equal(deg.cpu().numpy(), oracle.in_degrees(ip))
    assert np.array_equal(g.out_degrees().cpu().numpy(), np.diff(rp))
    # host-built CSR moved to the device gives the same arrays
    gh = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=idtype)
    gh = gh.formats(["csr", "csc"]).to(DEV)
    assert torch.equal(gh._index.csc().indices, csc.indices) and torch.equal(gh._index.csc().eids, csc.eids)


@pytest.mark.parametrize("name", ["tiny", "g200", "bip", "cora_like"])
def test_golden_vectors(name):
    g = dict(np.load(os.path.join(GOLD, name + ".npz")))
    if name == "tiny":
        gr = mk(5, 5, g["src"], g["dst"])
        out = ops.gspmm(gr, "copy_lhs", "sum", T(g["X"]), None)
        assert np.array_equal(out.cpu().numpy(), g["copy_u_sum"])  # small integers: exact in any order
        assert np.array_equal(gr.in_degrees().cpu().numpy(), g["in_degrees"])
        return
    n_src, n_dst = int(g["n_src"]), int(g["n_dst"])
    gr = mk(n_src, n_dst, g["src"], g["dst"])
    X, V = T(g["X"]), T(g["V"])
    assert rel(ops.gspmm(gr, "copy_lhs", "sum", X, None).cpu(), g["copy_u_sum"]) < RTOL
    assert rel(ops.gspmm(gr, "copy_lhs", "mean", X, None).cpu(), g["copy_u_mean"]) < RTOL
    H = g["W"].shape[1]
    assert rel(ops.gspmm(gr, "mul", "sum", X.view(n_src, H, -1), T(g["W"])).cpu(), g["u_mul_e_sum_f64"]) < RTOL
    assert np.array_equal(ops.gspmm(gr, "copy_lhs", "max", X, None).cpu().numpy(), g["copy_u_max"])
    assert np.array_equal(ops.gsddmm(gr, "add", X, V).cpu().numpy(), g["u_add_v"])
    assert np.array_equal(ops.gsddmm(gr, "mul", X, V).cpu().numpy(), g["u_mul_v"])
    assert rel(ops.gsddmm(gr, "dot", X, V).cpu(), g["u_dot_v_f64"]) < RTOL
    a = ops.edge_softmax(gr, T(g["Z"]))
    assert rel(a.cpu(), g["edge_softmax_f64"]) < RTOL
    dz = sparse.edge_softmax_bwd_raw(gr._index.csc(), T(g["edge_softmax_f64"].astype(np.float32)), T(g["dA"]))
    assert float(np.max(np.abs(dz.cpu().numpy() - g["edge_softmax_bwd_f64"]))) < 1e-5


SPMM_SHAPES = [
    # (n_src, n_dst, nnz, D)  -- D sweeps kernel/dgl-new.py:13 (1..128) plus odd / wide rows
    (500, 500, 6000, 1), (500, 500, 6000, 2), (500, 500, 6000, 4), (500, 500, 6000, 8),
    (500, 500, 6000, 16), (2000, 2000, 60000, 32), (2000, 2000, 100000, 64), (2000, 2000, 30000, 100),
    (2000, 2000, 30000, 128), (1500, 700, 20000, 256), (300, 300, 2000, 602), (200, 200, 1500, 1433),
    (400, 400, 3000, 7), (400, 400, 3000, 6), (50, 5000, 4000, 64), (3000, 40, 150000, 64),
    # column passes of 32 lanes (132 = 32 + 1 lanes, 320 = 32 + 32 + 16), wide odd widths on ragged 16-byte lanes
    (600, 600, 20000, 132), (600, 600, 20000, 320), (500, 500, 9000, 300), (500, 500, 9000, 150), (400, 400, 6000, 514),
    (300, 300, 5000, 1023), (3000, 40, 90000, 602),
]


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("shape", SPMM_SHAPES)
def test_copy_u_sum_and_mean(oracle, shape, idtype):
    n_src, n_dst, nnz, D = shape
    src, dst = random_graph(n_src, n_dst, nnz, seed=D)
    rng = np.random.default_rng(D)
    X = rng.random((n_src, D), dtype=np.float32)  # U[0,1) as kernel/dgl-new.py:15
    g = mk(n_src, n_dst, src, dst, idtype)
    ip, ix, ei = oracle.coo_to_csr(n_dst, dst, src)
    for red in ("sum", "mean"):
        out = ops.gspmm(g, "copy_lhs", red, T(X), None)
        assert out.shape == (n_dst, D)
        assert rel(out.cpu(), oracle.spmm(ip, ix, ei, "copy_lhs", red, X, None)) < RTOL
    # determinism: no atomics => bitwise identical reruns
    assert torch.equal(ops.gspmm(g, "copy_lhs", "sum", T(X), None), ops.gspmm(g, "copy_lhs", "sum", T(X), None))


@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "copy_lhs", "copy_rhs"])
@pytest.mark.parametrize("red", ["sum", "max", "min", "mean"])
def test_gspmm_all_ops_all_reducers(oracle, op, red):
    """kernel/dgl-new.py:50-51 lets any op x reduce through; efeat has the node feature's width."""
    n_src, n_dst, nnz, D = 700, 600, 9000, 12
    src, dst = random_graph(n_src, n_dst, nnz, seed=17)
    rng = np.random.default_rng(17)
    X = rng.random((n_src, D), dtype=np.float32)
    E = rng.random((nnz, D), dtype=np.float32) + 0.5
    g = mk(n_src, n_dst, src, dst)
    ip, ix, ei = oracle.coo_to_csr(n_dst, dst, src)
    out = ops.gspmm(g, op, red, T(X), T(E))
    ref = oracle.spmm(ip, ix, ei, op, red, X, E)
    assert rel(out.cpu(), ref) < RTOL


@pytest.mark.parametrize("H,F", [(1, 16), (4, 8), (8, 16), (3, 5), (8, 1), (1, 1)])
def test_u_mul_e_head_broadcast(oracle, H, F):
    n, nnz = 900, 15000
    src, dst = random_graph(n, n, nnz, seed=H * 10 + F)
    rng = np.random.default_rng(H)
    X = rng.random((n, H, F), dtype=np.float32)
    W = rng.random((nnz, H, 1), dtype=np.float32)
    g = mk(n, n, src, dst)
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    out = ops.gspmm(g, "mul", "sum", T(X), T(W))
    assert out.shape == (n, H, F)
    assert rel(out.cpu(), oracle.spmm(ip, ix, ei, "mul", "sum", X, W)) < RTOL
    # (N,D) x (E,1) of main_dgl_proteins_rgcn_for.py:52
    X2, W2 = X.reshape(n, H * F), W[:, 0, :]
    out = ops.gspmm(g, "mul", "mean", T(X2), T(W2))
    assert rel(out.cpu(), oracle.spmm(ip, ix, ei, "mul", "mean", X2, W2)) < RTOL


def test_general_broadcast_tables(oracle):
    n, nnz = 300, 4000
    src, dst = random_graph(n, n, nnz, seed=5)
    rng = np.random.default_rng(5)
    X = rng.random((n, 1, 6), dtype=np.float32)
    E = rng.random((nnz, 4, 1), dtype=np.float32)
    g = mk(n, n, src, dst)
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    for op in ("add", "mul"):
        out = ops.gspmm(g, op, "sum", T(X), T(E))
        assert out.shape == (n, 4, 6)
        assert rel(out.cpu(), oracle.spmm(ip, ix, ei, op, "sum", X, E)) < RTOL
    out = ops.gsddmm(g, "mul", T(X), T(rng.random((n, 4, 1), dtype=np.float32)))
    assert out.shape == (nnz, 4, 6)


def test_max_min_arg_indices_exact(oracle):
    n_src, n_dst, nnz, D = 400, 500, 5000, 9
    src, dst = random_graph(n_src, n_dst, nnz, seed=23)
    rng = np.random.default_rng(23)
    X = rng.random((n_src, D), dtype=np.float32)
    g = mk(n_src, n_dst, src, dst)
    csc = g._index.csc()
    ip, ix, ei = oracle.coo_to_csr(n_dst, dst, src)
    for red in ("max", "min"):
        out, au, ae = sparse.gspmm_raw(csc, "copy_lhs", red, T(X), None, want_arg=True)
        ref, ru, _ = oracle.spmm(ip, ix, ei, "copy_lhs", red, X, None, want_arg=True)
        assert np.array_equal(out.cpu().numpy(), ref)
        assert np.array_equal(au.cpu().numpy(), ru)


SDDMM_SHAPES = [(600, 600, 8000, 1), (600, 600, 8000, 2), (600, 600, 8000, 16), (600, 500, 8000, 64),
                (300, 400, 5000, 128), (300, 300, 3000, 7), (200, 200, 1000, 300)]


@pytest.mark.parametrize("fmt", ["coo", "csr_csc"])
@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "dot", "copy_lhs", "copy_rhs"])
@pytest.mark.parametrize("shape", SDDMM_SHAPES)
def test_gsddmm(oracle, shape, op, fmt):
    n_src, n_dst, nnz, D = shape
    src, dst = random_graph(n_src, n_dst, nnz, seed=D + 3)
    rng = np.random.default_rng(D)
    U = rng.random((n_src, D), dtype=np.float32)
    V = rng.random((n_dst, D), dtype=np.float32) + 0.5
    g = mk(n_src, n_dst, src, dst, formats=["csr", "csc"] if fmt == "csr_csc" else None)
    out = ops.gsddmm(g, op, T(U), T(V))
    ref = oracle.sddmm(src, dst, op, U, V)
    assert out.shape == ref.shape
    if op == "dot":
        assert rel(out.cpu(), ref) < RTOL
    else:
        assert np.array_equal(out.cpu().numpy(), ref)  # one rounding per element: bit-exact


@pytest.mark.parametrize("op,targets", [("add", "uv"), ("sub", "vu"), ("mul", "uu"), ("div", "vv"), ("copy_lhs", "uv"), ("copy_rhs", "uv"),
                                        ("copy_lhs", "vu")])
@pytest.mark.parametrize("shape", [(300, 400, 5000, 64), (300, 300, 4097, 4), (257, 513, 9000, 100), (64, 64, 600, 256)])
def test_gsddmm_walked_in_csr_order(oracle, shape, op, targets, monkeypatch):
    """Round 5: a graph that holds its COO AND an in-CSR -- the lean kernel over the CSR's order, whole output rows
    scattered by edge id (mgx_sddmm_coo_perm) -- against the oracle and against the walk in edge-id order, bit for bit."""
    from mi355x_graph import sparse
    n_src, n_dst, nnz, D = shape
    src, dst = random_graph(n_src, n_dst, nnz, seed=D + 11)
    rng = np.random.default_rng(D + 1)
    feats = {"u": rng.random((n_src, D), dtype=np.float32) + 0.25, "v": rng.random((n_dst, D), dtype=np.float32) + 0.5}
    Lh, Rh = feats[targets[0]], feats[targets[1]]
    g = mk(n_src, n_dst, src, dst)
    g._index.csc()   # materialised beside the edge list, as after any update_all() on the graph
    assert g._index.has_format("coo") and g._index.has_format("csc")
    monkeypatch.setenv("MGX_SDDMM_WALK", "csr")
    sparse.PROFILE = []
    try:
        out = ops.gsddmm(g, op, T(Lh), T(Rh), targets[0], targets[1])
    finally:
        recs, sparse.PROFILE = sparse.PROFILE, None
    assert [r.get("walk") for r in recs if r.get("kernel") == "sddmm"] == ["csr order"]   # the permuted walk really ran
    monkeypatch.setenv("MGX_SDDMM_WALK", "coo")
    plain = ops.gsddmm(g, op, T(Lh), T(Rh), targets[0], targets[1])
    assert torch.equal(out, plain)
    assert np.array_equal(out.cpu().numpy(), oracle.sddmm(src, dst, op, Lh, Rh, targets[0], targets[1]))
    # default (auto): a small graph keeps the edge list's own order
    monkeypatch.delenv("MGX_SDDMM_WALK")
    assert not sparse.HipBackend._sddmm_in_csr_order(g._index, op, T(Lh), T(Rh), targets[0], targets[1], D, D, D, None, None)


def test_gsddmm_targets_and_heads(oracle):
    n, nnz, H, F = 500, 7000, 4, 8
    src, dst = random_graph(n, n, nnz, seed=9)
    rng = np.random.default_rng(9)
    U = rng.random((n, H, F), dtype=np.float32)
    V = rng.random((n, H, F), dtype=np.float32)
    Ee = rng.random((nnz, H, 1), dtype=np.float32)
    g = mk(n, n, src, dst)
    assert rel(ops.gsddmm(g, "dot", T(U), T(V)).cpu(), oracle.sddmm(src, dst, "dot", U, V)) < RTOL
    out = ops.gsddmm(g, "mul", T(Ee), T(V), "e", "v")
    assert np.array_equal(out.cpu().numpy(), oracle.sddmm(src, dst, "mul", Ee, V, "e", "v"))
    out = ops.gsddmm(g, "sub", T(V), T(U), "v", "u")
    assert np.array_equal(out.cpu().numpy(), oracle.sddmm(src, dst, "sub", V, U, "v", "u"))
    el, er = rng.random((n, H, 1), dtype=np.float32), rng.random((n, H, 1), dtype=np.float32)
    out = ops.gsddmm(g, "add", T(el), T(er))  # GATConv's u_add_v
    assert np.array_equal(out.cpu().numpy(), oracle.sddmm(src, dst, "add", el, er))


@pytest.mark.parametrize("H", [1, 2, 3, 4, 8, 16, 64])
def test_edge_softmax_fwd_bwd(oracle, H):
    n, nnz = 800, 30000
    src, dst = random_graph(n, n, nnz, seed=H)
    rng = np.random.default_rng(H)
    Z = (rng.standard_normal((nnz, H)) * 4).astype(np.float32)
    dA = rng.standard_normal((nnz, H)).astype(np.float32)
    g = mk(n, n, src, dst)
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    z = T(Z).requires_grad_(True)
    a = ops.edge_softmax(g, z)
    ref = oracle.edge_softmax_fwd(ip, ei, Z)
    assert rel(a.detach().cpu(), ref) < RTOL
    a.backward(T(dA))
    ref_dz = oracle.edge_softmax_bwd(ip, ei, ref, dA)
    assert float(np.max(np.abs(z.grad.cpu().numpy() - ref_dz))) < 1e-5
    # (E,H,1) logits as GATConv passes them
    a3 = ops.edge_softmax(g, T(Z).view(nnz, H, 1))
    assert torch.equal(a3.view(nnz, H), a.detach())


@pytest.mark.parametrize("red", ["sum", "mean", "max", "min"])
def test_segment_reduce(oracle, red):
    rng = np.random.default_rng(1)
    lens = rng.integers(0, 60, size=300)
    lens[5] = 0
    x = rng.random((int(lens.sum()), 256), dtype=np.float32)
    off = np.concatenate([[0], np.cumsum(lens)])
    out = ops.segment_reduce(T(lens), T(x), red)
    assert rel(out.cpu(), oracle.segment_reduce(off, x, red)) < RTOL


def test_edge_cases(oracle):
    # empty graph, graph without edges, single hub row longer than any cache
    g = mk(4, 4, np.zeros(0, np.int64), np.zeros(0, np.int64))
    out = ops.gspmm(g, "copy_lhs", "sum", torch.rand(4, 8, device=DEV), None)
    assert out.shape == (4, 8) and float(out.abs().sum()) == 0.0
    assert ops.gsddmm(g, "add", torch.rand(4, 8, device=DEV), torch.rand(4, 8, device=DEV)).shape == (0, 8)
    n, nnz = 50, 40000
    rng = np.random.default_rng(2)
    src = rng.integers(0, n, nnz)
    dst = np.zeros(nnz, np.int64)  # one hub
    X = rng.random((n, 64), dtype=np.float32)
    g = mk(n, n, src, dst)
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    # a 40k-edge fp32 row: the reference CPU algorithm's own sequential rounding error is ~1.3e-4 here,
    # so both are held to the exact (fp64) sum; the HIP path must be at least as accurate as the oracle
    exact = np.zeros((n, 64))
    np.add.at(exact, dst, X[src].astype(np.float64))
    hip = ops.gspmm(g, "copy_lhs", "sum", T(X), None).cpu()
    assert rel(hip, exact) < RTOL
    assert rel(hip, exact) <= rel(oracle.spmm(ip, ix, ei, "copy_lhs", "sum", X, None), exact) + 1e-6
    Z = rng.standard_normal((nnz, 2)).astype(np.float32)
    assert rel(ops.edge_softmax(g, T(Z)).cpu(), oracle.edge_softmax_fwd(ip, ei, Z)) < RTOL
    with pytest.raises(mg.DGLError):
        ops.gspmm(g, "copy_lhs", "sum", torch.rand(n + 1, 4, device=DEV), None)
    with pytest.raises(mg.DGLError):
        ops.gspmm(g, "pow", "sum", T(X), None)
    with pytest.raises(mg.DGLError):
        ops.gspmm(g, "copy_lhs", "sum", T(X).double(), None)


def test_autograd_against_dense_torch():
    """Backward formulas vs torch autograd on the gathered/dense formulation."""
    n, nnz, H, F = 120, 1500, 2, 4
    src, dst = random_graph(n, n, nnz, seed=31)
    g = mk(n, n, src, dst)
    s, d = T(src), T(dst)
    torch.manual_seed(0)

    def check(fn_ours, fn_ref, *shapes):
        xs = [torch.rand(*sh, device=DEV, requires_grad=True) for sh in shapes]
        ys = [x.detach().clone().requires_grad_(True) for x in xs]
        a, b = fn_ours(*xs), fn_ref(*ys)
        assert rel(a.detach().cpu(), b.detach().cpu()) < RTOL
        w = torch.rand_like(a)
        (a * w).sum().backward()
        (b * w).sum().backward()
        for x, y in zip(xs, ys):  # gradients cancel towards 0: compare against the tensor's scale
            gx, gy = x.grad.cpu().double(), y.grad.cpu().double()
            assert float((gx - gy).abs().max() / gy.abs().max().clamp(min=1e-12)) < RTOL

    def agg(msg):
        out = torch.zeros((n,) + msg.shape[1:], device=DEV)
        return out.index_add(0, d, msg)

    deg = torch.bincount(d, minlength=n).clamp(min=1).float()
    check(lambda x: ops.gspmm(g, "copy_lhs", "sum", x, None), lambda x: agg(x[s]), (n, 8))
    check(lambda x: ops.gspmm(g, "copy_lhs", "mean", x, None), lambda x: agg(x[s]) / deg[:, None], (n, 8))
    check(lambda x, w: ops.gspmm(g, "mul", "sum", x, w), lambda x, w: agg(x[s] * w), (n, H, F), (nnz, H, 1))
    check(lambda x, w: ops.gspmm(g, "mul", "mean", x, w), lambda x, w: agg(x[s] * w) / deg[:, None], (n, 6), (nnz, 1))
    check(lambda x, w: ops.gspmm(g, "add", "sum", x, w), lambda x, w: agg(x[s] + w), (n, 6), (nnz, 6))
    check(lambda w: ops.gspmm(g, "copy_rhs", "sum", None, w), lambda w: agg(w), (nnz, 5))
    check(lambda x, y: ops.gsddmm(g, "add", x, y), lambda x, y: x[s] + y[d], (n, H, 1), (n, H, 1))
    check(lambda x, y: ops.gsddmm(g, "dot", x, y), lambda x, y: (x[s] * y[d]).sum(-1, keepdim=True), (n, H, F), (n, H, F))
    check(lambda x, y: ops.gsddmm(g, "mul", x, y), lambda x, y: x[s] * y[d], (n, 7), (n, 7))
    check(lambda x, y: ops.gsddmm(g, "sub", x, y), lambda x, y: x[s] - y[d], (n, 7), (n, 7))
    check(lambda x, y: ops.gsddmm(g, "div", x, y + 1), lambda x, y: x[s] / (y[d] + 1), (n, 3), (n, 3))
    check(lambda x: ops.gsddmm(g, "copy_lhs", x, None), lambda x: x[s], (n, 16))

    def ref_max(x):
        return torch.zeros(n, 8, device=DEV).scatter_reduce(0, d[:, None].expand(-1, 8), x[s], "amax", include_self=False)
    check(lambda x: ops.gspmm(g, "copy_lhs", "max", x, None), ref_max, (n, 8))

    def ref_softmax(z):
        m = torch.full((n, H), -1e30, device=DEV).scatter_reduce(0, d[:, None].expand(-1, H), z, "amax")
        e = torch.exp(z - m[d])
        return e / agg(e)[d]
    check(lambda z: ops.edge_softmax(g, z), ref_softmax, (nnz, H))


def test_canonical_edge_order_is_equivalent(oracle):
    """GraphIndex.canonical(): edges renumbered in in-CSR order; every op must give the same result as on
    the original graph after permuting edge tensors (perm[new] = old)."""
    n, nnz, H, F = 700, 20000, 4, 8
    src, dst = random_graph(n, n, nnz, seed=77)
    g = mk(n, n, src, dst)
    cidx, perm = g._index.canonical()
    assert perm is not None and cidx.csc().eids is None
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    assert np.array_equal(perm.cpu().numpy(), ei)
    rng = np.random.default_rng(7)
    X = T(rng.random((n, H, F), dtype=np.float32))
    el, er = T(rng.random((n, H, 1), dtype=np.float32)), T(rng.random((n, H, 1), dtype=np.float32))
    W = T(rng.random((nnz, H, 1), dtype=np.float32))
    p = perm.long()
    e0 = ops.gsddmm(g, "add", el, er)
    e1 = ops.gsddmm(cidx, "add", el, er)
    assert torch.equal(e1, e0[p])
    assert torch.equal(ops.edge_softmax(cidx, e1), ops.edge_softmax(g, e0)[p])
    assert rel(ops.gspmm(cidx, "mul", "sum", X, W[p]).cpu(), ops.gspmm(g, "mul", "sum", X, W).cpu()) < 1e-6
    assert rel(ops.gsddmm(cidx, "dot", X, X).cpu(), ops.gsddmm(g, "dot", X, X)[p].cpu()) < 1e-6
    # backward through the canonical graph (reverse CSR carries the renumbered ids)
    w1 = W[p].clone().requires_grad_(True)
    w0 = W.clone().requires_grad_(True)
    x1, x0 = X.clone().requires_grad_(True), X.clone().requires_grad_(True)
    ops.gspmm(cidx, "mul", "sum", x1, w1).pow(2).sum().backward()
    ops.gspmm(g, "mul", "sum", x0, w0).pow(2).sum().backward()
    assert rel(x1.grad.cpu(), x0.grad.cpu()) < 1e-5 and rel(w1.grad.cpu(), w0.grad[p].cpu()) < 1e-5
    # GATConv returns attention in the caller's edge-id order
    from mi355x_graph.nn import GATConv
    torch.manual_seed(0)
    conv = GATConv(16, 8, 2, allow_zero_in_degree=True).to(DEV)
    h = torch.rand(n, 16, device=DEV)
    out, att = conv(g, h, get_attention=True)
    s, d = torch.from_numpy(src).to(DEV), torch.from_numpy(dst).to(DEV)
    sums = torch.zeros(n, 2, 1, device=DEV).index_add(0, d, att)
    has = torch.bincount(d, minlength=n) > 0
    assert torch.allclose(sums[has], torch.ones_like(sums[has]), atol=1e-5)
    ref = torch.zeros(n, 2, 8, device=DEV).index_add(0, d, conv.fc(h).view(n, 2, 8)[s] * att) + conv.bias.view(1, 2, 8)
    assert float((out - ref).abs().max() / ref.abs().max()) < RTOL  # signed sums cancel: compare at the tensor's scale


def test_spmm_accumulate_and_scales(oracle):
    """MGX_SPMM_ACCUMULATE + src/dst scales (the multi-GPU overlapped path and the fused mean backward)."""
    n_src, n_dst, D = 900, 700, 64
    rng = np.random.default_rng(3)
    s1, d1 = random_graph(n_src, n_dst, 30000, seed=31)
    s2, d2 = random_graph(400, n_dst, 9000, seed=32)
    X, Y = rng.random((n_src, D), dtype=np.float32), rng.random((400, D), dtype=np.float32)
    ds = rng.random(n_dst).astype(np.float32) + 0.5
    ss = rng.random(n_src).astype(np.float32) + 0.5
    g1, g2 = mk(n_src, n_dst, s1, d1), mk(400, n_dst, s2, d2)
    out, _, _ = sparse.gspmm_raw(g1._index.csc(), "copy_lhs", "sum", T(X), None, src_scale=T(ss), dst_scale=T(ds))
    ret, _, _ = sparse.gspmm_raw(g2._index.csc(), "copy_lhs", "sum", T(Y), None, dst_scale=T(ds), accumulate_into=out)
    assert ret.data_ptr() == out.data_ptr()
    i1 = oracle.coo_to_csr(n_dst, d1, s1)
    i2 = oracle.coo_to_csr(n_dst, d2, s2)
    ref = (oracle.spmm(*i1, "copy_lhs", "sum", X * ss[:, None], None) + oracle.spmm(*i2, "copy_lhs", "sum", Y, None)) * ds[:, None]
    assert rel(out.cpu(), ref) < RTOL


def test_c_abi_from_plain_c(tmp_path):
    """The boundary is a C ABI: a C program (no Python, no torch) builds against include/mi355x_graph.h, links
    libmi355x_graph.so and gets the hand-computed answers."""
    import shutil
    import subprocess
    from mi355x_graph import _lib
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("gcc / ROCm headers not available")
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call([gcc, "-std=c99", "-D__HIP_PLATFORM_AMD__", os.path.join(here, "abi_c", "abi_smoke.c"), "-o", exe,
                           "-I", os.path.dirname(_lib.HEADER_PATH), "-I", "/opt/rocm/include", "-L", _lib.CSRC_DIR,
                           "-lmi355x_graph", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + _lib.CSRC_DIR, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "abi_smoke ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
def test_device_plan_builder_bit_exact(idtype, monkeypatch):
    """mgx_spmm_plan_count/_fill (csrc/plan.hip) against the torch formulation of schedule.build_plan."""
    from mi355x_graph import schedule
    n, nnz = 5000, 300000
    src, dst = random_graph(n, n, nnz, seed=99)
    csc = mk(n, n, src, dst, idtype)._index.csc()
    order = torch.randperm(n, device=DEV)
    for ordr in (None, order):
        for split in (64, 256):
            monkeypatch.setattr(mgx_config, "PLAN_BUILDER", "device")
            a = schedule.build_plan(csc, ordr, split)
            monkeypatch.setattr(mgx_config, "PLAN_BUILDER", "torch")
            b = schedule.build_plan(csc, ordr, split)
            assert a.num_items == b.num_items and a.num_hubs == b.num_hubs and a.num_slots == b.num_slots
            assert a.num_hubs > 0
            for x, y in ((a.item_row, b.item_row), (a.item_beg, b.item_beg), (a.item_end, b.item_end),
                         (a.item_node, b.item_node), (a.hub_row, b.hub_row), (a.hub_slot_ptr, b.hub_slot_ptr),
                         (a.slot_item, b.slot_item)):
                assert torch.equal(x, y)


@pytest.mark.parametrize("H", [1, 3, 8])
@pytest.mark.parametrize("canonical", [False, True])
def test_fused_gat_attention_matches_composition(oracle, H, canonical):
    """mgx_gat_attention_fwd/bwd == apply_edges(u_add_v) -> leaky_relu -> edge_softmax (and its autograd), also on hub
    rows (chunked path) and in canonical edge order."""
    import torch.nn.functional as F
    n, nnz = 900, 60000
    src, dst = random_graph(n, n, nnz, seed=H + 40)
    g = mk(n, n, src, dst)
    gi = g._index.canonical()[0] if canonical else g._index
    rng = np.random.default_rng(H)
    el0 = T((rng.standard_normal((n, H, 1)) * 2).astype(np.float32))
    er0 = T((rng.standard_normal((n, H, 1)) * 2).astype(np.float32))
    w = torch.rand(nnz, H, 1, device=DEV)
    el1, er1 = el0.clone().requires_grad_(True), er0.clone().requires_grad_(True)
    el2, er2 = el0.clone().requires_grad_(True), er0.clone().requires_grad_(True)
    a_f = ops.gat_attention(gi, el1, er1, 0.2)
    a_u = ops.edge_softmax(gi, F.leaky_relu(ops.gsddmm(gi, "add", el2, er2, "u", "v"), 0.2))
    assert a_f.shape == (nnz, H, 1)
    assert float((a_f - a_u).abs().max()) < 1e-6
    (a_f * w).sum().backward()
    (a_u * w).sum().backward()
    for x, y in ((el1.grad, el2.grad), (er1.grad, er2.grad)):
        assert float((x - y).abs().max() / y.abs().max().clamp(min=1e-12)) < 1e-4
    if not canonical:  # against the CPU oracle
        ip, ix, ei = oracle.coo_to_csr(n, dst, src)
        z = oracle.sddmm(src, dst, "add", el0.cpu().numpy(), er0.cpu().numpy())
        z = np.where(z > 0, z, 0.2 * z).astype(np.float32)
        ref = oracle.edge_softmax_fwd(ip, ei, z.reshape(nnz, H))
        assert rel(a_f.detach().cpu().view(nnz, H), ref) < RTOL


@pytest.mark.parametrize("n,H,F", [(1, 1, 4), (777, 1, 16), (5000, 8, 16), (3001, 4, 32), (2000, 2, 128), (300, 1, 256), (64, 8, 4), (4000, 1, 41), (500, 3, 7), (100, 200, 1)])
@pytest.mark.parametrize("both", [False, True])
def test_head_dot_matches_fp64(n, H, F, both):
    """mgx_head_dot_fwd/bwd == (feat * attn).sum(-1) and its autograd (fp64 reference), one or two attention vectors."""
    rng = np.random.default_rng(n + H + F)
    x0 = T(rng.standard_normal((n, H, F)).astype(np.float32))
    a0 = T(rng.standard_normal((1, H, F)).astype(np.float32))
    b0 = T(rng.standard_normal((1, H, F)).astype(np.float32))
    wa, wb = torch.rand(n, H, device=DEV), torch.rand(n, H, device=DEV)
    x1, a1, b1 = (t.clone().requires_grad_(True) for t in (x0, a0, b0))
    x2, a2, b2 = (t.double().clone().requires_grad_(True) for t in (x0, a0, b0))
    assert ops.head_dot_supported(x1)
    if both:
        el, er = ops.head_dot(x1, a1, b1)
        ((el * wa).sum() + (er * wb).sum()).backward()
        rl, rr = (x2 * a2).sum(-1), (x2 * b2).sum(-1)
        ((rl * wa.double()).sum() + (rr * wb.double()).sum()).backward()
        pairs = [(el, rl), (er, rr), (x1.grad, x2.grad), (a1.grad, a2.grad), (b1.grad, b2.grad)]
    else:
        el = ops.head_dot(x1, a1)
        (el * wa).sum().backward()
        rl = (x2 * a2).sum(-1)
        (rl * wa.double()).sum().backward()
        pairs = [(el, rl), (x1.grad, x2.grad), (a1.grad, a2.grad)]
    for got, ref in pairs:
        assert got.shape == ref.shape
        assert float((got.double() - ref).abs().max() / ref.abs().max().clamp(min=1e-12)) < 1e-5


def test_head_dot_rejects_unsupported_shapes():
    x = torch.zeros(10, 1, 100, device=DEV)
    assert not ops.head_dot_supported(x)
    with pytest.raises(mg.DGLError, match="mgx_head_dot"):
        ops.head_dot(x, torch.zeros(1, 1, 100, device=DEV))


@pytest.mark.parametrize("n,C", [(0, 5), (1, 1), (1000, 47), (100003, 47), (70001, 64), (5000, 256), (3, 200), (40000, 7)])
def test_column_sum_and_linear_bias_grad(n, C):
    """mgx_column_sum == x.sum(0) (fp64 reference); nn.Linear / bias_add built on it give autograd's gradients."""
    rng = np.random.default_rng(n + C)
    x = T(rng.standard_normal((n, C)).astype(np.float32))
    got = sparse.backend_for(x).column_sum(x)
    ref = x.double().sum(0)
    assert got.shape == (C,)
    assert float((got.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(x.abs().sum(0).max())) if n else float(got.abs().max()) == 0
    if n == 0:
        return
    from mi355x_graph.nn import Linear
    lin = Linear(13, C).to(DEV)
    ref_lin = torch.nn.Linear(13, C).to(DEV)
    ref_lin.load_state_dict(lin.state_dict())
    inp = torch.rand(n, 13, device=DEV)
    i1, i2 = inp.clone().requires_grad_(True), inp.clone().requires_grad_(True)
    w = torch.rand(n, C, device=DEV)
    y1, y2 = lin(i1), ref_lin(i2)
    assert torch.equal(y1, y2)
    (y1 * w).sum().backward()
    (y2 * w).sum().backward()
    for a, b in ((i1.grad, i2.grad), (lin.weight.grad, ref_lin.weight.grad), (lin.bias.grad, ref_lin.bias.grad)):
        assert float((a - b).abs().max() / b.abs().max().clamp(min=1e-12)) < 1e-4
    b1 = torch.zeros(C, device=DEV, requires_grad=True)
    b2 = torch.zeros(C, device=DEV, requires_grad=True)
    (ops.bias_add(x, b1) * w).sum().backward()
    ((x + b2) * w).sum().backward()
    assert float((b1.grad - b2.grad).abs().max() / b2.grad.abs().max().clamp(min=1e-12)) < 1e-4


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("shape", [(1, 1, 0), (7, 5, 3), (1000, 1000, 30000), (5000, 300, 200000), (300, 5000, 20000)])
def test_csr_transpose_bit_exact(oracle, idtype, shape):
    """mgx_csr_transpose: the out-CSR from the in-CSR without a COO round trip, bit-identical to the COO-built one (oracle:
    stable sort of the edge list by source).  Transposing twice restores the input, and a graph that holds only one
    compressed format computes the same aggregation as one built from COO."""
    n_src, n_dst, nnz = shape
    src, dst = random_graph(n_src, n_dst, nnz, seed=nnz + 5)
    g = mk(n_src, n_dst, src, dst, idtype)
    csc = g._index.csc()                                   # rows = dst, indices = src, eids -> edge id
    t = sparse.csr_transpose(csc)                          # rows = src, indices = dst
    rp, rx, re = oracle.coo_to_csr(n_src, src, dst)
    assert t.num_rows == n_src and t.num_cols == n_dst and t.indptr.dtype == idtype
    assert np.array_equal(t.indptr.cpu().numpy(), rp) and np.array_equal(t.indices.cpu().numpy(), rx)
    assert np.array_equal(t.eids.cpu().numpy(), re)
    # canonical in-CSR (eids = NULL): edge ids are the positions, rows ordered by position
    ip, ix = csc.indptr.cpu().numpy(), csc.indices.cpu().numpy()
    order = np.argsort(ix, kind="stable")
    tc = sparse.csr_transpose(sparse.CsrView(csc.num_rows, csc.num_cols, csc.indptr, csc.indices, None))
    assert np.array_equal(tc.eids.cpu().numpy(), order)
    assert np.array_equal(tc.indices.cpu().numpy(), np.repeat(np.arange(n_dst), np.diff(ip))[order])
    tt = sparse.csr_transpose(t)
    assert torch.equal(tt.indptr, csc.indptr) and torch.equal(tt.indices, csc.indices) and torch.equal(tt.eids, csc.eids)
    if nnz:
        # a graph restricted to csc has no COO: its out-CSR now comes from the transpose; backward must still be right
        g2 = mg.DGLGraph(mg.GraphIndex(n_src, n_dst, csc=csc, formats=("csr", "csc")), is_block=True)
        x1 = torch.rand(n_src, 6, device=DEV, requires_grad=True)
        x2 = x1.detach().clone().requires_grad_(True)
        w = torch.rand(n_dst, 6, device=DEV)
        (ops.gspmm(g2, "copy_lhs", "sum", x1, None) * w).sum().backward()
        (ops.gspmm(g, "copy_lhs", "sum", x2, None) * w).sum().backward()
        assert float((x1.grad - x2.grad).abs().max()) <= 1e-4 * float(x2.grad.abs().max() + 1e-12)


@pytest.mark.parametrize("H,F", [(1, 4), (1, 16), (1, 64), (8, 8), (8, 16), (4, 32), (3, 4), (12, 4), (2, 128), (1, 256),
                                 (1, 41), (1, 7), (1, 47), (1, 63), (1, 5), (2, 41)])  # odd widths: ragged 16-byte lanes (one head)
@pytest.mark.parametrize("canonical", [False, True])
def test_head_dot_sddmm_on_csr_walk(oracle, H, F, canonical):
    """u_dot_v per head on a CSR-only graph (the specialised row-constant kernel; (2,128)/(1,256) exceed its lane-group
    limits and take the generic body): both target orders, hub rows, canonical edge order, against the oracle; and the
    u_mul_e/sum edge gradient that reaches it through autograd."""
    n_src, n_dst, nnz = 700, 500, 40000
    src, dst = random_graph(n_src, n_dst, nnz, seed=H * 100 + F)
    rng = np.random.default_rng(H + F)
    U = (rng.random((n_src, H, F), dtype=np.float32) - 0.5)
    V = (rng.random((n_dst, H, F), dtype=np.float32) - 0.5)
    g = mk(n_src, n_dst, src, dst, formats=["csr", "csc"])
    gi = g._index.canonical()[0] if canonical else g._index
    perm = g._index.canonical()[1]
    ref = oracle.sddmm(src, dst, "dot", U, V)                                   # (E, H, 1) in edge-id order
    if canonical and perm is not None:
        ref = ref[perm.cpu().numpy()]
    scale = float(np.abs(ref).max())
    out = ops.gsddmm(gi, "dot", T(U), T(V), "u", "v")
    assert out.shape == ref.shape and float(np.abs(out.cpu().numpy() - ref).max()) < RTOL * scale
    out = ops.gsddmm(gi, "dot", T(V), T(U), "v", "u")
    assert float(np.abs(out.cpu().numpy() - ref).max()) < RTOL * scale
    # edge gradient of u_mul_e/sum: d a[e, h] = <X[u, h, :], dZ[v, h, :]>
    a = torch.rand(nnz, H, 1, device=DEV, requires_grad=True)
    z = ops.gspmm(gi, "mul", "sum", T(U), a)
    (z * T(V)).sum().backward()
    assert float((a.grad.cpu() - torch.from_numpy(ref)).abs().max()) < RTOL * scale


@pytest.mark.parametrize("n,M,K", [(0, 3, 5), (1, 1, 1), (5, 64, 128), (1000, 47, 64), (70001, 64, 100), (300000, 16, 7),
                                    (123457, 33, 113), (65536, 64, 64), (70000, 256, 128), (66000, 200, 300), (3000, 65, 129),
                                    (80000, 128, 602), (70000, 256, 1024), (50, 256, 1000), (169343, 256, 512), (0, 200, 300)])
def test_xty_matches_fp64(n, M, K):
    """mgx_xty (dW = dY^T X with fp32 MFMA) against the fp64 product; every tile-count template and ragged edges."""
    rng = np.random.default_rng(n + M + K)
    a = T(rng.standard_normal((n, M)).astype(np.float32))
    b = T(rng.standard_normal((n, K)).astype(np.float32))
    got = sparse.backend_for(a).xty(a, b)
    ref = a.double().t() @ b.double()
    assert got.shape == (M, K)
    scale = float((a.double().abs().t() @ b.double().abs()).max()) if n else 1.0
    assert float((got.double() - ref).abs().max()) <= 1e-5 * max(scale, 1e-30)
    with pytest.raises(mg.DGLError, match="mgx_xty"):
        sparse.backend_for(a).xty(torch.zeros(4, 257, device=DEV), torch.zeros(4, 8, device=DEV))


@pytest.mark.parametrize("n,M,K", [(70001, 64, 128), (65537, 64, 200), (100000, 47, 128), (70000, 64, 64), (66000, 16, 100), (5000, 64, 128),
                                    (70000, 64, 72), (70000, 128, 128), (0, 64, 128)])
def test_xty_with_column_sums_matches_fp64(n, M, K):
    """backend.xty(colsum=True): the weight gradient and, from the same pass where mgx_xty_colsum applies (else mgx_column_sum), the bias
    gradient = column sums of the first operand; both against fp64, the product bit for bit the plain mgx_xty result."""
    rng = np.random.default_rng(n + 7 * M + K)
    a = T(rng.standard_normal((n, M)).astype(np.float32))
    b = T(rng.standard_normal((n, K)).astype(np.float32))
    be = sparse.backend_for(a)
    got, sums = be.xty(a, b, colsum=True)
    assert torch.equal(got, be.xty(a, b))
    ref_s = a.double().sum(0)
    bound = a.double().abs().sum(0)
    assert sums.shape == (M,) and bool(((sums.double() - ref_s).abs() <= 1e-5 * bound + 1e-30).all())
    again, sums2 = be.xty(a, b, colsum=True)
    assert torch.equal(sums, sums2) and torch.equal(again, got)      # fixed summation order
    wide = T(rng.standard_normal((n, M + 8)).astype(np.float32))      # a row-strided first operand
    got_w, sums_w = be.xty(wide[:, :M], b, colsum=True)
    assert bool(((sums_w.double() - wide[:, :M].double().sum(0)).abs() <= 1e-5 * wide[:, :M].double().abs().sum(0) + 1e-30).all())


def test_linear_weight_grad_through_xty_matches_torch():
    from mi355x_graph.nn import Linear
    n = 70000  # >= XTY_MIN_ROWS: the weight gradient takes mgx_xty
    lin = Linear(100, 47).to(DEV)
    ref = torch.nn.Linear(100, 47).to(DEV)
    ref.load_state_dict(lin.state_dict())
    x = torch.rand(n, 100, device=DEV)
    w = torch.randn(n, 47, device=DEV)
    (lin(x) * w).sum().backward()
    (ref(x) * w).sum().backward()
    assert float((lin.weight.grad - ref.weight.grad).abs().max() / ref.weight.grad.abs().max()) < 1e-4
    assert float((lin.bias.grad - ref.bias.grad).abs().max() / ref.bias.grad.abs().max()) < 1e-4


@pytest.mark.parametrize("n", [500, 70000])
def test_linear_sum_matches_two_linears(n):
    """ops.linear_sum (SAGEConv's fc_self(h) + fc_neigh(h_neigh) with the add folded into the second GEMM) == the two
    nn.Linear calls, values and all five gradients (n >= 65536 takes mgx_xty for the weight gradients)."""
    torch.manual_seed(n)
    x1 = torch.rand(n, 40, device=DEV, requires_grad=True)
    x2 = torch.rand(n, 24, device=DEV, requires_grad=True)
    w1 = torch.randn(17, 40, device=DEV, requires_grad=True)
    w2 = torch.randn(17, 24, device=DEV, requires_grad=True)
    b = torch.randn(17, device=DEV, requires_grad=True)
    g = torch.randn(n, 17, device=DEV)
    y = ops.linear_sum(x1, w1, x2, w2, b)
    ref = torch.nn.functional.linear(x1, w1) + torch.nn.functional.linear(x2, w2, b)
    assert float((y - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    got = torch.autograd.grad((y * g).sum(), [x1, w1, x2, w2, b])
    want = torch.autograd.grad((ref * g).sum(), [x1, w1, x2, w2, b])
    for a_, b_ in zip(got, want):
        assert float((a_ - b_).abs().max()) < 1e-4 * float(b_.abs().max())
    # without a bias, and when an input needs no gradient
    y2 = ops.linear_sum(x1.detach(), w1, x2, w2, None)
    assert float((y2 - (ref - b)).abs().max()) < 1e-4 * float(ref.abs().max())
    assert torch.autograd.grad(y2.sum(), [w1, x2])[0].shape == w1.shape


@pytest.mark.parametrize("p", [0.1, 0.5, 0.9])
def test_relu_dropout_fused(p):
    """mgx_relu_dropout_fwd/bwd: y is relu(x)/(1-p) on a kept subset and 0 elsewhere, the kept share of the positive
    entries is 1-p (binomial bound), backward passes dy/(1-p) exactly where y != 0, consecutive calls draw different
    masks, evaluation mode is plain relu."""
    n, d = 40000, 64
    x = torch.randn(n, d, device=DEV)
    xr = x.clone().requires_grad_(True)
    y = ops.relu_dropout(xr, p, True)
    one = torch.tensor(1.0, dtype=torch.float32)
    scale = float(one / (one - torch.tensor(p, dtype=torch.float32)))        # 1/(1-p) as the kernel rounds it
    pos = x > 0
    kept = y != 0
    assert bool((kept <= pos).all())                                         # nothing negative survives
    assert torch.equal(y[kept], x[kept] * scale)                             # kept values: exactly relu(x) / (1 - p)
    share = float(kept.sum()) / float(pos.sum())
    sigma = (p * (1 - p) / float(pos.sum())) ** 0.5
    assert abs(share - (1 - p)) < 6 * sigma
    # per-column shares too: no structure along either axis
    col = kept.float().sum(0) / pos.float().sum(0).clamp(min=1)
    assert float((col - (1 - p)).abs().max()) < 0.03
    g = torch.randn(n, d, device=DEV)
    (y * g).sum().backward()
    assert torch.equal(xr.grad, torch.where(kept, g * scale, torch.zeros_like(g)))
    y2 = ops.relu_dropout(x, p, True)
    assert float(((y2 != 0) != kept).float().mean()) > 0.5 * min(p, 1 - p) * 0.5   # a fresh mask
    assert torch.equal(ops.relu_dropout(x, p, False), torch.relu(x))
    odd = torch.randn(7, 3, device=DEV)                                       # numel % 4 != 0: the PyTorch path
    assert ops.relu_dropout(odd, p, True).shape == odd.shape


@pytest.mark.parametrize("n,C", [(2, 4), (1000, 64), (169343, 256), (50001, 12)])
def test_batch_norm_matches_torch(n, C):
    """nn.BatchNorm1d built on mgx_column_pair_sums / mgx_column_affine: outputs, input / weight / bias gradients and the
    running statistics against torch.nn.BatchNorm1d (fp32 tolerance), training and evaluation mode."""
    from mi355x_graph.nn import BatchNorm1d
    torch.manual_seed(n + C)
    x = (torch.randn(n, C, device=DEV) * 2.0 + 0.5)
    ours, ref = BatchNorm1d(C).to(DEV), torch.nn.BatchNorm1d(C).to(DEV)
    with torch.no_grad():
        ours.weight.uniform_(0.5, 1.5)
        ours.bias.uniform_(-0.5, 0.5)
    ref.load_state_dict(ours.state_dict())
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    g = torch.randn(n, C, device=DEV)
    for step in range(2):
        y1, y2 = ours(x1), ref(x2)
        assert float((y1 - y2).abs().max()) < 2e-4 * float(y2.abs().max())
    (y1 * g).sum().backward()
    (y2 * g).sum().backward()
    for a_, b_ in ((x1.grad, x2.grad), (ours.weight.grad, ref.weight.grad), (ours.bias.grad, ref.bias.grad),
                   (ours.running_mean, ref.running_mean), (ours.running_var, ref.running_var)):
        # input gradients of a normalisation cancel to ~0 (exactly 0 for n = 2): absolute floor at fp32 rounding of O(1) terms
        assert float((a_ - b_).abs().max()) < 5e-6 + 5e-4 * float(b_.abs().max()), (float((a_ - b_).abs().max()), float(b_.abs().max()))
    assert int(ours.num_batches_tracked) == int(ref.num_batches_tracked) == 2
    ours.eval(), ref.eval()
    assert float((ours(x) - ref(x)).abs().max()) < 2e-4 * float(ref(x).abs().max())


@pytest.mark.parametrize("n,C", [(4096, 8), (300001, 64)])
def test_batch_norm_large_mean_small_std(n, C):
    """|mean| >> std (mean 1e3, std 1e-1): E[x^2] - mean^2 from plain fp32 sums would lose the variance entirely; the
    library's sums are taken relative to the first row (mgx_column_pair_sums modes 2 / 3), like Welford in torch."""
    from mi355x_graph.nn import BatchNorm1d
    torch.manual_seed(n)
    x = torch.randn(n, C, device=DEV) * 0.1 + 1000.0
    ours, ref = BatchNorm1d(C).to(DEV), torch.nn.BatchNorm1d(C).to(DEV)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    g = torch.randn(n, C, device=DEV)
    y1, y2 = ours(x1), ref(x2)
    exact = (x.double() - x.double().mean(0)) / (x.double().var(0, unbiased=False) + 1e-5).sqrt()
    e_ours, e_ref = float((y1.double() - exact).abs().max()), float((y2.double() - exact).abs().max())
    assert e_ours < max(4.0 * e_ref, 5e-3), (e_ours, e_ref)     # outputs are O(1); fp32 x carries ~6e-5 of its own
    assert float((ours.running_var - ref.running_var).abs().max()) < 1e-3 * float(ref.running_var.abs().max())
    (y1 * g).sum().backward()
    (y2 * g).sum().backward()
    gscale = float(x2.grad.abs().max())
    assert float((x1.grad - x2.grad).abs().max()) < 2e-2 * gscale, (float((x1.grad - x2.grad).abs().max()), gscale)
    assert float((ours.weight.grad - ref.weight.grad).abs().max()) < 2e-3 * float(ref.weight.grad.abs().max()) + 1e-2


@pytest.mark.parametrize("D,density", [(64, 0.08), (64, 1.0), (100, 0.3), (8, 0.0), (256, 0.5), (4, 0.02)])
def test_row_sparse_gradient_aggregation(oracle, D, density, monkeypatch):
    """mgx_row_nonzero_bits + mgx_spmm_copy_u_masked (the backward aggregation of copy_u when most gradient rows are zero,
    main_dgl_product_sage.py:105-106 trains on 8 % of the nodes): bitmap bit-exact, result == the plain g-SpMM == oracle,
    with hub rows (chunked path), every density incl. all-zero and dense, and through autograd."""
    monkeypatch.setattr(mgx_config, "SPARSE_GRAD_MIN_NNZ", 0)
    monkeypatch.setattr(mgx_config, "SPARSE_GRAD", True)
    n, nnz = 3000, 90000
    src, dst = random_graph(n, n, nnz, seed=D)
    src[:4000], dst[:2500] = 7, 11                       # a hub source and a hub destination
    g = mk(n, n, src, dst)
    rng = np.random.default_rng(D)
    X = rng.standard_normal((n, D)).astype(np.float32)
    keep = rng.random(n) < density
    X[~keep] = 0.0
    if density > 0:
        X[5, 1] = 1e-30                                    # a row whose only non-zero is tiny still counts
        keep[5] = True
    x = T(X)
    be = sparse.backend_for(x)
    bits = be.row_nonzero_bits(x).cpu().numpy().view(np.uint32)
    got = ((bits[np.arange(n) // 32] >> (np.arange(n) % 32)) & 1).astype(bool)
    assert np.array_equal(got, (X != 0).any(1))
    csr = g._index.csr()                                   # rows = sources: the backward aggregation walks this one
    out = sparse.gspmm_grad_raw(csr, x)
    plain, _, _ = sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None)
    rp, rx, re = oracle.coo_to_csr(n, src, dst)
    ref = oracle.spmm(rp, rx, re, "copy_lhs", "sum", X, None)
    scale = oracle.spmm(rp, rx, re, "copy_lhs", "sum", np.abs(X), None)
    assert float(np.max(np.abs(out.cpu().numpy() - ref) - RTOL * scale)) <= 1e-12
    assert float((out - plain).abs().max()) <= 1e-5 * max(float(plain.abs().max()), 1.0)
    # through autograd: d/dx of sum over the destination rows selected by a mask (the training-split pattern)
    h = T(rng.standard_normal((n, D)).astype(np.float32)).requires_grad_(True)
    sel = torch.from_numpy(np.nonzero(keep)[0]).to(DEV)
    w = torch.randn(sel.shape[0], D, device=DEV)
    (ops.gspmm(g, "copy_lhs", "mean", h, None)[sel] * w).sum().backward()
    h2 = h.detach().clone().requires_grad_(True)
    monkeypatch.setattr(mgx_config, "SPARSE_GRAD", False)
    (ops.gspmm(g, "copy_lhs", "mean", h2, None)[sel] * w).sum().backward()
    assert float((h.grad - h2.grad).abs().max()) <= 1e-5 * max(float(h2.grad.abs().max()), 1e-6)


def test_wide_rows_are_aggregated_on_a_line_padded_copy(oracle):
    """sparse.gspmm_raw: D = 602 (reddit's input width) on a dense graph goes through the copy padded to 608 columns and
    comes back as its [:, :602] view -- same numbers as the oracle, gradients flow through the autograd wrapper."""
    n, D = 1200, 602
    nnz = (1 << 20) + 4096
    src, dst = random_graph(n, n, nnz, seed=5)
    rng = np.random.default_rng(5)
    X = rng.random((n, D), dtype=np.float32)
    g = mk(n, n, src, dst, torch.int32)
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    x = T(X).requires_grad_(True)
    out = ops.gspmm(g, "copy_lhs", "mean", x, None)
    assert out.shape == (n, D) and not out.is_contiguous()  # the padded path was taken
    assert rel(out.detach().cpu(), oracle.spmm(ip, ix, ei, "copy_lhs", "mean", X, None)) < RTOL
    w = T(rng.random((n, D), dtype=np.float32))
    (out * w).sum().backward()
    # d/dX of sum(w * mean-aggregate) = aggregate over the reversed graph of w / deg
    deg = np.maximum(np.bincount(dst, minlength=n), 1).astype(np.float32)
    rp, rx, re = oracle.coo_to_csr(n, src, dst)
    ref = oracle.spmm(rp, rx, re, "copy_lhs", "sum", (w.cpu().numpy() / deg[:, None]).astype(np.float32), None)
    assert rel(x.grad.cpu(), ref) < RTOL


@pytest.mark.parametrize("D", [4, 16, 64, 100, 128, 260, 7])
def test_gather_rows_dense_strided_and_odd_widths(D):
    """mgx_gather_rows / mgx_gather_rows_strided (the pack of a partition's boundary rows, dist.py): bit-exact row copies out of a dense
    matrix, out of the left half of a wider one (row stride 2 D + 8), with int32 and int64 indices, repeated and out-of-order rows."""
    from mi355x_graph import sparse
    rng = np.random.default_rng(D)
    n, m = 5000, 12345
    wide = T(rng.standard_normal((n, 2 * D + 8)).astype(np.float32))
    dense = wide[:, :D].contiguous()
    be = sparse.backend_for(wide)
    for dtype in (torch.int32, torch.int64):
        idx = torch.from_numpy(rng.integers(0, n, m)).to(DEV).to(dtype)
        want = dense[idx.long()]
        assert torch.equal(be.gather_rows(dense, idx), want)
        assert torch.equal(be.gather_rows(wide[:, :D], idx), want)          # strided view (falls back to a dense copy for odd D)
        assert torch.equal(be.gather_rows(wide[:, 8:8 + D], idx), wide[:, 8:8 + D].contiguous()[idx.long()])
    assert be.gather_rows(dense, torch.zeros(0, dtype=torch.int32, device=DEV)).shape == (0, D)
