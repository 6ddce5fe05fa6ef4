"""The CPU (OpenMP) variants behind the same signatures (include/mi355x_graph_cpu.h, csrc/cpu_ops.cpp) -- product code with
its own tests against the oracle and the golden fixtures; opt-in (mi355x_graph.enable_cpu_backend()), never a fallback."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

import mi355x_graph as mg
from mi355x_graph import cpu_backend, ops, sparse
from conftest import random_graph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def cpu_on():
    was = mg.enable_cpu_backend(True)
    try:
        yield
    finally:
        mg.enable_cpu_backend(was)


def test_library_exports_what_the_header_declares():
    hdr = open(os.path.join(ROOT, "include", "mi355x_graph_cpu.h")).read()
    declared = sorted(set(re.findall(r"\b(mgx_cpu_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(cpu_backend.SIGNATURES)
    h = ctypes.CDLL(cpu_backend.CPU_LIB_PATH)
    for s in declared:
        assert hasattr(h, s), "libmi355x_graph_cpu.so does not export %s" % s
    assert cpu_backend.lib().mgx_cpu_num_threads() >= 1
    # the product library is independent of the checker: nothing of oracle/ is linked or loaded
    out = os.popen("ldd %s" % cpu_backend.CPU_LIB_PATH).read()
    assert "oracle" not in out and "liboracle" not in out


def test_opt_in_and_no_fallback():
    """Default: CPU tensors raise.  Enabled: CPU tensors compute on the CPU backend.  Either way a HIP tensor never reaches it."""
    g = mg.graph((torch.tensor([0, 1, 1]), torch.tensor([1, 0, 1])))
    x = torch.rand(2, 4)
    assert not mg.cpu_backend_enabled()
    with pytest.raises(mg.DGLError, match="MI355X"):
        ops.gspmm(g, "copy_lhs", "sum", x, None)
    was = mg.enable_cpu_backend(True)
    try:
        assert was is False and isinstance(sparse.backend_for(x), cpu_backend.CpuBackend)
        out = ops.gspmm(g, "copy_lhs", "sum", x, None)
        assert torch.allclose(out, torch.stack([x[1], x[0] + x[1]]))
        assert isinstance(sparse._BACKENDS["cuda"], sparse.HipBackend)  # untouched
    finally:
        mg.enable_cpu_backend(False)
    with pytest.raises(mg.DGLError, match="MI355X"):
        ops.gspmm(g, "copy_lhs", "sum", x, None)


@pytest.mark.parametrize("idtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "copy_lhs", "copy_rhs"])
@pytest.mark.parametrize("reduce", ["sum", "mean", "max", "min"])
def test_gspmm_against_the_oracle(oracle, cpu_on, op, reduce, idtype):
    n_src, n_dst, nnz, D = 300, 260, 5000, 12
    src, dst = random_graph(n_src, n_dst, nnz, seed=7)
    rng = np.random.default_rng(3)
    U = rng.random((n_src, D), dtype=np.float32) + 0.25
    E = rng.random((nnz, D), dtype=np.float32) + 0.5
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=idtype)
    out = ops.gspmm(g, op, reduce, torch.from_numpy(U), torch.from_numpy(E))
    ip, ix, ei = oracle.coo_to_csr(n_dst, dst, src)
    oop, Eo = ("add", -E) if op == "sub" else (("mul", 1.0 / E) if op == "div" else (op, E))
    ref = oracle.spmm(ip, ix, ei, oop, reduce, U, Eo.astype(np.float32))
    if reduce in ("max", "min"):
        assert np.array_equal(out.numpy(), ref)          # a selection: bit for bit
    else:
        assert np.allclose(out.numpy(), ref, rtol=1e-5, atol=1e-6)


def test_gspmm_broadcast_heads_and_arg_indices(oracle, cpu_on):
    n, nnz, H, F = 200, 3000, 4, 8
    src, dst = random_graph(n, n, nnz, seed=11)
    rng = np.random.default_rng(5)
    U = rng.random((n, H, F), dtype=np.float32)
    W = rng.random((nnz, H, 1), dtype=np.float32)
    g = mg.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=n).int()
    out = ops.gspmm(g, "mul", "sum", torch.from_numpy(U), torch.from_numpy(W))
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    assert np.allclose(out.numpy(), oracle.spmm(ip, ix, ei, "mul", "sum", U, W), rtol=1e-5, atol=1e-6)
    csc = g._index.csc()
    o, au, ae = sparse.gspmm_raw(csc, "copy_lhs", "max", torch.from_numpy(U[:, 0]), None, want_arg=True)
    ro, ru, _ = oracle.spmm(ip, ix, ei, "copy_lhs", "max", U[:, 0], None, want_arg=True)
    assert np.array_equal(o.numpy(), ro) and np.array_equal(au.numpy(), ru)


@pytest.mark.parametrize("fmt", ["coo", "csr_csc"])
@pytest.mark.parametrize("op", ["add", "sub", "mul", "div", "dot", "copy_lhs", "copy_rhs"])
def test_gsddmm_against_the_oracle(oracle, cpu_on, op, fmt):
    n_src, n_dst, nnz, D = 150, 170, 4000, 16
    src, dst = random_graph(n_src, n_dst, nnz, seed=9)
    rng = np.random.default_rng(2)
    U = rng.random((n_src, D), dtype=np.float32)
    V = rng.random((n_dst, D), dtype=np.float32) + 0.5
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=torch.int32)
    if fmt == "csr_csc":
        g = g.formats(["csr", "csc"])
    out = ops.gsddmm(g, op, torch.from_numpy(U), torch.from_numpy(V))
    ref = oracle.sddmm(src, dst, op, U, V)
    assert out.shape == ref.shape
    if op == "dot":
        assert np.allclose(out.numpy(), ref, rtol=1e-5, atol=1e-6)
    else:
        assert np.array_equal(out.numpy(), ref)


def test_edge_softmax_segment_reduce_and_autograd(oracle, cpu_on):
    n, nnz, H = 120, 2500, 3
    src, dst = random_graph(n, n, nnz, seed=4)
    rng = np.random.default_rng(8)
    z = (rng.standard_normal((nnz, H, 1)) * 3).astype(np.float32)
    g = mg.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=n).int()
    zt = torch.from_numpy(z).requires_grad_(True)
    a = ops.edge_softmax(g, zt)
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    ref = oracle.edge_softmax_fwd(ip, ei, z.reshape(nnz, H))
    assert np.allclose(a.detach().numpy().reshape(nnz, H), ref, rtol=1e-5, atol=1e-7)
    w = torch.from_numpy(rng.standard_normal((nnz, H, 1)).astype(np.float32))
    (a * w).sum().backward()
    gref = oracle.edge_softmax_bwd(ip, ei, ref, w.numpy().reshape(nnz, H))
    assert np.allclose(zt.grad.numpy().reshape(nnz, H), gref, rtol=1e-4, atol=1e-6)
    # segment reduce (AvgPooling of main_dgl_molhiv_gcn.py:75)
    seglen = torch.tensor([3, 0, 5, 1, 7])
    x = torch.from_numpy(rng.random((16, 6), dtype=np.float32))
    for red in ("sum", "mean", "max", "min"):
        out = ops.segment_reduce(seglen, x, red)
        off = np.concatenate([[0], np.cumsum(seglen.numpy())])
        assert np.allclose(out.numpy(), oracle.segment_reduce(off, x.numpy(), red), rtol=1e-6, atol=1e-7)
    # update_all(copy_src, mean) forward + backward: the SAGE aggregation of the reference on CPU tensors
    import mi355x_graph.function as fn
    h = torch.from_numpy(rng.random((n, 10), dtype=np.float32)).requires_grad_(True)
    gl = g.local_var()
    gl.srcdata["h"] = h
    gl.update_all(fn.copy_src("h", "m"), fn.mean("m", "neigh"))
    out = gl.dstdata["neigh"]
    assert np.allclose(out.detach().numpy(), oracle.spmm(ip, ix, ei, "copy_lhs", "mean", h.detach().numpy(), None), rtol=1e-5, atol=1e-6)
    out.sum().backward()
    rp, rx, re_ = oracle.coo_to_csr(n, src, dst)
    inv = (1.0 / np.maximum(np.diff(ip), 1)).astype(np.float32)
    gr = oracle.spmm(rp, rx, re_, "copy_lhs", "sum", np.repeat(inv[:, None], 10, 1), None)
    assert np.allclose(h.grad.numpy(), gr, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
def test_formats_bit_exact(oracle, dtype):
    be = cpu_backend.CpuBackend()
    rng = np.random.default_rng(1)
    for n_rows, n_cols, nnz in [(1, 1, 0), (7, 5, 30), (400, 300, 20000)]:
        row, col = rng.integers(0, n_rows, nnz), rng.integers(0, n_cols, nnz)
        v = be.coo_to_csr(n_rows, n_cols, torch.from_numpy(row).to(dtype), torch.from_numpy(col).to(dtype))
        ip, ix, ei = oracle.coo_to_csr(n_rows, row, col)
        assert np.array_equal(v.indptr.numpy(), ip) and np.array_equal(v.indices.numpy(), ix) and np.array_equal(v.eids.numpy(), ei)
        t = be.csr_transpose(v)
        tp, tx, te = oracle.coo_to_csr(n_cols, col, row)
        assert np.array_equal(t.indptr.numpy(), tp) and np.array_equal(t.indices.numpy(), tx) and np.array_equal(t.eids.numpy(), te)
        assert np.array_equal(be.degrees(v).numpy(), np.diff(ip))


def test_golden_fixtures(cpu_on):
    """The committed golden vectors (tests/golden/*.npz: scipy CSR @ dense, numpy, fp64 torch) through the CPU variants."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    assert files
    checked = 0
    for f in files:
        d = np.load(f)
        if not {"src", "dst", "X", "copy_u_sum"} <= set(d.files):
            continue
        n_dst = int(d["n_dst"]) if "n_dst" in d.files else int(d["copy_u_sum"].shape[0])
        n_src = int(d["X"].shape[0])
        g = mg.create_block((torch.from_numpy(d["src"].astype(np.int64)), torch.from_numpy(d["dst"].astype(np.int64))), n_src, n_dst, idtype=torch.int32)
        X = torch.from_numpy(d["X"].astype(np.float32))
        out = ops.gspmm(g, "copy_lhs", "sum", X, None)
        assert np.array_equal(out.numpy(), d["copy_u_sum"])       # scipy's CSR @ dense sums a row sequentially in fp32: the same bits
        assert np.array_equal(g.in_degrees().numpy(), d["in_degrees"])
        if "copy_u_mean" in d.files:
            assert np.allclose(ops.gspmm(g, "copy_lhs", "mean", X, None).numpy(), d["copy_u_mean"], rtol=1e-6, atol=1e-7)
        csc = g._index.csc()
        assert np.array_equal(csc.indptr.numpy(), d["csc_indptr"]) and np.array_equal(csc.indices.numpy(), d["csc_indices"])
        checked += 1
    assert checked >= 4


def test_bad_arguments_raise_not_crash():
    L = cpu_backend.lib()
    assert L.mgx_cpu_spmm_csr(None, None, 0, 0, None, None, 1, 1, 1, None, None, None, None, None, None, None, None, 0, None) == 1
    assert b"csr is NULL" in L.mgx_cpu_last_error()
    c = mg._lib.MgxCsr(1, 1, 0, None, None, None, 16, 0)
    assert L.mgx_cpu_spmm_csr(ctypes.byref(c), None, 0, 0, None, None, 1, 1, 1, None, None, None, None, None, None, None, None, 0, None) == 1
    assert L.mgx_cpu_sddmm_coo(1, 1, 1, None, None, 32, 99, None, None, 0, 2, 1, 1, 1, 1, None, None, None, None) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("op,reduce", [("copy_lhs", "sum"), ("copy_lhs", "mean"), ("mul", "sum"), ("add", "max"), ("copy_rhs", "min")])
def test_hip_kernels_and_cpu_variants_agree(cpu_on, op, reduce):
    """Two product implementations of one contract, HIP (libmi355x_graph.so) and CPU (libmi355x_graph_cpu.so), on the same inputs:
    selections (max / min values and arg indices) and element-wise g-SDDMM bit for bit, sums within 1e-4 of the row's sum of |terms|."""
    dev = torch.device("cuda:0")
    n_src, n_dst, nnz, D = 900, 700, 40000, 24
    src, dst = random_graph(n_src, n_dst, nnz, seed=21)
    rng = np.random.default_rng(6)
    U = torch.from_numpy(rng.standard_normal((n_src, D)).astype(np.float32))
    E = torch.from_numpy((rng.random((nnz, D)) + 0.5).astype(np.float32))
    edges = (torch.from_numpy(src), torch.from_numpy(dst))
    g_cpu = mg.create_block(edges, n_src, n_dst, idtype=torch.int32)
    g_hip = mg.create_block(edges, n_src, n_dst, idtype=torch.int32, device=dev)
    out_c = ops.gspmm(g_cpu, op, reduce, U, E)
    out_h = ops.gspmm(g_hip, op, reduce, U.to(dev), E.to(dev)).cpu()
    if reduce in ("max", "min"):
        assert torch.equal(out_c, out_h)
    else:
        scale = ops.gspmm(g_cpu, op, "sum", U.abs(), E.abs()) + 1e-6
        assert float(((out_c - out_h).abs() / scale).max()) <= 1e-4
    V = torch.from_numpy(rng.standard_normal((n_dst, D)).astype(np.float32))
    for sop in ("add", "mul", "sub"):
        assert torch.equal(ops.gsddmm(g_cpu, sop, U, V), ops.gsddmm(g_hip, sop, U.to(dev), V.to(dev)).cpu())
    z = torch.from_numpy((rng.standard_normal((nnz, 2, 1)) * 2).astype(np.float32))
    a_c, a_h = ops.edge_softmax(g_cpu, z), ops.edge_softmax(g_hip, z.to(dev)).cpu()
    assert float((a_c - a_h).abs().max()) <= 1e-6


@pytest.mark.gpu
def test_secondary_bench_leg_runs_and_reports_a_roofline(monkeypatch):
    """bench.py's `secondary` block (secondary_bench.py): one leg end to end on the GPU at a reduced size -- ms_per_step, the dominant
    hot-path call and SURVEY 8d's bytes for it."""
    monkeypatch.setenv("MGX_DATASET_SCALE", "0.2")
    sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
    import secondary_bench as sb
    out = sb.sage_arxiv(torch.device("cuda:0"), steps=3, warmup=2)
    r = out["roofline"]
    assert out["ms_per_step"] > 0 and r["bound"] == "hbm" and "g-SpMM copy_lhs" in r["kernel"] and r["D"] if "D" in r else True
    assert r["algorithmic_bytes_per_launch"] > 0 and 0 < r["frac"] < 1 and r["launches_timed"] >= 3
    out = sb._gat(torch.device("cuda:0"), "reddit-small", 2, 8, 16, 0.0, 2, 1, "test", "-")
    assert "fused GAT block" in out["roofline"]["kernel"] and out["ms_per_step"] > 0
