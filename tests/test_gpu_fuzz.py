"""GPU parity, randomized: seeded random graphs / widths / operators / schedules against the oracle.  Complements the fixed
shapes of test_gpu_parity.py -- every case draws the graph size, skew, feature width (incl. non-multiples of 4 and widths
past one wave), operator, reducer, index width, format restriction and schedule policy."""
import os

import numpy as np
import pytest
import torch

from mi355x_graph import config as mgx_config

import mi355x_graph as mg
from mi355x_graph import ops
from conftest import random_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = 1e-4
WIDTHS = [1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 32, 33, 41, 47, 64, 100, 128, 200, 256, 300]


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def close(out, ref):
    """Compared at the scale of the row's operands: sums of signed values cancel."""
    out, ref = np.asarray(out, np.float64), np.asarray(ref, np.float64)
    assert out.shape == ref.shape
    if ref.size == 0:
        return True
    return float(np.abs(out - ref).max()) <= RTOL * max(1.0, float(np.abs(ref).max()))


def close_rows(out, ref, row_scale):
    """north_star's 1e-4 RELATIVE bound, row by row: |out - ref| <= 1e-4 * (sum of |terms| reduced into that element).
    The tensor-scale `close` above would hide a row that is wrong by 1e-4 absolute when another row is large."""
    out, ref, row_scale = (np.asarray(t, np.float64) for t in (out, ref, row_scale))
    assert out.shape == ref.shape == row_scale.shape
    if ref.size == 0:
        return True
    bad = np.abs(out - ref) > RTOL * row_scale + 1e-30
    return not bool(bad.any())


def draw_graph(rng, seed):
    n_src = int(rng.integers(1, 3000))
    n_dst = int(rng.integers(1, 3000))
    nnz = int(rng.integers(0, 60000))
    src, dst = random_graph(n_src, n_dst, nnz, seed=seed, skew=bool(rng.integers(0, 2)))
    return n_src, n_dst, nnz, src, dst


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_gspmm(oracle, seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    n_src, n_dst, nnz, src, dst = draw_graph(rng, seed)
    D = int(rng.choice(WIDTHS))
    op = str(rng.choice(["copy_lhs", "copy_rhs", "mul", "add", "sub", "div"]))
    red = str(rng.choice(["sum", "mean", "max", "min"]))
    idtype = torch.int32 if rng.integers(0, 2) else torch.int64
    monkeypatch.setenv("MGX_SCHEDULE", str(rng.choice(["auto", "natural", "none"])))
    monkeypatch.setattr(mgx_config, "HUB_SPLIT", int(rng.choice([64, 256, 1024])))
    ewidth = D if rng.integers(0, 2) else 1          # full-width or scalar edge feature
    X = rng.standard_normal((n_src, D)).astype(np.float32)
    E = (rng.random((nnz, ewidth)).astype(np.float32) + 0.5)
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=idtype, device=DEV)
    if rng.integers(0, 2):
        g = g.formats(["csr", "csc"])
    ip, ix, ei = oracle.coo_to_csr(n_dst, dst, src)
    out = ops.gspmm(g, op, red, T(X), T(E))
    ref = oracle.spmm(ip, ix, ei, op, red, X, E)
    assert close(out.cpu().numpy(), ref), (n_src, n_dst, nnz, D, op, red, ewidth)
    if red in ("sum", "mean"):
        # per-row scale: the same reduction over |terms| (|x| + |e| bounds |x +- e| for add / sub)
        mag_op = "add" if op == "sub" else op
        scale = oracle.spmm(ip, ix, ei, mag_op, red, np.abs(X), np.abs(E))
        assert close_rows(out.cpu().numpy(), ref, scale), (n_src, n_dst, nnz, D, op, red, ewidth)
    else:  # max / min select one term: no accumulation error, only the rounding of the operator itself
        o = out.cpu().numpy().astype(np.float64)
        assert float(np.abs(o - ref).max(initial=0.0)) <= 2e-6 * max(1.0, float(np.abs(ref).max(initial=0.0)))
        assert bool((np.abs(o - ref) <= 2e-6 * np.abs(ref) + 1e-12).all()), (op, red)


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_gsddmm(oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    n_src, n_dst, nnz, src, dst = draw_graph(rng, 100 + seed)
    op = str(rng.choice(["add", "sub", "mul", "div", "dot", "copy_lhs", "copy_rhs"]))
    H = int(rng.choice([1, 1, 2, 4, 8]))
    F = int(rng.choice([1, 3, 4, 8, 16, 41, 64]))
    U = rng.standard_normal((n_src, H, F)).astype(np.float32)
    V = (rng.random((n_dst, H, F)).astype(np.float32) + 0.5)
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=torch.int32, device=DEV)
    if rng.integers(0, 2):
        g = g.formats(["csr", "csc"])
    out = ops.gsddmm(g, op, T(U), T(V))
    ref = oracle.sddmm(src, dst, op, U, V)
    assert close(out.cpu().numpy(), ref), (n_src, n_dst, nnz, H, F, op)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_edge_softmax_and_gat_attention(oracle, seed, monkeypatch):
    rng = np.random.default_rng(3000 + seed)
    n = int(rng.integers(2, 2500))
    nnz = int(rng.integers(1, 60000))
    src, dst = random_graph(n, n, nnz, seed=200 + seed)
    H = int(rng.choice([1, 2, 3, 4, 8, 16]))
    monkeypatch.setattr(mgx_config, "HUB_SPLIT", int(rng.choice([64, 256])))
    g = mg.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=n).int().to(DEV)
    z = (rng.standard_normal((nnz, H, 1)) * 3).astype(np.float32)
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    a = ops.edge_softmax(g, T(z))
    ref = oracle.edge_softmax_fwd(ip, ei, z.reshape(nnz, H))
    assert close(a.cpu().numpy().reshape(nnz, H), ref)
    da = rng.standard_normal((nnz, H)).astype(np.float32)
    zt = T(z).requires_grad_(True)
    (ops.edge_softmax(g, zt) * T(da).view(nnz, H, 1)).sum().backward()
    assert close(zt.grad.cpu().numpy().reshape(nnz, H), oracle.edge_softmax_bwd(ip, ei, ref, da))
    el = rng.standard_normal((n, H, 1)).astype(np.float32)
    er = rng.standard_normal((n, H, 1)).astype(np.float32)
    zz = oracle.sddmm(src, dst, "add", el, er)
    zz = np.where(zz > 0, zz, 0.2 * zz).astype(np.float32)
    fused = ops.gat_attention(g, T(el), T(er), 0.2)
    assert close(fused.cpu().numpy().reshape(nnz, H), oracle.edge_softmax_fwd(ip, ei, zz.reshape(nnz, H)))
