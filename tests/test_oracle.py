"""CPU suite: the oracle (oracle/oracle.c) against the committed golden vectors.

The goldens come from scipy / numpy / torch (tests/golden/make_golden.py), not from the oracle.
Integer results must be bit-exact; fp32 sums in storage order are bit-identical to scipy's
CSR @ dense; everything else within 1e-6 relative of the fp64 goldens.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["g200", "bip", "cora_like"]


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-6)))


def test_tiny_hand_computed(oracle):
    g = load("tiny")
    indptr, indices, eids = oracle.coo_to_csr(int(g["n"]), g["dst"], g["src"])
    assert np.array_equal(indptr, g["csc_indptr"]) and np.array_equal(indices, g["csc_indices"])
    assert np.array_equal(eids, g["csc_eids"])
    out = oracle.spmm(indptr, indices, eids, "copy_lhs", "sum", g["X"], None)
    assert np.array_equal(out, g["copy_u_sum"])
    assert np.array_equal(oracle.in_degrees(indptr), g["in_degrees"])
    mean = oracle.spmm(indptr, indices, eids, "copy_lhs", "mean", g["X"], None)
    assert np.array_equal(mean[4], np.zeros(2, np.float32))  # isolated node -> 0, not NaN
    assert np.allclose(mean[3], g["copy_u_sum"][3] / 4)
    s, d = oracle.add_self_loop(g["src"], g["dst"], int(g["n"]))
    assert np.array_equal(s, g["self_loop_src"]) and np.array_equal(d, g["self_loop_dst"])


@pytest.mark.parametrize("name", CASES)
def test_formats_bit_exact(oracle, name):
    g = load(name)
    indptr, indices, eids = oracle.coo_to_csr(int(g["n_dst"]), g["dst"], g["src"])
    assert np.array_equal(indptr, g["csc_indptr"])
    assert np.array_equal(indices, g["csc_indices"])
    assert np.array_equal(eids, g["csc_eids"])
    assert np.array_equal(oracle.in_degrees(indptr), g["in_degrees"])
    n = max(int(g["n_src"]), int(g["n_dst"]))
    bs, bd = oracle.to_bidirected(g["src"], g["dst"], n)
    assert np.array_equal(bs, g["bidir_src"]) and np.array_equal(bd, g["bidir_dst"])


@pytest.mark.parametrize("name", CASES)
def test_spmm(oracle, name):
    g = load(name)
    ip, ix, ei = g["csc_indptr"], g["csc_indices"], g["csc_eids"]
    out = oracle.spmm(ip, ix, ei, "copy_lhs", "sum", g["X"], None)
    assert np.array_equal(out, g["copy_u_sum"]), "sequential fp32 row sums must equal scipy's bit for bit"
    assert np.array_equal(oracle.spmm(ip, ix, ei, "copy_lhs", "mean", g["X"], None), g["copy_u_mean"])
    H = g["W"].shape[1]
    X3 = g["X"].reshape(g["X"].shape[0], H, -1)
    out = oracle.spmm(ip, ix, ei, "mul", "sum", X3, g["W"])
    assert rel_err(out, g["u_mul_e_sum_f64"]) < 1e-5
    mx, au, ae = oracle.spmm(ip, ix, ei, "copy_lhs", "max", g["X"], None, want_arg=True)
    assert np.array_equal(mx, g["copy_u_max"])
    has = g["in_degrees"] > 0
    assert np.all(au[~has] == -1)
    rows = np.nonzero(has)[0]
    assert np.array_equal(g["X"][au[rows], np.arange(g["X"].shape[1])[None, :]], mx[rows])
    # copy_e / sum of per-edge messages == index_add
    msg = g["X"][g["src"]]
    assert np.array_equal(oracle.spmm(ip, ix, ei, "copy_rhs", "sum", None, msg), g["copy_u_sum"])


@pytest.mark.parametrize("name", CASES)
def test_sddmm(oracle, name):
    g = load(name)
    assert np.array_equal(oracle.sddmm(g["src"], g["dst"], "add", g["X"], g["V"]), g["u_add_v"])
    assert np.array_equal(oracle.sddmm(g["src"], g["dst"], "mul", g["X"], g["V"]), g["u_mul_v"])
    assert rel_err(oracle.sddmm(g["src"], g["dst"], "dot", g["X"], g["V"]), g["u_dot_v_f64"]) < 1e-5
    assert np.array_equal(oracle.sddmm(g["src"], g["dst"], "copy_lhs", g["X"], None), g["X"][g["src"]])
    assert np.array_equal(oracle.sddmm(g["src"], g["dst"], "copy_rhs", None, g["V"]), g["V"][g["dst"]])


@pytest.mark.parametrize("name", CASES)
def test_edge_softmax(oracle, name):
    g = load(name)
    a = oracle.edge_softmax_fwd(g["csc_indptr"], g["csc_eids"], g["Z"])
    assert rel_err(a, g["edge_softmax_f64"]) < 1e-5
    sums = np.zeros((int(g["n_dst"]), a.shape[1]))
    np.add.at(sums, g["dst"], a)
    assert np.allclose(sums[g["in_degrees"] > 0], 1.0, atol=1e-5)
    dz = oracle.edge_softmax_bwd(g["csc_indptr"], g["csc_eids"], g["edge_softmax_f64"].astype(np.float32), g["dA"])
    assert np.max(np.abs(dz - g["edge_softmax_bwd_f64"])) < 1e-5


def test_segment_reduce_and_batch(oracle):
    rng = np.random.default_rng(3)
    lens = np.array([3, 0, 5, 1, 7], np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    x = rng.random((int(lens.sum()), 6), dtype=np.float32)
    ref = np.stack([x[off[i]:off[i + 1]].sum(0) if lens[i] else np.zeros(6, np.float32) for i in range(5)])
    assert np.allclose(oracle.segment_reduce(off, x, "sum"), ref, rtol=1e-6)
    mean = oracle.segment_reduce(off, x, "mean")
    assert np.allclose(mean, ref / np.maximum(lens, 1)[:, None], rtol=1e-6)
    n, s, d, bn, be = oracle.batch([(3, [0, 1], [1, 2]), (2, [0], [1]), (4, [3, 3, 0], [0, 1, 2])])
    assert n == 9 and list(s) == [0, 1, 3, 8, 8, 5] and list(d) == [1, 2, 4, 5, 6, 7]
    assert list(bn) == [3, 2, 4] and list(be) == [2, 1, 3]


def test_properties(oracle):
    """Linearity and adjointness <A x, y> = <x, A^T y> (backward = forward on the reversed graph)."""
    rng = np.random.default_rng(5)
    n, nnz, D = 150, 2000, 8
    src, dst = rng.integers(0, n, nnz), rng.integers(0, n, nnz)
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    rp, rx, re = oracle.coo_to_csr(n, src, dst)
    x, y = rng.random((n, D), dtype=np.float32), rng.random((n, D), dtype=np.float32)
    Ax = oracle.spmm(ip, ix, ei, "copy_lhs", "sum", x, None)
    ATy = oracle.spmm(rp, rx, re, "copy_lhs", "sum", y, None)
    assert abs(float((Ax.astype(np.float64) * y).sum() - (x.astype(np.float64) * ATy).sum())) < 1e-2
    A2x = oracle.spmm(ip, ix, ei, "copy_lhs", "sum", 2 * x, None)
    assert np.allclose(A2x, 2 * Ax, rtol=1e-6)
