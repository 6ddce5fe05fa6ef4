"""Generates tests/golden/*.npz -- inputs and expected outputs for the hot path.

The reference (dglai/dgl-0.5-benchmark) holds no golden vectors and its DGL dependency is not
importable here (ModuleNotFoundError), so the expected outputs come from INDEPENDENT
implementations present in this container -- scipy.sparse, numpy and plain torch ops -- never from
the oracle or the HIP library under test.  Run:  python tests/golden/make_golden.py
"""
import os

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def graph(n_src, n_dst, nnz, seed):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, n_dst + 1) ** 0.9
    w[rng.integers(0, n_dst, size=max(1, n_dst // 8))] = 0.0
    w /= w.sum()
    dst = rng.choice(n_dst, size=nnz, p=w).astype(np.int64)
    src = rng.integers(0, n_src, size=nnz).astype(np.int64)
    return src, dst


def csr_numpy(n_rows, row, col):
    perm = np.argsort(row, kind="stable")
    indptr = np.zeros(n_rows + 1, np.int64)
    np.add.at(indptr, row + 1, 1)
    return np.cumsum(indptr), col[perm], perm


def make(name, n_src, n_dst, nnz, D, H, seed):
    rng = np.random.default_rng(seed + 1000)
    src, dst = graph(n_src, n_dst, nnz, seed)
    X = rng.random((n_src, D), dtype=np.float32)
    V = rng.random((n_dst, D), dtype=np.float32)
    indptr, indices, eids = csr_numpy(n_dst, dst, src)
    out = {"n_src": n_src, "n_dst": n_dst, "src": src, "dst": dst, "X": X, "V": V,
           "csc_indptr": indptr, "csc_indices": indices, "csc_eids": eids}
    # copy_u / sum : scipy CSR @ X accumulates sequentially in row storage order
    A = sp.csr_matrix((np.ones(nnz, np.float32), indices, indptr), shape=(n_dst, n_src))
    # scipy sums duplicates lazily; csr @ dense walks the stored entries in order
    out["copy_u_sum"] = (A @ X).astype(np.float32)
    deg = np.diff(indptr)
    out["in_degrees"] = deg
    out["copy_u_mean"] = (out["copy_u_sum"] / np.maximum(deg, 1)[:, None].astype(np.float32)).astype(np.float32)
    # u_mul_e / sum with per-head weights: (N,H,F) x (E,H,1), fp64 torch reference
    F = D // H
    W = rng.random((nnz, H, 1), dtype=np.float32)
    out["W"] = W
    msg = torch.from_numpy(X).double().view(n_src, H, F)[torch.from_numpy(src)] * torch.from_numpy(W).double()
    acc = torch.zeros(n_dst, H, F, dtype=torch.float64)
    acc.index_add_(0, torch.from_numpy(dst), msg)
    out["u_mul_e_sum_f64"] = acc.numpy()
    # max with numpy
    mx = np.full((n_dst, D), -np.inf, np.float32)
    np.maximum.at(mx, dst, X[src])
    mx[deg == 0] = 0.0
    out["copy_u_max"] = mx
    # SDDMM (kernel/utils.py:8-16 dense definitions)
    out["u_add_v"] = X[src] + V[dst]
    out["u_mul_v"] = X[src] * V[dst]
    out["u_dot_v_f64"] = (X[src].astype(np.float64) * V[dst].astype(np.float64)).sum(-1, keepdims=True)
    # edge softmax per head, fp64
    Z = (rng.standard_normal((nnz, H)) * 3).astype(np.float32)
    out["Z"] = Z
    z = torch.from_numpy(Z).double()
    m = torch.full((n_dst, H), -float("inf"), dtype=torch.float64)
    m = m.scatter_reduce(0, torch.from_numpy(dst)[:, None].expand(-1, H), z, reduce="amax")
    s = torch.exp(z - m[torch.from_numpy(dst)])
    den = torch.zeros(n_dst, H, dtype=torch.float64).index_add_(0, torch.from_numpy(dst), s)
    a = s / den[torch.from_numpy(dst)]
    out["edge_softmax_f64"] = a.numpy()
    dA = rng.standard_normal((nnz, H)).astype(np.float32)
    out["dA"] = dA
    sds = a * torch.from_numpy(dA).double()
    accum = torch.zeros(n_dst, H, dtype=torch.float64).index_add_(0, torch.from_numpy(dst), sds)
    out["edge_softmax_bwd_f64"] = (sds - a * accum[torch.from_numpy(dst)]).numpy()
    # integer transforms
    n = max(n_src, n_dst)
    key = np.unique(np.concatenate([src * n + dst, dst * n + src]))
    out["bidir_src"], out["bidir_dst"] = key // n, key % n
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def tiny():
    """Hand-computed cases: path, star, isolated node, self loop, multi-edge."""
    src = np.array([0, 1, 2, 0, 0, 3, 3, 1], np.int64)
    dst = np.array([1, 2, 3, 3, 3, 3, 0, 2], np.int64)  # node 4 isolated; (3,3) self loop; (1,2) twice
    X = np.arange(5 * 2, dtype=np.float32).reshape(5, 2) + 1  # [[1,2],[3,4],[5,6],[7,8],[9,10]]
    # in-neighbours: 0<-{3}, 1<-{0}, 2<-{1,1}, 3<-{2,0,0,3}, 4<-{}
    copy_u_sum = np.array([[7, 8], [1, 2], [6, 8], [5 + 1 + 1 + 7, 6 + 2 + 2 + 8], [0, 0]], np.float32)
    deg = np.array([1, 1, 2, 4, 0], np.int64)
    np.savez_compressed(os.path.join(HERE, "tiny.npz"), src=src, dst=dst, X=X, n=5, copy_u_sum=copy_u_sum,
                        in_degrees=deg,
                        csc_indptr=np.array([0, 1, 2, 4, 8, 8], np.int64),
                        csc_indices=np.array([3, 0, 1, 1, 2, 0, 0, 3], np.int64),
                        csc_eids=np.array([6, 0, 1, 7, 2, 3, 4, 5], np.int64),
                        self_loop_src=np.concatenate([src, np.arange(5)]),
                        self_loop_dst=np.concatenate([dst, np.arange(5)]))


if __name__ == "__main__":
    tiny()
    make("g200", 200, 200, 5000, 16, 4, seed=7)         # the 200-node / 5000-edge multigraph of SURVEY 8c
    make("bip", 300, 120, 3000, 24, 3, seed=11)          # bipartite block, N_src != N_dst, H does not divide 64
    make("cora_like", 2708, 2708, 10556, 16, 4, seed=1)  # cora-sized (config 0)
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
