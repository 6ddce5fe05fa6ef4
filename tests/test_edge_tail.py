"""csrc/spmm_tail.inc (round 5): copy_u / sum|mean of a CONSTANT matrix of exactly 100 columns -- the layer-1 aggregation of ogbn-products'
input features (main_dgl_product_sage.py:61-62) -- as a compact [N, 96] block (three cache lines per row) + the last four columns laid out
once along the edge list.  Against the CPU oracle through the C ABI and against the one-matrix kernel; the layer with and without it."""
import numpy as np
import pytest
import torch

import mi355x_graph as mg
from mi355x_graph import _lib, config as mgx_config, ops, sparse

DEV = "cuda:0"


def _graph(n, seed):
    rng = np.random.default_rng(seed)
    deg = rng.poisson(30.0, n).astype(np.int64)
    deg[rng.integers(0, n, n // 20)] = 0
    deg[[5, 1234]] = [1500, 600]                         # hub rows: split into 256-edge chunks, partial slots + fix-up
    dst = np.repeat(np.arange(n), deg)
    src = rng.integers(0, n, dst.shape[0])
    perm = rng.permutation(dst.shape[0])
    return src[perm], dst[perm], deg


@pytest.mark.gpu
def test_edge_tail_against_the_oracle_and_the_one_matrix_kernel(oracle):
    n = 20000
    src, dst, deg = _graph(n, 3)
    g = mg.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=n).int().to(DEV)
    csc = g._index.csc()
    be = sparse.backend_for(csc.indptr)
    rng = np.random.default_rng(1)
    X = rng.standard_normal((n, 100)).astype(np.float32)
    wide = torch.zeros(n, 200, device=DEV)
    wide[:, :100] = torch.from_numpy(X).to(DEV)
    x = wide[:, :100]                                     # the left half of the layer's [x | neigh] buffer
    operands = be.edge_tail_of(csc, x)
    assert operands is not None
    a, tail = operands
    assert a.shape == (n, 96) and a.data_ptr() % 128 == 0 and torch.equal(a, x[:, :96])
    assert torch.equal(tail, x[:, 96:][csc.indices.long()])           # position order of THIS CSR
    plan, short = csc.spmm_plan_for(100)
    assert not short and plan is not None and plan.num_slots >= 6
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    absum = oracle.spmm(ip, ix, ei, "copy_lhs", "sum", np.abs(X), None)
    for red in ("sum", "mean"):
        out = torch.full((n, 100), float("nan"), device=DEV)
        be.spmm_copy_u_edge_tail(csc, red, a, tail, out)
        assert _lib.lib().mgx_last_spmm_kernel().decode() == "edge tail"
        ref = oracle.spmm(ip, ix, ei, "copy_lhs", red, X, None)
        scale = absum / np.maximum(deg, 1)[:, None] if red == "mean" else absum
        assert bool((np.abs(out.cpu().numpy() - ref) <= 1e-4 * scale + 1e-30).all())
        dense = torch.empty(n, 100, device=DEV)
        be.spmm_copy_u_strided(csc, red, x, dense)
        assert torch.equal(out[:, :96], dense[:, :96])               # the gathered part: the row kernel's own order of additions
        assert float((out[:, 96:] - dense[:, 96:]).abs().max()) <= 2e-6 * float(dense.abs().max())
    sc = torch.from_numpy(rng.random(n).astype(np.float32) + 0.5).to(DEV)
    base = torch.from_numpy(rng.standard_normal((n, 200)).astype(np.float32)).to(DEV)
    buf = base.clone()
    be.spmm_copy_u_edge_tail(csc, "sum", a, tail, buf[:, 100:], accumulate=True, dst_scale=sc)
    want = base[:, 100:].cpu().numpy() + oracle.spmm(ip, ix, ei, "copy_lhs", "sum", X, None) * sc.cpu().numpy()[:, None]
    assert bool((np.abs(buf[:, 100:].cpu().numpy() - want) <= 1e-4 * (absum * sc.cpu().numpy()[:, None] + np.abs(base[:, 100:].cpu().numpy())) + 1e-30).all())
    assert torch.equal(buf[:, :100], base[:, :100])
    assert be.edge_tail_of(csc, torch.zeros(n, 64, device=DEV)) is None and be.edge_tail_of(csc, torch.zeros(n + 1, 100, device=DEV)) is None


@pytest.mark.gpu
def test_the_first_layer_with_and_without_the_edge_tail(monkeypatch):
    import full_graph
    from mi355x_graph.datasets import synthetic_edges
    n = 30000
    src, dst = synthetic_edges(n, 500000, 2000, seed=3, device=torch.device(DEV), symmetric=True)
    g = mg.graph((src, dst), num_nodes=n).int().formats(["csr", "csc"]).to(DEV)
    gen = torch.Generator().manual_seed(2)
    feats = torch.rand(n, 100, generator=gen).to(DEV)
    labels = torch.randint(0, 47, (n,), generator=gen).to(DEV)
    monkeypatch.setattr(mgx_config, "EDGE_TAIL_MIN_NNZ", 0)
    monkeypatch.setattr(mgx_config, "PACKED_GATHER", False)
    built = {"n": 0}
    orig = sparse.HipBackend.edge_tail_of

    def spy(self, csr, x2d):
        built["n"] += 1
        return orig(self, csr, x2d)

    monkeypatch.setattr(sparse.HipBackend, "edge_tail_of", spy)
    results = {}
    for on in (True, False):
        monkeypatch.setattr(mgx_config, "EDGE_TAIL", on)
        built["n"] = 0
        torch.manual_seed(7)
        ops.ReluDropout._calls = 0
        model = full_graph.GraphSAGE(100, 64, 47, 3, 0.5, False, True).to(DEV)
        model.train()
        losses = []
        for step in range(3):
            model.zero_grad()
            loss = ops.nll_sum(model(g, feats), labels) / n
            loss.backward()
            losses.append(float(loss))
            if step == 1:
                feats.mul_(1.0)                            # an in-place write: the version counter moves, the layout is rebuilt
        results[on] = (losses, [p.grad.clone() for p in model.parameters()])
        assert built["n"] == (2 if on else 0)              # once, and once more after the write
    for a, b in zip(results[True][0], results[False][0]):
        assert abs(a - b) <= 1e-5 * abs(b)
    for a, b in zip(results[True][1], results[False][1]):
        assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-6)
