"""GPU suite: the fused GAT message-passing block (mgx_gat_fused_fwd/bwd, csrc/gatfused.hip) against the CPU oracle's
composition  sddmm(add) -> leaky_relu -> edge_softmax -> spmm(mul, sum)  -- the chain GATConv runs at
main_dgl_reddit_gat.py:31-55 -- and, for the gradients, against the library's own unfused operators (which are held to
the oracle in test_gpu_parity.py).  Tolerance: 1e-4 relative (north_star), stated at each assert."""
import numpy as np
import pytest
import torch

from mi355x_graph import config as mgx_config
import torch.nn.functional as F

import mi355x_graph as mg
from mi355x_graph import ops
from conftest import random_graph

pytestmark = pytest.mark.gpu
RTOL = 1e-4
DEV = "cuda:0"


def mk(n_src, n_dst, src, dst):
    return mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=torch.int32, device=DEV)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def hubby_graph(n, nnz, seed, hub_deg=3000):
    """random graph + one hub destination (chunked rows) + one hub source + a few nodes without in-edges"""
    src, dst = random_graph(n, n, nnz, seed=seed)
    rng = np.random.default_rng(seed)
    keep = dst >= 5                      # nodes 0..4 receive nothing
    src, dst = src[keep], dst[keep]
    hs = rng.integers(0, n, hub_deg)
    src = np.concatenate([src, hs, np.full(hub_deg, 7)])
    dst = np.concatenate([dst, np.full(hub_deg, 11), rng.integers(5, n, hub_deg)])
    return src.astype(np.int64), dst.astype(np.int64)


def unfused_attention(g, el, er, slope):
    return ops.edge_softmax(g, F.leaky_relu(ops.gsddmm(g, "add", el, er, "u", "v"), slope))


def unfused(g, feat, el, er, slope):
    return ops.gspmm(g, "mul", "sum", feat, unfused_attention(g, el, er, slope))


@pytest.mark.parametrize("n,H,Fd", [(900, 1, 16), (900, 8, 16), (700, 4, 8), (500, 2, 64), (300, 1, 4), (600, 3, 16), (400, 1, 256),
                                    (400, 16, 4), (350, 2, 128), (800, 1, 41), (500, 1, 7), (400, 1, 100), (300, 1, 5), (300, 1, 253)])
def test_fused_forward_matches_oracle_composition(oracle, n, H, Fd):
    nnz = 40 * n
    src, dst = hubby_graph(n, nnz, seed=H * 100 + Fd)
    E = src.shape[0]
    g = mk(n, n, src, dst)
    rng = np.random.default_rng(H + Fd)
    feat = rng.standard_normal((n, H, Fd)).astype(np.float32)
    el = (rng.standard_normal((n, H, 1)) * 2).astype(np.float32)
    er = (rng.standard_normal((n, H, 1)) * 2).astype(np.float32)
    assert ops.gat_fused_supported(g, T(feat))
    out = ops.gat_fused(g, T(feat), T(el), T(er), 0.2, 0.0, True)
    assert out.shape == (n, H, Fd)
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    z = oracle.sddmm(src, dst, "add", el, er)
    z = np.where(z > 0, z, 0.2 * z).astype(np.float32)
    a = oracle.edge_softmax_fwd(ip, ei, z.reshape(E, H))
    ref = oracle.spmm(ip, ix, ei, "mul", "sum", feat, a.reshape(E, H, 1))
    got = out.cpu().numpy()
    # north_star's 1e-4 RELATIVE bound, element by element: signed sums cancel, so the error is held against the sum of
    # |terms| reduced into THAT output element, sum_e a[e,h] |feat[u,h,f]| (a tensor-scale bound would hide a small row that is
    # wrong by 1e-4 absolute)
    row_scale = oracle.spmm(ip, ix, ei, "mul", "sum", np.abs(feat), a.reshape(E, H, 1)).astype(np.float64)
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    assert not (err > RTOL * row_scale + 1e-30).any(), float((err / (row_scale + 1e-30)).max())
    assert np.all(got[:5] == 0.0)  # nodes without in-edges aggregate to exactly 0


@pytest.mark.parametrize("n,H,Fd", [(900, 8, 16), (800, 1, 16), (500, 2, 64), (600, 3, 16), (300, 4, 4), (800, 1, 41), (500, 1, 7),
                                    (400, 1, 102)])
def test_fused_gradients_match_unfused_operators(n, H, Fd):
    nnz = 40 * n
    src, dst = hubby_graph(n, nnz, seed=H * 7 + Fd)
    g = mk(n, n, src, dst)
    torch.manual_seed(H + Fd)
    feat0 = torch.randn(n, H, Fd, device=DEV)
    el0, er0 = torch.randn(n, H, 1, device=DEV) * 2, torch.randn(n, H, 1, device=DEV) * 2
    w = torch.randn(n, H, Fd, device=DEV)
    ins1 = [t.clone().requires_grad_(True) for t in (feat0, el0, er0)]
    ins2 = [t.clone().requires_grad_(True) for t in (feat0, el0, er0)]
    o1 = ops.gat_fused(g, ins1[0], ins1[1], ins1[2], 0.2, 0.0, True)
    o2 = unfused(g, ins2[0], ins2[1], ins2[2], 0.2)
    assert float((o1 - o2).detach().abs().max()) < RTOL * float(feat0.abs().max())
    (o1 * w).sum().backward()
    (o2 * w).sum().backward()
    for name, x, y in zip(("d_feat", "d_el", "d_er"), ins1, ins2):
        err, ref = float((x.grad - y.grad).abs().max()), float(y.grad.abs().max())
        assert err < RTOL * max(ref, 1e-6), (name, err, ref)  # 1e-4 of the gradient tensor's scale (hub rows sum 3000 terms)
    # second backward through the same saved tensors (nstat[..., 3] is rewritten, not accumulated)
    ins3 = [t.clone().requires_grad_(True) for t in (feat0, el0, er0)]
    o3 = ops.gat_fused(g, ins3[0], ins3[1], ins3[2], 0.2, 0.0, True)
    (o3 * w).sum().backward(retain_graph=True)
    first = [t.grad.clone() for t in ins3]
    for t in ins3:
        t.grad = None
    (o3 * w).sum().backward()
    for a_, b_ in zip(first, ins3):
        assert torch.equal(a_, b_.grad)  # bitwise: no atomics anywhere


def test_fused_only_er_needs_grad_and_no_grad_paths():
    n, H, Fd = 400, 2, 8
    src, dst = hubby_graph(n, 30 * n, seed=3)
    g = mk(n, n, src, dst)
    feat, el = torch.randn(n, H, Fd, device=DEV), torch.randn(n, H, 1, device=DEV)
    er = torch.randn(n, H, 1, device=DEV, requires_grad=True)
    o = ops.gat_fused(g, feat, el, er, 0.2, 0.0, True)
    o.sum().backward()
    er2 = er.detach().clone().requires_grad_(True)
    unfused(g, feat, el, er2, 0.2).sum().backward()
    assert float((er.grad - er2.grad).abs().max()) < RTOL * max(float(er2.grad.abs().max()), 1e-3)
    with torch.no_grad():
        assert torch.equal(ops.gat_fused(g, feat, el, er, 0.2, 0.0, True), o.detach())


@pytest.mark.parametrize("p", [0.3, 0.6])
def test_fused_attention_dropout_is_consistent(p):
    """attn_drop inside the kernels: reproducible for a seed, a fresh mask per call, the forward and both backward walks
    regenerate the SAME mask (adjointness of the linear map feat -> out, and sum_v d_er = sum_u d_el), keep rate 1 - p,
    evaluation mode = no mask."""
    n, H, Fd = 1200, 4, 16
    src, dst = hubby_graph(n, 40 * n, seed=17)
    g = mk(n, n, src, dst)
    torch.manual_seed(5)
    feat = torch.randn(n, H, Fd, device=DEV)
    el, er = torch.randn(n, H, 1, device=DEV), torch.randn(n, H, 1, device=DEV)
    full = ops.gat_fused(g, feat, el, er, 0.2, 0.0, True)
    assert torch.equal(ops.gat_fused(g, feat, el, er, 0.2, p, False), full)        # evaluation: identity
    torch.manual_seed(123)
    ops.GATFused._calls = 0
    d1 = ops.gat_fused(g, feat, el, er, 0.2, p, True)
    d2 = ops.gat_fused(g, feat, el, er, 0.2, p, True)
    torch.manual_seed(123)
    ops.GATFused._calls = 0
    d1b = ops.gat_fused(g, feat, el, er, 0.2, p, True)
    assert torch.equal(d1, d1b) and not torch.equal(d1, d2)
    # E[dropout(a)] = a: with feat = 1 the output is sum_e keep/(1-p) a, mean 1 over many rows
    ones = torch.ones(n, H, Fd, device=DEV)
    frac = ops.gat_fused(g, ones, el, er, 0.2, p, True)[5:, :, 0]
    assert abs(float(frac.mean()) - 1.0) < 0.05, float(frac.mean())
    # same mask in forward and backward: <A x, y> == <x, A^T y> for the linear map x -> out at fixed (el, er, mask)
    x = torch.randn(n, H, Fd, device=DEV, requires_grad=True)
    y = torch.randn(n, H, Fd, device=DEV)
    elg, erg = el.clone().requires_grad_(True), er.clone().requires_grad_(True)
    torch.manual_seed(77)
    ops.GATFused._calls = 0
    out = ops.gat_fused(g, x, elg, erg, 0.2, p, True)
    lhs = float((out.detach().double() * y.double()).sum())
    (out * y).sum().backward()
    rhs = float((x.detach().double() * x.grad.double()).sum())
    assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), 1.0), (lhs, rhs)
    # every edge's d z lands once in d_el (out-CSR walk) and once in d_er (in-CSR walk)
    s_el, s_er = elg.grad.double().sum(0).flatten(), erg.grad.double().sum(0).flatten()
    assert float((s_el - s_er).abs().max()) < 1e-3 * max(float(elg.grad.abs().sum(0).max()), 1e-3), (s_el, s_er)


def test_gatconv_uses_the_fused_block_and_matches_the_unfused_module(monkeypatch):
    """dgl.nn.pytorch.GATConv (the module main_dgl_reddit_gat.py builds): fused and unfused paths give the same outputs
    and parameter gradients; get_attention falls back to the unfused path."""
    from mi355x_graph.nn import GATConv
    n = 1500
    src, dst = hubby_graph(n, 30 * n, seed=23)
    loops = np.arange(n)
    g = mk(n, n, np.concatenate([src, loops]), np.concatenate([dst, loops]))
    torch.manual_seed(0)
    conv = GATConv(24, 16, 8, 0.0, 0.0, 0.2, activation=F.elu).to(DEV)
    x = torch.randn(n, 24, device=DEV)
    calls = []
    real = ops.gat_fused
    monkeypatch.setattr(ops, "gat_fused", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    y1 = conv(g, x)
    assert calls, "GATConv did not take the fused path"
    y1.square().sum().backward()
    g1 = [p.grad.clone() for p in conv.parameters()]
    conv.zero_grad()
    monkeypatch.setenv("MGX_GAT_FUSED", "0")
    y2 = conv(g, x)
    y2.square().sum().backward()
    assert float((y1 - y2).detach().abs().max()) < RTOL * float(y2.detach().abs().max())
    for a_, b_ in zip(g1, [p.grad for p in conv.parameters()]):
        assert float((a_ - b_).abs().max()) < 2e-4 * float(b_.abs().max()) + 1e-6
    monkeypatch.setenv("MGX_GAT_FUSED", "1")
    y3, att = conv(g, x, get_attention=True)
    assert att.shape == (g.number_of_edges(), 8, 1) and float((y3 - y1).abs().max()) < RTOL * float(y1.abs().max())


@pytest.mark.parametrize("k,f,residual", [(16, 41, False), (8, 47, True), (32, 64, False)])
def test_one_head_widening_layer_aggregates_first(k, f, residual, monkeypatch):
    """GATConv(k, f, 1 head) with k < f (main_dgl_reddit_gat.py:62-64: the 16 -> 41 output layer): the fused block runs on the
    k-column input rows and the projection follows; outputs and every gradient equal the project-first order (1e-4 relative)."""
    from mi355x_graph.nn import GATConv
    n = 1200
    src, dst = hubby_graph(n, 40 * n, seed=29)
    loops = np.arange(n)
    g = mk(n, n, np.concatenate([src, loops]), np.concatenate([dst, loops]))
    torch.manual_seed(1)
    conv = GATConv(k, f, 1, 0.0, 0.0, 0.2, residual=residual).to(DEV)
    x0 = torch.randn(n, k, device=DEV)
    w = torch.randn(n, 1, f, device=DEV)
    widths, res = [], []
    real = ops.gat_fused
    monkeypatch.setattr(ops, "gat_fused", lambda g_, feat, *a, **kw: (widths.append(int(feat.shape[-1])), real(g_, feat, *a, **kw))[1])
    for first in ("1", "0"):
        monkeypatch.setattr(mgx_config, "GAT_AGG_FIRST", first == "1")
        x = x0.clone().requires_grad_(True)
        conv.zero_grad()
        y = conv(g, x)
        assert y.shape == (n, 1, f)
        (y * w).sum().backward()
        res.append([y.detach(), x.grad] + [p.grad.clone() for p in conv.parameters()])
    assert widths == [k, f]
    for a_, b_ in zip(*res):
        assert float((a_ - b_).abs().max()) < RTOL * float(b_.abs().max()) + 1e-6, (a_.shape, float((a_ - b_).abs().max()), float(b_.abs().max()))
    widths.clear()
    GATConv(f, k, 1).to(DEV)(g, torch.randn(n, f, device=DEV))  # a narrowing layer keeps the usual order
    assert widths == [k]


def pair_keep(seed, dst, src, rank, p):
    """The tile walks' attn_drop mask (csrc/gatfused.hip gat_hash, csrc/gat_tile.inc gat_pair_keep): a 32-bit multiply-xorshift
    mixer of (seed, source, destination + rank * 0x9E3779B1), kept when the hash is >= p * 2^32; in int64 arithmetic masked to 32 bits."""
    M = 0xFFFFFFFF
    x = (src ^ (seed & M)) & M
    x = x ^ (x >> 16)
    x = (x * 0x7feb352d) & M
    x = x ^ (x >> 15)
    x = x ^ ((dst + rank * 0x9E3779B1 + ((seed >> 32) & M)) & M)
    x = (x * 0x846ca68b) & M
    x = x ^ (x >> 16)
    return x >= min(int(p * 4294967296.0), 4294967295)


def rank_among_parallel_edges(src, dst, n):
    """rank[e] = how many edges with a smaller id join the same (source, destination) pair."""
    key = dst * n + src
    order = np.argsort(key, kind="stable")
    ks = key[order]
    first = np.concatenate([[True], ks[1:] != ks[:-1]])
    run0 = np.maximum.accumulate(np.where(first, np.arange(ks.shape[0]), 0))
    rank = np.empty_like(key)
    rank[order] = np.arange(ks.shape[0]) - run0
    return rank


@pytest.mark.parametrize("Fd,p", [(16, 0.0), (16, 0.3), (8, 0.0), (4, 0.25), (12, 0.4)])
def test_tile_walks_match_the_row_kernels(Fd, p, monkeypatch):
    """One head of up to 16 columns on a graph with dense neighbourhoods (main_dgl_reddit_gat.py on reddit): the three walks run
    as LDS-staged tile kernels (csrc/gat_tile.inc) -- same outputs and gradients (1e-4 relative) as the row kernels at p = 0, and
    with attn_drop as the unfused operators with the tile walks' mask (one bit per (seed, destination, source, rank among parallel
    edges)) applied to the attention.  A MULTIGRAPH (up to ~10 parallel edges per pair) with hub rows split by the plan on both sides."""
    n = 2500
    src, dst = random_graph(n, n, 600000, seed=31)
    rng = np.random.default_rng(5)
    hub_in, hub_out = rng.integers(0, n, 21000), rng.integers(0, n, 9000)   # > 2,048 edges: split work items on both sides
    src = np.concatenate([src, hub_in, np.full(9000, 17)]).astype(np.int64)
    dst = np.concatenate([dst, np.full(21000, 23), hub_out]).astype(np.int64)
    rank = rank_among_parallel_edges(src, dst, n)
    assert 2 <= rank.max() < 128
    feat0 = torch.randn(n, 1, Fd, device=DEV)
    el0, er0 = torch.randn(n, 1, 1, device=DEV), torch.randn(n, 1, 1, device=DEV)
    w = torch.randn(n, 1, Fd, device=DEV)
    monkeypatch.setenv("MGX_TILE", "1")
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MGX_GAT_TILE", mode)
        g = mk(n, n, src, dst)
        feat, el, er = (t.clone().requires_grad_(True) for t in (feat0, el0, er0))
        torch.manual_seed(7)
        ops.GATFused._calls = 0  # the mask's seed = (torch seed, call counter)
        if mode == "1" or p == 0.0:
            out = ops.gat_fused(g, feat, el, er, 0.2, p, True)
        else:  # the reference under attn_drop: unfused operators, the mask applied to the attention
            keep = pair_keep(7, T(dst), T(src), T(rank), p).float().view(-1, 1, 1) / (1.0 - p)
            out = ops.gspmm(g, "mul", "sum", feat, unfused_attention(g, el, er, 0.2) * keep)
        held = g._index.csc()._tile_plan
        assert (isinstance(held, dict) and "gat" in held) == (mode == "1"), "the tile walk was %staken" % ("not " if mode == "1" else "")
        (out * w).sum().backward()
        res[mode] = [out.detach(), feat.grad, el.grad, er.grad]
        if mode == "1":
            st = held["gat"].stats
            assert st["parallel_edges"] is True and st["max_pair_rank"] == int(rank.max()) and held["gat"].base.num_slots > 0
            ops.GATFused._calls = 0
            again = ops.gat_fused(g, feat0, el0, er0, 0.2, p, True)
            assert torch.equal(again, out.detach())  # no atomics: bitwise reruns
    for name, a_, b_ in zip(("out", "d_feat", "d_el", "d_er"), res["1"], res["0"]):
        assert float((a_ - b_).abs().max()) < RTOL * float(b_.abs().max()) + 1e-6, (name, float((a_ - b_).abs().max()), float(b_.abs().max()))


def test_tile_walks_with_more_than_128_parallel_edges_on_a_pair(monkeypatch):
    """The rank field of a tile plan's entries is 7 bits: a pair with 200 parallel edges re-uses mask keys (rank mod 128).  Both
    CSRs see the same multiset of bits for the pair, so forward and backward stay consistent: the same numbers as the unfused
    operators with that mask, at 1e-4."""
    n = 2000
    src, dst = random_graph(n, n, 500000, seed=37)
    src = np.concatenate([src, np.full(200, 3)]).astype(np.int64)   # 200 parallel edges 3 -> 5
    dst = np.concatenate([dst, np.full(200, 5)]).astype(np.int64)
    rank = rank_among_parallel_edges(src, dst, n)
    assert rank.max() >= 199
    monkeypatch.setenv("MGX_TILE", "1")
    monkeypatch.setenv("MGX_GAT_TILE", "1")
    g = mk(n, n, src, dst)
    p = 0.35
    el, er = torch.randn(n, 1, 1, device=DEV), torch.randn(n, 1, 1, device=DEV)
    w = torch.randn(n, 1, 16, device=DEV)
    res = []
    for tile in (True, False):
        feat = torch.randn(n, 1, 16, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)).requires_grad_(True)
        elg, erg = el.clone().requires_grad_(True), er.clone().requires_grad_(True)
        torch.manual_seed(11)
        ops.GATFused._calls = 0
        if tile:
            out = ops.gat_fused(g, feat, elg, erg, 0.2, p, True)
            assert g._index.csc()._tile_plan["gat"].stats["max_pair_rank"] >= 199
        else:
            keep = pair_keep(11, T(dst), T(src), T(rank % 128), p).float().view(-1, 1, 1) / (1.0 - p)
            out = ops.gspmm(g, "mul", "sum", feat, unfused_attention(g, elg, erg, 0.2) * keep)
        (out * w).sum().backward()
        res.append([out.detach(), feat.grad, elg.grad, erg.grad])
    for name, a_, b_ in zip(("out", "d_feat", "d_el", "d_er"), *res):
        assert float((a_ - b_).abs().max()) < RTOL * float(b_.abs().max()) + 1e-6, (name, float((a_ - b_).abs().max()), float(b_.abs().max()))


@pytest.mark.parametrize("mixed", [False, True])
def test_tile_walks_on_a_bipartite_block_and_mixed_with_the_row_kernels(mixed, monkeypatch):
    """A dense bipartite block (3,000 sources -> 1,000 destinations): the tile walks against the unfused operators; `mixed`: the
    out-CSR has no tile plan (as on a graph whose out-degrees fall below the tile rule), so the forward is a tile walk and the
    backward the row kernels -- they share the per-node statistics."""
    n_src, n_dst = 3000, 1000
    src, dst = random_graph(n_src, n_dst, 400000, seed=41)
    monkeypatch.setenv("MGX_TILE", "1")
    monkeypatch.setenv("MGX_GAT_TILE", "1")
    g = mk(n_src, n_dst, src, dst)
    if mixed:
        csr = g._index.csr()
        orig = type(csr).gat_tile_plan
        monkeypatch.setattr(type(csr), "gat_tile_plan", lambda self, F: None if self is csr else orig(self, F))
    feat = torch.randn(n_src, 1, 16, device=DEV)
    el, er = torch.randn(n_src, 1, 1, device=DEV), torch.randn(n_dst, 1, 1, device=DEV)
    w = torch.randn(n_dst, 1, 16, device=DEV)
    res = []
    for fused in (True, False):
        f_, l_, r_ = (t.clone().requires_grad_(True) for t in (feat, el, er))
        out = ops.gat_fused(g, f_, l_, r_, 0.2, 0.0, True) if fused else unfused(g, f_, l_, r_, 0.2)
        (out * w).sum().backward()
        res.append([out.detach(), f_.grad, l_.grad, r_.grad])
    assert "gat" in g._index.csc()._tile_plan
    for name, a_, b_ in zip(("out", "d_feat", "d_el", "d_er"), *res):
        assert float((a_ - b_).abs().max()) < RTOL * float(b_.abs().max()) + 1e-6, (name, float((a_ - b_).abs().max()), float(b_.abs().max()))


def test_tile_walks_fuzz_over_graph_shapes(monkeypatch):
    """The tile forms of the three walks forced onto 14 random graphs -- a handful of nodes to a few thousand, bipartite, sparse to
    dense, hub rows and hub sources, rows without in-edges, parallel edges -- for F in {4, 8, 12, 16}, with and without attn_drop:
    outputs and all gradients against the unfused operators (with the walks' mask applied under dropout)."""
    monkeypatch.setenv("MGX_TILE", "1")
    monkeypatch.setenv("MGX_GAT_TILE", "1")
    rng = np.random.default_rng(77)
    for trial in range(14):
        n_src = int(rng.choice([5, 40, 300, 2000, 3500]))
        n_dst = n_src if trial % 3 else int(rng.choice([7, 120, 900]))
        nnz = int(rng.choice([1, 60, 5000, 60000, 300000]))
        src = rng.integers(0, n_src, nnz)
        dst = rng.integers(0, max(1, n_dst - n_dst // 5), nnz)            # the last fifth of the rows receives nothing
        if trial % 4 == 2:
            src = np.concatenate([src, rng.integers(0, n_src, 4000), np.full(2500, n_src - 1)])
            dst = np.concatenate([dst, np.full(4000, 1 % n_dst), rng.integers(0, n_dst, 2500)])
        src, dst = src.astype(np.int64), dst.astype(np.int64)
        Fd = int(rng.choice([4, 8, 12, 16]))
        p = float(rng.choice([0.0, 0.35]))
        g = mk(n_src, n_dst, src, dst)
        feat = torch.randn(n_src, 1, Fd, device=DEV)
        el, er = torch.randn(n_src, 1, 1, device=DEV), torch.randn(n_dst, 1, 1, device=DEV)
        w = torch.randn(n_dst, 1, Fd, device=DEV)
        rank = np.zeros_like(src)
        if p > 0.0:  # rank among parallel edges, keyed on (destination, source): any multiple of n_src as the row stride
            key = dst * n_src + src
            order = np.argsort(key, kind="stable")
            ks = key[order]
            first = np.concatenate([[True], ks[1:] != ks[:-1]])
            run0 = np.maximum.accumulate(np.where(first, np.arange(ks.shape[0]), 0))
            rank[order] = np.arange(ks.shape[0]) - run0
        res = []
        for tile in (True, False):
            f_, l_, r_ = (t.clone().requires_grad_(True) for t in (feat, el, er))
            torch.manual_seed(5)
            ops.GATFused._calls = 0
            if tile:
                out = ops.gat_fused(g, f_, l_, r_, 0.2, p, True)
                assert "gat" in g._index.csc()._tile_plan, (trial, "tile walk not taken")
            else:
                a = unfused_attention(g, l_, r_, 0.2)
                if p > 0.0:
                    a = a * (pair_keep(5, T(dst), T(src), T(rank % 128), p).float().view(-1, 1, 1) / (1.0 - p))
                out = ops.gspmm(g, "mul", "sum", f_, a)
            (out * w).sum().backward()
            res.append([out.detach(), f_.grad, l_.grad, r_.grad])
        for name, a_, b_ in zip(("out", "d_feat", "d_el", "d_er"), *res):
            tol = RTOL * float(b_.abs().max()) + 1e-6
            assert float((a_ - b_).abs().max()) < tol, (trial, n_src, n_dst, src.shape[0], Fd, p, name, float((a_ - b_).abs().max()), float(b_.abs().max()))


def test_fused_rejects_mismatched_rows():
    n = 100
    src, dst = random_graph(n, n, 1000, seed=1)
    g = mk(n, n, src, dst)
    feat = torch.randn(n, 2, 8, device=DEV)
    with pytest.raises(mg.DGLError):
        ops.gat_fused(g, feat, torch.randn(n - 1, 2, 1, device=DEV), torch.randn(n, 2, 1, device=DEV))
    assert ops.gat_fused_supported(g, torch.randn(n, 1, 41, device=DEV))       # one head of any width: ragged windows
    assert not ops.gat_fused_supported(g, torch.randn(n, 2, 41, device=DEV))   # several ragged heads: the unfused path
    assert not ops.gat_fused_supported(g, torch.randn(n, 1, 3, device=DEV))


@pytest.mark.parametrize("H,F", [(1, 16), (1, 41), (2, 8), (1, 24), (4, 4)])
def test_packed_gather_operands_change_nothing(H, F, monkeypatch):
    """Narrow layers gather [feat | el] and [d_out | er, m, 1/s, t] rows packed into whole lines (one L2 request per edge
    instead of two, mgx_gat_fused_pack_workspace); config.GAT_PACK = False gathers the separate arrays: same bits."""
    from mi355x_graph import _lib
    n = 4000
    src, dst = random_graph(n, n, 60000, seed=H * 100 + F, skew=True)
    g = mg.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=n).int().formats(["csr", "csc"]).to(DEV)
    assert _lib.lib().mgx_gat_fused_pack_workspace(n, n, H, F) > 0
    assert _lib.lib().mgx_gat_fused_pack_workspace(n, n, 8, 16) == 0  # 8 x 16: the terms do not fit the row's lines
    assert _lib.lib().mgx_gat_fused_pack_workspace(n, n, 1, 8) == 0   # 8 + 1 floats in a 128-byte row: too much padding
    torch.manual_seed(1)
    feat0 = torch.randn(n, H, F, device=DEV)
    el0, er0 = torch.randn(n, H, device=DEV), torch.randn(n, H, device=DEV)
    w = torch.randn(n, H, F, device=DEV)
    res = []
    for nopack in (False, True):
        if nopack:
            monkeypatch.setattr(mgx_config, "GAT_PACK", False)
        feat, el, er = (t.clone().requires_grad_(True) for t in (feat0, el0, er0))
        out = ops.gat_fused(g, feat, el, er, 0.2)
        (out * w).sum().backward()
        res.append((out.detach(), feat.grad, el.grad, er.grad))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,H,Fd,p", [(900, 8, 16, 0.0), (700, 4, 32, 0.0), (600, 2, 64, 0.0), (500, 16, 8, 0.0), (900, 8, 16, 0.4)])
def test_el_formed_in_the_kernel_matches_the_gathered_el(oracle, n, H, Fd, p):
    """Round 3: with attn_l passed, multi-head layers whose rows are not packed (8 x 16, BASELINE config 3) form
    el[u,h] = <feat[u,h,:], attn_l[h,:]> from the gathered row instead of gathering it.  Same layer: forward against the oracle's
    composition on el = (feat * attn_l).sum(-1), gradients (through el's own producer) against the unfused operators, the three
    walks agree with each other under dropout (adjointness), reruns bitwise."""
    src, dst = hubby_graph(n, 40 * n, seed=H * 31 + Fd)
    E = src.shape[0]
    g = mk(n, n, src, dst)
    torch.manual_seed(H * Fd)
    feat0 = torch.randn(n, H, Fd, device=DEV)
    attn0 = torch.randn(1, H, Fd, device=DEV)
    er0 = torch.randn(n, H, 1, device=DEV) * 2
    if p == 0.0:
        with torch.no_grad():
            el = (feat0 * attn0).sum(-1, keepdim=True)
            out = ops.gat_fused(g, feat0, el, er0, 0.2, 0.0, True, attn_l=attn0)
            gathered = ops.gat_fused(g, feat0, el, er0, 0.2, 0.0, True)
        ip, ix, ei = oracle.coo_to_csr(n, dst, src)
        z = oracle.sddmm(src, dst, "add", el.cpu().numpy(), er0.cpu().numpy())
        z = np.where(z > 0, z, 0.2 * z).astype(np.float32)
        a = oracle.edge_softmax_fwd(ip, ei, z.reshape(E, H))
        featn = feat0.cpu().numpy()
        ref = oracle.spmm(ip, ix, ei, "mul", "sum", featn, a.reshape(E, H, 1))
        row_scale = oracle.spmm(ip, ix, ei, "mul", "sum", np.abs(featn), a.reshape(E, H, 1)).astype(np.float64)
        for got in (out, gathered):
            err = np.abs(got.cpu().numpy().astype(np.float64) - ref.astype(np.float64))
            assert not (err > RTOL * row_scale + 1e-30).any(), float((err / (row_scale + 1e-30)).max())
        res = []
        for in_kernel in (True, False):
            f_, a_, r_ = (t.clone().requires_grad_(True) for t in (feat0, attn0, er0))
            el_ = (f_ * a_).sum(-1, keepdim=True)
            o = ops.gat_fused(g, f_, el_, r_, 0.2, 0.0, True, attn_l=a_) if in_kernel else unfused(g, f_, el_, r_, 0.2)
            (o * torch.sin(torch.arange(o.numel(), device=DEV).view_as(o).float())).sum().backward()
            res.append([f_.grad, a_.grad, r_.grad])
        for name, x, y in zip(("d_feat", "d_attn_l", "d_er"), *res):
            err, scale = float((x - y).abs().max()), float(y.abs().max())
            assert err < RTOL * max(scale, 1e-6), (name, err, scale)
        with torch.no_grad():
            assert torch.equal(out, ops.gat_fused(g, feat0, el, er0, 0.2, 0.0, True, attn_l=attn0))
    else:  # dropout: the map feat -> out is linear for fixed attention: <A x, y> = <x, A^T y> with the SAME mask in all walks
        from mi355x_graph.ops import GATFused
        torch.manual_seed(5)
        calls = GATFused._calls
        with torch.no_grad():
            el = (feat0 * attn0).sum(-1, keepdim=True)
        f_ = feat0.clone().requires_grad_(True)
        o = ops.gat_fused(g, f_, el, er0, 0.2, p, True, attn_l=attn0)
        y = torch.randn_like(o)
        (o * y).sum().backward()
        lhs = float((o.detach().double() * y.double()).sum())
        rhs = float((f_.grad.double() * feat0.double()).sum())  # <x, A^T y> with x = feat (el held fixed: it is its own input)
        assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0), (lhs, rhs)
        GATFused._calls = calls  # same seed -> same mask: bitwise the same output
        with torch.no_grad():
            assert torch.equal(o.detach(), ops.gat_fused(g, feat0, el, er0, 0.2, p, True, attn_l=attn0))
