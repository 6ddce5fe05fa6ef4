"""GPU suite: the short-row g-SpMM (spmm_rowgroup32_kernel, csrc/spmm.hip, round 4) -- one work item per lane group.

Chosen by the library for CSRs whose work items average fewer than 14 edges: the halo CSRs of a partition (dist.py), arxiv-
shaped graphs (kernel/dgl-new.py:61), batched molecules (main_dgl_molhiv_gcn.py:46).  Against the CPU oracle through the C ABI;
rows that are not split are summed in storage order, i.e. BIT-EXACT against the oracle's sequential fp32 sum."""
import numpy as np
import pytest
import torch

import mi355x_graph as mg
from mi355x_graph import _lib, ops, sparse

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def last_kernel():
    return _lib.lib().mgx_last_spmm_kernel().decode()


def expected_kernel(csr, D):
    """The host layer's policy (sparse.CsrView.spmm_plan_for): short AND even work items -> one item per lane group."""
    return "rowgroup32" if csr.short_rows(D) else "rowwave32"


def short_row_graph(n_src, n_dst, avg, seed, hubs=2):
    """Mostly short rows (Poisson around `avg`, many empty), a few hub rows beyond the split threshold, duplicates."""
    rng = np.random.default_rng(seed)
    deg = rng.poisson(avg, n_dst)
    deg[rng.integers(0, n_dst, max(1, n_dst // 10))] = 0
    for h in rng.integers(0, n_dst, hubs):
        deg[h] = int(rng.integers(300, 900))
    dst = np.repeat(np.arange(n_dst), deg)
    src = rng.integers(0, n_src, dst.shape[0])
    perm = rng.permutation(dst.shape[0])          # edge ids are not in CSR order
    return src[perm].astype(np.int64), dst[perm].astype(np.int64), deg


@pytest.mark.parametrize("D", [4, 8, 16, 32, 64, 100, 128])
@pytest.mark.parametrize("avg", [1.5, 3.4, 11.0])
def test_short_rows_copy_u_against_the_oracle(oracle, D, avg):
    n_src, n_dst = 5000, 7000
    src, dst, deg = short_row_graph(n_src, n_dst, avg, seed=int(D * 10 + avg))
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=torch.int32, device=DEV)
    rng = np.random.default_rng(D)
    X = rng.standard_normal((n_src, D)).astype(np.float32)
    ip, ix, ei = oracle.coo_to_csr(n_dst, dst, src)
    x = torch.from_numpy(X).to(DEV)
    small = deg <= 32                               # rows one lane group walks alone: summed in storage order
    if D % 4 == 0:                                  # the kernel itself, whatever the policy says about this graph (flag forced)
        g._index.csc()._short = {nb: True for nb in (2, 4, 8, 16, 32, 64)}
    want_kernel = expected_kernel(g._index.csc(), D)
    for red in ("sum", "mean"):
        out = ops.gspmm(g, "copy_lhs", red, x, None).cpu().numpy()
        assert last_kernel() == want_kernel, (last_kernel(), want_kernel)
        ref = oracle.spmm(ip, ix, ei, "copy_lhs", red, X, None)
        if want_kernel == "rowgroup32":
            assert np.array_equal(out[small], ref[small]), (D, avg, red)
        scale = oracle.spmm(ip, ix, ei, "copy_lhs", red, np.abs(X), None)
        assert bool((np.abs(out - ref) <= 1e-4 * scale + 1e-30).all())
        assert float(np.abs(out[deg == 0]).sum()) == 0.0
    # accumulate + dst_scale + bitwise rerun
    csc = g._index.csc()
    base = torch.from_numpy(rng.standard_normal((n_dst, D)).astype(np.float32)).to(DEV)
    sc = torch.from_numpy(rng.random(n_dst).astype(np.float32) + 0.5).to(DEV)
    acc = base.clone()
    sparse.gspmm_raw(csc, "copy_lhs", "sum", x, None, dst_scale=sc, accumulate_into=acc)
    assert last_kernel() == want_kernel
    want = base.cpu().numpy() + oracle.spmm(ip, ix, ei, "copy_lhs", "sum", X, None) * sc.cpu().numpy()[:, None]
    assert float(np.abs(acc.cpu().numpy() - want).max()) <= 1e-4 * max(1.0, float(np.abs(want).max()))
    acc2 = base.clone()
    sparse.gspmm_raw(csc, "copy_lhs", "sum", x, None, dst_scale=sc, accumulate_into=acc2)
    assert torch.equal(acc, acc2)


def test_short_rows_strided_operands_and_copy_e(oracle):
    n_src, n_dst, D = 6000, 130000, 64          # (two hub rows: a two-part plan needs schedule.MIN_SHORT_ITEMS short rows to be taken)
    src, dst, deg = short_row_graph(n_src, n_dst, 3.4, seed=5)
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=torch.int32, device=DEV)
    csc = g._index.csc()
    be = sparse.backend_for(csc.indptr)
    rng = np.random.default_rng(1)
    wide_in = torch.from_numpy(rng.standard_normal((n_src, 2 * D + 8)).astype(np.float32)).to(DEV)
    wide_out = torch.zeros(n_dst, 3 * D, device=DEV)
    assert csc.short_rows(D) and csc.short_rows(32)      # the policy's own answer: Poisson rows + two split hubs are short and even
    be.spmm_copy_u_strided(csc, "mean", wide_in[:, D:2 * D], wide_out[:, D:2 * D])
    assert last_kernel() == "rowgroup32"
    ip, ix, ei = oracle.coo_to_csr(n_dst, dst, src)
    ref = oracle.spmm(ip, ix, ei, "copy_lhs", "mean", wide_in[:, D:2 * D].contiguous().cpu().numpy(), None)
    got = wide_out.cpu().numpy()
    assert np.array_equal(got[:, D:2 * D][deg <= 16], ref[deg <= 16])
    assert float(np.abs(got[:, :D]).sum()) == 0.0 and float(np.abs(got[:, 2 * D:]).sum()) == 0.0   # nothing written beside the block
    be.spmm_copy_u_strided(csc, "sum", wide_in[:, D:2 * D], wide_out[:, D:2 * D], accumulate=True)
    ref2 = ref + oracle.spmm(ip, ix, ei, "copy_lhs", "sum", wide_in[:, D:2 * D].contiguous().cpu().numpy(), None)
    assert float(np.abs(wide_out[:, D:2 * D].cpu().numpy() - ref2).max()) <= 1e-4 * float(np.abs(ref2).max())
    # copy_e / sum: rows of an edge matrix addressed by edge id (main_dgl_molhiv_gcn.py:46 after the UDF message)
    E = rng.standard_normal((src.shape[0], 32)).astype(np.float32)
    out = ops.gspmm(g, "copy_rhs", "sum", None, torch.from_numpy(E).to(DEV)).cpu().numpy()
    assert last_kernel() == "rowgroup32"
    ref_e = oracle.spmm(ip, ix, ei, "copy_rhs", "sum", None, E)
    assert np.array_equal(out[deg <= 32], ref_e[deg <= 32])
    assert float(np.abs(out - ref_e).max()) <= 1e-4 * float(np.abs(ref_e).max())


def test_policy_long_rows_keep_the_row_per_wave_kernel_and_grads_flow(oracle):
    n = 4000
    rng = np.random.default_rng(3)
    src, dst = rng.integers(0, n, 40 * n), rng.integers(0, n, 40 * n)          # 40 edges per row: row-per-wave
    g = mg.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=n).int().to(DEV)
    x = torch.rand(n, 64, device=DEV)
    ops.gspmm(g, "copy_lhs", "sum", x, None)
    assert last_kernel() == "rowwave32"
    src2, dst2 = rng.integers(0, n, 5 * n), rng.integers(0, n, 5 * n)          # 5 edges per row, both directions short
    g2 = mg.graph((torch.from_numpy(src2), torch.from_numpy(dst2)), num_nodes=n).int().to(DEV)
    xr = x.clone().requires_grad_(True)
    out = ops.gspmm(g2, "copy_lhs", "mean", xr, None)
    assert last_kernel() == "rowgroup32"
    w = torch.rand(n, 64, device=DEV)
    (out * w).sum().backward()
    assert last_kernel() == "rowgroup32"                                         # the reversed graph is short-rowed too
    rp, rx, re_ = oracle.coo_to_csr(n, src2, dst2)
    ip = oracle.coo_to_csr(n, dst2, src2)[0]
    inv = (1.0 / np.maximum(np.diff(ip), 1)).astype(np.float32)
    want = oracle.spmm(rp, rx, re_, "copy_lhs", "sum", w.cpu().numpy() * inv[:, None], None)
    assert float(np.abs(xr.grad.cpu().numpy() - want).max()) <= 1e-4 * float(np.abs(want).max())


def test_skewed_graph_takes_a_two_part_plan(oracle):
    """Power-law rows (the arxiv shape): the short items on the lane-group kernel, hub chunks and long rows in the plan's `rest` on the
    wave-per-item kernel; every row written exactly once, whatever the reduce / accumulate / dst_scale / strides."""
    n_src, n_dst, D = 30000, 160000, 32
    rng = np.random.default_rng(11)
    deg = rng.poisson(3.4, n_dst).astype(np.int64)
    heavy = rng.integers(0, n_dst, n_dst // 50)
    deg[heavy] = np.minimum(40 * rng.zipf(1.6, heavy.shape[0]), 900)   # 2 % of the rows hold half of the edges, like the arxiv stand-in
    deg[rng.integers(0, n_dst, n_dst // 8)] = 0
    deg[17] = 1900                                   # split into 256-edge chunks: partial slots + fix-up, all in `rest`
    dst = np.repeat(np.arange(n_dst), deg)
    src = rng.integers(0, n_src, dst.shape[0])
    perm = rng.permutation(dst.shape[0])
    src, dst = src[perm], dst[perm]
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=torch.int32, device=DEV)
    csc = g._index.csc()
    plan, short = csc.spmm_plan_for(D)
    assert short and plan is not None and plan.rest is not None
    lens = (plan.item_end - plan.item_beg)
    assert int(lens.max()) <= 32 and bool((plan.item_row >= 0).all()) and plan.num_slots == 0
    rest = plan.rest
    assert rest.num_slots >= 8 and int((rest.item_row < 0).sum()) == rest.num_slots
    covered = torch.cat([plan.item_node, rest.item_node[rest.item_row >= 0], rest.hub_row]).long()
    assert covered.shape[0] == n_dst and bool((torch.bincount(covered, minlength=n_dst) == 1).all())
    owner = rest.slot_item.long()                    # slot s is written by item slot_item[s] of `rest`
    assert torch.equal(rest.item_row[owner].long(), -(torch.arange(rest.num_slots, device=DEV) + 1))
    X = rng.standard_normal((n_src, D)).astype(np.float32)
    x = torch.from_numpy(X).to(DEV)
    ip, ix, ei = oracle.coo_to_csr(n_dst, dst, src)
    for red in ("sum", "mean"):
        out = ops.gspmm(g, "copy_lhs", red, x, None).cpu().numpy()
        assert last_kernel() == "rowgroup32"
        ref = oracle.spmm(ip, ix, ei, "copy_lhs", red, X, None)
        assert np.array_equal(out[deg <= 32], ref[deg <= 32])     # a lane group adds in storage order
        scale = oracle.spmm(ip, ix, ei, "copy_lhs", "sum", np.abs(X), None)
        if red == "mean":
            scale = scale / np.maximum(deg, 1)[:, None]
        assert bool((np.abs(out - ref) <= 1e-4 * scale + 1e-30).all())
    be = sparse.backend_for(csc.indptr)
    sc = torch.from_numpy(rng.random(n_dst).astype(np.float32) + 0.5).to(DEV)
    base = torch.from_numpy(rng.standard_normal((n_dst, 2 * D)).astype(np.float32)).to(DEV)
    wide = base.clone()
    be.spmm_copy_u_strided(csc, "sum", x, wide[:, D:], accumulate=True, dst_scale=sc)
    assert last_kernel() == "rowgroup32"
    want = base[:, D:].cpu().numpy() + oracle.spmm(ip, ix, ei, "copy_lhs", "sum", X, None) * sc.cpu().numpy()[:, None]
    assert float(np.abs(wide[:, D:].cpu().numpy() - want).max()) <= 1e-4 * float(np.abs(want).max())
    assert torch.equal(wide[:, :D], base[:, :D])
    E = rng.standard_normal((src.shape[0], 16)).astype(np.float32)
    out_e = ops.gspmm(g, "copy_rhs", "sum", None, torch.from_numpy(E).to(DEV)).cpu().numpy()
    ref_e = oracle.spmm(ip, ix, ei, "copy_rhs", "sum", None, E)
    assert np.array_equal(out_e[deg <= 32], ref_e[deg <= 32])
    assert float(np.abs(out_e - ref_e).max()) <= 1e-4 * float(np.abs(ref_e).max())
    # a two-part plan is refused where it does not belong (max has no split form here)
    import ctypes
    out_m = torch.empty(n_dst, D, device=DEV)
    ws = torch.empty(rest.num_slots, D, device=DEV)
    P = sparse._ptr
    status = _lib.lib().mgx_spmm_csr(ctypes.byref(csc.c_struct()), ctypes.byref(plan.c_struct()), sparse.OP["copy_lhs"], sparse.REDUCE["max"],
                                     P(x), None, D, 0, D, None, None, None, None, P(out_m), None, None, P(ws), 2, None)
    assert status != 0 and b"rest" in _lib.lib().mgx_last_error()


def test_one_step_structures_decide_without_reading_lengths_back():
    """A sampled block / a batch lives for one step: its views carry the average row length (host-known) and _short_choice uses the
    average rule alone -- no split tables, no host reads (ns-sage-dgl.py's epoch went 0.40 -> 0.51 s with an analysis per block)."""
    n = 50000
    gen = torch.Generator().manual_seed(0)
    src, dst = torch.randint(0, n, (400000,), generator=gen), torch.randint(0, n, (400000,), generator=gen)
    g = mg.graph((src, dst), num_nodes=n).int().to(DEV)
    from mi355x_graph import sampling
    blocks = sampling.MultiLayerNeighborSampler([10, 5]).sample_blocks(g, torch.arange(2000, device=DEV))
    for b in blocks:
        assert b._index.ephemeral
        for view in (b._index.csc(), b._index.csr()):
            assert isinstance(view.short_hint, float) and abs(view.short_hint - view.nnz / view.num_rows) < 1e-9
            want = view.short_hint < 16.0
            assert view.short_rows(16) == want and "split" not in str(list(view._short))    # D = 16: 16 lane groups, average below 16
            assert view.short_rows(128) == (view.short_hint < 6.0)                         # D = 128: 2 lane groups, 3 edges each
        x = torch.rand(b.number_of_src_nodes(), 16, device=DEV)
        ops.gspmm(b, "copy_lhs", "mean", x, None)
        assert last_kernel() == ("rowgroup32" if b._index.csc().short_hint < 16.0 else "rowwave32")
    moved = blocks[0]._index.csc().to(torch.device("cpu")).to(DEV)
    assert moved.short_hint == blocks[0]._index.csc().short_hint


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_two_part_plans_fuzz_against_the_oracle(oracle, seed):
    """Random skewed multigraphs, every width class, forced two-part plans at several limits (also where the policy would refuse them):
    copy_u and copy_e, sum and mean, plain / accumulate / dst_scale -- each output element within 1e-4 of the sum of |terms|."""
    from mi355x_graph import schedule
    rng = np.random.default_rng(100 + seed)
    n_src, n_dst = int(rng.integers(500, 6000)), int(rng.integers(500, 9000))
    deg = np.minimum(rng.zipf(1.5 + 0.3 * rng.random(), n_dst), 1500).astype(np.int64)
    deg[rng.integers(0, n_dst, n_dst // 5)] = 0
    dst = np.repeat(np.arange(n_dst), deg)
    src = rng.integers(0, n_src, dst.shape[0])
    perm = rng.permutation(dst.shape[0])
    src, dst = src[perm], dst[perm]
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=torch.int32, device=DEV)
    csc = g._index.csc()
    ip, ix, ei = oracle.coo_to_csr(n_dst, dst, src)
    be = sparse.backend_for(csc.indptr)
    for D in (4, 16, 32, 64, 100, 128):
        limit = int(rng.choice([1, 5, 16, 32]))
        split = schedule.split_short_items(csc, csc.plan(), any_share=True, limit=limit)
        if split is None:          # no item that short
            continue
        plan = split[0]
        csc._short = {nb: (plan if plan is not None and plan.rest is not None else True) for nb in (2, 4, 8, 16, 32, 64)}
        X = rng.standard_normal((n_src, D)).astype(np.float32)
        x = torch.from_numpy(X).to(DEV)
        absX = np.abs(X)
        for red in ("sum", "mean"):
            out = ops.gspmm(g, "copy_lhs", red, x, None).cpu().numpy()
            assert last_kernel() == "rowgroup32", (D, limit)
            ref = oracle.spmm(ip, ix, ei, "copy_lhs", red, X, None)
            bound = oracle.spmm(ip, ix, ei, "copy_lhs", red, absX, None)
            assert bool((np.abs(out - ref) <= 1e-4 * bound + 1e-30).all()), (D, limit, red)
        sc = (rng.random(n_dst).astype(np.float32) + 0.5)
        base = rng.standard_normal((n_dst, D)).astype(np.float32)
        acc = torch.from_numpy(base).to(DEV)
        sparse.gspmm_raw(csc, "copy_lhs", "sum", x, None, dst_scale=torch.from_numpy(sc).to(DEV), accumulate_into=acc)
        ref = base + oracle.spmm(ip, ix, ei, "copy_lhs", "sum", X, None) * sc[:, None]
        bound = np.abs(base) + oracle.spmm(ip, ix, ei, "copy_lhs", "sum", absX, None) * sc[:, None]
        assert bool((np.abs(acc.cpu().numpy() - ref) <= 1e-4 * bound + 1e-30).all()), (D, limit, "accumulate")
        if D <= 32:
            E = rng.standard_normal((src.shape[0], D)).astype(np.float32)
            out = ops.gspmm(g, "copy_rhs", "sum", None, torch.from_numpy(E).to(DEV)).cpu().numpy()
            ref = oracle.spmm(ip, ix, ei, "copy_rhs", "sum", None, E)
            bound = oracle.spmm(ip, ix, ei, "copy_rhs", "sum", None, np.abs(E))
            assert bool((np.abs(out - ref) <= 1e-4 * bound + 1e-30).all()), (D, limit, "copy_e")
        csc._short = {}
