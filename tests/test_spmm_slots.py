"""csrc/spmm_slots.inc (round 5): copy_u / sum|mean over MOSTLY-ZERO rows of 64 columns -- the dropout(relu(.)) input of a hidden GraphSAGE
layer (main_dgl_product_sage.py:93-96) -- gathered as one 128-byte slot per source row instead of the 256-byte row.

  mgx_rows_slots_pack      against a plain restatement of the slot format (include/mi355x_graph.h): bit for bit
  mgx_spmm_copy_u_slots    against the CPU oracle through the C ABI (the order of additions inside a row differs from the dense kernels':
                           1e-4 of the row's sum of magnitudes, as for every split row), on a two-part plan with hub rows and on a single
                           schedule, with mean / dst_scale / accumulate / strided operands, at every density incl. rows above 24 non-zeros
  ops.SageMeanCatFn        the layer with and without the slot form: same output, same gradients (fp32 rounding)"""
import ctypes

import numpy as np
import pytest
import torch

import mi355x_graph as mg
from mi355x_graph import _lib, config as mgx_config, ops, sparse

DEV = "cuda:0"


def _reference_slots(x):
    """[n, 32] uint32 slots of a host matrix [n, 64], and the number of rows above 24 non-zeros."""
    n = x.shape[0]
    out = np.zeros((n, 32), np.uint32)
    order = sorted(range(64), key=lambda col: (col % 4) * 16 + col // 4)   # (component, lane) of a 16-lane x float4 read
    over = 0
    xb = x.view(np.uint32)
    for r in range(n):
        nz = [c for c in order if x[r, c] != 0]
        meta = [[64, 64, 64] for _ in range(8)]
        flag = 0
        if len(nz) > 24:
            flag, over = 255, over + 1
        else:
            for j, c in enumerate(nz):
                meta[j // 3][j % 3] = c
                out[r, 4 * (j // 3) + 1 + j % 3] = xb[r, c]
        for i in range(8):
            out[r, 4 * i] = meta[i][0] | (meta[i][1] << 8) | (meta[i][2] << 16) | (flag << 24)
    return out, over


@pytest.mark.gpu
@pytest.mark.parametrize("density", [0.0, 0.1, 0.25, 0.5, 1.0])
def test_pack_follows_the_slot_format(density):
    be = sparse.backend_for(torch.empty(1, device=DEV))
    rng = np.random.default_rng(int(density * 100) + 3)
    for n in (1, 5, 1003):
        x = (rng.standard_normal((n, 64)) * (rng.random((n, 64)) < density)).astype(np.float32)
        if n > 4 and 0 < density < 1:
            x[2] = 0.0                                        # an empty row
            x[3, :25] = 1.5                                   # 25 non-zeros at least: an overflow row
            x[4] = 0.0
            x[4, [0, 5, 63]] = [-0.0, 2.0, -3.0]              # -0.0 is zero: not stored
        want, over = _reference_slots(x)
        wide = torch.zeros(n, 128, device=DEV)
        wide[:, :64] = torch.from_numpy(x).to(DEV)
        for xt in (torch.from_numpy(x).to(DEV), wide[:, :64]):  # contiguous rows and the left half of a layer's [h | neigh] buffer
            assert be.rows_slots_supported(xt)
            slots, ovf = be.rows_slots_pack(xt)
            assert np.array_equal(slots.cpu().numpy().view(np.uint32), want), (density, n)
            assert int(ovf) == over
    assert not be.rows_slots_supported(torch.zeros(8, 32, device=DEV)) and not be.rows_slots_supported(torch.zeros(8, 64, device=DEV)[:, ::2])
    with pytest.raises(_lib.DGLError):
        be.rows_slots_pack(torch.zeros(8, 32, device=DEV))


@pytest.mark.gpu
def test_the_layer_gemm_writes_the_same_slots_as_the_pack_pass():
    """mgx_rows_gemm_relu_dropout with a slots operand: byte for byte mgx_rows_slots_pack of the activation it stores, overflow rows counted."""
    be = sparse.backend_for(torch.empty(1, device=DEV))
    n, K, M = 50021, 128, 64
    torch.manual_seed(3)
    a = torch.randn(n, K, device=DEV)
    w = torch.randn(M, K, device=DEV) * 0.1
    b = torch.randn(M, device=DEV) * 0.1
    for p in (0.5, 0.05):                                   # 25 % non-zero: a handful of rows above 24; 48 %: most of them
        wide = torch.zeros(n, 2 * M, device=DEV)
        slots = torch.full((n, 32), -1, dtype=torch.int32, device=DEV)
        ovf = torch.zeros(1, dtype=torch.int64, device=DEV)
        y, mask = be.rows_gemm_relu_dropout(a, w, True, b, p, 12345, 777, out=wide[:, :M], slots=slots, overflow=ovf)
        plain = be.rows_gemm_relu_dropout(a, w, True, b, p, 12345, 777)
        assert torch.equal(y, plain[0]) and torch.equal(mask, plain[1])            # the activation itself is unchanged
        want, want_ovf = be.rows_slots_pack(y)
        assert torch.equal(slots, want) and int(ovf) == int(want_ovf)
        assert (int(ovf) > n // 2) == (p < 0.1)


@pytest.mark.gpu
def test_relu_dropout_backward_writes_scaled_slots():
    """mgx_relu_dropout_bwd_slots: the gradient as mgx_relu_dropout_bwd_strided forms it, and its rows times a row factor as slots -- byte
    for byte mgx_rows_slots_pack(result, row_scale)."""
    be = sparse.backend_for(torch.empty(1, device=DEV))
    n = 40003
    torch.manual_seed(4)
    x = torch.randn(n, 64, device=DEV)
    y, mask = be.relu_dropout_fwd(x, 0.5, 99, 1234)
    up = torch.randn(n, 128, device=DEV)[:, 64:]                 # the upstream gradient: a column block of a wider matrix
    sc = torch.rand(n, device=DEV) + 0.1
    plain = be.relu_dropout_bwd(up, mask, 0.5)
    wide = torch.zeros(n, 128, device=DEV)
    dx, slots, ovf = be.relu_dropout_bwd_slots(up, mask, 0.5, wide[:, :64], row_scale=sc)
    assert torch.equal(dx, plain) and torch.equal(wide[:, 64:], torch.zeros_like(wide[:, 64:]))
    want, want_ovf = be.rows_slots_pack(plain, row_scale=sc)
    assert torch.equal(slots, want) and int(ovf) == int(want_ovf)
    dx2, slots2, _ = be.relu_dropout_bwd_slots(up, mask, 0.5, torch.empty(n, 64, device=DEV))
    assert torch.equal(dx2, plain) and torch.equal(slots2, be.rows_slots_pack(plain)[0])


def _skewed_graph(n_src, n_dst, seed):
    """Mostly short rows, 2 % of the rows holding half of the edges, two rows beyond the split threshold: the policy's two-part plan."""
    rng = np.random.default_rng(seed)
    deg = rng.poisson(3.4, n_dst).astype(np.int64)
    heavy = rng.integers(0, n_dst, n_dst // 50)
    deg[heavy] = np.minimum(40 * rng.zipf(1.6, heavy.shape[0]), 900)
    deg[rng.integers(0, n_dst, n_dst // 8)] = 0
    deg[[7, 4001]] = [2300, 700]                                                   # hub rows: split into 256-edge chunks
    dst = np.repeat(np.arange(n_dst), deg)
    src = rng.integers(0, n_src, dst.shape[0])
    perm = rng.permutation(dst.shape[0])
    return src[perm], dst[perm], deg


@pytest.mark.gpu
@pytest.mark.parametrize("density", [0.05, 0.25, 0.6])
def test_copy_u_over_slots_against_the_oracle(oracle, density):
    n_src, n_dst, D = 30000, 160000, 64
    src, dst, deg = _skewed_graph(n_src, n_dst, seed=int(density * 100))
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=torch.int32, device=DEV)
    csc = g._index.csc()
    be = sparse.backend_for(csc.indptr)
    rng = np.random.default_rng(5)
    X = (rng.standard_normal((n_src, D)) * (rng.random((n_src, D)) < density)).astype(np.float32)
    X[11] = rng.standard_normal(D)                                                  # a dense row among the sources
    ip, ix, ei = oracle.coo_to_csr(n_dst, dst, src)
    absum = oracle.spmm(ip, ix, ei, "copy_lhs", "sum", np.abs(X), None)
    wide_in = torch.zeros(n_src, 2 * D, device=DEV)
    wide_in[:, :D] = torch.from_numpy(X).to(DEV)
    x = wide_in[:, :D]
    slots, ovf = be.rows_slots_pack(x)
    assert (int(ovf) > n_src // 2) == (density > 0.5)
    plan, short = csc.spmm_plan_for(D)
    assert plan is not None and short and plan.rest is not None and plan.rest.num_slots >= 9   # a two-part plan with split rows
    for red in ("sum", "mean"):
        out = torch.full((n_dst, D), float("nan"), device=DEV)
        be.spmm_copy_u_strided(csc, red, x, out, slots=slots)
        assert _lib.lib().mgx_last_spmm_kernel().decode() == "slots"
        ref = oracle.spmm(ip, ix, ei, "copy_lhs", red, X, None)
        scale = absum / np.maximum(deg, 1)[:, None] if red == "mean" else absum
        assert bool((np.abs(out.cpu().numpy() - ref) <= 1e-4 * scale + 1e-30).all()), (density, red)
        dense = torch.empty_like(out)
        be.spmm_copy_u_strided(csc, red, x, dense)
        assert _lib.lib().mgx_last_spmm_kernel().decode() != "slots"
        assert float((out - dense).abs().max()) <= 2e-6 * float(dense.abs().max())
    # accumulate + dst_scale into the right half of a wider matrix: what a partition's aggregation asks for
    sc = torch.from_numpy(rng.random(n_dst).astype(np.float32) + 0.5).to(DEV)
    base = torch.from_numpy(rng.standard_normal((n_dst, 2 * D)).astype(np.float32)).to(DEV)
    wide = base.clone()
    be.spmm_copy_u_strided(csc, "sum", x, wide[:, D:], accumulate=True, dst_scale=sc, slots=slots)
    want = base[:, D:].cpu().numpy() + oracle.spmm(ip, ix, ei, "copy_lhs", "sum", X, None) * sc.cpu().numpy()[:, None]
    assert bool((np.abs(wide[:, D:].cpu().numpy() - want) <= 1e-4 * (absum * sc.cpu().numpy()[:, None] + np.abs(base[:, D:].cpu().numpy())) + 1e-30).all())
    assert torch.equal(wide[:, :D], base[:, :D])
    # slots that carry a row factor (the reversed aggregation of a mean layer: A^T (D^-1 dy)): out = sum_u sc_src[u] * x[u]; the kernel
    # applies the factor itself only to the rows it reads from the dense matrix (those above 24 non-zeros)
    sc_src = torch.from_numpy(rng.random(n_src).astype(np.float32) + 0.25).to(DEV)
    scaled_slots, ovf2 = be.rows_slots_pack(x, row_scale=sc_src)
    assert int(ovf2) == int(ovf)
    want_slots, _ = be.rows_slots_pack((x * sc_src.view(-1, 1)).contiguous())
    keep = (want_slots[:, 0].view(torch.int32) >> 24) == 0                      # (overflow rows hold no values either way)
    assert torch.equal(scaled_slots[keep], want_slots[keep])
    out = torch.empty((n_dst, D), device=DEV)
    be.spmm_copy_u_strided(csc, "sum", x, out, slots=scaled_slots, src_scale=sc_src)
    assert _lib.lib().mgx_last_spmm_kernel().decode() == "slots"
    Xs = X * sc_src.cpu().numpy()[:, None]
    ref = oracle.spmm(ip, ix, ei, "copy_lhs", "sum", Xs, None)
    assert bool((np.abs(out.cpu().numpy() - ref) <= 1e-4 * oracle.spmm(ip, ix, ei, "copy_lhs", "sum", np.abs(Xs), None) + 1e-30).all())
    with pytest.raises(_lib.DGLError):
        be.spmm_copy_u_strided(csc, "sum", x, out, src_scale=sc_src)             # a factor without slots packed with it
    # ONE schedule (no short part): every item on the slot kernel, hub rows through the partial slots and the fix-up
    csc._short = {nb: None for nb in (2, 4, 8, 16, 32, 64)}
    plan1, short1 = csc.spmm_plan_for(D)
    assert not short1 and plan1 is not None and plan1.rest is None
    out = torch.empty((n_dst, D), device=DEV)
    be.spmm_copy_u_strided(csc, "sum", x, out, slots=slots)
    assert _lib.lib().mgx_last_spmm_kernel().decode() == "slots"
    assert bool((np.abs(out.cpu().numpy() - oracle.spmm(ip, ix, ei, "copy_lhs", "sum", X, None)) <= 1e-4 * absum + 1e-30).all())
    # and without any plan: natural row order straight from indptr
    nat = sparse.CsrView(csc.num_rows, csc.num_cols, csc.indptr, csc.indices, csc.eids)
    nat._plan = None
    nat._short = {nb: None for nb in (2, 4, 8, 16, 32, 64)}
    status = _lib.lib().mgx_spmm_copy_u_slots(ctypes.byref(nat.c_struct()), None, sparse.REDUCE["mean"], sparse._ptr(x), D, int(x.stride(0)),
                                              sparse._ptr(slots), None, None, sparse._ptr(out), D, None, 0, None)
    torch.cuda.synchronize()
    if max(deg) <= 100000:
        assert status == 0, _lib.lib().mgx_last_error()
        ref = oracle.spmm(ip, ix, ei, "copy_lhs", "mean", X, None)
        assert bool((np.abs(out.cpu().numpy() - ref) <= 1e-4 * absum / np.maximum(deg, 1)[:, None] + 1e-30).all())


@pytest.mark.gpu
def test_slots_refuse_what_they_cannot_do():
    g = mg.graph((torch.tensor([0, 1]), torch.tensor([1, 2])), num_nodes=3).int().to(DEV)
    csc = g._index.csc()
    x = torch.rand(3, 32, device=DEV)
    out = torch.empty(3, 32, device=DEV)
    slots = torch.zeros(3, 32, dtype=torch.int32, device=DEV)
    status = _lib.lib().mgx_spmm_copy_u_slots(ctypes.byref(csc.c_struct()), None, sparse.REDUCE["sum"], sparse._ptr(x), 32, 32, sparse._ptr(slots),
                                              None, None, sparse._ptr(out), 32, None, 0, None)
    assert status == 2                                                     # MGX_ERR_UNSUPPORTED: 64 columns only
    status = _lib.lib().mgx_spmm_copy_u_slots(ctypes.byref(csc.c_struct()), None, sparse.REDUCE["max"], sparse._ptr(x), 64, 64, sparse._ptr(slots),
                                              None, None, sparse._ptr(out), 64, None, 0, None)
    assert status == 1                                                     # MGX_ERR_INVALID_ARGUMENT: SUM or MEAN only


@pytest.mark.gpu
def test_the_sage_layer_with_and_without_slots(monkeypatch):
    """ops.SageMeanCatFn over a tagged relu + dropout output: the slot form (forced on this small graph) against the dense kernels."""
    import full_graph
    from mi355x_graph.datasets import synthetic_edges
    n = 30000
    src, dst = synthetic_edges(n, 500000, 2000, seed=3, device=torch.device(DEV), symmetric=True)
    g = mg.graph((src, dst), num_nodes=n).int().formats(["csr", "csc"]).to(DEV)
    gen = torch.Generator().manual_seed(2)
    feats = torch.rand(n, 100, generator=gen).to(DEV)
    labels = torch.randint(0, 47, (n,), generator=gen).to(DEV)
    monkeypatch.setattr(mgx_config, "PACKED_GATHER_MIN_NNZ", 0)
    monkeypatch.setattr(ops, "_ROWS_GEMM_MIN", 0)
    results = {}
    seen = {}
    orig = sparse.HipBackend.spmm_copy_u_strided

    def spy(self, csr, reduce, U2d, out2d, accumulate=False, dst_scale=None, slots=None, src_scale=None):
        seen["slots"] = seen.get("slots", 0) + (slots is not None)
        seen["scaled"] = seen.get("scaled", 0) + (src_scale is not None)
        return orig(self, csr, reduce, U2d, out2d, accumulate=accumulate, dst_scale=dst_scale, slots=slots, src_scale=src_scale)

    monkeypatch.setattr(sparse.HipBackend, "spmm_copy_u_strided", spy)
    orig_pack = sparse.HipBackend.rows_slots_pack

    def pack_spy(self, x2d, overflow=None, row_scale=None):
        seen["packs"] = seen.get("packs", 0) + 1
        return orig_pack(self, x2d, overflow, row_scale)

    monkeypatch.setattr(sparse.HipBackend, "rows_slots_pack", pack_spy)
    for packed in (True, False):
        monkeypatch.setattr(mgx_config, "PACKED_GATHER", packed)
        seen["slots"] = seen["packs"] = seen["scaled"] = 0
        g._index.__dict__.pop("_slot_gate", None)
        torch.manual_seed(7)
        ops.ReluDropout._calls = 0
        model = full_graph.GraphSAGE(100, 64, 47, 3, 0.5, False, True).to(DEV)
        model.train()
        losses = []
        for _ in range(2):
            model.zero_grad()
            loss = ops.nll_sum(model(g, feats), labels) / n
            loss.backward()
            losses.append(float(loss))
        results[packed] = (losses, [p.grad.clone() for p in model.parameters()])
        # the two hidden layers' forward aggregations (slots written by the producing layer's GEMM epilogue) and the reversed
        # aggregation of layer 2's backward -- on the relu + dropout gradient itself, 1 / deg folded into the slots its own kernel
        # writes (d h = d y W_self + A^T (D^-1 d y) W_neigh) --, in both passes; never a separate pack pass
        assert seen["slots"] == (6 if packed else 0) and seen["scaled"] == (2 if packed else 0) and seen["packs"] == 0
    (la, ga), (lb, gb) = results[True], results[False]
    for a, b in zip(la, lb):
        assert abs(a - b) <= 1e-5 * abs(b)
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-6)


@pytest.mark.gpu
def test_the_gate_turns_the_slot_form_off_for_dense_rows(monkeypatch):
    """relu alone leaves half of the entries: nearly every row has more than 24 non-zeros and would be read from the dense matrix anyway --
    the second call reads the first one's overflow count (waiting for it: the decision does not depend on timing) and takes the dense
    kernels for the next 64 calls."""
    n = 20000
    src = torch.randint(0, n, (400000,))
    dst = torch.randint(0, n, (400000,))
    g = mg.graph((src, dst), num_nodes=n).int().to(DEV)
    csc = g._index.csc()
    be = sparse.backend_for(csc.indptr)
    monkeypatch.setattr(mgx_config, "PACKED_GATHER_MIN_NNZ", 0)
    csc._short = {nb: None for nb in (2, 4, 8, 16, 32, 64)}
    h = ops._structural_zeros(torch.relu(torch.randn(n, 64, device=DEV)))
    assert ops.has_structural_zeros(h)
    assert ops._packed_rows(be, g._index, csc, h, h) is not None          # nothing known yet: packed, and watched
    assert ops._packed_rows(be, g._index, csc, h, h) is None              # the count is waited for: ~100 % overflow rows
    gate = g._index._slot_gate
    assert gate.last_fraction > 0.9 and gate.dense_until >= 64
    sparse_h = ops._structural_zeros(h * (torch.rand(n, 64, device=DEV) < 0.4))
    gate.dense_until = 0                                                   # (64 calls later)
    assert ops._packed_rows(be, g._index, csc, sparse_h, sparse_h) is not None
    assert ops._packed_rows(be, g._index, csc, sparse_h, sparse_h) is not None and gate.last_fraction < 0.1
    h.add_(1.0)                                                            # an in-place write voids the tag: dense kernels
    assert ops._packed_rows(be, g._index, csc, h, h) is None


@pytest.mark.gpu
def test_gspmm_probes_an_untagged_operand(monkeypatch, oracle):
    """dgl.ops.gspmm / update_all(copy_u, mean) of a [N, 64] operand nobody tagged (the reference's own modules: F.relu + nn.Dropout): it is
    packed, aggregated over slots, and the CSR's gate decides from the overflow count whether the next calls do the same."""
    import dgl.function as fn
    n = 30000
    rng = np.random.default_rng(8)
    src, dst = rng.integers(0, n, 600000), rng.integers(0, n, 600000)
    g = mg.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=n).int().to(DEV)
    csc = g._index.csc()
    csc._short = {nb: None for nb in (2, 4, 8, 16, 32, 64)}
    monkeypatch.setattr(mgx_config, "PACKED_GATHER_MIN_NNZ", 0)
    X = (rng.standard_normal((n, 64)) * (rng.random((n, 64)) < 0.2)).astype(np.float32)
    h = torch.from_numpy(X).to(DEV).requires_grad_(True)
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    ref = oracle.spmm(ip, ix, ei, "copy_lhs", "mean", X, None)
    scale = oracle.spmm(ip, ix, ei, "copy_lhs", "mean", np.abs(X), None)
    for _ in range(2):
        g.ndata["h"] = h
        g.update_all(fn.copy_u("h", "m"), fn.mean("m", "neigh"))
        assert _lib.lib().mgx_last_spmm_kernel().decode() == "slots"
        out = g.ndata["neigh"]
        assert bool((np.abs(out.detach().cpu().numpy() - ref) <= 1e-4 * scale + 1e-30).all())
        torch.cuda.synchronize()
    out.sum().backward()                                              # the reversed aggregation of the (dense) gradient is not probed
    assert h.grad is not None                                         # (it runs on autograd's thread: the kernel name here is still the forward's)
    assert csc._gate is not None and csc._gate.last_fraction < 0.05 and g._index.csr()._gate is None
    dense = torch.randn(n, 64, device=DEV)
    first = ops.gspmm(g, "copy_lhs", "sum", dense, None)              # probed: ~100 % of the rows above 24 non-zeros -> the dense kernels,
    assert _lib.lib().mgx_last_spmm_kernel().decode() != "slots" and csc._gate.last_fraction > 0.9
    calls = csc._gate.calls
    assert csc._gate.dense_until == calls + 64                        # ... for this call and the next 64 on this CSR
    assert torch.equal(first, ops.gspmm(g, "copy_lhs", "sum", dense, None))   # same operand, same bits
    monkeypatch.setattr(mgx_config, "PACKED_GATHER_PROBE", False)
    csc._gate.dense_until = 0
    ops.gspmm(g, "copy_lhs", "sum", h.detach(), None)
    assert _lib.lib().mgx_last_spmm_kernel().decode() != "slots"
