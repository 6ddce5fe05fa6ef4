"""mgx_rows_gemm (csrc/rowsgemm.hip): C = A x B (+ bias) (x a row factor on a column range) for tall A and small K, M -- the dense
projections of a SAGE layer (main_dgl_product_sage.py:23-24,64) -- against fp64 torch."""
import os
import sys

import pytest
import torch

from mi355x_graph import config as mgx_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import mi355x_graph as mg  # noqa: E402,F401
from mi355x_graph import sparse  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def reference(a, b, bt, bias, rs, scale_from):
    c = a.double() @ (b.double().t() if bt else b.double())
    if bias is not None:
        c = c + bias.double()
    if rs is not None:
        c[:, scale_from:] *= rs.double().view(-1, 1)
    bound = a.double().abs() @ (b.double().abs().t() if bt else b.double().abs())
    if bias is not None:
        bound = bound + bias.double().abs()
    if rs is not None:
        bound[:, scale_from:] *= rs.double().view(-1, 1)
    return c, bound


@pytest.mark.parametrize("K,M,bt,lda_pad,n", [
    (200, 64, True, 0, 70001),     # forward layer 1 of the products model: [x | agg] (200) -> 64
    (128, 64, True, 0, 65600),     # forward layer 2
    (128, 47, True, 0, 66000),     # forward layer 3: 47 classes (column guard)
    (64, 128, False, 0, 70000),    # backward layer 2: dY (64) -> d[h | neigh] (128)
    (47, 128, False, 0, 65599),    # backward layer 3: dword loads (K = 47)
    (64, 128, False, 8, 5000),     # a row-strided A (a column block of a wider matrix), short input
    (40, 64, False, 0, 1000), (41, 128, False, 0, 333), (32, 64, True, 0, 17), (48, 128, True, 0, 64),
])
def test_rows_gemm_matches_fp64(K, M, bt, lda_pad, n):
    gen = torch.Generator(device=DEV).manual_seed(K * 1000 + M)
    be = sparse.backend_for(torch.zeros(1, device=DEV))
    wide = torch.randn(n, K + lda_pad, device=DEV, generator=gen)
    a = wide[:, :K]
    b = torch.randn((M, K) if bt else (K, M), device=DEV, generator=gen)
    bias = torch.randn(M, device=DEV, generator=gen)
    rs = torch.rand(n, device=DEV, generator=gen) + 0.25
    for use_bias, use_rs, scale_from in ((False, False, 0), (True, False, 0), (False, True, M // 2), (True, True, 0), (False, True, M)):
        out = be.rows_gemm(a, b, b_transposed=bt, bias=bias if use_bias else None, row_scale=rs if use_rs else None, scale_from=scale_from)
        assert out is not None, (K, M, bt)
        ref, bound = reference(a, b, bt, bias if use_bias else None, rs if use_rs else None, scale_from)
        err = (out.double() - ref).abs()
        assert bool((err <= 1e-5 * bound + 1e-30).all()), (K, M, use_bias, use_rs, float((err / (bound + 1e-30)).max()))
    if M % 8 == 0:   # two compact outputs: the columns before / from M / 2 on (the halves of d[h | neigh])
        whole = be.rows_gemm(a, b, b_transposed=bt, bias=bias, row_scale=rs, scale_from=M // 2)
        left, right = be.rows_gemm(a, b, b_transposed=bt, bias=bias, row_scale=rs, scale_from=M // 2, split_col=M // 2)
        assert left.is_contiguous() and right.is_contiguous() and torch.equal(left, whole[:, :M // 2]) and torch.equal(right, whole[:, M // 2:])
    again = be.rows_gemm(a, b, b_transposed=bt, bias=bias, row_scale=rs, scale_from=0)
    assert torch.equal(again, be.rows_gemm(a, b, b_transposed=bt, bias=bias, row_scale=rs, scale_from=0))   # fixed summation order


def test_unsupported_shapes_return_none_and_bad_arguments_raise():
    be = sparse.backend_for(torch.zeros(1, device=DEV))
    a = torch.rand(100, 512, device=DEV)
    assert be.rows_gemm(a, torch.rand(512, 256, device=DEV)) is None            # B beyond the 64 KB stage: the caller uses a GEMM
    assert be.rows_gemm(torch.rand(100, 100, device=DEV), torch.rand(100, 64, device=DEV)) is None   # no kernel built for K = 100
    with pytest.raises(mg.DGLError):
        be.rows_gemm(torch.rand(100, 64, device=DEV), torch.rand(64, 128, device=DEV), row_scale=torch.rand(100, device=DEV), scale_from=129)


def test_sage_layer_takes_it_and_keeps_its_gradients():
    """The products layer shapes through full_graph.GraphSAGE: loss and every gradient with MGX_ROWS_GEMM's kernel against the library
    GEMM form of the same autograd node (toggled in-process)."""
    import full_graph
    from mi355x_graph import ops
    n = 70000
    gen = torch.Generator().manual_seed(3)
    src, dst = torch.randint(0, n, (8 * n,), generator=gen), torch.randint(0, n, (8 * n,), generator=gen)
    g = mg.graph((src, dst), num_nodes=n).int().to(DEV)
    x = torch.rand(n, 100, generator=gen).to(DEV)
    y = torch.randint(0, 47, (n,), generator=gen).to(DEV)
    rows = torch.arange(0, n, 9, device=DEV)
    results = []
    for on in (True, False):
        mgx_config.ROWS_GEMM = on
        try:
            torch.manual_seed(5)
            model = full_graph.GraphSAGE(100, 64, 47, 3, 0.0, False, True).to(DEV)
            loss = ops.nll_sum(model(g, x, rows=rows), y[rows]) / rows.shape[0]
            loss.backward()
            results.append((float(loss), [p.grad.clone() for p in model.parameters()]))
        finally:
            mgx_config.ROWS_GEMM = True
    (l1, g1), (l0, g0) = results
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    for a, b in zip(g1, g0):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-7


def test_layer_with_the_activation_in_the_epilogue_is_bitwise_the_composition():
    """ops.sage_mean_layer_act == ops.relu_dropout(ops.sage_mean_layer(...), out=...): same random stream position, same arithmetic ->
    the same bits in the activation, the loss and every gradient of the products-shaped model (dropout 0.5, three layers)."""
    import full_graph
    from mi355x_graph import ops
    n = 70000
    gen = torch.Generator().manual_seed(7)
    src, dst = torch.randint(0, n, (6 * n,), generator=gen), torch.randint(0, n, (6 * n,), generator=gen)
    g = mg.graph((src, dst), num_nodes=n).int().to(DEV)
    x = torch.rand(n, 100, generator=gen).to(DEV)
    y = torch.randint(0, 47, (n,), generator=gen).to(DEV)
    rows = torch.arange(0, n, 7, device=DEV)
    results = []
    for fused in ("1", "0"):
        mgx_config.SAGE_FUSED_ACT = fused == "1"
        try:
            torch.manual_seed(11)
            ops.ReluDropout._calls = 0
            model = full_graph.GraphSAGE(100, 64, 47, 3, 0.5, False, True).to(DEV)
            model.train()
            out = model(g, x, rows=rows)
            loss = ops.nll_sum(out, y[rows]) / rows.shape[0]
            loss.backward()
            results.append((out.detach().clone(), float(loss), [p.grad.clone() for p in model.parameters()], ops.ReluDropout._calls))
        finally:
            mgx_config.SAGE_FUSED_ACT = True
    (o1, l1, g1, c1), (o0, l0, g0, c0) = results
    assert c1 == c0 == 2                                   # two activations, two positions in the random stream either way
    assert torch.equal(o1, o0) and l1 == l0
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)


def test_rows_gemm_relu_dropout_equals_gemm_then_relu_dropout():
    be = sparse.backend_for(torch.zeros(1, device=DEV))
    gen = torch.Generator(device=DEV).manual_seed(5)
    n, K, M = 66001, 128, 64
    a = torch.randn(n, K, device=DEV, generator=gen)
    w = torch.randn(M, K, device=DEV, generator=gen)
    b = torch.randn(M, device=DEV, generator=gen)
    wide = torch.zeros(n, 2 * M, device=DEV)
    y, mask = be.rows_gemm_relu_dropout(a, w, True, b, 0.3, 12345, 777, out=wide[:, :M])
    z = be.rows_gemm(a, w, b_transposed=True, bias=b)
    y2, mask2 = be.relu_dropout_fwd(z, 0.3, 12345, 777)
    assert torch.equal(y, y2) and torch.equal(mask, mask2) and float(wide[:, M:].abs().sum()) == 0.0
    kept = float((mask2 > 0).float().mean())
    assert 0.2 < kept < 0.9
