"""GPU suite at BASELINE.json's FULL sizes, where the CPU oracle would take minutes: size-independent properties
of the domain instead of element-wise comparison.

  products shape  N = 2,449,029  E = 123,718,280   (config 3, the headline workload)      copy_u sum/mean, D = 64 / 100
  arxiv shape     N =   169,343  E ~ 2.3 M bidirected (config 1)                          D = 256
  reddit shape    N =   232,965  E = 11.6 M + self loops, 8 heads (config 2)              u_add_v, edge_softmax, u_mul_e

Properties: column checksums in fp64 (sum_v out[v] = sum_u outdeg(u) X[u]), linearity, adjointness
<A x, y> = <x, A^T y> (forward kernel vs the reversed-graph kernel used by backward), mean*deg = sum, softmax rows sum
to 1 and are shift invariant, CSR structure (monotone indptr, stable edge-id order inside rows, bincount = degrees),
COO -> CSR -> COO round trip, idempotence of to_bidirected, determinism (bitwise identical reruns)."""
import numpy as np
import pytest
import torch

import mi355x_graph as mg
from mi355x_graph import ops, sparse, transform
from mi355x_graph.datasets import SHAPES, synthetic_edges

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


@pytest.fixture(scope="module")
def products():
    spec = SHAPES["products"]
    src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], DEV, symmetric=True)
    g = mg.graph((src, dst), num_nodes=spec["n"]).int()
    return g, src, dst, spec["n"]


def test_products_csr_structure_bit_exact(products):
    g, src, dst, n = products
    E = src.shape[0]
    assert E == 123718280 and n == 2449029
    for view, rows, cols in ((g._index.csc(), dst, src), (g._index.csr(), src, dst)):
        ip, ix, ei = view.indptr.long(), view.indices.long(), view.eids.long()
        assert int(ip[0]) == 0 and int(ip[-1]) == E and bool((ip[1:] >= ip[:-1]).all())
        assert torch.equal(ip[1:] - ip[:-1], torch.bincount(rows, minlength=n))          # degrees, bit-exact
        assert torch.equal(cols[ei], ix) and torch.equal(torch.sort(ei)[0], torch.arange(E, device=DEV))  # a permutation
        starts = torch.zeros(E, dtype=torch.bool, device=DEV)
        starts[ip[:-1][ip[:-1] < E]] = True
        assert bool(((ei[1:] > ei[:-1]) | starts[1:]).all())                              # stable: edge ids ascend inside a row
        row_of = torch.repeat_interleave(torch.arange(n, device=DEV), ip[1:] - ip[:-1])
        assert torch.equal(rows[ei], row_of)                                              # CSR -> COO round trip
    assert torch.equal(g.in_degrees().long(), torch.bincount(dst, minlength=n))
    assert g.in_degrees().dtype == torch.int32


@pytest.mark.parametrize("D", [64, 100])
def test_products_spmm_properties(products, D):
    g, src, dst, n = products
    gen = torch.Generator(device=DEV).manual_seed(D)
    x = torch.rand(n, D, device=DEV, generator=gen)
    y = torch.rand(n, D, device=DEV, generator=gen)
    ax = ops.gspmm(g, "copy_lhs", "sum", x, None)
    outdeg = torch.bincount(src, minlength=n).double()
    indeg = torch.bincount(dst, minlength=n)
    # checksum: every source row is counted once per out-edge
    assert rel(ax.double().sum(0), (outdeg[:, None] * x.double()).sum(0)) < 1e-6
    # linearity
    axy = ops.gspmm(g, "copy_lhs", "sum", x + y, None)
    ay = ops.gspmm(g, "copy_lhs", "sum", y, None)
    assert rel(axy, ax + ay) < 1e-5
    # mean * max(deg, 1) = sum ; isolated rows are exactly 0
    am = ops.gspmm(g, "copy_lhs", "mean", x, None)
    assert rel(am * indeg.clamp(min=1)[:, None], ax) < 1e-6
    assert float(am[indeg == 0].abs().sum()) == 0.0
    # adjointness with the reversed-graph kernel (what backward runs)
    aty, _, _ = sparse.gspmm_raw(g._index.csr(), "copy_lhs", "sum", y, None)
    lhs = (ax.double() * y.double()).sum()
    rhs = (x.double() * aty.double()).sum()
    assert abs(float(lhs - rhs)) / abs(float(lhs)) < 1e-6
    # determinism
    assert torch.equal(ax, ops.gspmm(g, "copy_lhs", "sum", x, None))
    # autograd of mean at full size: grad of sum(out) w.r.t. x[u] = sum over out-edges of 1/deg(v)
    xr = x.clone().requires_grad_(True)
    ops.gspmm(g, "copy_lhs", "mean", xr, None).sum().backward()
    inv = 1.0 / indeg.clamp(min=1).double()
    want = torch.zeros(n, dtype=torch.float64, device=DEV).index_add_(0, src, inv[dst])
    assert rel(xr.grad[:, 0].double(), want) < 1e-5


def test_arxiv_shape_bidirected_and_sage_width():
    spec = SHAPES["arxiv"]
    src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], DEV, symmetric=False)
    g0 = mg.graph((src, dst), num_nodes=spec["n"])
    g = transform.to_bidirected(g0)
    s, d = g.edges()
    key = s * spec["n"] + d
    assert bool((key[1:] > key[:-1]).all())                                    # sorted by (src, dst), no duplicates
    rk = torch.unique(torch.cat([src * spec["n"] + dst, dst * spec["n"] + src]))
    assert torch.equal(key, rk)                                                # exactly the symmetrised edge set
    g2 = transform.to_bidirected(g)
    assert torch.equal(g2.edges()[0], s) and torch.equal(g2.edges()[1], d)     # idempotent
    gl = transform.add_self_loop(g)
    assert gl.number_of_edges() == g.number_of_edges() + spec["n"]
    assert torch.equal(gl.edges()[0][-spec["n"]:], torch.arange(spec["n"], device=DEV))
    g = g.int()
    x = torch.rand(spec["n"], 256, device=DEV)
    out = ops.gspmm(g, "copy_lhs", "sum", x, None)
    outdeg = torch.bincount(s, minlength=spec["n"]).double()
    assert rel(out.double().sum(0), (outdeg[:, None] * x.double()).sum(0)) < 1e-6


def test_reddit_shape_gat_pipeline_properties():
    spec = SHAPES["reddit-small"]
    H, F = 8, 16
    src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], DEV, symmetric=True)
    g = transform.add_self_loop(mg.graph((src, dst), num_nodes=spec["n"])).int()
    s, d = [t.long() for t in g.edges()]
    n, E = spec["n"], s.shape[0]
    el, er = torch.randn(n, H, 1, device=DEV), torch.randn(n, H, 1, device=DEV)
    e = ops.gsddmm(g, "add", el, er)
    outdeg, indeg = torch.bincount(s, minlength=n).double(), torch.bincount(d, minlength=n).double()
    want = (outdeg[:, None] * el[:, :, 0].double()).sum(0) + (indeg[:, None] * er[:, :, 0].double()).sum(0)
    assert rel(e[:, :, 0].double().sum(0), want) < 1e-6                        # SDDMM checksum
    a = ops.edge_softmax(g, e)
    rows = ops.gspmm(g, "copy_rhs", "sum", None, a)                            # per-destination sums
    assert float((rows - 1).abs().max()) < 1e-5 and float(a.min()) >= 0.0
    a2 = ops.edge_softmax(g, e + 3.25 * er.new_ones(1)[0])                     # shift invariance (constant per edge)
    assert float((a - a2).abs().max()) < 1e-5
    ft = torch.rand(n, H, F, device=DEV)
    out = ops.gspmm(g, "mul", "sum", ft, a)
    # convex combination: every output lies inside the range of the features
    assert float(out.max()) <= float(ft.max()) + 1e-5 and float(out.min()) >= float(ft.min()) - 1e-5
    # with uniform attention 1/deg the weighted sum equals the mean aggregator
    uni = (1.0 / indeg.clamp(min=1)).float()[d][:, None, None].expand(E, H, 1).contiguous()
    assert rel(ops.gspmm(g, "mul", "sum", ft, uni), ops.gspmm(g, "copy_lhs", "mean", ft, None)) < 1e-5
    # canonical (in-CSR) edge order gives the same layer output
    cidx, perm = g._index.canonical()
    assert rel(ops.gspmm(cidx, "mul", "sum", ft, a[perm.long()]), out) < 1e-6


def test_reddit_shape_fused_attention_head_dot_and_transpose_properties():
    """Full-size properties of the later kernels: fused GAT attention == the composed chain; the head-wise dot g-SDDMM is
    the adjoint of u_mul_e/sum in its edge argument (<A(ft, a), dZ> = <a, dot(ft[u], dZ[v])>), also on a CSR-only graph;
    mgx_csr_transpose of the in-CSR is bit-identical to the COO-built out-CSR; column sums against fp64."""
    import torch.nn.functional as F_
    spec = SHAPES["reddit-small"]
    H, F = 8, 8
    src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], DEV, symmetric=True)
    g = transform.add_self_loop(mg.graph((src, dst), num_nodes=spec["n"])).int()
    n, E = spec["n"], g.number_of_edges()
    el, er = torch.randn(n, H, 1, device=DEV), torch.randn(n, H, 1, device=DEV)
    fused = ops.gat_attention(g, el, er, 0.2)
    chain = ops.edge_softmax(g, F_.leaky_relu(ops.gsddmm(g, "add", el, er), 0.2))
    assert float((fused - chain).abs().max()) < 1e-6
    ft, dz = torch.randn(n, H, F, device=DEV), torch.randn(n, H, F, device=DEV)
    a = torch.rand(E, H, 1, device=DEV)
    for graph in (g, g.formats(["csr", "csc"])):                      # COO walk, then the CSR-walk head-dot kernel
        lhs = (ops.gspmm(graph, "mul", "sum", ft, a).double() * dz.double()).sum()
        rhs = (a.double() * ops.gsddmm(graph, "dot", ft, dz).double()).sum()
        assert abs(float(lhs - rhs)) < 1e-6 * float(lhs.abs().clamp(min=1.0)) * 10
    csc, csr = g._index.csc(), g._index.csr()
    t = sparse.csr_transpose(csc)
    assert torch.equal(t.indptr, csr.indptr) and torch.equal(t.indices, csr.indices) and torch.equal(t.eids, csr.eids)
    x = torch.rand(spec["n"] * 10, 47, device=DEV)                    # products-sized column sum (2.3 M x 47)
    got = sparse.backend_for(x).column_sum(x)
    assert rel(got.double(), x.double().sum(0)) < 1e-6


@pytest.fixture(scope="module")
def reddit_full():
    spec = SHAPES["reddit"]
    src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], DEV, symmetric=True)
    g = mg.graph((src, dst), num_nodes=spec["n"]).int()
    return g, src, dst, spec["n"]


@pytest.mark.parametrize("D", [602, 256])
def test_reddit_full_size_wide_rows_properties(reddit_full, D):
    """reddit at the dataset's own size (E = 114.6 M, 492 in-edges per node), the input width of main_dgl_reddit_sage.py
    (602: line-padded copy + 128-column passes) and a hidden width that takes two column passes: fp64 column checksum,
    adjointness with the reversed-graph kernel, mean * deg = sum, determinism."""
    g, src, dst, n = reddit_full
    assert src.shape[0] == 114615892
    gen = torch.Generator(device=DEV).manual_seed(D)
    x = torch.rand(n, D, device=DEV, generator=gen)
    ax = ops.gspmm(g, "copy_lhs", "sum", x, None)
    outdeg = torch.bincount(src, minlength=n).double()
    ref_cols = (outdeg.view(-1, 1) * x.double()).sum(0)          # sum_v out[v] = sum_u outdeg(u) X[u]
    assert rel(ax.double().sum(0), ref_cols) < 1e-6
    y = torch.rand(n, 8, device=DEV, generator=gen)
    aty = ops.gspmm(g.reverse(), "copy_lhs", "sum", y, None)     # A^T y, D = 8
    lhs = (ax[:, :8].double() * y.double()).sum()
    rhs = (x[:, :8].double() * aty.double()).sum()
    assert abs(float(lhs - rhs)) < 1e-6 * abs(float(rhs))
    mean = ops.gspmm(g, "copy_lhs", "mean", x, None)
    indeg = g.in_degrees().clamp(min=1).float().view(-1, 1)
    assert rel(mean * indeg, ax) < 1e-5
    assert torch.equal(ax, ops.gspmm(g, "copy_lhs", "sum", x, None))


def test_reddit_full_size_u_add_v_checksum(reddit_full):
    """The lean COO g-SDDMM at the kernel sweep's own size: sum_e (U[src e] + V[dst e]) per column in fp64."""
    g, src, dst, n = reddit_full
    D = 32
    gen = torch.Generator(device=DEV).manual_seed(9)
    u, v = torch.rand(n, D, device=DEV, generator=gen), torch.rand(n, D, device=DEV, generator=gen)
    out = ops.gsddmm(g, "add", u, v)
    assert out.shape == (src.shape[0], D)
    outdeg, indeg = torch.bincount(src, minlength=n).double(), torch.bincount(dst, minlength=n).double()
    ref = (outdeg.view(-1, 1) * u.double()).sum(0) + (indeg.view(-1, 1) * v.double()).sum(0)
    assert rel(out.double().sum(0), ref) < 1e-6
    e = torch.randint(0, src.shape[0], (4096,), device=DEV, generator=gen)
    assert torch.equal(out[e], u[src[e]] + v[dst[e]])            # bit-exact on sampled edges
