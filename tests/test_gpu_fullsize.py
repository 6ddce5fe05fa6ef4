"""GPU suite at BASELINE.json's FULL sizes: size-independent properties of the domain AND (round 4) element-wise parity
against the CPU oracle on all host cores -- the OpenMP oracle aggregates the full products graph in 0.3-0.5 s (bench.py's
cpu_baseline), so nothing stands in the way of holding every output row to it (second half of this file).

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


# ----------------------------------------------------------------------------- element-wise against the oracle, full size
def _host_cores():
    import os
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def assert_rows_close(out, ref, scale, terms, what):
    """north_star's 1e-4 relative bound per output ELEMENT, against the sum of |terms| reduced into it (`scale`), plus what the
    oracle's own sequential fp32 sum may be off by on long rows (1e-8 per term, measured in tests/test_tile_spmm.py: hub rows
    of 17 k terms)."""
    out, ref, scale = (np.asarray(t, np.float64) for t in (out, ref, scale))
    assert out.shape == ref.shape == scale.shape, what
    bound = (1e-4 + 1e-8 * np.asarray(terms, np.float64).reshape((-1,) + (1,) * (out.ndim - 1))) * scale + 1e-30
    err = np.abs(out - ref)
    bad = err > bound
    assert not bad.any(), "%s: %d elements off, worst err/bound %.3g" % (what, int(bad.sum()), float((err / bound).max()))


@pytest.fixture(scope="module")
def products_host(products, oracle):
    """The products-shaped graph on the host: the ORACLE's own CSR builds (orc_coo_to_csr, stable counting sort) of both
    directions, checked bit for bit against the device-built formats the kernels walk."""
    g, src, dst, n = products
    oracle.set_num_threads(_host_cores())
    s, d = src.to(torch.int32).cpu().numpy(), dst.to(torch.int32).cpu().numpy()
    host = {}
    for name, view, rows, cols in (("csc", g._index.csc(), d, s), ("csr", g._index.csr(), s, d)):
        ip, ix, ei = oracle.coo_to_csr(n, rows, cols)
        assert np.array_equal(ip, view.indptr.cpu().numpy()), name          # indptr / degrees bit-exact at full size
        assert np.array_equal(ix, view.indices.cpu().numpy()), name
        assert np.array_equal(ei, view.eids.cpu().numpy()), name
        host[name] = (ip, ix, ei)
    assert np.array_equal(oracle.in_degrees(host["csc"][0]), g.in_degrees().cpu().numpy())
    return host


@pytest.mark.parametrize("D", [64, 100])
def test_products_every_output_row_against_the_oracle(products, products_host, oracle, D):
    """kernel/dgl-new.py:20 / main_dgl_product_sage.py:62 at N = 2,449,029, E = 123,718,280: copy_u/mean forward (what the layer
    runs) and the reversed-graph copy_u/sum (what its backward runs), every element of every row, hub rows of 17 k terms
    included; features U[0,1) as kernel/dgl-new.py:15 draws them."""
    g, src, dst, n = products
    gen = torch.Generator(device=DEV).manual_seed(100 + D)
    x = torch.rand(n, D, device=DEV, generator=gen)
    xh = x.cpu().numpy()
    ip, ix, ei = products_host["csc"]
    deg = np.diff(ip)
    ref = oracle.spmm(ip, ix, ei, "copy_lhs", "mean", xh, None)
    got = ops.gspmm(g, "copy_lhs", "mean", x, None).cpu().numpy()
    assert_rows_close(got, ref, np.abs(ref), deg, "products copy_u/mean D=%d" % D)   # all terms >= 0: the sum IS the scale
    assert float(np.abs(got[deg == 0]).sum()) == 0.0
    rp, rx, re_ = products_host["csr"]
    ref_t = oracle.spmm(rp, rx, re_, "copy_lhs", "sum", xh, None)
    got_t, _, _ = sparse.gspmm_raw(g._index.csr(), "copy_lhs", "sum", x, None)
    assert_rows_close(got_t.cpu().numpy(), ref_t, np.abs(ref_t), np.diff(rp), "products reversed copy_u/sum D=%d" % D)


def test_arxiv_shape_sage_width_against_the_oracle(oracle):
    """config 1 (main_dgl_arxiv_sage.py:162,144): bidirected arxiv shape, D = 256, sum and mean, signed features."""
    spec = SHAPES["arxiv"]
    oracle.set_num_threads(_host_cores())
    src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], DEV, symmetric=False)
    g = transform.to_bidirected(mg.graph((src, dst), num_nodes=spec["n"])).int()
    s, d = [t.cpu().numpy() for t in g.edges()]
    rs, rd = oracle.to_bidirected(src.cpu().numpy(), dst.cpu().numpy(), spec["n"])
    assert np.array_equal(rs, s) and np.array_equal(rd, d)                       # the transform itself, bit-exact
    ip, ix, ei = oracle.coo_to_csr(spec["n"], d, s)
    x = torch.randn(spec["n"], 256, device=DEV)
    xh = x.cpu().numpy()
    scale = oracle.spmm(ip, ix, ei, "copy_lhs", "sum", np.abs(xh), None)
    for red in ("sum", "mean"):
        ref = oracle.spmm(ip, ix, ei, "copy_lhs", red, xh, None)
        sc = scale if red == "sum" else scale / np.maximum(np.diff(ip), 1)[:, None]
        assert_rows_close(ops.gspmm(g, "copy_lhs", red, x, None).cpu().numpy(), ref, sc, np.diff(ip), "arxiv D=256 " + red)


def test_reddit_shape_eight_head_pipeline_against_the_oracle_composition(oracle):
    """config 2: u_add_v -> leaky_relu -> edge_softmax -> u_mul_e/sum on the reddit shape (11.6 M edges + self loops, 8 heads of
    16), unfused operator by operator AND as the fused block, against the oracle's composition."""
    spec = SHAPES["reddit-small"]
    H, F = 8, 16
    oracle.set_num_threads(_host_cores())
    src, dst = synthetic_edges(spec["n"], spec["m"], spec["max_deg"], spec["seed"], DEV, symmetric=True)
    g = transform.add_self_loop(mg.graph((src, dst), num_nodes=spec["n"])).int()
    s, d = [t.cpu().numpy() for t in g.edges()]
    n, E = spec["n"], s.shape[0]
    gen = torch.Generator(device=DEV).manual_seed(8)
    el = torch.randn(n, H, 1, device=DEV, generator=gen) * 2
    er = torch.randn(n, H, 1, device=DEV, generator=gen) * 2
    ft = torch.randn(n, H, F, device=DEV, generator=gen)
    ip, ix, ei = oracle.coo_to_csr(n, d, s)
    z = oracle.sddmm(s, d, "add", el.cpu().numpy(), er.cpu().numpy())
    e_dev = ops.gsddmm(g, "add", el, er)
    assert np.array_equal(e_dev.cpu().numpy(), z)                                # element-wise g-SDDMM: bit-exact
    z = np.where(z > 0, z, np.float32(0.2) * z).astype(np.float32)
    a_ref = oracle.edge_softmax_fwd(ip, ei, z.reshape(E, H))
    a_dev = ops.edge_softmax(g, torch.nn.functional.leaky_relu(e_dev, 0.2))
    a_got = a_dev.cpu().numpy().reshape(E, H).astype(np.float64)
    assert bool((np.abs(a_got - a_ref) <= 1e-4 * a_ref + 1e-12).all())          # every probability within 1e-4 relative (hub rows: 20 k terms)
    fth = ft.cpu().numpy()
    ref = oracle.spmm(ip, ix, ei, "mul", "sum", fth, a_ref.reshape(E, H, 1))
    scale = oracle.spmm(ip, ix, ei, "mul", "sum", np.abs(fth), a_ref.reshape(E, H, 1))
    deg = np.diff(ip)
    unfused = ops.gspmm(g, "mul", "sum", ft, a_dev)
    assert_rows_close(unfused.cpu().numpy(), ref, scale, deg, "reddit 8-head unfused")
    assert ops.gat_fused_supported(g, ft)
    fused = ops.gat_fused(g, ft, el, er, 0.2, 0.0, True)
    assert_rows_close(fused.cpu().numpy(), ref, scale, deg, "reddit 8-head fused")


def test_int32_ids_with_a_gathered_matrix_beyond_4_gib(oracle):
    """int32 node ids but N * D * 4 >= 2^32 bytes: the lean kernel's 32-bit byte offsets do not reach, the launcher must take
    the 64-bit path (csrc/spmm.hip) -- 2.2 M rows of 512 floats = 4.5 GB, a skewed graph whose edges touch the last rows."""
    oracle.set_num_threads(_host_cores())
    n, D, nnz = 2_200_000, 512, 6_000_000
    assert n * D * 4 >= 2 ** 32
    rng = np.random.default_rng(41)
    src = rng.integers(0, n, nnz)
    src[: nnz // 4] = rng.integers(n - 50_000, n, nnz // 4)        # a quarter of the gathers beyond the 4 GiB mark
    dst = (rng.pareto(1.2, nnz) * 2000).astype(np.int64) % n        # heavy head: hub rows
    g = mg.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=n).int().to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.rand(n, D, device=DEV, generator=gen)
    xh = x.cpu().numpy()
    ip, ix, ei = oracle.coo_to_csr(n, dst, src)
    for red in ("sum", "mean"):
        ref = oracle.spmm(ip, ix, ei, "copy_lhs", red, xh, None)
        got = ops.gspmm(g, "copy_lhs", red, x, None).cpu().numpy()
        assert_rows_close(got, ref, np.abs(ref), np.diff(ip), "int32 ids, 4.5 GB matrix, " + red)
    # the strided in-place form used by the one-GEMM SAGE layer must refuse or handle the same size, never wrap around
    left = torch.empty(n, 2 * D, device=DEV)[:, :D]
    left.copy_(x)
    out = torch.empty(n, D, device=DEV)
    try:
        sparse.backend_for(x).spmm_copy_u_strided(g._index.csc(), "sum", left, out)
    except mg.DGLError as err:  # a loud refusal is the contract (callers fall back to the dense form)
        assert "4 GiB" in str(err)
    else:
        ref = oracle.spmm(ip, ix, ei, "copy_lhs", "sum", xh, None)
        assert_rows_close(out.cpu().numpy(), ref, np.abs(ref), np.diff(ip), "strided, 9 GB operand")


def test_reddit_full_size_tile_kernel_every_row_against_the_oracle(reddit_full, oracle):
    """kernel/dgl-new.py:61 at the dataset's own size (N = 232,965, E = 114,615,892, 492 in-edges per node): what gspmm selects here is
    the LDS-staged TILE kernel (64- and 16-column geometries) -- every output element against the oracle, hub rows of 21 k terms included."""
    from mi355x_graph import _lib
    g, src, dst, n = reddit_full
    oracle.set_num_threads(_host_cores())
    ip, ix, ei = oracle.coo_to_csr(n, dst.to(torch.int32).cpu().numpy(), src.to(torch.int32).cpu().numpy())
    csc = g._index.csc()
    assert np.array_equal(ip, csc.indptr.cpu().numpy()) and np.array_equal(ix, csc.indices.cpu().numpy())
    deg = np.diff(ip)
    for D in (64, 16):
        gen = torch.Generator(device=DEV).manual_seed(200 + D)
        x = torch.rand(n, D, device=DEV, generator=gen)
        got = ops.gspmm(g, "copy_lhs", "mean", x, None).cpu().numpy()
        assert _lib.lib().mgx_last_spmm_kernel().decode() == "tile"
        ref = oracle.spmm(ip, ix, ei, "copy_lhs", "mean", x.cpu().numpy(), None)
        assert_rows_close(got, ref, np.abs(ref), deg, "reddit E = 114.6 M, tile kernel, D=%d" % D)
