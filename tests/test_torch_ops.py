"""torch.library registrations (mi355x_graph/torch_ops.py): schema + fake (meta) implementations on CPU --
shape inference needs no GPU --, and on the GPU parity with the direct ctypes path, torch.library.opcheck, tracing through
torch.compile (aot_eager: no code generation), and the MGX_TORCH_OPS=1 switch of the autograd layer."""
import numpy as np
import pytest
import torch

import mi355x_graph as mg
from mi355x_graph import ops, torch_ops  # noqa: F401  (registers the ops)
from conftest import random_graph

OPS = ["gspmm", "gsddmm", "edge_softmax_fwd", "edge_softmax_bwd", "segment_reduce", "coo_to_csr", "csr_transpose", "in_degrees"]


def test_ops_are_registered_with_schemas():
    for name in OPS:
        op = getattr(torch.ops.mi355x_graph, name)
        assert "mi355x_graph::" + name in str(op.default._schema)


def test_fake_implementations_infer_shapes_without_a_gpu():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        n_dst, n_src, nnz = 7, 9, 30
        indptr = torch.empty(n_dst + 1, dtype=torch.int32)
        indices = torch.empty(nnz, dtype=torch.int32)
        eids = torch.empty(nnz, dtype=torch.int32)
        out, au, ae = torch.ops.mi355x_graph.gspmm(indptr, indices, eids, n_src, "copy_lhs", "sum", torch.empty(n_src, 4, 8), None)
        assert out.shape == (n_dst, 4, 8) and au.numel() == 0 and ae.numel() == 0
        out, au, ae = torch.ops.mi355x_graph.gspmm(indptr, indices, eids, n_src, "mul", "max", torch.empty(n_src, 4, 8), torch.empty(nnz, 4, 1))
        assert out.shape == au.shape == ae.shape == (n_dst, 4, 8) and au.dtype == torch.int32
        e = torch.ops.mi355x_graph.gsddmm(indptr, indices, eids, n_src, "dot", torch.empty(n_src, 2, 5), torch.empty(n_dst, 2, 5), "u", "v")
        assert e.shape == (nnz, 2, 1)
        assert torch.ops.mi355x_graph.gsddmm(indptr, indices, None, n_src, "add", torch.empty(n_src, 3), torch.empty(n_dst, 3), "u", "v").shape == (nnz, 3)
        assert torch.ops.mi355x_graph.edge_softmax_fwd(indptr, indices, eids, n_src, torch.empty(nnz, 4, 1)).shape == (nnz, 4, 1)
        assert torch.ops.mi355x_graph.segment_reduce(torch.empty(5, dtype=torch.int64), torch.empty(20, 6), "mean").shape == (4, 6)
        ip, ix, ei = torch.ops.mi355x_graph.coo_to_csr(indices, indices, n_dst, n_src)
        assert ip.shape == (n_dst + 1,) and ix.shape == ei.shape == (nnz,)
        tp, tx, te = torch.ops.mi355x_graph.csr_transpose(indptr, indices, eids, n_src)
        assert tp.shape == (n_src + 1,) and tx.shape == (nnz,)
        assert torch.ops.mi355x_graph.in_degrees(indptr).shape == (n_dst,)


@pytest.mark.gpu
def test_registered_ops_match_the_direct_path(oracle):
    dev = "cuda:0"
    n_src, n_dst, nnz = 500, 400, 9000
    src, dst = random_graph(n_src, n_dst, nnz, seed=1)
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), n_src, n_dst, idtype=torch.int32, device=dev)
    csc = g._index.csc()
    args = torch_ops.csr_args(csc)
    ph = torch_ops.plan_handle(csc)   # the CSR's execution schedule (hub rows split): same summation order as the direct path
    assert (ph != 0) == (torch_ops.NATIVE and csc.plan() is not None)
    x = torch.randn(n_src, 4, 8, device=dev)
    w = torch.rand(nnz, 4, 1, device=dev)
    out, au, ae = torch.ops.mi355x_graph.gspmm(*args, "mul", "sum", x, w, ph)
    assert torch.equal(out, ops.gspmm(g, "mul", "sum", x, w)) and au.numel() == 0
    out0 = torch.ops.mi355x_graph.gspmm(*args, "mul", "sum", x, w)[0]            # plan = 0: natural row order, same sums to rounding
    assert float((out0 - out).abs().max()) <= 1e-4 * float(out.abs().max())
    out, au, ae = torch.ops.mi355x_graph.gspmm(*args, "copy_lhs", "max", x, None)
    ip, ix, ei = oracle.coo_to_csr(n_dst, dst, src)
    ref, ru, _ = oracle.spmm(ip, ix, ei, "copy_lhs", "max", x.cpu().numpy(), None, want_arg=True)
    assert np.array_equal(out.cpu().numpy(), ref) and np.array_equal(au.cpu().numpy(), ru)
    y = torch.randn(n_dst, 4, 8, device=dev)
    assert torch.equal(torch.ops.mi355x_graph.gsddmm(*args, "dot", x, y, "u", "v", ph), ops.gsddmm(g.formats(["csr", "csc"]), "dot", x, y))
    z = torch.randn(nnz, 4, 1, device=dev, requires_grad=True)
    a = torch.ops.mi355x_graph.edge_softmax_fwd(*args, z, ph)
    z2 = z.detach().clone().requires_grad_(True)
    a2 = ops.edge_softmax(g, z2)
    assert torch.equal(a, a2)
    (a * w).sum().backward()          # autograd registered on the op itself
    (a2 * w).sum().backward()
    assert torch.equal(z.grad, z2.grad)
    r, c = torch.from_numpy(dst).to(dev).int(), torch.from_numpy(src).to(dev).int()
    p, i, e = torch.ops.mi355x_graph.coo_to_csr(r, c, n_dst, n_src)
    assert np.array_equal(p.cpu().numpy(), ip) and np.array_equal(i.cpu().numpy(), ix) and np.array_equal(e.cpu().numpy(), ei)
    tp, tx, te = torch.ops.mi355x_graph.csr_transpose(p, i, e, n_src)
    rp, rx, re = oracle.coo_to_csr(n_src, src, dst)
    assert np.array_equal(tp.cpu().numpy(), rp) and np.array_equal(tx.cpu().numpy(), rx) and np.array_equal(te.cpu().numpy(), re)
    assert np.array_equal(torch.ops.mi355x_graph.in_degrees(p).cpu().numpy(), np.diff(ip))
    off = torch.tensor([0, 100, 100, 350, n_src], device=dev)
    seg = torch.ops.mi355x_graph.segment_reduce(off, x, "mean")
    assert float((seg[0] - x[:100].mean(0)).abs().max()) < 1e-5 and float(seg[1].abs().max()) == 0.0


@pytest.mark.gpu
def test_opcheck_and_trace():
    dev = "cuda:0"
    src, dst = random_graph(60, 50, 700, seed=2)
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), 60, 50, idtype=torch.int32, device=dev)
    args = torch_ops.csr_args(g._index.csc())
    x = torch.randn(60, 8, device=dev)
    torch.library.opcheck(torch.ops.mi355x_graph.gspmm.default, (*args, "copy_lhs", "sum", x, None),
                          test_utils=("test_schema", "test_faketensor"))
    z = torch.randn(700, 2, device=dev, requires_grad=True)
    torch.library.opcheck(torch.ops.mi355x_graph.edge_softmax_fwd.default, (*args, z),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))

    def f(x, z):
        h = torch.ops.mi355x_graph.gspmm(*args, "copy_lhs", "mean", x, None)[0]
        return torch.relu(h).sum() + torch.ops.mi355x_graph.edge_softmax_fwd(*args, z).square().sum()

    compiled = torch.compile(f, backend="aot_eager", fullgraph=True)   # traced through the fake implementations; no codegen
    assert abs(float(compiled(x, z)) - float(f(x, z))) < 1e-4 * abs(float(f(x, z)))


@pytest.mark.gpu
def test_autograd_layer_through_registered_ops(monkeypatch):
    """MGX_TORCH_OPS=1: SAGEConv-style update_all forward/backward is bit-identical to the ctypes path."""
    import mi355x_graph.function as fn
    dev = "cuda:0"
    src, dst = random_graph(800, 800, 20000, seed=3)
    g = mg.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=800).int().to(dev)
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("MGX_TORCH_OPS", flag)
        x = torch.arange(800 * 16, device=dev, dtype=torch.float32).view(800, 16).sin().requires_grad_(True)
        gg = g.local_var()
        gg.srcdata["h"] = x
        gg.update_all(fn.copy_src("h", "m"), fn.mean("m", "neigh"))
        out = gg.dstdata["neigh"]
        out.square().sum().backward()
        res.append((out.detach().clone(), x.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.gpu
def test_plan_handle_keeps_its_owner_and_goes_stale_with_it():
    """a plan travels through the dispatcher as an address; the autograd node of edge_softmax keeps the owner alive,
    and a handle whose owner is gone is refused on the Python side instead of being dereferenced in C++."""
    import gc
    dev = torch.device("cuda:0")
    src, dst = random_graph(300, 300, 4000, seed=3)
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), 300, 300, idtype=torch.int32, device=dev)
    csc = g._index.csc()
    h = torch_ops.softmax_plan_handle(csc)
    if h:
        assert torch_ops.plan_owner(h) is csc.softmax_plan()
    z = torch.randn(4000, 2, 1, device=dev, requires_grad=True)
    a = torch.ops.mi355x_graph.edge_softmax_fwd(csc.indptr, csc.indices, csc.eids, csc.num_cols, z, h)
    if h:
        assert a.grad_fn is not None
    del g, csc
    gc.collect()
    a.sum().backward()  # the node holds the plan: nothing was freed under it
    assert z.grad is not None and bool(torch.isfinite(z.grad).all())
    del a
    gc.collect()
    if h:
        with pytest.raises(mg.DGLError):
            torch_ops.plan_owner(h)


@pytest.mark.gpu
def test_ops_run_on_the_device_of_their_tensors_not_the_current_one():
    """every op in csrc/torch_bind.cpp makes its tensors' device current for the call (several GPUs in one process)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs in one process")
    dev = torch.device("cuda:1")
    src, dst = random_graph(200, 200, 3000, seed=5)
    g = mg.create_block((torch.from_numpy(src), torch.from_numpy(dst)), 200, 200, idtype=torch.int32, device=dev)
    csc = g._index.csc()
    x = torch.rand(200, 16, device=dev)
    with torch.cuda.device(0):
        out, _, _ = torch.ops.mi355x_graph.gspmm(csc.indptr, csc.indices, csc.eids, csc.num_cols, "copy_lhs", "sum", x, None)
        deg = torch.ops.mi355x_graph.in_degrees(csc.indptr)
    assert out.device == dev and deg.device == dev
    ref = torch.zeros(200, 16, device=dev).index_add_(0, torch.from_numpy(dst).to(dev), x[torch.from_numpy(src).to(dev)])
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
