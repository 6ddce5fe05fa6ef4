"""GPU suite: the model families of the reference scripts (SAGE, GAT, molhiv GCN) run through the
DGL surface on the HIP backend and are compared -- outputs, loss and every parameter gradient --
against a restatement in plain PyTorch gather/index_add ops on the same device (dropout = 0)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import mi355x_graph as mg
from conftest import random_graph

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dgl-0.5-benchmark_amd")
sys.path.insert(0, PKG)
DEV = "cuda:0"


def nerr(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def grads_close(got, params, tol):
    """Per-tensor error relative to that tensor's scale; tensors whose true gradient is ~0 (e.g. attention
    vectors the softmax is nearly invariant to) are held to the scale of the largest gradient instead."""
    gmax = max(float(p.grad.abs().max()) for p in params if p.grad is not None)
    for a, p in zip(got, params):
        if p.grad is None:
            continue
        scale = max(float(p.grad.abs().max()), 1e-3 * gmax)
        assert float((a - p.grad).abs().max()) / scale < tol


def agg(dst, msg, n):
    return torch.zeros((n,) + msg.shape[1:], device=msg.device).index_add(0, dst, msg)


def test_sage_products_model_matches_gather_reference():
    import dgl
    import full_graph
    n, nnz = 3000, 40000
    src, dst = random_graph(n, n, nnz, seed=4)
    s, d = torch.from_numpy(src).to(DEV), torch.from_numpy(dst).to(DEV)
    g = dgl.graph((s, d), num_nodes=n).int().formats(["csr", "csc"])
    torch.manual_seed(0)
    model = full_graph.GraphSAGE(100, 64, 47, 3, dropout=0.0).to(DEV)
    x = torch.rand(n, 100, device=DEV)
    y = torch.randint(0, 47, (n,), device=DEV)
    idx = torch.nonzero(torch.rand(n, device=DEV) < 0.1).flatten()
    loss = F.nll_loss(model(g, x)[idx], y[idx])
    loss.backward()
    got = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    deg = torch.bincount(d, minlength=n).clamp(min=1).float()[:, None]
    h = x
    for i, layer in enumerate(model.layers):
        neigh = agg(d, h[s], n) / deg
        h = layer.fc_self(h) + layer.fc_neigh(neigh)
        if i < 2:
            h = F.relu(h)
    ref_loss = F.nll_loss(h.log_softmax(-1)[idx], y[idx])
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    grads_close(got, list(model.parameters()), 1e-4)


@pytest.mark.parametrize("heads,layers", [(8, 2), (1, 3)])
def test_gat_matches_gather_reference(heads, layers):
    import dgl
    import full_graph
    n, nnz = 1500, 20000
    src, dst = random_graph(n, n, nnz, seed=6)
    g = dgl.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=n)
    g = dgl.add_self_loop(g).int().to(DEV)
    s, d = g.edges()
    s, d = s.long(), d.long()
    torch.manual_seed(0)
    hs = [heads] * (layers - 1) + [1]
    model = full_graph.GAT(g, layers, 32, 16, 7, hs).to(DEV)
    x = torch.rand(n, 32, device=DEV)
    y = torch.randint(0, 7, (n,), device=DEV)
    mask = torch.rand(n, device=DEV) < 0.3
    loss = F.cross_entropy(model(x)[mask], y[mask])
    loss.backward()
    got = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()

    def gat_layer(conv, h):  # main_pyg_reddit_gat.py:99-112 states the same math
        H, Fo = conv._num_heads, conv._out_feats
        ft = conv.fc(h).view(n, H, Fo)
        el = (ft * conv.attn_l).sum(-1)
        er = (ft * conv.attn_r).sum(-1)
        e = F.leaky_relu(el[s] + er[d], 0.2)
        m = torch.full((n, H), -1e30, device=DEV).scatter_reduce(0, d[:, None].expand(-1, H), e, "amax")
        ex = torch.exp(e - m[d])
        a = ex / agg(d, ex, n)[d]
        out = agg(d, ft[s] * a[:, :, None], n)
        out = out + conv.bias.view(1, H, Fo)
        return conv.activation(out) if conv.activation else out

    h = x
    for l in range(layers - 1):
        h = gat_layer(model.gat_layers[l], h).flatten(1)
    ref = gat_layer(model.gat_layers[-1], h).mean(1)
    ref_loss = F.cross_entropy(ref[mask], y[mask])
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    grads_close(got, list(model.parameters()), 2e-4)
    with pytest.raises(mg.DGLError, match="0-in-degree"):
        g0 = dgl.graph((torch.tensor([0, 1]), torch.tensor([1, 2])), num_nodes=4).int().to(DEV)
        model.gat_layers[0](g0, torch.rand(4, 32, device=DEV))


def test_molhiv_gcn_batched_udf_and_readout():
    import dgl
    import graph_classification as gc
    from mi355x_graph.datasets import molhiv_like
    from dgl.dataloading import GraphDataLoader
    data = molhiv_like(num_graphs=96, seed=5)
    loader = GraphDataLoader(data, batch_size=32, shuffle=False)
    torch.manual_seed(0)
    model = gc.GCN(64, 1, 3, dropout=0.0).to(DEV)
    bg, labels = next(iter(loader))
    assert bg.batch_size == 32 and int(bg.batch_num_nodes().sum()) == bg.number_of_nodes()
    bg = bg.to(DEV).int().formats("coo")
    atom, bond = bg.ndata["feat"], bg.edata["feat"]
    out = model(bg, atom, bond)
    loss = F.binary_cross_entropy_with_logits(out.view(-1), labels.to(DEV).float())
    loss.backward()
    got = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    s, d = bg.edges()
    s, d = s.long(), d.long()
    n = bg.number_of_nodes()
    deg = torch.bincount(d, minlength=n).float()[:, None] + 1
    c = deg.pow(-0.5)
    x = model.atom_encoder(atom)
    for i, layer in enumerate(model.layers):
        xx = layer.fc(x)
        w = layer.bond_encoder(bond)
        h = agg(d, c[s] * c[d] * F.relu(xx[s] + w), n)
        x = h + F.relu(xx + layer.root_emb.weight) * 1. / deg
        if i < len(model.layers) - 1:
            x = F.relu(model.bns[i](x))
    seg = torch.repeat_interleave(torch.arange(32, device=DEV), bg.batch_num_nodes())
    pooled = agg(seg, x, 32) / bg.batch_num_nodes().float()[:, None]
    ref = model.graph_pred_fc(pooled)
    assert nerr(out, ref) < 1e-4
    F.binary_cross_entropy_with_logits(ref.view(-1), labels.to(DEV).float()).backward()
    grads_close(got, list(model.parameters()), 2e-4)
    # bit-exact integer work on the batch: degrees and batch offsets
    assert torch.equal(bg.in_degrees().long(), torch.bincount(d, minlength=n))
    parts = dgl.unbatch(bg.to("cpu"))
    assert [p.number_of_nodes() for p in parts] == bg.batch_num_nodes().tolist()


def test_gatconv_wide_heads_bias_and_grad():
    """heads * out_feats > 256 (8 x 41, reddit/ns-gat-dgl.py's output layer): the bias gradient leaves the column-sum
    kernel's range and must fall back to the plain broadcast add with the (1, H, F) bias shape."""
    import mi355x_graph as mg
    from mi355x_graph.nn import GATConv
    from conftest import random_graph
    n, nnz, H, F = 300, 4000, 8, 41
    src, dst = random_graph(n, n, nnz, seed=4)
    g = mg.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=n).add_self_loop().int().to("cuda:0")
    conv = GATConv(20, F, H).to("cuda:0")
    torch.nn.init.normal_(conv.bias)
    x = torch.rand(n, 20, device="cuda:0")
    out = conv(g, x)
    assert out.shape == (n, H, F)
    conv.bias.data.zero_()
    out0 = conv(g, x)
    torch.nn.init.normal_(conv.bias)
    out1 = conv(g, x)
    assert torch.allclose(out1 - out0, conv.bias.view(1, H, F).expand_as(out0), atol=1e-5)
    out1.sum().backward()
    assert torch.allclose(conv.bias.grad, torch.full_like(conv.bias, float(n)), rtol=1e-5)


def test_hip_graph_captured_training_step_matches_eager():
    """utils.GraphedStep: forward + backward + Adam of a 2-layer GraphSAGE and of a GAT captured in a HIP graph give the
    eager loss trajectory (no dropout: bitwise-deterministic kernels, so the trajectories agree to rounding)."""
    sys.path.insert(0, PKG)
    import full_graph
    from mi355x_graph.utils import GraphedStep
    from mi355x_graph.datasets import synthetic_edges
    dev = "cuda:0"
    n = 3000
    src, dst = synthetic_edges(n, 20000, 200, seed=5, symmetric=True)
    g = mg.graph((src, dst), num_nodes=n).add_self_loop().int().formats(["csr", "csc"]).to(dev)
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(n, 24, generator=gen).to(dev)
    y = torch.randint(0, 6, (n,), generator=gen).to(dev)
    idx = torch.nonzero(torch.rand(n, generator=gen) < 0.3).flatten().to(dev)
    builders = {"sage": lambda: full_graph.GraphSAGE(24, 16, 6, 2, dropout=0.0).to(dev),
                "gat": lambda: full_graph.GAT(g, 2, 24, 8, 6, [4, 1], feat_drop=0.0, attn_drop=0.0).to(dev)}
    for name, fwd in (("sage", lambda m: m(g, x)), ("gat", lambda m: m(x))):
        torch.manual_seed(3)
        model = builders[name]()
        eager = builders[name]()
        eager.load_state_dict(model.state_dict())
        opt_e = torch.optim.Adam(eager.parameters(), lr=0.01)
        opt_g = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
        model.train()
        eager.train()
        warm = 2
        step = GraphedStep(lambda: F.nll_loss(F.log_softmax(fwd(model), -1)[idx], y[idx]), opt_g, warmup=warm)
        losses_e = []
        for i in range(warm + 6):
            opt_e.zero_grad()
            le = F.nll_loss(F.log_softmax(fwd(eager), -1)[idx], y[idx])
            le.backward()
            opt_e.step()
            losses_e.append(float(le.detach()))
        losses_g = [float(step()) for _ in range(6)]
        assert np.allclose(losses_g, losses_e[warm:], rtol=2e-4), (name, losses_g, losses_e[warm:])
        assert losses_g[-1] < losses_g[0]


def test_ginconv_matches_dense_formulation():
    """dgl.nn.GINConv: (1 + eps) h_v + sum / mean / max over in-neighbours, then the apply function; gradients reach eps."""
    from mi355x_graph.nn import GINConv
    n, nnz, D = 400, 3000, 12
    src, dst = random_graph(n, n, nnz, seed=8)
    g = mg.graph((torch.from_numpy(src), torch.from_numpy(dst)), num_nodes=n).int().to("cuda:0")
    x = torch.randn(n, D, device="cuda:0")
    A = torch.zeros(n, n, dtype=torch.float64, device="cuda:0")
    A.index_put_((torch.from_numpy(dst).cuda(), torch.from_numpy(src).cuda()), torch.ones(nnz, dtype=torch.float64, device="cuda:0"), accumulate=True)
    lin = torch.nn.Linear(D, 5).to("cuda:0")
    for agg in ("sum", "mean"):
        conv = GINConv(lin, agg, init_eps=0.3, learn_eps=True).to("cuda:0")
        out = conv(g, x)
        neigh = A @ x.double()
        if agg == "mean":
            neigh = neigh / A.sum(1, keepdim=True).clamp(min=1)
        ref = lin((1.3 * x.double() + neigh).float())
        assert nerr(out, ref) < 1e-4
        out.sum().backward()
        assert conv.eps.grad is not None and float(conv.eps.grad.abs()) > 0
    out = GINConv(None, "max").to("cuda:0")(g, x)
    dense = torch.where(A.bool().unsqueeze(-1), x.unsqueeze(0).expand(n, n, D), torch.full((1, 1, 1), -float("inf"), device="cuda:0"))
    mx = dense.max(1)[0]
    mx = torch.where(torch.isinf(mx), torch.zeros_like(mx), mx)
    assert nerr(out, x + mx) < 1e-5


def test_hip_graph_capture_with_dropout_takes_the_same_forms_in_warm_up():
    """GraphedStep with dropout > 0: under capture relu_dropout is the two PyTorch ops (a replay must draw a new mask), which
    also decides which form the next SAGE layer takes (one GEMM over a CatBuffer only behind the fused kernel).  The warm-up
    steps must take the captured forms (ops.warming_up_for_capture), or the capture meets a BLAS path -- and its handle
    creation -- for the first time (regression: hipblasCreate inside the capture)."""
    sys.path.insert(0, PKG)
    import full_graph
    from mi355x_graph.utils import GraphedStep
    from mi355x_graph.datasets import synthetic_edges
    dev = "cuda:0"
    n = 4000
    src, dst = synthetic_edges(n, 30000, 200, seed=6, symmetric=True)
    g = mg.graph((src, dst), num_nodes=n).int().formats(["csr", "csc"]).to(dev)
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(n, 20, generator=gen).to(dev)
    y = torch.randint(0, 5, (n,), generator=gen).to(dev)
    idx = torch.nonzero(torch.rand(n, generator=gen) < 0.3).flatten().to(dev)
    torch.manual_seed(4)
    model = full_graph.GraphSAGE(20, 16, 5, 3, dropout=0.5).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
    model.train()
    step = GraphedStep(lambda: F.nll_loss(model(g, x)[idx], y[idx]), opt, warmup=2)
    losses = [float(step()) for _ in range(12)]
    assert all(np.isfinite(losses)) and min(losses[6:]) < losses[0]
    assert len({round(v, 6) for v in losses}) > 6          # replays draw new dropout masks
