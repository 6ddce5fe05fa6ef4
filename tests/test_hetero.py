"""Heterograph multi-relation dispatch.  The same checks run on CPU tensors (arithmetic by the
test-only oracle backend: API / host logic) and, marked gpu, on the MI355X (each relation's update_all / apply_edges goes
through the HIP g-SpMM / g-SDDMM kernels on that relation's own in-CSR).  References are dense fp64 products."""
import numpy as np
import pytest
import torch

import mi355x_graph as mg
import mi355x_graph.function as fn
from mi355x_graph.nn import HeteroGraphConv


@pytest.fixture
def cpu_backend():
    import oracle_backend
    oracle_backend.install()
    yield "cpu"
    oracle_backend.uninstall()


def rating_graph(nu, nm, per_rel, seed, device):
    """users x movies, one relation + its reverse per rating value (the GCMC encoder graph)."""
    rng = np.random.default_rng(seed)
    rels, dense = [], {}
    for r, k in enumerate(per_rel):
        u, m = rng.integers(0, nu, k), rng.integers(0, nm, k)
        rels.append(mg.bipartite((u, m), "user", "r%d" % r, "movie", num_nodes=(nu, nm)))
        rels.append(mg.bipartite((m, u), "movie", "rev-r%d" % r, "user", num_nodes=(nm, nu)))
        a = np.zeros((nm, nu))
        np.add.at(a, (m, u), 1.0)
        dense["r%d" % r] = a          # movie <- user
        dense["rev-r%d" % r] = a.T    # user <- movie
    return mg.hetero_from_relations(rels).int().to(device), dense


def check_hetero(device):
    nu, nm, D = 70, 45, 12
    g, dense = rating_graph(nu, nm, [300, 0, 150, 500], seed=3, device=device)
    assert g.ntypes == ["movie", "user"] and len(g.etypes) == 8 and g.number_of_edges("r1") == 0
    assert g.number_of_nodes("user") == nu and g.number_of_nodes() == nu + nm
    assert g.number_of_edges() == 2 * (300 + 150 + 500)
    with pytest.raises(mg.DGLError):
        g.number_of_edges("nope")
    with pytest.raises(mg.DGLError):
        g.update_all(fn.copy_u("h", "m"), fn.sum("m", "o"))  # several edge types: etype is required
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(device)
    rng = np.random.default_rng(0)
    xu, xm = rng.random((nu, D)), rng.random((nm, D))
    g.nodes["user"].data["h"] = t(xu)
    g.nodes["movie"].data["h"] = t(xm)
    # relation views share storage with the heterograph
    view = g["r0"]
    assert not view.is_block and view.number_of_src_nodes() == nu and view.number_of_dst_nodes() == nm
    assert view.srcdata["h"] is g.nodes["user"].data["h"]
    assert torch.equal(view.in_degrees().cpu().long(), torch.from_numpy(dense["r0"].sum(1)).long())
    assert torch.equal(g["rev-r0"].out_degrees().cpu().long(), torch.from_numpy(dense["r0"].sum(1)).long())
    # one relation
    g.update_all(fn.copy_u("h", "m"), fn.mean("m", "o"), etype="r3")
    deg = np.maximum(dense["r3"].sum(1, keepdims=True), 1)
    assert np.allclose(g.nodes["movie"].data["o"].cpu().numpy(), dense["r3"] @ xu / deg, rtol=1e-4, atol=1e-5)
    # all relations, every cross reducer
    funcs = {e: (fn.copy_u("h", "m"), fn.sum("m", "agg")) for e in g.etypes}
    parts_m = [dense["r%d" % r] @ xu for r in (0, 2, 3)]     # r1 has no edges: skipped, as in DGL
    parts_u = [dense["rev-r%d" % r] @ xm for r in (0, 2, 3)]
    for red, ref in (("sum", np.sum), ("mean", np.mean), ("max", np.max), ("min", np.min)):
        g.multi_update_all(funcs, red)
        assert np.allclose(g.nodes["movie"].data["agg"].cpu().numpy(), ref(np.stack(parts_m), 0), rtol=1e-4, atol=1e-5)
        assert np.allclose(g.nodes["user"].data["agg"].cpu().numpy(), ref(np.stack(parts_u), 0), rtol=1e-4, atol=1e-5)
    g.multi_update_all(funcs, "stack")
    assert np.allclose(g.nodes["movie"].data["agg"].cpu().numpy(), np.stack(parts_m, 1), rtol=1e-4, atol=1e-5)
    # apply_edges on a single-relation (decoder) graph, through nodes[...].data and edata
    u, m = rng.integers(0, nu, 400), rng.integers(0, nm, 400)
    dec = mg.bipartite((u, m), "user", "rate", "movie", num_nodes=(nu, nm)).int().to(device)
    with dec.local_scope():
        dec.nodes["user"].data["h"] = t(xu)
        dec.nodes["movie"].data["h"] = t(xm)
        dec.apply_edges(fn.u_dot_v("h", "h", "sr"))
        sr = dec.edata["sr"].cpu().numpy()
    assert "sr" not in dec.edata and np.allclose(sr[:, 0], (xu[u] * xm[m]).sum(1), rtol=1e-4, atol=1e-5)

    # HeteroGraphConv: per-relation module + weights, gradients reach every relation's weight
    class Rel(torch.nn.Module):
        def __init__(self):
            super(Rel, self).__init__()
            self.w = torch.nn.Parameter(torch.from_numpy(rng.random((D, 5)).astype(np.float32)))

        def forward(self, graph, feat, scale=1.0):
            with graph.local_scope():
                graph.srcdata["x"] = feat[0] @ self.w * scale
                graph.update_all(fn.copy_u("x", "m"), fn.sum("m", "y"))
                return graph.dstdata["y"]

    conv = HeteroGraphConv({e: Rel() for e in g.etypes}, aggregate="stack").to(device)
    out = conv(g, {"user": t(xu), "movie": t(xm)}, mod_kwargs={"r0": {"scale": 2.0}})
    assert out["movie"].shape == (nm, 3, 5) and out["user"].shape == (nu, 3, 5)
    ref0 = 2.0 * dense["r0"] @ (xu @ conv.mods["r0"].w.detach().cpu().numpy().astype(np.float64))
    assert np.allclose(out["movie"][:, 0].detach().cpu().numpy(), ref0, rtol=1e-4, atol=1e-4)
    (out["movie"].sum() + out["user"].sum()).backward()
    for e in ("r0", "r2", "r3", "rev-r0", "rev-r2", "rev-r3"):
        assert conv.mods[e].w.grad is not None and float(conv.mods[e].w.grad.abs().sum()) > 0
    assert conv.mods["r1"].w.grad is None  # empty relation: never called
    gw = conv.mods["r3"].w.grad.cpu().numpy()
    assert np.allclose(gw, xu.T @ dense["r3"].T @ np.ones((nm, 5)), rtol=1e-4, atol=1e-4)


def test_heterograph_cpu(cpu_backend):
    check_hetero("cpu")


@pytest.mark.gpu
def test_heterograph_gpu():
    check_hetero("cuda:0")


def test_heterograph_constructor_and_errors():
    g = mg.heterograph({("user", "follows", "user"): ([0, 1], [1, 2]), ("user", "plays", "game"): ([0, 2], [1, 0])},
                       num_nodes_dict={"user": 4, "game": 3})
    assert g.ntypes == ["game", "user"] and g.etypes == ["follows", "plays"]
    assert g.number_of_nodes("user") == 4 and g.number_of_nodes("game") == 3
    assert g["follows"].number_of_nodes() == 4 and not g["follows"].is_block
    assert g.to_canonical_etype("plays") == ("user", "plays", "game")
    s, d = g.edges(etype="plays")
    assert s.tolist() == [0, 2] and d.tolist() == [1, 0]
    g.nodes["game"].data["x"] = torch.zeros(3, 2)
    assert "x" in g["plays"].dstdata and "x" not in g["plays"].srcdata
    with pytest.raises(mg.DGLError):
        g.nodes["game"].data["bad"] = torch.zeros(4, 2)
    with pytest.raises(mg.DGLError):
        mg.heterograph({("a", "e", "b"): ([0], [5])}, num_nodes_dict={"a": 1, "b": 2})
    with pytest.raises(mg.DGLError):
        mg.bipartite(([0], [0]), "a", "e", "a")
    sub = g.edge_type_subgraph(["plays"])
    assert sub.etypes == ["plays"] and sub.number_of_edges() == 2
    assert isinstance(g.ndata["x"], dict) and list(g.ndata["x"]) == ["game"]
