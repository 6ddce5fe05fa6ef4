"""transform.GraphPool: the small graphs of a dataset in one storage, batch() of any subset from the pooled arrays, GraphDataLoader over
indices -- the host side of BASELINE configs[4] (main_dgl_molhiv_gcn.py:163: a shuffled mini-batch of 256 molecules per step).  Everything
here is integer / copy work: bit for bit against the general path (four torch.cat over the graphs of the batch)."""
import numpy as np
import torch

from mi355x_graph import dataloading, transform
from mi355x_graph.datasets import SmallGraphDataset, Subset, molhiv_like


def _general_batch(graphs):
    saved = [g._pool for g in graphs]
    for g in graphs:
        g._pool = None
    try:
        return transform.batch(graphs)
    finally:
        for g, p in zip(graphs, saved):
            g._pool = p


def _same(a, b):
    assert a.number_of_nodes() == b.number_of_nodes() and a.number_of_edges() == b.number_of_edges()
    assert torch.equal(a.edges()[0], b.edges()[0]) and torch.equal(a.edges()[1], b.edges()[1]) and a.idtype == b.idtype
    assert torch.equal(a.batch_num_nodes(), b.batch_num_nodes()) and torch.equal(a.batch_num_edges(), b.batch_num_edges())
    assert sorted(a.ndata.keys()) == sorted(b.ndata.keys()) and sorted(a.edata.keys()) == sorted(b.edata.keys())
    for k in a.ndata.keys():
        assert torch.equal(a.ndata[k], b.ndata[k])
    for k in a.edata.keys():
        assert torch.equal(a.edata[k], b.edata[k])


def test_pooled_batch_equals_the_general_path():
    data = molhiv_like(600)
    assert data.pool is not None and data.pool.members_clean()
    rng = np.random.default_rng(0)
    for size in (1, 2, 37, 256):
        ids = rng.permutation(len(data))[:size]
        graphs = [data[i][0] for i in ids]
        pooled = transform.batch(graphs)
        assert transform.GraphPool.of(graphs) is not None
        _same(pooled, _general_batch(graphs))
        parts = transform.unbatch(pooled)
        assert len(parts) == size and all(torch.equal(p.ndata["feat"], g.ndata["feat"]) and torch.equal(p.edges()[0], g.edges()[0])
                                          for p, g in zip(parts, graphs))
    # float features, no edge features (the ENZYMES stand-in)
    enz = SmallGraphDataset(80, 30.0, 10.0, 2, 90, 18, 0, 6, seed=9, int_features=False)
    graphs = [enz[i][0] for i in (5, 3, 70, 3)]      # a graph may appear twice in a batch
    _same(transform.batch(graphs), _general_batch(graphs))


def test_adopted_graphs_are_views_and_reassignment_leaves_the_pool():
    data = molhiv_like(50)
    g = data[7][0]
    g.ndata["feat"][0, 0] = 9                        # an in-place write lands in the pool: the pooled batch sees it
    b = transform.batch([data[7][0], data[8][0]])
    assert int(b.ndata["feat"][0, 0]) == 9
    g.ndata["feat"] = g.ndata["feat"] + 1            # a reassigned field: this graph is no member any more
    graphs = [data[7][0], data[8][0]]
    assert transform.GraphPool.of(graphs) is None and not data.pool.members_clean()
    out = transform.batch(graphs)                    # the general path, with the new values
    assert torch.equal(out.ndata["feat"][:g.number_of_nodes()], g.ndata["feat"])
    loader = dataloading.GraphDataLoader(data, batch_size=16)   # a dirty pool: the loader takes the general path as well
    assert not hasattr(loader, "pooled_dataset")
    assert sum(int(bg.batch_size) if hasattr(bg, "batch_size") else int(lab.shape[0]) for bg, lab in loader) == 50


def test_loader_over_indices_yields_the_same_batches():
    data = molhiv_like(700)
    for dataset in (data, Subset(data, list(range(50, 650, 2)))):
        torch.manual_seed(3)
        fast = list(dataloading.GraphDataLoader(dataset, batch_size=64, shuffle=True))
        torch.manual_seed(3)
        slow = list(dataloading.GraphDataLoader(dataset, batch_size=64, shuffle=True, collate_fn=dataloading._collate))
        assert len(fast) == len(slow) > 1
        for (ga, la), (gb, lb) in zip(fast, slow):
            _same(ga, gb)
            assert torch.equal(la, lb)
    assert hasattr(dataloading.GraphDataLoader(data, batch_size=8), "pooled_dataset")


def test_lists_that_do_not_qualify_are_left_alone():
    import mi355x_graph as mg
    a = mg.graph((torch.tensor([0, 1]), torch.tensor([1, 2])), num_nodes=3)
    b = mg.graph((torch.tensor([0]), torch.tensor([1])), num_nodes=2)
    a.ndata["x"] = torch.rand(3, 2)
    b.ndata["y"] = torch.rand(2, 2)                  # different field names
    assert transform.GraphPool.adopt([a, b]) is None and getattr(a, "_pool", None) is None
    b2 = mg.graph((torch.tensor([0]), torch.tensor([1])), num_nodes=2).int()
    a2 = mg.graph((torch.tensor([0]), torch.tensor([1])), num_nodes=2)
    assert transform.GraphPool.adopt([a2, b2]) is None          # different index widths
