"""Neighbor sampling / blocks.  CPU: sampler invariants (index work only).  GPU: the hot-path
kernels on sampled blocks (N_src != N_dst) against the oracle, and one mini-batch SAGE step against a gather
restatement."""
import numpy as np
import pytest
import torch

import mi355x_graph as mg
from mi355x_graph import sampling
from mi355x_graph.datasets import synthetic_edges


def make_graph(n=3000, m=40000, device="cpu"):
    s, d = synthetic_edges(n, m, 300, seed=11, symmetric=True)
    return mg.graph((s.to(device), d.to(device)), num_nodes=n), s, d


def test_sample_neighbors_invariants():
    g, s, d = make_graph()
    n = g.number_of_nodes()
    seeds = torch.randperm(n)[:500]
    gen = torch.Generator().manual_seed(0)
    src, dst, eid = sampling.sample_neighbors(g, seeds, 7, generator=gen)
    # every sampled edge exists (by edge id), belongs to a seed, and no seed has more than `fanout`
    assert torch.equal(s[eid], src) and torch.equal(d[eid], dst)
    assert bool(torch.isin(dst, seeds).all())
    deg = torch.bincount(d, minlength=n)
    cnt = torch.bincount(dst, minlength=n)
    assert torch.equal(cnt[seeds], torch.clamp(deg[seeds], max=7))   # rows with deg <= fanout keep ALL their edges
    assert eid.unique().numel() == eid.numel()                        # without replacement
    # fanout None = all in-edges
    src_all, dst_all, eid_all = sampling.sample_neighbors(g, seeds, None)
    assert int(eid_all.numel()) == int(deg[seeds].sum())
    # uniformity (coarse): over many draws each in-edge of a hub is picked about equally often
    hub = int(torch.argmax(deg))
    picks = torch.zeros(int(deg[hub]))
    pos = {int(e): i for i, e in enumerate(torch.nonzero(d == hub).flatten().tolist())}
    for t in range(300):
        _, _, e = sampling.sample_neighbors(g, torch.tensor([hub]), 10)
        for x in e.tolist():
            picks[pos[x]] += 1
    expect = 300 * 10 / float(deg[hub])
    assert float(picks.std()) < 4 * np.sqrt(expect) and float(picks.sum()) == 3000


def test_blocks_and_loader():
    g, s, d = make_graph()
    sampler = sampling.MultiLayerNeighborSampler([5, 10])
    loader = sampling.NodeDataLoader(g, torch.arange(1000), sampler, batch_size=256, shuffle=True, drop_last=False)
    assert len(loader) == 4
    seen = 0
    for inp, out, blocks in loader:
        b0, b1 = blocks
        seen += out.numel()
        assert b1.is_block and b1.number_of_dst_nodes() == out.numel()
        assert torch.equal(b1.srcdata[sampling.NID][:out.numel()], out)           # dst nodes are a prefix of src nodes
        assert torch.equal(b0.dstdata[sampling.NID], b1.srcdata[sampling.NID])    # layers chain
        assert torch.equal(b0.srcdata[sampling.NID], inp)
        assert int(b1.in_degrees().max()) <= 10 and int(b0.in_degrees().max()) <= 5
        for b in blocks:  # local edges map back to real global edges
            ls, ld = b.edges()
            gs, gd = b.srcdata[sampling.NID][ls.long()], b.dstdata[sampling.NID][ld.long()]
            e = b.edata[sampling.EID]
            assert torch.equal(s[e], gs) and torch.equal(d[e], gd)
    assert seen == 1000
    sub = g.subgraph(torch.arange(0, 3000, 2))
    ss, sd = sub.edges()
    assert bool(((sub.ndata[sampling.NID][ss.long()] % 2) == 0).all()) and sub.number_of_nodes() == 1500


@pytest.mark.gpu
def test_block_kernels_and_minibatch_step_on_gpu(oracle):
    import torch.nn.functional as F
    from mi355x_graph import ops
    from mi355x_graph.nn import SAGEConv, GATConv
    dev = "cuda:0"
    g, s, d = make_graph(device=dev)
    n = g.number_of_nodes()
    feats = torch.rand(n, 32, device=dev)
    sampler = sampling.MultiLayerNeighborSampler([10, 25])
    loader = sampling.NodeDataLoader(g, torch.arange(600), sampler, batch_size=300, shuffle=False)
    inp, out, blocks = next(iter(loader))
    blocks = [b.int() for b in blocks]
    x = feats[inp]
    # kernels on a block vs the oracle
    b = blocks[0]
    ls, ld = [t.cpu().numpy() for t in b.edges()]
    ip, ix, ei = oracle.coo_to_csr(b.number_of_dst_nodes(), ld, ls)
    got = ops.gspmm(b, "copy_lhs", "mean", x, None)
    ref = oracle.spmm(ip, ix, ei, "copy_lhs", "mean", x.cpu().numpy(), None)
    assert got.shape == (b.number_of_dst_nodes(), 32)
    assert float(np.max(np.abs(got.cpu().numpy() - ref) / (np.abs(ref) + 1e-5))) < 1e-4
    # a 2-layer mini-batch SAGE step vs a gather restatement
    torch.manual_seed(0)
    l0, l1 = SAGEConv(32, 16, "mean").to(dev), SAGEConv(16, 5, "mean").to(dev)
    y = torch.randint(0, 5, (out.numel(),), device=dev)
    h = l1(blocks[1], F.relu(l0(blocks[0], x)))
    loss = F.cross_entropy(h, y)
    loss.backward()
    got_grads = [p.grad.clone() for p in list(l0.parameters()) + list(l1.parameters())]
    for p in list(l0.parameters()) + list(l1.parameters()):
        p.grad = None

    def layer(conv, blk, hh):
        es, ed = [t.long() for t in blk.edges()]
        nd = blk.number_of_dst_nodes()
        deg = torch.bincount(ed, minlength=nd).clamp(min=1).float()[:, None]
        if conv._in_src_feats > conv._out_feats:  # lin_before_mp
            neigh = torch.zeros(nd, conv._out_feats, device=dev).index_add(0, ed, conv.fc_neigh(hh)[es]) / deg
        else:
            neigh = conv.fc_neigh(torch.zeros(nd, hh.shape[1], device=dev).index_add(0, ed, hh[es]) / deg)
        return conv.fc_self(hh[:nd]) + neigh
    ref_loss = F.cross_entropy(layer(l1, blocks[1], F.relu(layer(l0, blocks[0], x))), y)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    for a, p in zip(got_grads, list(l0.parameters()) + list(l1.parameters())):
        assert float((a - p.grad).abs().max() / p.grad.abs().max().clamp(min=1e-9)) < 2e-4
    # GATConv accepts blocks too
    conv = GATConv(32, 8, 2, allow_zero_in_degree=True).to(dev)
    assert conv(blocks[0], x).shape == (blocks[0].number_of_dst_nodes(), 2, 8)


@pytest.mark.gpu
def test_device_sampler_invariants_and_uniformity():
    """mgx_sample_neighbors (HIP): same invariants as the torch formulation, reproducible per seed, uniform."""
    dev = "cuda:0"
    g, s, d = make_graph(n=3000, m=60000, device=dev)
    g = g.int()
    n = g.number_of_nodes()
    s, d = s.to(dev), d.to(dev)
    deg = torch.bincount(d, minlength=n)
    seeds = torch.randperm(n, device=dev)[:700]
    for fanout in (1, 7, 25, 64):
        gen = torch.Generator().manual_seed(fanout)
        src, dst, eid = sampling.sample_neighbors(g, seeds, fanout, generator=gen)
        assert torch.equal(s[eid], src) and torch.equal(d[eid], dst)
        cnt = torch.bincount(dst, minlength=n)
        assert torch.equal(cnt[seeds], torch.clamp(deg[seeds], max=fanout))
        assert eid.unique().numel() == eid.numel()
        # grouped by seed in seed order, CSR positions ascending inside a seed
        seg = torch.repeat_interleave(torch.arange(seeds.numel(), device=dev), cnt[seeds])
        assert torch.equal(dst, seeds[seg])
        src2, dst2, eid2 = sampling.sample_neighbors(g, seeds, fanout, generator=torch.Generator().manual_seed(fanout))
        assert torch.equal(eid, eid2)  # same generator state -> same sample
    hub = int(torch.argmax(deg))
    in_edges = torch.nonzero(d == hub).flatten()
    picks = torch.zeros(n * 0 + int(s.numel()), device=dev)
    draws = 2000
    hub_t = torch.full((draws,), hub, device=dev)  # the same seed many times in one call: independent draws per slot
    _, _, e = sampling.sample_neighbors(g, hub_t, 10)
    picks.index_add_(0, e, torch.ones(e.numel(), device=dev))
    p = picks[in_edges]
    expect = draws * 10 / float(deg[hub])
    assert float(p.sum()) == draws * 10 and float(picks.sum()) == draws * 10
    assert float((p - expect).abs().max()) < 6 * np.sqrt(expect)  # binomial spread; a biased sampler fails this
