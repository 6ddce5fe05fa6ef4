"""CPU suite: the multi-GPU path (partition + halo exchange + gradient all-reduce) with world_size 2
over gloo.  The arithmetic of each rank runs on the TEST-ONLY oracle backend; what is under test is
the host logic of mi355x_graph/dist.py: a P-way partitioned forward/backward must reproduce the
1-process result (SURVEY 8e "parity under sharding")."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mi355x_graph as mg
from mi355x_graph import dist as mdist
import oracle_backend

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dgl-0.5-benchmark_amd")


def make_ring_problem(world):
    """`world` UNEVEN blocks of nodes; edges inside a block and between block b and b + 1 only: non-adjacent peer pairs exchange
    nothing (empty halo both ways), adjacent pairs exchange different amounts, the assignment is given (no partitioner)."""
    g = torch.Generator().manual_seed(4)
    sizes = [30 + 17 * ((3 * b) % 5) for b in range(world)]
    starts = np.concatenate([[0], np.cumsum(sizes)])
    n = int(starts[-1])
    assign = torch.repeat_interleave(torch.arange(world), torch.tensor(sizes))
    src, dst = [], []
    for b in range(world):
        k = 6 * sizes[b]
        src.append(torch.randint(int(starts[b]), int(starts[b + 1]), (k,), generator=g))
        dst.append(torch.randint(int(starts[b]), int(starts[b + 1]), (k,), generator=g))
        if b + 1 < world:
            k = 2 * min(sizes[b], sizes[b + 1]) + 5 * b
            u = torch.randint(int(starts[b]), int(starts[b + 1]), (k,), generator=g)
            v = torch.randint(int(starts[b + 1]), int(starts[b + 2]), (k,), generator=g)
            src += [u, v]
            dst += [v, u]
    src, dst = torch.cat(src), torch.cat(dst)
    feats = torch.rand(n, 12, generator=g)
    labels = torch.randint(0, 5, (n,), generator=g)
    train = torch.rand(n, generator=g) < 0.3
    return n, src, dst, feats, labels, train, assign


def make_problem(ring_world=0):
    if ring_world:
        return make_ring_problem(ring_world)[:6]
    from mi355x_graph.datasets import synthetic_edges
    n = 600
    src, dst = synthetic_edges(n, 4000, 60, seed=3, symmetric=True)
    g = torch.Generator().manual_seed(0)
    feats = torch.rand(n, 12, generator=g)
    labels = torch.randint(0, 5, (n,), generator=g)
    train = torch.rand(n, generator=g) < 0.3
    return n, src, dst, feats, labels, train


def build_model(batch_norm=False):
    sys.path.insert(0, PKG)
    import full_graph
    torch.manual_seed(1)
    return full_graph.GraphSAGE(12, 8, 5, 3, dropout=0.0, batch_norm=batch_norm)


def single_process_reference(device="cpu", batch_norm=False, ring_world=0):
    import torch.nn.functional as F
    n, src, dst, feats, labels, train = [t.to(device) if isinstance(t, torch.Tensor) else t for t in make_problem(ring_world)]
    g = mg.graph((src, dst), num_nodes=n).int()
    model = build_model(batch_norm).to(device)
    out = model(g, feats)
    loss = F.nll_loss(out[train], labels[train])
    loss.backward()
    return (out.detach().cpu(), loss.item(), [p.grad.cpu().clone() for p in model.parameters()], generic_path(g, feats).cpu(),
            [b.cpu().clone() for b in model.buffers()])


def generic_path(g, x):
    """apply_edges(u_add_v) + update_all(u_mul_e, sum): on a DistGraph this takes the generic route
    (HaloExchange of every source field + the [owned | halo] block), not the overlapped copy_u route."""
    import mi355x_graph.function as fn
    g = g.local_var()
    g.ndata["a"] = x[:, :1].contiguous()
    g.ndata["h"] = x
    g.apply_edges(fn.u_add_v("a", "a", "w"))
    g.update_all(fn.u_mul_e("h", "w", "m"), fn.sum("m", "o"))
    return g.ndata["o"]


def _worker(rank, world, port, q, device="cpu", batch_norm=False, ring=False):
    import torch.nn.functional as F
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if device == "cpu":
        oracle_backend.install()  # CPU tensors: arithmetic by the test-only oracle backend
    n, src, dst, feats, labels, train = [t.to(device) if isinstance(t, torch.Tensor) else t for t in make_problem(world if ring else 0)]
    if ring:
        assign = make_ring_problem(world)[6].to(device)
        stats = {"edge_cut": float((assign[src] != assign[dst]).float().mean())}
    else:
        assign, stats = mdist.partition_nodes(src, dst, n, world)
    block, plan, own = mdist.build_local_partition(src, dst, n, assign, rank, world)
    g = mdist.DistGraph(block, plan)
    model = build_model(batch_norm).to(device)
    if batch_norm:  # statistics over the union of both ranks' rows
        model = mdist.convert_batchnorm(model)
    if device == "cpu":
        mdist.broadcast_parameters(model)  # same seed => same init; exercised on the CPU run only
    x, y, m = feats[own], labels[own], train[own]
    g.set_static_input(x)  # explicit declaration (every rank alike): the layer-1 halo rows stay resident
    total = torch.tensor([float(train.sum())], device=device)
    bucket = mdist.GradBucket(model)  # gradients are views into one flat buffer: one all_reduce, no copies (bench.py's path)
    bucket.zero()
    out = model(g, x)
    loss = F.nll_loss(out[m], y[m], reduction="sum") / total  # global mean
    loss.backward()
    assert all(prm.grad.data_ptr() >= bucket.flat.data_ptr() for prm in model.parameters())  # still the views
    bucket.all_reduce()
    # the hidden layers' inputs are relu outputs (tagged by ops.relu_dropout): their halo rows crossed as bitmaps + non-zeros and
    # their gradients came back under the same bitmaps (dist.SparseHalo) -- the layer-1 input is dense and static
    assert mdist.SPARSE_EXCHANGES == [len(model.layers) - 1] * 2, mdist.SPARSE_EXCHANGES
    lsum = loss.detach().cpu().clone()
    dist.all_reduce(lsum)
    q.put((rank, own.cpu().numpy(), out.detach().cpu().numpy(), float(lsum),
           [p.grad.cpu().numpy() for p in model.parameters()], stats, plan.n_halo, sum(plan.send_splits),
           generic_path(g, x).cpu().numpy(), [b.cpu().numpy() for b in model.buffers()]))
    # second training forward: the halo rows of the constant input features are resident, hidden layers exchange again
    n_before = g._comm.n_exchanges
    out2 = model(g, x)
    n_layers = len(model.layers)
    assert g._comm.n_exchanges - n_before == n_layers - 1, (n_before, g._comm.n_exchanges)
    assert torch.equal(out2, out) or torch.allclose(out2, out, rtol=1e-6, atol=1e-7)
    x.add_(0.0)  # an in-place write bumps the version counter: the cache must not be trusted any more
    n_before = g._comm.n_exchanges
    model(g, x)
    assert g._comm.n_exchanges - n_before == n_layers
    # identity, not address: a per-step temporary with the same values (e.g. feat_drop(feat)) never hits the cache
    n_before = g._comm.n_exchanges
    model(g, x.clone())
    assert g._comm.n_exchanges - n_before == n_layers
    n_before = g._comm.n_exchanges
    model(g, x)  # the declared tensor is still resident
    assert g._comm.n_exchanges - n_before == n_layers - 1
    dist.barrier()
    dist.destroy_process_group()


def _run_two_way(device, batch_norm=False, world=2, ring=False):
    if device == "cpu":
        oracle_backend.install()
    try:
        ref_out, ref_loss, ref_grads, ref_generic, ref_buffers = single_process_reference(device, batch_norm, world if ring else 0)
    finally:
        oracle_backend.uninstall()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + (7 if device != "cpu" else 0)
    port += 11 if batch_norm else 0
    port += 23 * (world - 2) + (5 if ring else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, device, batch_norm, ring)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got = torch.zeros_like(ref_out)
    got_generic = torch.zeros_like(ref_generic)
    seen = np.zeros(ref_out.shape[0], bool)
    for rank, own, out, lsum, grads, stats, n_halo, n_send, gen, buffers in res:
        assert len(buffers) == len(ref_buffers) and (len(buffers) > 0) == batch_norm
        for b, r in zip(buffers, ref_buffers):  # running_mean / running_var / num_batches_tracked
            assert np.allclose(b, r.numpy(), rtol=1e-4, atol=1e-6)
        got[own] = torch.from_numpy(out)
        got_generic[own] = torch.from_numpy(gen)
        assert not seen[own].any()
        seen[own] = True
        assert abs(lsum - ref_loss) < 1e-5
        for g, r in zip(grads, ref_grads):
            assert np.allclose(g, r.numpy(), rtol=1e-4, atol=1e-6)
        assert 0.0 < stats["edge_cut"] < 0.6 and (ring or (n_halo > 0 and n_send > 0))
    assert seen.all()
    assert torch.allclose(got, ref_out, rtol=1e-4, atol=1e-6)
    assert torch.allclose(got_generic, ref_generic, rtol=1e-4, atol=1e-6)


@pytest.mark.timeout(300)
def test_two_way_partition_matches_single_process():
    _run_two_way("cpu")


@pytest.mark.timeout(300)
def test_three_way_partition_matches_single_process():
    """world_size 3: uneven per-peer splits, peers with different halo sizes, the fixed per-peer scatter order."""
    _run_two_way("cpu", world=3)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [4, 8])
def test_ring_of_uneven_parts_matches_single_process(world):
    """world_size 4 and 8 with a GIVEN assignment: uneven parts, peer pairs with an empty halo in both directions (only
    adjacent blocks are linked), DistCopyU's split exchange and GradBucket's flat all_reduce across every rank."""
    _run_two_way("cpu", world=world, ring=True)


@pytest.mark.timeout(300)
def test_two_way_partition_with_batchnorm_matches_single_process():
    """arxiv-style model (BatchNorm1d between layers): GlobalBatchNorm1d reproduces the 1-process statistics,
    outputs, gradients and running estimates."""
    _run_two_way("cpu", batch_norm=True)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_way_partition_on_gpu_matches_single_process():
    """Two ranks sharing cuda:0 (gloo transport staged through the host): HIP kernels + HaloExchange +
    mgx_gather_rows / mgx_scatter_add_rows, against the 1-process GPU result."""
    _run_two_way("cuda:0")


def test_partition_is_balanced_and_cuts_less_than_random():
    from mi355x_graph.datasets import synthetic_edges
    n = 40000
    src, dst = synthetic_edges(n, 500000, 1000, seed=5, symmetric=True)  # 25 % of edges leave their community
    oracle_backend.install()
    try:
        for parts, bound in ((2, 0.25), (4, 0.40), (8, 0.55)):
            assign, stats = mdist.partition_nodes(src, dst, n, parts)
            load = torch.bincount(assign[dst], minlength=parts).float()
            assert load.max() / load.mean() < 1.10
            rnd = torch.randint(0, parts, (n,))
            assert stats["edge_cut"] < 0.7 * float((rnd[src] != rnd[dst]).float().mean())
            assert stats["edge_cut"] < bound
    finally:
        oracle_backend.uninstall()


def test_structural_zero_tag_is_voided_by_an_in_place_write():
    """dist.SparseHalo may drop gradient entries at the zeros of a relu (+ dropout) output only: the tag carries the tensor's version
    counter, so a later in-place write -- which may create zeros the producer's backward does not annihilate -- voids it."""
    from mi355x_graph import ops
    y = ops.relu_dropout(torch.randn(6, 8), 0.0, True)
    assert mdist.structural_zeros(y)
    assert not mdist.structural_zeros(y[:3]) and not mdist.structural_zeros(y * 1.0)      # new tensor objects carry no tag
    y.sub_(0.25)
    assert not mdist.structural_zeros(y)
    z = mdist.mark_structural_zeros(torch.relu(torch.randn(4, 8)))
    assert mdist.structural_zeros(z)
    z[0, 0] = 0.0
    assert not mdist.structural_zeros(z)


def test_halo_plan_consistency():
    """Every rank's receive list from q must equal q's send list to it (same order)."""
    from mi355x_graph.datasets import synthetic_edges
    n, world = 800, 4
    src, dst = synthetic_edges(n, 6000, 80, seed=9, symmetric=False)
    oracle_backend.install()
    try:
        assign, _ = mdist.partition_nodes(src, dst, n, world)
    finally:
        oracle_backend.uninstall()
    parts = [mdist.build_local_partition(src, dst, n, assign, r, world) for r in range(world)]
    for r, (block, plan, own) in enumerate(parts):
        assert plan.n_own == int((assign == r).sum())
        assert block.number_of_edges() == int((assign[dst] == r).sum())
        for q in range(world):
            assert plan.recv_splits[q] == parts[q][1].send_splits[r]
        assert plan.recv_splits[r] == 0 and plan.send_splits[r] == 0
        # global ids of what peers send me, in order == my halo order
        got = []
        for q in range(world):
            a, b = parts[q][1].send_ranges[r]
            got.append(parts[q][2][parts[q][1].send_idx[a:b].long()])
        got = torch.cat(got)
        ls, ld = block.edges()
        es, ed = src[assign[dst] == r], dst[assign[dst] == r]
        full_ids = torch.cat([own, got])
        assert torch.equal(full_ids[ls.long()], es) and torch.equal(own[ld.long()], ed)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_rccl_api_path_single_rank():
    """One rank over the real RCCL backend: exercises all_to_all_single(async_op=True) / Work.wait(), the flat gradient
    all_reduce and broadcast exactly as bench.py --gpus N calls them (degenerate sizes: no halo with one part)."""
    import torch.nn.functional as F
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(29500 + os.getpid() % 2000 + 13)
    dev = torch.device("cuda:0")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    except Exception as e:  # pragma: no cover
        pytest.skip("RCCL not usable here: %r" % (e,))
    try:
        n, src, dst, feats, labels, train = [t.to(dev) if isinstance(t, torch.Tensor) else t for t in make_problem()]
        assign, _ = mdist.partition_nodes(src, dst, n, 1)
        block, plan, own = mdist.build_local_partition(src, dst, n, assign, 0, 1)
        assert plan.n_halo == 0 and plan.n_own == n
        g = mdist.DistGraph(block, plan)
        model = build_model().to(dev)
        mdist.broadcast_parameters(model)
        g._comm.trace = []  # bench.py's measurement of the exposed exchange: events around every wait() on the RCCL work handle
        packed0 = mdist.SPARSE_EXCHANGES[0]
        out = model(g, feats)
        loss = F.nll_loss(out[train], labels[train])
        loss.backward()
        torch.cuda.synchronize()
        # one traced message per exchange, two (bitmaps, then values) for a forward exchange in the packed form (dist.SparseHalo)
        assert len(g._comm.trace) == g._comm.n_exchanges + (mdist.SPARSE_EXCHANGES[0] - packed0) > 0
        assert mdist.SPARSE_EXCHANGES[0] - packed0 == len(model.layers) - 1
        assert all(a.elapsed_time(b) >= 0.0 and nb == 0 for a, b, nb in g._comm.trace)  # one rank: nothing to receive
        g._comm.trace = None
        mdist.allreduce_gradients(model)
        ref_out, ref_loss, ref_grads, ref_gen, _ = single_process_reference("cuda:0")
        assert torch.allclose(out.detach().cpu(), ref_out, rtol=1e-4, atol=1e-6)
        for p, r in zip(model.parameters(), ref_grads):
            assert torch.allclose(p.grad.cpu(), r, rtol=1e-4, atol=1e-6)
        assert torch.allclose(generic_path(g, feats).cpu(), ref_gen, rtol=1e-4, atol=1e-6)
    finally:
        dist.destroy_process_group()


def test_metis_partition_api_for_cluster_gcn():
    """dgl.transform.metis_partition(g, psize) as cluster-sage/dgl/partition_utils.py:9-16 uses it,
    plus subgraph_collate_fn's g.subgraph(nids) (sampler.py:63-71)."""
    import dgl
    from dgl.transform import metis_partition
    from dgl import backend as F
    from mi355x_graph.datasets import synthetic_edges
    n = 6000
    src, dst = synthetic_edges(n, 60000, 300, seed=2, symmetric=True)
    g = dgl.graph((src, dst), num_nodes=n)
    g.ndata["feat"] = torch.rand(n, 4)
    parts = metis_partition(g, 50)
    nids = [F.asnumpy(val.ndata[dgl.NID]) for k, val in parts.items()]
    allids = np.concatenate(nids)
    assert len(allids) == n and len(np.unique(allids)) == n            # a partition of the node set
    sizes = np.array([len(x) for x in nids])
    assert sizes.max() < 6 * sizes.mean()
    intra = sum(val.number_of_edges() for val in parts.values()) / g.number_of_edges()
    rnd = torch.randint(0, 50, (n,))
    assert intra > 3 * float((rnd[src] == rnd[dst]).float().mean())    # far fewer cut edges than a random split
    batch = np.concatenate(nids[:5]).astype(np.int64)                    # ClusterIter batch -> subgraph_collate_fn
    g1 = g.subgraph(batch)
    nid = g1.ndata[dgl.NID]
    assert torch.equal(nid, torch.from_numpy(batch)) and torch.equal(g1.ndata["feat"], g.ndata["feat"][nid])
    s1, d1 = g1.edges()
    assert bool(g.has_edges_between(nid[s1.long()], nid[d1.long()]).all())
    g1.create_formats_()
    assert g.in_degree(0) == int((dst == 0).sum()) and g.out_degree(0) == int((src == 0).sum())
    fs, fd = g.find_edges(0)
    assert int(fs[0]) == int(src[0]) and int(fd[0]) == int(dst[0])
    # dgl_cluster_sampler.py:99-101: the {ntype: ids} spelling and the feature-name listing
    g2 = g.subgraph({"_U": batch})
    assert torch.equal(g2.edges()[0], s1) and torch.equal(g2.edges()[1], d1)
    schemes = g.node_attr_schemes()
    assert set(schemes) == set(g.ndata.keys()) and schemes["feat"].shape == tuple(g.ndata["feat"].shape[1:])
    with pytest.raises(dgl.DGLError):
        g.subgraph({"a": batch, "b": batch})


def test_cached_partition_roundtrip(tmp_path):
    """mdist.cached_partition: second call comes from disk and is identical; another graph of the same size does not."""
    from mi355x_graph.datasets import synthetic_edges
    n = 3000
    src, dst = synthetic_edges(n, 20000, 100, seed=1, symmetric=True)
    a1, s1 = mdist.cached_partition(src, dst, n, 4, cache_dir=str(tmp_path))
    a2, s2 = mdist.cached_partition(src, dst, n, 4, cache_dir=str(tmp_path))
    assert s1["cached"] is False and s2["cached"] is True and torch.equal(a1, a2)
    assert abs(s1["edge_cut"] - s2["edge_cut"]) < 1e-12
    src2, dst2 = synthetic_edges(n, 20000, 100, seed=2, symmetric=True)
    _, s3 = mdist.cached_partition(src2[:src.shape[0]], dst2[:src.shape[0]], n, 4, cache_dir=str(tmp_path))
    assert s3["cached"] is False


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_line_of_a_two_rank_run_matches_the_one_rank_run():
    """`python bench.py --gpus 2` end to end (self-launch -> torch.distributed.run children -> partition -> DistGraph model ->
    JSON line): both ranks on cuda:0 over gloo staging (MGX_BENCH_SHARE_GPU=1), a 2 % graph, dropout 0.  The line must say
    n_gpus 2, carry the partition statistics, and end at the loss of the 1-rank run of the same command."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--scale", "0.02", "--steps", "2", "--warmup", "1", "--no-pmc",
            "--no-controls", "--no-cpu-baseline", "--no-plain", "--dropout", "0"]
    env = dict(os.environ, MGX_BENCH_SHARE_GPU="1", MGX_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    lines = {}
    for gpus in (1, 2):
        p = subprocess.run(base + ["--gpus", str(gpus)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
        lines[gpus] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    one, two = lines[1], lines[2]
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["steps"] == 2 and two["warmup"] == 1
    part = two["config"]["partition"]
    assert 0.0 < part["edge_cut_pct"] < 60.0 and len(part["per_rank"]) == 2
    assert all(r["owned_rows"] > 0 and r["halo_rows"] > 0 and r["send_rows"] > 0 for r in part["per_rank"])
    assert sum(r["owned_rows"] for r in part["per_rank"]) == one["roofline"]["rows"]
    assert two["value"] > 0 and two["ms_per_step"] > 0 and two["scaling"] == "strong"
    assert abs(two["config"]["final_loss"] - one["config"]["final_loss"]) <= 1e-4 * max(1.0, abs(one["config"]["final_loss"]))
    assert two["roofline"] is not None and two["roofline"]["frac"] > 0
