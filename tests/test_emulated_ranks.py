"""P ranks of the partitioned program in ONE process (mi355x_graph/emulate.py).

CPU: the emulated world itself (uneven all_to_all, all_reduce in rank order, broadcast, failures and mismatched
collectives surface instead of hanging) and the 4- / 8-way ring partition through dist.py on the test-only oracle backend.
GPU (-m gpu): what VERDICT r03 item 1 asks -- for P = 4 and 8 on a skewed graph of > 1 M edges every rank's local partition is
built on cuda:0 and every rank's DistSageMeanCatFn forward + backward runs through the HIP kernels (strided / accumulating
aggregations, per-rank schedules, return_csr); the stitched output, the loss and every gradient are held to the 1-GPU HIP
result (SURVEY 8e "parity under sharding": P-way == 1-GPU to fp32 tolerance)."""
import os
import sys

import numpy as np
import pytest
import torch

import mi355x_graph as mg
from mi355x_graph import dist as mdist, emulate
import oracle_backend

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dgl-0.5-benchmark_amd")
sys.path.insert(0, PKG)


# ----------------------------------------------------------------------------- the emulated world itself (CPU)
def test_all_to_all_moves_uneven_slices_and_all_reduce_sums_in_rank_order():
    P = 5
    # rank r sends (r + q) % 3 rows to rank q, each row = [r, q, i]
    def rows(r, q):
        return (r + 2 * q) % 3 if r != q else 0

    def body(rank):
        ctx = emulate.current()
        assert ctx.rank == rank and ctx.size == P and mdist.world_size() == P
        send = torch.tensor([[rank, q, i] for q in range(P) for i in range(rows(rank, q))], dtype=torch.float32).view(-1, 3)
        in_splits = [rows(rank, q) for q in range(P)]
        out_splits = [rows(q, rank) for q in range(P)]
        recv = torch.full((sum(out_splits), 3), -1.0)
        work = mdist._Comm().all_to_all_async(recv, send, out_splits, in_splits)
        local = torch.tensor([float(rank)])           # work between post and wait
        work.wait()
        want = torch.tensor([[q, rank, i] for q in range(P) for i in range(rows(q, rank))], dtype=torch.float32).view(-1, 3)
        assert torch.equal(recv, want)
        t = torch.tensor([1.0 + rank, 10.0 * rank])
        mdist.all_reduce(t)
        assert torch.equal(t, torch.tensor([sum(1.0 + r for r in range(P)), 10.0 * sum(range(P))]))
        m = torch.tensor([float(rank)])
        mdist.all_reduce(m, op=torch.distributed.ReduceOp.MAX)
        assert float(m) == P - 1
        b = torch.tensor([float(rank)])
        mdist.broadcast(b, 3)
        assert float(b) == 3.0
        return rank * 2 + float(local)

    assert emulate.EmuWorld(P).run(body) == [3.0 * r for r in range(P)]


def test_a_failing_rank_and_a_mismatched_collective_raise_instead_of_hanging():
    def fails(rank):
        t = torch.ones(1)
        mdist.all_reduce(t)
        if rank == 2:
            raise ValueError("rank 2 gives up")
        mdist.all_reduce(t)

    with pytest.raises(ValueError, match="rank 2 gives up"):
        emulate.EmuWorld(4).run(fails)

    def mismatched(rank):
        t = torch.ones(1)
        if rank != 1:
            mdist.all_reduce(t)      # rank 1 never posts this one
        emulate.current().barrier()

    with pytest.raises(emulate.EmuError):
        emulate.EmuWorld(3).run(mismatched)

    def wrong_sizes(rank):
        send = torch.zeros((2, 1))
        recv = torch.zeros((3, 1))   # nobody sends 3 rows to anybody
        mdist._Comm().all_to_all(recv, send, [3, 0], [0, 2] if rank == 0 else [2, 0])

    with pytest.raises(emulate.EmuError, match="expects"):
        emulate.EmuWorld(2).run(wrong_sizes)


def test_trace_and_priced_epoch():
    """Two ranks, hand-made stretches: the priced epoch follows the slower link and hides what the window covers."""
    ranks = []
    for r in range(2):
        ranks.append([
            {"kind": "all_to_all", "info": {"recv_rows": [0, 1000] if r == 0 else [2000, 0], "row_bytes": 256},
             "pre": {"pack": 0.1}, "window": {"owned": 0.5 if r == 0 else 0.2}},
            {"kind": "all_reduce", "info": {"bytes": 4}, "pre": {"dense": 1.0}, "window": {}},
            {"kind": None, "pre": {"opt": 0.3}},
        ])
    # 100 GB/s, no latency: 1 -> 0 carries 256 kB = 2.56 us, 0 -> 1 carries 512 kB = 5.12 us; both hidden by the windows
    hid = emulate.price_epoch(ranks, 100.0, latency_us=0.0, allreduce_us=0.0)
    assert abs(hid["epoch_ms"] - (0.1 + 0.5 + 1.0 + 0.3)) < 1e-9
    assert all(x["exposed"] == 0.0 for x in hid["exchanges"][0])
    # 1 GB/s: 0.512 ms each way needed -> both ranks wait until 0.1 + 0.512; rank 1's window is 0.2
    slow = emulate.price_epoch(ranks, 1.0, latency_us=0.0, allreduce_us=0.0)
    assert abs(slow["exchanges"][0][1]["exposed"] - (0.512 - 0.2)) < 1e-9
    assert abs(slow["exchanges"][0][0]["exposed"] - (0.512 - 0.5)) < 1e-9
    assert abs(slow["epoch_ms"] - (0.1 + 0.512 + 1.0 + 0.3)) < 1e-9
    ser = emulate.price_epoch(ranks, 1.0, latency_us=0.0, allreduce_us=0.0, overlap=False)
    assert abs(ser["epoch_ms"] - (0.1 + 0.5 + 0.512 + 1.0 + 0.3)) < 1e-9

    # a real trace on the CPU clock: structure and labels survive typical_epoch
    def body(rank):
        ctx = emulate.current()
        traces = []
        for _ in range(3):
            ctx.start_trace()
            comm = mdist._Comm()
            comm.mark("pack")
            send, recv = torch.ones((2, 4)), torch.zeros((2, 4))
            w = comm.all_to_all_async(recv, send, [2, 0] if rank else [0, 2], [2, 0] if rank else [0, 2])
            comm.mark("window")
            w.wait()
            comm.mark("after")
            mdist.all_reduce(torch.ones(1))
            traces.append(ctx.stop_trace())
        return emulate.typical_epoch(traces)

    tl = emulate.EmuWorld(2).run(body)
    assert [s["kind"] for s in tl[0]] == ["all_to_all", "all_reduce", None]
    assert "pack" in tl[0][0]["pre"] and "window" in tl[0][0]["window"] and "after" in tl[1][1]["pre"]
    assert tl[0][0]["info"]["recv_rows"] == [0, 2] and tl[0][0]["info"]["row_bytes"] == 16
    assert emulate.price_epoch(tl, 64.0)["epoch_ms"] > 0


def test_solo_epochs_replay_the_recorded_payloads():
    """record_epoch() keeps what a rank received; solo_epochs() runs the rank's epochs alone with every collective completed from the
    kept payloads (same sizes: copied; another size: zeros), counts the collectives, and leaves the live world usable afterwards."""
    seen = {}

    def body(rank):
        ctx = emulate.current()
        comm = mdist._Comm()
        state = {"recv": None, "red": None, "odd": None, "extra": False}

        def step():
            send = torch.full((2, 4), float(rank + 1))
            recv = torch.zeros((2, 4))
            w = comm.all_to_all_async(recv, send, [2, 0] if rank else [0, 2], [2, 0] if rank else [0, 2])
            w.wait()
            red = torch.full((3,), float(rank + 1))
            mdist.all_reduce(red)
            n = 5 if ctx._solo is not None else 3          # a message whose size changes from epoch to epoch
            odd = torch.ones((n, 2))
            comm.all_to_all_async(odd, torch.ones((n, 2)), [n, 0] if rank else [0, n], [n, 0] if rank else [0, n]).wait()
            if state["extra"]:
                mdist.all_reduce(torch.ones(1))
            state.update(recv=recv, red=red, odd=odd)

        kept = ctx.record_epoch(step)
        assert len(kept) == 3
        live = (state["recv"].clone(), state["red"].clone())
        ctx.barrier()
        ms = ctx.solo_epochs(step, kept, epochs=2, warmup=1)
        assert ms >= 0 and ctx._solo is None
        assert torch.equal(state["recv"], live[0]) and torch.equal(state["red"], live[1])   # the kept payloads
        assert state["odd"].shape == (5, 2) and float(state["odd"].abs().sum()) == 0.0      # no counterpart of that size: zeros
        state["extra"] = True
        with pytest.raises(emulate.EmuError):
            ctx.solo_epochs(step, kept, epochs=1, warmup=0)                                  # one collective more than recorded
        state["extra"] = False
        ctx.barrier()
        step()                                                                                # the live world still works
        seen[rank] = float(state["recv"][0, 0])
        return ms

    emulate.EmuWorld(2).run(body)
    assert seen == {0: 2.0, 1: 1.0}


@pytest.mark.parametrize("world", [4, 8])
def test_emulated_ring_partition_matches_single_process_on_cpu(world):
    """The same ring problem tests/test_dist.py runs over gloo processes, as threads of one process."""
    import torch.nn.functional as F
    import test_dist
    oracle_backend.install()
    try:
        ref_out, ref_loss, ref_grads, ref_generic, _ = test_dist.single_process_reference("cpu", False, world)
        n, src, dst, feats, labels, train, assign = test_dist.make_ring_problem(world)

        def body(rank):
            block, plan, own = mdist.build_local_partition(src, dst, n, assign, rank, world)
            g = mdist.DistGraph(block, plan)
            model = test_dist.build_model()
            mdist.broadcast_parameters(model)
            x, y, m = feats[own], labels[own], train[own]
            g.set_static_input(x)
            bucket = mdist.GradBucket(model)
            bucket.zero()
            out = model(g, x)
            loss = F.nll_loss(out[m], y[m], reduction="sum") / float(train.sum())
            loss.backward()
            bucket.all_reduce()
            lsum = loss.detach().clone()
            mdist.all_reduce(lsum)
            return own, out.detach(), float(lsum), [p.grad.clone() for p in model.parameters()], test_dist.generic_path(g, x)

        res = emulate.EmuWorld(world).run(body)
    finally:
        oracle_backend.uninstall()
    got, gen = torch.zeros_like(ref_out), torch.zeros_like(ref_generic)
    for own, out, lsum, grads, generic in res:
        got[own], gen[own] = out, generic
        assert abs(lsum - ref_loss) < 1e-5
        for a, b in zip(grads, ref_grads):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    assert torch.allclose(got, ref_out, rtol=1e-4, atol=1e-6) and torch.allclose(gen, ref_generic, rtol=1e-4, atol=1e-6)


# ----------------------------------------------------------------------------- every rank on the HIP path (GPU)
def _products_like(device, scale_nodes=40000, edges=600000):
    from mi355x_graph.datasets import synthetic_edges
    src, dst = synthetic_edges(scale_nodes, edges, 3000, seed=11, device=device, symmetric=True)  # 1.2 M directed, power law
    gen = torch.Generator().manual_seed(5)
    feats = torch.rand(scale_nodes, 100, generator=gen)
    labels = torch.randint(0, 47, (scale_nodes,), generator=gen)
    train = torch.rand(scale_nodes, generator=gen) < 0.08
    return scale_nodes, src, dst, feats, labels, train


def _products_model(device):
    import full_graph
    torch.manual_seed(77)
    # dropout 1e-12: the fused relu_dropout kernel runs (and writes the next layer's [h | neigh] buffer, so every layer is the
    # one-GEMM form, as in bench.py) but its 32-bit threshold p * 2^32 is 0 and the scale 1/(1-p) is 1.0f: no element is dropped
    m = full_graph.GraphSAGE(100, 64, 47, 3, 1e-12, False, True).to(device)
    m.rows_are_distinct = True
    return m


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [4, 8])
def test_every_rank_of_a_partition_on_the_hip_path_matches_one_gpu(world):
    from mi355x_graph import ops
    dev = torch.device("cuda:0")
    n, src, dst, feats, labels, train = _products_like(dev)
    assert src.shape[0] >= 1_000_000
    indeg = torch.bincount(dst, minlength=n)
    assert int(indeg.max()) > 40 * float(indeg.float().mean())   # skewed: hub rows far above the mean degree
    total_train = float(train.sum())

    # ---- 1-GPU reference: the bench's step (default model, loss on the train rows)
    g1 = mg.graph((src, dst), num_nodes=n).int().formats(["csr", "csc"]).to(dev)
    x1, y1 = feats.to(dev), labels.to(dev)
    idx1 = torch.nonzero(train).flatten().to(dev)
    ref_model = _products_model(dev)
    ref_model.train()
    ref_loss = ops.nll_sum(ref_model(g1, x1, rows=idx1), y1[idx1]) / total_train
    ref_loss.backward()
    ref_grads = [p.grad.clone() for p in ref_model.parameters()]
    ref_out = ref_model(g1, x1).detach()
    assert g1._index.csc().plan() is not None

    assign, stats = mdist.partition_nodes(src, dst, n, world)
    assert 0.0 < stats["edge_cut"] < 0.9
    parts = [mdist.build_local_partition(src, dst, n, assign, r, world) for r in range(world)]
    taken = {"cat": 0}
    orig = mdist.DistSageMeanCatFn.forward

    def counting(*a, **k):
        taken["cat"] += 1
        return orig(*a, **k)

    def body(rank):
        block, plan, own = parts[rank]
        g = mdist.DistGraph(block, plan)
        model = _products_model(dev)
        mdist.broadcast_parameters(model)
        own_c = own.cpu()
        x, y = feats[own_c].to(dev), labels[own_c].to(dev)
        g.set_static_input(x)
        idx = torch.nonzero(train[own_c]).flatten().to(dev)
        bucket = mdist.GradBucket(model)
        model.train()
        losses = []
        for _ in range(2):  # twice: the second pass takes the resident layer-1 halo and the kept CatBuffer copy
            bucket.zero()
            loss = ops.nll_sum(model(g, x, rows=idx), y[idx]) / total_train
            loss.backward()
            bucket.all_reduce()
            lsum = loss.detach().clone()
            mdist.all_reduce(lsum)
            losses.append(float(lsum))
        out = model(g, x).detach()
        return own, out, losses, [p.grad.clone() for p in model.parameters()], plan

    mdist.DistSageMeanCatFn.forward = staticmethod(counting)
    packed0 = list(mdist.SPARSE_EXCHANGES)
    try:
        res = emulate.EmuWorld(world, dev).run(body)
    finally:
        mdist.DistSageMeanCatFn.forward = staticmethod(orig)
    assert taken["cat"] == world * 3 * 3   # every layer of every rank took the one-GEMM layer with the exchange inside
    # the two hidden layers' inputs are relu(+dropout) outputs: their halo rows crossed as bitmaps + non-zeros in all three forward
    # passes of every rank, their gradients came back under the same bitmaps in both backward passes (dist.SparseHalo, csrc/rowpack.hip)
    assert [a - b for a, b in zip(mdist.SPARSE_EXCHANGES, packed0)] == [world * 2 * 3, world * 2 * 2]

    got = torch.full_like(ref_out, float("nan"))
    for own, out, losses, grads, plan in res:
        got[own] = out
        for l in losses:
            assert abs(l - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
        assert plan.n_halo > 0 and sum(plan.send_splits) > 0 and plan.return_csr().nnz == sum(plan.send_splits)
        for a, b in zip(grads, ref_grads):
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= 1e-4 * max(scale, 1e-3), (float((a - b).abs().max()), scale)
    err = (got - ref_out).abs()
    assert not torch.isnan(got).any()
    assert float(err.max()) <= 1e-4 * max(1.0, float(ref_out.abs().max()))
    # every rank ends on the same gradients bit for bit (rank-ordered sum)
    for a, b in zip(res[0][3], res[-1][3]):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_fused_activation_on_a_partition_is_its_composition(monkeypatch):
    """DistGraph.sage_mean_layer_act (relu + dropout in the layer GEMM's epilogue, written into the next layer's buffer, tagged for the
    packed halo exchange) against the composition it stands for -- the layer, then ops.relu_dropout -- with REAL dropout: the same
    position in the random stream, so the same bits: losses and every gradient of every rank are equal."""
    import full_graph
    from mi355x_graph import config as mgx_config, ops
    dev = torch.device("cuda:0")
    world = 4
    n, src, dst, feats, labels, train = _products_like(dev)
    total_train = float(train.sum())
    assign, _ = mdist.partition_nodes(src, dst, n, world)
    parts = [mdist.build_local_partition(src, dst, n, assign, r, world) for r in range(world)]
    monkeypatch.setattr(ops, "_ROWS_GEMM_MIN", 0)      # the partitions of the test graph are below the tall-matrix threshold
    taken = {"act": 0}
    orig = mdist.DistSageMeanCatFn.forward

    def counting(ctx, *a):
        taken["act"] += 1 if len(a) > 9 and a[9] is not None else 0
        return orig(ctx, *a)

    def body(rank):
        block, plan, own = parts[rank]
        g = mdist.DistGraph(block, plan)
        torch.manual_seed(77)
        model = full_graph.GraphSAGE(100, 64, 47, 3, 0.5, False, True).to(dev)
        model.rows_are_distinct = True
        mdist.broadcast_parameters(model)
        own_c = own.cpu()
        x, y = feats[own_c].to(dev), labels[own_c].to(dev)
        g.set_static_input(x)
        idx = torch.nonzero(train[own_c]).flatten().to(dev)
        bucket = mdist.GradBucket(model)
        model.train()
        losses = []
        for _ in range(2):
            bucket.zero()
            loss = ops.nll_sum(model(g, x, rows=idx), y[idx]) / total_train
            loss.backward()
            bucket.all_reduce()
            losses.append(float(loss))
        return losses, [p.grad.clone() for p in model.parameters()]

    results = {}
    for fused in (True, False):
        monkeypatch.setattr(mgx_config, "SAGE_FUSED_ACT", fused)
        ops.ReluDropout._calls = 0
        torch.manual_seed(5)
        taken["act"] = 0
        packed0 = list(mdist.SPARSE_EXCHANGES)
        mdist.DistSageMeanCatFn.forward = staticmethod(counting)
        try:
            results[fused] = emulate.EmuWorld(world, dev).run(body)
        finally:
            mdist.DistSageMeanCatFn.forward = staticmethod(orig)
        assert taken["act"] == (world * 2 * 2 if fused else 0)        # both hidden layers of every rank, both passes
        # either way the hidden layers' inputs crossed as bitmaps + non-zeros: the fused form tags its output as ops.relu_dropout does
        assert [a - b for a, b in zip(mdist.SPARSE_EXCHANGES, packed0)] == [world * 2 * 2, world * 2 * 2]
    for (la, ga), (lb, gb) in zip(results[True], results[False]):
        assert la == lb
        for a, b in zip(ga, gb):
            assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_emulate_ranks_line_carries_the_scaling_model(tmp_path):
    """`python bench.py --emulate-ranks 2,4` end to end on a 2 % graph: the N = 1 line gains config.partition.predicted with, per P, the
    per-rank compute, the halo bytes and the priced epochs; the emulated partitions end on one and the same loss (dropout 0); the report
    file is written."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    report = str(tmp_path / "scale.txt")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--scale", "0.02", "--steps", "2", "--warmup", "1", "--dropout", "0",
           "--emulate-ranks", "2,4", "--report", report]
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ms_per_step"] > 0
    pred = line["config"]["partition"]["predicted"]
    assert "MODEL" in pred["kind"] and abs(pred["epoch_ms_1gpu_measured"] - line["ms_per_step"]) < 1e-2
    for P in (2, 4):
        m = pred["P=%d" % P]
        assert "error" not in m, m
        assert len(m["per_rank"]) == P and all(r["owned_rows"] > 0 and r["compute_ms"] > 0 for r in m["per_rank"])
        assert sum(r["owned_rows"] for r in m["per_rank"]) == line["roofline"]["rows"]
        # every emulated world takes the same number of optimizer steps (not the headline's): they end on the same loss
        assert abs(m["final_loss"] - pred["P=2"]["final_loss"]) <= 1e-4 * max(1.0, abs(pred["P=2"]["final_loss"]))
        assert m["exchanges_per_epoch"] == 4 and m["max_pair_bytes_per_exchange"] > 0
        fast, slow = m["predicted"]["64 GB/s per link"], m["predicted"]["32 GB/s per link"]
        assert 0 < fast["epoch_ms_overlapped"] <= slow["epoch_ms_overlapped"]
        assert fast["epoch_ms_overlapped"] <= fast["epoch_ms_not_overlapped"] + 1e-9
        assert m["predicted"]["no exchange cost (compute only, lock step)"]["epoch_ms_overlapped"] <= fast["epoch_ms_overlapped"] + 1e-9
    text = open(report).read()
    assert "== P = 2" in text and "== P = 4" in text and "halo MB per exchange" in text
    # the emulated world against a REAL 2-process run of the same program (both ranks on cuda:0, gloo staging): the emulation took
    # warm-up 1 + one discarded traced step + 2 timed steps = 4 optimizer steps, so does `--warmup 1 --steps 3`
    real = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--scale", "0.02", "--steps", "3", "--warmup", "1",
                           "--dropout", "0", "--no-pmc", "--no-controls", "--no-cpu-baseline", "--no-plain"],
                          env=dict(env, MGX_BENCH_SHARE_GPU="1", MGX_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert real.returncode == 0, real.stdout[-1500:] + real.stderr[-3000:]
    two = json.loads([ln for ln in real.stdout.splitlines() if ln.startswith("{")][-1])
    assert two["n_gpus"] == 2
    assert abs(two["config"]["final_loss"] - pred["P=2"]["final_loss"]) <= 1e-4 * max(1.0, abs(two["config"]["final_loss"]))
    real_rows = sorted(r["owned_rows"] for r in two["config"]["partition"]["per_rank"])
    assert real_rows == sorted(r["owned_rows"] for r in pred["P=2"]["per_rank"])            # the same partition
    assert sorted(r["halo_rows"] for r in two["config"]["partition"]["per_rank"]) == sorted(r["halo_rows"] for r in pred["P=2"]["per_rank"])


def _gat_problem(world):
    import test_dist
    n, src, dst, feats, labels, train, assign = test_dist.make_ring_problem(world)
    loops = torch.arange(n)
    return n, torch.cat([src, loops]), torch.cat([dst, loops]), feats, labels, train, assign   # dgl.add_self_loop: no 0-in-degree node


def _gat_model(g, device="cpu"):
    import full_graph
    torch.manual_seed(3)
    # layer 1: 12 -> 2 heads x 8 (input narrower than its projection: the INPUT rows travel); layer 2: 16 -> 1 head x 5 (the PROJECTED rows)
    return full_graph.GAT(g, 2, 12, 8, 5, [2, 1], feat_drop=0.0, attn_drop=0.0).to(device)


def _run_gat_partition(world, device):
    import torch.nn.functional as F
    n, src, dst, feats, labels, train, assign = [t.to(device) if isinstance(t, torch.Tensor) else t for t in _gat_problem(world)]
    g1 = mg.graph((src, dst), num_nodes=n).int()
    ref_model = _gat_model(g1, device)
    ref_out = ref_model(feats)
    ref_loss = F.nll_loss(ref_out.log_softmax(-1)[train], labels[train])
    ref_loss.backward()
    ref_grads = [p.grad.clone() for p in ref_model.parameters()]

    def body(rank):
        block, plan, own = mdist.build_local_partition(src, dst, n, assign, rank, world)
        g = mdist.DistGraph(block, plan)
        model = _gat_model(g, device)
        mdist.broadcast_parameters(model)
        x, y, m = feats[own], labels[own], train[own]
        bucket = mdist.GradBucket(model)
        bucket.zero()
        out = model(x)
        loss = F.nll_loss(out.log_softmax(-1)[m], y[m], reduction="sum") / float(train.sum())
        loss.backward()
        bucket.all_reduce()
        lsum = loss.detach().clone()
        mdist.all_reduce(lsum)
        return own, out.detach(), float(lsum), [p.grad.clone() for p in model.parameters()], g._comm.n_exchanges

    res = emulate.EmuWorld(world, None if device == "cpu" else device).run(body)
    got = torch.zeros_like(ref_out.detach())
    for own, out, lsum, grads, n_ex in res:
        got[own] = out
        assert n_ex == 3        # one exchange per layer forward, one backward for layer 2 (the input features need no gradient)
        assert abs(lsum - float(ref_loss)) < 1e-5 * max(1.0, abs(float(ref_loss)))
        for a, b in zip(grads, ref_grads):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), float((a - b).abs().max())
    assert torch.allclose(got, ref_out.detach(), rtol=1e-4, atol=1e-6)


def test_gat_on_a_partition_matches_single_process_on_cpu():
    """GATConv on a dist.DistGraph (main_dgl_reddit_gat.py's model, partitioned): the layer's input rows or its projected rows cross the
    halo exchange once per layer, the block then runs as a sampled block does; 4 ranks, uneven ring, against the 1-process model."""
    oracle_backend.install()
    try:
        _run_gat_partition(4, "cpu")
    finally:
        oracle_backend.uninstall()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_gat_on_a_partition_on_the_hip_path():
    """The same on cuda:0: the fused GAT kernels on the bipartite [owned | halo] -> owned blocks of every rank."""
    _run_gat_partition(4, "cuda:0")


def _run_sage_project_first(world, device):
    """dist.DistSageProjectFirstFn (reddit's 602 -> 16 layer on a partition: the boundary rows travel at the OUTPUT width): a 2-layer SAGE
    whose first layer narrows 40 -> 8 (constant input), against the 1-process model."""
    import torch.nn.functional as F
    import test_dist
    import full_graph
    n, src, dst, feats, labels, train, assign = [t.to(device) if isinstance(t, torch.Tensor) else t for t in test_dist.make_ring_problem(world)]
    torch.manual_seed(9)
    wide = torch.rand(n, 40).to(device)

    def model():
        torch.manual_seed(4)
        return full_graph.GraphSAGE(40, 8, 5, 2, dropout=0.0).to(device)

    g1 = mg.graph((src, dst), num_nodes=n).int()
    ref = model()
    ref_out = ref(g1, wide)
    F.nll_loss(ref_out[train], labels[train]).backward()
    ref_grads = [p.grad.clone() for p in ref.parameters()]
    taken = {"n": 0}
    orig = mdist.DistSageProjectFirstFn.forward

    def counting(*a, **k):
        taken["n"] += 1
        return orig(*a, **k)

    def body(rank):
        block, plan, own = mdist.build_local_partition(src, dst, n, assign, rank, world)
        g = mdist.DistGraph(block, plan)
        m = model()
        mdist.broadcast_parameters(m)
        x, y, msk = wide[own], labels[own], train[own]
        bucket = mdist.GradBucket(m)
        bucket.zero()
        out = m(g, x)
        (F.nll_loss(out[msk], y[msk], reduction="sum") / float(train.sum())).backward()
        bucket.all_reduce()
        return own, out.detach(), [p.grad.clone() for p in m.parameters()]

    mdist.DistSageProjectFirstFn.forward = staticmethod(counting)
    try:
        res = emulate.EmuWorld(world, None if device == "cpu" else device).run(body)
    finally:
        mdist.DistSageProjectFirstFn.forward = staticmethod(orig)
    assert taken["n"] == world          # layer 1 (40 -> 8, constant input) projects first on every rank; layer 2 (8 -> 5) does not
    got = torch.zeros_like(ref_out.detach())
    for own, out, grads in res:
        got[own] = out
        for a, b in zip(grads, ref_grads):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), float((a - b).abs().max())
    assert torch.allclose(got, ref_out.detach(), rtol=1e-4, atol=1e-6)


def test_sage_projecting_first_on_a_partition_matches_single_process_on_cpu():
    oracle_backend.install()
    try:
        _run_sage_project_first(3, "cpu")
    finally:
        oracle_backend.uninstall()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_sage_projecting_first_on_a_partition_on_the_hip_path():
    _run_sage_project_first(4, "cuda:0")
