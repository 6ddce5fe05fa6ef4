#!/usr/bin/env python
"""bench.py -- full-graph GraphSAGE on an ogbn-products-shaped graph (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`,
   one rank per GPU over RCCL)

A "step" is one training epoch of main_dgl_product_sage.py's loop (forward, nll_loss on the train
split, backward, Adam step, loss.item() as the host sync; lines 101-110) on the synthetic
products-shaped graph (N = 2,449,029, E = 123,718,280 directed, D = 100 -> 64 -> 64 -> 47, fp32).
`value` = aggregated edges/s = (3 forward + 2 backward g-SpMMs) * E / epoch time, whole job.
N > 1 partitions the SAME graph (edge-cut + RCCL all_to_all halo exchange): strong scaling.

The JSON line also carries
  roofline      the dominant hot-path kernel (copy_u/sum g-SpMM at D = 64, 4 launches per epoch):
                algorithmic bytes (4(N+1) + 4E + 4ND + 4ND) / its mean launch duration,
                measured live with HIP events on the launch stream, against the 8 TB/s HBM peak;
  cpu_baseline  the CPU oracle (OpenMP port of DGL's CPU algorithm) timed on the host cores on the
                g-SpMMs of one epoch (rank 0, N = 1 only, bounded to ~30 s).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "dgl-0.5-benchmark_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# Dense layers of the (unmodified) model are rocBLAS/hipBLASLt GEMMs; their default heuristics pick 3 ms kernels for
# the tall-skinny weight-gradient shapes (K = 2.45 M).  PyTorch's TunableOp selections for exactly these shapes --
# the full graph and the per-rank owned-row counts of the deterministic P = 2, 4, 8 partitions -- were recorded once on
# MI355X by experiments/tune_dense.py (dgl-0.5-benchmark_amd/tunableop_products<device>.csv) and are only LOADED here
# (tuning off), which is a user-level PyTorch setting, not part of the message-passing library.
if os.environ.get("MGX_BENCH_TUNABLEOP", "1") == "1":
    os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
    os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "0")
    os.environ.setdefault("PYTORCH_TUNABLEOP_RECORD_UNTUNED", "0")
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", os.path.join(PKG, "tunableop_products.csv"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def spmm_algorithmic_bytes(n_dst, n_src, nnz, D):
    """indptr + indices + every source row once + every output row once."""
    return 4 * (n_dst + 1) + 4 * nnz + 4 * n_src * D + 4 * n_dst * D


def host_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(g, feat_dim, hidden, budget_s=30.0):
    """Times the oracle's g-SpMM (row-parallel OpenMP, sequential in-row fp32 accumulation -- the
    algorithm of DGL's CPU kernel) on the epoch's five aggregations, inputs already in host memory."""
    import numpy as np
    from oracle import oracle as orc
    orc.build()
    native = orc.use_native_build()  # timed on this host: compile for its ISA (falls back to the shipped x86-64-v3 build)
    cores = host_cores()
    orc.set_num_threads(cores)
    csc, csr = g._index.csc(), g._index.csr()
    n, nnz = csc.num_rows, csc.nnz
    host = {}
    for name, v in (("csc", csc), ("csr", csr)):
        host[name] = (v.indptr.cpu().numpy().astype(np.int32), v.indices.cpu().numpy().astype(np.int32))
    rng = np.random.default_rng(0)
    plan = [("fwd L1 copy_u/mean D=%d" % feat_dim, "csc", feat_dim, "mean"),
            ("fwd L2 copy_u/mean D=%d" % hidden, "csc", hidden, "mean"),
            ("fwd L3 copy_u/mean D=%d" % hidden, "csc", hidden, "mean"),
            ("bwd L3 copy_u/sum D=%d" % hidden, "csr", hidden, "sum"),
            ("bwd L2 copy_u/sum D=%d" % hidden, "csr", hidden, "sum")]
    feats = {}
    done, spent, edges = [], 0.0, 0
    for label, fmt, D, red in plan:
        if spent > budget_s:
            break
        if D not in feats:
            feats[D] = rng.random((n, D), dtype=np.float32)
        ip, ix = host[fmt]
        t0 = time.perf_counter()
        orc.spmm(ip, ix, None, "copy_lhs", red, feats[D], None)
        dt = time.perf_counter() - t0
        spent += dt
        edges += nnz
        done.append("%s: %.2fs" % (label, dt))
    out = {"value": edges / spent, "unit": "edges/s", "cores": cores, "kind": "port", "isa": "native" if native else "x86-64-v3",
           "sample": "g-SpMM part only (no dense layers): %d of the epoch's 5 aggregations on the full graph "
                     "(N=%d, E=%d), OpenMP over rows; %s" % (len(done), n, nnz, "; ".join(done))}
    # independent second CPU number: PyTorch's own CSR SpMM on the same cores, one D=hidden aggregation
    try:
        torch.set_num_threads(cores)
        ip, ix = host["csr"]
        a = torch.sparse_csr_tensor(torch.from_numpy(ip), torch.from_numpy(ix), torch.ones(nnz), size=(n, n))
        x = torch.from_numpy(feats[hidden])
        a @ x[:, :1].contiguous()  # builds any internal handle outside the timed call
        t0 = time.perf_counter()
        a @ x
        dt = time.perf_counter() - t0
        out["second"] = {"value": nnz / dt, "unit": "edges/s", "kind": "torch.sparse_csr @ X (copy_u/sum, D=%d)" % hidden,
                         "cores": cores, "seconds": round(dt, 3)}
    except Exception as err:  # a missing CPU sparse kernel must not lose the bench line
        out["second"] = {"value": None, "kind": "torch.sparse_csr @ X", "error": str(err)[:200]}
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--dataset", default="products")
    p.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debug only; makes the line invalid)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    args = p.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the message-passing library has no CPU path")
    # MGX_BENCH_SHARE_GPU=1 + MGX_DIST_BACKEND=gloo: smoke-run the N > 1 code path with all ranks on cuda:0
    # (1-GPU boxes); the driver's real runs use one GPU per rank over RCCL.
    share = os.environ.get("MGX_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("MGX_DIST_BACKEND", "nccl")
    device = torch.device("cuda:%d" % (0 if share else local_rank))
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    import dgl
    import full_graph
    from mi355x_graph import dist as mdist, sparse
    from mi355x_graph.datasets import SHAPES, synthetic_edges

    cfg = full_graph.SAGE_CONFIGS[args.dataset]
    spec = SHAPES[cfg["dataset"]]
    n = max(64, int(spec["n"] * args.scale))
    m = max(64, int(spec["m"] * args.scale))
    src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], device, symmetric=spec["symmetric"])
    num_edges = int(src.shape[0])
    gen = torch.Generator(device="cpu").manual_seed(spec["seed"] + 100)
    feats = torch.rand(n, spec["feat"], generator=gen)
    labels = torch.randint(0, spec["classes"], (n,), generator=gen)
    train_mask = torch.rand(n, generator=gen) < 0.08  # ogbn-products trains on 8 % of the nodes

    torch.manual_seed(1234)
    model = full_graph.GraphSAGE(spec["feat"], cfg["hidden"], spec["classes"], cfg["num_layers"], cfg["dropout"],
                                 cfg["batch_norm"], cfg["neigh_bias"]).to(device)
    part_stats = {}
    if world == 1:
        g = dgl.graph((src, dst), num_nodes=n).int().formats(["csr", "csc"]).to(device)
        x, y = feats.to(device), labels.to(device)
        train_idx = torch.nonzero(train_mask).flatten().to(device)
        total_train = float(train_idx.numel())
    else:
        if rank == 0:
            assign, part_stats = mdist.cached_partition(src, dst, n, world)
        else:
            assign = torch.empty(n, dtype=torch.int64, device=device)
        mdist.broadcast(assign, 0)
        block, hplan, own = mdist.build_local_partition(src, dst, n, assign, rank, world)
        g = mdist.DistGraph(block, hplan)
        own_cpu = own.cpu()
        x, y = feats[own_cpu].to(device), labels[own_cpu].to(device)
        train_idx = torch.nonzero(train_mask[own_cpu]).flatten().to(device)
        total_train = float(train_mask.sum())
        mdist.broadcast_parameters(model)
        part_stats.update({"halo_rows": hplan.n_halo, "owned_rows": hplan.n_own, "send_rows": int(sum(hplan.send_splits)),
                           "local_edges": block.number_of_edges(),
                           "halo_bytes_per_exchange": {"D=%d" % d: hplan.n_halo * d * 4 for d in (spec["feat"], cfg["hidden"])},
                           "exchanges_per_epoch": "2 forward (layer-1 input halo is resident) + 2 backward"})
    del src, dst
    opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"])

    def step():
        # main_dgl_product_sage.py:101-110; for P > 1 the mean loss is taken over the GLOBAL train set
        model.train()
        opt.zero_grad()
        out = model(g, x)[train_idx]
        loss = F.nll_loss(out, y[train_idx], reduction="sum") / total_train
        loss.backward()
        if world > 1:
            mdist.allreduce_gradients(model)
        opt.step()
        return loss.item()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    sparse.PROFILE = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    records, sparse.PROFILE = sparse.PROFILE, None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        mdist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        lt = torch.tensor([loss], dtype=torch.float64, device=device)  # each rank holds its share of the global mean
        mdist.all_reduce(lt, op=dist.ReduceOp.SUM)
        loss = float(lt.item())

    # ---- roofline of the dominant kernel: copy_u g-SpMM at D = hidden (4 of the 5 launches per epoch)
    D = cfg["hidden"]
    sel = [r for r in records if r["op"] == "copy_lhs" and r["out_len"] == D]
    durs = [r["start"].elapsed_time(r["end"]) * 1e-3 for r in sel]
    roofline = None
    if durs:
        r0 = sel[0]
        avg = sum(durs) / len(durs)
        algo = sum(spmm_algorithmic_bytes(r["n_rows"], r["n_cols"], r["nnz"], D) for r in sel) / len(sel)
        achieved = algo / avg / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1 and args.scale == 1.0:
            tj = json.load(open(tpath))
            if tj.get("dataset") == args.dataset and tj.get("D") == D:
                traffic = tj.get("hbm_bytes_per_launch")
        roofline = {"bound": "hbm", "kernel": "mgx::spmm_rowwave32_kernel<4,16,copy_lhs> (copy_u/sum, D=%d)" % D,
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(algo),
                    "no_reuse_gather_bytes_per_launch": int(algo - 4 * r0["n_cols"] * D + 4 * r0["nnz"] * D),
                    "avg_launch_ms": round(avg * 1e3, 4),
                    "launches_timed": len(durs), "rows": r0["n_rows"], "nnz": r0["nnz"]}

    epoch = elapsed / args.steps
    agg_edges = full_graph.spmm_edges_per_epoch(cfg["num_layers"], num_edges)
    line = {
        "metric": "aggregated edges/s, full-graph 3-layer GraphSAGE (hidden 64) on an ogbn-products-shaped graph; "
                  "epoch time in ms_per_step",
        "value": agg_edges / epoch, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": epoch * 1e3, "epoch_time_s": epoch, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[3]: 3-layer GraphSAGE full-graph on ogbn-products shape "
                               "(N=%d, E=%d directed, D=%d->%d->%d->%d), %s" %
                               (n, num_edges, spec["feat"], cfg["hidden"], cfg["hidden"], spec["classes"],
                                "1 GPU" if world == 1 else "edge-cut partition over %d GPUs, RCCL all_to_all halo" % world),
                   "spmm_per_epoch": 2 * cfg["num_layers"] - 1, "final_loss": loss,
                   "schedule": os.environ.get("MGX_SCHEDULE", "auto"), "partition": part_stats,
                   "dense_gemm_selection": "tunableop file" if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") == "1" else "default"},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(g, spec["feat"], cfg["hidden"])
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
