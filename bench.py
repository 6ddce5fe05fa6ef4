#!/usr/bin/env python
"""bench.py -- full-graph GraphSAGE on an ogbn-products-shaped graph (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  The driver launches it as `python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N ...`; a plain `python bench.py --gpus N` (no WORLD_SIZE in the environment) starts those N ranks
itself as CHILD processes, before this process has touched the GPU, and exits with their status.

A "step" is one training epoch of main_dgl_product_sage.py's loop (forward, nll_loss on the train split, backward, Adam
step, loss.item() as the host sync; lines 101-110) on the synthetic products-shaped graph (N = 2,449,029,
E = 123,718,280 directed, D = 100 -> 64 -> 64 -> 47, fp32).  `value` = aggregated edges/s = (3 forward + 2 backward
g-SpMMs) * E / epoch time, whole job.  N > 1 partitions the SAME graph (edge-cut + RCCL all_to_all halo exchange):
strong scaling.

The JSON line also carries
  roofline      the dominant hot-path kernel (copy_u/sum g-SpMM at D = 64, 4 launches per epoch): algorithmic bytes
                (4(N+1) + 4E + 4ND + 4ND; the two backward launches accumulate into the self gradient and so
                read the output rows as well: + 4ND for them) / its mean launch duration, measured live with HIP events on
                the launch stream, against the 8 TB/s HBM peak; `kernels` lists every g-SpMM shape of the epoch (D = 100 too);
                `traffic` = FETCH_SIZE*2 + WRITE_SIZE per launch from rocprofv3 --pmc child passes of the same kernel on
                the same graph, run after the timed region (N = 1; --no-pmc skips them -> null);
                `controls` = the same kernel on control graphs of the same size (dgl-0.5-benchmark_amd/kernel_controls.py);
  epoch_ms_last_layer_backward_on_loss_rows   the default model with MGX_SAGE_SPARSE_LAST=1 (exact; not the headline)
  epoch_ms_plain_model   the same epoch with the reference's exact module graph (torch.nn.Linear, F.relu, nn.Dropout,
                separate add, log_softmax over all nodes then [train_idx], F.nll_loss) -- only update_all() is this
                package's -- timed right after the headline loop.  The headline (default) model computes the same function
                with this package's dense-side forms: a SAGE layer as one autograd node and one GEMM on [h | neigh], fused
                relu+dropout, the loss tail on the training rows only;
  value_reference_modules / ms_per_step_reference_modules   the same unmodified modules with the OPT-IN switch a user of an
                unmodified script is told to set (MGX_ACCELERATE_LINEAR=1 == mi355x_graph.utils.accelerate_linear()): torch.nn.Linear's
                backward on tall matrices through this package's column-sum / X^T Y kernels; `..._torch_linear` = nothing patched
                (== epoch_ms_plain_model); `reference_modules` records which state produced which number;
  secondary     the other BASELINE configs timed in this same run (dgl-0.5-benchmark_amd/secondary_bench.py): arxiv SAGE, the 2 x 8-head
                and the script-default 3 x 1-head reddit GATs, molhiv GCN eager + captured -- ms_per_step, steps and the roofline
                entry (SURVEY 8d bytes / live HIP-event mean) of each one's dominant hot-path call;
  cpu_baseline  the CPU oracle (OpenMP port of DGL's CPU algorithm) timed on the host cores on the g-SpMMs of one epoch
                (rank 0, N = 1 only, bounded to ~30 s).
"""
import argparse
import datetime
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

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


# Before torch is imported, for EVERY way this file is started (plain, self-launched children, the driver's
# `python -m torch.distributed.run ... bench.py`): dmabuf IPC -- RCCL and device-tensor sharing across processes fail with
# `hipIpcGetMemHandle: invalid argument` on this driver without it -- and a bounded OpenMP team per rank (torch.distributed.run
# exports OMP_NUM_THREADS=1 itself when it is unset, which this leaves alone).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))))

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "dgl-0.5-benchmark_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# Dense layers of the model are rocBLAS/hipBLASLt GEMMs; selections recorded once on MI355X for exactly these shapes
# are only LOADED here (tuning off) -- a user-level PyTorch setting, see dgl-0.5-benchmark_amd/tunable.py.
import tunable  # noqa: E402

tunable.setup()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def spmm_algorithmic_bytes(n_dst, n_src, nnz, D, accumulate=False):
    """indptr + indices + every source row once + every output row once.  `accumulate`: the extended count for an
    accumulating launch (out += A x, the backward aggregation inside ops.SageMeanLayerFn), which also reads every output row
    once -- reported beside the contract's figure (`frac_incl_accumulate_read`), never under `frac`."""
    return 4 * (n_dst + 1) + 4 * nnz + 4 * n_src * D + 4 * n_dst * D * (2 if accumulate else 1)


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
    # a whole CPU epoch for scale (optional in the contract): the five aggregations above + the dense layers of the same model
    # (fc_self / fc_neigh GEMMs, relu, log_softmax / nll on the train rows, their backward) by PyTorch on the same cores, once
    try:
        if len(done) == len(plan) and spent < budget_s:
            torch.set_num_threads(cores)
            dims = [feat_dim, hidden, hidden]
            outs = [hidden, hidden, 47]
            t0 = time.perf_counter()
            for d_in, d_out in zip(dims, outs):
                xin = torch.from_numpy(feats[d_in]).requires_grad_(d_in != feat_dim)
                nin = torch.from_numpy(feats[d_in])  # stands for mean_agg(x): same shape, values do not matter for the time
                lin_s, lin_n = torch.nn.Linear(d_in, d_out, bias=False), torch.nn.Linear(d_in, d_out)
                y = torch.relu(lin_s(xin) + lin_n(nin))
                y.sum().backward()
            dense_s = time.perf_counter() - t0
            out["epoch"] = {"epoch_s": round(spent + dense_s, 3), "spmm_s": round(spent, 3), "dense_s": round(dense_s, 3),
                            "note": "five oracle aggregations + the three SAGE layers' dense forward / backward by PyTorch CPU, "
                                    "%d threads; no optimizer step" % cores}
    except Exception as err:
        out["epoch"] = {"error": str(err)[:200]}
    # the library's own CPU (OpenMP) variants (include/mi355x_graph_cpu.h, csrc/cpu_ops.cpp: product code, opt-in) on one D = hidden
    # aggregation, same cores -- beside the oracle's number, never instead of it
    try:
        import ctypes
        from mi355x_graph import cpu_backend, sparse as msparse
        cpu_backend.lib().mgx_cpu_set_num_threads(cores)
        ip, ix = host["csc"]
        view = msparse.CsrView(n, n, torch.from_numpy(ip), torch.from_numpy(ix), None)
        xh = torch.from_numpy(feats[hidden])
        be_cpu = cpu_backend.CpuBackend()
        be_cpu.spmm(view, "copy_lhs", "mean", xh[:, :4].contiguous(), None, 4, 0, 4, None, None, None, None, False)  # threads started outside the clock
        t0 = time.perf_counter()
        be_cpu.spmm(view, "copy_lhs", "mean", xh, None, hidden, 0, hidden, None, None, None, None, False)
        dt = time.perf_counter() - t0
        out["product_cpu_variants"] = {"value": nnz / dt, "unit": "edges/s", "kind": "mgx_cpu_spmm_csr (copy_u/mean, D=%d), libmi355x_graph_cpu.so" % hidden,
                                       "cores": cores, "seconds": round(dt, 3)}
    except Exception as err:  # must not lose the bench line
        out["product_cpu_variants"] = {"value": None, "error": str(err)[:200]}
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


def profile_avg_us(tag=None):
    """Average duration (us) of the g-SpMM kernels in the committed rocprofv3 --kernel-trace --stats summary of this same
    command (profiles/<tag>_kernel_stats.txt, newest round by default), keyed by feature width, so that the live HIP-event
    mean and the traced mean sit side by side in the line.  {} when no summary travels with the tree."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_stats.txt")))
    if tag:
        files = [f for f in files if os.path.basename(f).startswith(tag + "_")]
    if not files:
        return {}
    out, group = {}, {}
    for ln in open(files[-1]):
        # the row kernels only: spmm_tile*_kernel<W, NL, ...> carries other template arguments.  A launch over a two-part
        # plan (short items: spmm_rowgroup32_kernel<VEC, G, ...>; the rest: spmm_rowwave32_kernel<VEC, G, ...>) is the SUM of the two.
        m = re.match(r"void mgx::spmm_(rowwave32|rowgroup32)_kernel<(\d+), (\d+),.*?\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s*$", ln)
        if m:
            fam, vec, g, calls, total_ms = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5))
            for width in (64, 100):  # lane-group width G covers D = 4 G columns (D = 100: G = 32 with idle lanes)
                if vec * g >= width > vec * g // 2:
                    if fam == "rowwave32" and width not in out:
                        out[width] = {"avg_us": float(m.group(6)), "calls": calls, "file": os.path.basename(files[-1])}
                    elif fam == "rowgroup32" and width not in group:
                        group[width] = (calls, total_ms)
    for width, (calls, total_ms) in group.items():
        if width in out and calls == out[width]["calls"]:  # one lane-group launch per wave-per-item launch: the same calls, in two parts
            out[width]["avg_us"] = round(out[width]["avg_us"] + total_ms * 1e3 / calls, 2)
            out[width]["kernels"] = "spmm_rowgroup32_kernel + spmm_rowwave32_kernel (two-part plan)"
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as child processes (torch.distributed.run) and exit
    with their status.  Runs before anything in this process has initialised the GPU (device_count() does not)."""
    share = os.environ.get("MGX_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not share:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible (MGX_BENCH_SHARE_GPU=1 MGX_DIST_BACKEND=gloo runs "
                         "all ranks on cuda:0 for a smoke test)\n" % (args.gpus, have))
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    sys.exit(subprocess.call(cmd, env=env))


def pmc_traffic(widths, reps, scale, timeout_s=240, slots=False):
    """HBM-side bytes per launch of the g-SpMM kernels on the benchmark graph from rocprofv3 counter passes run as CHILD
    processes (rocprofv3 -- python3 kernel_controls.py ...; --pmc alone, one pass per counter group as gfx950's TCC slots
    require).  Returns (list per width | None, note); slots=True: two more entries -- the pack pass and the slot-form D = 64 launches."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    import kernel_controls as kc
    sets, t0 = [], time.time()
    work = tempfile.mkdtemp(prefix="mgx_pmc_")
    try:
        for i, counters in enumerate((["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"])):
            left = timeout_s - (time.time() - t0)
            if left < 20:
                return None, "pmc passes ran out of time"
            out = os.path.join(work, "p%d" % i)
            cmd = [rocprof, "--pmc"] + counters + ["--output-format", "csv", "-d", out, "--", sys.executable,
                                                     os.path.join(PKG, "kernel_controls.py"), "--graphs", "products",
                                                     "--widths", ",".join(str(w) for w in widths), "--reps", str(reps),
                                                     "--scale", str(scale)] + (["--slots"] if slots else [])
            env = dict(os.environ, TMPDIR=work)
            r = subprocess.run(cmd, cwd=work, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=left)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (" ".join(counters), r.returncode,
                                                                     r.stderr.decode(errors="replace")[-200:])
            sets.append(kc.parse_pmc_dir(out))
        return kc.pmc_to_traffic(sets, reps), "rocprofv3 --pmc child passes (FETCH_SIZE x2 + WRITE_SIZE; %d launches each)" % reps
    except (subprocess.TimeoutExpired, OSError) as err:
        return None, "pmc passes: %s" % (str(err)[:160],)
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--dataset", default="products")
    p.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debug only; makes the line invalid)")
    p.add_argument("--mixing", type=float, default=None,
                   help="share of edges that leave their planted community (the benchmark graph: 0.25).  A CONTROL for the scaling "
                        "model -- any other value makes the line a control, not the benchmark (config.workload says so)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic = null)")
    p.add_argument("--no-controls", action="store_true", help="skip the control graphs (roofline.controls)")
    p.add_argument("--no-plain", action="store_true", help="skip the plain-torch-module epoch (epoch_ms_plain_model)")
    p.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configs (the `secondary` block: arxiv SAGE, reddit GATs, molhiv)")
    p.add_argument("--no-variants", action="store_true", help="skip the exact variants reported beside the headline (last layer on the loss rows, layer 1 projected first)")
    p.add_argument("--dropout", type=float, default=None, help="override the model's dropout (tests compare N = 1 and N > 1 at 0)")
    p.add_argument("--emulate-ranks", default=None, metavar="P[,P...]",
                   help="N = 1 only: run every rank of a P-way partition of the same graph one at a time on this GPU (mi355x_graph/"
                        "emulate.py) and put the scaling model into config.partition.predicted; given explicitly it also skips the "
                        "plain-model / controls / counter / CPU legs.  Default for a plain N = 1 run: 2,4,8")
    p.add_argument("--no-scale-model", action="store_true", help="skip the emulated-ranks scaling model")
    p.add_argument("--report", default=None, help="with --emulate-ranks: also write the human-readable table to this file")
    p.add_argument("--tune-dense", action="store_true",
                   help="experiment, with --emulate-ranks: PyTorch TunableOp chooses the dense GEMMs of the per-rank shapes in the warm-up")
    args = p.parse_args()
    if args.emulate_ranks is not None:
        args.no_plain = args.no_controls = args.no_pmc = args.no_cpu_baseline = args.no_secondary = True
    elif not args.no_scale_model and args.gpus == 1:
        args.emulate_ranks = "2,4,8"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return

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
        # generous collective timeout: rank 0 partitions while the others wait in a broadcast
        tmo = datetime.timedelta(minutes=60)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
        world = dist.get_world_size()  # what the backend really connected
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but the process group has %d rank(s); reporting n_gpus=%d" % (args.gpus, world, world),
              file=sys.stderr)

    t_start = time.perf_counter()

    def progress(msg):  # rank 0, stderr: where a multi-rank run spends its set-up time (never on stdout: that is the JSON line)
        if rank == 0 and world > 1:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    import dgl
    import full_graph
    from mi355x_graph import dist as mdist, ops, sparse
    from mi355x_graph.datasets import SHAPES, synthetic_edges

    cfg = dict(full_graph.SAGE_CONFIGS[args.dataset])
    if args.dropout is not None:
        cfg["dropout"] = args.dropout
    spec = SHAPES[cfg["dataset"]]
    n = max(64, int(spec["n"] * args.scale))
    m = max(64, int(spec["m"] * args.scale))
    gen_kw = {} if args.mixing is None else {"mixing": args.mixing}
    src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], device, symmetric=spec["symmetric"], **gen_kw)
    num_edges = int(src.shape[0])
    torch.cuda.synchronize()
    progress("graph generated on the device (%d edges)" % num_edges)
    gen = torch.Generator(device="cpu").manual_seed(spec["seed"] + 100)
    feats = torch.rand(n, spec["feat"], generator=gen)
    labels = torch.randint(0, spec["classes"], (n,), generator=gen)
    train_mask = torch.rand(n, generator=gen) < 0.08  # ogbn-products trains on 8 % of the nodes

    def make_model(plain=False):
        torch.manual_seed(1234)
        return full_graph.GraphSAGE(spec["feat"], cfg["hidden"], spec["classes"], cfg["num_layers"], cfg["dropout"],
                                    cfg["batch_norm"], cfg["neigh_bias"], plain=plain).to(device)

    progress("host features / labels drawn")
    model = make_model()
    part_stats = {}
    bucket = None
    if world == 1:
        g = dgl.graph((src, dst), num_nodes=n).int().formats(["csr", "csc"]).to(device)
        x, y = feats.to(device), labels.to(device)
        train_idx = torch.nonzero(train_mask).flatten().to(device)
        total_train = float(train_idx.numel())
    else:
        # every rank generated the edge list itself (same seed): make sure they agree before rank 0's partition is applied
        finger = mdist.edge_fingerprint(src, dst)
        fp = torch.tensor([finger, -finger], dtype=torch.int64, device=device)
        mdist.all_reduce(fp, op=dist.ReduceOp.MAX)
        if int(fp[0]) != finger or int(fp[1]) != -finger:
            raise SystemExit("bench.py: ranks generated different graphs (edge checksums differ)")
        t0 = time.perf_counter()
        if rank == 0:
            assign, part_stats = mdist.cached_partition(src, dst, n, world)
        else:
            assign = torch.empty(n, dtype=torch.int64, device=device)
        part_stats["partition_s_rank0"] = round(time.perf_counter() - t0, 2)
        progress("partitioned")
        mdist.broadcast(assign, 0)
        progress("assignment broadcast")
        t0 = time.perf_counter()
        block, hplan, own = mdist.build_local_partition(src, dst, n, assign, rank, world)
        g = mdist.DistGraph(block, hplan)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        progress("local partition built")
        own_cpu = own.cpu()
        x, y = feats[own_cpu].to(device), labels[own_cpu].to(device)
        g.set_static_input(x)  # the layer-1 input is constant: its halo rows are exchanged once and stay resident
        train_idx = torch.nonzero(train_mask[own_cpu]).flatten().to(device)
        total_train = float(train_mask.sum())
        t2 = time.perf_counter()
        if cfg["batch_norm"]:  # statistics over the union of every rank's rows (main_dgl_arxiv_sage.py:70-77 on a partition)
            model = mdist.convert_batchnorm(model)
        mdist.broadcast_parameters(model)
        bucket = mdist.GradBucket(model)
        torch.cuda.synchronize()
        part_stats["build_local_s_rank0"] = round(t1 - t0, 2)
        part_stats["slice_features_s_rank0"] = round(t2 - t1, 2)
        part_stats["broadcast_parameters_s_rank0"] = round(time.perf_counter() - t2, 2)
        progress("features sliced, parameters broadcast")
        # per-rank halo statistics, gathered on every rank (tiny)
        mine = torch.tensor([hplan.n_own, hplan.n_halo, int(sum(hplan.send_splits)), block.number_of_edges(),
                             int(hplan.halo.num_edges())], dtype=torch.int64, device=device)
        allr = torch.zeros(world * 5, dtype=torch.int64, device=device)
        allr[rank * 5:(rank + 1) * 5] = mine
        mdist.all_reduce(allr)
        allr = allr.view(world, 5).cpu().tolist()
        D_hid = cfg["hidden"]
        part_stats.update({
            "edge_cut_pct": round(100.0 * sum(r[4] for r in allr) / max(num_edges, 1), 2),
            "per_rank": [{"owned_rows": r[0], "halo_rows": r[1], "send_rows": r[2], "local_edges": r[3], "halo_edges": r[4],
                          "halo_bytes_per_exchange_D%d" % D_hid: r[1] * D_hid * 4} for r in allr],
            "max_halo_bytes_per_exchange": max(r[1] for r in allr) * D_hid * 4,
            "exchanges_per_epoch": "2 forward (the layer-1 input halo is resident) + 2 backward, D=%d each; dense-row bytes above -- with "
                                   "MGX_SPARSE_HALO (default) the relu + dropout rows cross as bitmaps + non-zeros and their gradients under "
                                   "the same bitmaps (dist.SparseHalo): bytes_received_per_epoch below is what really moved" % D_hid,
            "sparse_halo": mdist.SPARSE_HALO})
    if world > 1 or not args.emulate_ranks:
        del src, dst
    opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"])

    def make_step(model, opt, bucket):
        model.rows_are_distinct = True  # train_idx = nonzero() of a mask

        def step():
            # main_dgl_product_sage.py:101-110; for P > 1 the mean loss is taken over the GLOBAL train set
            model.train()
            if bucket is not None:
                bucket.zero()  # gradients are views into one flat buffer (zero_grad would drop them)
            else:
                opt.zero_grad()
            y_train = y[train_idx]  # main_dgl_product_sage.py:106 gathers the labels inside train()
            # default model: log_softmax on the train rows only (row-wise, so the same numbers); plain: the reference's line
            if model.plain:
                loss = F.nll_loss(model(g, x)[train_idx], y_train, reduction="sum") / total_train
            else:
                loss = ops.nll_sum(model(g, x, rows=train_idx), y_train) / total_train
            loss.backward()
            if bucket is not None:
                bucket.all_reduce()
            opt.step()
            return loss.item()
        return step

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup, profile):
        import gc
        gc.collect()  # before the warm-up: what it releases is back in the allocator's pool by the time the clock starts
        for _ in range(warmup):
            step()
        fence()
        if profile:
            sparse.PROFILE = []
            if world > 1:
                g._comm.trace = []  # exposed exchange time per rank (dist._TimedWork): what the scaling model predicts, measured
        # as timeit does: no cyclic garbage collection inside the timed loop (a full collection of this process' heap is an 80 ms host
        # stall that landed in the last step of one leg, run after run: 17.5 ms steps, one of 97 ms)
        gc_was = gc.isenabled()
        gc.disable()
        t0 = time.perf_counter()
        marks = []
        try:
            for _ in range(steps):
                loss = step()
                marks.append(time.perf_counter())  # (step() ends with loss.item(): the device has finished the step)
            fence()
            elapsed = time.perf_counter() - t0
        finally:
            if gc_was:
                gc.enable()
        if os.environ.get("MGX_BENCH_STEP_TIMES") == "1" and rank == 0:
            print("[bench] step times, ms: " + " ".join("%.2f" % ((b - a) * 1e3) for a, b in zip([t0] + marks[:-1], marks)),
                  file=sys.stderr, flush=True)
        records = None
        if profile:
            records, sparse.PROFILE = sparse.PROFILE, None
        return elapsed, loss, records

    progress("starting warm-up + timed steps")
    elapsed, loss, records = timed(make_step(model, opt, bucket), args.steps, args.warmup, True)
    progress("timed region done")
    if world > 1:
        tr = g._comm.trace or []  # the timed steps' exchanges (set in timed(), after the warm-up)
        g._comm.trace = None
        exposed = sum(a.elapsed_time(b) for a, b, _ in tr) / max(args.steps, 1)
        ex = torch.zeros(world * 2, dtype=torch.float64, device=device)
        ex[rank * 2], ex[rank * 2 + 1] = exposed, sum(nb for _, _, nb in tr) / max(args.steps, 1)
        mdist.all_reduce(ex)
        ex = ex.view(world, 2).cpu().tolist()
        for r, rec in enumerate(part_stats.get("per_rank", [])):
            rec["exposed_exchange_ms_per_epoch"] = round(ex[r][0], 4)
            rec["bytes_received_per_epoch"] = int(ex[r][1])
        part_stats["exposed_exchange_note"] = ("per rank and epoch: compute-stream time between two HIP events around every all_to_all wait() "
                                               "in the timed region (0 over gloo staging, which exchanges synchronously)")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        mdist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        lt = torch.tensor([loss], dtype=torch.float64, device=device)  # each rank holds its share of the global mean
        mdist.all_reduce(lt, op=dist.ReduceOp.SUM)
        loss = float(lt.item())

    # ---- roofline: every copy_u g-SpMM shape of the epoch from the live HIP-event records; the headline entry is the
    # dominant one, D = hidden (4 of the 5 launches per epoch)
    D = cfg["hidden"]
    kernels = []
    sparse_recs = [r for r in records if r.get("variant") == "row-sparse"]
    records = [r for r in records if r.get("variant") != "row-sparse"]
    traced = profile_avg_us()
    for width in sorted({r["out_len"] for r in records if r["op"] == "copy_lhs"}):
        sel = [r for r in records if r["op"] == "copy_lhs" and r["out_len"] == width]
        durs = [r["start"].elapsed_time(r["end"]) * 1e-3 for r in sel]
        r0 = sel[0]
        avg = sum(durs) / len(durs)
        # the contract's figure for EVERY launch; the accumulating launches' extra read of the output rows is
        # real traffic of the fused layer and is reported beside it, under its own keys
        algo = spmm_algorithmic_bytes(r0["n_rows"], r0["n_cols"], r0["nnz"], width)
        algo_rmw = sum(spmm_algorithmic_bytes(r["n_rows"], r["n_cols"], r["nnz"], width, r.get("accumulate", False)) for r in sel) / len(sel)
        achieved = algo / avg / 1e9
        variants = sorted({r.get("variant", "row") for r in sel})
        entry = "mgx_spmm_tile_copy_u" if variants == ["tile"] else "mgx_spmm_csr / mgx_spmm_copy_u_strided"
        # the launches of this width by FORM: dense 256-byte rows, or 128-byte slots of a mostly-zero operand (mgx_spmm_copy_u_slots:
        # the forward aggregations of relu + dropout outputs and the reversed aggregation of the gradient behind one)
        forms = []
        for name, pick in (("dense rows", lambda r: "slots" not in r.get("variant", "")), ("128-byte slots (mgx_spmm_copy_u_slots)", lambda r: "slots" in r.get("variant", ""))):
            fd = [r["start"].elapsed_time(r["end"]) * 1e-3 for r in sel if pick(r)]
            if fd and len(fd) != len(durs):
                favg = sum(fd) / len(fd)
                forms.append({"form": name, "launches_per_epoch": len(fd) // max(args.steps, 1), "avg_launch_ms": round(favg * 1e3, 4),
                              "frac": round(algo / favg / 1e9 / HBM_PEAK_GBPS, 4)})
        if any("slots" in v for v in variants):
            entry += " / mgx_spmm_copy_u_slots"
        kernels.append({"bound": "hbm", "kernel": "g-SpMM copy_u/sum|mean, D=%d (%s; %s kernel)" % (width, entry, "+".join(variants)),
                        "D": width,
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                        "algorithmic_bytes_per_launch": int(algo),
                        "frac_incl_accumulate_read": round(algo_rmw / avg / 1e9 / HBM_PEAK_GBPS, 4),
                        "bytes_incl_accumulate_read_per_launch": int(algo_rmw),
                        "no_reuse_gather_bytes_per_launch": int(algo - 4 * r0["n_cols"] * width + 4 * r0["nnz"] * width),
                        "avg_launch_ms": round(avg * 1e3, 4), "launches_timed": len(durs),
                        "profile_avg_us": traced.get(width),
                        "launches_per_epoch": len(durs) // max(args.steps, 1),
                        "accumulating_launches_per_epoch": sum(1 for r in sel if r.get("accumulate")) // max(args.steps, 1),
                        "forms": forms or None,
                        "rows": r0["n_rows"], "nnz": r0["nnz"]})

    # backward aggregations of a GRADIENT go through the row-sparse kernel (zero rows skipped): listed apart, never mixed into
    # the dense launches' average -- the compulsory-bytes formula assumes every source row is read
    row_sparse = None
    if sparse_recs:
        durs = sorted(r["start"].elapsed_time(r["end"]) for r in sparse_recs)
        per_epoch = len(durs) // max(args.steps, 1)
        row_sparse = {"kernel": "g-SpMM copy_u/sum of a gradient, zero rows skipped (mgx_row_nonzero_bits + mgx_spmm_copy_u_masked)",
                      "launches_per_epoch": per_epoch, "ms_fastest": round(durs[0], 4), "ms_slowest": round(durs[-1], 4),
                      "ms_mean": round(sum(durs) / len(durs), 4),
                      "note": "products trains on 8 % of the nodes: the last layer's gradient has 92 % zero rows (fast launches); "
                              "earlier layers' gradients are dense (slow launches = the dense kernel + the row flagging pass)"}

    epoch = elapsed / args.steps
    agg_edges = full_graph.spmm_edges_per_epoch(cfg["num_layers"], num_edges)
    # like for like with cpu_baseline (which times the five aggregations alone): the same five aggregations' DEVICE time, hub
    # fix-up launches included (the events bracket the whole mgx_spmm_* call), per epoch
    spmm_ms_per_epoch = sum(r["start"].elapsed_time(r["end"]) for r in records + sparse_recs) / max(args.steps, 1)
    line = {
        "metric": "aggregated edges/s, full-graph 3-layer GraphSAGE (hidden 64) on an ogbn-products-shaped graph; "
                  "epoch time in ms_per_step",
        "value": agg_edges / epoch, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": epoch * 1e3, "epoch_time_s": epoch, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "value_spmm_only_edges_per_s": (agg_edges / (spmm_ms_per_epoch * 1e-3)) if spmm_ms_per_epoch > 0 and world == 1 else None,
        "spmm_ms_per_epoch": round(spmm_ms_per_epoch, 4) if world == 1 else None,
        "config": {"workload": "%s %d-layer GraphSAGE full-graph on %s shape "
                               "(N=%d, E=%d directed, D=%s), %s" %
                               ("configs[3]:" if args.dataset == "products" else "(not the headline workload)", cfg["num_layers"],
                                "ogbn-products" if args.dataset == "products" else "the " + args.dataset,
                                n, num_edges, "->".join(str(v) for v in [spec["feat"]] + [cfg["hidden"]] * (cfg["num_layers"] - 1) + [spec["classes"]]),
                                "1 GPU" if world == 1 else "edge-cut partition over %d GPUs, RCCL all_to_all halo" % world)
                               + ("" if args.mixing is None else " -- CONTROL GRAPH: %g of the edges leave their community (benchmark: 0.25)" % args.mixing),
                   "spmm_per_epoch": 2 * cfg["num_layers"] - 1, "final_loss": loss,
                   "schedule": os.environ.get("MGX_SCHEDULE", "auto"), "partition": part_stats,
                   "module_graph": "full_graph.GraphSAGE (default form): a SAGE layer as ONE GEMM on [h | mean_agg(h)], the same form on "
                                   "one GPU and on every rank of a partition (dist.DistSageMeanCatFn: halo exchange inside the layer)",
                   "value_is": "`value` / `ms_per_step`: this package's own module graph (module_graph above) -- same function, loss and gradients as "
                               "the reference's modules (tests/test_next_rows_gpu.py), all five aggregations over every edge.  "
                               "`value_reference_modules` / `ms_per_step_reference_modules`: the reference's UNMODIFIED module graph "
                               "(main_dgl_product_sage.py:15-99: torch.nn.Linear x 2 + add, F.relu, nn.Dropout, log_softmax over all nodes) with only "
                               "update_all() from this package; see `reference_modules` for the torch.nn.functional.linear state of each number",
                   "dropout": cfg["dropout"],
                   "dense_gemm_selection": tunable.status()},
    }

    single = rank == 0 and world == 1
    # ---- the reference's exact module graph on the same graph, same step function
    if single and not args.no_plain:
        del model, opt
        torch.cuda.empty_cache()
        from mi355x_graph import utils as mutils
        was_accelerated = mutils.accelerate_linear(False)  # this leg: PyTorch's own torch.nn.Linear backward (MGX_ACCELERATE_LINEAR=0)
        pm = make_model(plain=True)
        popt = torch.optim.Adam(pm.parameters(), lr=cfg["lr"])
        psteps = min(args.steps, 10)
        pel, ploss, _ = timed(make_step(pm, popt, None), psteps, min(args.warmup, 3), False)
        line["epoch_ms_plain_model"] = round(pel / psteps * 1e3, 3)
        line["ms_per_step_reference_modules_torch_linear"] = round(pel / psteps * 1e3, 3)
        line["value_reference_modules_torch_linear"] = agg_edges / (pel / psteps)
        line["plain_model"] = {"modules": "torch.nn.Linear, F.relu, nn.Dropout, fc_self(h) + fc_neigh(neigh) as two GEMMs and an "
                                          "add (main_dgl_product_sage.py:15-99); update_all() by this package",
                               "steps": psteps, "final_loss": ploss,
                               "value_edges_per_s": agg_edges / (pel / psteps)}
        del pm, popt
        torch.cuda.empty_cache()
        # ---- the same unmodified modules with mi355x_graph.utils.accelerate_linear(): torch.nn.Linear's backward through this package's
        # column-sum / X^T Y kernels (PyTorch's bias-gradient reduction alone is 19 ms of the epoch above)
        mutils.accelerate_linear(True)
        try:
            pm = make_model(plain=True)
            popt = torch.optim.Adam(pm.parameters(), lr=cfg["lr"])
            pel2, ploss2, _ = timed(make_step(pm, popt, None), psteps, min(args.warmup, 3), False)
            line["epoch_ms_plain_model_accelerated_linear"] = round(pel2 / psteps * 1e3, 3)
            line["plain_model"]["accelerated_linear"] = {"what": "OPT-IN since round 5 (MGX_ACCELERATE_LINEAR=1 or mi355x_graph.utils.accelerate_linear(); "
                                                                 "`import dgl` alone leaves torch alone): same modules, same script; F.linear's "
                                                                 "backward on tall matrices by mgx_column_sum / mgx_xty",
                                                         "final_loss": ploss2}
            line["ms_per_step_reference_modules"] = round(pel2 / psteps * 1e3, 3)
            line["value_reference_modules"] = agg_edges / (pel2 / psteps)
            line["reference_modules"] = {
                "module_graph": "full_graph.GraphSAGE(plain=True) == main_dgl_product_sage.py:15-99 module for module; the step is main_dgl_product_sage.py:101-110",
                "ms_per_step_reference_modules": {"linear": "MGX_ACCELERATE_LINEAR=1 (opt-in; what INTEGRATION.md tells the user of an unmodified script to "
                                                            "set): torch.nn.functional.linear's BACKWARD on >= 65536-row fp32 matrices by mgx_xty / "
                                                            "mgx_column_sum, forward untouched", "final_loss": ploss2, "steps": psteps},
                "ms_per_step_reference_modules_torch_linear": {"linear": "nothing patched: `import dgl` as it is by default; PyTorch's own Linear backward "
                                                                         "(19 ms of it one bias-gradient reduce_kernel, profiles/r03_plain_epoch_timeline.txt)",
                                                               "final_loss": ploss, "steps": psteps},
                "accelerate_linear_during_headline_loop": bool(was_accelerated)}
            del pm, popt
        finally:
            mutils.accelerate_linear(was_accelerated)
        torch.cuda.empty_cache()
        # ---- the default model again with the backward of the LAST layer formed on the loss rows only (ops.SageMeanCatRowsFn,
        # MGX_SAGE_SPARSE_LAST=1): same forward, same gradients -- the output gradient is zero outside the 8 % training rows, so the
        # dense gradients are taken on those rows and the reversed aggregation skips the rows that are zero by construction.  Off in
        # the headline (value / ms_per_step) so that every one of its five aggregations gathers every source row; reported beside it.
        if os.environ.get("MGX_SAGE_SPARSE_LAST", "0") != "1" and not args.no_variants:
            os.environ["MGX_SAGE_SPARSE_LAST"] = "1"
            try:
                sm = make_model()
                sopt = torch.optim.Adam(sm.parameters(), lr=cfg["lr"])
                sel_, sloss, _ = timed(make_step(sm, sopt, None), psteps, min(args.warmup, 3), False)
                line["epoch_ms_last_layer_backward_on_loss_rows"] = round(sel_ / psteps * 1e3, 3)
                line["last_layer_backward_on_loss_rows"] = {"switch": "MGX_SAGE_SPARSE_LAST=1", "steps": psteps, "final_loss": sloss,
                                                            "value_edges_per_s": agg_edges / (sel_ / psteps)}
                del sm, sopt
            finally:
                os.environ["MGX_SAGE_SPARSE_LAST"] = "0"
            torch.cuda.empty_cache()
        # ---- the default model with layer 1 projected BEFORE its aggregation (ops.SageMeanStaticInputProjectFn,
        # MGX_SAGE_L1_PROJECT_FIRST=1): mean_agg(x) W^T = mean_agg(x W^T), so the layer-1 aggregation runs at 64 columns instead of
        # 100; its weight gradient is taken against the constant mean_agg(x), aggregated once and kept.  Same loss and gradients
        # (test); still five aggregations over every edge per epoch.  NOT the headline: the headline keeps the reference's widths.
        if os.environ.get("MGX_SAGE_L1_PROJECT_FIRST", "0") != "1" and not args.no_variants:
            os.environ["MGX_SAGE_L1_PROJECT_FIRST"] = "1"
            try:
                lm = make_model()
                lopt = torch.optim.Adam(lm.parameters(), lr=cfg["lr"])
                built0 = ops.STATIC_AGGREGATIONS_BUILT[0]
                lel, lloss, _ = timed(make_step(lm, lopt, None), psteps, min(args.warmup, 3), False)
                line["epoch_ms_layer1_projected_first"] = round(lel / psteps * 1e3, 3)
                line["layer1_projected_first"] = {"switch": "MGX_SAGE_L1_PROJECT_FIRST=1", "steps": psteps, "final_loss": lloss,
                                                  "constant_input_aggregated": "%d time(s) in %d steps" % (
                                                      ops.STATIC_AGGREGATIONS_BUILT[0] - built0, psteps + min(args.warmup, 3)),
                                                  "value_edges_per_s": agg_edges / (lel / psteps),
                                                  "aggregation_widths": [cfg["hidden"]] * (2 * cfg["num_layers"] - 1)}
                del lm, lopt
            finally:
                os.environ["MGX_SAGE_L1_PROJECT_FIRST"] = "0"
            torch.cuda.empty_cache()
            # both exact variants together (still not the headline)
            if os.environ.get("MGX_SAGE_SPARSE_LAST", "0") != "1":
                os.environ["MGX_SAGE_L1_PROJECT_FIRST"] = os.environ["MGX_SAGE_SPARSE_LAST"] = "1"
                try:
                    bm = make_model()
                    bopt = torch.optim.Adam(bm.parameters(), lr=cfg["lr"])
                    bel, bloss, _ = timed(make_step(bm, bopt, None), psteps, min(args.warmup, 3), False)
                    line["epoch_ms_both_exact_variants"] = round(bel / psteps * 1e3, 3)
                    del bm, bopt
                finally:
                    os.environ["MGX_SAGE_L1_PROJECT_FIRST"] = os.environ["MGX_SAGE_SPARSE_LAST"] = "0"
                torch.cuda.empty_cache()

    # ---- the other BASELINE configurations, driver-timed in this same run: arxiv SAGE, the 8-head and the
    # script-default reddit GATs, molhiv GCN eager + captured -- each with ms_per_step and the roofline entry of its dominant call
    if single and args.dataset == "products" and not args.no_secondary and args.scale == 1.0:
        import secondary_bench
        model = opt = None
        torch.cuda.empty_cache()
        line["secondary"] = secondary_bench.run(device, progress=lambda msg: print("[bench] " + msg, file=sys.stderr, flush=True))

    # ---- the 2 / 4 / 8-GPU curve as a MODEL measured on this one GPU: every rank of the partitioned program run here, one at a
    # time, exchanges priced per xGMI link (scale_model.py; never a measured multi-GPU number, and labelled so)
    if rank == 0 and world == 1 and args.emulate_ranks:
        import scale_model
        model = opt = None  # (the plain-model leg above may already have dropped them)
        torch.cuda.empty_cache()
        models = []
        for P in [int(t) for t in args.emulate_ranks.split(",") if t.strip()]:
            try:
                models.append(scale_model.run(device, src, dst, n, feats, labels, train_mask, cfg, spec, P,
                                              steps=min(args.steps, 5), warmup=min(args.warmup, 2), dropout=args.dropout,
                                              tune_dense=args.tune_dense,
                                              progress=lambda msg: print("[scale model] " + msg, file=sys.stderr, flush=True)))
            except Exception as err:  # the model must not lose the bench line
                models.append({"ranks": P, "error": "%s: %s" % (type(err).__name__, str(err)[:300])})
            torch.cuda.empty_cache()
        pred = {"kind": "MODEL, not a multi-GPU measurement: every rank's device time measured on this GPU (mi355x_graph/emulate.py), "
                        "exchanges priced per xGMI link (scale_model.py)",
                "reading": "per P and link rate: epoch_ms_overlapped = every stretch between two collectives timed by HIP events, one rank at a "
                           "time (the host never runs ahead across a collective: conservative, sensitive to the host's launch speed); "
                           "epoch_ms_solo_plus_exposed = the slowest rank's epochs run back to back alone with recorded payloads "
                           "(solo_epoch_ms: what one rank's process does per epoch when the exchange is free) + that rank's exposed exchange",
                "epoch_ms_1gpu_measured": round(epoch * 1e3, 3)}
        for m in models:
            if "error" in m:
                pred["P=%d" % m["ranks"]] = m
                continue
            keep = {k: m[k] for k in m if k in ("edge_cut_pct", "exchanges_per_epoch", "messages_per_epoch", "halo_bytes_per_epoch", "imbalance",
                                      "compute_ms_max", "compute_ms_mean", "solo_epoch_ms_max", "solo_epoch_ms_mean", "max_pair_bytes_per_exchange", "max_recv_bytes_per_exchange",
                                      "final_loss", "predicted")}
            keep["per_rank"] = [{k: p[k] for k in ("owned_rows", "halo_rows", "local_edges", "halo_edges", "compute_ms", "solo_epoch_ms",
                                                   "overlap_window_ms_per_exchange")} for p in m["per_rank"]]
            pred["P=%d" % m["ranks"]] = keep
        line["config"]["partition"] = {"predicted": pred}
        if args.report:
            good = [m for m in models if "error" not in m]
            with open(args.report, "w") as fh:
                fh.write(scale_model.report(good, epoch * 1e3, header="Scaling model of %s\n(bench.py --emulate-ranks %s --steps %d)"
                                            % (line["config"]["workload"], args.emulate_ranks, args.steps)))
        del src, dst

    head = next((k for k in kernels if k["D"] == D), kernels[0] if kernels else None)
    roofline = dict(head) if head else None
    if roofline is not None:
        roofline["kernels"] = kernels
        roofline["row_sparse_backward"] = row_sparse
        if single and args.dataset == "products" and not args.no_controls:
            import kernel_controls as kc
            try:
                roofline["controls"] = kc.run_controls(device, [k for k in kc.CONTROLS if k != "products"], (D,), args.scale, reps=5)
                roofline["controls"]["note"] = ("same kernel, same N and E, other edge structure (kernel_controls.py); counter "
                                                "traffic per control: profiles/r03_controls.txt")
            except Exception as err:  # a control graph must not lose the bench line
                roofline["controls"] = {"error": str(err)[:200]}
        if single and args.dataset == "products" and not args.no_pmc:
            widths = [k["D"] for k in kernels]
            slot_forms = [f for k in kernels for f in (k.get("forms") or []) if "slots" in f["form"]]
            traffic, note = pmc_traffic(widths, 3, args.scale, slots=bool(slot_forms))
            roofline["traffic_source"] = note
            if traffic:
                for k, t in zip(kernels, traffic):
                    if "hbm_read_bytes" in t and "hbm_write_bytes" in t:
                        k["traffic"] = t["hbm_read_bytes"] + t["hbm_write_bytes"]
                        k["traffic_detail"] = t
                        for f in (k.get("forms") or []):
                            if "slots" not in f["form"]:
                                f["traffic"] = k["traffic"]
                if slot_forms and len(traffic) >= len(kernels) + 2:  # [.., pack pass, slot launches]
                    t = traffic[len(kernels) + 1]
                    if "hbm_read_bytes" in t and "hbm_write_bytes" in t:
                        for f in slot_forms:
                            f["traffic"] = t["hbm_read_bytes"] + t["hbm_write_bytes"]
                            f["traffic_detail"] = t
                roofline["traffic"] = head["traffic"]
                if "traffic_detail" in head:
                    roofline["traffic_detail"] = head["traffic_detail"]
        elif roofline is not None:
            roofline["traffic_source"] = "not collected (N > 1, --no-pmc or another dataset)"
    line["roofline"] = roofline

    if single and not args.no_cpu_baseline:
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
