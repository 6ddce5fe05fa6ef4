"""Scaling model of the partitioned full-graph epoch, measured on ONE GPU (bench.py --emulate-ranks / config.partition.predicted).

The metric is the 1/2/4/8-GPU curve (BASELINE.json); the reference itself is single-GPU (README.md:12), so the P-way program is
this package's own (mi355x_graph/dist.py).  A 1-GPU box cannot run RCCL across ranks, but it can run every rank's
share of the work: `mi355x_graph.emulate.EmuWorld` executes the P ranks of the real program -- same partitioner, same local
CSRs and schedules, same autograd nodes (dist.DistSageMeanCatFn), same kernels -- one at a time on cuda:0, with each
all_to_all replaced by device copies between the ranks' buffers, and records per rank the device time of every stretch of
work between collectives (HIP events) and the rows every exchange moves between every pair.  `emulate.price_epoch` then
replays the ranks in lock step with each exchange priced at a stated per-link rate: slice q -> r on its own xGMI link, full
duplex, startable when both ranks have posted.

What the model does NOT contain: RCCL's copy kernels competing with the aggregation for CUs and HBM during the window, and the host cost
of the RCCL calls themselves.  Two timings of a rank are reported (mi355x_graph/emulate.py's docstring): the traced stretches -- the time
between two events on the rank's stream, idle gaps included, with the host starting from an empty queue after every collective:
conservative, and as sensitive to the host's launch speed as a host-bound stretch is -- and the rank's SOLO epochs, run back to back with
recorded payloads: a rank at P = 8 is then device-bound (busy 4.3 of 4.8 ms under the profiler, docs/LOG_r05.md section 9b).
"""
import gc
import os
import tempfile
import time

import torch

LINK_RATES_GBPS = (32.0, 48.0, 64.0)  # per link and direction; see report(): MI355X xGMI is 7 links x ~153 GB/s bidirectional


def run(device, src, dst, n, feats, labels, train_mask, cfg, spec, P, steps=5, warmup=2, dropout=None, progress=None, tune_dense=False,
        host_profile=None, solo_extra=0, solo_clock=None):
    """Partition the graph P ways and run `warmup` + `steps` training epochs of every rank (the step of bench.py) inside an
    emulated world.  Returns a dict: partition statistics, per-rank typical-epoch stages (median over the steps), byte matrices, priced epochs."""
    import full_graph
    from mi355x_graph import dist as mdist, emulate, ops

    say = progress or (lambda msg: None)
    t0 = time.perf_counter()
    assign, pstats = mdist.cached_partition(src, dst, n, P)
    say("P=%d: partitioned (%.1f s, cut %.1f %%)" % (P, time.perf_counter() - t0, 100 * pstats["edge_cut"]))
    t0 = time.perf_counter()
    parts = [mdist.build_local_partition(src, dst, n, assign, r, P) for r in range(P)]
    torch.cuda.synchronize(device)
    say("P=%d: %d local partitions built (%.1f s)" % (P, P, time.perf_counter() - t0))
    total_train = float(train_mask.sum())
    p_drop = cfg["dropout"] if dropout is None else dropout

    def rank_main(rank):
        block, plan, own = parts[rank]
        ctx = emulate.current()
        g = mdist.DistGraph(block, plan)
        torch.manual_seed(1234)
        model = full_graph.GraphSAGE(spec["feat"], cfg["hidden"], spec["classes"], cfg["num_layers"], p_drop,
                                     cfg["batch_norm"], cfg["neigh_bias"]).to(device)
        if cfg["batch_norm"]:
            model = mdist.convert_batchnorm(model)
        model.rows_are_distinct = True
        own_c = own.cpu()
        x, y = feats[own_c].to(device), labels[own_c].to(device)
        g.set_static_input(x)
        train_idx = torch.nonzero(train_mask[own_c]).flatten().to(device)
        mdist.broadcast_parameters(model)
        bucket = mdist.GradBucket(model)
        opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"])

        def step():
            model.train()
            bucket.zero()
            y_train = y[train_idx]
            loss = ops.nll_sum(model(g, x, rows=train_idx), y_train) / total_train
            loss.backward()
            g._comm.mark("gradient all_reduce + optimizer")
            bucket.all_reduce()
            opt.step()
            g._comm.mark("dense")
            return loss.item()

        if tune_dense and rank == 0:
            # experiment (bench.py --tune-dense): PyTorch TunableOp picks the GEMMs of the per-rank shapes during the warm-up.  The
            # environment variable wins over torch.cuda.tunable.tuning_enable(); results go to a scratch file, never to the staged
            # copy of the committed selections.
            torch.cuda.tunable.enable(True)
            torch.cuda.tunable.set_filename(os.path.join(tempfile.gettempdir(), "mgx_tunableop_emulated_%d.csv" % os.getpid()))
            torch.cuda.tunable.set_max_tuning_duration(10)
            torch.cuda.tunable.set_max_tuning_iterations(5)
            os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
        for _ in range(warmup):
            step()
        if tune_dense:
            ctx.barrier()
            os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
        ctx.start_trace()  # one traced epoch thrown away: the first use of the timing events
        step()
        ctx.stop_trace()
        traces, loss = [], None
        for _ in range(steps):
            ctx.start_trace()
            loss = step()
            traces.append(ctx.stop_trace())
        lt = torch.tensor([loss], dtype=torch.float64, device=device)
        mdist.all_reduce(lt)
        torch.cuda.synchronize(device)
        # the same epochs with this rank ALONE on the device and every collective completed by a copy of what arrived in one recorded
        # epoch (emulate.py, "solo epochs"): host and device overlap as in a rank's own process, no token passing, no events
        kept = ctx.record_epoch(step)
        ctx.barrier()
        solo_ms = ctx.solo_epochs(step, kept, epochs=max(steps, 3), warmup=1)
        if host_profile and rank == 0:  # experiments/prof_rank_host.py: the Python side of rank 0's solo epochs
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
            ctx.solo_epochs(step, kept, epochs=max(steps, 3), warmup=0)
            prof.disable()
            prof.dump_stats(host_profile)
        if solo_extra and rank == 0:  # experiments/solo_trace.sh: a long stretch of one rank's solo epochs for a kernel trace to look at
            torch.cuda.synchronize(device)
            stamp = [time.clock_gettime_ns(c) for c in (time.CLOCK_MONOTONIC, time.CLOCK_BOOTTIME, time.CLOCK_REALTIME)]
            ms = ctx.solo_epochs(step, kept, epochs=solo_extra, warmup=0)
            stamp += [time.clock_gettime_ns(c) for c in (time.CLOCK_MONOTONIC, time.CLOCK_BOOTTIME, time.CLOCK_REALTIME)]
            if solo_clock:
                with open(solo_clock, "w") as fh:
                    fh.write(" ".join(str(v) for v in stamp) + " %d %.6f\n" % (solo_extra, ms))
        del kept
        ctx.barrier()
        return {"stages": emulate.typical_epoch(traces), "loss": float(lt.item()), "solo_epoch_ms": solo_ms, "n_own": plan.n_own, "n_halo": plan.n_halo,
                "send_rows": int(sum(plan.send_splits)), "recv_splits": list(plan.recv_splits), "local_edges": block.number_of_edges(),
                "halo_edges": int(plan.halo.num_edges()), "train_rows": int(train_idx.numel())}

    t0 = time.perf_counter()
    gc.collect()
    gc.disable()  # a collection inside a traced stretch is a millisecond of "device time" for whichever rank it hits (bench.py does the same)
    try:
        res = emulate.EmuWorld(P, device).run(rank_main)
    finally:
        gc.enable()
    say("P=%d: %d + %d epochs of every rank (%.1f s)" % (P, warmup, steps, time.perf_counter() - t0))
    return summarize(res, P, cfg["hidden"], pstats, n, int(src.shape[0]))


def _stage_ms(st, part):
    return sum(st.get(part, {}).values())


def summarize(res, P, D, pstats, n, num_edges):
    from mi355x_graph import emulate
    ranks = [r["stages"] for r in res]
    per_rank = []
    for r in res:
        by_label, window_ms = {}, []
        for st in r["stages"]:
            for part in ("pre", "window"):
                for label, ms in st.get(part, {}).items():
                    by_label[label] = by_label.get(label, 0.0) + ms
            if st["kind"] == "all_to_all":
                window_ms.append(round(_stage_ms(st, "window"), 4))
        per_rank.append({"owned_rows": r["n_own"], "halo_rows": r["n_halo"], "send_rows": r["send_rows"],
                         "local_edges": r["local_edges"], "halo_edges": r["halo_edges"], "train_rows": r["train_rows"],
                         "compute_ms": round(sum(by_label.values()), 4),
                         "solo_epoch_ms": None if r.get("solo_epoch_ms") is None else round(r["solo_epoch_ms"], 4),
                         "compute_ms_by_stretch": {k: round(v, 4) for k, v in sorted(by_label.items())},
                         "overlap_window_ms_per_exchange": window_ms})
    # halo byte matrix of ONE hidden-width exchange sent as DENSE rows: [receiver][sender] -- the reference point
    matrix = [[rows * D * 4 for rows in r["recv_splits"]] for r in res]
    # ... and what every message of the epoch really carried (dist.SparseHalo sends relu + dropout outputs as bitmaps + non-zeros and
    # takes their gradients back under the same bitmaps): per all_to_all stage, in program order
    a2a_idx = [k for k, st in enumerate(ranks[0]) if st["kind"] == "all_to_all"]
    messages = []
    for k in a2a_idx:
        mk = [[ranks[r][k]["info"]["recv_rows"][q] * ranks[r][k]["info"]["row_bytes"] for q in range(P)] for r in range(P)]
        messages.append({"tag": ranks[0][k]["info"].get("tag", "rows"), "max_pair_bytes": max(max(row) for row in mk),
                         "max_recv_bytes": max(sum(row) for row in mk), "total_bytes": sum(sum(row) for row in mk), "matrix": mk})
    pair_max = max(max(row) for row in matrix)
    a2a = [ranks[0][k] for k in a2a_idx if ranks[0][k]["info"].get("tag", "rows") != "values"]  # logical exchanges (a packed forward one = 2 messages)
    dense_total = sum(sum(row) for row in matrix) * len(a2a)
    sent_total = sum(m["total_bytes"] for m in messages)
    own = [r["n_own"] for r in res]
    edges = [r["local_edges"] for r in res]
    out = {"ranks": P, "edge_cut_pct": round(100.0 * sum(r["halo_edges"] for r in res) / max(num_edges, 1), 2),
           "num_clusters": pstats.get("num_clusters"), "final_loss": res[0]["loss"],
           "exchanges_per_epoch": len(a2a),
           "messages_per_epoch": [{k: v for k, v in m.items() if k != "matrix"} for m in messages],
           "message_bytes_matrices": [m["matrix"] for m in messages],
           "halo_bytes_per_epoch": {"sent": sent_total, "as_dense_rows": dense_total,
                                    "ratio": round(sent_total / dense_total, 4) if dense_total else None},
           "per_rank": per_rank,
           "halo_bytes_matrix_D%d" % D: matrix,
           "max_pair_bytes_per_exchange": pair_max,
           "max_recv_bytes_per_exchange": max(sum(row) for row in matrix),
           "imbalance": {"owned_rows_max_over_mean": round(max(own) * P / float(sum(own)), 3),
                         "in_edges_max_over_mean": round(max(edges) * P / float(sum(edges)), 3),
                         "compute_ms_max_over_mean": round(max(p["compute_ms"] for p in per_rank) * P
                                                           / sum(p["compute_ms"] for p in per_rank), 3)},
           "compute_ms_max": max(p["compute_ms"] for p in per_rank),
           "compute_ms_mean": round(sum(p["compute_ms"] for p in per_rank) / P, 4),
           "predicted": {}}
    solo = [p["solo_epoch_ms"] for p in per_rank]
    if all(v is not None for v in solo):
        out["solo_epoch_ms_max"], out["solo_epoch_ms_mean"] = max(solo), round(sum(solo) / P, 4)
    else:
        solo = None
    for rate in LINK_RATES_GBPS:
        over = emulate.price_epoch(ranks, rate, overlap=True)
        ser = emulate.price_epoch(ranks, rate, overlap=False)
        out["predicted"]["%g GB/s per link" % rate] = {
            "epoch_ms_overlapped": round(over["epoch_ms"], 3), "epoch_ms_not_overlapped": round(ser["epoch_ms"], 3),
            # a rank's solo epoch (exchange free of charge) + what the lock-step replay leaves exposed for that rank
            "epoch_ms_solo_plus_exposed": None if solo is None else round(max(solo[r] + sum(x[r]["exposed"] for x in over["exchanges"])
                                                                              for r in range(P)), 3),
            "exposed_exchange_ms_worst_rank": round(max(sum(x[r]["exposed"] for x in over["exchanges"]) for r in range(P)), 3),
            "exchange_ms_per_exchange_worst_rank": [round(max(x[r]["done"] - x[r]["posted"] for r in range(P)), 3)
                                                    for x in over["exchanges"]]}
    out["predicted"]["no exchange cost (compute only, lock step)"] = {
        "epoch_ms_overlapped": round(emulate.price_epoch(ranks, 1e9, latency_us=0.0, allreduce_us=0.0)["epoch_ms"], 3),
        "epoch_ms_solo_plus_exposed": None if solo is None else round(max(solo), 3)}
    out["stages_rank0"] = [{"kind": st["kind"], "pre_ms": round(_stage_ms(st, "pre"), 4),
                            "window_ms": round(_stage_ms(st, "window"), 4)} for st in ranks[0]]
    return out


def report(models, one_gpu_ms=None, header=""):
    """Human-readable table of run()'s results for several P (profiles/r04_scale_model.txt)."""
    lines = [header] if header else []
    lines.append("Per-link rates priced: %s GB/s per direction.  MI355X: 8 GPUs fully connected, 7 xGMI links per GPU at ~153 GB/s"
                 % ", ".join("%g" % r for r in LINK_RATES_GBPS))
    lines.append("bidirectional each = 76.8 GB/s per direction peak; 64 is ~83 % of that, 48 / 32 are what an all_to_all of this size may")
    lines.append("deliver when it shares HBM and CUs with the aggregation it hides behind (assumption, unmeasured: no multi-GPU box).")
    if one_gpu_ms:
        lines.append("1 GPU (measured, same process, the non-partitioned module form): %.3f ms / epoch" % one_gpu_ms)
    for m in models:
        P = m["ranks"]
        lines.append("")
        lines.append("== P = %d   edge cut %.2f %%   %d exchanges / epoch   final loss %.6f" % (P, m["edge_cut_pct"], m["exchanges_per_epoch"], m["final_loss"]))
        lines.append("   imbalance (max / mean): owned rows %.3f, in-edges %.3f, compute %.3f"
                     % (m["imbalance"]["owned_rows_max_over_mean"], m["imbalance"]["in_edges_max_over_mean"], m["imbalance"]["compute_ms_max_over_mean"]))
        lines.append("   rank  owned rows  halo rows  send rows   in-edges  halo edges  compute ms  solo epoch ms  windows ms (per exchange)")
        for r, p in enumerate(m["per_rank"]):
            lines.append("   %4d  %10d %10d %10d %10d %11d  %10.3f  %13s  %s" % (r, p["owned_rows"], p["halo_rows"], p["send_rows"], p["local_edges"],
                                                                                p["halo_edges"], p["compute_ms"],
                                                                                "-" if p.get("solo_epoch_ms") is None else "%.3f" % p["solo_epoch_ms"],
                                                                                " ".join("%.3f" % w for w in p["overlap_window_ms_per_exchange"])))
        labels = sorted({k for p in m["per_rank"] for k in p["compute_ms_by_stretch"]})
        lines.append("   compute by stretch, ms (mean over ranks / worst rank):")
        for lab in labels:
            vals = [p["compute_ms_by_stretch"].get(lab, 0.0) for p in m["per_rank"]]
            lines.append("      %-34s %8.3f / %8.3f" % (lab, sum(vals) / len(vals), max(vals)))
        key = [k for k in m if k.startswith("halo_bytes_matrix")][0]
        lines.append("   halo MB per exchange, receiver (row) x sender (column), %s:" % key[len("halo_bytes_matrix_"):])
        for row in m[key]:
            lines.append("      " + " ".join("%7.2f" % (b / 1e6) for b in row) + "   | recv %8.2f" % (sum(row) / 1e6))
        lines.append("   as dense rows: max pair %.2f MB, max received by one rank %.2f MB per exchange" % (m["max_pair_bytes_per_exchange"] / 1e6, m["max_recv_bytes_per_exchange"] / 1e6))
        if m.get("messages_per_epoch"):
            lines.append("   messages of one epoch as sent (program order; a packed forward exchange = bitmaps + values, its gradients = values only):")
            for i, msg in enumerate(m["messages_per_epoch"]):
                lines.append("      #%d %-16s total %8.2f MB   max pair %7.2f MB   max received by one rank %8.2f MB"
                             % (i, msg["tag"], msg["total_bytes"] / 1e6, msg["max_pair_bytes"] / 1e6, msg["max_recv_bytes"] / 1e6))
            hb = m["halo_bytes_per_epoch"]
            lines.append("   bytes on the links per epoch: %.1f MB sent against %.1f MB as dense rows = %.3f"
                         % (hb["sent"] / 1e6, hb["as_dense_rows"] / 1e6, hb["ratio"] or 0.0))
            for i, mk in enumerate(m.get("message_bytes_matrices", [])):
                if m["messages_per_epoch"][i]["tag"] in ("values", "gradient values") and i <= 2:
                    lines.append("   MB of message #%d (%s), receiver (row) x sender (column):" % (i, m["messages_per_epoch"][i]["tag"]))
                    for row in mk:
                        lines.append("      " + " ".join("%7.2f" % (b / 1e6) for b in row) + "   | recv %8.2f" % (sum(row) / 1e6))
        lines.append("   predicted epoch, ms:")
        for k, v in m["predicted"].items():
            if "epoch_ms_not_overlapped" in v:
                sp = (" speed-up %.2fx / %.2fx" % (one_gpu_ms / v["epoch_ms_overlapped"], one_gpu_ms / v["epoch_ms_not_overlapped"])) if one_gpu_ms else ""
                lines.append("      %-24s overlapped %7.3f   not overlapped %7.3f   exposed (worst rank) %6.3f   per exchange %s%s"
                             % (k, v["epoch_ms_overlapped"], v["epoch_ms_not_overlapped"], v["exposed_exchange_ms_worst_rank"],
                                " ".join("%.3f" % x for x in v["exchange_ms_per_exchange_worst_rank"]), sp))
            else:
                sp = (" speed-up %.2fx" % (one_gpu_ms / v["epoch_ms_overlapped"])) if one_gpu_ms else ""
                lines.append("      %-24s %7.3f%s" % (k, v["epoch_ms_overlapped"], sp))
            if v.get("epoch_ms_solo_plus_exposed") is not None:
                sp = (" speed-up %.2fx" % (one_gpu_ms / v["epoch_ms_solo_plus_exposed"])) if one_gpu_ms else ""
                lines.append("      %-24s solo epoch of the slowest rank + its exposed exchange %7.3f%s" % ("", v["epoch_ms_solo_plus_exposed"], sp))
    return "\n".join(lines) + "\n"
