"""The BASELINE.json configurations other than the headline, timed inside bench.py's default N = 1 run (`secondary` in the JSON line).

VERDICT r04 item 2: arxiv SAGE (configs[1], main_dgl_arxiv_sage.py:141-148), the 2-layer 8-head GAT on the reddit shape with the
README's 11.6 M edges (configs[2]) and the reference script's own 3 x 1-head GAT on DGL's 114.6 M-edge reddit
(main_dgl_reddit_gat.py:87-96), molhiv GCN at batch size 256, eager and as one captured HIP graph (configs[4],
main_dgl_molhiv_gcn.py:143-145).  Each entry: `ms_per_step` (one training epoch, host-synchronised as the scripts do), the step
count, and a `roofline` entry for the workload's dominant hot-path call: SURVEY 8d's algorithmic bytes / the mean duration between
two HIP events on the launch stream around that C-ABI call (sparse.PROFILE), against the 8 TB/s HBM peak.

The same launchers run stand-alone through generate_result.py (profiles/rNN_generate_result.txt); this module only reuses their
model / step functions so that the driver's own bench run carries the numbers.
"""
import gc
import os
import time

import torch
import torch.nn as nn

HBM_PEAK_GBPS = 8000.0


def _bytes_spmm(r):
    """copy_u 4(N_dst+1) + 4E + 4 N_src D + 4 N_dst D; copy_e 4(N+1) + 4E + 4ED + 4ND; u_mul_e adds 4EH + 4E."""
    n_dst, n_src, nnz, D = r["n_rows"], r["n_cols"], r["nnz"], r["out_len"]
    if r["op"] == "copy_rhs":
        return 4 * (n_dst + 1) + 4 * nnz + 4 * nnz * D + 4 * n_dst * D
    b = 4 * (n_dst + 1) + 4 * nnz + 4 * n_src * D + 4 * n_dst * D
    if r["op"] != "copy_lhs":
        b += 4 * nnz * r.get("heads", 1) + 4 * nnz
    return b


def _bytes_gat(r):
    """Fused GAT block (csrc/gatfused.hip; the per-walk figures of profiles/gat_roofline.py, every array once): forward = one in-CSR
    walk; backward = the in-CSR walk (d_er, t) + the out-CSR walk (d_feat, d_el) when the layer's input needs a gradient."""
    n, nnz, H, F = max(r["n_src"], r["n_dst"]), r["nnz"], r["H"], r["F"]
    idx, nd, nh = 4 * (n + 1) + 4 * nnz, 4 * n * H * F, 4 * n * H
    if r["kernel"] == "gat_fwd":
        return idx + nd + 2 * nh + nd + 4 * nh
    dst = idx + 3 * nd + nh + 4 * nh + 2 * nh
    src = idx + 2 * nd + 4 * nh + nh + nd + nh
    return dst + (src if r.get("source_walk", True) else 0)


def _bytes_sddmm(r):
    """8E + 4 N_src D + 4 N_dst D + 4 E D_out (an operand addressed by edge id streams E rows instead)."""
    nnz, out = r["nnz"], r["out_len"]
    rows = {"u": r["n_src"], "v": r["n_dst"], "e": nnz}
    lt, rt = r["targets"][0], r["targets"][1]
    return 8 * nnz + 4 * rows[lt] * r["l_len"] + 4 * rows[rt] * r["r_len"] + 4 * nnz * out


def _family(r):
    """Calls of one kernel family at one width -- whatever the graph (the batches of molhiv differ in size from step to step)."""
    k = r.get("kernel", "spmm")
    if k == "spmm":
        return ("g-SpMM %s/%s" % (r["op"], r["reduce"]), r["out_len"])
    if k in ("gat_fwd", "gat_bwd"):
        return ("fused GAT block %s (%s form), H=%d%s" % ("forward" if k == "gat_fwd" else "backward", r["form"], r["H"],
                                                           "" if r.get("source_walk", True) else ", no source walk"), r["H"] * r["F"])
    if k == "sddmm":
        return ("g-SDDMM %s (%s)" % (r["op"], r["targets"]), r["out_len"])
    return (k, 0)


_BYTES = {"spmm": _bytes_spmm, "gat_fwd": _bytes_gat, "gat_bwd": _bytes_gat, "sddmm": _bytes_sddmm}


def dominant_roofline(records, steps):
    """The hot-path call family with the largest summed device time in the profiled steps -> a `roofline` object: SURVEY 8d's
    algorithmic bytes summed over its launches / their summed duration (== bytes per launch / mean duration on one graph)."""
    fams = {}
    for r in records:
        if r.get("kernel", "spmm") not in _BYTES or r.get("variant") == "row-sparse":
            continue
        fams.setdefault(_family(r), []).append(r)
    if not fams:
        return None
    tot = {k: sum(x["start"].elapsed_time(x["end"]) for x in v) for k, v in fams.items()}
    key = max(tot, key=tot.get)
    sel = fams[key]
    algo_sum = sum(_BYTES[x.get("kernel", "spmm")](x) for x in sel)
    achieved = algo_sum / (tot[key] * 1e-3) / 1e9
    all_ms = sum(tot.values())
    rows = [x.get("n_rows", x.get("n_dst")) for x in sel]
    return {"bound": "hbm", "kernel": "%s, D=%d" % key, "rows": int(sum(rows) / len(rows)), "nnz": int(sum(x["nnz"] for x in sel) / len(sel)),
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
            "algorithmic_bytes_per_launch": int(algo_sum / len(sel)), "avg_launch_ms": round(tot[key] / len(sel), 4), "launches_timed": len(sel),
            "launches_per_step": round(len(sel) / max(steps, 1), 2),
            "share_of_hot_path_device_time": round(tot[key] / all_ms, 3) if all_ms > 0 else None,
            "hot_path_ms_per_step": round(all_ms / max(steps, 1), 4)}


def _timed(step, steps, warmup, sync=None, profile_steps=None):
    """Warm-up, then `steps` steps between two device synchronisations -- the number reported -- and AFTER them `profile_steps` more
    steps with HIP events around every hot-path call (sparse.PROFILE) for the roofline entry: the events cost host time, which a
    launch-bound loop (molhiv eager: ~300 launches per batch) would show in its step time."""
    from mi355x_graph import sparse
    sync = sync or torch.cuda.synchronize
    for _ in range(warmup):
        step()
    sync()
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        elapsed = time.perf_counter() - t0
        n_prof = steps if profile_steps is None else profile_steps
        sparse.PROFILE = []
        try:
            for _ in range(n_prof):
                step()
            sync()
        finally:
            records, sparse.PROFILE = sparse.PROFILE, None
    finally:
        if was:
            gc.enable()
    return elapsed / steps * 1e3, records, n_prof


def sage_arxiv(device, steps=30, warmup=5):
    """configs[1]: 3-layer GraphSAGE, hidden 256, BatchNorm, on the bidirected arxiv shape (main_dgl_arxiv_sage.py:141-148,162)."""
    import full_graph
    cfg, data, g, model, train_idx, opt = full_graph.build_sage("arxiv", device)
    ms, recs, n_prof = _timed(lambda: full_graph.sage_train_step(model, g, data.features, data.labels, train_idx, opt), steps, warmup, profile_steps=5)
    return {"workload": "configs[1]: 3-layer GraphSAGE (hidden 256, BatchNorm) full-graph on the ogbn-arxiv shape, bidirected "
                        "(N=%d, E=%d)" % (g.number_of_nodes(), g.number_of_edges()),
            "reference": "main_dgl_arxiv_sage.py:141-148", "ms_per_step": round(ms, 4), "steps": steps, "warmup": warmup,
            "value_edges_per_s": full_graph.spmm_edges_per_epoch(cfg["num_layers"], g.number_of_edges()) / (ms * 1e-3),
            "roofline": dominant_roofline(recs, n_prof)}


def _gat(device, dataset, layers, heads, hidden, dropout, steps, warmup, label, ref):
    import dgl
    import full_graph
    from mi355x_graph.datasets import NodeData
    data = NodeData(dataset, device=device)
    g = dgl.add_self_loop(data.graph).int().to(device)
    torch.manual_seed(7)
    model = full_graph.GAT(g, layers, data.features.shape[1], hidden, data.num_classes, [heads] * (layers - 1) + [1],
                           feat_drop=dropout, attn_drop=dropout).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=0.003, weight_decay=2.4e-5)
    loss_fcn = nn.CrossEntropyLoss()
    ms, recs, n_prof = _timed(lambda: full_graph.gat_train_step(model, data.features, data.labels, data.train_mask, opt, loss_fcn), steps, warmup,
                              profile_steps=4)
    return {"workload": "%s (N=%d, E=%d incl. self loops, in=%d, hidden=%d, heads=%d, %d layers, dropout %g)"
                        % (label, g.number_of_nodes(), g.number_of_edges(), data.features.shape[1], hidden, heads, layers, dropout),
            "reference": ref, "ms_per_step": round(ms, 4), "steps": steps, "warmup": warmup, "roofline": dominant_roofline(recs, n_prof)}


def gat8_reddit_small(device, steps=20, warmup=4):
    """configs[2] as worded: 2 layers, 8 heads, the README's 11.6 M-edge reddit."""
    return _gat(device, "reddit-small", 2, 8, 16, 0.0, steps, warmup, "configs[2]: 2-layer 8-head GAT full-graph on the reddit shape, 11.6 M edges",
                "main_dgl_reddit_gat.py:31-55 (model class), reddit/ns-gat-dgl.py:39-42 (8 heads x 2 layers)")


def gat_reddit(device, steps=10, warmup=3):
    """The reference script's own defaults on DGL's reddit: 3 layers, 1 head, hidden 16, 114.6 M directed edges."""
    return _gat(device, "reddit", 3, 1, 16, 0.18074706609292976, steps, warmup,
                "configs[2] at the script's defaults: 3-layer 1-head GAT full-graph on DGL's reddit, 114.6 M edges", "main_dgl_reddit_gat.py:87-96")


def gcn_molhiv(device, epochs=2, num_graphs=32901, batch_size=256):
    """configs[4]: the reference's 5-layer GCN-with-bond-encoder on the molhiv shape, batch size 256 -- eager, then the same loop as one
    captured HIP graph replayed per batch.  A step = one EPOCH (129 batches).  (BASELINE words it "GIN"; SURVEY Appendix D.)"""
    import graph_classification as gc_
    from dgl.dataloading import GraphDataLoader
    from mi355x_graph.datasets import molhiv_like
    torch.set_num_threads(max(1, min(int(os.environ.get("MGX_HOST_THREADS", "4")), os.cpu_count() or 1)))
    data = molhiv_like(num_graphs)
    loader = GraphDataLoader(data, batch_size=batch_size, shuffle=True, num_workers=0)
    loss_fn = nn.BCEWithLogitsLoss()
    out = {"workload": "configs[4]: 5-layer GCN with bond encoder (UDF message + fn.sum, AvgPooling) on the ogbg-molhiv shape, "
                       "%d graphs, batch size %d, emb 256; a step = one epoch of %d batches" % (num_graphs, batch_size, len(loader)),
           "reference": "main_dgl_molhiv_gcn.py:143-145", "steps": epochs, "warmup": 1}
    torch.manual_seed(0)
    model = gc_.GCN(256, 1, 5, 0.5).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=0.001)
    ms, recs, n_prof = _timed(lambda: gc_.train_epoch(model, device, loader, opt, loss_fn), epochs, 1, profile_steps=1)
    out["eager"] = {"ms_per_step": round(ms, 2), "roofline": dominant_roofline(recs, n_prof),
                    "note": "host-bound: ~300 launches per batch (profiles/r02_molhiv_kernel_stats.txt)"}
    del model, opt
    torch.manual_seed(0)
    model = gc_.convert_masked_batchnorm(gc_.GCN(256, 1, 5, 0.5).to(device))
    opt = torch.optim.Adam(model.parameters(), lr=0.001, capturable=True)
    n_pad, e_pad = gc_.GraphedBatchTrainer.static_shape(data, batch_size)
    trainer = gc_.GraphedBatchTrainer(model, opt, loss_fn, device, batch_size, n_pad, e_pad)

    def graphed_epoch():
        gc_.train_epoch_graphed(trainer, loader)
        torch.cuda.synchronize()
    ms2, _, _ = _timed(graphed_epoch, epochs, 1, profile_steps=0)
    out["captured"] = {"ms_per_step": round(ms2, 2), "static_batch_shape": [int(n_pad), int(e_pad)],
                       "replayed": trainer.stats["replayed"], "oversize_batches_split": trainer.stats["split"],
                       "note": "one HIP graph of the padded batch step replayed per batch (graph_classification.GraphedBatchTrainer); "
                               "no per-call events inside a replay"}
    out["ms_per_step"] = out["captured"]["ms_per_step"]
    return out


LEGS = [("sage_arxiv", sage_arxiv), ("gat8_reddit_small", gat8_reddit_small), ("gat_reddit", gat_reddit), ("gcn_molhiv", gcn_molhiv)]


def run(device, only=None, budget_s=75.0, progress=None):
    """Every leg in turn (each releases its graph before the next starts); a leg that fails reports its error, never loses the line.
    Legs that would start after `budget_s` seconds are skipped and say so."""
    out, t0 = {}, time.perf_counter()
    for name, fn in LEGS:
        if only and name not in only:
            continue
        if time.perf_counter() - t0 > budget_s:
            out[name] = {"skipped": "the secondary block's %.0f s budget was used up by the legs before it" % budget_s}
            continue
        t1 = time.perf_counter()
        try:
            out[name] = fn(device)
        except Exception as err:  # noqa: BLE001
            out[name] = {"error": "%s: %s" % (type(err).__name__, str(err)[:300])}
        out[name]["leg_wall_s"] = round(time.perf_counter() - t1, 2)
        if progress:
            progress("secondary %s: %s" % (name, {k: v for k, v in out[name].items() if k in ("ms_per_step", "error", "leg_wall_s")}))
        gc.collect()
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    only = set(sys.argv[1].split(",")) if len(sys.argv) > 1 else None
    print(json.dumps(run(torch.device("cuda:0"), only=only, budget_s=1e9, progress=lambda m: print(m, file=sys.stderr, flush=True))))
