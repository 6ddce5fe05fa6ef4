"""The clustered control graph (kernel_controls.clustered_edges): does a graph with real neighbourhood overlap change what bounds the
g-SpMM?  (a) its average local clustering on a scaled-down copy (exact, CPU-sized), (b) distinct-source reuse per tile of the
locality schedule, (c) the row kernel and the LDS-staged tile kernel on it at full size.

  python experiments/exp_clustered_control.py [--scale 1.0] [--widths 64]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import kernel_controls as kc  # noqa: E402
from mi355x_graph import schedule, sparse, tileplan  # noqa: E402

dev = torch.device("cuda:0")


def clustering(n, src, dst, sample=2000, seed=0):
    """Average local clustering coefficient over `sample` random nodes of the simple undirected graph (exact per node)."""
    key = torch.unique(torch.minimum(src, dst) * n + torch.maximum(src, dst))
    u, v = key // n, key % n
    keep = u != v
    u, v = u[keep], v[keep]
    a = torch.cat([u, v])
    b = torch.cat([v, u])
    order = torch.argsort(a)
    a, b = a[order], b[order]
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=a.device)
    ptr[1:] = torch.cumsum(torch.bincount(a, minlength=n), 0)
    edge_set = a * n + b  # sorted by a, not by b: sort fully for searchsorted
    edge_set = torch.sort(edge_set)[0]
    g = torch.Generator(device="cpu").manual_seed(seed)
    nodes = torch.randint(0, n, (sample,), generator=g).tolist()
    tot, cnt = 0.0, 0
    for x in nodes:
        nb = b[ptr[x]:ptr[x + 1]]
        k = nb.numel()
        if k < 2:
            continue
        if k > 400:
            nb = nb[torch.randperm(k, device=nb.device)[:400]]
            k = 400
        pairs = (nb.view(-1, 1) * n + nb.view(1, -1)).flatten()
        pos = torch.searchsorted(edge_set, pairs).clamp(max=edge_set.numel() - 1)
        links = int((edge_set[pos] == pairs).sum())  # ordered pairs
        tot += links / (k * (k - 1))
        cnt += 1
    return tot / max(cnt, 1)


def reuse(csr, order, R):
    n = csr.num_rows
    pos = torch.empty(n, dtype=torch.int64, device=dev)
    pos[order.long()] = torch.arange(n, device=dev)
    deg = (csr.indptr[1:] - csr.indptr[:-1]).long()
    row = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    key = (pos[row] // R) * csr.num_cols + csr.indices.long()
    return csr.nnz / torch.unique(key).numel()


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--scale", type=float, default=1.0)
    p.add_argument("--widths", default="64")
    p.add_argument("--graphs", default="clustered,products")
    p.add_argument("--clustering-scale", type=float, default=1.0, help="size of the copy the local clustering is measured on")
    p.add_argument("--clustering-only", action="store_true")
    args = p.parse_args()
    be = sparse.backend_for(torch.zeros(1, device=dev))
    for kind in args.graphs.split(","):
        ns, (s, d) = kc.control_edges(kind, dev, args.clustering_scale)
        print("%s: average local clustering %.3f over 2000 sampled nodes (scale %g: N=%d E=%d)" % (kind, clustering(ns, s, d), args.clustering_scale, ns, s.numel()), flush=True)
        del s, d
        torch.cuda.empty_cache()
        if args.clustering_only:
            continue
        n, (src, dst) = kc.control_edges(kind, dev, args.scale)
        csr = sparse.coo_to_csr(n, n, dst.int().contiguous(), src.int().contiguous())
        del src, dst
        t0 = time.time()
        plan = csr.plan()
        order = csr._row_order[0]
        print("  N=%d E=%d; schedule %.1fs; edges / distinct sources per tile of the schedule: %s" %
              (n, csr.nnz, time.time() - t0, ", ".join("R=%d %.2f" % (R, reuse(csr, order, R)) for R in (16, 64, 256, 336))), flush=True)
        for D in [int(w) for w in args.widths.split(",")]:
            x = torch.rand(n, D, device=dev)
            ms = kc.time_spmm(csr, x)
            algo = kc.spmm_algorithmic_bytes(n, n, csr.nnz, D)
            print("  row kernel D=%d: %.3f ms = %.1f GB/s algorithmic (%.1f %% of 8 TB/s)" % (D, ms, algo / ms / 1e6, algo / ms / 1e6 / 80), flush=True)
            ref = sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None)[0]
            for cfg in ((14, 6, 2, 2), (14, 6, 2, 3)):
                base = schedule.plan_for(csr, split=tileplan.TILE_SPLIT)
                tp = tileplan.build_tile_plan(csr, base, *cfg)
                tileplan.validate(tp, csr)
                out = be.spmm_tile_copy_u(csr, tp, "sum", x)
                err = float(((out - ref).abs() / (ref.abs() + 1)).max())
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(6)]
                for a, b in ev:
                    a.record()
                    be.spmm_tile_copy_u(csr, tp, "sum", x)
                    b.record()
                torch.cuda.synchronize()
                t = sorted(a.elapsed_time(b) for a, b in ev)[2]
                print("  tile kernel %s D=%d: %.3f ms (%.1f %% of 8 TB/s); staged %.0f %%, gathered rows / edge %.3f, lds fill %.2f; max rel diff %.1e"
                      % ("x".join(map(str, cfg)), D, t, algo / t / 1e6 / 80, 100.0 * tp.stats["staged_edges"] / tp.stats["edges"],
                         tp.stats["gathered_rows_per_edge"], tp.stats["lds_slot_fill"], err), flush=True)
                del tp
        del csr, plan
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
