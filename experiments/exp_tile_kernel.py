"""LDS-staged tile g-SpMM (csrc/spmm_tile.hip) against the row-per-wave kernel: correctness on a small graph with hubs, then
time on a dataset-shaped graph.

  python experiments/exp_tile_kernel.py [reddit|proteins|products] [--widths 64,128] [--configs 12x6x4x2,...] [--split 2048]
  config = consumers x nacc x loaders x tau
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
from mi355x_graph import schedule, sparse, tileplan  # noqa: E402
from mi355x_graph.datasets import SHAPES, synthetic_edges  # noqa: E402

os.environ["MGX_TILE"] = "0"  # gspmm_raw below is the ROW kernel (the baseline); the tile kernel is called directly
dev = torch.device("cuda:0")
be = sparse.backend_for(torch.zeros(1, device=dev))


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2], t[0]


def make_csr(n, src, dst):
    return sparse.coo_to_csr(n, n, dst.int().contiguous(), src.int().contiguous())


def check_small():
    torch.manual_seed(0)
    n, m = 6000, 300000
    src, dst = synthetic_edges(n, m, 3000, 11, dev, symmetric=True)
    csr = make_csr(n, src, dst)
    base = schedule.build_plan(csr, torch.randperm(n, device=dev), split=512, order_kind="cluster")
    ok = True
    for (nc, nacc, nl, tau, lg) in ((12, 6, 4, 2, 4), (14, 5, 2, 2, 4), (7, 12, 1, 3, 4), (14, 8, 2, 2, 4), (7, 8, 1, 2, 4),
                                    (7, 8, 1, 3, 3), (7, 6, 1, 2, 3), (7, 8, 1, 3, 2), (7, 4, 1, 2, 2), (7, 6, 1, 2, 2)):
        tp = tileplan.build_tile_plan(csr, base, nc, nacc, nl, tau, lanes_log2=lg)
        tileplan.validate(tp, csr)
        shapes = {4: ((64, 64), (128, 128), (100, 100), (64, 160), (256, 256), (36, 36)),
                  3: ((32, 32), (24, 24), (64, 64), (36, 36), (32, 96), (16, 16)),
                  2: ((16, 16), (8, 8), (12, 12), (32, 32), (20, 20), (16, 48))}[lg]
        for D, stride in shapes:
            xw = torch.rand(n, stride, device=dev)
            x = xw[:, :D]
            ref = torch.zeros(n, D, dtype=torch.float64, device=dev).index_add_(0, dst, x.double()[src])
            for reduce in ("sum", "mean"):
                want = ref if reduce == "sum" else ref / csr.degrees().clamp(min=1).double().unsqueeze(1)
                got = be.spmm_tile_copy_u(csr, tp, reduce, x)
                err = float(((got.double() - want).abs() / (want.abs() + 1.0)).max())
                acc0 = torch.rand(n, D, device=dev)
                got2 = be.spmm_tile_copy_u(csr, tp, reduce, x, out2d=acc0.clone(), accumulate=True)
                err2 = float(((got2.double() - want - acc0.double()).abs() / (want.abs() + 1.0)).max())
                bad = err > 1e-5 or err2 > 1e-5
                ok = ok and not bad
                if bad or D == shapes[0][0]:
                    print("small %s D=%d stride=%d %s: rel err %.2e, accumulate %.2e %s" % ((nc, nacc, nl, tau, lg), D, stride, reduce,
                                                                                        err, err2, "FAIL" if bad else "ok"), flush=True)
        # bitwise rerun
        x = torch.rand(n, shapes[0][0], device=dev)
        a, b = be.spmm_tile_copy_u(csr, tp, "sum", x), be.spmm_tile_copy_u(csr, tp, "sum", x)
        ok = ok and bool(torch.equal(a, b))
    print("small graph:", "PASS" if ok else "FAIL", flush=True)
    return ok


def main():
    p = argparse.ArgumentParser()
    p.add_argument("dataset", nargs="?", default="reddit")
    p.add_argument("--widths", default="64,128")
    p.add_argument("--configs", default="12x6x4x2,12x6x4x3,12x4x4x2,14x8x2x2")
    p.add_argument("--split", type=int, default=2048)
    p.add_argument("--lg", type=int, default=4, help="log2 lanes per row: 4 = 64-column passes, 3 = 32, 2 = 16")
    p.add_argument("--scale", type=float, default=1.0)
    p.add_argument("--balance", default="0", help="workgroup slots the tiles are balanced over (0 = tiles of equal item count), e.g. 0,512")
    p.add_argument("--skip-small", action="store_true")
    p.add_argument("--small-only", action="store_true")
    args = p.parse_args()
    if not args.skip_small and not check_small():
        sys.exit(1)
    if args.small_only:
        return
    spec = SHAPES[args.dataset]
    n, m = int(spec["n"] * args.scale), int(spec["m"] * args.scale)
    src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], dev, symmetric=spec["symmetric"])
    csr = make_csr(n, src, dst)
    E = csr.nnz
    del src, dst
    t0 = time.time()
    row_plan = csr.plan()  # the row kernel's own plan (split 256, LP order)
    torch.cuda.synchronize()
    print("%s: N=%d E=%d; row plan %.1fs (%d items)" % (args.dataset, n, E, time.time() - t0, row_plan.num_items), flush=True)
    base = schedule.plan_for(csr, split=args.split)
    print("tile base plan: split %d -> %d items, %d hubs" % (args.split, base.num_items, base.num_hubs), flush=True)
    widths = [int(w) for w in args.widths.split(",")]
    xs = {D: torch.rand(n, D, device=dev) for D in widths}
    row_out = {}
    for D in widths:
        x = xs[D]
        med, best = timeit(lambda: sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None))
        row_out[D] = sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None)[0]
        print("row kernel  D=%-4d %.3f ms (best %.3f)  delivered %.1f TB/s" % (D, med, best, E * D * 4 / med / 1e9), flush=True)
    for cfg, bal in [(c, int(b)) for c in args.configs.split(",") for b in args.balance.split(",")]:
        nc, nacc, nl, tau = [int(v) for v in cfg.split("x")]
        t0 = time.time()
        tp = tileplan.build_tile_plan(csr, base, nc, nacc, nl, tau, balance=bal, lanes_log2=args.lg)
        cfg = "%s balance %d" % (cfg, bal)
        torch.cuda.synchronize()
        tb = time.time() - t0
        tileplan.validate(tp, csr)
        st = tp.stats
        print("config %s: build %.1fs tiles %d chunks %d staged %.1f%% gathered rows/edge %.3f lds fill %.2f dir fill %.2f" %
              (cfg, tb, st["tiles"], st["chunks"], 100.0 * st["staged_edges"] / st["edges"], st["gathered_rows_per_edge"],
               st["lds_slot_fill"], st["dir_slot_fill"]), flush=True)
        cost = tileplan.tile_costs(tp)
        span, ideal = tileplan.simulate_dispatch(cost)
        per = (st["tiles"] + 7) // 8
        xs_sum = [float(cost[i * per:(i + 1) * per].sum()) for i in range(8)]
        print("  dispatch model: makespan %.0f against %.0f divisible (%.2f); per-XCD work max / mean %.3f; tile cost max / mean %.2f" %
              (span, ideal, span / max(ideal, 1e-9), max(xs_sum) / (sum(xs_sum) / 8), float(cost.max() / cost.mean())), flush=True)
        for D in widths:
            x = xs[D]
            out = be.spmm_tile_copy_u(csr, tp, "sum", x)
            err = float(((out - row_out[D]).abs() / (row_out[D].abs() + 1.0)).max())
            med, best = timeit(lambda: be.spmm_tile_copy_u(csr, tp, "sum", x))
            print("  tile D=%-4d %.3f ms (best %.3f)  max rel diff vs row kernel %.2e" % (D, med, best, err), flush=True)
        del tp
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
