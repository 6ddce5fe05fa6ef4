"""Products-shaped g-SpMM as P accumulating passes over DISJOINT source sets: pass k gathers only the sources whose position in the
locality schedule falls into a block b with b % P == k (blocks of S rows), so that at any time an XCD's L2 has to hold 1/P of a
community's rows.  Costs P row walks (index streams, a read-modify-write of the output per extra pass) against fewer L2 misses.

  python experiments/exp_source_passes.py [--scale 1.0] [--D 64] [--blocks 3072,6144,12288] [--passes 2,4]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import kernel_controls as kc  # noqa: E402
from mi355x_graph import schedule, sparse  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--graph", default="products")
    p.add_argument("--scale", type=float, default=1.0)
    p.add_argument("--D", type=int, default=64)
    p.add_argument("--blocks", default="3072,6144,12288")
    p.add_argument("--passes", default="2,4")
    args = p.parse_args()
    n, (src, dst) = kc.control_edges(args.graph, dev, args.scale)
    src, dst = src.long(), dst.long()
    csr = sparse.coo_to_csr(n, n, dst.int().contiguous(), src.int().contiguous())
    csr.plan()
    order = csr._row_order[0]
    pos = torch.empty(n, dtype=torch.int64, device=dev)
    pos[order.long()] = torch.arange(n, device=dev)
    x = torch.rand(n, args.D, device=dev)
    base = timeit(lambda: sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None))
    ref = sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None)[0]
    print("%s N=%d E=%d D=%d: one pass %.3f ms" % (args.graph, n, csr.nnz, args.D, base), flush=True)
    split = int(os.environ.get("MGX_SPLIT", 256))
    for S in [int(v) for v in args.blocks.split(",")]:
        for P in [int(v) for v in args.passes.split(",")]:
            blk = (pos[src] // S) % P
            parts = []
            for k in range(P):
                m = blk == k
                c = sparse.coo_to_csr(n, n, dst[m].int().contiguous(), src[m].int().contiguous())
                c._plan = schedule.build_plan(c, order, split, "cluster")
                c._tile_plan = None
                parts.append(c)
            out = torch.empty(n, args.D, device=dev)

            def run():
                sparse.gspmm_raw(parts[0], "copy_lhs", "sum", x, None, accumulate_into=None)
                o = sparse.gspmm_raw(parts[0], "copy_lhs", "sum", x, None)[0]
                for c in parts[1:]:
                    sparse.gspmm_raw(c, "copy_lhs", "sum", x, None, accumulate_into=o)
                return o

            def run_timed():
                o = sparse.gspmm_raw(parts[0], "copy_lhs", "sum", x, None)[0]
                for c in parts[1:]:
                    sparse.gspmm_raw(c, "copy_lhs", "sum", x, None, accumulate_into=o)
                return o

            o = run_timed()
            err = float(((o - ref).abs() / (ref.abs() + 1)).max())
            t = timeit(run_timed)
            each = [timeit(lambda c=c: sparse.gspmm_raw(c, "copy_lhs", "sum", x, None)) for c in parts]
            print("  blocks of %5d rows, %d passes: %.3f ms total (passes alone: %s)  max rel diff %.1e" %
                  (S, P, t, ", ".join("%.3f" % v for v in each), err), flush=True)
            del parts, out


if __name__ == "__main__":
    main()
