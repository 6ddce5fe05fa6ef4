"""Kernel trace of ONE rank's solo epochs (emulate.py): run under `rocprofv3 --kernel-trace --output-format csv -d DIR --` with
    python3 experiments/solo_trace.py run CLOCKFILE [P] [epochs] [scale]
then `python3 experiments/solo_trace.py read DIR CLOCKFILE`: device-busy fraction of the solo stretch, its kernels by total time, the gaps
by the kernel that precedes them."""
import collections
import csv
import glob
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def run(clock, P, epochs, scale):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dgl-0.5-benchmark_amd"))
    import tunable
    tunable.setup()
    import torch
    import dgl  # noqa: F401
    import full_graph
    import scale_model
    from mi355x_graph.datasets import SHAPES, synthetic_edges
    dev = torch.device("cuda:0")
    spec, cfg = SHAPES["products"], full_graph.SAGE_CONFIGS["products"]
    n, m = int(spec["n"] * scale), int(spec["m"] * scale)
    src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], dev, symmetric=True)
    gen = torch.Generator().manual_seed(1)
    feats = torch.rand(n, spec["feat"], generator=gen)
    labels = torch.randint(0, spec["classes"], (n,), generator=gen)
    train_mask = torch.rand(n, generator=gen) < 0.08
    res = scale_model.run(dev, src, dst, n, feats, labels, train_mask, cfg, spec, P, steps=3, warmup=2, solo_extra=epochs, solo_clock=clock,
                          progress=lambda s: print(s, flush=True))
    print("solo epoch per rank (ms):", [r["solo_epoch_ms"] for r in res["per_rank"]])


def read(out, clock):
    f = sorted(glob.glob(out + "/**/*kernel_trace.csv", recursive=True))[-1]
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", ""))
                for r in csv.DictReader(open(f)))
    vals = open(clock).read().split()
    stamps, epochs, ms = [int(v) for v in vals[:6]], int(vals[6]), float(vals[7])
    lo, hi = ks[0][0], ks[-1][1]
    pick = next((i for i in range(3) if lo <= stamps[i] <= hi and lo <= stamps[3 + i] <= hi), None)
    if pick is None:
        print("no host clock matches the trace's timestamps", lo, hi, stamps)
        return
    a, b = stamps[pick], stamps[3 + pick]
    sel = [k for k in ks if k[0] >= a and k[1] <= b]
    busy = sum(e - s for s, e, _ in sel)
    print("solo stretch: %d epochs, %.3f ms per epoch by the host clock; %d kernels = %.1f per epoch; device busy %.3f ms per epoch (%.1f %%)"
          % (epochs, ms, len(sel), len(sel) / epochs, busy / 1e6 / epochs, 100.0 * busy / (b - a)))
    h = collections.defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        k = n.split("(")[0][:100]
        h[k][0] += 1
        h[k][1] += e - s
    print("kernels per epoch by total time:")
    for k, v in sorted(h.items(), key=lambda kv: -kv[1][1])[:45]:
        print("%7.1f x %8.1f us = %8.3f ms  %s" % (v[0] / epochs, v[1] / v[0] / 1e3, v[1] / 1e6 / epochs, k))
    gaps = collections.defaultdict(lambda: [0, 0])
    for (s0, e0, n0), (s1, e1, n1) in zip(sel[:-1], sel[1:]):
        if s1 > e0:
            k = n0.split("(")[0][:60] + "  ->  " + n1.split("(")[0][:60]
            gaps[k][0] += 1
            gaps[k][1] += s1 - e0
    tot = sum(v[1] for v in gaps.values())
    print("idle between kernels: %.3f ms per epoch; the largest by the pair around them:" % (tot / 1e6 / epochs))
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
        print("%7.1f x %8.1f us = %8.3f ms  %s" % (v[0] / epochs, v[1] / v[0] / 1e3, v[1] / 1e6 / epochs, k))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8, int(sys.argv[4]) if len(sys.argv) > 4 else 100,
            float(sys.argv[5]) if len(sys.argv) > 5 else 1.0)
    else:
        read(sys.argv[2], sys.argv[3])
