"""Turns rocprofv3 CSV output (gpurun_out/<tag>/...) into the compact, committed summaries under
profiles/: per-kernel time table from --kernel-trace --stats, and per-launch HBM traffic of the
hot-path kernels from the separate --pmc passes.

gfx950 corrections (guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE is reported in KiB and tallies
128-byte requests at 64 bytes, so read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE (KiB) is exact:
write bytes = WRITE_SIZE * 1024.

  python profiles/summarize.py r01 gpurun_out/r01_bench_trace gpurun_out/r01_bench_fetch gpurun_out/r01_bench_write [gpurun_out/r01_bench_l2]
"""
import collections
import csv
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def find(d, suffix):
    f = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return max(f, key=os.path.getmtime) if f else None  # newest run in that directory


def short(name):
    return name if len(name) <= 100 else name[:97] + "..."


def kernel_stats(d, out):
    rows = list(csv.DictReader(open(find(d, "kernel_stats.csv"))))
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (top kernels by total time)\n")
        f.write("%-100s %8s %12s %12s %8s\n" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
        for r in rows[:40]:
            f.write("%-100s %8s %12.3f %12.2f %8s\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e3, r["Percentage"]))
    return rows


def pmc(d):
    rows = list(csv.DictReader(open(find(d, "counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag, trace = sys.argv[1], sys.argv[2]
    stats = kernel_stats(trace, os.path.join(HERE, tag + "_kernel_stats.txt"))
    counters = {}
    for d in sys.argv[3:]:
        for k, v in pmc(d).items():
            counters.setdefault(k, {}).update({c: vals for c, vals in v.items()})
    lines = ["# per-launch PMC averages of the mgx:: kernels (separate --pmc passes; gfx950 corrections applied)",
             "%-70s %8s %14s %14s %10s" % ("kernel", "launches", "hbm_read_MB", "hbm_write_MB", "L2_hit")]
    summary = {}
    for k, c in sorted(counters.items()):
        if "mgx::" not in k:
            continue
        n = max(len(v) for v in c.values())
        rd = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1) * 1024 * 2
        wr = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1) * 1024
        hit = sum(c.get("TCC_HIT_sum", [0]))
        miss = sum(c.get("TCC_MISS_sum", [0]))
        l2 = hit / (hit + miss) if hit + miss else float("nan")
        lines.append("%-70s %8d %14.1f %14.1f %10.3f" % (short(k)[:70], n, rd / 1e6, wr / 1e6, l2))
        summary[k] = {"launches": n, "hbm_read_bytes": rd, "hbm_write_bytes": wr, "l2_hit_rate": l2}
    if len(sys.argv) > 3:
        open(os.path.join(HERE, tag + "_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
    if len(sys.argv) > 3:
        print(open(os.path.join(HERE, tag + "_pmc_summary.txt")).read())
    print(open(os.path.join(HERE, tag + "_kernel_stats.txt")).read()[:3000])


if __name__ == "__main__":
    main()
