#!/bin/bash
# L2 requests (TCC_HIT_sum + TCC_MISS_sum) per launch of the kernel sweep on the reddit-shaped graph at its own size: are the
# narrow widths bound by requests?  (one request = one 128-byte line as counted by the L2)
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
mkdir -p gpurun_out
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/rq -- python3 dgl-0.5-benchmark_amd/kernel_bench.py --datasets reddit > gpurun_out/rq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
out = ["# TCC_HIT_sum + TCC_MISS_sum per launch (rocprofv3 --pmc, own pass), kernel_bench.py --datasets reddit (E = 114.6 M)"]
f = glob.glob("gpurun_out/rq/**/*counter_collection.csv", recursive=True)
acc = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "spmm_rowwave32" in k or "sddmm_coo32" in k:
        acc.setdefault(k[:66], collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    hit, miss = c.get("TCC_HIT_sum", [0]), c.get("TCC_MISS_sum", [0])
    n = max(len(hit), 1)
    req = (sum(hit) + sum(miss)) / n
    out.append("%-68s launches %2d  L2 requests / launch %7.1f M  (per edge %.2f)  hit rate %.3f" %
               (k, n, req / 1e6, req / 114.6e6, sum(hit) / max(sum(hit) + sum(miss), 1)))
open("gpurun_out/r02_reddit_l2_requests.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf gpurun_out/rq
