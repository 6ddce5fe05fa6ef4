#!/bin/bash
# L2 requests (TCC_HIT_sum + TCC_MISS_sum) per launch of the fused GAT kernels on the reddit-shaped graph, with the gathered
# operands packed into whole-line rows and with the separate arrays (MGX_GAT_NO_PACK=1).  Counters in their own passes.
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
mkdir -p gpurun_out
for np in 0 1; do
  if [ $np = 1 ]; then export MGX_GAT_NO_PACK=1; else unset MGX_GAT_NO_PACK; fi
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/gatpack_$np -- python3 dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --heads 1 --num-layers 3 --num-hidden 16 --epochs 4 > gpurun_out/gatpack_$np.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
out = ["# TCC_HIT_sum + TCC_MISS_sum per launch (rocprofv3 --pmc, own pass), reddit-shaped graph E = 114.8 M, 3-layer 1-head GAT",
       "# packed = gathered operands in whole-line rows [feat | el], [d_out | er, m, 1/s, t]; separate = MGX_GAT_NO_PACK=1"]
for np, label in ((0, "packed"), (1, "separate")):
    f = glob.glob("gpurun_out/gatpack_%d/**/*counter_collection.csv" % np, recursive=True)
    if not f:
        out.append("%s: no csv" % label); continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "gat_fused_kernel" in k:
            acc.setdefault(k[:60], collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        hit, miss = c.get("TCC_HIT_sum", [0]), c.get("TCC_MISS_sum", [0])
        n = max(len(hit), 1)
        req = (sum(hit) + sum(miss)) / n
        out.append("%-9s %-62s launches %2d  L2 requests / launch %.1f M  (per edge %.2f)  hit rate %.3f" %
                   (label, k, n, req / 1e6, req / 114.8e6, sum(hit) / max(sum(hit) + sum(miss), 1)))
open("gpurun_out/r02_gat_pack_requests.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf gpurun_out/gatpack_0 gpurun_out/gatpack_1
