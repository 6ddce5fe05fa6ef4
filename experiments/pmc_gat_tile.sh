#!/bin/bash
# Counters of the fused GAT walks on the reddit shape: tile kernels (default) against the row kernels (MGX_GAT_TILE=0); separate --pmc passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R/dgl-0.5-benchmark_amd
for v in 1 0; do
  export MGX_GAT_TILE=$v
  CMD="python3 $R/dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --epochs 4"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_gt_c$v -- $CMD > $O/pmc_gt_c$v.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $O/pmc_gt_b$v -- $CMD > $O/pmc_gt_b$v.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for tag in ("c1", "b1", "c0", "b0"):
    fs = glob.glob("$O/pmc_gt_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        print("pass", tag, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "gat_tile_kernel" in k or "gat_fused_kernel" in k:
            agg[k[:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(agg.items()):
        print(tag, k, {n: "%.4g" % (sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
rm -rf $O/pmc_gt_c1 $O/pmc_gt_b1 $O/pmc_gt_c0 $O/pmc_gt_b0
