#!/bin/bash
# L2 / LDS / VALU counters of the 16-column tile kernel against the row kernel on the reddit shape, D = 16 (separate --pmc passes;
# never combined with tracing).  gpurun -- 'bash experiments/pmc_tile_narrow.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/experiments/exp_tile_kernel.py reddit --skip-small --lg 2 --widths 16 --configs 7x3x1x2"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc_tn_b -- $CMD > $O/pmc_tn_b.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_tn_c -- $CMD > $O/pmc_tn_c.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_tn_d -- $CMD > $O/pmc_tn_d.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in "bcd":
    fs = glob.glob("$O/pmc_tn_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        print("pass", tag, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "spmm_tile_narrow_kernel" in k or "spmm_rowwave32" in k:
            agg[k[:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(tag, k, {n: "%.4g" % (sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
rm -rf $O/pmc_tn_b $O/pmc_tn_c $O/pmc_tn_d
