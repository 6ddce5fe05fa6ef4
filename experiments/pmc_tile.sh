#!/bin/bash
# SQ counters of the tile kernel on the reddit shape (separate --pmc passes; never combined with tracing)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
CFG=${1:-14x6x2x3}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/pmc_list.txt 2>&1
CMD="python3 $R/experiments/exp_tile_kernel.py reddit --skip-small --widths 64 --configs $CFG"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_tile_a -- $CMD > $O/pmc_tile_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_tile_b -- $CMD > $O/pmc_tile_b.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_tile_c -- $CMD > $O/pmc_tile_c.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_tile_d -- $CMD > $O/pmc_tile_d.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in "abcd":
    fs = glob.glob("$O/pmc_tile_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        print("pass", tag, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "spmm_tile_kernel" in k or "spmm_rowwave32" in k:
            agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(tag, k, {n: "%.4g" % (sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
