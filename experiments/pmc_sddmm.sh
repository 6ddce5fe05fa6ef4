#!/bin/bash
# HBM traffic of the COO g-SDDMM (u_add_v) on the reddit-shaped graph: FETCH_SIZE / WRITE_SIZE per launch, separate passes
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/sddmm_$c -- python3 dgl-0.5-benchmark_amd/kernel_bench.py --datasets ${1:-reddit} > gpurun_out/sddmm_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/sddmm_%s/**/*counter_collection.csv" % c, recursive=True)
    if not f:
        print(c, "no csv"); continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "sddmm" in k or "spmm_rowwave32" in k:
            acc.setdefault(k[:70], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        # gfx950: FETCH_SIZE in KiB x 2 (guide), WRITE_SIZE in KiB
        scale = 2048.0 if c == "FETCH_SIZE" else 1024.0
        print(c, k, "launches", len(v), "GB/launch min %.2f median %.2f max %.2f" % (min(v) * scale / 1e9, sorted(v)[len(v)//2] * scale / 1e9, max(v) * scale / 1e9))
PY
