"""Print the kernels of the LAST bench epoch in launch order with start offsets, durations and idle gaps.

Usage (on the GPU box):  rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline
                         python experiments/epoch_timeline.py DIR > gpurun_out/timeline.txt
The epoch boundary is found from the optimizer's multi-tensor kernel (one Adam step per epoch)."""
import csv
import glob
import os
import sys

d = sys.argv[1]
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
# group consecutive optimizer kernels: an epoch ends at the last kernel of each group
ends = [m for j, m in enumerate(marks) if j + 1 == len(marks) or marks[j + 1] - m > 8]
if len(ends) < 2:
    sys.exit("need at least two epochs in the trace")
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
agg = {}
print("# last epoch: %d kernels" % (b - a))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    short = name[:110]
    print("%9.3f ms  dur %8.1f us  gap %7.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, (s - prev_end) / 1e3, short))
    busy += e - s
    prev_end = max(prev_end, e)
    k = name[:70]
    agg[k] = agg.get(k, 0) + (e - s)
print("# epoch span %.3f ms, busy %.3f ms" % ((prev_end - t0) / 1e6, busy / 1e6))
print("# by kernel:")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:25]:
    print("%9.3f ms  %s" % (v / 1e6, k))
