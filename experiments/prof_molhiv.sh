cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_molhiv_trace -- python3 $R/dgl-0.5-benchmark_amd/graph_classification.py --epochs 2 --num_graphs 8192 > $R/gpurun_out/r02_molhiv_trace.log 2>&1
f=$(find $R/gpurun_out/r02_molhiv_trace -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6, "kernels", sum(int(r["Calls"]) for r in rows))
for r in rows[:25]:
    print("%-110s %7s %10.3f %9.2f %6s" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
tail -3 $R/gpurun_out/r02_molhiv_trace.log
