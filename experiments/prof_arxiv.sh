cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_arxiv_trace
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/dgl-0.5-benchmark_amd/full_graph.py --model sage --dataset arxiv --epochs 12 > $O/log.txt 2>&1
tail -2 $O/log.txt
python3 - <<'PY'
import csv, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_arxiv_trace"
f = sorted(glob.glob(out + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:28]:
    print("%6d %9.3f ms %8.1f us  %s" % (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
