cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ns_prof -o ns -- python3 dgl-0.5-benchmark_amd/sampling_sage.py --num-epochs 3 > $O/ns_prof.log 2>&1
tail -3 $O/ns_prof.log
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/ns_prof/ns_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms %.1f" % (tot/1e6))
for r in rows[:28]:
    print("%-110s calls %6s avg_us %9.1f  %5.1f%%" % (r['Name'][:110], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
