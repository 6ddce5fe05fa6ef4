cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
timeout 900 python -m pytest tests/test_gat_fused.py -q -m gpu -x --tb=short -k "tile" 2>&1 | tail -6
export MGX_GAT_TILE=1
for d in 0.0; do
  echo "== MGX_GAT_TILE=1 dropout $d"
  timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --dropout $d --epochs 12 2>&1 | tail -1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gat_tile_prof -o gt -- python3 dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --dropout 0.0 --epochs 8 > $O/gat_tile_prof.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/gat_tile_prof/gt_kernel_stats.csv')))
for r in rows:
    if 'gat' in r['Name']:
        print("%-100s calls %5s avg_us %10.1f" % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3))
PY
