cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
for v in 1 0; do
  echo "== MGX_GAT_TILE=$v"
  MGX_GAT_TILE=$v timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --heads 1 --num-layers 3 --num-hidden 16 --epochs 12 2>&1 | tail -2
done
for cfg in 7x4x1x2 7x3x1x3 7x4x1x3; do
  echo "== MGX_GAT_TILE_CFG=$cfg"
  MGX_GAT_TILE_CFG=$cfg timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --heads 1 --num-layers 3 --num-hidden 16 --epochs 12 2>&1 | tail -1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gat_tile_prof -o gt -- python3 dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --heads 1 --num-layers 3 --num-hidden 16 --epochs 8 > $O/gat_tile_prof.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/gat_tile_prof/gt_kernel_stats.csv')))
for r in rows:
    if 'gat' in r['Name'] or 'mgx' in r['Name']:
        print("%-100s calls %5s avg_us %10.1f" % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3))
PY
