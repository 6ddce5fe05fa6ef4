cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
timeout 900 python -m pytest tests/test_gat_fused.py tests/test_models_gpu.py tests/test_gpu_parity.py -q -m gpu -x --tb=short 2>&1 | tail -3
timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit-small --heads 8 --num-layers 2 --epochs 20 2>&1 | tail -1
timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset arxiv --epochs 20 2>&1 | tail -1
timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset cora --epochs 30 2>&1 | tail -1
MGX_GAT_TILE=0 timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --epochs 12 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gat8_prof -o g8 -- python3 dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit-small --heads 8 --num-layers 2 --epochs 8 > $O/gat8_prof.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/gat8_prof/g8_kernel_stats.csv')))
for r in rows:
    if 'gat' in r['Name']:
        print("%-100s calls %5s avg_us %10.1f" % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3))
PY
