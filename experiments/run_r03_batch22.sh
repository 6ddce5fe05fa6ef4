cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
timeout 1200 python -m pytest tests/test_tile_spmm.py tests/test_gat_fused.py -q -m gpu -x --tb=short 2>&1 | tail -8
for d in 0.18 0.0; do
  echo "== reddit GAT, reference defaults, dropout $d"
  timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --dropout $d --epochs 12 2>&1 | tail -1
done
timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset arxiv --epochs 14 2>&1 | tail -1
timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset cora --epochs 30 2>&1 | tail -1
