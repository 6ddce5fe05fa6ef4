cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
timeout 900 python -m pytest tests/test_gat_fused.py tests/test_tile_spmm.py -q -m gpu -x --tb=short 2>&1 | tail -8
for v in 1 0; do
  echo "== MGX_GAT_AGG_FIRST=$v"
  MGX_GAT_AGG_FIRST=$v timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --heads 1 --num-layers 3 --num-hidden 16 --epochs 12 2>&1 | tail -3
done
