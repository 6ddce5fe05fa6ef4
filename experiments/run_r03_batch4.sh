cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gat_fused.py -q -m gpu -x --tb=short 2>&1 | tail -15
for v in 0 1; do
  if [ $v = 1 ]; then export MGX_GAT_EL_GATHER=1; else unset MGX_GAT_EL_GATHER; fi
  echo "== MGX_GAT_EL_GATHER=${MGX_GAT_EL_GATHER:-unset}"
  timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit-small --heads 8 --num-layers 2 --epochs 14 2>&1 | tail -2
  timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset arxiv --heads 4 --num-layers 3 --epochs 14 2>&1 | tail -2
done
