cd $GRAFT_REPO_ROOT
for f in 0 1; do
  echo "== MGX_GAT_FUSED=$f reddit-small 8 heads 2 layers"
  MGX_GAT_FUSED=$f python3 dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit-small --heads 8 --num-layers 2 --epochs 12 2>&1 | grep -E "Training time|epoch 11"
  echo "== MGX_GAT_FUSED=$f reddit 1 head 3 layers hidden 16"
  MGX_GAT_FUSED=$f python3 dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --heads 1 --num-layers 3 --num-hidden 16 --epochs 12 2>&1 | grep -E "Training time|epoch 11"
  echo "== MGX_GAT_FUSED=$f arxiv 4 heads 3 layers"
  MGX_GAT_FUSED=$f python3 dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset arxiv --heads 4 --num-layers 3 --num-hidden 16 --epochs 12 2>&1 | grep -E "Training time|epoch 11"
done
