cd $GRAFT_REPO_ROOT
echo "== eager gcn"; python3 dgl-0.5-benchmark_amd/graph_classification.py --epochs 3 2>&1 | tail -4
echo "== hipgraph gcn"; python3 dgl-0.5-benchmark_amd/graph_classification.py --epochs 4 --hipgraph 2>&1 | tail -6
echo "== eager gin"; python3 dgl-0.5-benchmark_amd/graph_classification.py --epochs 3 --model gin 2>&1 | tail -3
echo "== hipgraph gin"; python3 dgl-0.5-benchmark_amd/graph_classification.py --epochs 4 --model gin --hipgraph 2>&1 | tail -5
