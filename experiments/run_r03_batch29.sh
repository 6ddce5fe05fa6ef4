cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
timeout 900 python dgl-0.5-benchmark_amd/graph_classification.py --model gin --epochs 4 --hipgraph 2>&1 | tail -3
timeout 900 python dgl-0.5-benchmark_amd/graph_classification.py --model gin --epochs 3 2>&1 | tail -2
