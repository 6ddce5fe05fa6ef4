cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
timeout 900 python -m pytest tests/test_gat_fused.py -q -m gpu -x --tb=short -k "tile_walks" 2>&1 | tail -15
