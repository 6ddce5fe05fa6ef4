cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
export MGX_LIB_PATH=$GRAFT_REPO_ROOT/experiments/tile_spmm/libmgx_stamps.so
timeout 600 python experiments/exp_tile_stamps.py reddit 7x3x1x2 16 2 2>&1 | grep -v amdgpu
timeout 600 python experiments/exp_tile_stamps.py reddit 7x3x1x3 32 3 2>&1 | grep -v amdgpu
timeout 600 python experiments/exp_tile_stamps.py reddit 7x6x1x3 64 4 2>&1 | grep -v amdgpu
timeout 600 python experiments/exp_tile_stamps.py proteins 7x3x1x2 16 2 2>&1 | grep -v amdgpu
unset MGX_LIB_PATH
timeout 900 python -m pytest tests/test_gat_fused.py -q -m gpu -x --tb=short 2>&1 | tail -3
