cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
echo "== shipped"
timeout 600 python experiments/exp_wgrad_shapes.py 2>&1 | grep -v amdgpu | head -16
echo "== software-pipelined operands"
MGX_LIB_PATH=$GRAFT_REPO_ROOT/experiments/tile_spmm/libmgx_xtypipe.so timeout 600 python experiments/exp_wgrad_shapes.py 2>&1 | grep -v amdgpu | head -16
