cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_tile_spmm.py tests/test_gat_fused.py tests/test_gpu_fullsize.py -q -m gpu -x --tb=short > gpurun_out/check2.log 2>&1; grep "passed\|failed" gpurun_out/check2.log | tail -2
