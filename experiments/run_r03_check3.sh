cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gat_fused.py -q -m gpu -x --tb=short > gpurun_out/check3.log 2>&1; grep "passed\|failed\|Error" gpurun_out/check3.log | tail -4
