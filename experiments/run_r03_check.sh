cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --tb=short > gpurun_out/r03_pytest_gpu.log 2>&1; grep "passed\|failed" gpurun_out/r03_pytest_gpu.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
