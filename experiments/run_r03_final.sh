cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --tb=short > $O/r03_pytest_gpu.log 2>&1; tail -4 $O/r03_pytest_gpu.log | head -2
bash profiles/collect.sh r03 2>&1 | tail -2
cd $GRAFT_REPO_ROOT
bash profiles/collect_kernel_bench.sh r03 2>&1 | tail -1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
