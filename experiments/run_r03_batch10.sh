cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x --tb=short > $O/r03_pytest_gpu.log 2>&1; tail -4 $O/r03_pytest_gpu.log
bash profiles/collect_kernel_bench.sh r03 2>&1 | tail -3
grep -n "hidden size: \(1\|2\|4\|8\|16\|32\)," $O/r03_kernel_bench.txt | sed -n 1,60p | cut -c1-150
