#!/bin/bash
# Round 5, fourth GPU batch: the whole suite after the prune, g-SDDMM in CSR order (fixed), the default bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
timeout 1200 python3 -m pytest tests -m gpu -x -q > $O/r05_b4_pytest.log 2>&1; echo "pytest rc $?"
grep -n "passed\|failed" $O/r05_b4_pytest.log | tail -3
timeout 600 python3 experiments/exp_sddmm_perm.py > $O/r05_sddmm_perm.txt 2>&1; echo "sddmm rc $?"
cat $O/r05_sddmm_perm.txt
timeout 900 python3 bench.py --steps 10 --warmup 3 > $O/r05_b4_bench_line.json 2> $O/r05_b4_bench_line.err; echo "bench rc $?"
tail -c 400 $O/r05_b4_bench_line.json
