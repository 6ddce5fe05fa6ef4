#!/bin/bash
# Round 5, third GPU batch: the whole suite, the packed-row gather prototype, g-SDDMM in CSR order, the scaling model after the fused
# unpack-add.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
timeout 1200 python3 -m pytest tests -m gpu -x -q > $O/r05_b3_pytest.log 2>&1; echo "pytest rc $?"
grep -n "passed\|failed" $O/r05_b3_pytest.log | tail -3
timeout 400 python3 experiments/exp_packed_rows.py > $O/r05_sparse_row_bound.txt 2>&1; echo "packed rc $?"
timeout 600 python3 experiments/exp_sddmm_perm.py > $O/r05_sddmm_perm.txt 2>&1; echo "sddmm rc $?"
timeout 900 python3 bench.py --emulate-ranks 2,4,8 --steps 10 --warmup 3 --report $O/r05_scale_model.txt > $O/r05_b3_emu_line.json 2> $O/r05_b3_emu_line.err; echo "emu rc $?"
tail -25 $O/r05_sparse_row_bound.txt; cat $O/r05_sddmm_perm.txt; grep -n "64 GB/s" $O/r05_scale_model.txt
