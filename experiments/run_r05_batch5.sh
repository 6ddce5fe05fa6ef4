#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
timeout 300 python3 experiments/exp_rowpack.py > $O/r05_rowpack.txt 2>&1; echo "rowpack rc $?"
cat $O/r05_rowpack.txt
timeout 300 python3 dgl-0.5-benchmark_amd/kernel_bench.py --datasets reddit-small --hidden 32,48 --no-spmm > $O/r05_sddmm_32.txt 2>&1
MGX_SDDMM_WALK=coo timeout 300 python3 dgl-0.5-benchmark_amd/kernel_bench.py --datasets reddit-small --hidden 32,48 --no-spmm >> $O/r05_sddmm_32.txt 2>&1
grep "hidden size" $O/r05_sddmm_32.txt
