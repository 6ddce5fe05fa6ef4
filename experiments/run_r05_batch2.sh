#!/bin/bash
# Round 5, second GPU batch: the new kernels' tests (rowpack, sparse halo on the HIP path of emulated ranks), the packed-row gather
# prototype and the scaling model with the sparse halo exchange.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
timeout 900 python3 -m pytest tests/test_rowpack.py tests/test_emulated_ranks.py tests/test_dist.py tests/test_accelerate_linear.py tests/test_torch_ops.py -m gpu -x -q > $O/r05_b2_pytest.log 2>&1; echo "pytest rc $?"
tail -4 $O/r05_b2_pytest.log
timeout 400 python3 experiments/exp_packed_rows.py > $O/r05_sparse_row_bound.txt 2>&1; echo "packed rc $?"
timeout 900 python3 bench.py --emulate-ranks 2,4,8 --steps 10 --warmup 3 --report $O/r05_scale_model.txt > $O/r05_b2_emu_line.json 2> $O/r05_b2_emu_line.err; echo "emu rc $?"
MGX_SPARSE_HALO=0 timeout 900 python3 bench.py --emulate-ranks 8 --steps 10 --warmup 3 --report $O/r05_scale_model_dense_halo.txt > $O/r05_b2_emu_dense_line.json 2> $O/r05_b2_emu_dense_line.err; echo "emu dense rc $?"
tail -30 $O/r05_sparse_row_bound.txt
