#!/bin/bash
# Round 5, first GPU batch: the suite, the default bench line (now with `secondary` + reference-module keys), the two bounded kernel
# experiments and the kernel timeline of the reference's unmodified module graph with MGX_ACCELERATE_LINEAR=1.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/r05_b1_pytest.log 2>&1; echo "pytest rc $?" >> $O/r05_b1_pytest.log
tail -3 $O/r05_b1_pytest.log
timeout 600 python3 bench.py --steps 10 --warmup 3 > $O/r05_b1_bench_line.json 2> $O/r05_b1_bench_line.err; echo "bench rc $?"
timeout 300 python3 experiments/exp_d100_split.py > $O/r05_d100_split.txt 2>&1; echo "d100 rc $?"
timeout 400 python3 experiments/exp_packed_rows.py > $O/r05_sparse_row_bound.txt 2>&1; echo "packed rc $?"
cd /tmp && export TMPDIR=/tmp
export MGX_PLAIN_MODEL=1 MGX_ACCELERATE_LINEAR=1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05_plain_trace -- python3 $R/dgl-0.5-benchmark_amd/full_graph.py --model sage --dataset products --epochs 8 > $O/r05_plain_trace.log 2>&1
unset MGX_PLAIN_MODEL MGX_ACCELERATE_LINEAR
python3 $R/experiments/epoch_timeline.py $O/r05_plain_trace > $O/r05_plain_epoch_timeline.txt 2>&1
rm -rf $O/r05_plain_trace
tail -c 1500 $O/r05_b1_bench_line.json
