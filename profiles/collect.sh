#!/bin/bash
# Commands that produced the summaries in this directory (run on the MI355X box through gpurun, from the
# repo root; rocprofv3 needs a writable cwd/TMPDIR).  Counters are collected in their own passes
# (--pmc only, never combined with tracing), as the pool requires.
#   gpurun -- 'bash profiles/collect.sh r01'   then   python profiles/summarize.py r01 gpurun_out/r01_bench_trace ...
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_bench_trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_trace.log 2>&1
python $R/experiments/epoch_timeline.py $O/${TAG}_bench_trace > $O/${TAG}_epoch_timeline.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_bench_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_bench_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${TAG}_bench_l2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_l2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_gat_trace -- python $R/dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --heads 1 --num-layers 3 --num-hidden 16 --epochs 6 > $O/${TAG}_gat_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_gat8_trace -- python $R/dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit-small --heads 8 --num-layers 2 --epochs 6 > $O/${TAG}_gat8_trace.log 2>&1
python $R/bench.py --steps 10 --warmup 3 > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench_line.err
tail -c 600 $O/${TAG}_bench_line.json
