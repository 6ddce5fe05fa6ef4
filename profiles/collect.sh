#!/bin/bash
# Commands that produced the summaries in this directory (run on the MI355X box through gpurun, from the
# repo root; rocprofv3 needs a writable cwd/TMPDIR).  Counters are collected in their own passes
# (--pmc only, never combined with tracing), as the pool requires.
#   gpurun -- 'bash profiles/collect.sh r05'   then the summaries are written by profiles/summarize.py (called below)
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-pmc --no-controls --no-plain --no-scale-model --no-secondary --no-variants"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_bench_trace -- $BENCH --steps 5 --warmup 2 > $O/${TAG}_bench_trace.log 2>&1
python3 $R/experiments/epoch_timeline.py $O/${TAG}_bench_trace > $O/${TAG}_epoch_timeline.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_bench_fetch -- $BENCH --steps 3 --warmup 1 > $O/${TAG}_bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_bench_write -- $BENCH --steps 3 --warmup 1 > $O/${TAG}_bench_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${TAG}_bench_l2 -- $BENCH --steps 3 --warmup 1 > $O/${TAG}_bench_l2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_gat_trace -- python3 $R/dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit --epochs 6 > $O/${TAG}_gat_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_gat8_trace -- python3 $R/dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit-small --heads 8 --num-layers 2 --epochs 6 > $O/${TAG}_gat8_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_molhiv_trace -- python3 $R/dgl-0.5-benchmark_amd/graph_classification.py --epochs 2 --num_graphs 8192 > $O/${TAG}_molhiv_trace.log 2>&1
python3 $R/profiles/summarize.py $TAG $O/${TAG}_bench_trace $O/${TAG}_bench_fetch $O/${TAG}_bench_write $O/${TAG}_bench_l2 > $O/${TAG}_summarize.log 2>&1
python3 $R/profiles/summarize.py ${TAG}_gat_reddit $O/${TAG}_gat_trace >> $O/${TAG}_summarize.log 2>&1
python3 $R/profiles/summarize.py ${TAG}_gat8_redditsmall $O/${TAG}_gat8_trace >> $O/${TAG}_summarize.log 2>&1
python3 $R/profiles/summarize.py ${TAG}_molhiv $O/${TAG}_molhiv_trace >> $O/${TAG}_summarize.log 2>&1
# only what summarize.py wrote into profiles/ on this box (a blanket copy would overwrite the files generated straight into $O)
cp $R/profiles/${TAG}_*kernel_stats.txt $R/profiles/${TAG}_pmc_summary.txt $O/ 2>/dev/null
(python3 $R/profiles/gat_roofline.py $O/${TAG}_gat8_trace 232965 11839883 8 16; python3 $R/profiles/gat_roofline.py $O/${TAG}_gat8_trace 232965 11839883 1 41) > $O/${TAG}_gat8_roofline.txt 2>&1
python3 $R/dgl-0.5-benchmark_amd/generate_result.py --out $O/${TAG}_generate_result.csv > $O/${TAG}_generate_result.txt 2>&1
python3 $R/bench.py --emulate-ranks 2,4,8 --steps 10 --warmup 3 --report $O/${TAG}_scale_model.txt > $O/${TAG}_emu_bench_line.json 2> $O/${TAG}_emu_bench_line.err
python3 $R/bench.py --steps 10 --warmup 3 > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench_line.err
# one rank's solo epochs under the kernel trace (device-busy share, kernels, gaps) and the Python side of the same epochs
cd $R
bash experiments/solo_trace.sh 8 100 > /dev/null 2>&1; cp $O/solo_trace_P8.txt $O/${TAG}_solo_trace_P8.txt
bash experiments/solo_trace.sh 2 40 > /dev/null 2>&1; cp $O/solo_trace_P2.txt $O/${TAG}_solo_trace_P2.txt
python3 experiments/prof_rank_host.py 1.0 8 10 > $O/${TAG}_rank_host_profile.txt 2>&1
python3 experiments/exp_rowpack.py > $O/${TAG}_rowpack.txt 2>/dev/null
tail -c 600 $O/${TAG}_bench_line.json
