#!/bin/bash
# reddit-shaped 2-layer SAGE at the reference's own size (E = 114.6 M, D = 602 -> 16 -> 41): epoch + per-kernel times
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
mkdir -p gpurun_out
python dgl-0.5-benchmark_amd/full_graph.py --model sage --dataset reddit --epochs 12 > gpurun_out/reddit_sage.log 2>&1
tail -4 gpurun_out/reddit_sage.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/reddit_sage_prof -o rs -- python3 dgl-0.5-benchmark_amd/full_graph.py --model sage --dataset reddit --epochs 8 > gpurun_out/reddit_sage_prof.log 2>&1
python profiles/summarize.py gpurun_out/reddit_sage_prof 2>/dev/null | head -30 > gpurun_out/reddit_sage_stats.txt || true
ls gpurun_out/reddit_sage_prof/* | tail -5
f=$(ls gpurun_out/reddit_sage_prof/*/*kernel_stats.csv 2>/dev/null | tail -1); [ -n "$f" ] && column -s, -t "$f" | cut -c1-220 | sed -n 1,16p
