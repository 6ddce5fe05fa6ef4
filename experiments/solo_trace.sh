#!/bin/bash
# where one rank's solo epoch goes (P = 8 by default): device-busy share, kernels, gaps  ->  gpurun_out/solo_trace.txt
P=${1:-8}; EPOCHS=${2:-100}; SCALE=${3:-1.0}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/solo_trace_raw
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 experiments/solo_trace.py run $OUT/clock.txt $P $EPOCHS $SCALE > $OUT/run.txt 2>&1
tail -2 $OUT/run.txt
python3 experiments/solo_trace.py read $OUT $OUT/clock.txt > gpurun_out/solo_trace_P$P.txt 2>&1
rm -rf $OUT/*/  # the raw trace stays on the box
head -5 gpurun_out/solo_trace_P$P.txt
