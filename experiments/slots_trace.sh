#!/bin/bash
# kernel stats of the headline epoch with the slot form on (experiments: where the forward D = 64 aggregation's time goes now)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/slots_trace
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-pmc --no-controls --no-plain --no-scale-model --no-secondary --no-variants --steps 5 --warmup 2 > $OUT/line.json 2> $OUT/err.txt
python3 experiments/epoch_timeline.py $OUT > gpurun_out/slots_epoch_timeline.txt 2>&1
rm -rf $OUT/*/
head -75 gpurun_out/slots_epoch_timeline.txt | cut -c1-150
