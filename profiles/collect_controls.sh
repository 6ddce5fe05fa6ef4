#!/bin/bash
# Counter passes of the dominant kernel on the benchmark graph and the control graphs (dgl-0.5-benchmark_amd/kernel_controls.py),
# one rocprofv3 --pmc pass per counter group (never combined with tracing), then profiles/summarize_controls.py.
#   gpurun -- 'bash profiles/collect_controls.sh r02'
set -u
TAG=${1:-r02}
WIDTHS=${2:-64}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_controls
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 $R/dgl-0.5-benchmark_amd/kernel_controls.py --widths $WIDTHS --reps 6 > $O/timing.json 2> $O/timing.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/dgl-0.5-benchmark_amd/kernel_controls.py --widths $WIDTHS --reps 3 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/dgl-0.5-benchmark_amd/kernel_controls.py --widths $WIDTHS --reps 3 > $O/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2 -- python3 $R/dgl-0.5-benchmark_amd/kernel_controls.py --widths $WIDTHS --reps 3 > $O/l2.log 2>&1
python3 $R/profiles/summarize_controls.py $TAG $O $WIDTHS | tee $O/summary.txt
