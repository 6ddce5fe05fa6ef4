cd /tmp && export TMPDIR=/tmp
export MGX_SAGE_L1_PROJECT_FIRST=1
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pf_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_trace -- python3 $R/bench.py --no-cpu-baseline --no-pmc --no-controls --no-plain --no-scale-model --steps 5 --warmup 2 > $R/gpurun_out/pf_line.txt 2>&1
python3 $R/experiments/epoch_timeline.py $R/gpurun_out/pf_trace > $R/gpurun_out/pf_timeline.txt 2>&1
awk '{ if ($4+0 > 150) print }' $R/gpurun_out/pf_timeline.txt | cut -c1-200 | head -40
