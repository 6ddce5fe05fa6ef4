cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rs_prof -o rs -- python3 dgl-0.5-benchmark_amd/full_graph.py --model sage --dataset reddit --epochs 8 > $O/rs_prof.log 2>&1
python3 experiments/epoch_timeline.py $O/rs_prof 2>&1 | head -60 | cut -c1-170
rm -rf $O/rs_prof
