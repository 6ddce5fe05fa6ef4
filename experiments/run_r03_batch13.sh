cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
MGX_PLAIN_MODEL=1 timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model sage --dataset products --epochs 10 2>&1 | tail -2
export MGX_PLAIN_MODEL=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/plain_prof -o pl -- python3 dgl-0.5-benchmark_amd/full_graph.py --model sage --dataset products --epochs 6 > $O/plain_prof.log 2>&1
python3 experiments/epoch_timeline.py $O/plain_prof > $O/plain_epoch_timeline.txt 2>&1
head -90 $O/plain_epoch_timeline.txt | cut -c1-170
