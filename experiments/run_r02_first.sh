set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest1.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r02_pytest1.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r02_bench1.json 2> gpurun_out/r02_bench1.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r02_bench1.json
tail -5 gpurun_out/r02_bench1.err
python experiments/exp_partition_time.py > gpurun_out/r02_partition_time.txt 2>&1; echo "part rc=$?"
cat gpurun_out/r02_partition_time.txt | tail -5
MGX_BENCH_SHARE_GPU=1 MGX_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r02_bench_share2.json 2> gpurun_out/r02_bench_share2.err; echo "share2 rc=$?"
tail -c 2500 gpurun_out/r02_bench_share2.json; tail -5 gpurun_out/r02_bench_share2.err
