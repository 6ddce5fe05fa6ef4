cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gat_fused.py tests/test_hetero.py -q -m gpu -x --tb=short 2>&1 | tail -4
timeout 600 python dgl-0.5-benchmark_amd/link_prediction.py --data_name ml-1m --train_max_iter 30 2>&1 | tail -2
timeout 600 python dgl-0.5-benchmark_amd/kernel_bench.py --datasets reddit-small --no-spmm --sddmm-binary dot --hidden 16,64,128 2>&1 | grep hidden
