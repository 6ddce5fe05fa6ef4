cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
timeout 900 python -m pytest tests/test_tile_spmm.py tests/test_next_rows_gpu.py -q -m gpu -x --tb=short 2>&1 | tail -6
for mw in 16 48; do
  echo "== MGX_TILE_MIN_WIDTH=$mw"
  MGX_TILE_MIN_WIDTH=$mw timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model sage --dataset reddit --epochs 12 2>&1 | tail -3
done
MGX_SAGE_PROJECT_FIRST=0 timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --model sage --dataset reddit --epochs 12 2>&1 | tail -3
timeout 900 python experiments/exp_tile_kernel.py products --skip-small --lg 2 --widths 16,8 --configs 7x3x1x2,7x4x1x3 2>&1 | grep -v amdgpu | tail -12
timeout 900 python experiments/exp_tile_kernel.py products --skip-small --lg 3 --widths 32 --configs 7x3x1x3 2>&1 | grep -v amdgpu | tail -6
timeout 900 python experiments/exp_tile_kernel.py reddit --skip-small --lg 2 --widths 4,12 --configs 7x3x1x2 2>&1 | grep -v amdgpu | tail -6
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/reddit_sage_prof -o rs -- python3 dgl-0.5-benchmark_amd/full_graph.py --model sage --dataset reddit --epochs 8 > gpurun_out/reddit_sage_prof.log 2>&1
f=$(ls gpurun_out/reddit_sage_prof/*/*kernel_stats.csv 2>/dev/null | tail -1); [ -n "$f" ] && column -s, -t "$f" | cut -c1-200 | sed -n 1,14p
