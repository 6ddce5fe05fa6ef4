cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python experiments/exp_tile_kernel.py --small-only > $O/tile_small.log 2>&1; tail -1 $O/tile_small.log
for lpt in 1 0; do
  echo "== MGX_TILE_LPT=$lpt"; MGX_TILE_LPT=$lpt timeout 900 python experiments/exp_tile_kernel.py reddit --skip-small --widths 64,128 --configs 7x8x1x3,14x6x2x3 2>&1 | grep -v amdgpu | tail -6
done
for sp in 1024 4096; do
  echo "== split $sp"; timeout 900 python experiments/exp_tile_kernel.py reddit --skip-small --widths 64 --split $sp --configs 7x8x1x3 2>&1 | grep -v amdgpu | tail -3
done
timeout 900 python experiments/exp_clustered_control.py --graphs clustered > $O/r03_clustered_control2.txt 2>&1; cat $O/r03_clustered_control2.txt | grep -v amdgpu
timeout 900 python -m pytest tests/test_graphed_batches.py tests/test_next_rows_gpu.py tests/test_tile_spmm.py -q -m gpu 2>&1 | tail -4
