cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 400 python experiments/exp_tile_kernel.py --small-only > $O/tile_small.log 2>&1; grep -c " ok" $O/tile_small.log; grep "FAIL\|PASS\|Error\|error" $O/tile_small.log | head -10
if grep -q PASS $O/tile_small.log; then
  timeout 900 python experiments/exp_tile_kernel.py reddit --skip-small --lg 3 --widths 32,24 --configs 7x8x1x3,7x8x1x2,7x6x1x3 2>&1 | grep -v amdgpu | tail -12
  timeout 900 python experiments/exp_tile_kernel.py reddit --skip-small --lg 2 --widths 16,8 --configs 7x8x1x3,7x8x1x2,7x4x1x3 2>&1 | grep -v amdgpu | tail -12
  timeout 900 python experiments/exp_tile_kernel.py proteins --skip-small --lg 2 --widths 16 --configs 7x8x1x3 2>&1 | grep -v amdgpu | tail -4
fi
