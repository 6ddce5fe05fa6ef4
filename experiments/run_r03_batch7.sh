cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python experiments/exp_tile_kernel.py reddit --skip-small --lg 4 --widths 64,128 --configs 7x6x1x3,7x5x1x3,7x4x1x3,7x6x1x2,7x5x1x2 2>&1 | grep -v amdgpu | tail -18
timeout 900 python experiments/exp_tile_kernel.py reddit --skip-small --lg 3 --widths 32,64 --configs 7x3x1x2,7x5x1x2,7x3x1x3 2>&1 | grep -v amdgpu | tail -12
timeout 900 python experiments/exp_tile_kernel.py proteins --skip-small --lg 4 --widths 64,128 --configs 7x8x1x3,7x6x1x3,7x5x1x3,7x4x1x3 2>&1 | grep -v amdgpu | tail -14
timeout 900 python experiments/exp_tile_kernel.py proteins --skip-small --lg 2 --widths 16 --configs 7x3x1x2,7x4x1x2 2>&1 | grep -v amdgpu | tail -6
timeout 900 python experiments/exp_tile_kernel.py proteins --skip-small --lg 3 --widths 32 --configs 7x3x1x2,7x5x1x3,7x6x1x2 2>&1 | grep -v amdgpu | tail -8
