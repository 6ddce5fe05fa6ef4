cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
for b in 1 0; do
  echo "== MGX_TILE_BANK_ORDER=$b"
  MGX_TILE_BANK_ORDER=$b timeout 900 python experiments/exp_tile_kernel.py reddit --skip-small --lg 2 --widths 16,8 --configs 7x3x1x2 2>&1 | grep "tile D"
  MGX_TILE_BANK_ORDER=$b timeout 900 python experiments/exp_tile_kernel.py reddit --skip-small --lg 3 --widths 32 --configs 7x3x1x3 2>&1 | grep "tile D"
  MGX_TILE_BANK_ORDER=$b timeout 900 python experiments/exp_tile_kernel.py proteins --skip-small --lg 2 --widths 16 --configs 7x3x1x2 2>&1 | grep "tile D"
done
bash experiments/pmc_tile_narrow.sh 2>&1 | grep "^b \|^c " | cut -c1-330
