cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp
MGX_TILE_RING=3 timeout 400 python experiments/exp_tile_kernel.py --small-only > $O/tile_small_ring3.log 2>&1; grep -c " ok" $O/tile_small_ring3.log; grep "FAIL\|PASS\|Error\|error" $O/tile_small_ring3.log | head -10
if grep -q PASS $O/tile_small_ring3.log; then
for r in 2 3; do
  echo "== MGX_TILE_RING=$r"
  MGX_TILE_RING=$r timeout 900 python experiments/exp_tile_kernel.py reddit --skip-small --lg 2 --widths 16,8 --configs 7x3x1x2,7x4x1x3,7x3x1x3 2>&1 | grep "tile D"
  MGX_TILE_RING=$r timeout 900 python experiments/exp_tile_kernel.py proteins --skip-small --lg 2 --widths 16 --configs 7x3x1x2,7x4x1x3 2>&1 | grep "tile D"
done
fi
