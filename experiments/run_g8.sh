cd $GRAFT_REPO_ROOT
for g in 0 8; do
  echo "== MGX_SPMM_G=$g"
  MGX_SPMM_G=$g python3 dgl-0.5-benchmark_amd/kernel_controls.py --graphs products,banded,uniform --widths 64,100 --reps 6
done
