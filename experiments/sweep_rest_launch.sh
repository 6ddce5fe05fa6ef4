export PYTHONPATH=$GRAFT_REPO_ROOT/dgl-0.5-benchmark_amd
for r in 8 16 32 64; do echo "MGX_ROWS_PER_BLOCK=$r"; MGX_ROWS_PER_BLOCK=$r python experiments/exp_two_part_products.py 2>&1 | grep -E "D  64 (accumulate|mean) +one launch"; done
for sp in 128 512; do echo "MGX_SPLIT=$sp"; MGX_SPLIT=$sp python experiments/exp_two_part_products.py 2>&1 | grep -E "D  64 (accumulate|mean) +one launch"; done
