cd $GRAFT_REPO_ROOT
python3 dgl-0.5-benchmark_amd/kernel_controls.py --graphs products,banded,uniform --widths 64,100 --reps 8
