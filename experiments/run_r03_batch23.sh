cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
for o in generated dst src; do
  echo "# ---- g-SDDMM add, reddit (E = 114.6 M), edge list order: $o"
  timeout 900 python dgl-0.5-benchmark_amd/kernel_bench.py --datasets reddit --no-spmm --hidden 16,64,128 --edge-order $o 2>&1 | grep "hidden\|sorted"
done
for o in generated dst; do
  echo "# ---- g-SDDMM add, reddit-small (E = 11.6 M), edge list order: $o"
  timeout 900 python dgl-0.5-benchmark_amd/kernel_bench.py --datasets reddit-small --no-spmm --hidden 16,64,128 --edge-order $o 2>&1 | grep "hidden\|sorted"
done
