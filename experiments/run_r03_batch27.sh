cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
timeout 600 python experiments/exp_gat_plan_stats.py reddit 2>&1 | grep -v amdgpu | tail -4
