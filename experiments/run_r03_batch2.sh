cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python experiments/exp_clustered_control.py --clustering-only --graphs clustered,products,mixing1 > $O/r03_clustering_full.txt 2>&1; cat $O/r03_clustering_full.txt | grep -v amdgpu
timeout 1800 python -m pytest tests -q -m gpu > $O/r03_pytest_gpu.log 2>&1; tail -6 $O/r03_pytest_gpu.log
bash profiles/collect.sh r03 > $O/r03_collect.log 2>&1; tail -3 $O/r03_collect.log
bash profiles/collect_kernel_bench.sh r03 > $O/r03_collect_kb.log 2>&1; tail -2 $O/r03_collect_kb.log
bash profiles/collect_controls.sh r03 > $O/r03_collect_controls.log 2>&1; tail -12 $O/r03_collect_controls.log
