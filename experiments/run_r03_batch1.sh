cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python experiments/exp_tile_kernel.py --small-only > $O/tile_small.log 2>&1; tail -1 $O/tile_small.log
timeout 900 python experiments/exp_tile_kernel.py reddit --skip-small --widths 64,128,256 --configs 7x8x1x3,7x10x1x3,14x6x2x3 > $O/r03_tile_reddit.log 2>&1; tail -14 $O/r03_tile_reddit.log
timeout 900 python experiments/exp_tile_kernel.py proteins --skip-small --widths 64,128 --configs 7x8x1x3,14x6x2x3 > $O/r03_tile_proteins.log 2>&1; tail -9 $O/r03_tile_proteins.log
timeout 900 python experiments/exp_tile_kernel.py products --skip-small --widths 64 --configs 7x8x1x3,7x8x1x2 > $O/r03_tile_products.log 2>&1; tail -6 $O/r03_tile_products.log
timeout 1200 python experiments/exp_clustered_control.py > $O/r03_clustered_control.txt 2>&1; tail -14 $O/r03_clustered_control.txt
timeout 1200 python experiments/exp_partition_quality.py --scale 0.25 > $O/r03_partition_quality.txt 2>&1; tail -16 $O/r03_partition_quality.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r03_pytest_gpu.log 2>&1; tail -12 $O/r03_pytest_gpu.log
(timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --dataset reddit --epochs 12 2>&1 | tail -3; MGX_SAGE_PROJECT_FIRST=0 timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --dataset reddit --epochs 12 2>&1 | tail -3; MGX_SAGE_PROJECT_FIRST=0 MGX_TILE=0 timeout 600 python dgl-0.5-benchmark_amd/full_graph.py --dataset reddit --epochs 12 2>&1 | tail -3) > $O/r03_reddit_sage.txt 2>&1; cat $O/r03_reddit_sage.txt
