#!/bin/bash
# kernel_bench.py (counterpart of kernel/dgl-new.py) for the reference's three datasets at BASELINE's reddit size (11.6 M edges)
# and at the dataset's own size (114.6 M):
#   (a) ONE process for reddit-small,arxiv,proteins -- the reference's own invocation (kernel/dgl-new.py:61), with the GPU clocks
#       sampled before / after (round 2 saw proteins D = 128 at 3x its stand-alone time inside such a sweep);
#   (b) one process per dataset;
#   (c) the dense graphs (proteins, reddit) again with MGX_TILE=0: the row-per-wave kernel the LDS-staged tile kernel replaces.
#   gpurun -- 'bash profiles/collect_kernel_bench.sh r03'
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
export PYTHONPATH=$R/dgl-0.5-benchmark_amd
OUT=$O/${TAG}_kernel_bench.txt
clk() { /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -2 | tr '\n' ' '; echo; }
{
echo "# dgl-0.5-benchmark_amd/kernel_bench.py (counterpart of kernel/dgl-new.py); [min max] of the 8 timed repetitions behind every mean"
echo "# ---- (a) one process, the reference's dataset list; clocks before: $(clk)"
python3 $R/dgl-0.5-benchmark_amd/kernel_bench.py --datasets reddit-small,arxiv,proteins 2>&1 | grep -v "amdgpu.ids"
echo "# clocks after: $(clk)"
for ds in reddit-small arxiv proteins reddit; do
  echo "# ---- (b) own process: $ds"
  python3 $R/dgl-0.5-benchmark_amd/kernel_bench.py --datasets $ds 2>&1 | grep -v "amdgpu.ids"
done
for ds in proteins reddit; do
  echo "# ---- (c) own process, MGX_TILE=0 (row-per-wave kernel): $ds"
  MGX_TILE=0 python3 $R/dgl-0.5-benchmark_amd/kernel_bench.py --datasets $ds --no-sddmm 2>&1 | grep -v "amdgpu.ids"
done
} > $OUT
tail -3 $OUT
