#!/bin/bash
# kernel_bench.py (counterpart of kernel/dgl-new.py), ONE PROCESS PER DATASET (see the note at the head of the output), for
# the reference's three datasets at BASELINE's reddit size (11.6 M edges) and at the dataset's own size (114.6 M).
#   gpurun -- 'bash profiles/collect_kernel_bench.sh r02'
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
export PYTHONPATH=$R/dgl-0.5-benchmark_amd
OUT=$O/${TAG}_kernel_bench.txt
cat > $OUT <<'HDR'
# dgl-0.5-benchmark_amd/kernel_bench.py (counterpart of kernel/dgl-new.py), one process per dataset: in ONE process the proteins D = 128 g-SpMM
# measured 6.4-6.9 ms after the reddit-small and arxiv sweeps against 2.07 ms on its own (same kernel and sizes; not an allocator-cache effect:
# empty_cache() before every width changes nothing) -- observed, not understood; the per-dataset numbers below are reproducible.
# Datasets: reddit-small (BASELINE config 2's 11.6 M edges), arxiv, proteins, reddit (the dataset's own 114.6 M edges, what kernel/dgl-new.py:61 loads).
HDR
for ds in reddit-small arxiv proteins reddit; do
  python3 $R/dgl-0.5-benchmark_amd/kernel_bench.py --datasets $ds 2>&1 | grep -v "amdgpu.ids" >> $OUT
done
tail -3 $OUT
