#!/bin/bash
# Counter passes (separate --pmc runs) of the 8-head reddit-small GAT epoch: HBM-side bytes per launch of the fused GAT kernels.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
GAT="python3 $R/dgl-0.5-benchmark_amd/full_graph.py --model gat --dataset reddit-small --heads 8 --num-layers 2 --epochs 4"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_gat8_fetch -- $GAT > $O/${TAG}_gat8_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_gat8_write -- $GAT > $O/${TAG}_gat8_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${TAG}_gat8_l2 -- $GAT > $O/${TAG}_gat8_l2.log 2>&1
python3 - <<PY
import sys
sys.path.insert(0, "$R/profiles")
import summarize
c = {}
for d in ("$O/${TAG}_gat8_fetch", "$O/${TAG}_gat8_write", "$O/${TAG}_gat8_l2"):
    for k, v in summarize.pmc(d).items():
        c.setdefault(k, {}).update(v)
lines = ["# fused GAT kernels, reddit-small + self loops (N = 232,965, E = 11.84 M), 2 layers (8 heads x 16, then 1 head x 41): HBM-side",
         "# bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE KiB x 1024 x 2, WRITE_SIZE KiB x 1024) and L2 hit rate",
         "%-58s %8s %12s %12s %8s %10s" % ("kernel", "launches", "read MB", "write MB", "L2 hit", "L2 req/edge")]
E = 11839883.0
for k in sorted(c):
    if "mgx::gat_" not in k:
        continue
    v = c[k]
    n = len(v.get("FETCH_SIZE", [0]))
    rd = sum(v.get("FETCH_SIZE", [0])) / max(n, 1) * 2048
    wr = sum(v.get("WRITE_SIZE", [0])) / max(len(v.get("WRITE_SIZE", [1])), 1) * 1024
    hit, miss = sum(v.get("TCC_HIT_sum", [0])), sum(v.get("TCC_MISS_sum", [0]))
    nl2 = max(len(v.get("TCC_HIT_sum", [1])), 1)
    lines.append("%-58s %8d %12.1f %12.1f %8.3f %10.2f" % (k.split("(")[0].replace("void ", "")[:58], n, rd / 1e6, wr / 1e6, hit / max(hit + miss, 1),
                                                            (hit + miss) / nl2 / E))
open("$O/${TAG}_gat8_pmc.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
