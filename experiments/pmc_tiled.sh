cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/t_fetch -- python3 $R/experiments/exp_tiled_plan.py > /tmp/t_fetch.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/t_l2 -- python3 $R/experiments/exp_tiled_plan.py > /tmp/t_l2.log 2>&1
python3 - <<PY
import sys
sys.path.insert(0, "$R/dgl-0.5-benchmark_amd")
import kernel_controls as kc
for d in ("/tmp/t_fetch", "/tmp/t_l2"):
    for i, b in enumerate(kc.parse_pmc_dir(d)):
        print(d, "block", i, "launches", b["launches"], {k: round(v / max(b["launches"], 1), 1) for k, v in b["counters"].items()})
PY
