#!/bin/bash
# Round 5: the whole GPU suite + smoke() + default bench on the tree as committed.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
timeout 1200 python3 -m pytest tests -m gpu -x -q > $O/r05_b6_pytest.log 2>&1; echo "pytest rc $?"
grep -n "passed\|failed" $O/r05_b6_pytest.log | tail -3
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python3 bench.py > $O/r05_b6_bench_line.json 2> $O/r05_b6_bench_line.err; echo "bench rc $?"
python3 - <<'PY'
import json
l=json.loads(open('gpurun_out/r05_b6_bench_line.json').read().strip().splitlines()[-1])
print(l['ms_per_step'], l['ms_per_step_reference_modules'], l['roofline']['frac'], {k:v.get('ms_per_step') for k,v in l['secondary'].items()})
PY
