#!/bin/bash
F="--no-cpu-baseline --no-pmc --no-controls --no-plain --no-scale-model --no-secondary --no-variants"
python3 bench.py $F > gpurun_out/ab_on.json 2> gpurun_out/ab_on.err
python3 -c "
import sys
sys.path.insert(0, 'dgl-0.5-benchmark_amd')
sys.argv = ['bench.py'] + '$F'.split()
from mi355x_graph import config
config.PACKED_GATHER = False
exec(compile(open('bench.py').read(), 'bench.py', 'exec'))
" > gpurun_out/ab_off.json 2> gpurun_out/ab_off.err
python3 - <<'PY'
import json
for tag in ("on", "off"):
    try:
        l = json.loads([x for x in open("gpurun_out/ab_%s.json" % tag) if x.startswith("{")][-1])
        print(tag, "ms_per_step", round(l["ms_per_step"], 3), "frac", l["roofline"]["frac"], "spmm_ms", l.get("spmm_ms_per_epoch"))
        for k in l["roofline"]["kernels"]:
            print("   ", k["kernel"][:90], k.get("calls"), k.get("avg_us") or k.get("mean_ms"), k.get("frac"))
    except Exception as e:
        print(tag, "failed", e)
        print(open("gpurun_out/ab_%s.err" % tag).read()[-1500:])
PY
