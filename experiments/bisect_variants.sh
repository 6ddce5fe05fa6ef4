for envs in "MGX_ROWS_GEMM=0" "MGX_ROWGROUP=0" "MGX_ACCELERATE_LINEAR=0"; do
  env $envs python bench.py --no-pmc --no-controls --no-cpu-baseline --no-scale-model > gpurun_out/bis.txt 2>/dev/null
  python - "$envs" <<PY
import json, sys
l=[x for x in open("gpurun_out/bis.txt") if x.startswith('{"metric')][-1]
d=json.loads(l)
print(sys.argv[1], {k:v for k,v in d.items() if k.startswith("epoch_ms") or k in ("ms_per_step",)})
PY
done
