cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
for v in 0 1; do
  echo "== MGX_SAGE_BWD_OWN_MATRIX=$v"
  MGX_SAGE_BWD_OWN_MATRIX=$v timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-controls --no-plain 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.3f' % d['ms_per_step'], [(k['D'], k['avg_launch_ms'], k.get('accumulating_launches_per_epoch')) for k in d['roofline']['kernels']])
"
done
