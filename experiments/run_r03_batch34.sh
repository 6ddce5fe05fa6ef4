cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=dgl-0.5-benchmark_amd
timeout 900 python -m pytest tests/test_next_rows_gpu.py -q -m gpu -x --tb=short 2>&1 | tail -6
for v in 1 0; do
  echo "== MGX_SAGE_SPARSE_LAST=$v"
  MGX_SAGE_SPARSE_LAST=$v timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-controls --no-plain 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.3f loss %.5f' % (d['ms_per_step'], d['config']['final_loss']), [(k['D'], k['avg_launch_ms'], k['launches_per_epoch']) for k in d['roofline']['kernels']], d['roofline'].get('row_sparse_backward'))
"
done
