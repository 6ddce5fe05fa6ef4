cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
MGX_BENCH_SHARE_GPU=1 MGX_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 4 --scale 0.05 --steps 5 --warmup 2 2>gpurun_out/b37.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: d[k] for k in ('n_gpus', 'ms_per_step', 'value', 'scaling')}, d['config'].get('partition', {}).get('edge_cut_pct'), d['config']['final_loss'])
"
tail -4 gpurun_out/b37.err | cut -c1-200
