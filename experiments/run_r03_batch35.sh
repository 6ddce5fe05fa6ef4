cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python bench.py --steps 10 --warmup 3 --no-pmc --no-controls --no-cpu-baseline 2>gpurun_out/b35.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: d[k] for k in ('ms_per_step', 'epoch_ms_plain_model', 'epoch_ms_last_layer_backward_on_loss_rows', 'last_layer_backward_on_loss_rows')})
print(d['roofline']['frac'], d['roofline']['row_sparse_backward'])
"
tail -3 gpurun_out/b35.err
