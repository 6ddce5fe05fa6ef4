#!/bin/bash
# kernel trace of the emulated 8-rank epoch: how much of a rank's stretch between collectives is device work, how much is launch gaps
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_emu_trace
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --emulate-ranks 8 --steps 5 --warmup 2 > $OUT/line.json 2> $OUT/err.txt
tail -3 $OUT/err.txt
python3 - <<'PY'
import csv, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r04_emu_trace"
f = sorted(glob.glob(out + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
t0, t1 = ks[0][0], ks[-1][1]
print("kernels", len(ks), "span ms", (t1 - t0) / 1e6)
# the emulated phase = everything after the largest idle gap following the 1-GPU epochs; print busy fraction per 100 ms window
win = 50e6
import collections
busy = collections.Counter(); cnt = collections.Counter()
for s, e, n in ks:
    w = int((s - t0) // win); busy[w] += e - s; cnt[w] += 1
for w in sorted(busy):
    print("window %3d  busy %.3f  kernels %d  mean kernel us %.1f" % (w, busy[w] / win, cnt[w], busy[w] / cnt[w] / 1e3))
# last 40 % of kernels = the emulated ranks' timed epochs: name histogram
tail = ks[int(len(ks) * 0.6):]
h = collections.defaultdict(lambda: [0, 0])
for s, e, n in tail:
    k = n.split("(")[0][:90]; h[k][0] += 1; h[k][1] += e - s
tot = sum(v[1] for v in h.values()); span = tail[-1][1] - tail[0][0]
print("tail: kernels %d, busy ms %.2f over %.2f ms wall (%.1f %%)" % (len(tail), tot / 1e6, span / 1e6, 100.0 * tot / span))
for k, v in sorted(h.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8d %10.3f ms  %7.1f us  %s" % (v[0], v[1] / 1e6, v[1] / v[0] / 1e3, k))
PY
