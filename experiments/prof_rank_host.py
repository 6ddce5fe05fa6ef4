"""Host side of ONE rank's training step in a P-way partition (the launch floor of LOG r04 section 19): cProfile over rank 0's thread of an
emulated world's SOLO epochs (emulate.py), at a scale where device time is negligible (default 0.02) or at full size.
    python experiments/prof_rank_host.py [scale] [P] [steps]"""
import os
import pstats
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dgl-0.5-benchmark_amd"))
import tunable  # noqa: E402
tunable.setup()
import torch  # noqa: E402
import dgl  # noqa: E402,F401
import full_graph  # noqa: E402
import scale_model  # noqa: E402
from mi355x_graph.datasets import SHAPES, synthetic_edges  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")
spec = SHAPES["products"]
cfg = full_graph.SAGE_CONFIGS["products"]
n, m = int(spec["n"] * scale), int(spec["m"] * scale)
src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], dev, symmetric=True)
gen = torch.Generator().manual_seed(1)
feats = torch.rand(n, spec["feat"], generator=gen)
labels = torch.randint(0, spec["classes"], (n,), generator=gen)
train_mask = torch.rand(n, generator=gen) < 0.08
out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rank_host_%d.prof" % os.getpid())
t0 = time.perf_counter()
res = scale_model.run(dev, src, dst, n, feats, labels, train_mask, cfg, spec, P, steps=steps, warmup=3, host_profile=out,
                      progress=lambda s: print(s, flush=True))
print("scale %.3f P %d: compute per rank (ms): %s" % (scale, P, [round(r["compute_ms"], 3) for r in res["per_rank"]]))
print("solo epoch per rank (ms): %s" % [r["solo_epoch_ms"] for r in res["per_rank"]])
for r in (1, 2):
    print("  rank %d by stretch (ms): %s" % (r, res["per_rank"][r]["compute_ms_by_stretch"]))
st = pstats.Stats(out)
print("== %d steps of rank 0: by own time" % steps)
st.sort_stats("tottime").print_stats(45)
print("== by cumulative time")
st.sort_stats("cumulative").print_stats(60)
print("== who reads the device back")
st.print_callers("'item'")
st.print_callers("'cpu'")
st.print_callers("tolist")
