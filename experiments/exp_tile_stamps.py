"""Where a tile of the LDS-staged g-SpMM spends its cycles (diagnostic build of csrc/spmm_tile.hip with -DMGX_TILE_STAMPS: s_memtime
stamps per wave, written to a side buffer; the product build executes no stamp).

  MGX_LIB_PATH=experiments/tile_spmm/libmgx_stamps.so python experiments/exp_tile_stamps.py reddit 14x6x2x2 [D] [lanes_log2]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
from mi355x_graph import _lib, schedule, sparse, tileplan  # noqa: E402
from mi355x_graph.datasets import SHAPES, synthetic_edges  # noqa: E402

os.environ["MGX_TILE"] = "0"
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "reddit"
cfg = sys.argv[2] if len(sys.argv) > 2 else "14x6x2x2"
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
LG = int(sys.argv[4]) if len(sys.argv) > 4 else 4  # lanes per row (log2): 3 / 2 = the narrow kernels
nc, nacc, nl, tau = [int(v) for v in cfg.split("x")]
spec = SHAPES[name]
n, m = spec["n"], spec["m"]
src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], dev, symmetric=spec["symmetric"])
csr = sparse.coo_to_csr(n, n, dst.int().contiguous(), src.int().contiguous())
del src, dst
base = schedule.plan_for(csr, split=2048)
tp = tileplan.build_tile_plan(csr, base, nc, nacc, nl, tau, lanes_log2=LG)
be = sparse.backend_for(torch.zeros(1, device=dev))
x = torch.rand(n, D, device=dev)
L = _lib.lib()
fn = L.mgx_debug_set_tile_stamps
fn.argtypes = [ctypes.c_void_p]
fn.restype = None
for _ in range(3):
    be.spmm_tile_copy_u(csr, tp, "sum", x)
torch.cuda.synchronize()
T = tp.num_tiles
stamps = torch.zeros((T, 16, 8), dtype=torch.int64, device=dev)
fn(stamps.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
be.spmm_tile_copy_u(csr, tp, "sum", x)
e1.record()
torch.cuda.synchronize()
fn(None)
ms = e0.elapsed_time(e1)
s = stamps.cpu().double()
ld, co = s[:, :nl], s[:, nl:nl + nc]
tot = co[:, :, 3]
t0 = co[:, :, 6]
span = float((t0 + tot).max() - t0.min())
print("%s %s D=%d: %.3f ms; s_memtime ticks from the first tile's start to the last tile's end %.0f (= %.0f MHz if one tick is one cycle)"
      % (name, cfg, D, ms, span, span / ms / 1e3))
f = lambda v: "%.0f (%.0f%%)" % (v, 100.0 * v / tot.mean())
print("per tile, mean over tiles and consumer waves, ticks: total %.0f = barrier wait %s + LDS work %s + direct-phase barrier %s + direct %s + rest %s"
      % (tot.mean(), f(co[:, :, 0].mean()), f(co[:, :, 1].mean()), f(co[:, :, 5].mean()), f(co[:, :, 2].mean()),
         f((tot - co[:, :, 0] - co[:, :, 1] - co[:, :, 5] - co[:, :, 2]).mean())))
print("  LDS work: max / mean over the waves of a tile %.2f; supersteps per wave and chunk %.1f; ticks per superstep and wave %.0f (x %d waves -> %.1f per CU; 4 ds_read_b128 = 16 LDS cycles)"
      % ((co[:, :, 1].max(1)[0] / co[:, :, 1].mean(1).clamp(min=1)).mean(), (co[:, :, 4].sum(1) / ld[:, 0, 4].clamp(min=1) / nc).mean(),
         co[:, :, 1].sum() / co[:, :, 4].sum().clamp(min=1), nc, co[:, :, 1].sum() / co[:, :, 4].sum().clamp(min=1) / nc))
print("  per chunk: barrier wait %.0f ticks, LDS work %.0f ticks (mean wave), %.0f (slowest wave)"
      % ((co[:, :, 0].sum(1) / nc / ld[:, 0, 4].clamp(min=1)).mean(), (co[:, :, 1].sum(1) / nc / ld[:, 0, 4].clamp(min=1)).mean(),
         (co[:, :, 1].max(1)[0] / ld[:, 0, 4].clamp(min=1)).mean()))
print("loader waves, ticks: total %.0f = wait for DMA %.0f + barrier %.0f + issue %.0f; chunks per tile %.1f; per chunk: wait %.0f barrier %.0f issue %.0f"
      % (ld[:, :, 3].mean(), ld[:, :, 0].mean(), ld[:, :, 1].mean(), ld[:, :, 2].mean(), ld[:, 0, 4].mean(),
         (ld[:, :, 0].mean(1) / ld[:, 0, 4].clamp(min=1)).mean(), (ld[:, :, 1].mean(1) / ld[:, 0, 4].clamp(min=1)).mean(),
         (ld[:, :, 2].mean(1) / ld[:, 0, 4].clamp(min=1)).mean()))
tile_t = tot.max(1)[0]
print("tile duration, ticks: mean %.0f, min %.0f, max %.0f; sum / 256 CUs = %.0f against a span of %.0f (%d tiles, %.2f per CU): %.0f%% of the CU-time is inside tiles"
      % (tile_t.mean(), tile_t.min(), tile_t.max(), tile_t.sum() / 256, span, T, T / 256, 100.0 * tile_t.sum() / 256 / span))
xcc = co[:, 0, 7].long() & 0xF
print("tiles per XCC id:", torch.bincount(xcc, minlength=8).tolist())
