"""bound, then prototype, of the compressed-row gather for the D = 64 forward aggregations whose input is a
relu + dropout output (75 % zeros) -- experiments/csrc/packed_proto.hip.  Products graph, the wave-per-item part of the two-part
plan (rows above 16 edges + hub chunks: 89 % of the edges, 1.86 ms of the 2.23 ms call in profiles/r04_kernel_stats.txt).

  python experiments/exp_packed_rows.py [--scale 1.0] > gpurun_out/r05_sparse_row_bound.txt
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import torch  # noqa: E402

import dgl  # noqa: E402
import kernel_controls as kc  # noqa: E402
from mi355x_graph import _lib, sparse  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--scale", type=float, default=1.0)
p.add_argument("--reps", type=int, default=8)
args = p.parse_args()
dev = torch.device("cuda:0")
src_dir = os.path.join(ROOT, "experiments", "csrc")
so = os.path.join(src_dir, "libpacked_proto.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(src_dir, "packed_proto.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-shared",
                           "-o", so, os.path.join(src_dir, "packed_proto.hip")])
proto = ctypes.CDLL(so)
# (mode, unroll, grid, item_row, item_beg, item_end, xcd_start, indices, slots, dense, ld, out, ldo, partial, rpb, mean, stream)
proto.packed_spmm.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 7 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                              ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
proto.packed_pack.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

n, (src, dst) = kc.control_edges("products", dev, args.scale)
g = dgl.graph((src, dst), num_nodes=n).int().formats(["csc"]).to(dev)
del src, dst
csc = g._index.csc()
E = csc.nnz
be = sparse.backend_for(csc.indptr)
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
print("# products graph x %g: N = %d, E = %d" % (args.scale, n, E))


def timed(fn, reps=args.reps, discard=2):
    ts = []
    for i in range(reps + discard):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        if i >= discard:
            ts.append(a.elapsed_time(b))
    return sum(ts) / len(ts)


# ---------------------------------------------------------------- (a) what one line per edge costs: the dense kernel at D = 32 / 16
print("\n## (a) the library's dense call by width (copy_u/mean, whole graph, its own plan choice)")
for D in (64, 32, 16):
    x = torch.rand(n, D, device=dev)
    out = torch.empty(n, D, device=dev)
    ms = timed(lambda: be.spmm_copy_u_strided(csc, "mean", x, out))
    print("D = %3d  %.3f ms   (%s)" % (D, ms, _lib.lib().mgx_last_spmm_kernel().decode()))

# ---------------------------------------------------------------- the two-part plan of D = 64
plan, short = csc.spmm_plan_for(64)
assert short and plan.rest is not None, "expected the two-part plan on this graph"
rest = plan.rest
rest_edges = int((rest.item_end - rest.item_beg).sum())
print("\n## two-part plan at D = 64: %d short items, %d items in `rest` (%d edges = %.1f %%, %d partial slots, %d hubs)"
      % (plan.num_items, rest.num_items, rest_edges, 100.0 * rest_edges / E, rest.num_slots, rest.num_hubs))

x = torch.rand(n, 64, device=dev) * (torch.rand(n, 64, device=dev) < 0.25)
nnz_row = (x != 0).sum(1)
print("input: relu+dropout-like, %.1f %% non-zero, rows above 24 non-zeros: %.3f %%" % (100.0 * float((x != 0).float().mean()),
                                                                                   100.0 * float((nnz_row > 24).float().mean())))
out_full = torch.empty(n, 64, device=dev)
ms_full = timed(lambda: be.spmm_copy_u_strided(csc, "mean", x, out_full))
out_ref = out_full
print("\n## (reference) the dense call on this input (short part + rest + hub fix-up): %.3f ms;  by kernel in the epoch trace\n"
      "   (gpurun_out/r05_plain_epoch_timeline.txt): short rows 0.34 ms + rest 1.83 ms + fix-up 0.035 ms" % ms_full)

# ---------------------------------------------------------------- pack
slots = torch.zeros(n, 32, dtype=torch.int32, device=dev)
over = torch.zeros(1, dtype=torch.int64, device=dev)
rc = proto.packed_pack(n, ctypes.c_void_p(x.data_ptr()), 64, ctypes.c_void_p(slots.data_ptr()), ctypes.c_void_p(over.data_ptr()), stream)
assert rc == 0
torch.cuda.synchronize()
print("packed: %d rows in 128-byte slots (%.0f MB against %.0f MB dense), %d overflow rows" % (n, n * 128 / 1e6, n * 256 / 1e6, int(over)))

rpb = 16
xs = rest.xcd_item_start
if not xs[-1]:
    per = ((rest.num_items + 7) // 8 + rpb - 1) // rpb * rpb
    xs = [min(i * per, rest.num_items) for i in range(9)]
xcd_dev = torch.tensor(xs, dtype=torch.int64, device=dev)
grid = max((xs[i + 1] - xs[i] + rpb - 1) // rpb for i in range(8)) * 8
out = torch.zeros(n, 64, device=dev)
part2 = torch.zeros(max(rest.num_slots, 1), 64, device=dev)


def packed(mode, U):
    rc = proto.packed_spmm(mode, U, grid, ctypes.c_void_p(rest.item_row.data_ptr()), ctypes.c_void_p(rest.item_beg.data_ptr()),
                           ctypes.c_void_p(rest.item_end.data_ptr()), ctypes.c_void_p(xcd_dev.data_ptr()), ctypes.c_void_p(csc.indices.data_ptr()),
                           ctypes.c_void_p(slots.data_ptr()), ctypes.c_void_p(x.data_ptr()), 64, ctypes.c_void_p(out.data_ptr()), 64,
                           ctypes.c_void_p(part2.data_ptr()), rpb, 1, stream)
    assert rc == 0, rc


print("\n## (b) packed 128-byte slots over `rest` (%d workgroups, %d items each)" % (grid, rpb))
print("%-62s %9s" % ("variant", "ms"))
for mode, U, what in ((0, 2, "gather floor (one line per edge, values summed as fetched)"), (0, 4, "gather floor, 32 rows in flight per wave"),
                      (1, 2, "+ register unpack mix (4 bpermute + popcounts per edge), synthetic"), (1, 4, "same, 32 rows in flight"),
                      (2, 2, "LDS accumulate (ds_add_f32 per value), exact"), (2, 4, "LDS accumulate, 32 rows in flight"),
                      (2, 8, "LDS accumulate, 64 rows in flight"),
                      (4, 4, "LDS atomics, padding column skipped"),
                      (3, 2, "LDS read-add-write (no atomics), exact"), (3, 4, "LDS read-add-write, 32 rows in flight"),
                      (3, 8, "LDS read-add-write, 64 rows in flight")):
    ms = timed(lambda: packed(mode, U))
    print("mode %d U %d  %-50s %9.3f" % (mode, U, what, ms))

# ---------------------------------------------------------------- modes 2 / 3 are exact: compare with the dense kernel's rows
packed(3, 4)
torch.cuda.synchronize()
direct = rest.item_row[rest.item_row >= 0].long()
err3 = (out[direct] - out_ref[direct]).abs().max() / out_ref[direct].abs().max()
print("\nmode 3 against the dense kernel on the directly written rows of `rest`: max |diff| / max |ref| = %.2e" % float(err3))
packed(2, 4)
torch.cuda.synchronize()
direct = rest.item_row[rest.item_row >= 0].long()
err = (out[direct] - out_ref[direct]).abs().max() / out_ref[direct].abs().max()
print("\nmode 2 against the dense kernel on the %d directly written rows of `rest`: max |diff| / max |ref| = %.2e" % (direct.numel(), float(err)))
if rest.num_slots:
    # hub rows: the packed kernel left their chunks in `part2`; combine them in slot order and compare with the dense call's rows
    hub_ptr = rest.hub_slot_ptr.long()
    seg = torch.repeat_interleave(torch.arange(rest.num_hubs, device=dev), hub_ptr[1:] - hub_ptr[:-1])
    hub_sum = torch.zeros(rest.num_hubs, 64, device=dev).index_add_(0, seg, part2)
    hub_rows = rest.hub_row.long()
    deg = (csc.indptr[hub_rows + 1] - csc.indptr[hub_rows]).clamp(min=1).float().view(-1, 1)
    herr = (hub_sum / deg - out_ref[hub_rows]).abs().max() / out_ref[hub_rows].abs().max()
    print("the %d hub rows (partial slots combined on the host side of this script): max |diff| / max |ref| = %.2e" % (rest.num_hubs, float(herr)))
print("\n# decision rule (VERDICT r04 item 5): build it if short part (0.34) + fix-up (0.035) + packed rest < 1.8 ms, i.e. packed rest < 1.42 ms; "
      "dense today: %.3f ms" % ms_full)
print("# Reading (docs/LOG_r05.md sections 5, 9c).  One 128-byte line per edge over the wave-per-item part: 0.93 - 1.03 ms against 1.83 ms for the dense\n"
      "# rows -- both at ~15 TB/s of L2 -> CU row bytes.  A register-side unpack (mode 1 is half of one) or LDS float atomics (mode 2 / 4) give the\n"
      "# gain back; mode 3 -- the accumulator rows in LDS updated by plain read - add - write, no atomics: the columns of a source row are distinct,\n"
      "# lane groups own different rows, a wave's LDS operations run in order -- is exact and meets the rule.  Built: csrc/spmm_slots.inc.")
