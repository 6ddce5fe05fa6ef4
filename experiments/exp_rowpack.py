"""Where the `pack` / `unpack` stretches of a rank's epoch go (dist.SparseHalo): every step of one packed halo exchange timed alone on
~1 M boundary rows of 64 columns, 20 % non-zero (the products partition at P = 8), against the dense gather it replaces.

  python experiments/exp_rowpack.py > gpurun_out/r05_rowpack.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import torch  # noqa: E402

from mi355x_graph import sparse  # noqa: E402

dev = torch.device("cuda:0")
be = sparse.backend_for(torch.empty(1, device=dev))
n_own, S, D = 310_000, 1_000_000, 64
torch.manual_seed(0)
h = torch.rand(n_own, 2 * D, device=dev)[:, :D]          # the left half of a layer's [h | neigh] buffer (row-strided)
h.mul_((torch.rand(n_own, D, device=dev) < 0.2).float())
idx = torch.randint(0, n_own, (S,), device=dev, dtype=torch.int32)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


masks, counts = be.rows_pack_count(h, idx)
off = torch.zeros(S + 1, dtype=torch.int64, device=dev)
torch.cumsum(counts, 0, dtype=torch.int64, out=off[1:])
total = int(off[-1])
vals = be.rows_pack_values(h, idx, masks, off, total)
dense = torch.empty(S, D, device=dev)
g = torch.rand(S, D, device=dev)
print("# %d boundary rows of %d columns out of %d owned rows, %.1f %% non-zero: %.1f MB of values + %.1f MB of bitmaps against %.1f MB dense"
      % (S, D, n_own, 100.0 * total / (S * D), total * 4 / 1e6, S * 8 / 1e6, S * D * 4 / 1e6))
print("%-58s %8s" % ("step", "us"))
for name, fn in [
    ("dense pack: mgx_gather_rows_strided (what it replaces)", lambda: be.gather_rows(h, idx)),
    ("rows_pack_count (bitmaps + counts)", lambda: be.rows_pack_count(h, idx)),
    ("torch.cumsum over the counts", lambda: torch.cumsum(counts, 0, dtype=torch.int64, out=off[1:])),
    ("rows_mask_count (receiver)", lambda: be.rows_mask_count(masks, D)),
    ("8 boundary offsets -> host (.cpu(): the one sync)", lambda: off[::S // 8].cpu()),
    ("rows_pack_values of the layer input (gathered rows)", lambda: be.rows_pack_values(h, idx, masks, off, total)),
    ("rows_pack_values of a dense gradient (identity rows)", lambda: be.rows_pack_values(g, None, masks, off, total)),
    ("rows_unpack -> dense rows", lambda: be.rows_unpack(masks, off, vals, D, out=dense)),
]:
    print("%-58s %8.1f" % (name, timed(fn)))
ret = sparse.coo_to_csr(n_own, S, idx, torch.arange(S, dtype=torch.int32, device=dev))
dh = torch.zeros(n_own, D, device=dev)
print("%-58s %8.1f" % ("rows_unpack_add_csr (packed rows into their owners)", timed(lambda: be.rows_unpack_add_csr(ret, masks, off, vals, dh))))
print("%-58s %8.1f" % ("  against: copy_u over return_csr of the dense rows", timed(lambda: be.spmm_copy_u_strided(ret, "sum", dense, dh, accumulate=True))))

# ---- the halo-SOURCE aggregation straight from the packed rows (round 5, section 9b: would it pay to skip the unpack?): 310 k owned
# rows with ~11 halo in-edges each, sources uniformly random among the 1 M received rows -- the shape of a P = 8 rank's halo CSC
E = 3_400_000
hd = torch.randint(0, n_own, (E,), device=dev, dtype=torch.int32)
hs = torch.randint(0, S, (E,), device=dev, dtype=torch.int32)
halo = sparse.coo_to_csr(n_own, S, hd, hs)
out = torch.zeros(n_own, 2 * D, device=dev)[:, D:]
inv = torch.rand(n_own, device=dev)
t_un = timed(lambda: be.rows_unpack(masks, off, vals, D, out=dense))
t_ag = timed(lambda: be.spmm_copy_u_strided(halo, "sum", dense, out, accumulate=True, dst_scale=inv))
t_pk = timed(lambda: be.rows_unpack_add_csr(halo, masks, off, vals, out))
print("# halo-source aggregation, %d edges into %d rows from %d received rows" % (E, n_own, S))
print("%-58s %8.1f" % ("rows_unpack + copy_u over the dense rows (now)", t_un + t_ag))
print("%-58s %8.1f" % ("  of which the aggregation", t_ag))
print("%-58s %8.1f" % ("rows_unpack_add_csr over the halo CSC (no 1/deg yet)", t_pk))
