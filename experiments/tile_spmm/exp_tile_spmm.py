"""EXPERIMENT: LDS-tile g-SpMM (tile_spmm.hip) against the library's row-per-wave kernel on dense-neighbourhood graphs.
  python experiments/tile_spmm/exp_tile_spmm.py reddit 602,128,64 [scale]
Plan construction is plain torch on the device (one-off per graph)."""
import ctypes
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "dgl-0.5-benchmark_amd"))
from mi355x_graph import sparse, schedule  # noqa: E402
from mi355x_graph.datasets import SHAPES, synthetic_edges  # noqa: E402

R, RW, C = 64, 16, 64
lib = ctypes.CDLL(os.path.join(HERE, "libtile_spmm.so"))
lib.tile_spmm.restype = ctypes.c_int
lib.tile_spmm.argtypes = [ctypes.c_void_p] * 12 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def build_tile_plan(csr, order):
    """csr: in-CSR (rows = destinations).  order: schedule (row permutation)."""
    dev = csr.indptr.device
    n, ncols = csr.num_rows, csr.num_cols
    T = (n + R - 1) // R
    pos = torch.empty(n, dtype=torch.int64, device=dev)
    pos[order] = torch.arange(n, device=dev)
    deg = (csr.indptr[1:] - csr.indptr[:-1]).long()
    rowpos = torch.repeat_interleave(pos, deg)                      # schedule position of each edge's row
    src = csr.indices.long()
    key = (rowpos // R) * ncols + src
    dkey, inv = torch.unique(key, sorted=True, return_inverse=True)  # distinct (tile, source)
    del key
    dtile = torch.div(dkey, ncols, rounding_mode="floor")
    dsrc = (dkey - dtile * ncols).int()
    first = torch.searchsorted(dtile, torch.arange(T + 1, device=dev))
    rank = torch.arange(dkey.numel(), device=dev) - first[dtile]
    per_tile = first[1:] - first[:-1]
    nchunks = (per_tile + C - 1) // C
    tile_chunk_ptr = torch.zeros(T + 1, dtype=torch.int64, device=dev)
    tile_chunk_ptr[1:] = torch.cumsum(nchunks, 0)
    NCH = int(tile_chunk_ptr[-1])
    gch = tile_chunk_ptr[dtile] + rank // C
    slot = rank % C
    chunk_src = torch.full((NCH * C,), -1, dtype=torch.int32, device=dev)
    chunk_src[gch * C + slot] = dsrc
    j = rowpos % R
    key2 = gch[inv] * R + j                                          # (chunk, wave, local row): wave = j // 16
    slot_e = slot[inv].to(torch.uint8)
    del inv
    key2, perm = torch.sort(key2)
    ent = slot_e[perm].contiguous()
    del perm
    counts = torch.bincount(key2, minlength=NCH * R)
    assert int(counts.max()) <= 65535, "multigraph too dense for 16-bit counts"
    per_cw = counts.view(NCH * 4, RW).sum(1)
    ent_ptr = torch.zeros(NCH * 4 + 1, dtype=torch.int64, device=dev)
    ent_ptr[1:] = torch.cumsum(per_cw, 0)
    tile_row = torch.full((T * R,), -1, dtype=torch.int32, device=dev)
    tile_row[:n] = order.int()
    # work items: a tile's chunks, at most MAXCH per item (hub-heavy tiles are cut; their items add atomically)
    MAXCH = int(os.environ.get("TILE_MAXCH", "256"))
    pieces = (nchunks + MAXCH - 1) // MAXCH
    item_tile = torch.repeat_interleave(torch.arange(T, device=dev), pieces)
    pfirst = torch.zeros(T + 1, dtype=torch.int64, device=dev)
    pfirst[1:] = torch.cumsum(pieces, 0)
    k = torch.arange(item_tile.numel(), device=dev) - pfirst[item_tile]
    item_beg = tile_chunk_ptr[item_tile] + k * MAXCH
    item_end = torch.minimum(item_beg + MAXCH, tile_chunk_ptr[item_tile + 1])
    item_whole = (pieces[item_tile] == 1).to(torch.uint8)
    print("chunks per tile: mean %.0f max %d; %d items" % (float(nchunks.float().mean()), int(nchunks.max()), item_tile.numel()))
    return dict(T=T, NCH=NCH, I=int(item_tile.numel()), item_tile=item_tile.int(), item_beg=item_beg.int(), item_end=item_end.int(),
                item_whole=item_whole, chunk_src=chunk_src, counts=counts.to(torch.int16),
                ent_ptr=ent_ptr.int(), ent=ent, tile_row=tile_row, distinct=int(dkey.numel()))


def tile_spmm(plan, x, out, row_scale, npass):
    rc = lib.tile_spmm(x.data_ptr(), out.data_ptr(), plan["item_tile"].data_ptr(), plan["item_beg"].data_ptr(),
                       plan["item_end"].data_ptr(), plan["item_whole"].data_ptr(), plan["chunk_src"].data_ptr(),
                       plan["counts"].data_ptr(), plan["ent_ptr"].data_ptr(), plan["ent"].data_ptr(), plan["tile_row"].data_ptr(),
                       0 if row_scale is None else row_scale.data_ptr(), x.shape[1], plan["I"], npass,
                       torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "reddit"
    widths = [int(w) for w in (sys.argv[2] if len(sys.argv) > 2 else "602,128,64").split(",")]
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    dev = torch.device("cuda")
    spec = SHAPES[name]
    n, m = int(spec["n"] * scale), int(spec["m"] * scale)
    src, dst = synthetic_edges(n, m, min(spec["max_deg"], n - 1), spec["seed"], dev, symmetric=spec["symmetric"])
    csr = sparse.coo_to_csr(n, n, dst.int(), src.int())   # rows = destinations
    E = csr.nnz
    t0 = time.time()
    order = schedule.locality_order(csr)
    torch.cuda.synchronize()
    t1 = time.time()
    plan = build_tile_plan(csr, order)
    torch.cuda.synchronize()
    print("%s: N = %d, E = %d; schedule %.2f s, tile plan %.2f s; %d tiles, %d chunks, edges / distinct = %.2f" %
          (name, n, E, t1 - t0, time.time() - t1, plan["T"], plan["NCH"], E / plan["distinct"]), flush=True)
    for D in widths:
        x = torch.rand(n, D, device=dev)
        ref, _, _ = sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None)
        t_ref = timeit(lambda: sparse.gspmm_raw(csr, "copy_lhs", "sum", x, None))
        for npass in sorted({(D + 255) // 256, (D + 127) // 128, (D + 63) // 64}):
            out = torch.zeros(n, D, device=dev)
            tile_spmm(plan, x, out, None, npass)
            torch.cuda.synchronize()
            err = float(((out - ref).abs() / ref.abs().clamp(min=1.0)).max())
            t = timeit(lambda: (out.zero_(), tile_spmm(plan, x, out, None, npass)))
            print("D = %4d: row-per-wave %.3f ms (gather %.1f TB/s) | tile, %d column pass(es) %.3f ms  max rel err %.2e" %
                  (D, t_ref, E * D * 4 / t_ref / 1e9, npass, t, err), flush=True)


if __name__ == "__main__":
    main()
