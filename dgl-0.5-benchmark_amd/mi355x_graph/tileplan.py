"""Tile plans (mgx_tile_plan) for the LDS-staged g-SpMM of dense neighbourhoods (csrc/spmm_tile.hip).

Why: on graphs with hundreds of in-edges per node (reddit: 492, proteins: 597; kernel/dgl-new.py:61,
main_dgl_reddit_sage.py:73-80) the row-per-wave kernel delivers every edge's source row from L2 to a CU -- E*D*4 bytes at
the chip's 17-19 TB/s L2 gather rate, 3-5 % of the HBM roofline -- while destination rows that the locality schedule placed
next to each other read the SAME source rows 4.6x (reddit-shaped, 256 rows) to 7.7x (1024 rows) over.  A tile plan cuts the
schedule into TILES of R work items (one 1024-thread workgroup each), and for every tile

  * lists the tile's STAGED sources -- the ones gathered at least `tau` times inside the tile -- in CHUNKS of 127 rows; a
    chunk is gathered ONCE from L2 into LDS (LDS-DMA, 4-deep ring) and every edge into it is then served from LDS;
  * turns the edges into per-wave STREAMS the kernel can walk without looking at the graph again: a wave owns NACC rows per
    16-lane group (4 groups: 4 * NACC rows), one accumulator register set per row, statically indexed; a stream STEP holds one
    LDS slot (1 byte) per lane group, rows are padded to 4-step SUPERSTEPS with the chunk's all-zero slot, so the inner loop
    is branch-free: 4 x (slot -> ds_read_b128) then 4 x add;
  * sends the remaining edges (sources used once in the tile: no reuse to exploit) down a DIRECT stream of source ids that
    the same waves gather from global memory into the same accumulators afterwards.

Work items are the (row | hub chunk, edge range) items of an mgx_spmm_plan, so hub rows stay split and their partial sums go
through the same fix-up kernel.  Inside a tile, items are ranked by length; 4 consecutive ranks share a wave-instruction
(similar lengths: little padding) and the quads are dealt to the waves boustrophedon (similar work per wave: the per-chunk
barrier waits for the slowest).  Integer preprocessing, once per CSR, like the CSR build itself; torch sorts on the device.
"""
import ctypes
import os

import torch

from . import config as _settings

GROUPS = 4          # 16-lane groups of a wave (one 64-column pass: 16 lanes x float4)
CHUNK_SLOTS = 128   # LDS rows per chunk: 127 staged sources + the all-zero slot
ZERO_SLOT = CHUNK_SLOTS - 1
# Narrow rows (round 3, later): 8 or 4 lanes per row -> 8 / 16 rows per wave-instruction, 32 / 16 columns per pass, 256 slots per chunk
GEOMETRY = {4: (4, 128), 3: (8, 256), 2: (16, 256)}  # log2(lanes per row) -> (lane groups per wave, LDS slots per chunk)
TILE_WAVES = (8, 16)  # workgroups of 8 waves (two per CU, 2-chunk ring) or 16 waves (one per CU, 4-chunk ring)
CNT_STRIDE = 16     # superstep counts per (chunk | tile, wave) in lds_cnt / dir_cnt
NO_ITEM = -(2 ** 31)
STREAM_TAIL = 128   # padding SUPERSTEPS behind every stream (the kernels prefetch whole windows of 16-32 supersteps, two ahead)


class MgxTilePlan(ctypes.Structure):
    _fields_ = [
        ("num_tiles", ctypes.c_int64),
        ("num_chunks", ctypes.c_int64),
        ("lds_steps", ctypes.c_int64),
        ("dir_steps", ctypes.c_int64),
        ("consumers", ctypes.c_int32),
        ("nacc", ctypes.c_int32),
        ("loaders", ctypes.c_int32),
        ("lanes_log2", ctypes.c_int32),
        ("tile_chunk_ptr", ctypes.c_void_p),
        ("chunk_ids", ctypes.c_void_p),
        ("lds_off", ctypes.c_void_p),
        ("lds_cnt", ctypes.c_void_p),
        ("lds_stream", ctypes.c_void_p),
        ("dir_off", ctypes.c_void_p),
        ("dir_cnt", ctypes.c_void_p),
        ("dir_stream", ctypes.c_void_p),
        ("tile_item", ctypes.c_void_p),
        ("zero_row", ctypes.c_void_p),
        ("tile_order", ctypes.c_void_p),
        ("tile_node", ctypes.c_void_p),
        ("lds_stream16", ctypes.c_void_p),
    ]


class TilePlan(object):
    """Device tables of one tile plan (see build_tile_plan); `base` is the SpmmPlan whose items it tiles."""

    def __init__(self, base, consumers, nacc, loaders, tables, stats, lanes_log2=4):
        self.base, self.consumers, self.nacc, self.loaders = base, int(consumers), int(nacc), int(loaders)
        self.lanes_log2 = int(lanes_log2)
        self.groups, self.chunk_slots = GEOMETRY[self.lanes_log2]
        self.__dict__.update(tables)
        self.stats = stats
        self._c = None

    @property
    def rows_per_tile(self):
        return self.consumers * self.nacc * self.groups

    @property
    def num_tiles(self):
        return int(self.tile_chunk_ptr.shape[0]) - 1

    @property
    def num_chunks(self):
        return int(self.chunk_ids.shape[0]) // self.chunk_slots

    def c_struct(self):
        if self._c is None:
            p = lambda t: t.data_ptr() if t.numel() else None
            self._c = MgxTilePlan(self.num_tiles, self.num_chunks, int(self.lds_stream.shape[0]) // self.groups - STREAM_TAIL,
                                  int(self.dir_stream.shape[0]) // (4 * self.groups) - STREAM_TAIL,
                                  self.consumers, self.nacc, self.loaders, self.lanes_log2, p(self.tile_chunk_ptr), p(self.chunk_ids), p(self.lds_off),
                                  p(self.lds_cnt), p(self.lds_stream), p(self.dir_off), p(self.dir_cnt), p(self.dir_stream),
                                  p(self.tile_item), p(self.zero_row),
                                  p(self.tile_order) if getattr(self, "tile_order", None) is not None else None,
                                  p(self.tile_node) if getattr(self, "tile_node", None) is not None else None,
                                  p(self.lds_stream16) if getattr(self, "lds_stream16", None) is not None else None)
        return self._c


def _excl_cumsum(x):
    out = torch.zeros(x.shape[0] + 1, dtype=torch.int64, device=x.device)
    torch.cumsum(x, 0, out=out[1:])
    return out


def _streams(seg_key, nseg, payload, pad, dtype, seg_order=None, groups=GROUPS, bank_classes=0):
    """Edges with segment key (seg * 4 + g), seg = a (unit, cw, j) row of the stream -> (superstep counts per seg, first superstep
    of each seg, stream).  A row has max_g(count) steps rounded up to whole SUPERSTEPS (4 steps); layout
    [superstep][lane group g][step u]: one payload per entry, `pad` where a group has run out.  `seg_order`: the order in which
    the rows are laid out in the stream (default: seg order).  Edges keep their storage order inside a group (stable sort)."""
    dev = seg_key.device
    cnt4 = torch.bincount(seg_key, minlength=nseg * groups)
    steps = cnt4.view(nseg, groups).max(dim=1)[0]
    ssteps = (steps + 3) // 4
    if seg_order is None:
        base = _excl_cumsum(ssteps)  # first superstep of every row
        total = int(base[-1])
        base = base[:-1]
    else:
        laid = _excl_cumsum(ssteps[seg_order])
        total = int(laid[-1])
        base = torch.empty(nseg, dtype=torch.int64, device=dev)
        base[seg_order] = laid[:-1]
    stream = torch.full(((total + STREAM_TAIL) * groups * 4,), pad, dtype=dtype, device=dev)
    if seg_key.numel():
        if bank_classes:
            # narrow rows: a 16-lane pass of ds_read_b128 reads 16 / lanes-per-row rows, and two of them conflict when their slots
            # are equal mod `bank_classes` (a 64-byte row is one bank quarter, a 128-byte row one half).  Inside a group's list the
            # order is free (a sum): every group walks its entries class by class, starting from a class of its own, so that the
            # groups of a pass are mostly in different classes at the same step.
            cls = ((payload.long() & 0xFF) + (seg_key % groups)) % bank_classes
            order = torch.sort(seg_key * bank_classes + cls, stable=True)[1]
        else:
            order = torch.sort(seg_key, stable=True)[1]
        sk = seg_key[order]
        seg_start = _excl_cumsum(cnt4)
        rank = torch.arange(sk.shape[0], device=dev) - seg_start[sk]
        idx = ((base[sk // groups] + rank // 4) * groups + (sk % groups)) * 4 + rank % 4
        stream[idx] = payload[order].to(dtype)
    return ssteps, base, total, stream


XCDS = 8
NUM_CUS = 256
WG_SLOTS = 2 * NUM_CUS  # 8-wave workgroups, two per CU


def build_tile_plan(csr, base, consumers=12, nacc=6, loaders=4, tau=2, balance=False, lanes_log2=4, pair_rank=False):
    """csr: CsrView (int32, device or host); base: SpmmPlan over it (items in schedule order; None = natural rows).
    pair_rank (the plans of the fused GAT walks, gat_tile.inc): every entry also carries the RANK k of its edge among the parallel
    edges of its (row, source) pair, in edge-id order -- (destination, source, k) is then a key per edge that both CSRs of a graph
    agree on, which is what the walks key attn_drop by.  Staged entries: a second stream of 16-bit entries, slot | k << 8
    (`lds_stream16`); direct entries: k in bits 24-30 of the source id.  k is kept mod 128: a pair with MORE than 128 parallel
    edges (the heavy-tailed stand-ins have hub pairs with thousands; the datasets have none) re-uses keys, i.e. its edges with
    equal k mod 128 share a mask bit -- both CSRs still see the same multiset of bits per pair, so the three walks stay consistent."""
    dev = csr.indptr.device
    NC, NACC = int(consumers), int(nacc)
    if NC + int(loaders) not in TILE_WAVES or not 1 <= NACC <= CNT_STRIDE or int(loaders) not in (1, 2, 4):
        raise ValueError("tile plan: consumers + loaders must be 8 or 16, nacc in 1..16, loaders 1, 2 or 4")
    GROUPS, CHUNK_SLOTS = GEOMETRY[int(lanes_log2)]  # shadow the module defaults: lane groups per wave, LDS slots per chunk
    ZERO_SLOT = CHUNK_SLOTS - 1
    R = NC * NACC * GROUPS
    n_src = csr.num_cols
    if base is not None:
        item_row = base.item_row.long()
        beg, end = base.item_beg.long(), base.item_end.long()
    else:
        item_row = torch.arange(csr.num_rows, device=dev)
        beg, end = csr.indptr[:-1].long(), csr.indptr[1:].long()
    I = int(item_row.shape[0])
    lens = end - beg
    E = int(lens.sum())
    # ---- tiles: consecutive items of the schedule, at most R of them.  balance = S (workgroup slots of the chip: 2 per CU) cuts them
    # to the same number of EDGES each, and so many that the launch is a whole number of rounds over the S slots: XCD x takes its
    # stretch of tiles in order as slots free, so a launch of 1.1 rounds of equal tiles -- or of 2.9 rounds of unequal ones -- idles
    # most of the chip for its last round (tile_costs / simulate_dispatch model this; profiles/r03_tile_dispatch.txt).
    T0 = max((I + R - 1) // R, 1)
    S = WG_SLOTS if balance is True else int(balance)
    if S and T0 >= S // 2:
        rounds = (T0 + S - 1) // S
        budget = max(E // max(rounds * S - S // 32, 1), 1)
        te = (torch.cumsum(lens, 0) - lens) // budget              # tile by edge count ...
        first = torch.ones(I, dtype=torch.bool, device=dev)
        first[1:] = te[1:] != te[:-1]
        start = torch.cummax(torch.where(first, torch.arange(I, device=dev), torch.zeros(I, dtype=torch.int64, device=dev)), 0)[0]
        sub = (torch.arange(I, device=dev) - start) // R                # ... cut again where it holds more than R items
        key = te * (I // R + 2) + sub
        newt = torch.ones(I, dtype=torch.int64, device=dev)
        newt[1:] = (key[1:] != key[:-1]).long()
        it_tile = torch.cumsum(newt, 0) - 1
        it_first = torch.cummax(torch.where(newt.bool(), torch.arange(I, device=dev), torch.zeros(I, dtype=torch.int64, device=dev)), 0)[0]
        T = int(it_tile[-1]) + 1
    else:
        it_tile = torch.arange(I, device=dev) // R
        it_first = it_tile * R
        T = T0
    # ---- position of every item inside its tile: rank by length, quads of ranks, quads dealt to the waves back and forth
    maxlen = int(lens.max()) if I else 0
    order = torch.sort(it_tile * (maxlen + 1) + (maxlen - lens), stable=True)[1]
    rank = torch.empty(I, dtype=torch.int64, device=dev)
    rank[order] = torch.arange(I, device=dev) - it_first[order]
    quad, g = rank // GROUPS, rank % GROUPS
    rnd, w = quad // NC, quad % NC
    cw = torch.where(rnd % 2 == 0, w, NC - 1 - w)
    pos = (cw * NACC + rnd) * GROUPS + g
    tile_item = torch.full((T * R,), NO_ITEM, dtype=torch.int64, device=dev)
    tile_item[it_tile * R + pos] = item_row
    tile_node = torch.zeros((T * R,), dtype=torch.int32, device=dev)  # the NODE of every position's item (hub chunks: their row)
    tile_node[it_tile * R + pos] = (base.item_node if base is not None and base.item_node is not None else item_row).to(torch.int32)
    # ---- edges: (tile, position in tile, source)
    e_item = torch.repeat_interleave(torch.arange(I, device=dev), lens)
    first = _excl_cumsum(lens)[:-1]
    # E-sized temporaries are kept as narrow as their range allows and freed as soon as their last reader is past (reddit has
    # 115 M edges and up to five plans per graph: a dozen live int64 arrays were > 10 GB of transient memory)
    e_src = csr.indices[beg[e_item] + (torch.arange(E, device=dev) - first[e_item])]  # the CSR's own index type
    e_rank = None
    if pair_rank:  # rank of every edge among the edges of its (row, source) pair, by edge id
        node = (base.item_node if base is not None and base.item_node is not None else item_row).long()
        e_csr = beg[e_item] + (torch.arange(E, device=dev) - first[e_item])
        eid = csr.eids[e_csr].long() if getattr(csr, "eids", None) is not None else e_csr
        o1 = torch.sort(eid)[1]
        pk, o2 = torch.sort((node[e_item] * n_src + e_src)[o1], stable=True)
        order = o1[o2]
        runs = torch.unique_consecutive(pk, return_counts=True)[1]  # (torch.cummax over 115 M elements takes 0.36 s on this stack)
        run0 = torch.repeat_interleave(_excl_cumsum(runs)[:-1], runs)
        e_rank = torch.empty(E, dtype=torch.int32, device=dev)
        e_rank[order] = (torch.arange(E, device=dev) - run0).to(torch.int32)
        del pk, o1, o2, order, runs, run0, node, eid, e_csr
    e_tile, e_pos = it_tile[e_item], pos[e_item].to(torch.int32)
    del e_item, first
    # ---- sources gathered >= tau times inside a tile are staged; per tile they are ordered by multiplicity (dense chunks first)
    skey, perm = torch.sort(e_tile * n_src + e_src.long())
    uniq, inv, counts = torch.unique_consecutive(skey, return_inverse=True, return_counts=True)
    del skey
    g_tile, g_src = uniq // n_src, uniq % n_src
    sel = torch.nonzero(counts >= tau).flatten()
    st, sc = g_tile[sel], counts[sel]
    maxc = int(sc.max()) if sel.numel() else 0
    o2 = torch.sort(st * (maxc + 1) + (maxc - sc), stable=True)[1]
    sel = sel[o2]
    st = g_tile[sel]
    per_tile = torch.bincount(st, minlength=T)
    rank_in_tile = torch.arange(sel.shape[0], device=dev) - _excl_cumsum(per_tile)[st]
    real = CHUNK_SLOTS - 1
    tile_chunk_ptr = _excl_cumsum((per_tile + real - 1) // real)
    NCH = int(tile_chunk_ptr[-1])
    s_chunk = tile_chunk_ptr[st] + rank_in_tile // real
    s_slot = rank_in_tile % real
    chunk_ids = torch.full((NCH * CHUNK_SLOTS,), -1, dtype=torch.int32, device=dev)
    chunk_ids[s_chunk * CHUNK_SLOTS + s_slot] = g_src[sel].to(torch.int32)
    grp_chunk = torch.full((uniq.shape[0],), -1, dtype=torch.int64, device=dev)
    grp_slot = torch.zeros(uniq.shape[0], dtype=torch.int64, device=dev)
    grp_chunk[sel], grp_slot[sel] = s_chunk, s_slot
    # back to storage order of the edges (so that the streams keep it inside every lane group)
    e_chunk = torch.empty(E, dtype=torch.int64, device=dev)
    e_slot = torch.empty(E, dtype=torch.int16, device=dev)   # < 256 LDS slots per chunk
    e_chunk[perm], e_slot[perm] = grp_chunk[inv], grp_slot[inv].to(torch.int16)
    del perm, inv, grp_chunk, grp_slot, uniq, g_tile
    staged = e_chunk >= 0
    # ---- LDS streams: rows (chunk, cw, j), laid out per (tile, consumer wave) -- contiguous over the chunks of the tile, so the
    # kernel's prefetch of the next supersteps runs through chunk boundaries
    per_unit = NC * NACC * GROUPS
    sk = e_chunk[staged] * per_unit + e_pos[staged]
    del e_chunk
    nseg = NCH * NC * NACC
    seg = torch.arange(nseg, device=dev)
    seg_chunk, seg_cw = seg // (NC * NACC), (seg // NACC) % NC
    chunk_tile = torch.repeat_interleave(torch.arange(T, device=dev), tile_chunk_ptr[1:] - tile_chunk_ptr[:-1])
    seg_unit = (chunk_tile[seg_chunk] * NC + seg_cw) if NCH else seg
    seg_order = torch.sort(seg_unit, stable=True)[1]  # stable: (chunk, j) order kept inside a (tile, cw)
    bank_classes = {16: 4, 8: 2}.get(GROUPS, 0)  # 4 lanes per row: 64-byte rows; 8 lanes: 128-byte rows; 16 lanes: whole rows, no conflicts
    lds_cnt, lds_base, lds_total, lds_bytes = _streams(sk, nseg, e_slot[staged], ZERO_SLOT, torch.uint8, seg_order, GROUPS, bank_classes)
    lds_off = _excl_cumsum(torch.bincount(seg_unit, weights=lds_cnt.double(), minlength=T * NC).long()) if NCH else \
        torch.zeros(T * NC + 1, dtype=torch.int64, device=dev)
    lds16 = None
    max_rank = int(e_rank.max()) if e_rank is not None and E else 0
    if e_rank is not None:  # the same stream with 16-bit entries: slot | (rank mod 128) << 8
        lds16 = _streams(sk, nseg, e_slot[staged].long() + ((e_rank[staged].long() & 127) << 8), ZERO_SLOT, torch.int16, seg_order,
                         GROUPS, bank_classes)[3]
    del sk, e_slot
    # ---- direct streams: one per (tile, consumer wave)
    direct = ~staged
    dk = e_tile[direct] * per_unit + e_pos[direct]
    dir_payload = e_src[direct].to(torch.int32)
    if e_rank is not None and n_src < (1 << 24):
        dir_payload = dir_payload | ((e_rank[direct] & 127) << 24)
    del e_tile, e_pos, e_src, direct
    dir_cnt, dir_base, dir_total, dir_stream = _streams(dk, T * NC * NACC, dir_payload, -1, torch.int32, None, GROUPS)
    dir_off = torch.cat([dir_base.view(-1)[::NACC], torch.tensor([dir_total], device=dev)])
    if int(lds_cnt.max() if lds_cnt.numel() else 0) > 65535:
        raise ValueError("tile plan: more than 65535 supersteps in one (chunk, row)")

    def cnt8(c, units, dtype):  # [units * NC, CNT_STRIDE] with NACC live columns
        out = torch.zeros((units * NC, CNT_STRIDE), dtype=dtype, device=dev)
        if units:
            out[:, :NACC] = c.view(units * NC, NACC).to(dtype)
        return out.view(-1)

    n_staged = int(staged.sum())
    stats = {"tiles": T, "rows_per_tile": R, "chunks": NCH, "edges": E, "staged_edges": n_staged, "direct_edges": E - n_staged,
             "staged_sources": int(sel.shape[0]), "lds_supersteps": int(lds_bytes.shape[0]) // (4 * GROUPS) - STREAM_TAIL,
             "dir_supersteps": int(dir_stream.shape[0]) // (4 * GROUPS) - STREAM_TAIL,
             "gathered_rows_per_edge": (int(sel.shape[0]) + E - n_staged) / max(E, 1),
             "lds_slot_fill": n_staged / max(int(lds_bytes.shape[0]) - STREAM_TAIL * GROUPS * 4, 1),
             "dir_slot_fill": (E - n_staged) / max(int(dir_stream.shape[0]) - STREAM_TAIL * GROUPS * 4, 1),
             "parallel_edges": None if e_rank is None else max_rank > 0,
             "max_pair_rank": None if e_rank is None else max_rank,
             "pair_rank_streams": lds16 is not None and n_src < (1 << 24)}
    tables = {
        "tile_chunk_ptr": tile_chunk_ptr.to(torch.int32),
        "chunk_ids": chunk_ids,
        "lds_off": lds_off.to(torch.int32),            # [T * NC + 1] first superstep of (tile, cw): chunks, then rows j, contiguous
        "lds_cnt": cnt8(lds_cnt, NCH, torch.int16),    # supersteps of (chunk, cw, j), 16 uint16 per (chunk, cw)
        "lds_stream": lds_bytes.view(torch.int32) if lds_bytes.numel() else torch.zeros(0, dtype=torch.int32, device=dev),
        "dir_off": dir_off.to(torch.int32),            # [T * NC + 1]
        "dir_cnt": cnt8(dir_cnt, T, torch.int32),      # supersteps of (tile, cw, j), 16 int32 per (tile, cw)
        "dir_stream": dir_stream,                      # [dir_supersteps * 16] source ids, -1 = padding
        "tile_item": tile_item.to(torch.int32),        # [T * R] item_row of the item at every position, NO_ITEM = none
        "zero_row": torch.zeros(64, dtype=torch.float32, device=dev),
        "tile_node": tile_node,
        "lds_stream16": None if lds16 is None else lds16.view(torch.int32),  # [lds_supersteps * groups * 2] words: 4 x (slot | rank << 8)
        "tile_order": None,  # optional dispatch order (ABI field); longest-tile-first was measured and lost (docs/LOG_r01_r03.md)
    }
    if lds_total * 4 >= 2 ** 31 or dir_total * 16 >= 2 ** 31:
        raise ValueError("tile plan: stream offsets exceed 31 bits")
    return TilePlan(base, NC, NACC, loaders, tables, stats, lanes_log2)


def tile_costs(plan, dir_weight=1.6, chunk_cost=2.0):
    """Estimated duration of every tile in superstep units: a workgroup lasts as long as its slowest consumer wave -- its LDS
    supersteps, its direct supersteps (each worth `dir_weight`: rows from L2 / the fabric) -- plus a barrier per chunk."""
    NC, T = plan.consumers, plan.num_tiles
    lds = (plan.lds_off[1:] - plan.lds_off[:-1]).view(T, NC).double()
    dr = (plan.dir_off[1:] - plan.dir_off[:-1]).view(T, NC).double()
    chunks = (plan.tile_chunk_ptr[1:] - plan.tile_chunk_ptr[:-1]).double()
    return (lds + dir_weight * dr).max(dim=1)[0] + chunk_cost * chunks


def simulate_dispatch(cost, slots_per_xcd=64, xcds=8, order=None):
    """The launch as the hardware runs it: workgroup b goes to XCD b % 8, which takes its workgroups in order whenever one of its
    `slots_per_xcd` workgroup slots frees.  cost: per-tile durations in dispatch-slot order (slot = xcd * per + i, as the kernel maps
    blockIdx).  Returns (makespan, sum(cost) / (xcds * slots_per_xcd)): what the launch takes against perfectly divisible work."""
    import heapq
    c = cost.cpu().tolist() if hasattr(cost, "cpu") else list(cost)
    T = len(c)
    per = (T + xcds - 1) // xcds
    worst = 0.0
    for x in range(xcds):
        mine = c[x * per:min((x + 1) * per, T)] if order is None else [c[t] for t in order[x]]
        heap = [0.0] * slots_per_xcd
        for d in mine:
            heapq.heappush(heap, heapq.heappop(heap) + d)
        worst = max(worst, max(heap))
    return worst, sum(c) / (xcds * slots_per_xcd)


def validate(plan, csr):
    """Bounds of every index the kernel follows blindly (a wrong table would be an out-of-bounds access on the device)."""
    NC, NACC, R = plan.consumers, plan.nacc, plan.rows_per_tile
    GROUPS, CHUNK_SLOTS = plan.groups, plan.chunk_slots
    ZERO_SLOT = CHUNK_SLOTS - 1
    T, NCH = plan.num_tiles, plan.num_chunks
    num_slots = plan.base.num_slots if plan.base is not None else 0

    def ok(cond, what):
        if not bool(cond):
            raise ValueError("tile plan: " + what)

    tcp = plan.tile_chunk_ptr.long()
    ok(tcp.shape[0] == T + 1 and int(tcp[0]) == 0 and int(tcp[-1]) == NCH and bool((tcp[1:] >= tcp[:-1]).all()), "tile_chunk_ptr")
    ids = plan.chunk_ids
    ok(ids.shape[0] == NCH * CHUNK_SLOTS and (NCH == 0 or (int(ids.min()) >= -1 and int(ids.max()) < csr.num_cols)), "chunk_ids range")
    ok(NCH == 0 or bool((ids.view(NCH, CHUNK_SLOTS)[:, ZERO_SLOT] == -1).all()), "zero slot")
    for off, cnt, stream, units, per in ((plan.lds_off, plan.lds_cnt, plan.lds_stream, NCH, GROUPS),
                                         (plan.dir_off, plan.dir_cnt, plan.dir_stream, T, 4 * GROUPS)):
        off = off.long()
        ok(off.shape[0] == T * NC + 1 and int(off[0]) == 0 and bool((off[1:] >= off[:-1]).all()), "stream offsets")
        c = cnt.view(units * NC, CNT_STRIDE).long() & 0xFFFFFFFF if cnt.dtype == torch.int32 else cnt.view(units * NC, CNT_STRIDE).long() & 0xFFFF
        ok(bool((c[:, NACC:] == 0).all()), "superstep counts beyond nacc")
        per_wave = c.sum(1).view(units, NC)  # supersteps of (unit, cw)
        if units == NCH:  # chunks of a tile add up to the tile's stream
            per_tile = torch.zeros((T, NC), dtype=torch.int64, device=c.device)
            if NCH:
                per_tile.index_add_(0, torch.repeat_interleave(torch.arange(T, device=c.device), tcp[1:] - tcp[:-1]), per_wave)
            per_wave = per_tile
        ok(bool((per_wave.view(-1) == off[1:] - off[:-1]).all()), "superstep counts")
        ok(stream.shape[0] == (int(off[-1]) + STREAM_TAIL) * per, "stream length")
    if plan.lds_stream.numel():
        b = plan.lds_stream.view(torch.uint8)
        ok(int(b.max()) <= ZERO_SLOT, "lds slots")
    if plan.dir_stream.numel():
        ids = plan.dir_stream
        if plan.stats.get("pair_rank_streams"):  # bits 24-30: the edge's rank among parallel edges (gat_tile.inc masks them off)
            ids = torch.where(ids >= 0, ids & 0xFFFFFF, ids)
        ok(int(ids.min()) >= -1 and int(ids.max()) < csr.num_cols, "direct ids")
    ti = plan.tile_item.long()
    live = ti[ti != NO_ITEM]
    ok(ti.shape[0] == T * R and (live.numel() == 0 or (int(live.max()) < csr.num_rows and int(live.min()) >= -num_slots)), "tile_item")
    ok(plan.zero_row.numel() == 64 and not bool(plan.zero_row.any()), "zero row")
    if getattr(plan, "tile_order", None) is not None:
        ok(bool((torch.sort(plan.tile_order.long())[0] == torch.arange(T, device=plan.tile_order.device)).all()), "tile_order is a permutation")
    return True


def emulate(plan, x, out_rows, num_slots=0):
    """Walks a tile plan on the host exactly as the kernel does (same streams, same padding) and returns
    (out [out_rows, D], partial [num_slots, D]) in float64 -- the structural check of build_tile_plan used by tests/ (CPU)."""
    import numpy as np
    x = x.detach().cpu().double().numpy()
    D = x.shape[1]
    NC, NACC, R = plan.consumers, plan.nacc, plan.rows_per_tile
    GROUPS, CHUNK_SLOTS = plan.groups, plan.chunk_slots
    ZERO_SLOT = CHUNK_SLOTS - 1
    tcp = plan.tile_chunk_ptr.cpu().numpy()
    ids = plan.chunk_ids.cpu().numpy()
    lds_off, lds_cnt = plan.lds_off.cpu().numpy(), plan.lds_cnt.cpu().numpy().astype(np.uint16).reshape(-1, CNT_STRIDE)
    lds_stream = plan.lds_stream.cpu().numpy().view(np.uint8).reshape(-1, GROUPS, 4)  # [superstep][group][step]
    dir_off, dir_cnt = plan.dir_off.cpu().numpy(), plan.dir_cnt.cpu().numpy().reshape(-1, CNT_STRIDE)
    dir_stream = plan.dir_stream.cpu().numpy().reshape(-1, GROUPS, 4)
    if plan.stats.get("pair_rank_streams"):
        dir_stream = np.where(dir_stream >= 0, dir_stream & 0xFFFFFF, dir_stream)
    tile_item = plan.tile_item.cpu().numpy()
    out = np.zeros((out_rows, D))
    partial = np.zeros((max(num_slots, 1), D))
    for t in range(plan.num_tiles):
        acc = np.zeros((R, D))
        images = []
        for c in range(tcp[t], tcp[t + 1]):
            lds = np.zeros((CHUNK_SLOTS, D))
            cid = ids[c * CHUNK_SLOTS:(c + 1) * CHUNK_SLOTS]
            assert cid[ZERO_SLOT] == -1
            lds[cid >= 0] = x[cid[cid >= 0]]
            images.append(lds)
        for cw in range(NC):
            ss = lds_off[t * NC + cw]
            for c in range(tcp[t], tcp[t + 1]):
                lds = images[c - tcp[t]]
                for j in range(NACC):
                    for _ in range(int(lds_cnt[c * NC + cw, j])):
                        for g in range(GROUPS):
                            for u in range(4):
                                acc[(cw * NACC + j) * GROUPS + g] += lds[lds_stream[ss, g, u]]
                        ss += 1
            assert ss == lds_off[t * NC + cw + 1]
        for cw in range(NC):
            k = t * NC + cw
            ss = dir_off[k]
            for j in range(NACC):
                for _ in range(int(dir_cnt[k, j])):
                    for g in range(GROUPS):
                        for u in range(4):
                            sid = dir_stream[ss, g, u]
                            if sid >= 0:
                                acc[(cw * NACC + j) * GROUPS + g] += x[sid]
                    ss += 1
            assert ss == dir_off[k + 1]
        for p in range(R):
            it = tile_item[t * R + p]
            if it == NO_ITEM:
                assert not acc[p].any()
            elif it >= 0:
                out[it] += acc[p]
            else:
                partial[-(it + 1)] += acc[p]
    return out, partial


TILE_SPLIT = _settings.TILE_HUB_SPLIT  # hub threshold of the tile kernel's work items


def config(lanes_log2=4):
    """(consumers, nacc, loaders, tau) of the 64- / 32- / 16-column kernel: 8-wave workgroups (7 consumers + 1 loader), two or three
    per CU; 168 / 168 / 336 items per tile (profiles/r03_tile_narrow.txt).  The constants live in mi355x_graph/config.py."""
    return tuple(_settings.TILE_CONFIG[int(lanes_log2)])


def gat_config():
    """(consumers, nacc, loaders, tau) of the tile plans behind the fused GAT walks (gat_tile.inc: 7 + 1 waves, nacc 3 or 4)."""
    return tuple(_settings.GAT_TILE_CONFIG)


def lanes_log2_for(width):
    """Lanes per feature row (log2) of the tile kernel that takes rows of `width` columns in the fewest passes: 4 lanes x float4
    (16-column passes) up to 16 columns, 8 lanes up to 32, 16 lanes (64-column passes) beyond."""
    return 2 if width <= 16 else (3 if width <= 32 else 4)


def tile_plan_wanted(csr):
    """Policy: dense neighbourhoods only -- the kernel pays when a staged source is re-used often enough inside a tile.
    MGX_TILE=0 disables, =1 forces (tests)."""
    mode = os.environ.get("MGX_TILE", "auto")
    if mode == "0" or csr.idx_bits != 32 or not csr.indptr.is_cuda or csr.nnz == 0:
        return False
    if mode == "1":
        return True
    return csr.nnz >= (1 << 22) and csr.nnz >= 128 * csr.num_rows
