"""Execution schedules (mgx_spmm_plan) for the row-segmented g-SpMM: hub-row splitting and a
locality-aware row order.  Built once per CSR and cached on it -- preprocessing, like the CSR build
that the reference's cold-start repetitions hide (kernel/dgl-new.py:8,21).

Why: the gather of neighbour rows is the whole cost of copy_u/sum on ogbn-products-sized graphs
(E*D*4 = 31.7 GB of row reads against 1.76 GB of compulsory traffic).  Rows of one
community read the same source rows; walking them back to back on ONE XCD turns those reads into hits
in that XCD's 4 MiB L2.  Node ids are untouched: only the order in which destination rows are
processed changes, so results and the DGL-visible numbering are unaffected.

  * split: rows with more than `split` edges become several work items whose partial sums are
    combined in a fixed order (deterministic), so a 17k-edge hub never serialises on one wavefront;
  * order: semi-synchronous label propagation (a few rounds of sort + run-length on the device)
    groups rows into clusters (8 rounds: config.LP_ROUNDS); the schedule is the rows sorted by (final label, earlier labels).
"""
import ctypes
import os

import torch

from ._lib import DGLError
from . import config


class MgxSpmmPlan(ctypes.Structure):
    _fields_ = [
        ("num_items", ctypes.c_int64),
        ("item_row", ctypes.c_void_p),
        ("item_beg", ctypes.c_void_p),
        ("item_end", ctypes.c_void_p),
        ("num_hubs", ctypes.c_int64),
        ("hub_row", ctypes.c_void_p),
        ("hub_slot_ptr", ctypes.c_void_p),
        ("num_slots", ctypes.c_int64),
        ("slot_item", ctypes.c_void_p),
        ("item_node", ctypes.c_void_p),
        ("xcd_item_start", ctypes.c_int64 * 9),
        ("xcd_item_start_dev", ctypes.c_void_p),
        ("rest", ctypes.c_void_p),
    ]


class SpmmPlan(object):
    def __init__(self, item_row, item_beg, item_end, hub_row, hub_slot_ptr, num_slots, order_kind, slot_item=None,
                 item_node=None):
        self.item_row, self.item_beg, self.item_end = item_row, item_beg, item_end
        self.hub_row, self.hub_slot_ptr, self.num_slots = hub_row, hub_slot_ptr, int(num_slots)
        self.slot_item = slot_item
        self.item_node = item_node
        self.order_kind = order_kind
        self.rest = None  # second part of a two-part plan (split_short_items): long items + the hub tables
        self._c = None
        self.xcd_item_start = self._balance_xcds()
        self._xcd_dev = (torch.tensor(self.xcd_item_start, dtype=torch.int64, device=item_row.device)
                         if self.xcd_item_start[-1] and item_row.is_cuda else None)

    def _balance_xcds(self, xcds=8, granule=16):
        """Cut the schedule into 8 contiguous stretches of equal EDGE count (one per XCD), boundaries on multiples of the
        workgroup's item count.  One device cumsum + 7 searches + one host read, once per plan."""
        n = int(self.item_row.shape[0])
        if n < xcds * granule * 4:
            return [0] * (xcds + 1)
        ln = (self.item_end - self.item_beg).to(torch.int64) + 1  # +1: an empty item still costs its fixed overhead
        csum = torch.cumsum(ln, 0)
        targets = (torch.arange(1, xcds, device=csum.device, dtype=torch.float64) * (float(csum[-1]) / xcds)).to(torch.int64)
        cuts = (torch.searchsorted(csum, targets) // granule * granule).clamp(max=n).tolist()
        out = [0] + [int(c) for c in cuts] + [n]
        for i in range(1, len(out)):
            out[i] = max(out[i], out[i - 1])
        return out

    @property
    def num_items(self):
        return int(self.item_row.shape[0])

    @property
    def num_hubs(self):
        return int(self.hub_row.shape[0])

    @property
    def total_slots(self):
        """Rows of the partial workspace a g-SpMM over this plan needs (its own slots + those of its `rest` part)."""
        return self.num_slots + (self.rest.total_slots if self.rest is not None else 0)

    def c_struct(self):
        if self._c is None:
            self._c = MgxSpmmPlan(self.num_items, self.item_row.data_ptr(), self.item_beg.data_ptr(),
                                  self.item_end.data_ptr(), self.num_hubs,
                                  self.hub_row.data_ptr() if self.num_hubs else None,
                                  self.hub_slot_ptr.data_ptr() if self.num_hubs else None, self.num_slots,
                                  self.slot_item.data_ptr() if self.num_slots else None,
                                  None if self.item_node is None else self.item_node.data_ptr(),
                                  (ctypes.c_int64 * 9)(*self.xcd_item_start),
                                  None if self._xcd_dev is None else self._xcd_dev.data_ptr(),
                                  None if self.rest is None else ctypes.addressof(self.rest.c_struct()))
        return self._c


SHORT_ITEM_EDGES = 32     # the longest item a lane group of spmm_rowgroup32_kernel walks alone (kLong in csrc/spmm.hip)
MIN_SHORT_ITEMS = 100_000  # fewer short items than this do not pay for a second launch (see split_short_items)


def short_item_limit(lane_groups):
    """Edges up to which an item counts as short when `lane_groups` (= 64 / G) items share a wave: 4 per lane group of the wave, at most
    32.  Measured on the products graph (experiments/exp_two_part_products.py): D = 64 (4 lane groups) best at 16 (8: -1.7 %, 16: -5.4 %,
    24: -4.8 %, 32: -4.1 % of the one-launch time), D = 16 (16 lane groups) at 24 .. 32 (-18 %), D = 100 (2 lane groups) at none."""
    return min(SHORT_ITEM_EDGES, 4 * int(lane_groups))


def split_short_items(csr, base, any_share=False, limit=SHORT_ITEM_EDGES):
    """The schedule of `csr` (`base`, or its natural rows when it has no plan) as a TWO-PART plan (mgx_spmm_plan::rest): the direct items
    of at most `limit` edges, in schedule order, for the lane-group kernel; every other item -- longer rows, the 256-edge chunks of
    split rows with the hub tables -- in `rest` for the wave-per-item kernel.  Returns (plan, short_lengths, short_edges); plan is
    `base` itself (may be None) when nothing is long.  None when something is long but fewer than MIN_SHORT_ITEMS items are short:
    what the lane-group kernel saves grows with the number of short items (0.05 - 0.2 ns each: arxiv x 1 165 k items -8 us, arxiv x 4
    662 k -120 us, products 1.3 M -110 us) and the second launch costs 10 - 20 us whatever it holds (reddit-small / 10, 18 k short
    items: 0.03 -> 0.05 ms); nothing is materialised then (any_share=True: experiments).  One host read."""
    dev = csr.indptr.device
    if base is not None:
        row, beg, end, node = base.item_row, base.item_beg, base.item_end, base.item_node
    else:
        row = torch.arange(csr.num_rows, dtype=torch.int32, device=dev)
        beg, end, node = csr.indptr[:-1], csr.indptr[1:], row
    lens = end - beg
    long_ = (lens > limit) | (row < 0)
    n_long, long_edges = [int(v) for v in torch.stack([long_.sum(), (lens * long_).sum()]).tolist()]
    n_items = int(lens.shape[0])
    n_short, short_edges = n_items - n_long, csr.nnz - long_edges
    if n_short == 0 or (n_long > 0 and n_short < MIN_SHORT_ITEMS and not any_share):
        return None
    if n_long == 0:
        return base, lens, short_edges
    keep = ~long_
    kind = base.order_kind if base is not None else "natural"
    i32 = lambda k: torch.zeros(k, dtype=torch.int32, device=dev)
    head = SpmmPlan(row[keep].contiguous(), beg[keep].contiguous(), end[keep].contiguous(), i32(0), i32(1), 0, kind, i32(0), node[keep].contiguous())
    if base is not None and base.num_slots:
        new_index = torch.cumsum(long_.to(torch.int64), 0) - 1
        hub_row, hub_ptr, slots = base.hub_row, base.hub_slot_ptr, base.num_slots
        slot_item = new_index[base.slot_item.long()].to(torch.int32).contiguous()
    else:
        hub_row, hub_ptr, slots, slot_item = i32(0), i32(1), 0, i32(0)
    head.rest = SpmmPlan(row[long_].contiguous(), beg[long_].contiguous(), end[long_].contiguous(), hub_row, hub_ptr, slots, kind, slot_item,
                         node[long_].contiguous())
    return head, lens[keep], short_edges


def label_propagation(indptr, indices, n, rounds=5, seed=0, node_w=None, max_weight=None):
    """Semi-synchronous LP on a square CSR; returns the label history [(n,) int64 per round].
    With `max_weight`, a node may not join a cluster whose weight (sum of node_w) already reached it,
    which keeps clusters from snowballing into one giant component on power-law graphs."""
    dev = indptr.device
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    cols = indices.long()
    if cols.numel() and int(cols.max().item()) >= n:  # block graph [dst-prefix | extra sources]: cluster the square part
        keep = cols < n
        rows, cols = rows[keep], cols[keep]
    labels = torch.arange(n, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    history = []
    for it in range(rounds):
        key = rows * n + labels[cols]
        key, _ = torch.sort(key)
        ukey, cnt = torch.unique_consecutive(key, return_counts=True)
        del key
        urow = torch.div(ukey, n, rounding_mode="floor")
        ulab = ukey - urow * n
        score = (cnt << 32) | ulab  # unique per row: the larger label wins ties -> deterministic
        best = torch.zeros(n, dtype=torch.int64, device=dev)
        best.scatter_reduce_(0, urow, score, reduce="amax", include_self=True)
        new = best & 0xFFFFFFFF
        has = best > 0
        # update a random half of the nodes per round (plain synchronous LP oscillates)
        flip = torch.rand(n, generator=gen, device=dev) < (0.5 if it < rounds - 1 else 1.1)
        upd = has & flip
        if max_weight is not None:
            w = node_w if node_w is not None else torch.ones(n, dtype=torch.int64, device=dev)
            cw = torch.zeros(n, dtype=w.dtype, device=dev).index_add_(0, labels, w)
            upd &= (cw[new] < max_weight) | (new == labels)
        labels = torch.where(upd, new, labels)
        history.append(labels.clone())
    return history


def locality_order(csr, rounds=None):
    """Row permutation placing rows of one (nested) cluster next to each other."""
    n = csr.num_rows
    if n > csr.num_cols:
        raise DGLError("locality_order needs the destination nodes to be a prefix of the source nodes")
    hist = label_propagation(csr.indptr, csr.indices, n, config.LP_ROUNDS if rounds is None else rounds)
    order = torch.arange(n, device=csr.device)
    # stable sorts from the finest (earliest) to the coarsest (final) labels = lexicographic order
    for labels in hist[max(0, len(hist) - 3):]:
        order = order[torch.sort(labels[order], stable=True)[1]]
    return order


def _build_plan_device(csr, order, split, order_kind):
    """mgx_spmm_plan_count / _fill: the tables are built by the library (csrc/plan.hip); one host read of the totals."""
    import ctypes
    from . import _lib
    from .sparse import _ptr, _stream
    dev, n, idt = csr.device, csr.num_rows, csr.indptr.dtype
    L = _lib.lib()
    with torch.cuda.device(dev):
        ws_bytes = L.mgx_spmm_plan_workspace(n)
        if ws_bytes < 0:
            _lib.check(3)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        totals = torch.empty(3, dtype=torch.int64, device=dev)
        ordr = None if order is None else order.to(idt).contiguous()
        _lib.check(L.mgx_spmm_plan_count(ctypes.byref(csr.c_struct()), split, _ptr(ordr), _ptr(totals), _ptr(ws), ws_bytes, _stream(dev)))
        items, hubs, slots = [int(v) for v in totals.tolist()]
        i32 = lambda k: torch.empty(max(k, 1), dtype=torch.int32, device=dev)[:k]
        item_row, item_node = i32(items), i32(items)
        item_beg = torch.empty(items, dtype=idt, device=dev)
        item_end = torch.empty(items, dtype=idt, device=dev)
        hub_row, hub_ptr, slot_item = i32(hubs), torch.zeros(hubs + 1, dtype=torch.int32, device=dev), i32(slots)
        _lib.check(L.mgx_spmm_plan_fill(ctypes.byref(csr.c_struct()), split, _ptr(ordr), _ptr(item_row), _ptr(item_beg), _ptr(item_end),
                                        _ptr(item_node), _ptr(hub_row), _ptr(hub_ptr), _ptr(slot_item), _ptr(ws), ws_bytes, _stream(dev)))
    return SpmmPlan(item_row, item_beg, item_end, hub_row, hub_ptr, slots, order_kind, slot_item, item_node)


def build_plan(csr, order=None, split=1024, order_kind="natural"):
    if csr.indptr.is_cuda and config.PLAN_BUILDER == "device":
        return _build_plan_device(csr, order, split, order_kind)
    dev = csr.device
    n = csr.num_rows
    if n >= 2 ** 31:
        raise DGLError("schedules support fewer than 2^31 rows")
    indptr = csr.indptr.long()
    deg = indptr[1:] - indptr[:-1]
    if order is None:
        order = torch.arange(n, device=dev)
    nchunk = torch.clamp((deg + split - 1) // split, min=1)
    nchunk_o = nchunk[order]
    num_items = int(nchunk_o.sum().item())
    if num_items == n:  # nothing to split
        rows = order
        beg, end = indptr[rows], indptr[rows + 1]
        item_row = rows.to(torch.int32)
        hub_row = torch.zeros(0, dtype=torch.int32, device=dev)
        hub_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
        slots = 0
        slot_item = torch.zeros(0, dtype=torch.int32, device=dev)
    else:
        item_off = torch.cumsum(nchunk_o, 0) - nchunk_o
        pos = torch.repeat_interleave(torch.arange(n, device=dev), nchunk_o)
        chunk = torch.arange(num_items, device=dev) - item_off[pos]
        rows = order[pos]
        beg = indptr[rows] + chunk * split
        end = torch.minimum(beg + split, indptr[rows + 1])
        is_hub = nchunk[rows] > 1
        slot = torch.cumsum(is_hub.long(), 0) - 1
        item_row = torch.where(is_hub, -(slot + 1), rows).to(torch.int32)
        hub_pos = torch.nonzero(nchunk_o > 1).flatten()
        hub_row = order[hub_pos].to(torch.int32)
        hub_ptr = torch.zeros(hub_pos.shape[0] + 1, dtype=torch.int64, device=dev)
        torch.cumsum(nchunk_o[hub_pos], 0, out=hub_ptr[1:])
        slots = int(hub_ptr[-1].item())
        hub_ptr = hub_ptr.to(torch.int32)
        slot_item = torch.nonzero(is_hub).flatten().to(torch.int32)  # slots are numbered in item order
    idt = csr.indptr.dtype
    return SpmmPlan(item_row.contiguous(), beg.to(idt).contiguous(), end.to(idt).contiguous(), hub_row.contiguous(),
                    hub_ptr.contiguous(), slots, order_kind, slot_item.contiguous(), rows.to(torch.int32).contiguous())


# nnz below which the whole gathered matrix is cache resident anyway and clustering cannot pay



_SMALL_NNZ = 200_000


def _no_hubs_cheaply(csr):
    """Small graphs (sampled blocks, batched molecules): one fused max over the degrees, one host sync."""
    deg = csr.indptr[1:] - csr.indptr[:-1]
    return int(deg.max().item()) <= config.HUB_SPLIT


def plan_for(csr, split=None):
    """Default policy: always split hubs; cluster the row order only for big square graphs.
    MGX_SCHEDULE=natural|cluster|none overrides (none: no plan at all).  `split` overrides the hub threshold (the row
    order computed for the first plan of a CSR is reused)."""
    mode = os.environ.get("MGX_SCHEDULE", "auto")
    if mode == "none" or csr.num_rows == 0 or csr.nnz == 0:
        return None
    if mode == "auto" and csr.nnz < _SMALL_NNZ and csr.nnz // max(csr.num_rows, 1) < 64 and _no_hubs_cheaply(csr):
        return None
    if split is None:
        split = config.HUB_SPLIT
    want_cluster = mode == "cluster" or (mode == "auto" and csr.nnz >= config.CLUSTER_MIN_NNZ)
    order, kind = csr._row_order
    if kind is None:
        order, kind = None, "natural"
        square_like = csr.num_rows == csr.num_cols or getattr(csr, "dst_is_src_prefix", False)
        if want_cluster and square_like and csr.indices.numel():
            order, kind = locality_order(csr), "cluster"
        csr._row_order = (order, kind)
    plan = build_plan(csr, order, split, kind)
    if kind == "natural" and plan.num_hubs == 0:
        return None  # natural order, nothing split: the plan-free path is identical and leaner
    return plan
