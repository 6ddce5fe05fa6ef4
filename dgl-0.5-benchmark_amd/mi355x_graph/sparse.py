"""Raw (non-autograd) sparse primitives on torch tensors, dispatched to the HIP C ABI.

Mirrors the role of DGL's python/dgl/sparse.py (_gspmm / _gsddmm, UPSTREAM) for the call sites
kernel/dgl-new.py:20,39 and main_dgl_product_sage.py:62: shape/broadcast bookkeeping and output
allocation happen here (caller-allocates convention of the seam), the arithmetic happens in
csrc/*.hip.  Tensors on a HIP device go to `HipBackend`; there is no built-in backend for CPU
tensors -- message-passing arithmetic on CPU tensors raises DGLError.  (tests/ registers a checker
backend for multi-process gloo tests; the product never does.)
"""
import ctypes
import os
import functools

import numpy as np
import torch

from . import _lib
from ._lib import DGLError, MgxCsr, OP, REDUCE, TARGET
from . import config


# Every tile plan is bounds-checked once when it is built (the kernels follow its tables blindly: a wrong entry would be an
# out-of-bounds access on the device).  A dozen reductions with host syncs per plan; MGX_TILE_VALIDATE=0 skips them.
PROFILE = None  # set to a list by bench.py to collect per-launch HIP-event timings of mgx_spmm_csr


class timed_call(object):
    """`with timed_call(dev, kernel=..., **meta):` around ONE C-ABI call: when bench.py has set PROFILE to a list, HIP events on the
    launch stream bracket the call and a record {kernel, meta..., start, end} is appended (the g-SpMM entry points below build
    theirs inline, with the same keys).  Costs one attribute read otherwise."""
    __slots__ = ("rec", "dev")

    def __init__(self, dev, **meta):
        self.rec, self.dev = (meta if PROFILE is not None else None), dev

    def __enter__(self):
        if self.rec is not None:
            self.rec["start"] = torch.cuda.Event(enable_timing=True)
            self.rec["end"] = torch.cuda.Event(enable_timing=True)
            self.rec["start"].record(torch.cuda.current_stream(self.dev))
        return self

    def __exit__(self, *exc):
        if self.rec is not None and exc[0] is None and PROFILE is not None:
            self.rec["end"].record(torch.cuda.current_stream(self.dev))
            PROFILE.append(self.rec)
        return False


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _prod(shape):
    p = 1
    for s in shape:
        p *= int(s)
    return p


class CsrView(object):
    """An immutable CSR over torch tensors (in-CSR: rows = destination nodes)."""

    __slots__ = ("__weakref__", "num_rows", "num_cols", "indptr", "indices", "eids", "_c", "_deg", "_inv_deg", "_plan", "_sm_plan",
                 "_row_order", "dst_is_src_prefix", "_tile_plan", "_short", "short_hint", "_rows", "_gate")

    def __init__(self, num_rows, num_cols, indptr, indices, eids):
        self.num_rows, self.num_cols = int(num_rows), int(num_cols)
        self.indptr, self.indices, self.eids = indptr, indices, eids
        self._c = None
        self._deg = None
        self._inv_deg = None
        self._plan = False  # False = not built yet; None = run without a plan
        self._sm_plan = False
        self._tile_plan = False
        self._short = {}        # lane groups per wave -> None | True | two-part SpmmPlan (see _short_choice); "split": its tables
        self.short_hint = None  # the owner's word on short rows where reading the lengths back is impossible (graph capture) or not
                                # worth it (a sampled block): True / False, or the average row length (see _short_choice)
        self._row_order = (None, None)  # (row order, kind) computed with the first plan
        self._rows = None  # row id of every stored position (the expanded indptr), built on first use by the permuted g-SDDMM walk
        self._gate = None  # ops._SlotGate of untagged 64-column operands aggregated over this CSR (ops._gspmm_over_slots)
        self.dst_is_src_prefix = False  # block graphs whose destination nodes are the first source nodes

    @property
    def nnz(self):
        return int(self.indices.shape[0])

    @property
    def device(self):
        return self.indptr.device

    @property
    def idx_bits(self):
        return 32 if self.indptr.dtype == torch.int32 else 64

    def c_struct(self):
        if self._c is None:
            self._c = MgxCsr(self.num_rows, self.num_cols, self.nnz, self.indptr.data_ptr(),
                             self.indices.data_ptr() if self.indices.numel() else None,
                             None if self.eids is None else self.eids.data_ptr(), self.idx_bits, 0)
        return self._c

    def row_of_position(self):
        """[nnz] the row of every stored entry: with `indices` and `eids` the CSR as an edge list in ITS order (sorted by row)."""
        if self._rows is None:
            ids = torch.arange(self.num_rows, dtype=self.indptr.dtype, device=self.indptr.device)
            self._rows = torch.repeat_interleave(ids, (self.indptr[1:] - self.indptr[:-1]).long(), output_size=self.nnz)
        return self._rows

    def degrees(self):
        if self._deg is None:
            if self.indptr.is_cuda:
                self._deg = backend_for(self.indptr).degrees(self)
            else:  # integer graph preparation on the host, like the reference (bit-exact either way)
                self._deg = self.indptr[1:] - self.indptr[:-1]
        return self._deg

    def inv_degrees(self):
        """1 / max(deg, 1) as fp32 -- the factor of fn.mean (main_dgl_product_sage.py:62)."""
        if self._inv_deg is None:
            self._inv_deg = backend_for(self.indptr).inv_degrees(self)
        return self._inv_deg

    def short_rows(self, width):
        """True when the summing g-SpMM of `width` columns gives the work items a LANE GROUP each (see spmm_plan_for)."""
        return self._short_choice(width) is not None

    def spmm_plan_for(self, width):
        """(plan, short) for a full-width copy_u / copy_e sum of `width` columns: the CSR's schedule and False, or -- short rows -- the
        plan to pass with MGX_SPMM_SHORT_ROWS set: the schedule itself when no item is long, else its two-part form (mgx_spmm_plan::rest)."""
        choice = self._short_choice(width)
        if choice is None:
            return self.plan(), False
        return (self.plan() if choice is True else choice), True

    def _short_choice(self, width):
        """None: a wave per work item (spmm_rowwave32_kernel).  Otherwise the SHORT work items -- at most schedule.short_item_limit(B)
        edges, B = 64 / G the items a wave walks side by side, G = lanes per feature row -- get a lane group each
        (spmm_rowgroup32_kernel): True = every item is short, pass the CSR's own schedule; a SpmmPlan = pass this two-part plan, whose
        `rest` (longer rows, chunks of split rows) keeps the wave-per-item kernel in a second launch of the same call.
        Taken when (measurements: docs/LOG_r04.md section 2)
          enough items are short  all of them, or at least schedule.MIN_SHORT_ITEMS (what the second launch has to pay for),
          short on average        the short items average fewer than 3 edges per lane group of the wave, at most 16,
          and even                sum over batches of B consecutive short items of max(length) * B <= MGX_ROWGROUP_IMBALANCE (6) x their
                                  edges: a wave lasts as long as the longest item of its batch.
        Decided once per CSR and lane-group count (two host reads); a structure rebuilt on the device inside a captured step cannot be
        read back and says so itself through `short_hint` (graph_classification.GraphedBatchTrainer: molecules, degree <= 6)."""
        lanes, G = (int(width) + 3) // 4, 1
        while G < lanes and G < 64:
            G *= 2
        nb = 64 // G
        if nb < 2 or self.idx_bits != 32 or not self.indptr.is_cuda or self.nnz == 0 or width % 4:
            return None
        if self.short_hint is not None:
            if self.short_hint is True or self.short_hint is False:
                return True if self.short_hint else None
            # a number: the average row length of a structure that lives for one step (a sampled block) -- the average rule alone,
            # from host-known sizes; rows above 32 edges are taken by the whole wave inside the kernel
            return True if float(self.short_hint) < (16.0 if nb >= 8 else 3.0 * nb) else None
        if nb not in self._short:
            if torch.cuda.is_current_stream_capturing():
                return None  # the lengths cannot be read back inside a capture: the wave-per-item kernel, nothing cached
            from . import schedule
            limit = schedule.short_item_limit(nb)
            if ("split", limit) not in self._short:
                self._short[("split", limit)] = schedule.split_short_items(self, self.plan(), limit=limit)
            split, choice = self._short[("split", limit)], None
            if split is not None:
                plan, lens, edges = split
                n_short = int(lens.shape[0])
                pad = (-n_short) % nb
                padded = torch.cat([lens, lens.new_zeros(pad)]) if pad else lens
                worst = float(padded.view(-1, nb).max(dim=1)[0].sum()) * nb  # one host read
                if edges / n_short < (16.0 if nb >= 8 else 3.0 * nb) and worst <= config.SHORT_ROWS_IMBALANCE * edges:
                    choice = True if (plan is None or plan.rest is None) else plan
            self._short[nb] = choice
        return self._short[nb]

    def plan(self):
        """Execution schedule for the summing g-SpMM (schedule.py); built on first use, device only."""
        if self._plan is False:
            from . import schedule
            self._plan = schedule.plan_for(self) if self.indptr.is_cuda else None
        return self._plan

    def _tile_base_plan(self):
        """The work items every tile plan of this CSR is cut from (hub rows split at tileplan.TILE_SPLIT edges): built once."""
        from . import schedule, tileplan
        if "base" not in self._tile_plan:
            self._tile_plan["base"] = schedule.plan_for(self, split=tileplan.TILE_SPLIT)
        return self._tile_plan["base"]

    def gat_tile_plan(self, F):
        """Tile plan for the fused GAT walks over this CSR (one head of 4 .. 16 columns, dense neighbourhoods: gat_tile.inc), else None.
        Its own plan: 4 lanes per row with the node of every position and every entry's rank among parallel edges (attn_drop keys)."""
        from . import tileplan
        if F % 4 != 0 or not 4 <= F <= 16 or os.environ.get("MGX_GAT_TILE", "1") == "0" or self.num_cols >= (1 << 24):
            return None
        if self._tile_plan is False:
            self._tile_plan = {} if tileplan.tile_plan_wanted(self) else None
        if self._tile_plan is None:
            return None
        if "gat" not in self._tile_plan:
            from . import schedule
            self.plan()
            nc, nacc, nl, tau = tileplan.gat_config()
            base = self._tile_base_plan()
            held = tileplan.build_tile_plan(self, base, nc, nacc, nl, tau, lanes_log2=2, pair_rank=True)
            if config.TILE_VALIDATE:
                tileplan.validate(held, self)
            self._tile_plan["gat"] = held
        return self._tile_plan["gat"]

    def tile_plan(self, width=64):
        """Tile plan of the LDS-staged g-SpMM (tileplan.py) for rows of `width` columns on graphs with dense neighbourhoods, else
        None; built on first use, one per kernel geometry (64- / 32- / 16-column passes)."""
        from . import tileplan
        lg = tileplan.lanes_log2_for(width)
        if self._tile_plan is False:
            self._tile_plan = {} if tileplan.tile_plan_wanted(self) else None
        if self._tile_plan is None:
            return None
        if lg not in self._tile_plan:
            from . import schedule
            self.plan()  # computes (and caches) the locality row order first
            nc, nacc, nl, tau = tileplan.config(lg)
            base = self._tile_base_plan()
            tp = tileplan.build_tile_plan(self, base, nc, nacc, nl, tau, lanes_log2=lg)
            if config.TILE_VALIDATE:
                tileplan.validate(tp, self)
            self._tile_plan[lg] = tp
        return self._tile_plan[lg]

    def softmax_plan(self):
        """Schedule of the edge-softmax / fused attention kernels: the g-SpMM plan."""
        return self.plan()

    def to(self, device):
        out = CsrView(self.num_rows, self.num_cols, self.indptr.to(device), self.indices.to(device),
                      None if self.eids is None else self.eids.to(device))
        out.short_hint = self.short_hint  # a property of the structure, not of where it lives
        return out

    def astype(self, dtype):
        if self.indptr.dtype == dtype:
            return self
        out = CsrView(self.num_rows, self.num_cols, self.indptr.to(dtype), self.indices.to(dtype),
                      None if self.eids is None else self.eids.to(dtype))
        out.short_hint = self.short_hint
        return out


# ----------------------------------------------------------------------------- broadcasting
def _is_head_bcast(shape, oshape):
    """True when `shape` == oshape[:j] + (1,)*(n-j): element k of the output uses operand element
    k // (out_len / len) -- the (N,H,F) x (E,H,1) pattern; expressed to the C ABI as a NULL table."""
    if len(shape) != len(oshape):
        return False
    j = len(shape)
    while j > 0 and shape[j - 1] == 1:
        j -= 1
    return tuple(shape[:j]) == tuple(oshape[:j])


@functools.lru_cache(maxsize=256)
def _bcast_plan(lshape, rshape):
    nd = max(len(lshape), len(rshape))
    ls = (1,) * (nd - len(lshape)) + tuple(lshape)
    rs = (1,) * (nd - len(rshape)) + tuple(rshape)
    out = []
    for a, b in zip(ls, rs):
        if a != b and a != 1 and b != 1:
            raise DGLError("Feature shapes %s and %s are not broadcastable" % (lshape, rshape))
        out.append(max(a, b))
    out = tuple(out)

    def table(shape):
        if shape == out or _is_head_bcast(shape, out):
            return None
        idx = np.indices(out).reshape(nd, -1)
        strides = np.ones(nd, np.int64)
        for d in range(nd - 2, -1, -1):
            strides[d] = strides[d + 1] * shape[d + 1]
        off = np.zeros(idx.shape[1], np.int64)
        for d in range(nd):
            if shape[d] != 1:
                off += idx[d] * strides[d]
        return off

    return out, table(ls), table(rs)


_table_cache = {}


def _device_table(arr, device):
    if arr is None:
        return None
    key = (arr.tobytes(), str(device))
    t = _table_cache.get(key)
    if t is None:
        t = torch.from_numpy(arr).to(device)
        _table_cache[key] = t
    return t


# ----------------------------------------------------------------------------- backends
def _short_variant(plan):
    """Label of a MGX_SPMM_SHORT_ROWS launch in bench.py's records: which kernels one call runs."""
    if plan is not None and plan.rest is not None:
        return "lane-group (spmm_rowgroup32_kernel, %d short items) + row (spmm_rowwave32_kernel, %d items)" % (plan.num_items, plan.rest.num_items)
    return "lane-group (spmm_rowgroup32_kernel)"


class HipBackend(object):
    """Calls the gfx950 library through its C ABI on PyTorch's current HIP stream."""

    name = "hip"

    def _check_dev(self, *tensors):
        dev = None
        for t in tensors:
            if t is None:
                continue
            if not t.is_cuda:
                raise DGLError("expected a HIP device tensor, got device %s" % t.device)
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise DGLError("tensors on different devices: %s vs %s" % (dev, t.device))
        return dev

    def coo_to_csr(self, num_rows, num_cols, row, col):
        dev = self._check_dev(row, col)
        L = _lib.lib()
        nnz = row.shape[0]
        bits = 32 if row.dtype == torch.int32 else 64
        indptr = torch.empty(num_rows + 1, dtype=row.dtype, device=dev)
        indices = torch.empty(nnz, dtype=row.dtype, device=dev)
        eids = torch.empty(nnz, dtype=row.dtype, device=dev)
        with torch.cuda.device(dev):
            ws_bytes = L.mgx_coo_to_csr_workspace(num_rows, nnz, bits)
            if ws_bytes < 0:
                _lib.check(3)
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
            _lib.check(L.mgx_coo_to_csr(num_rows, nnz, _ptr(row), _ptr(col), bits, _ptr(indptr), _ptr(indices),
                                        _ptr(eids), _ptr(ws), ws_bytes, _stream(dev)))
        return CsrView(num_rows, num_cols, indptr, indices, eids)

    def csr_transpose(self, csr):
        dev = self._check_dev(csr.indptr, csr.indices, csr.eids)
        L = _lib.lib()
        dt = csr.indptr.dtype
        indptr = torch.empty(csr.num_cols + 1, dtype=dt, device=dev)
        indices = torch.empty(csr.nnz, dtype=dt, device=dev)
        eids = torch.empty(csr.nnz, dtype=dt, device=dev)
        with torch.cuda.device(dev):
            ws_bytes = L.mgx_csr_transpose_workspace(csr.num_cols, csr.nnz, csr.idx_bits)
            if ws_bytes < 0:
                _lib.check(3)
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
            _lib.check(L.mgx_csr_transpose(ctypes.byref(csr.c_struct()), _ptr(indptr), _ptr(indices), _ptr(eids), _ptr(ws), ws_bytes,
                                           _stream(dev)))
        return CsrView(csr.num_cols, csr.num_rows, indptr, indices, eids)

    def degrees(self, csr):
        dev = csr.device
        deg = torch.empty(csr.num_rows, dtype=csr.indptr.dtype, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_csr_degrees(csr.num_rows, _ptr(csr.indptr), csr.idx_bits, _ptr(deg), _stream(dev)))
        return deg

    def inv_degrees(self, csr):
        dev = csr.device
        inv = torch.empty(csr.num_rows, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_csr_inv_degrees(csr.num_rows, _ptr(csr.indptr), csr.idx_bits, _ptr(inv), _stream(dev)))
        return inv

    def spmm(self, csr, op, reduce, U, E, u_len, e_len, out_len, u_off, e_off, src_scale, dst_scale, want_arg,
             accumulate_into=None):
        dev = self._check_dev(csr.indptr, U, E, src_scale, dst_scale, accumulate_into)
        if accumulate_into is not None:
            # the C ABI takes a dense [num_rows, out_len] output: a strided view (e.g. gspmm_raw's line-padded wide result)
            # would be written at the wrong offsets -- refuse it loudly (column blocks go through spmm_copy_u_strided)
            if not accumulate_into.is_contiguous() or accumulate_into.numel() != csr.num_rows * out_len:
                raise DGLError("gspmm: accumulate_into must be a contiguous tensor of %d x %d elements (got shape %s, strides %s)"
                               % (csr.num_rows, out_len, tuple(accumulate_into.shape), tuple(accumulate_into.stride())))
            out = accumulate_into.view(csr.num_rows, out_len)
        else:
            out = torch.empty((csr.num_rows, out_len), dtype=torch.float32, device=dev)
        arg_u = arg_e = None
        if want_arg:
            if op != "copy_rhs":
                arg_u = torch.empty((csr.num_rows, out_len), dtype=csr.indptr.dtype, device=dev)
            if op != "copy_lhs":
                arg_e = torch.empty((csr.num_rows, out_len), dtype=csr.indptr.dtype, device=dev)
        if (op == "copy_lhs" and reduce in ("sum", "mean") and src_scale is None and u_off is None and e_off is None
                and u_len == out_len and out_len >= config.TILE_MIN_WIDTH and out_len % 4 == 0 and csr.idx_bits == 32
                and csr.num_cols * out_len * 4 < 2 ** 32 and U.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 and not want_arg):
            tp = csr.tile_plan(out_len)
            if tp is not None:  # dense neighbourhoods: the LDS-staged tile kernel
                self.spmm_tile_copy_u(csr, tp, reduce, U.view(csr.num_cols, out_len), out, accumulate_into is not None, dst_scale)
                return out, None, None
        if (op == "copy_lhs" and reduce in ("sum", "mean") and src_scale is None and u_off is None and e_off is None
                and u_len == out_len and out_len % 4 != 0 and config.TILE_MIN_WIDTH <= 4 and csr.idx_bits == 32 and not want_arg
                and csr.num_cols * (out_len + 3) * 4 < 2 ** 32):
            padded = (out_len + 3) // 4 * 4
            tp = csr.tile_plan(padded)
            if tp is not None:  # odd widths (1, 2, 41 ...) on dense neighbourhoods: through zero-padded copies of both operands
                up = torch.zeros((csr.num_cols, padded), dtype=torch.float32, device=dev)
                up[:, :out_len] = U.view(csr.num_cols, out_len)
                res = self.spmm_tile_copy_u(csr, tp, reduce, up, None, False, dst_scale)[:, :out_len]
                if accumulate_into is not None:
                    out += res
                else:
                    out.copy_(res)
                return out, None, None
        # MGX_SPMM_SHORT_ROWS: full-width copy_u / copy_e sums over short, even work items take the lane-group-per-item kernel
        plan, short = None, False
        if reduce in ("sum", "mean"):
            if (src_scale is None and u_off is None and e_off is None
                    and ((op == "copy_lhs" and u_len == out_len) or (op == "copy_rhs" and e_len == out_len))):
                plan, short = csr.spmm_plan_for(out_len)
            else:
                plan = csr.plan()
        partial = None
        if plan is not None and plan.total_slots:
            partial = torch.empty((plan.total_slots, out_len), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rec = None
            if PROFILE is not None:  # bench.py: HIP events on the launch stream around this launch
                rec = {"op": op, "reduce": reduce, "out_len": out_len, "n_rows": csr.num_rows, "n_cols": csr.num_cols,
                       "nnz": csr.nnz, "accumulate": accumulate_into is not None, "start": torch.cuda.Event(enable_timing=True),
                       "end": torch.cuda.Event(enable_timing=True)}
                if short:
                    rec["variant"] = _short_variant(plan)
                rec["start"].record(torch.cuda.current_stream(dev))
            _lib.check(_lib.lib().mgx_spmm_csr(
                ctypes.byref(csr.c_struct()), None if plan is None else ctypes.byref(plan.c_struct()),
                OP[op], REDUCE[reduce], _ptr(U), _ptr(E), u_len, e_len, out_len,
                _ptr(u_off), _ptr(e_off), _ptr(src_scale), _ptr(dst_scale), _ptr(out), _ptr(arg_u), _ptr(arg_e),
                _ptr(partial), (1 if accumulate_into is not None else 0) | (2 if short else 0), _stream(dev)))
            if rec is not None:
                rec["end"].record(torch.cuda.current_stream(dev))
                PROFILE.append(rec)
        return out, arg_u, arg_e

    def row_nonzero_bits(self, x2d):
        """Bitmap of the rows of x2d [n, D] that hold a non-zero (uint32 words, n padded to 64 rows)."""
        dev = self._check_dev(x2d)
        n, D = x2d.shape
        bits = torch.empty(2 * ((n + 63) // 64), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_row_nonzero_bits(n, D, _ptr(x2d), _ptr(bits), _stream(dev)))
        return bits

    def spmm_copy_u_masked(self, csr, reduce, U2d, bits, dst_scale=None, accumulate_into=None):
        """copy_u / sum|mean that skips source rows whose bit is clear (exact when those rows are zero)."""
        dev = self._check_dev(csr.indptr, U2d, bits, dst_scale, accumulate_into)
        D = int(U2d.shape[1])
        if accumulate_into is not None and (not accumulate_into.is_contiguous() or accumulate_into.numel() != csr.num_rows * D):
            raise DGLError("spmm_copy_u_masked: accumulate_into must be a contiguous tensor of %d x %d elements" % (csr.num_rows, D))
        out = accumulate_into.view(csr.num_rows, D) if accumulate_into is not None else \
            torch.empty((csr.num_rows, D), dtype=torch.float32, device=dev)
        plan = csr.plan()
        partial = torch.empty((plan.num_slots, D), dtype=torch.float32, device=dev) if plan is not None and plan.num_slots else None
        with torch.cuda.device(dev):
            rec = None
            if PROFILE is not None:
                rec = {"op": "copy_lhs", "reduce": reduce, "out_len": D, "n_rows": csr.num_rows, "n_cols": csr.num_cols,
                       "nnz": csr.nnz, "variant": "row-sparse", "start": torch.cuda.Event(enable_timing=True),
                       "end": torch.cuda.Event(enable_timing=True)}
                rec["start"].record(torch.cuda.current_stream(dev))
            _lib.check(_lib.lib().mgx_spmm_copy_u_masked(
                ctypes.byref(csr.c_struct()), None if plan is None else ctypes.byref(plan.c_struct()), REDUCE[reduce], _ptr(U2d), D,
                _ptr(bits), _ptr(dst_scale), _ptr(out), _ptr(partial), 1 if accumulate_into is not None else 0, _stream(dev)))
            if rec is not None:
                rec["end"].record(torch.cuda.current_stream(dev))
                PROFILE.append(rec)
        return out

    def edge_tail_of(self, csr, x2d):
        """(block_a [n, 96] compact, 128-byte aligned; edge_tail [nnz, 4] = x2d[csr.indices, 96:100]) of a CONSTANT x2d [n, 100]: the operands
        of spmm_copy_u_edge_tail, laid out once per (CSR, matrix).  None when mgx_spmm_copy_u_edge_tail would not take them."""
        plan, short = csr.spmm_plan_for(100)
        if (x2d.dim() != 2 or x2d.shape != (csr.num_cols, 100) or x2d.dtype != torch.float32 or x2d.stride(1) != 1 or csr.idx_bits != 32
                or short or (plan is not None and plan.rest is not None) or csr.num_cols * 384 >= 2 ** 32 or csr.tile_plan(100) is not None):
            return None
        dev = self._check_dev(csr.indptr, x2d)
        block_a = torch.empty((csr.num_cols, 96), dtype=torch.float32, device=dev)   # (the allocator aligns to 512 bytes)
        block_a.copy_(x2d[:, :96])
        tail = torch.empty((csr.nnz, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_edge_tail_fill(ctypes.byref(csr.c_struct()), _ptr(x2d), int(x2d.stride(0)), _ptr(tail), _stream(dev)))
        return block_a, tail

    def spmm_copy_u_edge_tail(self, csr, reduce, block_a, tail, out2d, accumulate=False, dst_scale=None):
        """copy_u / sum|mean of the constant [num_cols, 100] matrix held as edge_tail_of() into out2d [num_rows, 100] (row-strided view allowed)."""
        dev = self._check_dev(csr.indptr, block_a, tail, out2d, dst_scale)
        if (block_a.shape != (csr.num_cols, 96) or tail.shape != (csr.nnz, 4) or not block_a.is_contiguous() or not tail.is_contiguous()
                or out2d.shape != (csr.num_rows, 100) or out2d.stride(1) != 1):
            raise DGLError("spmm_copy_u_edge_tail: operands of edge_tail_of() and a [num_rows, 100] output expected")
        plan, _ = csr.spmm_plan_for(100)
        partial = torch.empty((plan.total_slots, 100), dtype=torch.float32, device=dev) if plan is not None and plan.total_slots else None
        with torch.cuda.device(dev):
            rec = None
            if PROFILE is not None:
                rec = {"op": "copy_lhs", "reduce": reduce, "out_len": 100, "n_rows": csr.num_rows, "n_cols": csr.num_cols, "nnz": csr.nnz,
                       "accumulate": bool(accumulate), "variant": "row+edge tail", "strides": (96, int(out2d.stride(0))),
                       "start": torch.cuda.Event(enable_timing=True), "end": torch.cuda.Event(enable_timing=True)}
                rec["start"].record(torch.cuda.current_stream(dev))
            _lib.check(_lib.lib().mgx_spmm_copy_u_edge_tail(ctypes.byref(csr.c_struct()), None if plan is None else ctypes.byref(plan.c_struct()),
                                                            REDUCE[reduce], _ptr(block_a), _ptr(tail), _ptr(dst_scale), _ptr(out2d),
                                                            int(out2d.stride(0)), _ptr(partial), 1 if accumulate else 0, _stream(dev)))
            if rec is not None:
                rec["end"].record(torch.cuda.current_stream(dev))
                PROFILE.append(rec)
        return out2d

    def rows_slots_supported(self, x2d, csr=None):
        """Can x2d [n, 64] travel as 128-byte slots (mgx_rows_slots_pack) into the g-SpMM over `csr` (mgx_spmm_copy_u_slots)?"""
        return (x2d.dim() == 2 and x2d.shape[1] == 64 and x2d.dtype == torch.float32 and x2d.stride(1) == 1 and x2d.stride(0) % 4 == 0
                and x2d.data_ptr() % 16 == 0 and x2d.shape[0] < (1 << 25) and x2d.shape[0] * int(x2d.stride(0)) * 4 < 2 ** 32
                and (csr is None or (csr.idx_bits == 32 and csr.num_cols == x2d.shape[0] and csr.tile_plan(64) is None)))

    def rows_slots_pack(self, x2d, overflow=None, row_scale=None):
        """(slots [n, 32] int32 = one 128-byte slot per row of x2d [n, 64] (times row_scale[r] when given), overflow int64[1] on the device
        += rows with more than 24 non-zeros, which the consumer reads from x2d itself): include/mi355x_graph.h, mgx_rows_slots_pack."""
        dev = self._check_dev(x2d, overflow, row_scale)
        n = int(x2d.shape[0])
        slots = torch.empty((n, 32), dtype=torch.int32, device=dev)
        if overflow is None:
            overflow = torch.zeros(1, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_rows_slots_pack(n, int(x2d.shape[1]), _ptr(x2d), int(x2d.stride(0)), _ptr(row_scale), _ptr(slots),
                                                      _ptr(overflow), _stream(dev)))
        return slots, overflow

    def spmm_copy_u_strided(self, csr, reduce, U2d, out2d, accumulate=False, dst_scale=None, slots=None, src_scale=None):
        """copy_u / sum|mean reading rows of U2d [num_cols, D] and writing rows of out2d [num_rows, D] IN PLACE, both row-strided
        views (stride(1) == 1) -- column blocks of wider matrices (mgx_spmm_copy_u_strided).  `slots` = rows_slots_pack(U2d)[0]: the
        work items gather the 128-byte slots instead of the 256-byte rows (mgx_spmm_copy_u_slots); `src_scale` (with slots only): the
        row factors the slots were packed with -- out = sum_u src_scale[u] * U2d[u]."""
        dev = self._check_dev(csr.indptr, U2d, out2d, dst_scale, src_scale)
        if src_scale is not None and slots is None:
            raise DGLError("spmm_copy_u_strided: src_scale comes with slots packed with the same factors")
        D = int(U2d.shape[1])
        if (U2d.dim() != 2 or out2d.dim() != 2 or U2d.stride(1) != 1 or out2d.stride(1) != 1 or out2d.shape[1] != D
                or U2d.shape[0] != csr.num_cols or out2d.shape[0] != csr.num_rows):
            raise DGLError("spmm_copy_u_strided: expected row-strided [num_cols, D] -> [num_rows, D] views")
        if (D >= config.TILE_MIN_WIDTH and D % 4 == 0 and csr.idx_bits == 32 and csr.num_cols * int(U2d.stride(0)) * 4 < 2 ** 32
                and U2d.stride(0) % 4 == 0 and out2d.stride(0) % 4 == 0 and U2d.data_ptr() % 16 == 0 and out2d.data_ptr() % 16 == 0):
            tp = csr.tile_plan(D)
            if tp is not None:
                return self.spmm_tile_copy_u(csr, tp, reduce, U2d, out2d, accumulate, dst_scale)
        plan, short = csr.spmm_plan_for(D)
        partial = torch.empty((plan.total_slots, D), dtype=torch.float32, device=dev) if plan is not None and plan.total_slots else None
        with torch.cuda.device(dev):
            rec = None
            if PROFILE is not None:
                rec = {"op": "copy_lhs", "reduce": reduce, "out_len": D, "n_rows": csr.num_rows, "n_cols": csr.num_cols,
                       "nnz": csr.nnz, "accumulate": bool(accumulate), "strides": (int(U2d.stride(0)), int(out2d.stride(0))),
                       "start": torch.cuda.Event(enable_timing=True), "end": torch.cuda.Event(enable_timing=True)}
                if short:
                    rec["variant"] = _short_variant(plan)
                rec["start"].record(torch.cuda.current_stream(dev))
            if slots is not None:
                if slots.shape != (U2d.shape[0], 32) or slots.dtype != torch.int32 or not slots.is_contiguous():
                    raise DGLError("spmm_copy_u_strided: slots must be the [num_cols, 32] int32 result of rows_slots_pack")
                if rec is not None:
                    rec["variant"] = (rec.get("variant", "") + "+slots").lstrip("+")
                _lib.check(_lib.lib().mgx_spmm_copy_u_slots(
                    ctypes.byref(csr.c_struct()), None if plan is None else ctypes.byref(plan.c_struct()), REDUCE[reduce], _ptr(U2d), D,
                    int(U2d.stride(0)), _ptr(slots), _ptr(src_scale), _ptr(dst_scale), _ptr(out2d), int(out2d.stride(0)), _ptr(partial),
                    (1 if accumulate else 0) | (2 if short else 0), _stream(dev)))
            else:
                _lib.check(_lib.lib().mgx_spmm_copy_u_strided(
                    ctypes.byref(csr.c_struct()), None if plan is None else ctypes.byref(plan.c_struct()), REDUCE[reduce], _ptr(U2d), D,
                    int(U2d.stride(0)), _ptr(dst_scale), _ptr(out2d), int(out2d.stride(0)), _ptr(partial),
                    (1 if accumulate else 0) | (2 if short else 0), _stream(dev)))
            if rec is not None:
                rec["end"].record(torch.cuda.current_stream(dev))
                PROFILE.append(rec)
        return out2d

    def spmm_tile_copy_u(self, csr, tile_plan, reduce, U2d, out2d=None, accumulate=False, dst_scale=None):
        """copy_u / sum|mean through the LDS-staged tile kernel (mgx_spmm_tile_copy_u); U2d / out2d may be row-strided views."""
        dev = self._check_dev(csr.indptr, U2d, out2d, dst_scale)
        D = int(U2d.shape[1])
        if out2d is None:
            out2d = torch.empty((csr.num_rows, D), dtype=torch.float32, device=dev)
        if (U2d.dim() != 2 or out2d.dim() != 2 or U2d.stride(1) != 1 or out2d.stride(1) != 1 or out2d.shape[1] != D
                or U2d.shape[0] != csr.num_cols or out2d.shape[0] != csr.num_rows):
            raise DGLError("spmm_tile_copy_u: expected row-strided [num_cols, D] -> [num_rows, D] views")
        base = tile_plan.base
        partial = torch.empty((base.num_slots, D), dtype=torch.float32, device=dev) if base is not None and base.num_slots else None
        with torch.cuda.device(dev):
            rec = None
            if PROFILE is not None:
                rec = {"op": "copy_lhs", "reduce": reduce, "out_len": D, "n_rows": csr.num_rows, "n_cols": csr.num_cols,
                       "nnz": csr.nnz, "accumulate": bool(accumulate), "variant": "tile",
                       "start": torch.cuda.Event(enable_timing=True), "end": torch.cuda.Event(enable_timing=True)}
                rec["start"].record(torch.cuda.current_stream(dev))
            _lib.check(_lib.lib().mgx_spmm_tile_copy_u(
                ctypes.byref(csr.c_struct()), None if base is None else ctypes.byref(base.c_struct()), ctypes.byref(tile_plan.c_struct()),
                REDUCE[reduce], _ptr(U2d), D, int(U2d.stride(0)), _ptr(dst_scale), _ptr(out2d), int(out2d.stride(0)), _ptr(partial),
                1 if accumulate else 0, _stream(dev)))
            if rec is not None:
                rec["end"].record(torch.cuda.current_stream(dev))
                PROFILE.append(rec)
        return out2d

    def sddmm(self, graph_index, op, L, R, lhs_target, rhs_target, l_len, r_len, out_len, reduce_size, l_off, r_off):
        """graph_index supplies either COO (edge-id order) or the in-CSR."""
        nnz = graph_index.num_edges()
        dev = self._check_dev(L, R)
        out = torch.empty((nnz, out_len), dtype=torch.float32, device=dev)
        lib = _lib.lib()
        if self._sddmm_in_csr_order(graph_index, op, L, R, lhs_target, rhs_target, l_len, r_len, out_len, l_off, r_off):
            # the edge list in the in-CSR's order (sorted by destination) with the edge id as the output row: sddmm_coo32_kernel<PERM>
            csc = graph_index.csc()
            with torch.cuda.device(dev), timed_call(dev, kernel="sddmm", op=op, out_len=out_len, l_len=l_len, r_len=r_len, nnz=nnz,
                                                    n_src=graph_index.num_src, n_dst=graph_index.num_dst, targets=lhs_target + rhs_target,
                                                    walk="csr order"):
                st = lib.mgx_sddmm_coo_perm(graph_index.num_src, graph_index.num_dst, nnz, _ptr(csc.indices), _ptr(csc.row_of_position()),
                                            _ptr(csc.eids), 32, OP[op], _ptr(L), _ptr(R), TARGET[lhs_target], TARGET[rhs_target], out_len,
                                            _ptr(out), _stream(dev))
            if st != _lib.ERR_UNSUPPORTED:
                _lib.check(st)
                return out
        with torch.cuda.device(dev), timed_call(dev, kernel="sddmm", op=op, out_len=out_len, l_len=l_len, r_len=r_len, nnz=nnz,
                                                n_src=graph_index.num_src, n_dst=graph_index.num_dst, targets=lhs_target + rhs_target):
            if graph_index.has_format("coo") or not graph_index.has_format("csc"):
                src, dst = graph_index.coo()
                self._check_dev(src, L, R)
                bits = 32 if src.dtype == torch.int32 else 64
                _lib.check(lib.mgx_sddmm_coo(
                    graph_index.num_src, graph_index.num_dst, nnz, _ptr(src), _ptr(dst), bits, OP[op], _ptr(L), _ptr(R),
                    TARGET[lhs_target], TARGET[rhs_target], l_len, r_len, out_len, reduce_size, _ptr(l_off), _ptr(r_off),
                    _ptr(out), _stream(dev)))
            else:
                csr = graph_index.csc()
                self._check_dev(csr.indptr, L, R)
                plan = csr.plan()
                _lib.check(lib.mgx_sddmm_csr(
                    ctypes.byref(csr.c_struct()), None if plan is None else ctypes.byref(plan.c_struct()),
                    OP[op], _ptr(L), _ptr(R), TARGET[lhs_target], TARGET[rhs_target],
                    l_len, r_len, out_len, reduce_size, _ptr(l_off), _ptr(r_off), _ptr(out), _stream(dev)))
        return out

    SDDMM_PERM_MIN_EDGES = 1 << 20
    SDDMM_PERM_MIN_WIDTH = 32  # measured (profiles/r05_sddmm_perm.txt): D = 64 / 128 gain 17 - 27 %, D = 16 loses 20 % (a 64-byte output row
                               # is half a line: scattering it costs more than the operand misses it saves)

    @staticmethod
    def _sddmm_in_csr_order(gidx, op, L, R, lt, rt, l_len, r_len, out_len, l_off, r_off):
        """Walk the in-CSR's order with the edge id as the output row (mgx_sddmm_coo_perm) instead of the edge list's own order?
        MGX_SDDMM_WALK = coo | csr | auto (default).  auto: a big graph that holds an in-CSR whose edge ids are NOT its positions, an
        element-wise op on u / v operands as wide as the output -- and an edge list that is not already sorted by an endpoint (checked
        once per graph; a sorted list has the same locality AND streams its output, profiles/r03_sddmm_edge_order.txt)."""
        mode = os.environ.get("MGX_SDDMM_WALK", "auto")
        if (mode == "coo" or not gidx.has_format("csc") or op == "dot" or l_off is not None or r_off is not None or out_len < 4
                or (L is not None and (lt not in "uv" or l_len != out_len)) or (R is not None and (rt not in "uv" or r_len != out_len))):
            return False
        csc = gidx.csc()
        if csc.eids is None or csc.idx_bits != 32 or not csc.indptr.is_cuda:
            return False
        if mode == "csr":
            return True
        if getattr(gidx, "ephemeral", False) or csc.nnz < HipBackend.SDDMM_PERM_MIN_EDGES or out_len < HipBackend.SDDMM_PERM_MIN_WIDTH:
            return False
        if not gidx.has_format("coo"):
            return True  # no edge list to walk: the lean kernel over the CSR's order replaces the generic CSR body
        order = getattr(gidx, "_coo_sorted", None)
        if order is None:
            src, dst = gidx.coo()
            order = bool(((dst[1:] >= dst[:-1]).all() | (src[1:] >= src[:-1]).all()).item()) if dst.numel() > 1 else True
            try:
                gidx._coo_sorted = order
            except AttributeError:
                pass
        return not order

    def edge_softmax_fwd(self, csr, z2d):
        dev = self._check_dev(csr.indptr, z2d)
        a = torch.empty_like(z2d)
        plan, ws = self._softmax_plan(csr, z2d.shape[1], dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_edge_softmax_fwd(ctypes.byref(csr.c_struct()), plan, z2d.shape[1], _ptr(z2d), _ptr(a),
                                                       _ptr(ws), _stream(dev)))
        return a

    def gat_attention_fwd(self, csr, el2d, er2d, slope):
        dev = self._check_dev(csr.indptr, el2d, er2d)
        H = el2d.shape[1]
        a = torch.empty((csr.nnz, H), dtype=torch.float32, device=dev)
        plan, ws = self._softmax_plan(csr, H, dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_gat_attention_fwd(ctypes.byref(csr.c_struct()), plan, H, _ptr(el2d), _ptr(er2d),
                                                        ctypes.c_float(slope), _ptr(a), _ptr(ws), _stream(dev)))
        return a

    def gat_attention_bwd(self, csr, el2d, er2d, slope, a2d, da2d):
        dev = self._check_dev(csr.indptr, el2d, er2d, a2d, da2d)
        H = el2d.shape[1]
        de = torch.empty_like(a2d)
        plan, ws = self._softmax_plan(csr, H, dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_gat_attention_bwd(ctypes.byref(csr.c_struct()), plan, H, _ptr(el2d), _ptr(er2d),
                                                        ctypes.c_float(slope), _ptr(a2d), _ptr(da2d), _ptr(de), _ptr(ws),
                                                        _stream(dev)))
        return de

    @staticmethod
    def gat_fused_supported(csr, H, F):
        """Shapes mgx_gat_fused_* takes: F a power of two >= 4 with H*F <= 256, or ONE head of any width 4 < F <= 256;
        32-bit indices and byte offsets."""
        regular = F >= 4 and F % 4 == 0 and ((F // 4) & (F // 4 - 1)) == 0 and H * F <= 256
        ragged = H == 1 and 4 < F <= 256
        return ((regular or ragged) and csr.idx_bits == 32
                and max(csr.num_cols, csr.num_rows) * H * max(F * 4, 16) < 2 ** 32 and csr.nnz < 2 ** 31)

    @staticmethod
    def _plan_ptr(plan):
        return None if plan is None else ctypes.byref(plan.c_struct())

    def _gat_ws(self, plans, H, F, dev):
        L = _lib.lib()
        need = max([L.mgx_gat_fused_workspace(self._plan_ptr(p), H, F) for p in plans] + [0])
        return torch.empty(need // 4, dtype=torch.float32, device=dev) if need else None

    @staticmethod
    def _gat_tile_plans(csc, csr, H, F, p):
        """(tile plan of the in-CSR, of the out-CSR | None) when a layer's walks run as tile kernels, else None.  With attn_drop the
        two forms key the mask differently (edge id / (destination, source, rank among parallel edges)), so a layer takes the tile
        form for ALL three walks or for none: it needs both plans with their rank streams."""
        if H != 1:
            return None
        t_dst = csc.gat_tile_plan(F)
        if t_dst is None:
            return None
        t_src = csr.gat_tile_plan(F) if csr is not None else None
        if p > 0.0 and (t_src is None or not t_dst.stats["pair_rank_streams"] or not t_src.stats["pair_rank_streams"]):
            return None
        return t_dst, t_src

    def gat_fused_fwd(self, csc, feat3d, el2d, er2d, slope, p, seed, attn_l=None, csr=None):
        """feat3d [n_src, H, F], el2d [n_src, H], er2d [n_dst, H] -> (out [n_dst, H, F], nstat [n_dst, H, 4], form).
        csr: the out-CSR, needed to choose the tile form when p > 0 (see _gat_tile_plans).  `form` ("tile" | "row") is the
        kernel family that ran: with p > 0 it fixes how the dropout mask is keyed, and the backward must be given it back."""
        dev = self._check_dev(csc.indptr, feat3d, el2d, er2d)
        H, F = int(feat3d.shape[1]), int(feat3d.shape[2])
        out = torch.empty((csc.num_rows, H, F), dtype=torch.float32, device=dev)
        nstat = torch.empty((csc.num_rows, H, 4), dtype=torch.float32, device=dev)
        tiles = self._gat_tile_plans(csc, csr, H, F, p)
        if tiles is not None:  # dense neighbourhoods, one narrow head: the LDS-staged tile walk
            tp = tiles[0]
            ws = self._gat_ws([tp.base], H, F, dev)
            pack = self._gat_pack_ws(csc, H, F, dev)
            with torch.cuda.device(dev), timed_call(dev, kernel="gat_fwd", form="tile", H=H, F=F, nnz=csc.nnz, n_src=csc.num_cols, n_dst=csc.num_rows):
                _lib.check(_lib.lib().mgx_gat_tile_fwd(ctypes.byref(csc.c_struct()), self._plan_ptr(tp.base), ctypes.byref(tp.c_struct()),
                                                       H, F, _ptr(feat3d), _ptr(el2d), _ptr(er2d), ctypes.c_float(slope), ctypes.c_float(p),
                                                       ctypes.c_uint64(seed), _ptr(out), _ptr(nstat), _ptr(ws), _ptr(pack), _stream(dev)))
            return out, nstat, "tile"
        plan = csc.plan()
        ws = self._gat_ws([plan], H, F, dev)
        pack = self._gat_pack_ws(csc, H, F, dev)
        with torch.cuda.device(dev), timed_call(dev, kernel="gat_fwd", form="row", H=H, F=F, nnz=csc.nnz, n_src=csc.num_cols, n_dst=csc.num_rows):
            _lib.check(_lib.lib().mgx_gat_fused_fwd(ctypes.byref(csc.c_struct()), self._plan_ptr(plan), H, F, _ptr(feat3d), _ptr(el2d),
                                                    _ptr(attn_l), _ptr(er2d), ctypes.c_float(slope), ctypes.c_float(p), ctypes.c_uint64(seed),
                                                    _ptr(out), _ptr(nstat), _ptr(ws), _ptr(pack), _stream(dev)))
        return out, nstat, "row"

    @staticmethod
    def _gat_pack_ws(csc, H, F, dev):
        if not config.GAT_PACK:
            return None
        need = _lib.lib().mgx_gat_fused_pack_workspace(csc.num_cols, csc.num_rows, H, F)
        return torch.empty(need // 4, dtype=torch.float32, device=dev) if need else None

    def gat_fused_bwd(self, csc, csr, feat3d, el2d, slope, p, seed, out3d, d_out3d, nstat, need_src, attn_l=None, form=None):
        """-> (d_feat | None, d_el | None, d_er); nstat[..., 3] is overwritten with <out, d_out> per head.
        `form`: what gat_fused_fwd returned.  With p > 0 the backward runs that same kernel family or raises -- the two families
        key the dropout mask differently (edge id / (destination, source, rank)), so re-deriving the choice here (an environment
        switch flipped, a plan that failed to build in between) would silently apply another mask."""
        dev = self._check_dev(csc.indptr, csr.indptr, feat3d, el2d, out3d, d_out3d, nstat)
        H, F = int(feat3d.shape[1]), int(feat3d.shape[2])
        d_er = torch.empty((csc.num_rows, H), dtype=torch.float32, device=dev)
        d_feat = torch.empty_like(feat3d) if need_src else None
        d_el = torch.empty((csc.num_cols, H), dtype=torch.float32, device=dev) if need_src else None
        tiles = self._gat_tile_plans(csc, csr, H, F, p)
        if p > 0.0 and form == "row":
            tiles = None
        elif p > 0.0 and form == "tile" and (tiles is None or tiles[1] is None):
            raise DGLError("gat_fused backward: the forward pass drew its dropout mask with the tile kernels, but their plans are not "
                           "available any more (MGX_GAT_TILE / MGX_TILE changed between forward and backward?)")
        if tiles is not None and tiles[1] is not None:
            t_dst, t_src = tiles
            if not need_src:  # the kernel's source walk is skipped when both are NULL
                d_feat = d_el = None
            ws = self._gat_ws([t_dst.base, t_src.base], H, F, dev)
            pack = self._gat_pack_ws(csc, H, F, dev)
            with torch.cuda.device(dev), timed_call(dev, kernel="gat_bwd", form="tile", H=H, F=F, nnz=csc.nnz, n_src=csc.num_cols, n_dst=csc.num_rows,
                                                    source_walk=bool(need_src)):
                _lib.check(_lib.lib().mgx_gat_tile_bwd(ctypes.byref(csc.c_struct()), self._plan_ptr(t_dst.base), ctypes.byref(t_dst.c_struct()),
                                                       ctypes.byref(csr.c_struct()), self._plan_ptr(t_src.base), ctypes.byref(t_src.c_struct()),
                                                       H, F, _ptr(feat3d), _ptr(el2d), ctypes.c_float(slope), ctypes.c_float(p),
                                                       ctypes.c_uint64(seed), _ptr(out3d), _ptr(d_out3d), _ptr(nstat), _ptr(d_feat),
                                                       _ptr(d_el), _ptr(d_er), _ptr(ws), _ptr(pack), _stream(dev)))
            return d_feat, d_el, d_er
        p_dst, p_src = csc.plan(), csr.plan()
        ws = self._gat_ws([p_dst, p_src], H, F, dev)
        pack = self._gat_pack_ws(csc, H, F, dev)
        with torch.cuda.device(dev), timed_call(dev, kernel="gat_bwd", form="row", H=H, F=F, nnz=csc.nnz, n_src=csc.num_cols, n_dst=csc.num_rows,
                                                source_walk=bool(need_src)):
            _lib.check(_lib.lib().mgx_gat_fused_bwd(ctypes.byref(csc.c_struct()), self._plan_ptr(p_dst), ctypes.byref(csr.c_struct()),
                                                    self._plan_ptr(p_src), H, F, _ptr(feat3d), _ptr(el2d), _ptr(attn_l), ctypes.c_float(slope),
                                                    ctypes.c_float(p), ctypes.c_uint64(seed), _ptr(out3d), _ptr(d_out3d), _ptr(nstat),
                                                    _ptr(d_feat), _ptr(d_el), _ptr(d_er), _ptr(ws), _ptr(pack), _stream(dev)))
        return d_feat, d_el, d_er

    @staticmethod
    def head_dot_supported(H, F):
        return H * F <= 256 and (F <= 64 or F in (128, 256))

    def head_dot_fwd(self, feat3d, attn_a, attn_b):
        """feat3d [n, H, F], attn_a/attn_b [H, F] (attn_b may be None) -> (out_a [n, H], out_b [n, H] | None)."""
        dev = self._check_dev(feat3d, attn_a, attn_b)
        n, H, F = feat3d.shape
        out_a = torch.empty((n, H), dtype=torch.float32, device=dev)
        out_b = torch.empty((n, H), dtype=torch.float32, device=dev) if attn_b is not None else None
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_head_dot_fwd(n, H, F, _ptr(feat3d), _ptr(attn_a), _ptr(attn_b), _ptr(out_a), _ptr(out_b),
                                                   _stream(dev)))
        return out_a, out_b

    def head_dot_bwd(self, feat3d, attn_a, attn_b, d_a, d_b, need_feat_grad):
        dev = self._check_dev(feat3d, attn_a, attn_b, d_a, d_b)
        n, H, F = feat3d.shape
        d_feat = torch.empty_like(feat3d) if need_feat_grad else None
        g_a = torch.empty_like(attn_a)
        g_b = torch.empty_like(attn_b) if attn_b is not None else None
        ws = torch.empty(_lib.lib().mgx_head_dot_bwd_workspace(H, F) // 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_head_dot_bwd(n, H, F, _ptr(feat3d), _ptr(attn_a), _ptr(attn_b), _ptr(d_a), _ptr(d_b),
                                                   _ptr(d_feat), _ptr(g_a), _ptr(g_b), _ptr(ws), _stream(dev)))
        return d_feat, g_a, g_b

    @staticmethod
    def _softmax_plan(csr, H, dev):
        plan = csr.softmax_plan()
        if plan is None:
            return None, None
        ws = None
        if plan.num_slots:
            ws = torch.empty((plan.num_slots + plan.num_hubs) * 2 * H, dtype=torch.float32, device=dev)
        return ctypes.byref(plan.c_struct()), ws

    def edge_softmax_bwd(self, csr, a2d, da2d):
        dev = self._check_dev(csr.indptr, a2d, da2d)
        dz = torch.empty_like(a2d)
        plan, ws = self._softmax_plan(csr, a2d.shape[1], dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_edge_softmax_bwd(ctypes.byref(csr.c_struct()), plan, a2d.shape[1], _ptr(a2d), _ptr(da2d),
                                                       _ptr(dz), _ptr(ws), _stream(dev)))
        return dz

    def segment_reduce(self, offsets, x2d, reduce, want_arg):
        dev = self._check_dev(offsets, x2d)
        n = offsets.shape[0] - 1
        out = torch.empty((n, x2d.shape[1]), dtype=torch.float32, device=dev)
        arg = torch.empty((n, x2d.shape[1]), dtype=torch.int64, device=dev) if want_arg else None
        with torch.cuda.device(dev), timed_call(dev, kernel="segment_reduce", reduce=reduce, segments=int(n), rows=int(x2d.shape[0]), D=int(x2d.shape[1])):
            _lib.check(_lib.lib().mgx_segment_reduce(n, _ptr(offsets), x2d.shape[1], REDUCE[reduce], _ptr(x2d), _ptr(out), _ptr(arg), _stream(dev)))
        return out, arg

    @staticmethod
    def _row_strided(t):
        return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] and t.stride(0) % 4 == 0 and t.shape[1] % 4 == 0

    def relu_dropout_fwd(self, x, p, seed, offset, out=None, counter=None):
        """`out`: a row-strided [rows, cols] view to write into (x then 2-D, possibly row-strided itself).  `counter`: a
        1-element int64 device tensor read by the launch when it RUNS instead of the host-side `offset` (HIP-graph replays)."""
        dev = self._check_dev(x, out, counter)
        mask = torch.empty(x.numel() // 4, dtype=torch.uint8, device=dev)
        if counter is not None:
            y = out if out is not None else torch.empty(x.shape, dtype=torch.float32, device=dev)
            if not (self._row_strided(x) and self._row_strided(y) and x.shape == y.shape):
                raise DGLError("relu_dropout_fwd: counter mode needs [rows, cols] operands with unit column stride, cols % 4 == 0")
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().mgx_relu_dropout_fwd_counter(x.shape[0], x.shape[1], _ptr(x), x.stride(0), ctypes.c_float(p),
                                                                    ctypes.c_uint64(seed), _ptr(counter), _ptr(y), y.stride(0),
                                                                    _ptr(mask), _stream(dev)))
            return y, mask
        if out is None and x.is_contiguous():
            y = torch.empty_like(x)
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().mgx_relu_dropout_fwd(x.numel(), _ptr(x), ctypes.c_float(p), ctypes.c_uint64(seed),
                                                           ctypes.c_uint64(offset), _ptr(y), _ptr(mask), _stream(dev)))
            return y, mask
        y = out if out is not None else torch.empty(x.shape, dtype=torch.float32, device=dev)
        if not (self._row_strided(x) and self._row_strided(y) and x.shape == y.shape):
            raise DGLError("relu_dropout_fwd: strided operands must be [rows, cols] views with unit column stride, cols % 4 == 0")
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_relu_dropout_fwd_strided(x.shape[0], x.shape[1], _ptr(x), x.stride(0), ctypes.c_float(p),
                                                                ctypes.c_uint64(seed), ctypes.c_uint64(offset), _ptr(y), y.stride(0),
                                                                _ptr(mask), _stream(dev)))
        return y, mask

    def relu_dropout_bwd(self, dy, mask, p, out=None):
        """dy: contiguous, or a row-strided [rows, cols] view (the gradient's column block); dx is dense, or `out`: a row-strided [rows, cols]
        view to write into (the left half of a wider matrix)."""
        dev = self._check_dev(dy, mask, out)
        if out is not None:
            if dy.dim() != 2 or out.shape != dy.shape or not self._row_strided(out) or not (dy.is_contiguous() or self._row_strided(dy)):
                raise DGLError("relu_dropout_bwd: `out` must be a [rows, cols] view with unit column stride of the gradient's shape")
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().mgx_relu_dropout_bwd_strided(dy.shape[0], dy.shape[1], _ptr(dy), dy.stride(0), _ptr(mask),
                                                                    ctypes.c_float(p), _ptr(out), out.stride(0), _stream(dev)))
            return out
        dx = torch.empty(dy.shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if dy.is_contiguous():
                _lib.check(_lib.lib().mgx_relu_dropout_bwd(dy.numel(), _ptr(dy), _ptr(mask), ctypes.c_float(p), _ptr(dx), _stream(dev)))
            else:
                if not self._row_strided(dy):
                    raise DGLError("relu_dropout_bwd: a non-contiguous gradient must be a [rows, cols] view with unit column stride")
                _lib.check(_lib.lib().mgx_relu_dropout_bwd_strided(dy.shape[0], dy.shape[1], _ptr(dy), dy.stride(0), _ptr(mask),
                                                                    ctypes.c_float(p), _ptr(dx), dx.stride(0), _stream(dev)))
        return dx

    def relu_dropout_bwd_slots(self, dy, mask, p, out, row_scale=None):
        """relu_dropout_bwd into `out` (a row-strided [rows, 64] view) + (slots, overflow) of row_scale * result, as rows_slots_pack would
        give them -- one pass (mgx_relu_dropout_bwd_slots)."""
        dev = self._check_dev(dy, mask, out, row_scale)
        if (dy.dim() != 2 or dy.shape[1] != 64 or out.shape != dy.shape or not self._row_strided(out) or out.data_ptr() % 16
                or not (dy.is_contiguous() or self._row_strided(dy)) or dy.data_ptr() % 16):
            raise DGLError("relu_dropout_bwd_slots: [rows, 64] operands with unit column stride, 16-byte aligned")
        n = int(dy.shape[0])
        slots = torch.empty((n, 32), dtype=torch.int32, device=dev)
        overflow = torch.zeros(1, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_relu_dropout_bwd_slots(n, _ptr(dy), dy.stride(0), _ptr(mask), ctypes.c_float(p), _ptr(out), out.stride(0),
                                                             _ptr(row_scale), _ptr(slots), _ptr(overflow), _stream(dev)))
        return out, slots, overflow

    def column_pair_sums(self, a2d, b2d=None, shifted=False):
        """(sum a, sum a*a) per column, or (sum a, sum a*b) when b2d is given.  `shifted`: relative to the first row p of
        the squared / second operand -- (sum (a-p), sum (a-p)^2) with p = a[0], or (sum a, sum a*(b-p)) with p = b[0]."""
        dev = self._check_dev(a2d, b2d)
        n, C = a2d.shape
        out = torch.empty((2, C), dtype=torch.float32, device=dev)
        ws = torch.empty(2 * _lib.lib().mgx_column_sum_workspace(C) // 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_column_pair_sums(n, C, (0 if b2d is None else 1) + (2 if shifted else 0), _ptr(a2d), _ptr(b2d), _ptr(out[0]), _ptr(out[1]),
                                                       _ptr(ws), _stream(dev)))
        return out[0], out[1]

    def column_affine(self, a2d, A, Cc, b2d=None, B=None):
        """a*A[c] + b*B[c] + Cc[c] with per-column coefficient vectors."""
        dev = self._check_dev(a2d, b2d, A, B, Cc)
        out = torch.empty_like(a2d)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_column_affine(a2d.shape[0], a2d.shape[1], _ptr(a2d), _ptr(b2d), _ptr(A), _ptr(B), _ptr(Cc),
                                                    _ptr(out), _stream(dev)))
        return out

    COLUMN_SUM_MAX = 256
    XTY_TILE = (64, 128)      # one tile of mgx_xty
    XTY_MAX = (256, 1024)     # the grid of tiles in one launch goes up to this output shape
    XTY_MIN_ROWS = 1 << 16    # shorter reductions stay with the GEMM library

    def xty(self, a2d, b2d, colsum=False):
        """a2d [n, M], b2d [n, K] -> a^T b [M, K] (the tall-skinny weight-gradient product).  colsum=True: (a^T b, column sums of a) --
        the bias gradient from the same pass over a when the shape allows (mgx_xty_colsum), else by mgx_column_sum."""
        dev = self._check_dev(a2d, b2d)
        n, M = a2d.shape
        K = b2d.shape[1]
        if M > self.XTY_MAX[0] or K > self.XTY_MAX[1]:
            raise DGLError("mgx_xty: at most %d x %d outputs, got %d x %d" % (self.XTY_MAX + (M, K)))
        out = torch.empty((M, K), dtype=torch.float32, device=dev)
        sums = torch.empty(M, dtype=torch.float32, device=dev) if colsum else None
        have_sums = False
        L = _lib.lib()
        tm, tk = self.XTY_TILE
        ntiles = ((M + tm - 1) // tm) * ((K + tk - 1) // tk)

        def one(a_t, b_t, o_t, ws, want_sums):
            if want_sums:
                st = L.mgx_xty_colsum(n, a_t.shape[1], b_t.shape[1], _ptr(a_t), max(a2d.stride(0), M), _ptr(b_t), max(b2d.stride(0), K),
                                      _ptr(o_t), out.stride(0), _ptr(sums), _ptr(ws), _stream(dev))
                if st != _lib.ERR_UNSUPPORTED:
                    _lib.check(st)
                    return True
            _lib.check(L.mgx_xty(n, a_t.shape[1], b_t.shape[1], _ptr(a_t), max(a2d.stride(0), M), _ptr(b_t), max(b2d.stride(0), K),
                                 _ptr(o_t), out.stride(0), _ptr(ws), _stream(dev)))
            return False

        with torch.cuda.device(dev):
            if ntiles == 1 or ntiles >= 4:
                # a single 64 x 128 tile, or the grid of tiles in ONE launch (operand rows shared through L2): measured
                # 256 x 512 at n = 169 k 1.07 -> 0.56 ms, 128 x 602 at n = 233 k 0.78 -> 0.57 ms (experiments/exp_wgrad_shapes.py)
                ws = torch.empty(max(L.mgx_xty_workspace(M, K), 4) // 4, dtype=torch.float32, device=dev)
                have_sums = one(a2d, b2d, out, ws, colsum and ntiles == 1)
            else:
                # two or three tiles (the stacked 64 x 200 of the products layer): one launch per tile, each with the exact tile
                # shape, is faster than the grid's padded 64 x 128 tiles (n = 2.45 M: 0.93 against 1.04 ms)
                ws = torch.empty(max(L.mgx_xty_workspace(min(M, tm), min(K, tk)), 4) // 4, dtype=torch.float32, device=dev)
                for m0 in range(0, M, tm):
                    for k0 in range(0, K, tk):
                        a_t, b_t, o_t = a2d[:, m0:m0 + tm], b2d[:, k0:k0 + tk], out[m0:m0 + tm, k0:k0 + tk]
                        if one(a_t, b_t, o_t, ws, colsum and M <= tm and k0 == 0):
                            have_sums = True
        if not colsum:
            return out
        return out, (sums if have_sums else self.column_sum(a2d if a2d.is_contiguous() else a2d.contiguous()))

    def rows_gemm(self, a2d, b2d, b_transposed=False, bias=None, row_scale=None, scale_from=0, out=None, split_col=0):
        """a2d [n, K] (row-strided view allowed) x B -> [n, M]; B = b2d [K, M], or [M, K] when b_transposed (nn.Linear's layout);
        + bias; columns >= scale_from times row_scale[r].  split_col > 0 (a multiple of 4): TWO compact outputs, the columns before
        and from split_col on, returned as a pair.  None when mgx_rows_gemm has no kernel for the shape (use a GEMM)."""
        dev = self._check_dev(a2d, b2d, bias, row_scale, out)
        n, K = a2d.shape
        M = b2d.shape[0] if b_transposed else b2d.shape[1]
        if (a2d.stride(1) != 1 or b2d.stride(1) != 1 or (b2d.shape[1] if b_transposed else b2d.shape[0]) != K
                or a2d.dtype != torch.float32 or b2d.dtype != torch.float32):
            return None
        out2 = None
        if split_col:
            out = torch.empty((n, split_col), dtype=torch.float32, device=dev)
            out2 = torch.empty((n, M - split_col), dtype=torch.float32, device=dev)
        elif out is None:
            out = torch.empty((n, M), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _lib.lib().mgx_rows_gemm(n, K, M, _ptr(a2d), a2d.stride(0), _ptr(b2d), b2d.stride(0), 1 if b_transposed else 0,
                                          _ptr(bias), _ptr(row_scale), int(scale_from), _ptr(out), out.stride(0), _ptr(out2),
                                          0 if out2 is None else out2.stride(0), int(split_col), _stream(dev))
        if st == _lib.ERR_UNSUPPORTED:
            return None
        _lib.check(st)
        return out if out2 is None else (out, out2)

    def rows_gemm_supported(self, K, M, lda):
        return bool(_lib.lib().mgx_rows_gemm_supported(int(K), int(M), int(lda)))

    def rows_gemm_relu_dropout(self, a2d, b2d, b_transposed, bias, p, seed, offset, out=None, slots=None, overflow=None):
        """dropout(relu(a2d x B + bias), p) written into `out` (a row-strided [n, M] view, or a new matrix) + the mask of
        relu_dropout_fwd -- bit for bit what rows_gemm followed by relu_dropout_fwd(seed, offset) gives, without storing the product.
        `slots` ([n, 32] int32, M == 64): the rows as 128-byte slots too -- what rows_slots_pack(result) would write --, `overflow`
        (int64[1]) += rows above 24 non-zeros.  None when mgx_rows_gemm has no kernel for the shape."""
        dev = self._check_dev(a2d, b2d, bias, out)
        n, K = a2d.shape
        M = b2d.shape[0] if b_transposed else b2d.shape[1]
        if (a2d.stride(1) != 1 or b2d.stride(1) != 1 or (b2d.shape[1] if b_transposed else b2d.shape[0]) != K or M % 4
                or a2d.dtype != torch.float32 or b2d.dtype != torch.float32 or not self.rows_gemm_supported(K, M, a2d.stride(0))):
            return None
        y = out if out is not None else torch.empty((n, M), dtype=torch.float32, device=dev)
        if not self._row_strided(y) or y.shape != (n, M) or y.data_ptr() % 16:
            return None
        mask = torch.empty(n * M // 4, dtype=torch.uint8, device=dev)
        if slots is not None and (M != 64 or slots.shape != (n, 32) or slots.dtype != torch.int32 or not slots.is_contiguous()):
            raise DGLError("rows_gemm_relu_dropout: slots are [n, 32] int32 for 64 output columns")
        with torch.cuda.device(dev):
            st = _lib.lib().mgx_rows_gemm_relu_dropout(n, K, M, _ptr(a2d), a2d.stride(0), _ptr(b2d), b2d.stride(0),
                                                       1 if b_transposed else 0, _ptr(bias), ctypes.c_float(p), ctypes.c_uint64(seed),
                                                       ctypes.c_uint64(offset), _ptr(y), y.stride(0), _ptr(mask), _ptr(slots),
                                                       _ptr(overflow) if slots is not None else None, _stream(dev))
        if st == _lib.ERR_UNSUPPORTED:  # e.g. an operand that is only dword aligned: mgx_rows_gemm_supported() never sees the pointer
            return None
        _lib.check(st)
        return y, mask

    def column_sum(self, x2d):
        dev = self._check_dev(x2d)
        n, C = x2d.shape
        out = torch.empty(C, dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.lib().mgx_column_sum_workspace(C) // 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_column_sum(n, C, _ptr(x2d), _ptr(out), _ptr(ws), _stream(dev)))
        return out

    def sample_neighbors(self, csr, seeds, fanout, rng_seed):
        """seeds: graph-idtype tensor on the device.  Returns (src, eid, counts) with the picks of seed i at
        [offsets[i], offsets[i+1])."""
        dev = self._check_dev(csr.indptr, seeds)
        deg = (csr.indptr[seeds.long() + 1] - csr.indptr[seeds.long()]).long()
        counts = torch.clamp(deg, max=fanout)
        offsets = torch.zeros(seeds.shape[0] + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=offsets[1:])
        total = int(offsets[-1].item())
        src = torch.empty(total, dtype=csr.indptr.dtype, device=dev)
        eid = torch.empty(total, dtype=csr.indptr.dtype, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_sample_neighbors(ctypes.byref(csr.c_struct()), seeds.shape[0], _ptr(seeds), int(fanout),
                                                       ctypes.c_uint64(rng_seed & (2 ** 64 - 1)), _ptr(offsets), _ptr(src),
                                                       _ptr(eid), _stream(dev)))
        return src, eid, counts

    # ---- halo rows as bitmaps + packed values (csrc/rowpack.hip; dist.SparseHalo)
    ROWPACK_MAX_D = 256

    def rows_pack_supported(self, x2d):
        return (x2d.dim() == 2 and x2d.dtype == torch.float32 and x2d.stride(1) == 1 and x2d.shape[1] % 4 == 0
                and 4 <= x2d.shape[1] <= self.ROWPACK_MAX_D and x2d.stride(0) % 4 == 0 and x2d.data_ptr() % 16 == 0)

    def rows_pack_count(self, x2d, idx):
        """(masks [n, ceil(D / 64)] int64 bit patterns of x2d[idx[i], :] != 0, counts [n] int32); idx None = every row in order."""
        dev = self._check_dev(x2d, idx)
        n = int(x2d.shape[0] if idx is None else idx.shape[0])
        D = int(x2d.shape[1])
        masks = torch.empty((n, (D + 63) // 64), dtype=torch.int64, device=dev)
        counts = torch.empty(n, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_rows_pack_count(n, _ptr(idx), 0 if idx is None else (32 if idx.dtype == torch.int32 else 64), D, _ptr(x2d),
                                                      x2d.stride(0), _ptr(masks), _ptr(counts), _stream(dev)))
        return masks, counts

    def rows_mask_count(self, masks, D):
        dev = self._check_dev(masks)
        counts = torch.empty(masks.shape[0], dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_rows_mask_count(masks.shape[0], int(D), _ptr(masks), _ptr(counts), _stream(dev)))
        return counts

    def rows_pack_values(self, x2d, idx, masks, offsets, total):
        """The values of x2d[idx[i], :] under masks[i], row i from offsets[i] on, as one [total] vector."""
        dev = self._check_dev(x2d, idx, masks, offsets)
        values = torch.empty(int(total), dtype=torch.float32, device=dev)
        n = int(masks.shape[0])
        if int(total) == 0:  # every mask empty (a layer whose boundary rows are all zero): nothing to write, and an empty tensor has no address
            return values
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_rows_pack_values(n, _ptr(idx), 0 if idx is None else (32 if idx.dtype == torch.int32 else 64),
                                                       int(x2d.shape[1]), _ptr(x2d), x2d.stride(0), _ptr(masks), _ptr(offsets), _ptr(values),
                                                       _stream(dev)))
        return values

    def rows_unpack(self, masks, offsets, values, D, out=None):
        """Dense [n, D] rows: values under the masks, zeros elsewhere."""
        dev = self._check_dev(masks, offsets, values, out)
        n = int(masks.shape[0])
        if out is None:
            out = torch.empty((n, int(D)), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_rows_unpack(n, int(D), _ptr(masks), _ptr(offsets), _ptr(values), _ptr(out), out.stride(0), _stream(dev)))
        return out

    def rows_unpack_add_csr(self, csr, masks, offsets, values, out):
        """out[v] += the packed rows at the positions listed in row v of `csr` (int32), in CSR order; out row-strided."""
        dev = self._check_dev(csr.indptr, masks, offsets, values, out)
        if csr.idx_bits != 32 or values.numel() >= 2 ** 32:
            raise DGLError("rows_unpack_add_csr: int32 CSR and fewer than 2^32 packed values only")
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_rows_unpack_add_csr(csr.num_rows, _ptr(csr.indptr), _ptr(csr.indices), int(out.shape[1]), _ptr(masks),
                                                          _ptr(offsets), _ptr(values), _ptr(out), out.stride(0), _stream(dev)))
        return out

    def gather_rows(self, x2d, idx):
        """out[i] = x2d[idx[i]]; x2d may be a row-strided view (stride(1) == 1) when D and the stride are multiples of 4."""
        dev = self._check_dev(x2d, idx)
        D = int(x2d.shape[1])
        out = torch.empty((idx.shape[0], D), dtype=torch.float32, device=dev)
        bits = 32 if idx.dtype == torch.int32 else 64
        with torch.cuda.device(dev):
            if (x2d.dim() == 2 and x2d.stride(1) == 1 and D % 4 == 0 and x2d.stride(0) % 4 == 0 and x2d.data_ptr() % 16 == 0 and D > 0
                    and x2d.stride(0) >= D):
                _lib.check(_lib.lib().mgx_gather_rows_strided(idx.shape[0], _ptr(idx), bits, D, _ptr(x2d), int(x2d.stride(0)), _ptr(out),
                                                              D, _stream(dev)))
            else:
                x2d = x2d.contiguous()
                _lib.check(_lib.lib().mgx_gather_rows(idx.shape[0], _ptr(idx), bits, D, _ptr(x2d), _ptr(out), _stream(dev)))
        return out

    def scatter_add_rows(self, x2d, idx, rows2d):
        dev = self._check_dev(x2d, idx, rows2d)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mgx_scatter_add_rows(idx.shape[0], _ptr(idx), 32 if idx.dtype == torch.int32 else 64,
                                                       x2d.shape[1], _ptr(rows2d), _ptr(x2d), _stream(dev)))
        return x2d


_BACKENDS = {"cuda": HipBackend()}


def register_backend(device_type, backend):
    """Hook used by tests/ only (a checker backend for CPU tensors in gloo tests)."""
    _BACKENDS[device_type] = backend


def backend_for(tensor):
    b = _BACKENDS.get(tensor.device.type)
    if b is None:
        raise DGLError(
            "message passing on %s tensors is not supported: this backend runs on MI355X (HIP) devices only; "
            "move the graph and features to a cuda/hip device" % tensor.device.type)
    return b


def coo_to_csr_host(num_rows, num_cols, row, col):
    """Stable COO->CSR for CPU index tensors (host half of the C ABI; needs no GPU)."""
    nnz = row.shape[0]
    bits = 32 if row.dtype == torch.int32 else 64
    row, col = row.contiguous(), col.contiguous()
    indptr = torch.empty(num_rows + 1, dtype=row.dtype)
    indices = torch.empty(nnz, dtype=row.dtype)
    eids = torch.empty(nnz, dtype=row.dtype)
    _lib.check(_lib.lib().mgx_coo_to_csr_host(num_rows, nnz, _ptr(row), _ptr(col), bits, _ptr(indptr), _ptr(indices), _ptr(eids)))
    return CsrView(num_rows, num_cols, indptr, indices, eids)


def csr_transpose(csr):
    """Device CSR -> its transpose (same edge ids); rows ordered by the positions in `csr`."""
    return backend_for(csr.indptr).csr_transpose(csr)


def coo_to_csr(num_rows, num_cols, row, col):
    if row.is_cuda:
        return _BACKENDS["cuda"].coo_to_csr(num_rows, num_cols, row.contiguous(), col.contiguous())
    return coo_to_csr_host(num_rows, num_cols, row, col)


# ----------------------------------------------------------------------------- raw ops
def _as_f32(t, what):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise DGLError("%s must be float32 on this backend, got %s" % (what, t.dtype))
    return t.contiguous()


_WIDE_PAD_MIN = 176  # above this width the g-SpMM runs in column passes of 128 columns (csrc/spmm.hip launch_fast_v)


def gspmm_raw(csr, op, reduce, U, E, src_scale=None, dst_scale=None, want_arg=False, accumulate_into=None, dense_out=False):
    """out[v] = reduce_{p in row v} op(U[indices[p]], E[eids[p]]).

    U: (num_cols, *ushape) or None; E: (nnz, *eshape) or None.  Returns (out, arg_u, arg_e) with out of
    shape (num_rows, *bcast(ushape, eshape)).  `out` may be a row-strided VIEW (the line-padded wide path below) unless
    `dense_out` is set -- callers that hand the result to a raw kernel (accumulate_into) ask for a dense one."""
    if op not in ("add", "mul", "copy_lhs", "copy_rhs"):
        raise DGLError("gspmm_raw: unsupported op %r (sub/div are rewritten by the caller)" % op)
    U = None if op == "copy_rhs" else _as_f32(U, "gspmm lhs feature")
    E = None if op == "copy_lhs" else _as_f32(E, "gspmm rhs feature")
    if op != "copy_rhs" and U is None:
        raise DGLError("gspmm: op %r needs node features" % op)
    if op != "copy_lhs" and E is None:
        raise DGLError("gspmm: op %r needs edge features" % op)
    if U is not None and U.shape[0] != csr.num_cols:
        raise DGLError("gspmm: expected %d source rows, got %d" % (csr.num_cols, U.shape[0]))
    if E is not None and E.shape[0] != csr.nnz:
        raise DGLError("gspmm: expected %d edge rows, got %d" % (csr.nnz, E.shape[0]))
    if (op == "copy_lhs" and reduce in ("sum", "mean") and U.dim() == 2 and U.is_cuda and accumulate_into is None
            and U.shape[1] > _WIDE_PAD_MIN and U.shape[1] % 32 and csr.nnz >= max(config.WIDE_PAD_MIN_NNZ, 64 * csr.num_cols) and config.WIDE_PAD):
        # Wide rows that are not whole 128-byte lines (reddit's 602 input features: 2408-byte rows) on a dense graph: every
        # 512-byte column pass of a gathered row straddles one more line and 16-byte lanes are misaligned.  Aggregating a copy
        # padded to whole lines and returning the [:, :D] view is faster by more than the copy costs once a row is gathered
        # ~64 times (reddit-shaped, 492 in-edges per node: D = 602 18.4 ms -> 0.3 + 14.4 ms; D = 300 9.7 -> 0.2 + 8.0 ms).
        D = U.shape[1]
        out, _, _ = gspmm_raw(csr, op, reduce, torch.nn.functional.pad(U, (0, (-D) % 32)), None, src_scale, dst_scale)
        return (out[:, :D].contiguous() if dense_out else out[:, :D]), None, None
    ushape = tuple(U.shape[1:]) if U is not None else ()
    eshape = tuple(E.shape[1:]) if E is not None else ()
    if U is not None and E is not None:
        oshape, u_tab, e_tab = _bcast_plan(ushape, eshape)
    else:
        oshape, u_tab, e_tab = (ushape if U is not None else eshape), None, None
    ref = U if U is not None else E
    dev = ref.device
    out, arg_u, arg_e = backend_for(ref).spmm(
        csr, op, reduce, U, E, _prod(ushape), _prod(eshape), _prod(oshape),
        _device_table(u_tab, dev), _device_table(e_tab, dev), src_scale, dst_scale, want_arg,
        **({"accumulate_into": accumulate_into} if accumulate_into is not None else {}))
    shape = (csr.num_rows,) + tuple(oshape)
    out = out.view(shape)
    if arg_u is not None:
        arg_u = arg_u.view(shape)
    if arg_e is not None:
        arg_e = arg_e.view(shape)
    return out, arg_u, arg_e


# nnz below which a separate pass over the gradient to flag its zero rows cannot pay


def gspmm_grad_raw(csr, dZ, dst_scale=None, accumulate_into=None, dense_out=False):
    """copy_u / sum over `csr` of a GRADIENT matrix dZ [num_cols, ...]: the backward aggregation of copy_u (dX = A^T dZ).
    Gradients of a loss taken on a subset of the nodes are zero in most rows at the last layer (ogbn-products: 92 %), so the
    rows of dZ are flagged first (one streaming pass) and the aggregation skips the all-zero ones -- same sum, fewer gathers.
    OFF by default (MGX_SPARSE_GRAD=1 enables it): measured on the products epoch the 92 %-zero launch drops only from
    2.41 to 1.65 ms (2.66 M work items of ~4 live edges each are bound by the per-item chain meta -> ids -> flags -> rows ->
    store, not by bytes) while the dense-gradient launch of the layer below pays 0.55 ms for the flagging pass and the flag
    gathers: no net gain (23.4 vs 23.5 ms).  Exact either way; kept with its test as a measured variant."""
    be = backend_for(dZ)
    flat = dZ.contiguous().view(dZ.shape[0], -1)
    if (config.SPARSE_GRAD and be.name == "hip" and csr.nnz >= config.SPARSE_GRAD_MIN_NNZ and csr.idx_bits == 32
            and flat.shape[1] % 4 == 0 and flat.shape[1] >= 4 and dZ.dtype == torch.float32 and dZ.shape[0] == csr.num_cols
            and csr.num_cols * flat.shape[1] * 4 < 2 ** 32 and flat.data_ptr() % 16 == 0):
        bits = be.row_nonzero_bits(flat)
        out = be.spmm_copy_u_masked(csr, "sum", flat, bits, dst_scale, accumulate_into)
        return out.view((csr.num_rows,) + tuple(dZ.shape[1:]))
    return gspmm_raw(csr, "copy_lhs", "sum", dZ, None, dst_scale=dst_scale, accumulate_into=accumulate_into, dense_out=dense_out)[0]


def gsddmm_raw(gidx, op, L, R, lhs_target="u", rhs_target="v"):
    """out[e] = op(L[t_l(e)], R[t_r(e)]) in edge-id order; `dot` reduces the last dim to size 1."""
    L = None if op == "copy_rhs" else _as_f32(L, "gsddmm lhs feature")
    R = None if op == "copy_lhs" else _as_f32(R, "gsddmm rhs feature")
    if op not in OP:
        raise DGLError("gsddmm: unsupported op %r" % op)
    if op != "copy_rhs" and L is None:
        raise DGLError("gsddmm: op %r needs the lhs operand" % op)
    if op != "copy_lhs" and R is None:
        raise DGLError("gsddmm: op %r needs the rhs operand" % op)
    expect = {"u": gidx.num_src, "v": gidx.num_dst, "e": gidx.num_edges()}
    for t, tgt, name in ((L, lhs_target, "lhs"), (R, rhs_target, "rhs")):
        if t is not None and t.shape[0] != expect[tgt]:
            raise DGLError("gsddmm: %s has %d rows, target %r has %d" % (name, t.shape[0], tgt, expect[tgt]))
    lshape = tuple(L.shape[1:]) if L is not None else ()
    rshape = tuple(R.shape[1:]) if R is not None else ()
    reduce_size = 1
    if op == "dot":
        if not lshape or not rshape or lshape[-1] != rshape[-1]:
            raise DGLError("gsddmm dot: last dims differ: %s vs %s" % (lshape, rshape))
        reduce_size = lshape[-1]
        oshape, l_tab, r_tab = _bcast_plan(lshape[:-1], rshape[:-1])
        out_len = _prod(oshape)
        oshape = tuple(oshape) + (1,)
    elif L is not None and R is not None:
        oshape, l_tab, r_tab = _bcast_plan(lshape, rshape)
        out_len = _prod(oshape)
    else:
        oshape, l_tab, r_tab = (lshape if L is not None else rshape), None, None
        out_len = _prod(oshape)
    ref = L if L is not None else R
    dev = ref.device
    out = backend_for(ref).sddmm(gidx, op, L, R, lhs_target, rhs_target, _prod(lshape), _prod(rshape), out_len,
                                 reduce_size, _device_table(l_tab, dev), _device_table(r_tab, dev))
    return out.view((gidx.num_edges(),) + tuple(oshape))


def edge_softmax_fwd_raw(csr, z):
    z = _as_f32(z, "edge_softmax logits")
    if z.shape[0] != csr.nnz:
        raise DGLError("edge_softmax: expected %d edge rows, got %d" % (csr.nnz, z.shape[0]))
    a = backend_for(z).edge_softmax_fwd(csr, z.view(z.shape[0], -1))
    return a.view(z.shape)


def edge_softmax_bwd_raw(csr, a, da):
    a, da = _as_f32(a, "edge_softmax out"), _as_f32(da, "edge_softmax grad")
    if a.shape[0] != csr.nnz or tuple(da.shape) != tuple(a.shape):
        raise DGLError("edge_softmax backward: expected two (%d, ...) tensors of one shape, got %s and %s"
                       % (csr.nnz, tuple(a.shape), tuple(da.shape)))
    dz = backend_for(a).edge_softmax_bwd(csr, a.view(a.shape[0], -1), da.view(da.shape[0], -1))
    return dz.view(a.shape)


def segment_reduce_raw(offsets, x, reduce="sum", want_arg=False, total=None):
    """`total`: the caller's host-side sum of the segment lengths, checked against x.shape[0] without a device sync."""
    x = _as_f32(x, "segment_reduce input")
    if total is None:
        total = int(offsets[-1].item()) if offsets.numel() else 0
    if total != x.shape[0]:
        raise DGLError("segment_reduce: segment lengths sum to %d, value has %d rows" % (total, x.shape[0]))
    out, arg = backend_for(x).segment_reduce(offsets, x.view(x.shape[0], -1), reduce, want_arg)
    shape = (offsets.shape[0] - 1,) + tuple(x.shape[1:])
    return out.view(shape), (None if arg is None else arg.view(shape))


def gather_rows_raw(x, idx):
    x = _as_f32(x, "gather_rows input")
    out = backend_for(x).gather_rows(x.view(x.shape[0], -1), idx)
    return out.view((idx.shape[0],) + tuple(x.shape[1:]))


def scatter_add_rows_raw(x, idx, rows):
    """x[idx[i]] += rows[i] in place; idx must be unique."""
    backend_for(x).scatter_add_rows(x.view(x.shape[0], -1), idx, _as_f32(rows, "rows").view(rows.shape[0], -1))
    return x
