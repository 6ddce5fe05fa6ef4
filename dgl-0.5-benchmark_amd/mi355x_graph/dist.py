"""Multi-GPU full-graph message passing: edge-cut node partition + RCCL all_to_all halo exchange.

New capability (the reference is single-GPU, README.md:12): one process per GPU, each
owning a set of destination nodes with ALL their in-edges (1-D row partition), so every aggregation
kernel is local; the only exchange step per layer is the boundary ("halo") source rows:

  forward   send X_own[send_idx] to every peer (packed by mgx_gather_rows into one buffer)
            -> torch.distributed.all_to_all_single over RCCL/xGMI (per-peer split sizes)
            -> local g-SpMM over [owned | halo] source rows
  backward  halo-row gradients travel the transposed all_to_all and are added into their owners
            with mgx_scatter_add_rows (one call per peer, fixed order => deterministic)

`DistGraph` keeps the DGLGraph surface (srcdata/dstdata/update_all/apply_edges/in_degrees over the
OWNED nodes) so the model classes of full_graph.py run unmodified on a partition.  Weight
gradients are summed with one flat all_reduce; the loss is normalised by the global train count so
the P-way run computes the same mean loss as the 1-GPU run (main_dgl_product_sage.py:105-106).
"""
import os

import torch
from torch.autograd.function import once_differentiable
import torch.distributed as dist
import torch.nn as nn

from ._lib import DGLError
from . import core, emulate, ops, schedule, sparse
from . import config
from . import function as fn
from .graph import DGLGraph, Frame, GraphIndex


# ----------------------------------------------------------------------------- partitioning
def partition_nodes(src, dst, num_nodes, num_parts, rounds=5, clusters=None):
    """Edge-cut node partition: label-propagation clusters packed into `num_parts` bins balanced on
    in-edge count (the per-GPU SpMM work), largest cluster first.  Returns (assign [N] int64, stats).
    METIS is not available offline; this is the build's own partitioner."""
    dev = src.device
    if num_parts == 1:
        return torch.zeros(num_nodes, dtype=torch.int64, device=dev), {"edge_cut": 0.0, "num_clusters": 1}
    indeg = torch.bincount(dst.long(), minlength=num_nodes)
    node_w = indeg + 1  # +1 so that isolated nodes still spread out
    if clusters is None:
        csc = sparse.coo_to_csr(num_nodes, num_nodes, dst.contiguous(), src.contiguous())
        cap = int(node_w.sum().item()) // (num_parts * 4) + 1
        hist = schedule.label_propagation(csc.indptr, csc.indices, num_nodes, rounds)
        clusters = _cap_clusters(hist, node_w, cap)
    uniq, inv = torch.unique(clusters, return_inverse=True)
    c_edges = torch.zeros(uniq.shape[0], dtype=torch.int64, device=dev).index_add_(0, inv, node_w)
    c_nodes = torch.zeros_like(c_edges)
    # cluster graph (who talks to whom, how much), then linear-deterministic-greedy placement:
    # heaviest cluster first, into the part it is most connected to, discounted by that part's load
    import numpy as np
    C = int(uniq.shape[0])
    cs, cd = inv[src.long()], inv[dst.long()]
    cross = cs != cd
    pair, cnt = torch.unique(cs[cross] * C + cd[cross], return_counts=True)
    pa = torch.div(pair, C, rounding_mode="floor")
    pb = pair - pa * C
    # symmetrise: affinity(a,b) = edges a->b + b->a
    a_all = torch.cat([pa, pb]).cpu().numpy()
    b_all = torch.cat([pb, pa]).cpu().numpy()
    w_all = torch.cat([cnt, cnt]).cpu().numpy().astype(np.float64)
    order_e = np.argsort(a_all, kind="stable")
    a_all, b_all, w_all = a_all[order_e], b_all[order_e], w_all[order_e]
    ptr = np.zeros(C + 1, np.int64)
    np.add.at(ptr, a_all + 1, 1)
    ptr = np.cumsum(ptr)
    weight = c_edges.cpu().numpy().astype(np.float64)
    capacity = 1.03 * weight.sum() / num_parts + weight.max() * 0.0
    part = np.full(C, -1, np.int64)
    load = np.zeros(num_parts)
    order = np.argsort(-weight, kind="stable")
    for sweep in range(3):  # first sweep places, later sweeps move clusters with full knowledge
        for c in order:
            nb, w = b_all[ptr[c]:ptr[c + 1]], w_all[ptr[c]:ptr[c + 1]]
            placed = part[nb] >= 0
            conn = np.bincount(part[nb][placed], weights=w[placed], minlength=num_parts) if placed.any() else np.zeros(num_parts)
            cur = part[c]
            if cur >= 0:
                load[cur] -= weight[c]
            room = load + weight[c] <= capacity
            if not room.any():
                room = load == load.min()
            score = np.where(room, (conn + 1e-9) * (1.0 - load / capacity), -np.inf)
            best = int(np.argmax(score))
            if conn.max() <= 0:  # no placed neighbour: lightest part
                best = int(np.argmin(np.where(room, load, np.inf)))
            part[c] = best
            load[best] += weight[c]
    assign = torch.from_numpy(part).to(dev)[inv]
    if num_nodes * num_parts <= 200_000_000:  # the refinement keeps a dense [N, P] neighbour-count table
        assign = _refine(assign, src.long(), dst.long(), node_w, num_parts, 1.03)
    cut = float((assign[src.long()] != assign[dst.long()]).float().mean().item()) if src.numel() else 0.0
    return assign, {"edge_cut": cut, "num_clusters": int(uniq.shape[0])}


def edge_fingerprint(src, dst):
    """Order-dependent-free checksum of an edge list (python int < 2^61): keys the partition cache and lets ranks check
    that they hold the same graph."""
    mix = (src.long() * 1000003 + dst.long() * 7919) % 2147483629
    return int(mix.sum().item()) % (1 << 61)


def cached_partition(src, dst, num_nodes, num_parts, cache_dir=None, **kwargs):
    """partition_nodes with the result kept on disk, as the reference caches its METIS partitions
    (cluster-sage/dgl/sampler.py:34-41): `<cache_dir>/partition_n<N>_e<E>_p<P>_<fingerprint>.pt`.  The fingerprint is a
    checksum of the edge list, so a different graph of the same size never reuses a stale file."""
    cache_dir = cache_dir or os.environ.get("MGX_CACHE_DIR") or os.path.join(os.path.expanduser("~"), ".cache", "mi355x_graph")
    finger = edge_fingerprint(src, dst)
    path = os.path.join(cache_dir, "partition_n%d_e%d_p%d_%x.pt" % (num_nodes, src.shape[0], num_parts, finger))
    if os.path.exists(path):
        try:
            blob = torch.load(path, map_location=src.device, weights_only=True)
            if blob["assign"].shape[0] == num_nodes:
                stats = dict(blob["stats"], cached=True)
                return blob["assign"].to(src.device), stats
        except Exception:  # unreadable cache: recompute
            pass
    assign, stats = partition_nodes(src, dst, num_nodes, num_parts, **kwargs)
    try:
        os.makedirs(cache_dir, exist_ok=True)
        tmp = path + ".tmp%d" % os.getpid()
        torch.save({"assign": assign.cpu(), "stats": stats}, tmp)
        os.replace(tmp, path)
    except OSError:
        pass  # read-only location: run without the cache
    return assign, dict(stats, cached=False)


def _refine(assign, src, dst, node_w, num_parts, slack, sweeps=4):
    """Balanced label-propagation refinement at node level: a node moves to the part holding most of
    its neighbours when that reduces the cut and the target part has room (heaviest gains first)."""
    n = assign.shape[0]
    dev = assign.device
    total = float(node_w.sum().item())
    cap = slack * total / num_parts
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    for it in range(sweeps):
        cnt = torch.zeros(n * num_parts, dtype=torch.float32, device=dev)
        ones = torch.ones(src.shape[0], dtype=torch.float32, device=dev)
        cnt.index_add_(0, dst * num_parts + assign[src], ones)
        cnt.index_add_(0, src * num_parts + assign[dst], ones)
        cnt = cnt.view(n, num_parts)
        cur = cnt.gather(1, assign[:, None]).squeeze(1)
        best_cnt, best = cnt.max(1)
        gain = best_cnt - cur
        cand = (gain > 0) & (best != assign) & (torch.rand(n, generator=gen, device=dev) < 0.5)
        if not bool(cand.any()):
            break
        load = torch.zeros(num_parts, dtype=torch.float64, device=dev).index_add_(0, assign, node_w.double())
        idx = torch.nonzero(cand).flatten()
        # per target part: accept candidates in order of decreasing gain while the part has room
        key = best[idx].double() * 1e9 - gain[idx].double()
        idx = idx[torch.sort(key)[1]]
        tgt = best[idx]
        w = node_w[idx].double()
        first = torch.ones_like(tgt, dtype=torch.bool)
        first[1:] = tgt[1:] != tgt[:-1]
        csum = torch.cumsum(w, 0)
        base = (csum - w)[first]
        seg = torch.cumsum(first.long(), 0) - 1
        within = csum - base[seg]
        room = (cap - load)[tgt]
        ok = within <= room
        moved = idx[ok]
        assign = assign.clone()
        assign[moved] = tgt[ok]
    return assign


def _cap_clusters(hist, node_w, cap):
    """Clusters heavier than `cap` fall back to the finer labels of earlier LP rounds; what is still
    too heavy is cut into chunks of cumulative weight <= cap (node-id order)."""
    n = node_w.shape[0]
    dev = node_w.device
    labels = hist[-1].clone()

    def heavy_mask(lab):
        u, inv = torch.unique(lab, return_inverse=True)
        w = torch.zeros(u.shape[0], dtype=torch.int64, device=dev).index_add_(0, inv, node_w)
        return (w > cap)[inv]

    for level in range(len(hist) - 2, -1, -1):
        big = heavy_mask(labels)
        if not bool(big.any()):
            return labels
        labels = torch.where(big, hist[level] + n * (len(hist) - 1 - level), labels)
    big = heavy_mask(labels)
    if bool(big.any()):
        order = torch.sort(labels, stable=True)[1]
        lab_s, w_s = labels[order], node_w[order]
        csum = torch.cumsum(w_s, 0)
        first = torch.ones_like(lab_s, dtype=torch.bool)
        first[1:] = lab_s[1:] != lab_s[:-1]
        start_csum = (csum - w_s)[first]                      # cumulative weight before each cluster
        cid = torch.cumsum(first.long(), 0) - 1
        chunk = torch.div(csum - w_s - start_csum[cid], cap, rounding_mode="floor")
        new = lab_s * 0 + cid * (int(chunk.max().item()) + 1) + chunk + n * (len(hist) + 1)
        labels = labels.clone()
        labels[order] = torch.where(big[order], new, lab_s)
    return labels


class HaloPlan(object):
    """What one rank sends / receives per layer."""

    def __init__(self, rank, world, n_own, n_halo, send_idx, send_splits, recv_splits, loc=None, halo=None, inv_deg=None):
        self.loc, self.halo, self.inv_deg = loc, halo, inv_deg  # split structure for the overlapped copy_u path
        self.rank, self.world = rank, world
        self.n_own, self.n_halo = n_own, n_halo
        self.send_idx = send_idx            # local owned-row ids, grouped by destination peer
        self.send_splits = send_splits      # python ints, len world
        self.recv_splits = recv_splits
        off = 0
        self.send_ranges = []
        for c in send_splits:
            self.send_ranges.append((off, off + c))
            off += c
        self._ret = None

    def return_csr(self):
        """CSR over the OWNED rows whose entries are positions of the returned halo-gradient buffer (`back`, grouped by
        peer like send_idx): adding the rows that come back into their owners is then ONE copy_e/sum g-SpMM accumulating
        into the gradient -- rows that went to several peers are summed in peer order (stable sort), so the result is
        deterministic, and there is one launch instead of one scatter per peer."""
        if self._ret is None:
            pos = torch.arange(self.send_idx.shape[0], dtype=self.send_idx.dtype, device=self.send_idx.device)
            self._ret = sparse.coo_to_csr(self.n_own, max(int(pos.shape[0]), 1), self.send_idx.contiguous(), pos)
        return self._ret

    def add_returned_rows(self, gx, back):
        if back.shape[0]:
            sparse.gspmm_raw(self.return_csr(), "copy_rhs", "sum", None, back, accumulate_into=gx)
        return gx


def build_local_partition(src, dst, num_nodes, assign, rank, world, idtype=torch.int32):
    """From the GLOBAL edge list (every rank holds it in this benchmark) build rank's local block graph
    over [owned | halo] sources, its halo plan, and the owned global ids.  No communication."""
    dev = src.device
    src, dst = src.long(), dst.long()
    a_src, a_dst = assign[src], assign[dst]
    own = torch.nonzero(assign == rank).flatten()
    n_own = int(own.shape[0])
    g2l = torch.full((num_nodes,), -1, dtype=torch.int64, device=dev)
    g2l[own] = torch.arange(n_own, device=dev)
    mine = a_dst == rank
    es, ed, eo = src[mine], dst[mine], a_src[mine]
    remote = eo != rank
    # halo = distinct remote sources, ordered by (owner, global id): the order peers send them in
    key = eo[remote] * num_nodes + es[remote]
    hkey = torch.unique(key)
    h_owner = torch.div(hkey, num_nodes, rounding_mode="floor")
    n_halo = int(hkey.shape[0])
    recv_splits = torch.bincount(h_owner, minlength=world).cpu().tolist()
    l_src = g2l[es]
    l_src[remote] = n_own + torch.searchsorted(hkey, key)
    l_dst = g2l[ed]
    # what I must send: distinct (peer, my node) pairs over the edges leaving my part
    out = (a_src == rank) & (a_dst != rank)
    skey = torch.unique(a_dst[out] * num_nodes + src[out])
    s_peer = torch.div(skey, num_nodes, rounding_mode="floor")
    send_idx = g2l[skey - s_peer * num_nodes]
    send_splits = torch.bincount(s_peer, minlength=world).cpu().tolist()
    block = DGLGraph(GraphIndex(n_own + n_halo, n_own, coo=(l_src.to(idtype).contiguous(), l_dst.to(idtype).contiguous())),
                     is_block=True)
    block._index.csc().dst_is_src_prefix = True  # [owned | halo]: lets the schedule cluster the owned x owned part
    # the same edges split by source ownership: the owned x owned part can run while the halo rows travel
    keep = ~remote
    loc = GraphIndex(n_own, n_own, coo=(l_src[keep].to(idtype).contiguous(), l_dst[keep].to(idtype).contiguous()))
    halo = GraphIndex(n_halo, n_own, coo=((l_src[remote] - n_own).to(idtype).contiguous(),
                                          l_dst[remote].to(idtype).contiguous()))
    inv_deg = 1.0 / torch.bincount(l_dst, minlength=n_own).clamp(min=1).to(torch.float32)
    plan = HaloPlan(rank, world, n_own, n_halo, send_idx.to(idtype).contiguous(), send_splits, recv_splits,
                    loc, halo, inv_deg)
    return block, plan, own


# ----------------------------------------------------------------------------- exchange
class _Comm(object):
    """Thin wrapper so tests can run the same code over gloo (CPU) and the bench over RCCL."""

    def __init__(self, group=None):
        self.group = group
        self.n_exchanges = 0
        self.trace = None  # bench.py sets a list: (event before wait, event after wait, bytes received) per exchange of the timed region

    def mark(self, label):
        """Names the stretch of work that starts here -- recorded only by an emulated rank (emulate.EmuRank.mark)."""
        emu = emulate.current()
        if emu is not None:
            emu.mark(label)

    def all_to_all(self, out, inp, out_splits, in_splits):
        self.n_exchanges += 1
        emu = emulate.current()
        if emu is not None:  # P ranks in one process (emulate.py): in-process row copies
            return emu.all_to_all(out, inp, out_splits, in_splits)
        if dist.get_backend(self.group) == "gloo":  # tests: pairwise isend/irecv staged through host memory
            world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
            outs = list(out.split(out_splits, 0))
            inps = list(inp.split(in_splits, 0))
            outs[rank].copy_(inps[rank])
            reqs, landing = [], []
            for p in range(world):
                if p == rank:
                    continue
                if in_splits[p]:
                    reqs.append(dist.isend(inps[p].detach().cpu().contiguous(), p, group=self.group))
                if out_splits[p]:
                    buf = torch.empty(outs[p].shape, dtype=outs[p].dtype)
                    landing.append((outs[p], buf))
                    reqs.append(dist.irecv(buf, p, group=self.group))
            for r in reqs:
                r.wait()
            for dst_t, buf in landing:
                dst_t.copy_(buf)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def all_to_all_async(self, out, inp, out_splits, in_splits, count=True, tag="rows"):
        """Returns a handle whose wait() orders the CURRENT stream after the exchange (RCCL runs it on the
        process group's own stream, so kernels launched in between overlap with it).  count=False: the second message of ONE
        logical halo exchange (SparseHalo's values after its bitmaps) -- n_exchanges counts exchanges, not messages."""
        emu = emulate.current()
        if emu is not None:
            self.n_exchanges += 1 if count else 0
            return emu.all_to_all_async(out, inp, out_splits, in_splits, tag=tag)
        if dist.get_backend(self.group) == "gloo":
            self.all_to_all(out, inp, out_splits, in_splits)
            self.n_exchanges -= 0 if count else 1
            return _Done()
        self.n_exchanges += 1 if count else 0
        work = dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=True)
        if self.trace is None or not out.is_cuda:
            return work
        return _TimedWork(work, self.trace, int(out.numel() * out.element_size()))


class _Done(object):
    def wait(self):
        return True


class _TimedWork(object):
    """An RCCL work handle whose wait() is bracketed by two HIP events on the compute stream: what lies between them is the time the
    compute stream stood still for the collective -- the EXPOSED part of the exchange, the number the scaling model predicts."""

    def __init__(self, work, trace, nbytes):
        self.work, self.trace, self.nbytes = work, trace, nbytes

    def wait(self):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ok = self.work.wait()
        b.record()
        self.trace.append((a, b, self.nbytes))
        return ok


# ----------------------------------------------------------------------------- sparse halo exchange (round 5)
SPARSE_HALO = os.environ.get("MGX_SPARSE_HALO", "1") == "1"
SPARSE_EXCHANGES = [0, 0]  # forward / backward exchanges that took the packed form in this process (tests, bench.py's report)


def structural_zeros(t):
    """True when the caller marked `t` as the output of relu (+ dropout): its zeros are structural -- whoever produced it multiplies
    the gradient arriving at a zero position by zero (ops.relu_dropout tags its result; torch.relu outputs can be tagged with
    mark_structural_zeros).  Only then may the halo exchange drop those gradient entries.  The tag holds the tensor's version counter
    at tagging time: an in-place write afterwards (h.sub_(c), h[i] = 0 ...) may create zeros that are NOT annihilated by the producer's
    backward, so it voids the tag (ops.has_structural_zeros)."""
    from . import ops
    return ops.has_structural_zeros(t)


def mark_structural_zeros(t):
    t._mgx_structural_zeros = int(t._version)
    return t


def _bounds(splits, device):
    out, off = [0], 0
    for c in splits:
        off += int(c)
        out.append(off)
    return torch.tensor(out, dtype=torch.int64, device=device)


def _exclusive_scan(counts):
    off = torch.zeros(counts.shape[0] + 1, dtype=torch.int64, device=counts.device)
    if counts.shape[0]:
        torch.cumsum(counts, 0, dtype=torch.int64, out=off[1:])
    return off


class SparseHalo(object):
    """One layer's halo exchange with the boundary rows as bitmaps + packed non-zeros (csrc/rowpack.hip), both directions:

      forward   masks of the boundary rows -> all_to_all #1 (sizes known: rows x mask words); the value counts per peer are read
                back ONCE (the only host synchronisation: all_to_all_single takes host sizes) -> values -> all_to_all #2 (async);
                finish(): dense halo rows for the aggregation kernel, zeros restored
      backward  the gradient of a halo row travels as the values under the SAME mask (no mask, no size exchange: both sides
                kept the forward's); finish_back(): dense [send rows, D], zeros elsewhere, for the one copy_u over return_csr

    Valid for inputs whose zeros are structural (see structural_zeros): a gradient entry at a zero position is multiplied by relu's /
    dropout's zero at the owner, so dropping it changes no result.  Exact: values are moved, never rounded.  At D = 64 and 75 % zeros a
    row is 8 + 64 bytes forward and 64 bytes back instead of 256 each way."""

    def __init__(self, comm, plan, be, D):
        self.comm, self.plan, self.be, self.D = comm, plan, be, int(D)
        dev = plan.send_idx.device
        if getattr(plan, "_sparse_bounds", None) is None:
            plan._sparse_bounds = (_bounds(plan.send_splits, dev), _bounds(plan.recv_splits, dev))
        self.work = self.bwork = None

    def post(self, h):
        plan, be, comm = self.plan, self.be, self.comm
        SPARSE_EXCHANGES[0] += 1
        W = (self.D + 63) // 64
        idx = plan.send_idx if plan.send_idx.numel() else None
        if idx is None:
            self.smask = torch.empty((0, W), dtype=torch.int64, device=h.device)
            scnt = torch.empty(0, dtype=torch.int32, device=h.device)
        else:
            self.smask, scnt = be.rows_pack_count(h, idx)
        self.rmask = torch.empty((plan.n_halo, W), dtype=torch.int64, device=h.device)
        w1 = comm.all_to_all_async(self.rmask, self.smask, plan.recv_splits, plan.send_splits, tag="bitmaps")
        self.soff = _exclusive_scan(scnt)
        w1.wait()
        self.roff = _exclusive_scan(be.rows_mask_count(self.rmask, self.D) if plan.n_halo else scnt[:0])
        sb, rb = plan._sparse_bounds
        host = torch.cat([self.soff[sb], self.roff[rb]]).cpu().tolist()  # the one host read of the exchange
        k = len(plan.send_splits) + 1
        self.s_splits = [host[i + 1] - host[i] for i in range(k - 1)]
        self.r_splits = [host[k + i + 1] - host[k + i] for i in range(k - 1)]
        self.total_s, self.total_r = host[k - 1], host[2 * k - 1]
        svals = (be.rows_pack_values(h, idx, self.smask, self.soff, self.total_s) if idx is not None
                 else torch.empty(0, dtype=torch.float32, device=h.device))
        self.rvals = torch.empty(self.total_r, dtype=torch.float32, device=h.device)
        self._sent = svals  # alive until the exchange has completed
        self.work = comm.all_to_all_async(self.rvals, svals, self.r_splits, self.s_splits, count=False, tag="values")
        return self

    def finish(self):
        """The received halo rows, dense [n_halo, D]."""
        self.work.wait()
        self._sent = None
        out = torch.empty((self.plan.n_halo, self.D), dtype=torch.float32, device=self.rmask.device)
        if self.plan.n_halo:
            self.be.rows_unpack(self.rmask, self.roff, self.rvals, self.D, out=out)
        self.rvals = None
        return out

    def bytes_forward(self):
        return 8 * int(self.rmask.numel()) + 4 * self.total_r

    def post_back(self, g_halo):
        SPARSE_EXCHANGES[1] += 1
        gv = (self.be.rows_pack_values(g_halo, None, self.rmask, self.roff, self.total_r) if self.plan.n_halo
              else torch.empty(0, dtype=torch.float32, device=g_halo.device))
        self.bvals = torch.empty(self.total_s, dtype=torch.float32, device=g_halo.device)
        self._sent = gv
        self.bwork = self.comm.all_to_all_async(self.bvals, gv, self.s_splits, self.r_splits, tag="gradient values")
        return self

    def finish_back_into(self, dh):
        """dh[v] += the returned gradient rows of owned row v (peer order), straight from the packed form when the backend has the fused
        kernel (mgx_rows_unpack_add_csr); else through the dense rows and the plan's copy_u over return_csr."""
        plan, be = self.plan, self.be
        if not hasattr(be, "rows_unpack_add_csr") or plan.return_csr().idx_bits != 32 or not plan.send_idx.numel():
            back = self.finish_back()
            if back.shape[0]:
                be.spmm_copy_u_strided(plan.return_csr(), "sum", back, dh, accumulate=True)
            return dh
        self.bwork.wait()
        self._sent = None
        be.rows_unpack_add_csr(plan.return_csr(), self.smask, self.soff, self.bvals, dh)
        self.bvals = None
        return dh

    def finish_back(self):
        """The returned gradient rows, dense [send rows, D] in send_idx order (zeros where the forward row was zero)."""
        self.bwork.wait()
        self._sent = None
        n = int(self.plan.send_idx.shape[0])
        out = torch.empty((n, self.D), dtype=torch.float32, device=self.bvals.device)
        if n:
            self.be.rows_unpack(self.smask, self.soff, self.bvals, self.D, out=out)
        self.bvals = None
        return out


def sparse_halo_applies(h, plan):
    """Every rank takes the same decision: the tag is set by the program, the width by the model."""
    if not SPARSE_HALO or not structural_zeros(h) or h.dim() != 2 or h.dtype != torch.float32:
        return False
    be = sparse.backend_for(h)
    return hasattr(be, "rows_pack_count") and be.rows_pack_supported(h)


class DistCopyU(torch.autograd.Function):
    """update_all(copy_u, sum|mean) on a partition with the exchange hidden behind the local work:

      forward   pack boundary rows -> all_to_all (async, RCCL stream) || g-SpMM over owned sources
                -> wait -> g-SpMM over halo sources accumulating into the same output
      backward  g-SpMM^T producing halo-row gradients -> all_to_all (async) || g-SpMM^T over owned rows
                -> wait -> add received rows into their owners (per peer, fixed order)
    Both halves write through dst_scale = 1/max(deg,1) for `mean` (deg = full in-degree).
    `static_cache` (a dict, or None): the caller declared THIS tensor constant (DistGraph.set_static_input: the node
    features of layer 1), so its halo rows stay resident on the receiving rank after the first exchange.  The dict holds
    the tensor itself (identity, not an address that the allocator may hand to another tensor) and its version counter:
    an in-place update exchanges again.  Every rank runs the same program on the same declaration, so all ranks hit or
    miss together.  The aggregation itself always runs."""

    @staticmethod
    def forward(ctx, x, plan, comm, reduce, static_cache=None, sparse_exchange=False):
        x = x.contiguous()
        feat = tuple(x.shape[1:])
        halo_x = None
        if static_cache is not None and static_cache.get("recv") is not None and static_cache.get("version") == x._version:
            recv, work = static_cache["recv"], _Done()
        elif sparse_exchange and static_cache is None:
            halo_x = SparseHalo(comm, plan, sparse.backend_for(x), x.shape[1]).post(x)
            recv, work = None, None
        else:
            send = sparse.gather_rows_raw(x, plan.send_idx)
            recv = torch.empty((plan.n_halo,) + feat, dtype=x.dtype, device=x.device)
            work = comm.all_to_all_async(recv, send, plan.recv_splits, plan.send_splits)
            if static_cache is not None:
                static_cache["version"], static_cache["recv"] = x._version, recv
        scale = plan.inv_deg if reduce == "mean" else None
        # dense_out: the halo aggregation below accumulates into `out` through the raw kernel (a line-padded view would be
        # written at the wrong offsets)
        out, _, _ = sparse.gspmm_raw(plan.loc.csc(), "copy_lhs", "sum", x, None, dst_scale=scale, dense_out=True)
        if halo_x is not None:
            recv = halo_x.finish()
        else:
            work.wait()
        if plan.n_halo:
            sparse.gspmm_raw(plan.halo.csc(), "copy_lhs", "sum", recv, None, dst_scale=scale, accumulate_into=out)
        ctx.plan, ctx.comm, ctx.reduce, ctx.halo_x = plan, comm, reduce, halo_x
        return out

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dZ):
        plan, comm = ctx.plan, ctx.comm
        dZ = dZ.contiguous()
        if ctx.reduce == "mean":
            dZ = dZ * plan.inv_deg.view((-1,) + (1,) * (dZ.dim() - 1))
        feat = tuple(dZ.shape[1:])
        if plan.n_halo:
            g_halo = sparse.gspmm_grad_raw(plan.halo.csr(), dZ)
        else:
            g_halo = torch.empty((0,) + feat, dtype=dZ.dtype, device=dZ.device)
        halo_x = ctx.halo_x
        if halo_x is not None:
            halo_x.post_back(g_halo.contiguous())
        else:
            back = torch.empty((plan.send_idx.shape[0],) + feat, dtype=dZ.dtype, device=dZ.device)
            work = comm.all_to_all_async(back, g_halo, plan.send_splits, plan.recv_splits)
        gx = sparse.gspmm_grad_raw(plan.loc.csr(), dZ, dense_out=True)  # add_returned_rows accumulates into it
        if halo_x is not None:
            back = halo_x.finish_back()
        else:
            work.wait()
        plan.add_returned_rows(gx, back)
        return gx, None, None, None, None, None


class DistSageMeanCatFn(torch.autograd.Function):
    """ops.SageMeanCatFn on a partition: one mean-aggregator GraphSAGE layer (main_dgl_product_sage.py:52-64) as ONE GEMM on
    [h | neigh] with the halo exchange hidden behind the local aggregation -- the layer every rank of `bench.py --gpus N` runs, so
    that N = 1 and N > 1 time the same module form.

      forward   pack boundary rows of h -> all_to_all (async) || aggregation over owned sources, left half -> right half in place
                -> wait -> aggregation over the received halo rows accumulating into the right half -> ONE GEMM
      backward  ONE GEMM gives d[h | neigh]; halo-row gradients first -> all_to_all (async) || reversed aggregation over owned
                rows accumulating into the left half -> wait -> returned rows added into their owners (fixed order)
    `static_cache`: DistGraph.set_static_input's declaration for the layer-1 input (its halo rows stay resident)."""

    @staticmethod
    def forward(ctx, plan, comm, cat, h, w_self, w_neigh, bias, static_cache, sparse_exchange=False, act=None):
        be = sparse.backend_for(h)
        comm.mark("pack")
        sparse_exchange = sparse_exchange and static_cache is None  # (decided by the caller on the tensor it holds: DistGraph.sage_mean_layer)
        halo_x = None
        if static_cache is not None and static_cache.get("recv") is not None and static_cache.get("version") == h._version:
            recv, work = static_cache["recv"], _Done()
        else:
            if sparse_exchange:
                # relu + dropout output: the rows travel as bitmaps + non-zeros, their gradients come back under the same bitmaps
                halo_x = SparseHalo(comm, plan, be, h.shape[1]).post(h)
                recv, work = None, halo_x.work
            else:
                # boundary rows packed straight out of the (row-strided) left half of the layer's buffer: mgx_gather_rows_strided
                send = be.gather_rows(h, plan.send_idx) if plan.send_idx.numel() else h.new_empty((0, h.shape[1]))
                recv = torch.empty((plan.n_halo, h.shape[1]), dtype=h.dtype, device=h.device)
                work = comm.all_to_all_async(recv, send, plan.recv_splits, plan.send_splits)
            if static_cache is not None:
                static_cache["version"], static_cache["recv"] = h._version, recv
        if not cat.holds(h):  # the layer-1 input lives elsewhere: copied into the left half unless it is the same unmodified tensor
            same = cat.static_key is not None and cat.static_key[0] is h and cat.static_key[1] == h._version
            if not same or h.requires_grad or not config.SAGE_STATIC_CAT:
                cat.left.copy_(h)
                cat.static_key = None if h.requires_grad else (h, h._version)
        cat.generation += 1
        comm.mark("owned-source aggregation")
        from . import ops
        tail_ops = ops._edge_tail_operands(be, plan.loc.csc(), cat, h) if hasattr(be, "edge_tail_of") else None
        if tail_ops is not None:  # the constant 100-column input, as on one GPU: [n_own, 96] + its last four columns along the owned edges
            be.spmm_copy_u_edge_tail(plan.loc.csc(), "sum", tail_ops[0], tail_ops[1], cat.right, dst_scale=plan.inv_deg)
        else:
            # (a relu + dropout output is gathered as 128-byte slots, as on one GPU: ops._packed_rows)
            be.spmm_copy_u_strided(plan.loc.csc(), "sum", cat.left, cat.right, dst_scale=plan.inv_deg,
                                   slots=ops._packed_rows(be, plan.loc, plan.loc.csc(), h, cat.left) if hasattr(be, "rows_slots_pack") else None)
        if halo_x is not None:
            comm.mark("unpack")
            recv = halo_x.finish()
        else:
            work.wait()
        comm.mark("halo-source aggregation")
        if plan.n_halo:
            be.spmm_copy_u_strided(plan.halo.csc(), "sum", recv, cat.right, accumulate=True, dst_scale=plan.inv_deg)
        ctx.plan, ctx.comm, ctx.cat, ctx.generation = plan, comm, cat, cat.generation
        ctx.halo_x = halo_x
        comm.mark("dense")
        wcat = torch.cat([w_self, w_neigh], dim=1)
        if act is not None:
            # dropout(relu(.)) in the GEMM's epilogue, written into the next layer's left half (ops.SageMeanCatFn's fused form: same
            # position in the random stream as ops.relu_dropout would take, same bits); the N x out pre-activation is never stored
            p, into = act
            seed = torch.initial_seed() & (2 ** 64 - 1)
            offset = (ops.ReluDropout._calls * 0x9E3779B97F4A7C15) & (2 ** 63 - 1)
            ops.ReluDropout._calls += 1
            out = None if into is None else into.t
            fused = be.rows_gemm_relu_dropout(cat.buf, wcat, True, bias, float(p), seed, offset, out=out)
            y, mask = fused if fused is not None else be.relu_dropout_fwd(ops._rows_linear(be, cat.buf, wcat, bias), float(p), seed, offset, out=out)
            ctx.save_for_backward(w_self, w_neigh, mask)
            ctx.p = float(p)
            return y
        ctx.save_for_backward(w_self, w_neigh)
        ctx.p = None
        return ops._rows_linear(be, cat.buf, wcat, bias)

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dy):
        from . import ops
        plan, comm, cat = ctx.plan, ctx.comm, ctx.cat
        if cat.generation != ctx.generation:
            raise DGLError("SAGEConv: a later forward pass overwrote the [h | neigh] buffer this backward pass needs")
        be = sparse.backend_for(dy)
        if ctx.p is not None:
            w_self, w_neigh, mask = ctx.saved_tensors
            if not dy.is_contiguous() and not (hasattr(be, "_row_strided") and be._row_strided(dy)):
                dy = dy.contiguous()
            dy = be.relu_dropout_bwd(dy, mask, ctx.p)  # the gradient of the pre-activation, as ops.ReluDropout.backward forms it
        else:
            w_self, w_neigh = ctx.saved_tensors
        dy = dy.contiguous()
        need = ctx.needs_input_grad
        K = cat.K
        dh = None
        if need[3]:
            dh_own, dn = ops._rows_dgrad(be, dy, torch.cat([w_self, w_neigh], dim=1), plan.inv_deg, K)  # d h, d neigh / deg: [n_own, K] each
            halo_x = ctx.halo_x
            g_halo = torch.empty((plan.n_halo, K), dtype=dy.dtype, device=dy.device)
            comm.mark("halo-row gradients")
            if plan.n_halo:
                be.spmm_copy_u_strided(plan.halo.csr(), "sum", dn, g_halo)
            if halo_x is not None:  # the values under the forward's bitmaps only: what lies elsewhere meets relu's / dropout's zero at the owner
                comm.mark("pack")
                halo_x.post_back(g_halo)
            else:
                back = torch.empty((plan.send_idx.shape[0], K), dtype=dy.dtype, device=dy.device)
                work = comm.all_to_all_async(back, g_halo, plan.send_splits, plan.recv_splits)
            comm.mark("owned-row reversed aggregation")
            be.spmm_copy_u_strided(plan.loc.csr(), "sum", dn, dh_own, accumulate=True)
        # the parameter gradients need nothing from the peers: formed while the halo-row gradients travel (the scaling model,
        # profiles/r04_scale_model.txt, has the exchange at 2x the reversed aggregation it used to hide behind at P = 8)
        comm.mark("dense (inside the exchange window)")
        dws = dwn = db = None
        if (need[4] or need[5]) and need[6]:
            dw, db = ops._weight_bias_grad(dy, cat.buf)  # the bias gradient from the weight-gradient kernel's own pass over dy
            dws, dwn = dw[:, :K].contiguous(), dw[:, K:].contiguous()
        elif need[4] or need[5]:
            dw = ops._weight_grad(dy, cat.buf)
            dws, dwn = dw[:, :K].contiguous(), dw[:, K:].contiguous()
        elif need[6]:
            db = be.column_sum(dy)
        if need[3]:
            if halo_x is not None:
                comm.mark("return-add")
                halo_x.finish_back_into(dh_own)  # packed values -> their owners' rows, one kernel
            else:
                work.wait()
                comm.mark("return-add")
                if back.shape[0]:  # row v += the returned rows whose owner is v: copy_u over (owned row -> position in `back`)
                    be.spmm_copy_u_strided(plan.return_csr(), "sum", back, dh_own, accumulate=True)
            dh = dh_own
        comm.mark("dense")
        return None, None, None, dh, dws, dwn, db, None, None, None


class DistSageProjectFirstFn(torch.autograd.Function):
    """ops.SageMeanProjectFirstFn on a partition: y = h W_self^T + mean_agg(h W_neigh^T) + b with the aggregation -- and the halo exchange
    -- at the OUTPUT width (main_dgl_reddit_sage.py:73-80: 602 -> 16, so a boundary row travels as 16 floats instead of 602):

      forward   s = h W_self^T (+ b), z = h W_neigh^T; pack boundary rows of z -> all_to_all (async) || aggregation of z over owned
                sources accumulating into s -> wait -> aggregation of the received z rows accumulating into s
      backward  dn = dy / deg; halo-row gradients of z first -> all_to_all (async) || reversed aggregation over owned rows -> wait ->
                returned rows added; dW_self = dy^T h, dW_neigh = dz^T h, dh = dy W_self + dz W_neigh."""

    @staticmethod
    def forward(ctx, plan, comm, h, w_self, w_neigh, bias):
        be = sparse.backend_for(h)
        comm.mark("dense")
        s = torch.nn.functional.linear(h, w_self, bias)
        z = torch.nn.functional.linear(h, w_neigh)
        comm.mark("pack")
        send = be.gather_rows(z, plan.send_idx) if plan.send_idx.numel() else z.new_empty((0, z.shape[1]))
        recv = torch.empty((plan.n_halo, z.shape[1]), dtype=z.dtype, device=z.device)
        work = comm.all_to_all_async(recv, send, plan.recv_splits, plan.send_splits)
        comm.mark("owned-source aggregation")
        sparse.gspmm_raw(plan.loc.csc(), "copy_lhs", "sum", z, None, dst_scale=plan.inv_deg, accumulate_into=s)
        work.wait()
        comm.mark("halo-source aggregation")
        if plan.n_halo:
            sparse.gspmm_raw(plan.halo.csc(), "copy_lhs", "sum", recv, None, dst_scale=plan.inv_deg, accumulate_into=s)
        comm.mark("dense")
        ctx.plan, ctx.comm = plan, comm
        ctx.save_for_backward(h, w_self, w_neigh)
        return s

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dy):
        from . import ops
        h, w_self, w_neigh = ctx.saved_tensors
        plan, comm = ctx.plan, ctx.comm
        need = ctx.needs_input_grad
        be = sparse.backend_for(dy)
        dy = dy.contiguous()
        dn = dy * plan.inv_deg.view(-1, 1)  # d(sum / deg)
        K = dy.shape[1]
        back = torch.empty((plan.send_idx.shape[0], K), dtype=dy.dtype, device=dy.device)
        comm.mark("halo-row gradients")
        g_halo = sparse.gspmm_grad_raw(plan.halo.csr(), dn) if plan.n_halo else dy.new_empty((0, K))
        work = comm.all_to_all_async(back, g_halo, plan.send_splits, plan.recv_splits)
        comm.mark("owned-row reversed aggregation")
        dz = sparse.gspmm_grad_raw(plan.loc.csr(), dn, dense_out=True)
        comm.mark("dense (inside the exchange window)")
        dws = ops._weight_grad(dy, h) if need[3] else None
        db = be.column_sum(dy) if need[5] else None
        work.wait()
        comm.mark("return-add")
        plan.add_returned_rows(dz, back)
        comm.mark("dense")
        dwn = ops._weight_grad(dz, h) if need[4] else None
        dh = (dy @ w_self).addmm_(dz, w_neigh) if need[2] else None
        return None, None, dh, dws, dwn, db


class HaloExchange(torch.autograd.Function):
    """x_own [n_own, ...] -> [n_own + n_halo, ...]; backward adds halo gradients into their owners."""

    @staticmethod
    def forward(ctx, x, plan, comm):
        x = x.contiguous()
        feat = tuple(x.shape[1:])
        full = torch.empty((plan.n_own + plan.n_halo,) + feat, dtype=x.dtype, device=x.device)
        full[:plan.n_own] = x
        send = sparse.gather_rows_raw(x, plan.send_idx)
        comm.all_to_all(full[plan.n_own:], send, plan.recv_splits, plan.send_splits)
        ctx.plan, ctx.comm = plan, comm
        return full

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, g):
        plan, comm = ctx.plan, ctx.comm
        g = g.contiguous()
        gx = g[:plan.n_own].clone()
        back = torch.empty((plan.send_idx.shape[0],) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        comm.all_to_all(back, g[plan.n_own:], plan.send_splits, plan.recv_splits)
        plan.add_returned_rows(gx, back)
        return gx, None, None


class DistGraph(DGLGraph):
    """The owned nodes of one partition behind the DGLGraph surface."""

    def __init__(self, block, plan, comm=None):
        self._block = block
        self._plan = plan
        self._comm = comm or _Comm()
        self._index = block._index
        self._is_block = False
        self._src_frame = Frame(plan.n_own, kind="node")
        self._dst_frame = self._src_frame
        self._edge_frame = Frame(block.number_of_edges(), kind="edge")
        self._batch_num_nodes = None
        self._batch_num_edges = None
        self._static_halo = {}  # set_static_input(): tensor / version / resident halo rows; shared by local_var() clones

    def _clone(self, index=None, frames=None):
        g = DistGraph.__new__(DistGraph)
        g._block, g._plan, g._comm = self._block, self._plan, self._comm
        g._static_halo = self._static_halo
        g._index, g._is_block = self._index, False
        if frames is None:
            g._src_frame = self._src_frame.clone()
            g._dst_frame = g._src_frame
            g._edge_frame = self._edge_frame.clone()
        else:
            g._src_frame, g._dst_frame, g._edge_frame = frames
        g._batch_num_nodes = g._batch_num_edges = None
        return g

    def set_static_input(self, x):
        """Declare `x` (this rank's rows of the constant input features) static: update_all(copy_u, sum|mean) on exactly
        this tensor object exchanges its halo rows once and keeps them resident; every rank must make the same call.
        `None` withdraws the declaration."""
        self._static_halo.clear()
        if x is not None:
            self._static_halo.update(tensor=x, version=None, recv=None)

    def number_of_nodes(self, ntype=None):
        return self._plan.n_own

    num_nodes = number_of_nodes

    def number_of_src_nodes(self, ntype=None):
        return self._plan.n_own

    def number_of_dst_nodes(self, ntype=None):
        return self._plan.n_own

    def int(self):
        return self

    def to(self, device, **kw):
        if torch.device(device) != self.device:
            raise DGLError("DistGraph lives on its rank's device")
        return self

    def formats(self, formats=None):
        return self._index.format_status() if formats is None else self

    def halo_exchange(self, x):
        return HaloExchange.apply(x, self._plan, self._comm)

    def out_degrees(self, u="__ALL__"):
        # the local block holds the owned nodes' IN-edges only; their out-edges into other parts live on other ranks
        raise DGLError("out_degrees() is not available on a partition (dist.DistGraph holds in-edges only); "
                       "take the degrees on the whole graph before partitioning")

    out_degree = out_degrees

    def _local(self, fields):
        src = Frame(self._plan.n_own + self._plan.n_halo, kind="node")
        for f in fields:
            if f not in self._src_frame:
                raise DGLError("Cannot find field %r in the source node features" % (f,))
            src[f] = self.halo_exchange(self._src_frame[f])
        return self._block._clone(frames=(src, self._dst_frame, self._edge_frame))

    @staticmethod
    def _u_fields(func):
        if isinstance(func, fn.CopyMessageFunction):
            return [func.in_field] if func.target == "u" else []
        if isinstance(func, fn.BinaryMessageFunction):
            out = []
            if func.lhs == "u":
                out.append(func.lhs_field)
            if func.rhs == "u":
                out.append(func.rhs_field)
            return out
        raise DGLError("DistGraph supports builtin message functions only")

    def update_all(self, message_func, reduce_func, apply_node_func=None, etype=None):
        if (isinstance(message_func, fn.CopyMessageFunction) and message_func.target == "u"
                and isinstance(reduce_func, fn.SimpleReduceFunction) and reduce_func.name in ("sum", "mean")
                and reduce_func.msg_field == message_func.out_field and self._plan.loc is not None
                and apply_node_func is None):
            x = self._src_frame[message_func.in_field]
            static = self._static_halo if self._static_halo.get("tensor") is x else None  # declared by set_static_input
            self._dst_frame[reduce_func.out_field] = DistCopyU.apply(x, self._plan, self._comm, reduce_func.name, static,
                                                                     static is None and sparse_halo_applies(x, self._plan))
            return
        blk = self._local(self._u_fields(message_func))
        core.update_all(blk, message_func, reduce_func, apply_node_func)

    def apply_edges(self, func, edges="__ALL__", etype=None):
        blk = self._local(self._u_fields(func))
        core.apply_edges(blk, func)

    def sage_project_first(self, h, w_self, w_neigh, bias):
        """SAGEConv with the projection before the aggregation on this partition (ops.sage_project_first dispatches here); None when
        the reference's order (aggregate, then project) moves fewer columns -- the rule of the one-GPU form."""
        K, D = w_self.shape
        plan = self._plan
        if (plan.loc is None or h.dim() != 2 or h.dtype != torch.float32 or h.shape[0] != plan.n_own or h.shape[1] != D
                or K % 4 or K > 128  # the one-GPU form's widths (ops.sage_project_first): N = 1 and N > 1 run the same module graph
                or not torch.is_grad_enabled() or not config.SAGE_PROJECT_FIRST
                or (bias is not None and K > getattr(sparse.backend_for(h), "COLUMN_SUM_MAX", 256))):
            return None
        aggs_now = 2 if h.requires_grad else 1
        if 2 * max(K, 16) * 1.1 > aggs_now * D:
            return None
        return DistSageProjectFirstFn.apply(plan, self._comm, h, w_self, w_neigh, bias)

    def _cat_layer_applies(self, h, cat):
        """The one-GEMM layer's conditions on this partition: a float32 [n_own, K] input that the CatBuffer holds, 32-bit offsets everywhere."""
        plan = self._plan
        return not (cat is None or plan.loc is None or h.dim() != 2 or h.dtype != torch.float32 or not h.is_cuda or cat.K != h.shape[1]
                    or h.shape[0] != plan.n_own or cat.buf.shape[0] != plan.n_own or h.shape[1] % 4
                    or plan.loc.csc().indptr.dtype != torch.int32 or not torch.is_grad_enabled()
                    or not config.SAGE_CAT or not config.SAGE_FUSED_LAYER
                    or plan.return_csr().num_cols * h.shape[1] * 4 >= (1 << 32) or (plan.n_own + plan.n_halo) * 2 * h.shape[1] * 4 >= (1 << 32))

    def sage_mean_layer(self, h, w_self, w_neigh, bias, cat, act=None):
        """The one-GEMM SAGE layer on this partition (ops.sage_mean_layer dispatches here); None when it does not apply."""
        if not self._cat_layer_applies(h, cat):
            return None
        plan = self._plan
        static = self._static_halo if self._static_halo.get("tensor") is h else None
        return DistSageMeanCatFn.apply(plan, self._comm, cat, h, w_self, w_neigh, bias, static, static is None and sparse_halo_applies(h, plan), act)

    def sage_mean_layer_act(self, h, w_self, w_neigh, bias, cat, p, out):
        """dropout(relu(sage_mean_layer(...)), p) as ONE node (ops.sage_mean_layer_act dispatches here): the layer's GEMM applies the
        activation in its epilogue and writes `out` -- the next layer's left half --, tagged as ops.relu_dropout tags its result, so the
        next layer's halo exchange takes the packed form.  None when that form does not apply (the caller composes the two)."""
        from . import ops
        if not self._cat_layer_applies(h, cat):
            return None
        be = sparse.backend_for(h)
        if (not hasattr(be, "rows_gemm_relu_dropout") or not config.SAGE_FUSED_ACT or not config.ROWS_GEMM
                or not (0.0 < p < 1.0) or ops.capture_path() or w_self.shape[0] % 4 or h.shape[0] < ops._ROWS_GEMM_MIN
                or not be.rows_gemm_supported(2 * cat.K, w_self.shape[0], cat.buf.stride(0))
                or (bias is not None and w_self.shape[0] > be.COLUMN_SUM_MAX)
                or (out is not None and (out.shape != (h.shape[0], w_self.shape[0]) or out.stride(1) != 1 or out.stride(0) % 4
                                         or out.data_ptr() % 16 or out.requires_grad))):
            return None
        y = self.sage_mean_layer(h, w_self, w_neigh, bias, cat, act=(float(p), None if out is None else ops._Into(out)))
        return ops._structural_zeros(y)


# ----------------------------------------------------------------------------- training helpers
def _staged(group):
    """gloo (tests / single-GPU smoke runs) moves device tensors through host memory."""
    return emulate.current() is None and dist.get_backend(group) == "gloo"


_EMU_OPS = {dist.ReduceOp.SUM: "sum", dist.ReduceOp.MAX: "max", dist.ReduceOp.MIN: "min"}


def world_size(group=None):
    """Ranks of the running job: the emulated world's, else the process group's, else 1."""
    emu = emulate.current()
    if emu is not None:
        return emu.size
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
    emu = emulate.current()
    if emu is not None:
        return emu.all_reduce(t, _EMU_OPS[op])
    if _staged(group) and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def broadcast(t, src=0, group=None):
    emu = emulate.current()
    if emu is not None:
        return emu.broadcast(t, src)
    if _staged(group) and t.is_cuda:
        h = t.cpu()
        dist.broadcast(h, src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src, group=group)
    return t


def allreduce_gradients(model, group=None):
    """One flat (bucketed) all_reduce(sum) over every parameter gradient -- the payload is tiny for
    these models (~30k floats for products SAGE), so a single call is latency-optimal on xGMI."""
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class GradBucket(object):
    """Every parameter gradient of `model` as a view into ONE flat buffer, set up once: autograd accumulates into the
    views in place, so the per-step gradient all_reduce is a single call on the buffer with no concatenate / copy-back.
    Use `bucket.zero()` where the training loop calls `optimizer.zero_grad()` (set_to_none would drop the views)."""

    def __init__(self, model, group=None):
        self.group = group
        self.params = [p for p in model.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self._attach()

    def _attach(self, keep=False):
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            if keep and p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)  # a gradient autograd allocated afresh after the views were dropped: keep its value
            elif keep and p.grad is None:
                view.zero_()
            p.grad = view
            off += n

    def attached(self):
        """True while every p.grad still aliases its slice of the flat buffer (optimizer.zero_grad(set_to_none=True) or
        model.zero_grad() drop the views; the reduce would then see a stale buffer and the ranks would diverge silently)."""
        off, base, esz = 0, self.flat.data_ptr(), self.flat.element_size()
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != base + off * esz or not p.grad.is_contiguous():
                return False
            off += p.numel()
        return True

    def zero(self):
        if not self.attached():
            self._attach()
        self.flat.zero_()

    def all_reduce(self):
        if not self.attached():  # re-attach, carrying over the gradients of this step
            self._attach(keep=True)
        all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)


def broadcast_parameters(model, src=0, group=None):
    for p in list(model.parameters()) + list(model.buffers()):
        broadcast(p.data, src, group=group)


# ----------------------------------------------------------------------------- BatchNorm over a partitioned node set
class _GlobalBatchNormFn(torch.autograd.Function):
    """Training-mode batch normalisation whose statistics span every rank's rows (the arxiv model's
    BatchNorm1d, main_dgl_arxiv_sage.py:70-77, must see the whole node set).  One all_reduce of
    [sum, sum of squares, count] forward, one of [sum dy, sum dy*xhat] backward -- 2C+1 floats each."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        C = x.shape[1]
        stats = torch.empty(2 * C + 1, dtype=torch.float64, device=x.device)
        ctx.lib_path = ops.batch_norm_supported(x)
        if ctx.lib_path:  # the library's column kernels for the local reductions and the maps (ops.BatchNormFn's scheme)
            x = x.contiguous()
            be = sparse.backend_for(x)
            # local sums relative to this rank's first row p (no fp32 cancellation), re-based to raw moments in fp64
            s, ss = be.column_pair_sums(x, shifted=True)
            pv, nl = x[0].double(), float(x.shape[0])
            s, ss = s.double(), ss.double()
            stats[:C], stats[C:2 * C], stats[2 * C] = s + nl * pv, ss + 2.0 * pv * s + nl * pv * pv, x.shape[0]
            all_reduce(stats, group=group)
            n = stats[2 * C]
            mean = stats[:C] / n
            var = (stats[C:2 * C] / n - mean * mean).clamp(min=0.0)
            invstd = torch.rsqrt(var + eps)
            A = (invstd if weight is None else weight.double() * invstd).float()
            Cc = (-mean * A.double() if bias is None else bias.double() - mean * A.double()).float()
            y = be.column_affine(x, A.contiguous(), Cc.contiguous())
            ctx.save_for_backward(x, weight, mean.float(), invstd.float())
            ctx.group, ctx.count = group, n
            ctx.mark_non_differentiable(mean, var, n)
            return y, mean, var, n
        xd = x.double()
        stats[:C] = xd.sum(0)
        stats[C:2 * C] = (xd * xd).sum(0)
        stats[2 * C] = x.shape[0]
        all_reduce(stats, group=group)
        n = stats[2 * C]
        mean = stats[:C] / n
        var = (stats[C:2 * C] / n - mean * mean).clamp(min=0.0)  # biased, as BatchNorm normalises with
        invstd = torch.rsqrt(var + eps)
        xhat = ((xd - mean) * invstd).to(x.dtype)
        ctx.save_for_backward(xhat, weight, invstd.to(x.dtype))
        ctx.group, ctx.count = group, n
        y = xhat * weight + bias if weight is not None else xhat
        ctx.mark_non_differentiable(mean, var, n)
        return y, mean, var, n

    @staticmethod
    @once_differentiable  # raw kernels inside: second-order gradients would silently be wrong
    def backward(ctx, dy, _dm, _dv, _dn):
        C = dy.shape[1]
        red = torch.empty(2 * C, dtype=torch.float64, device=dy.device)
        if ctx.lib_path:
            x, weight, mean, invstd = ctx.saved_tensors
            dy = dy.contiguous()
            be = sparse.backend_for(dy)
            sdy, sdyx = be.column_pair_sums(dy, x, shifted=True)      # sum dy * (x - x[0])
            sdyxhat = (sdyx - (mean - x[0]) * sdy) * invstd           # this rank's rows
            red[:C], red[C:] = sdy.double(), sdyxhat.double()
            all_reduce(red, group=ctx.group)
            n = ctx.count
            A = invstd if weight is None else weight * invstd
            B = (-A.double() * invstd.double() * red[C:] / n).float()
            Cc = (-A.double() * red[:C] / n - B.double() * mean.double()).float()
            dx = be.column_affine(dy, A.contiguous(), Cc.contiguous(), x, B.contiguous())
            # LOCAL contributions: the caller's gradient all_reduce (allreduce_gradients) sums them over ranks
            return dx, (sdyxhat if weight is not None else None), (sdy if weight is not None else None), None, None
        xhat, weight, invstd = ctx.saved_tensors
        dyd = dy.double()
        red[:C] = dyd.sum(0)
        red[C:] = (dyd * xhat.double()).sum(0)
        all_reduce(red, group=ctx.group)
        sum_dy, sum_dy_xhat = red[:C], red[C:]
        g = weight.double() * invstd.double() if weight is not None else invstd.double()
        dx = (g * (dyd - sum_dy / ctx.count - xhat.double() * (sum_dy_xhat / ctx.count))).to(dy.dtype)
        dw = db = None
        if weight is not None:
            # LOCAL contributions: the caller's gradient all_reduce (allreduce_gradients) sums them over ranks
            dw = (dyd * xhat.double()).sum(0).to(weight.dtype)
            db = dyd.sum(0).to(weight.dtype)
        return dx, dw, db, None, None


class GlobalBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d over rows that are partitioned across ranks: same parameters, buffers and state_dict;
    training-mode statistics (and the running estimates) are those of the union of all ranks' rows."""

    def __init__(self, *args, **kwargs):
        self.process_group = kwargs.pop("process_group", None)
        super(GlobalBatchNorm1d, self).__init__(*args, **kwargs)

    def forward(self, x):
        use_batch = self.training or not self.track_running_stats
        if not use_batch or world_size(self.process_group) == 1:
            return super(GlobalBatchNorm1d, self).forward(x)
        if x.dim() != 2:
            raise DGLError("GlobalBatchNorm1d expects (rows, channels) input")
        y, mean, var, n = _GlobalBatchNormFn.apply(x, self.weight, self.bias, self.eps, self.process_group)
        if self.training and self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                unbiased = var * (n / (n - 1).clamp(min=1.0))
                self.running_mean.mul_(1 - m).add_(mean.to(self.running_mean.dtype), alpha=m)
                self.running_var.mul_(1 - m).add_(unbiased.to(self.running_var.dtype), alpha=m)
        return y


def convert_batchnorm(module, process_group=None):
    """Replaces every nn.BatchNorm1d under `module` by a GlobalBatchNorm1d carrying the same state (the counterpart
    of nn.SyncBatchNorm.convert_sync_batchnorm for node sets that are partitioned, not replicated)."""
    out = module
    if isinstance(module, nn.BatchNorm1d) and not isinstance(module, GlobalBatchNorm1d):
        out = GlobalBatchNorm1d(module.num_features, module.eps, module.momentum, module.affine,
                                module.track_running_stats, process_group=process_group)
        out.load_state_dict(module.state_dict())
        out.to(next(iter(module.state_dict().values())).device if module.state_dict() else "cpu")
        out.train(module.training)
    for name, child in module.named_children():
        out.add_module(name, convert_batchnorm(child, process_group))
    return out
