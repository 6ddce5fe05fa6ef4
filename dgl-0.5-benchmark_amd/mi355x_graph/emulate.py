"""P ranks of the partitioned program as P host threads of ONE process on ONE device.

Why: the multi-GPU path (dist.py) needs one GPU per rank over RCCL, and a 1-GPU box cannot give that.  What a
1-GPU box CAN give is every rank's own work -- the same local CSRs, schedules, kernels, autograd nodes and `return_csr` a
real run uses -- executed one rank at a time, with each collective replaced by in-process copies between the ranks'
buffers.  That yields (a) parity of a P-way partition against the 1-GPU result on the HIP path for P = 4, 8
(tests/test_emulated_ranks.py) and (b) per-rank device time between collectives plus the exact halo byte matrix, from
which bench.py --emulate-ranks builds the scaling model (profiles/r04_scale_model.txt).

How: `EmuWorld(P).run(fn)` starts P threads that all execute `fn(rank)` -- the SPMD program -- but only the holder of a
token runs.  A collective is two steps, as with RCCL: *post* (hand over the send buffer, keep running -- the async
all_to_all of DistSageMeanCatFn) and *complete* (`wait()`): the rank gives the token to the next rank until every rank has
posted, then copies its slices out of the peers' send buffers.  So a rank's kernels between two collectives are never
interleaved with another rank's, HIP events around those stretches measure that rank alone, and the order of launches on
the (single, shared) stream is the order a data dependency requires.  Autograd runs its backward nodes in the calling
thread (`set_multithreading_enabled(False)`), otherwise every rank's HIP nodes would share one engine thread and the first
blocking collective inside a backward node would deadlock.

dist.py routes its collectives here whenever the calling thread belongs to an EmuWorld (`current()`); nothing in the
product path changes for a real process group.

Two ways of timing a rank (scale_model.py reports both):
  traced stretches   HIP events around every stretch between two collectives, one rank at a time.  Every wait() hands the token to
                     the next rank, so a rank re-starts with an empty queue after each of its ~7 collectives: the host never runs
                     ahead across a collective and every launch gap of a host-bound stretch counts.  CONSERVATIVE.
  solo epochs        `record_epoch()` keeps what the rank received in one live epoch; `solo_epochs()` then runs the rank's epochs
                     back to back, ALONE (no token passing, no events), every collective completed by a device copy of the kept payload
                     on the compute stream -- the way an RCCL work handle's wait() orders the stream without blocking the host.  The
                     wall clock of those epochs is what ONE rank's process does per epoch when the exchange itself costs nothing:
                     host and device overlap as they do in a real process (the host still stops at the step's own reads: the
                     packed exchange's sizes, loss.item()).  The payloads are one epoch old, so the numbers a solo epoch computes
                     are not a training trajectory -- it is a timing device only.
"""
import threading
import time

import torch

_TLS = threading.local()


def current():
    """The EmuRank of the calling thread, or None outside an emulated world."""
    return getattr(_TLS, "rank_ctx", None)


class EmuError(RuntimeError):
    pass


class _Handle(object):
    """What all_to_all_async returns: wait() completes the collective for this rank."""

    def __init__(self, ctx, seq, finish):
        self.ctx, self.seq, self.finish = ctx, seq, finish

    def wait(self):
        if self.finish is not None:
            payloads = self.ctx._complete(self.seq)
            fin, self.finish = self.finish, None
            fin(payloads)
            self.ctx._release(self.seq)
        return True


class _SoloHandle(object):
    """A collective of a solo epoch: wait() puts the payload kept by record_epoch() into `out` (stream-ordered, the host goes on)."""

    def __init__(self, ctx, out):
        self.ctx, self.out = ctx, out

    def wait(self):
        if self.out is not None:
            self.ctx._solo_fill(self.out)
            self.out = None
        return True


class EmuRank(object):
    """One rank's view of the world: the collectives dist.py needs, and the trace of what ran between them."""

    def __init__(self, world, rank):
        self.world, self.rank = world, rank
        self.size = world.size
        self._seq = 0
        self.n_exchanges = 0
        self._recording = None  # record_epoch(): what this rank received, collective after collective
        self._solo = None       # solo_epochs(): [kept payloads, position]
        # trace: ("seg", label, start, end) stretches of this rank's own work and ("post" | "wait", kind, seq, info) points
        self.trace = None
        self._label = "other"
        self._open = None

    # ---- tracing
    def _now(self):
        dev = self.world.device
        if dev is not None and dev.type == "cuda":
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(dev))
            return ev
        return time.perf_counter()

    def start_trace(self):
        self.trace = []
        self._open = self._now()

    def stop_trace(self):
        self._close()
        t, self.trace, self._open = self.trace, None, None
        return t

    def _close(self):
        if self.trace is not None and self._open is not None:
            self.trace.append(("seg", self._label, self._open, self._now()))
            self._open = None

    def mark(self, label):
        """Names the stretch of work that starts here (dist.py: pack / owned-source / halo-source / dense / ...)."""
        if self.trace is not None:  # ONE event ends the running stretch and starts the next (the apparatus is on the timeline it measures)
            ev = self._now()
            if self._open is not None:
                self.trace.append(("seg", self._label, self._open, ev))
            self._open = ev
        self._label = label

    # ---- a rank's epochs alone (module docstring: "solo epochs")
    def record_epoch(self, step):
        """One LIVE epoch (`step()`, all ranks together as always) whose received payloads are kept: the argument of solo_epochs()."""
        self._recording = []
        try:
            step()
            return self._recording
        finally:
            self._recording = None

    def _keep(self, t):
        if self._recording is not None:
            self._recording.append(t.detach().clone())

    def _solo_fill(self, out):
        kept, pos = self._solo
        if pos >= len(kept):
            raise EmuError("solo epoch: rank %d issues more collectives than the recorded epoch's %d" % (self.rank, len(kept)))
        self._solo[1] = pos + 1
        src = kept[pos]
        if out.numel():
            # a message whose size depends on THIS epoch's dropout mask (the gradient values of a packed exchange) has no kept
            # counterpart of the same size: zeros of the right size -- the same bytes written
            if src.shape == out.shape and src.dtype == out.dtype:
                out.detach().copy_(src)
            else:
                out.detach().zero_()

    def solo_epochs(self, step, kept, epochs, warmup=1):
        """This rank's epochs back to back with nobody else running: ms per epoch by the wall clock (device synchronised at both
        ends).  Call between two barriers so that the other ranks are parked."""
        dev = self.world.device
        sync = (lambda: torch.cuda.synchronize(dev)) if dev is not None and dev.type == "cuda" else (lambda: None)
        trace, self.trace = self.trace, None
        self._solo = [kept, 0]
        try:
            for _ in range(warmup):
                self._solo[1] = 0
                step()
            sync()
            t0 = time.perf_counter()
            for _ in range(epochs):
                self._solo[1] = 0
                step()
                if self._solo[1] != len(kept):
                    raise EmuError("solo epoch: rank %d issued %d collectives, the recorded epoch %d" % (self.rank, self._solo[1], len(kept)))
            sync()
            return (time.perf_counter() - t0) * 1e3 / max(epochs, 1)
        finally:
            self._solo, self.trace = None, trace

    # ---- the two halves of every collective
    def _post(self, kind, payload, info=None):
        seq = self._seq
        self._seq += 1
        w = self.world
        with w.cv:
            w._check()
            slot = w.posted.setdefault(seq, {"kind": kind, "data": [None] * self.size, "count": 0, "taken": 0})
            if slot["kind"] != kind:
                w._fail(EmuError("rank %d posts %s as collective #%d, another rank posted %s" % (self.rank, kind, seq, slot["kind"])))
            slot["data"][self.rank] = payload
            slot["count"] += 1
        if self.trace is not None:
            ev = self._now()
            if self._open is not None:
                self.trace.append(("seg", self._label, self._open, ev))
            self.trace.append(("post", kind, seq, info))
            self._open = ev
        return seq

    def _complete(self, seq):
        w = self.world
        if self.trace is not None:
            self._close()
            self.trace.append(("wait", w.posted[seq]["kind"], seq, None))
        with w.cv:
            w.waiting[self.rank] = seq
            while w.posted[seq]["count"] < self.size:
                w._check()
                if all(w.done[r] or (r in w.waiting and w.posted[w.waiting[r]]["count"] < self.size) for r in range(self.size)):
                    # only the token holder runs, so nobody is left who could post: the ranks wait on different collectives
                    w._fail(EmuError("rank %d waits on collective #%d (%s) that %d rank(s) never post"
                                     % (self.rank, seq, w.posted[seq]["kind"], self.size - w.posted[seq]["count"])))
                w._pass_token(self.rank)
                while w.turn != self.rank and w.error is None:
                    w.cv.wait()
            w.waiting.pop(self.rank, None)
            w._check()
            data = w.posted[seq]["data"]
        return data

    def _release(self, seq):
        w = self.world
        with w.cv:
            slot = w.posted[seq]
            slot["taken"] += 1
            if slot["taken"] == self.size:
                del w.posted[seq]
        if self.trace is not None:
            self._open = self._now()

    # ---- collectives (the signatures dist._Comm / dist.all_reduce / dist.broadcast use)
    def all_to_all_async(self, out, inp, out_splits, in_splits, tag="rows"):
        """tag: what the message carries -- "rows" (dense boundary rows), "bitmaps" / "values" (the two messages of a packed forward
        exchange, dist.SparseHalo), "gradient values" -- for the byte accounting of the scaling model."""
        self.n_exchanges += 1
        rank = self.rank
        if self._solo is not None:
            return _SoloHandle(self, out)
        info = {"recv_rows": list(out_splits), "row_bytes": int(out[0].numel() * out.element_size()) if out.shape[0] else
                int(inp[0].numel() * inp.element_size()) if inp.shape[0] else 0, "tag": tag}
        seq = self._post("all_to_all", (inp, list(in_splits)), info)

        def finish(payloads):
            outs = out.split(list(out_splits), 0)
            for p in range(self.size):
                p_inp, p_splits = payloads[p]
                piece = p_inp.split(p_splits, 0)[rank]
                if tuple(piece.shape) != tuple(outs[p].shape):
                    raise EmuError("all_to_all #%d: rank %d expects %s from rank %d, which sends %s"
                                   % (seq, rank, tuple(outs[p].shape), p, tuple(piece.shape)))
                if piece.numel():
                    outs[p].copy_(piece)
            self._keep(out)
        return _Handle(self, seq, finish)

    def all_to_all(self, out, inp, out_splits, in_splits):
        self.all_to_all_async(out, inp, out_splits, in_splits).wait()

    def all_reduce(self, t, op="sum"):
        if self._solo is not None:
            self._solo_fill(t)
            return t
        seq = self._post("all_reduce", t.detach().clone(), {"bytes": int(t.numel() * t.element_size())})
        parts = self._complete(seq)
        acc = parts[0].clone()
        for p in parts[1:]:  # rank order on every rank: the same bits everywhere
            if op == "sum":
                acc.add_(p)
            elif op == "max":
                acc = torch.maximum(acc, p)
            elif op == "min":
                acc = torch.minimum(acc, p)
            else:
                raise EmuError("all_reduce: unsupported op %r" % (op,))
        t.detach().copy_(acc)
        self._keep(t)
        self._release(seq)
        return t

    def broadcast(self, t, src=0):
        if self._solo is not None:
            return t
        seq = self._post("broadcast", t.detach().clone() if self.rank == src else None)
        parts = self._complete(seq)
        if self.rank != src:
            t.detach().copy_(parts[src])
        self._release(seq)
        return t

    def barrier(self):
        if self._solo is not None:
            return
        seq = self._post("barrier", None)
        self._complete(seq)
        self._release(seq)


class EmuWorld(object):
    def __init__(self, size, device=None):
        self.size = int(size)
        self.device = None if device is None else torch.device(device)
        self.cv = threading.Condition()
        self.turn = 0
        self.posted = {}
        self.done = [False] * self.size
        self.waiting = {}
        self.error = None
        self.ranks = [EmuRank(self, r) for r in range(self.size)]

    # called with self.cv held
    def _check(self):
        if self.error is not None:
            raise EmuError("another emulated rank failed: %r" % (self.error,))

    def _fail(self, err):
        self.error = err
        self.cv.notify_all()
        raise err

    def _pass_token(self, rank):
        for k in range(1, self.size):
            cand = (rank + k) % self.size
            if not self.done[cand]:
                self.turn = cand
                self.cv.notify_all()
                return

    def run(self, fn, *args, **kwargs):
        """Runs fn(rank, *args) as every rank; returns the list of results in rank order.  Re-raises the first failure."""
        results = [None] * self.size
        errors = [None] * self.size

        def body(rank):
            ctx = self.ranks[rank]
            _TLS.rank_ctx = ctx
            try:
                with self.cv:
                    while self.turn != rank and self.error is None:
                        self.cv.wait()
                    self._check()
                if self.device is not None and self.device.type == "cuda":
                    torch.cuda.set_device(self.device)
                with torch.autograd.set_multithreading_enabled(False):
                    results[rank] = fn(rank, *args, **kwargs)
            except BaseException as err:  # noqa: B902 -- every failure must wake the other ranks
                errors[rank] = err
                with self.cv:
                    if self.error is None:
                        self.error = err
                    self.cv.notify_all()
            finally:
                _TLS.rank_ctx = None
                with self.cv:
                    self.done[rank] = True
                    if self.error is None and self.turn == rank:
                        self._pass_token(rank)
                    self.cv.notify_all()

        self.turn, self.error = 0, None
        self.done = [False] * self.size
        self.waiting = {}
        threads = [threading.Thread(target=body, args=(r,), name="emu-rank-%d" % r, daemon=True) for r in range(self.size)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        first = next((e for e in errors if e is not None and not (isinstance(e, EmuError) and "another emulated rank" in str(e))), None)
        first = first or next((e for e in errors if e is not None), None)
        if first is not None:
            raise first
        return results


# ----------------------------------------------------------------------------- traces -> mean epoch -> priced epoch
def _ms(a, b):
    return a.elapsed_time(b) if hasattr(a, "elapsed_time") else (b - a) * 1e3


def typical_epoch(traces):
    """One rank's traces of K identical epochs (start_trace() ... stop_trace() around each; device synchronised since) ->
    the typical epoch -- per stretch the MEDIAN over the K epochs, so that one host stall (an allocator refill, a lazily built
    table) in one epoch does not pass for device work -- as stages in program order, one per collective plus the tail:
        {"kind", "info", "pre": {label: ms}, "window": {label: ms}}     work before the post / between post and wait
        {"kind": None, "pre": {label: ms}}                               what follows the last collective
    Posts and waits must nest one at a time (post k, work, wait k) -- what dist.py issues."""
    epochs = []
    for tr in traces:
        stages, cur, phase, open_post = [], {}, "pre", None
        for it in tr:
            if it[0] == "seg":
                cur[it[1]] = cur.get(it[1], 0.0) + _ms(it[2], it[3])
            elif it[0] == "post":
                if open_post is not None:
                    raise EmuError("typical_epoch: collective #%d posted while #%d is still open" % (it[2], open_post[2]))
                open_post, pre, cur = it, cur, {}
            else:
                if open_post is None or open_post[2] != it[2]:
                    raise EmuError("typical_epoch: wait on #%d without its post" % (it[2],))
                stages.append({"kind": open_post[1], "info": open_post[3], "pre": pre, "window": cur})
                open_post, cur = None, {}
        stages.append({"kind": None, "pre": cur})
        epochs.append(stages)
    shape = [s["kind"] for s in epochs[0]]
    for e in epochs[1:]:
        if [s["kind"] for s in e] != shape:
            raise EmuError("typical_epoch: the epochs of one rank issue different collectives")
    def median(vals):
        v = sorted(vals)
        h = len(v) // 2
        return v[h] if len(v) % 2 else 0.5 * (v[h - 1] + v[h])

    out = []
    for i, st in enumerate(epochs[0]):
        m = {"kind": st["kind"], "info": st.get("info")}
        for part in ("pre", "window"):
            if part in st:
                labels = []
                for e in epochs:
                    labels += [l for l in e[i][part] if l not in labels]
                m[part] = {l: median([e[i][part].get(l, 0.0) for e in epochs]) for l in labels}
        out.append(m)
    return out


def price_epoch(ranks, link_gbps, latency_us=10.0, allreduce_us=40.0, overlap=True):
    """Lock-step replay of every rank's typical epoch with the collectives PRICED instead of emulated.

    all_to_all: the slice q -> r rides its own xGMI link (MI355X: every pair of the 8 GPUs is linked, full duplex): it starts
    once BOTH ranks have posted, takes latency + bytes / link rate, and a rank's collective is complete when everything it
    receives AND everything it sends has landed (RCCL's grouped send/recv is one kernel per rank).  `overlap=False` prices
    the exchange as if it were posted at the wait() (nothing hidden behind the window).  Any other collective: everyone
    meets, + allreduce_us.  Returns {"epoch_ms", "rank_ms", "exchanges": [[{posted, enter, done, exposed} per rank]]}."""
    P = len(ranks)
    clock = [0.0] * P
    bw = link_gbps * 1e9
    exchanges = []
    n_stage = len(ranks[0])
    for r in ranks:
        if [s["kind"] for s in r] != [s["kind"] for s in ranks[0]]:
            raise EmuError("price_epoch: ranks issue different collectives")
    for k in range(n_stage):
        kind = ranks[0][k]["kind"]
        posted, enter = [0.0] * P, [0.0] * P
        for r in range(P):
            clock[r] += sum(ranks[r][k]["pre"].values())
            posted[r] = clock[r]
            if kind is not None:
                clock[r] += sum(ranks[r][k]["window"].values())
            enter[r] = clock[r]
        if kind is None:
            break
        if kind == "all_to_all":
            if not overlap:
                posted = list(enter)
            done = list(posted)
            for r in range(P):
                info = ranks[r][k]["info"]
                for q in range(P):
                    nbytes = info["recv_rows"][q] * info["row_bytes"]
                    if q == r or nbytes == 0:
                        continue
                    land = max(posted[q], posted[r]) + latency_us * 1e-3 + nbytes / bw * 1e3
                    done[r] = max(done[r], land)
                    done[q] = max(done[q], land)
            rec = []
            for r in range(P):
                clock[r] = max(enter[r], done[r])
                rec.append({"posted": posted[r], "enter": enter[r], "done": done[r], "exposed": clock[r] - enter[r]})
            exchanges.append(rec)
        else:
            t = max(enter) + allreduce_us * 1e-3
            clock = [t] * P
    return {"epoch_ms": max(clock), "rank_ms": list(clock), "exchanges": exchanges}
