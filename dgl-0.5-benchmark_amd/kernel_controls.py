"""The dominant kernel (copy_u/sum g-SpMM) on the benchmark graph AND on control inputs of the same size (SURVEY 8d
"Synthetic inputs": R-MAT / Chung-Lu degree laws plus a uniform-degree control, co-reported).

The products-shaped generator (mi355x_graph.datasets.synthetic_edges) plants communities; how well the gather of
neighbour rows hits in L2 -- and with it the roofline fraction -- depends on that structure.  The controls bracket it:

  products   the benchmark input: power-law endpoints, 75 % of the edges inside planted communities, ids permuted
  mixing1    same degree law, NO communities (mixing = 1.0): what the kernel does with no locality to find
  rmat       R-MAT a=.57 b=.19 c=.19 d=.05 (Graph500 parameters), ids permuted, both directions stored
  uniform    every destination has the same in-degree, sources uniformly random (no skew, no locality)
  banded     every source within +-2048 rows of its destination: every gather is an L2 hit (the kernel's ceiling)

All have N = 2,449,029 rows and E = 123,718,280 stored edges (scaled by --scale).  Timing: HIP events around
mgx_spmm_csr on the launch stream, first repetitions discarded (kernel/dgl-new.py:8,18-23).  Run it under
`rocprofv3 --pmc ...` (separate passes) for FETCH_SIZE / WRITE_SIZE / TCC_HIT / TCC_MISS per variant:
profiles/collect_controls.sh.
"""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import torch  # noqa: E402

N_PRODUCTS, M_PRODUCTS, MAXDEG_PRODUCTS, SEED_PRODUCTS = 2449029, 61859140, 17481, 4
CONTROLS = ("products", "mixing1", "rmat", "uniform", "banded", "clustered")


def spmm_algorithmic_bytes(n_dst, n_src, nnz, D):
    """indptr + indices + every source row once + every output row once."""
    return 4 * (n_dst + 1) + 4 * nnz + 4 * n_src * D + 4 * n_dst * D


def rmat_edges(n, m, seed, device, a=0.57, b=0.19, c=0.19):
    """m R-MAT edges over n nodes (ids drawn in [0, 2^ceil(log2 n)), pairs with an endpoint >= n redrawn), randomly
    relabelled, both directions stored -- duplicates kept, like the benchmark generator."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    bits = max(1, (n - 1).bit_length())
    src_parts, dst_parts, have = [], [], 0
    while have < m:
        k = min(1 << 24, int((m - have) * 1.25) + 1024)
        s = torch.zeros(k, dtype=torch.int64, device=device)
        d = torch.zeros(k, dtype=torch.int64, device=device)
        for _ in range(bits):
            r = torch.rand(k, generator=gen, device=device)
            down = r >= a + b            # quadrants c, d: row bit set
            right = ((r >= a) & (r < a + b)) | (r >= a + b + c)  # quadrants b, d: column bit set
            s = (s << 1) | down.long()
            d = (d << 1) | right.long()
        ok = (s < n) & (d < n)
        s, d = s[ok][:m - have], d[ok][:m - have]
        src_parts.append(s)
        dst_parts.append(d)
        have += int(s.shape[0])
    s, d = torch.cat(src_parts), torch.cat(dst_parts)
    perm = torch.randperm(n, generator=gen, device=device)
    s, d = perm[s], perm[d]
    return torch.cat([s, d]), torch.cat([d, s])


def uniform_edges(n, e, seed, device):
    """e directed edges: destination v gets floor/ceil(e/n) in-edges, sources uniformly random."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    dst = (torch.arange(e, device=device, dtype=torch.int64) * n) // e
    src = torch.randint(0, n, (e,), generator=gen, device=device)
    return src, dst


def banded_edges(n, e, seed, device, half_width=2048):
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    dst = (torch.arange(e, device=device, dtype=torch.int64) * n) // e
    off = torch.randint(-half_width, half_width + 1, (e,), generator=gen, device=device)
    return (dst + off).clamp_(0, n - 1), dst


def clustered_edges(n, m, max_deg, seed, device, family=60, family_share=0.75):
    """A control with real NEIGHBOURHOOD OVERLAP: products' node and edge counts and power-law endpoints, but
    `family_share` of the edges stay inside small FAMILIES of ~`family` nodes -- dense groups (a node with 50 edges has ~37 of them
    among its ~60 family members), so two neighbours of a node are likely neighbours of each other; the rest goes half inside the
    usual communities, half anywhere.  Average local clustering, measured on the full-size graph
    (experiments/exp_clustered_control.py, profiles/r03_clustered_control.txt): this control ~0.3 (co-purchase graphs: ~0.4), the
    benchmark generator 0.16 (its power-law hubs close triangles inside a community), a structure-free power law 0.04."""
    from mi355x_graph.datasets import synthetic_edges
    m1 = int(m * family_share)
    s1, d1 = synthetic_edges(n, m1, max_deg, seed, device, mixing=0.0, avg_comm=family, symmetric=True)
    s2, d2 = synthetic_edges(n, m - m1, max_deg, seed + 1, device, mixing=0.5, symmetric=True)
    return torch.cat([s1, s2]), torch.cat([d1, d2])


def control_edges(kind, device, scale=1.0):
    from mi355x_graph.datasets import synthetic_edges
    n = max(4096, int(N_PRODUCTS * scale))
    m = max(4096, int(M_PRODUCTS * scale))
    if kind == "products":
        return n, synthetic_edges(n, m, min(MAXDEG_PRODUCTS, n - 1), SEED_PRODUCTS, device, symmetric=True)
    if kind == "mixing1":
        return n, synthetic_edges(n, m, min(MAXDEG_PRODUCTS, n - 1), SEED_PRODUCTS, device, mixing=1.0, symmetric=True)
    if kind == "rmat":
        return n, rmat_edges(n, m, SEED_PRODUCTS, device)
    if kind == "uniform":
        return n, uniform_edges(n, 2 * m, SEED_PRODUCTS, device)
    if kind == "banded":
        return n, banded_edges(n, 2 * m, SEED_PRODUCTS, device)
    if kind == "clustered":
        return n, clustered_edges(n, m, min(MAXDEG_PRODUCTS, n - 1), SEED_PRODUCTS, device)
    raise ValueError("unknown control graph %r (have %s)" % (kind, ", ".join(CONTROLS)))


def time_spmm(csc, X, reps=6, discard=2):
    """Mean launch duration (ms) of copy_u/sum on `csc`, HIP events on the launch stream."""
    from mi355x_graph import sparse
    keep, sparse.PROFILE = sparse.PROFILE, []
    try:
        for _ in range(reps):
            sparse.gspmm_raw(csc, "copy_lhs", "sum", X, None)
        torch.cuda.synchronize()
        recs = sparse.PROFILE
    finally:
        sparse.PROFILE = keep
    d = [r["start"].elapsed_time(r["end"]) for r in recs][discard:]
    return sum(d) / len(d)


MARKER = "mgx::rows_kernel"  # one 1-row mgx_gather_rows launch closes every (graph, width) block in a counter trace


def _marker(device):
    from mi355x_graph import sparse
    sparse.gather_rows_raw(torch.zeros(1, 1, device=device), torch.zeros(1, dtype=torch.int32, device=device))


def parse_pmc_dir(path):
    """rocprofv3 --pmc output of THIS script -> per (graph, width) block, in launch order, the counter sums over the
    block's g-SpMM kernels (main kernel + hub fix-up) and the number of main-kernel launches.  Blocks are delimited by
    the marker launch.  Values are raw counter units (FETCH_SIZE / WRITE_SIZE in KiB)."""
    import csv
    import glob
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return []
    rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))
    disp = {}
    for r in rows:
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "c": {}})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    def fresh():
        return {"counters": {}, "launches": 0, "kernel": None, "ns": 0, "names": {}}

    def close(b):  # a call over a two-part plan is two main kernels: launches = calls of the most frequent one, kernel = all names
        if b["names"]:
            b["launches"] = max(b["names"].values())
            b["kernel"] = " + ".join(b["names"])
        del b["names"]
        return b

    blocks, cur = [], fresh()
    for k in sorted(disp):
        d = disp[k]
        if MARKER in d["name"]:
            blocks.append(close(cur))
            cur = fresh()
        elif "mgx::spmm" in d["name"]:
            for c, v in d["c"].items():
                cur["counters"][c] = cur["counters"].get(c, 0.0) + v
            cur["ns"] += d["ns"]
            if "fixup" not in d["name"]:
                name = d["name"].split("(")[0].replace("void ", "")
                cur["names"][name] = cur["names"].get(name, 0) + 1
    return blocks


def pmc_to_traffic(block_sets, launches_per_block=None):
    """[{counter: blocks}] from the separate passes -> per block {"hbm_read_bytes", "hbm_write_bytes", "l2_hit"} per
    launch, with the gfx950 corrections of guides/MI355X_MICROARCH.md (FETCH_SIZE in KiB tallies 128-B requests at 64 B:
    x 1024 x 2; WRITE_SIZE KiB exact: x 1024)."""
    n = max((len(b) for b in block_sets), default=0)
    out = [dict() for _ in range(n)]
    for blocks in block_sets:
        for i, b in enumerate(blocks):
            L = launches_per_block or max(b["launches"], 1)
            c = b["counters"]
            if b["kernel"]:
                out[i]["kernel"] = b["kernel"]
            if "FETCH_SIZE" in c:
                out[i]["hbm_read_bytes"] = int(c["FETCH_SIZE"] * 1024 * 2 / L)
            if "WRITE_SIZE" in c:
                out[i]["hbm_write_bytes"] = int(c["WRITE_SIZE"] * 1024 / L)
            if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0) > 0:
                out[i]["l2_hit"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0)), 4)
    return out


def run_controls(device, kinds=CONTROLS, widths=(64,), scale=1.0, reps=6, peak_gbps=8000.0):
    """{kind: {"D=..": {"ms", "achieved_GBps", "frac", ...}}}: same kernel, same sizes, different edge structure."""
    import dgl
    out = {}
    gen = torch.Generator(device=device).manual_seed(7)
    for kind in kinds:
        n, (src, dst) = control_edges(kind, device, scale)
        g = dgl.graph((src, dst), num_nodes=n).int().formats(["csc"]).to(device)
        del src, dst
        csc = g._index.csc()
        entry = {"rows": n, "nnz": csc.nnz}
        plan = csc.plan()
        entry["schedule"] = "none" if plan is None else plan.order_kind
        for D in widths:
            X = torch.rand(n, D, generator=gen, device=device)
            ms = time_spmm(csc, X, reps)
            algo = spmm_algorithmic_bytes(n, n, csc.nnz, D)
            entry["D=%d" % D] = {"ms": round(ms, 4), "achieved_GBps": round(algo / ms / 1e6, 1),
                                 "frac": round(algo / ms / 1e6 / peak_gbps, 4), "algorithmic_bytes": algo}
            _marker(device)
            del X
        out[kind] = entry
        del g, csc, plan
        torch.cuda.empty_cache()
    return out


def slot_block(device, kind, scale, reps, density=0.21):
    """One more block of a counter trace: the D = 64 aggregation over 128-byte slots (mgx_spmm_copy_u_slots) of a relu + dropout-like
    operand on the `kind` graph -- the form three of the benchmark epoch's four D = 64 launches take."""
    import dgl
    from mi355x_graph import sparse
    n, (src, dst) = control_edges(kind, device, scale)
    g = dgl.graph((src, dst), num_nodes=n).int().formats(["csc"]).to(device)
    del src, dst
    csc = g._index.csc()
    be = sparse.backend_for(csc.indptr)
    gen = torch.Generator(device=device).manual_seed(7)
    X = torch.rand(n, 64, generator=gen, device=device) * (torch.rand(n, 64, generator=gen, device=device) < density)
    out = torch.empty(n, 64, device=device)
    slots, _ = be.rows_slots_pack(X)
    torch.cuda.synchronize()
    _marker(device)                       # (closes the pack pass: its own block, dropped by the reader)
    for _ in range(reps):
        be.spmm_copy_u_strided(csc, "sum", X, out, slots=slots)
    torch.cuda.synchronize()
    _marker(device)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--graphs", default=",".join(CONTROLS))
    p.add_argument("--widths", default="64")
    p.add_argument("--scale", type=float, default=1.0)
    p.add_argument("--reps", type=int, default=6)
    p.add_argument("--slots", action="store_true", help="after the width blocks of the FIRST graph: the pack pass and the slot-form D = 64 launches as two more blocks")
    args = p.parse_args()
    kinds = [k for k in args.graphs.split(",") if k]
    res = run_controls(torch.device("cuda:0"), kinds, [int(w) for w in args.widths.split(",")], args.scale, args.reps)
    if args.slots:
        slot_block(torch.device("cuda:0"), kinds[0], args.scale, args.reps)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
