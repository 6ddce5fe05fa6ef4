"""Experiment: 2-D tiling of the copy_u/sum g-SpMM at the PLAN level (no kernel change).

Sources of a cluster are cut into blocks of B rows; a destination row's edges are grouped by (source block) and every
group becomes its own work item, scheduled by (cluster, block, row): the XCD walking that stretch of the schedule then
gathers from ONE source block (B * D * 4 bytes, sized to stay in its 4 MiB L2).  Rows with several items go through the
partial-slot / fix-up path that hub rows already use.  Inter-cluster edges ride with block 0.
  B=8192 SPLIT=256 python experiments/exp_tiled_plan.py
"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import torch
import dgl
from mi355x_graph import schedule, sparse
import kernel_controls as kc

dev = torch.device("cuda:0")
scale = float(os.environ.get("SCALE", "1"))
D = int(os.environ.get("D", "64"))
SPLIT = int(os.environ.get("SPLIT", "256"))
n, (src, dst) = kc.control_edges(os.environ.get("GRAPH", "products"), dev, scale)
g = dgl.graph((src, dst), num_nodes=n).int().formats(["csc"]).to(dev)
del src, dst
csc = g._index.csc()
X = torch.rand(n, D, device=dev)
base_ms = kc.time_spmm(csc, X)
kc._marker(dev)
ref, _, _ = sparse.gspmm_raw(csc, "copy_lhs", "sum", X, None)
print(json.dumps({"variant": "baseline", "ms": round(base_ms, 4), "items": csc.plan().num_items if csc.plan() else n}), flush=True)

hist = schedule.label_propagation(csc.indptr, csc.indices, n, 5)
order = torch.arange(n, device=dev)
for labels in hist[-3:]:
    order = order[torch.sort(labels[order], stable=True)[1]]
label = hist[-1]
pos = torch.empty(n, dtype=torch.int64, device=dev)
pos[order] = torch.arange(n, device=dev)
# rank of a node inside its cluster (schedule order)
lab_sorted = label[order]
first = torch.ones(n, dtype=torch.bool, device=dev)
first[1:] = lab_sorted[1:] != lab_sorted[:-1]
start_of = torch.cummax(torch.where(first, torch.arange(n, device=dev), torch.zeros(n, dtype=torch.int64, device=dev)), 0)[0]
rank = torch.empty(n, dtype=torch.int64, device=dev)
rank[order] = torch.arange(n, device=dev) - start_of
csize = torch.bincount(label, minlength=n)
print(json.dumps({"clusters": int((csize > 0).sum()), "max_cluster": int(csize.max()),
                  "nodes_in_clusters_over_8k": float((csize[label] > 8192).float().mean()),
                  "over_16k": float((csize[label] > 16384).float().mean()), "over_32k": float((csize[label] > 32768).float().mean())}), flush=True)

indptr = csc.indptr.long()
deg = indptr[1:] - indptr[:-1]
erow = torch.repeat_interleave(torch.arange(n, device=dev), deg)
ecol = csc.indices.long()

for B in [int(b) for b in os.environ.get("B", "8192,16384").split(",")]:
    t0 = time.perf_counter()
    same = label[ecol] == label[erow]
    key = torch.where(same, rank[ecol] // B, torch.zeros_like(ecol))
    nkey = int(key.max().item()) + 1
    # cluster-major, then source block, then row (schedule order)
    sort_key = (pos[erow] - rank[erow]) * (nkey * 1) + key          # cluster start position identifies the cluster
    sort_key = sort_key * n + pos[erow]
    perm = torch.sort(sort_key, stable=True)[1]
    e_row, e_key = erow[perm], key[perm]
    idx_perm = ecol[perm].to(torch.int32).contiguous()
    del sort_key
    E = e_row.shape[0]
    seg_first = torch.ones(E, dtype=torch.bool, device=dev)
    seg_first[1:] = (e_row[1:] != e_row[:-1]) | (e_key[1:] != e_key[:-1])
    seg_beg = torch.nonzero(seg_first).flatten()
    seg_end = torch.cat([seg_beg[1:], torch.tensor([E], device=dev)])
    seg_row = e_row[seg_beg]
    # chunks of <= SPLIT edges
    seg_len = seg_end - seg_beg
    nchunk = (seg_len + SPLIT - 1) // SPLIT
    item_seg = torch.repeat_interleave(torch.arange(seg_beg.shape[0], device=dev), nchunk)
    item_off = torch.cumsum(nchunk, 0) - nchunk
    chunk = torch.arange(item_seg.shape[0], device=dev) - item_off[item_seg]
    item_beg = seg_beg[item_seg] + chunk * SPLIT
    item_end = torch.minimum(item_beg + SPLIT, seg_end[item_seg])
    item_node = seg_row[item_seg]
    # rows with more than one item -> slots (contiguous per row), others write directly; rows without edges: 1 empty item
    items_per_row = torch.bincount(item_node, minlength=n)
    empty = torch.nonzero(items_per_row == 0).flatten()
    if empty.numel():
        item_node = torch.cat([item_node, empty])
        item_beg = torch.cat([item_beg, torch.zeros_like(empty)])
        item_end = torch.cat([item_end, torch.zeros_like(empty)])
        items_per_row = torch.bincount(item_node, minlength=n)
    multi = items_per_row > 1
    hub_rows = torch.nonzero(multi).flatten()
    hub_ptr = torch.zeros(hub_rows.shape[0] + 1, dtype=torch.int64, device=dev)
    torch.cumsum(items_per_row[hub_rows], 0, out=hub_ptr[1:])
    hub_index = torch.full((n,), -1, dtype=torch.int64, device=dev)
    hub_index[hub_rows] = torch.arange(hub_rows.shape[0], device=dev)
    # ordinal of an item among the items of its row, in item order
    o2 = torch.sort(item_node, stable=True)[1]
    node_sorted = item_node[o2]
    f2 = torch.ones_like(node_sorted, dtype=torch.bool)
    f2[1:] = node_sorted[1:] != node_sorted[:-1]
    st = torch.cummax(torch.where(f2, torch.arange(node_sorted.shape[0], device=dev), torch.zeros_like(node_sorted)), 0)[0]
    ordinal = torch.empty_like(item_node)
    ordinal[o2] = torch.arange(node_sorted.shape[0], device=dev) - st
    is_multi = multi[item_node]
    slot = hub_ptr[hub_index[item_node].clamp(min=0)] + ordinal
    item_row = torch.where(is_multi, -(slot + 1), item_node).to(torch.int32).contiguous()
    num_slots = int(hub_ptr[-1].item())
    slot_item = torch.empty(max(num_slots, 1), dtype=torch.int32, device=dev)[:num_slots]
    if num_slots:
        slot_item[slot[is_multi]] = torch.nonzero(is_multi).flatten().to(torch.int32)
    plan = schedule.SpmmPlan(item_row, item_beg.to(torch.int32).contiguous(), item_end.to(torch.int32).contiguous(),
                             hub_rows.to(torch.int32).contiguous(), hub_ptr.to(torch.int32).contiguous(), num_slots, "tiled",
                             slot_item.contiguous(), item_node.to(torch.int32).contiguous())
    csr2 = sparse.CsrView(n, n, csc.indptr, idx_perm, None)
    csr2._plan = plan
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    out, _, _ = sparse.gspmm_raw(csr2, "copy_lhs", "sum", X, None)
    err = float(((out - ref).abs().max() / ref.abs().max()).item())
    kc._marker(dev)   # closes the block that holds the reference / check launches
    ms = kc.time_spmm(csr2, X)
    kc._marker(dev)
    print(json.dumps({"variant": "tiled", "B": B, "ms": round(ms, 4), "items": int(item_row.shape[0]), "slots": num_slots,
                      "hub_rows": int(hub_rows.shape[0]), "rel_err": err, "build_s": round(build_s, 2)}), flush=True)
    del plan, csr2, out
