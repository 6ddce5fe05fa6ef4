"""Edges per XCD range of the shipped schedule (xcd_remap gives each XCD a contiguous 1/8 of the work ITEMS)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import torch, dgl
import kernel_controls as kc
dev = torch.device("cuda:0")
for kind in os.environ.get("GRAPHS", "products,mixing1,rmat").split(","):
    n, (src, dst) = kc.control_edges(kind, dev)
    g = dgl.graph((src, dst), num_nodes=n).int().formats(["csc"]).to(dev)
    csc = g._index.csc()
    plan = csc.plan()
    ln = (plan.item_end - plan.item_beg).long()
    items = ln.shape[0]
    rpb = 16
    nblocks = ((items + rpb - 1) // rpb + 7) // 8 * 8
    per = nblocks // 8 * rpb
    sums = [int(ln[i * per:(i + 1) * per].sum()) for i in range(8)]
    tot = sum(sums)
    print(json.dumps({"graph": kind, "items": items, "edges_per_xcd_range_pct_of_mean": [round(100.0 * s * 8 / tot, 1) for s in sums]}), flush=True)
    del g, csc, plan
