"""Upper bound on what ANY on-chip cache of hot source rows (LDS staging of the most-gathered neighbours) can buy the
copy_u/sum g-SpMM on the benchmark graph: delete the edges whose source is among the K most-gathered nodes -- as if those
gathers were served for free -- and time the same kernel on what is left.
K = 640 rows: what 160 KiB of LDS holds at D = 64;  K = 16,384: what one XCD's 4 MiB L2 holds;  K = 131,072: all eight L2s.
"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import torch
import dgl
import kernel_controls as kc

dev = torch.device("cuda:0")
D = int(os.environ.get("D", "64"))
n, (src, dst) = kc.control_edges("products", dev, float(os.environ.get("SCALE", "1")))
E = src.shape[0]
cnt = torch.bincount(src, minlength=n)
X = torch.rand(n, D, device=dev)
rows = []
for K in (0, 640, 16384, 131072):
    if K:
        hot = torch.zeros(n, dtype=torch.bool, device=dev)
        hot[torch.topk(cnt, K).indices] = True
        keep = ~hot[src]
        s, d = src[keep], dst[keep]
    else:
        s, d = src, dst
    g = dgl.graph((s, d), num_nodes=n).int().formats(["csc"]).to(dev)
    csc = g._index.csc()
    ms = kc.time_spmm(csc, X)
    kc._marker(dev)
    rows.append({"K_hot_sources_removed": K, "edges_left": int(s.shape[0]), "gathers_removed_pct": round(100.0 * (1 - s.shape[0] / E), 1),
                 "ms": round(ms, 3)})
    print(json.dumps(rows[-1]), flush=True)
    del g, csc
base = rows[0]["ms"]
for r in rows[1:]:
    print("K = %6d: %4.1f %% of the gathers gone -> %.3f ms (%.1f %% less time)" % (r["K_hot_sources_removed"], r["gathers_removed_pct"], r["ms"],
                                                                                 100.0 * (1 - r["ms"] / base)))
