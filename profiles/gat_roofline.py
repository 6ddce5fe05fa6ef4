"""Per-kernel algorithmic bytes / time / fraction of the 8 TB/s HBM roofline for one fused GAT layer (csrc/gatfused.hip),
from a rocprofv3 --kernel-trace --stats run of full_graph.py --model gat (profiles/collect.sh).

  python profiles/gat_roofline.py <trace dir> N E H F  > profiles/<tag>_gat_roofline.txt

Algorithmic (compulsory) bytes, 4-byte indices and floats, D = H*F; every array counted once:
  gat_fused FWD  indptr + indices + feat + el + er (read)                  + out + nstat (write, 16 B per (node, head))
  gat_fused DST  indptr + indices + feat + el + nstat + d_out + out (read) + d_er + t (write)
  gat_fused SRC  indptr + indices + d_out + nstat + feat + el (read)       + d_feat + d_el (write)
The unfused chain these replace moved, besides the same node arrays, six E x H edge tensors (logits / attention /
their gradients, written and re-read): 6 * 2 * 4 * E * H bytes per layer and direction pair.
"""
import csv
import glob
import os
import sys


def main():
    d, N, E, H, F = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    D = H * F
    f = max(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    idx = 4 * (N + 1) + 4 * E
    nd, nh = 4 * N * D, 4 * N * H
    algo = {
        ", 0, ": ("forward: online softmax + aggregation", idx + nd + 2 * nh + nd + 4 * nh),
        ", 1, ": ("backward, in-CSR walk (d_er, t)", idx + 3 * nd + nh + 4 * nh + 2 * nh),
        ", 2, ": ("backward, out-CSR walk (d_feat, d_el)", idx + 2 * nd + 4 * nh + nh + nd + nh),
    }
    print("# fused GAT layer, N = %d, E = %d, H = %d, F = %d (D = %d); peak 8000 GB/s" % (N, E, H, F, D))
    print("%-44s %-38s %6s %9s %10s %9s %7s" % ("kernel", "role", "calls", "avg us", "algo MB", "GB/s", "frac"))
    G = 1
    while G * 4 < D:
        G <<= 1
    lph = G if (F % 4 or (F // 4) & (F // 4 - 1)) else F // 4
    want = "gat_fused_kernel<%d, %d," % (G, lph)
    for r in rows:
        name = r["Name"]
        if "mgx::gat_" not in name or want not in name:   # the layer with this (H, F): lane-group width G, lanes per head
            continue
        for key, (role, b) in algo.items():
            if (key in name) if key.startswith("gat_") else ("gat_fused_kernel" in name and key in name.split("gat_fused_kernel")[1][:16]):
                us = float(r["AverageNs"]) / 1e3
                gbs = b / us / 1e3
                print("%-44s %-38s %6s %9.1f %10.1f %9.1f %7.4f" % (name.split("(")[0].replace("void ", "")[:44], role, r["Calls"], us, b / 1e6, gbs, gbs / 8000.0))
                break
    print("# E x H edge tensors the unfused chain wrote and re-read per layer (fwd + bwd): %.1f MB" % (6 * 2 * 4 * E * H / 1e6))


if __name__ == "__main__":
    main()
