"""Quality of dist.partition_nodes (LP clusters + greedy placement + refinement; the stand-in for METIS,
partition_utils.py:9-16 in the reference's cluster-sage sampler) BEYOND the benchmark generator's planted structure: edge cut,
in-edge balance and halo rows at P = 2 / 4 / 8 on the control graphs (R-MAT, no-community power law, clustered, products),
next to a uniformly random assignment of the same sizes.

  python experiments/exp_partition_quality.py [--scale 0.25] [--graphs rmat,mixing1,clustered,products]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
import kernel_controls as kc  # noqa: E402
from mi355x_graph import dist as mdist  # noqa: E402

dev = torch.device("cuda:0")


def describe(assign, src, dst, parts):
    a_s, a_d = assign[src], assign[dst]
    cut = float((a_s != a_d).float().mean())
    load = torch.bincount(a_d, minlength=parts).double()
    remote = a_s != a_d
    halo = torch.unique(a_d[remote] * assign.numel() + src[remote]).numel()  # distinct (receiving part, remote source) pairs
    return cut, float(load.max() / load.mean()), halo


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--scale", type=float, default=0.25)
    p.add_argument("--graphs", default="rmat,mixing1,clustered,products")
    args = p.parse_args()
    print("# dist.partition_nodes against a random assignment (same graph); halo = sum over parts of distinct remote sources")
    print("%-10s %2s %10s %10s %8s %8s %12s %12s %8s" % ("graph", "P", "cut", "cut_rand", "ratio", "balance", "halo_rows", "halo_rand", "seconds"))
    for kind in args.graphs.split(","):
        n, (src, dst) = kc.control_edges(kind, dev, args.scale)
        src, dst = src.long(), dst.long()
        for parts in (2, 4, 8):
            t0 = time.time()
            assign, stats = mdist.partition_nodes(src, dst, n, parts)
            torch.cuda.synchronize()
            dt = time.time() - t0
            cut, bal, halo = describe(assign, src, dst, parts)
            rnd = torch.randint(0, parts, (n,), device=dev)
            rcut, _, rhalo = describe(rnd, src, dst, parts)
            print("%-10s %2d %9.1f%% %9.1f%% %8.2f %8.3f %12d %12d %8.1f" % (kind, parts, 100 * cut, 100 * rcut, cut / rcut, bal, halo, rhalo, dt),
                  flush=True)
        del src, dst
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
