"""Tile plans behind the fused GAT walks on a dataset shape: parallel edges, largest rank, stream sizes.
  python experiments/exp_gat_plan_stats.py [reddit]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dgl-0.5-benchmark_amd"))
os.environ["MGX_GAT_TILE"] = "1"
import dgl  # noqa: E402
from mi355x_graph.datasets import NodeData  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "reddit"
dev = torch.device("cuda:0")
data = NodeData(name, device=dev)
g = dgl.add_self_loop(data.graph).int().to(dev)
for tag, csr in (("in-CSR", g._index.csc()), ("out-CSR", g._index.csr())):
    tp = csr.gat_tile_plan(16)
    print(tag, None if tp is None else {k: tp.stats[k] for k in ("tiles", "chunks", "edges", "staged_edges", "parallel_edges", "max_pair_rank",
                                                                 "pair_rank_streams", "lds_slot_fill", "dir_slot_fill")}, flush=True)
