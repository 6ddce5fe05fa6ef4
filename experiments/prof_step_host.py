"""Host-side profile of one products training step (cProfile over 20 steps): where the ~0.3 ms before the first big kernel and the
per-launch Python time go."""
import cProfile
import os
import pstats
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dgl-0.5-benchmark_amd"))
import tunable  # noqa: E402
tunable.setup()
import torch  # noqa: E402
import dgl  # noqa: E402
import full_graph  # noqa: E402
from mi355x_graph import ops  # noqa: E402
from mi355x_graph.datasets import SHAPES, synthetic_edges  # noqa: E402

dev = torch.device("cuda:0")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
spec = SHAPES["products"]
cfg = full_graph.SAGE_CONFIGS["products"]
n, m = int(spec["n"] * scale), int(spec["m"] * scale)
src, dst = synthetic_edges(n, m, spec["max_deg"], spec["seed"], dev, symmetric=True)
g = dgl.graph((src, dst), num_nodes=n).int().formats(["csr", "csc"]).to(dev)
del src, dst
gen = torch.Generator().manual_seed(1)
x = torch.rand(n, spec["feat"], generator=gen).to(dev)
y = torch.randint(0, spec["classes"], (n,), generator=gen).to(dev)
idx = torch.nonzero(torch.rand(n, generator=gen) < 0.08).flatten().to(dev)
torch.manual_seed(1234)
model = full_graph.GraphSAGE(spec["feat"], cfg["hidden"], spec["classes"], cfg["num_layers"], cfg["dropout"], cfg["batch_norm"],
                             cfg["neigh_bias"]).to(dev)
model.rows_are_distinct = True
opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"])


def step():
    model.train()
    opt.zero_grad()
    loss = ops.nll_sum(model(g, x, rows=idx), y[idx]) / idx.shape[0]
    loss.backward()
    opt.step()
    return loss.item()


for _ in range(5):
    step()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.print_callers("item")
