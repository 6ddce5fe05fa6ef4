"""Why is bench.py's `epoch_ms_layer1_projected_first` leg 24 ms when the same model as the MAIN model runs 17.7 ms?  Reproduce the
legs' order at full size and time every step."""
import os
import sys
import time

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
spec = SHAPES["products"]
cfg = full_graph.SAGE_CONFIGS["products"]
n, m = spec["n"], spec["m"]
src, dst = synthetic_edges(n, m, spec["max_deg"], spec["seed"], dev, symmetric=True)
g = dgl.graph((src, dst), num_nodes=n).int().formats(["csr", "csc"]).to(dev)
del src, dst
gen = torch.Generator().manual_seed(1)
x = torch.rand(n, spec["feat"], generator=gen).to(dev)
y = torch.randint(0, spec["classes"], (n,), generator=gen).to(dev)
idx = torch.nonzero(torch.rand(n, generator=gen) < 0.08).flatten().to(dev)


def run(tag, steps=6):
    torch.manual_seed(1234)
    model = full_graph.GraphSAGE(spec["feat"], cfg["hidden"], spec["classes"], cfg["num_layers"], cfg["dropout"], cfg["batch_norm"],
                                 cfg["neigh_bias"]).to(dev)
    model.rows_are_distinct = True
    opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"])
    ts = []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = ops.nll_sum(model(g, x, rows=idx), y[idx]) / idx.shape[0]
        loss.backward()
        opt.step()
        loss.item()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%-40s %s" % (tag, " ".join("%.2f" % t for t in ts)), flush=True)
    del model, opt
    torch.cuda.empty_cache()


def run_plain(tag, steps=4):
    torch.manual_seed(1234)
    model = full_graph.GraphSAGE(spec["feat"], cfg["hidden"], spec["classes"], cfg["num_layers"], cfg["dropout"], cfg["batch_norm"],
                                 cfg["neigh_bias"], plain=True).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"])
    ts = []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = torch.nn.functional.nll_loss(model(g, x)[idx], y[idx], reduction="sum") / idx.shape[0]
        loss.backward()
        opt.step()
        loss.item()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%-40s %s" % (tag, " ".join("%.2f" % t for t in ts)), flush=True)
    del model, opt
    torch.cuda.empty_cache()


from mi355x_graph import utils as mutils  # noqa: E402
run("default")
os.environ["MGX_SAGE_L1_PROJECT_FIRST"] = "1"
run("layer 1 projected first (before plain)")
os.environ["MGX_SAGE_L1_PROJECT_FIRST"] = "0"
was = mutils.accelerate_linear(False)
run_plain("plain, PyTorch linear backward")
os.environ["MGX_SAGE_L1_PROJECT_FIRST"] = "1"
run("layer 1 projected first (after plain, linear off)")
os.environ["MGX_SAGE_L1_PROJECT_FIRST"] = "0"
mutils.accelerate_linear(True)
run_plain("plain, accelerated linear")
os.environ["MGX_SAGE_L1_PROJECT_FIRST"] = "1"
run("layer 1 projected first (after both plain)")
