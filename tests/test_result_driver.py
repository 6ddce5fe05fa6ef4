"""CPU suite: the result driver's parsing rule (counterpart of the reference's generate_result.py:29-44)."""
import importlib.util
import math
import os

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dgl-0.5-benchmark_amd")


def load():
    spec = importlib.util.spec_from_file_location("generate_result", os.path.join(PKG, "generate_result.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_parse_results_takes_mean_of_last_ten_and_final_accuracies():
    gr = load()
    lines = ["noise"] + ["Training time/epoch %.3f" % (0.1 * i) for i in range(1, 16)]
    lines += ["  Final Train: 91.20 ± 0.10", "Final Test: 78.40 ± 0.30", "Final Test: 79.00 ± 0.20"]
    rec = gr.parse_results("\n".join(lines))
    assert abs(rec["epoch_time"] - sum(0.1 * i for i in range(6, 16)) / 10) < 1e-9   # the LAST ten values
    assert rec["final_train_acc"].strip() == "91.20 ± 0.10" and rec["final_test_acc"].strip() == "79.00 ± 0.20"
    short = gr.parse_results("Training time/epoch 0.5\nTraining time/epoch 0.7")
    assert abs(short["epoch_time"] - 0.6) < 1e-9 and short["final_test_acc"] == ""
    assert math.isnan(gr.parse_results("nothing here")["epoch_time"])
    md = gr.to_markdown({"a": rec, "b": short})
    assert md.splitlines()[0].startswith("| run | epoch_time") and len(md.splitlines()) == 4
    names = [r[0] for r in gr.RUNS]
    assert len(set(names)) == len(names) and all(os.path.exists(os.path.join(PKG, r[1])) for r in gr.RUNS)


def test_loss_helpers_match_torch():
    """ops.select_distinct_rows / ops.nll_sum (the default model's loss tail) == x[rows] / F.nll_loss(reduction='sum'),
    values and gradients; host tensors, no kernels involved."""
    import torch
    import torch.nn.functional as F
    from mi355x_graph import ops
    torch.manual_seed(0)
    x = torch.randn(50, 7, requires_grad=True)
    rows = torch.randperm(50)[:20]
    t = torch.randint(0, 7, (20,))
    l1 = F.nll_loss(x[rows].log_softmax(-1), t, reduction="sum")
    (g1,) = torch.autograd.grad(l1, x)
    l2 = ops.nll_sum(ops.select_distinct_rows(x, rows).log_softmax(-1), t)
    (g2,) = torch.autograd.grad(l2, x)
    assert abs(float(l1.detach() - l2.detach())) < 1e-4 and torch.allclose(g1, g2, atol=1e-6)
