"""mi355x_graph.utils.accelerate_linear(): torch.nn.Linear on tall device matrices with this package's column-sum / X^T Y kernels in
the backward (the dense half of an unmodified reference model, main_dgl_product_sage.py:23-24)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
from mi355x_graph import utils  # noqa: E402


def test_switch_is_reversible_and_inert_on_cpu():
    import torch.nn.functional as F
    utils.accelerate_linear(False)  # (an earlier `import dgl` in this process switches it on)
    original = F.linear
    assert utils.accelerate_linear(True) is False
    try:
        assert F.linear is not original and utils.accelerate_linear(True) is True      # idempotent
        lin = torch.nn.Linear(5, 3)
        x = torch.rand(70000, 5, requires_grad=True)
        y = lin(x)                                                                         # CPU tensors: PyTorch's own function
        assert "TallLinearFn" not in type(y.grad_fn).__name__
        y.sum().backward()
        assert torch.allclose(lin.bias.grad, torch.full((3,), 70000.0))
    finally:
        assert utils.accelerate_linear(False) is True
    assert F.linear is original


def test_the_drop_in_import_switches_it_on_and_the_environment_opts_out():
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import torch.nn.functional as F; o = F.linear; import dgl; print(F.linear is not o)"
            % os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
    on = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={k: v for k, v in os.environ.items() if k != "MGX_ACCELERATE_LINEAR"})
    off = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MGX_ACCELERATE_LINEAR="0"))
    assert on.stdout.strip().endswith("True"), on.stderr[-2000:]
    assert off.stdout.strip().endswith("False"), off.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("in_f,out_f,bias", [(100, 64, True), (64, 47, True), (64, 64, False), (300, 40, True)])
def test_gradients_match_pytorch(in_f, out_f, bias):
    dev = torch.device("cuda:0")
    torch.manual_seed(in_f + out_f)
    n = 90000
    x0 = torch.randn(n, in_f, device=dev)
    w = torch.randn(n, out_f, device=dev)
    ref = torch.nn.Linear(in_f, out_f, bias=bias).to(dev)
    fast = torch.nn.Linear(in_f, out_f, bias=bias).to(dev)
    fast.load_state_dict(ref.state_dict())
    xr, xf = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
    utils.accelerate_linear(False)          # (an earlier `import dgl` in this process switches it on): PyTorch's own backward first
    (ref(xr) * w).sum().backward()
    utils.accelerate_linear(True)
    try:
        y = fast(xf)
        assert "TallLinearFn" in type(y.grad_fn).__name__
        (y * w).sum().backward()
        small = fast(xf[:100])                                                             # short inputs keep PyTorch's function
        assert "TallLinearFn" not in type(small.grad_fn).__name__
    finally:
        utils.accelerate_linear(False)
    assert torch.equal(y.detach(), ref(x0).detach())
    assert torch.allclose(xf.grad, xr.grad, rtol=1e-5, atol=1e-5)
    scale = float(ref.weight.grad.abs().max())
    assert float((fast.weight.grad - ref.weight.grad).abs().max()) <= 1e-4 * scale        # other fp32 summation order over 90 k rows
    if bias:
        assert float((fast.bias.grad - ref.bias.grad).abs().max()) <= 1e-4 * float(ref.bias.grad.abs().max())
