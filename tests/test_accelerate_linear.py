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
    utils.accelerate_linear(False)  # (another test of this process may have left it on)
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


def test_the_drop_in_import_leaves_torch_alone_unless_asked():
    """Opt-in (round 5): `import dgl` replaces nothing by default; MGX_ACCELERATE_LINEAR=1 switches it on and says so on stderr."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import torch.nn.functional as F; o = F.linear; import dgl; print(F.linear is not o)"
            % os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
    default = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                             env={k: v for k, v in os.environ.items() if k != "MGX_ACCELERATE_LINEAR"})
    on = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MGX_ACCELERATE_LINEAR="1"))
    assert default.stdout.strip().endswith("False"), default.stderr[-2000:]
    assert "MGX_ACCELERATE_LINEAR" not in default.stderr
    assert on.stdout.strip().endswith("True"), on.stderr[-2000:]
    assert "MGX_ACCELERATE_LINEAR=1" in on.stderr and "accelerate_linear(False)" in on.stderr


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
    utils.accelerate_linear(False)          # PyTorch's own backward first
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


@pytest.mark.gpu
def test_what_it_does_not_recognise_takes_pytorchs_function():
    """an active torch.func transform, tensor-subclass parameters and inputs that need no weight gradient."""
    dev = torch.device("cuda:0")
    n = 70000
    x = torch.randn(n, 8, device=dev)
    lin = torch.nn.Linear(8, 4).to(dev)

    class Tagged(torch.Tensor):  # stands for DTensor / a parametrization output: anything that is not a plain Tensor / Parameter
        pass

    utils.accelerate_linear(True)
    try:
        # torch.func.grad: the replacement has no vmap / functorch rule, so it must step aside (PyTorch's linear is transformable)
        f = lambda w: torch.nn.functional.linear(x, w, lin.bias).sum()  # noqa: E731
        gw = torch.func.grad(f)(lin.weight.detach())
        assert torch.allclose(gw, x.sum(0).expand(4, 8), rtol=1e-4, atol=1e-2)
        per_sample = torch.func.vmap(lambda row: torch.nn.functional.linear(row, lin.weight, lin.bias))(x[:16])
        assert per_sample.shape == (16, 4)
        # a subclass weight: PyTorch's function
        wsub = lin.weight.detach().as_subclass(Tagged).requires_grad_(True)
        y = torch.nn.functional.linear(x, wsub, lin.bias)
        assert "TallLinearFn" not in type(y.grad_fn).__name__
        # only the input needs a gradient: taken, and x itself is not kept for the backward
        xin = x.clone().requires_grad_(True)
        y = torch.nn.functional.linear(xin, lin.weight.detach(), None)
        assert "TallLinearFn" in type(y.grad_fn).__name__
        assert y.grad_fn.saved_tensors[0] is None and y.grad_fn.saved_tensors[1] is not None
        y.sum().backward()
        assert torch.allclose(xin.grad, lin.weight.detach().sum(0).expand(n, 8), rtol=1e-5, atol=1e-5)
        # double backward says so instead of returning wrong numbers
        xin2 = x.clone().requires_grad_(True)
        (g,) = torch.autograd.grad(torch.nn.functional.linear(xin2, lin.weight, lin.bias).sum(), xin2, create_graph=True)
        with pytest.raises(RuntimeError):
            g.sum().backward()
    finally:
        utils.accelerate_linear(False)
