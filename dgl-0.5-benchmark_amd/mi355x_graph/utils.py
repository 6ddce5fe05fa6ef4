"""dgl.utils helpers used by the scripts (expand_as_pair, main_dgl_product_sage.py:10,21)."""


def expand_as_pair(input_, g=None):
    if isinstance(input_, tuple):
        return input_
    if g is not None and getattr(g, "is_block", False):
        if isinstance(input_, dict):
            raise TypeError("heterograph inputs are not supported by this backend")
        return input_, input_[:g.number_of_dst_nodes()]
    return input_, input_


class GraphedStep(object):
    """One training step (forward + backward + optimizer) captured in a HIP graph and replayed.

    Full-graph epochs on small graphs are launch-bound (a cora-sized GraphSAGE epoch is ~60 kernels of a few microseconds
    each); every operator of this backend launches on the current stream with caller-allocated buffers and no host
    synchronisation once a graph's formats and execution plans exist, so the whole step can be stream-captured.
    `loss_fn()` must return the scalar loss tensor and be free of host synchronisations (index with integer index
    tensors, not boolean masks); the optimizer must be capture-safe (e.g. Adam(capturable=True)).

        step = GraphedStep(lambda: F.nll_loss(model(g, x)[train_idx], y[train_idx]), optimizer)
        for epoch in range(n): loss = step()          # loss: 0-dim tensor; .item() it only when it is needed
    """

    def __init__(self, loss_fn, optimizer, warmup=3):
        import torch
        self._torch = torch
        self.loss_fn, self.optimizer = loss_fn, optimizer
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        from . import ops
        # lazily built formats / plans / workspaces and autotuned GEMMs settle here; operators that take another form under
        # capture (relu_dropout) take it in these steps too, so the captured step meets nothing for the first time
        with torch.cuda.stream(side), ops.warming_up_for_capture():
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        # captured on the warm-up stream: autograd remembers the stream a parameter's gradient accumulator first ran on, and
        # capturing elsewhere turns every accumulation into a cross-stream fork / join inside the graph
        with torch.cuda.graph(self.graph, stream=side):
            self.loss = self.loss_fn()
            self.loss.backward()
            optimizer.step()

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn()
        loss.backward()
        self.optimizer.step()
        return loss

    def __call__(self):
        self.graph.replay()
        return self.loss


# ---------------------------------------------------------------------------------------------------------------------------
# torch.nn.Linear on tall device matrices (the dense half of an UNMODIFIED reference model, main_dgl_product_sage.py:23-24):
# PyTorch's backward forms the bias gradient with a generic reduction that takes 19 ms for a 2.45 M x 47 fp32 matrix on this
# stack (profiles/r03_plain_epoch_timeline.txt: one at::native::reduce_kernel = 42 % of the reference-module epoch) and the weight
# gradient with a library GEMM of 1 ms where the tall-skinny product kernel takes 0.4.  accelerate_linear() swaps the function
# torch.nn.Linear.forward calls for one whose BACKWARD uses this package's column-sum and X^T Y kernels; forward stays addmm.
#
# OPT-IN (round 5; it was the default of `import dgl` in round 4): replacing a function of another library process-wide is outside
# DGL's surface, so nothing here runs unless the user asks for it -- `mi355x_graph.utils.accelerate_linear()` from a launcher, or
# MGX_ACCELERATE_LINEAR=1 in the environment of an unmodified script (read by `import dgl`, which then says so on stderr once).
# Reversible (accelerate_linear(False)).  Anything the replacement does not recognise takes PyTorch's function: CPU, other dtypes,
# fewer than 65 536 rows, > 2 dims, autocast, torch.compile, a stream capture, an active torch.func transform (vmap / grad / jacrev),
# tensor subclasses in ANY operand (DTensor / FSDP parameters, parametrizations).  Double backward through it raises
# (once_differentiable), as for every autograd node of this package.
_ORIGINAL_LINEAR = None
LINEAR_MIN_ROWS = 1 << 16


def _tall_linear_fn():
    import torch
    from . import sparse

    class TallLinearFn(torch.autograd.Function):
        @staticmethod
        def forward(x, weight, bias):
            return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

        @staticmethod
        def setup_context(ctx, inputs, output):
            x, weight, bias = inputs
            # x is needed for the weight gradient only, the weight for the input gradient only
            ctx.save_for_backward(x if weight.requires_grad else None, weight if x.requires_grad else None)
            ctx.has_bias = bias is not None

        @staticmethod
        @torch.autograd.function.once_differentiable  # the kernels below are not differentiable: a double backward says so
        def backward(ctx, g):
            x, weight = ctx.saved_tensors
            g = g.contiguous()
            be = sparse.backend_for(g)
            dx = g @ weight if ctx.needs_input_grad[0] else None
            dw = db = None
            want_db = ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                out_f, in_f = g.shape[1], x.shape[1]
                if out_f <= be.XTY_MAX[0] and in_f <= be.XTY_MAX[1] and x.stride(1) == 1:
                    if want_db:  # the bias gradient from the weight-gradient kernel's own pass over g (mgx_xty_colsum), else two kernels
                        dw, db = be.xty(g, x, colsum=True)
                    else:
                        dw = be.xty(g, x)  # [out, in] = g^T x over the rows
                else:
                    dw = g.t() @ x
            if want_db and db is None:
                db = be.column_sum(g) if g.shape[1] <= be.COLUMN_SUM_MAX else g.sum(0)
            return dx, dw, db

    return TallLinearFn


def _functorch_active():
    """True inside torch.func.vmap / grad / jacrev / functionalize: those need an autograd.Function with a vmap rule."""
    import torch
    try:
        return torch._C._functorch.peek_interpreter_stack() is not None
    except AttributeError:
        return False


def linear_accelerated():
    """Whether torch.nn.functional.linear is currently this package's replacement."""
    return _ORIGINAL_LINEAR is not None


def accelerate_linear(enable=True):
    """Route torch.nn.functional.linear (hence every torch.nn.Linear) through TallLinearFn for tall fp32 device matrices.
    accelerate_linear(False) restores PyTorch's function.  Returns the previous state.  Never called implicitly: see above."""
    global _ORIGINAL_LINEAR
    import torch
    import torch.nn.functional as F
    was = _ORIGINAL_LINEAR is not None
    if enable and not was:
        original, fn = F.linear, _tall_linear_fn()
        plain = (torch.Tensor, torch.nn.Parameter)

        def linear(input, weight, bias=None):  # noqa: A002 -- torch's own parameter names
            if (type(input) is torch.Tensor and input.is_cuda and input.dim() == 2 and input.dtype == torch.float32
                    and type(weight) in plain and weight.dtype == torch.float32 and weight.dim() == 2
                    and input.shape[0] >= LINEAR_MIN_ROWS and torch.is_grad_enabled() and (input.requires_grad or weight.requires_grad)
                    and (bias is None or (type(bias) in plain and bias.dtype == torch.float32))
                    and not torch.cuda.is_current_stream_capturing() and not torch.is_autocast_enabled()
                    and not torch.compiler.is_compiling() and not _functorch_active()):
                return fn.apply(input, weight, bias)
            return original(input, weight, bias)

        _ORIGINAL_LINEAR = original
        F.linear = linear
    elif not enable and was:
        F.linear = _ORIGINAL_LINEAR
        _ORIGINAL_LINEAR = None
    return was
