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
# Process-wide and reversible; switched on by the drop-in `import dgl` (MGX_ACCELERATE_LINEAR=0 opts out), never by `import
# mi355x_graph`.  Anything it does not recognise (CPU, other dtypes, fewer than 65 536 rows, > 2 dims, autocast, torch.compile, a
# stream capture, tensor subclasses) takes the original function.
_ORIGINAL_LINEAR = None
LINEAR_MIN_ROWS = 1 << 16


def _tall_linear_fn():
    import torch
    from . import sparse

    class TallLinearFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, weight, bias):
            ctx.save_for_backward(x, weight)
            ctx.has_bias = bias is not None
            return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

        @staticmethod
        @torch.autograd.function.once_differentiable  # the kernels below are not differentiable: a double backward says so
        def backward(ctx, g):
            x, weight = ctx.saved_tensors
            g = g.contiguous()
            be = sparse.backend_for(g)
            dx = g @ weight if ctx.needs_input_grad[0] else None
            dw = None
            if ctx.needs_input_grad[1]:
                out_f, in_f = weight.shape
                if out_f <= be.XTY_MAX[0] and in_f <= be.XTY_MAX[1] and x.stride(1) == 1:
                    dw = be.xty(g, x)  # [out, in] = g^T x over the rows
                else:
                    dw = g.t() @ x
            db = None
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = be.column_sum(g) if g.shape[1] <= be.COLUMN_SUM_MAX else g.sum(0)
            return dx, dw, db

    return TallLinearFn


def accelerate_linear(enable=True):
    """Route torch.nn.functional.linear (hence every torch.nn.Linear) through TallLinearFn for tall fp32 device matrices.
    accelerate_linear(False) restores PyTorch's function.  Returns the previous state."""
    global _ORIGINAL_LINEAR
    import torch
    import torch.nn.functional as F
    was = _ORIGINAL_LINEAR is not None
    if enable and not was:
        original, fn = F.linear, _tall_linear_fn()

        def linear(input, weight, bias=None):  # noqa: A002 -- torch's own parameter names
            if (input.is_cuda and input.dim() == 2 and input.dtype == torch.float32 and weight.dtype == torch.float32
                    and input.shape[0] >= LINEAR_MIN_ROWS and torch.is_grad_enabled() and (input.requires_grad or weight.requires_grad)
                    and (bias is None or bias.dtype == torch.float32) and not torch.cuda.is_current_stream_capturing()
                    and not torch.is_autocast_enabled() and not torch.compiler.is_compiling() and type(input) is torch.Tensor):
                return fn.apply(input, weight, bias)
            return original(input, weight, bias)

        _ORIGINAL_LINEAR = original
        F.linear = linear
    elif not enable and was:
        F.linear = _ORIGINAL_LINEAR
        _ORIGINAL_LINEAR = None
    return was
