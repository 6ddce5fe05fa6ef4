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
