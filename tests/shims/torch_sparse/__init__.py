"""TEST-ONLY stand-in so that the reference's kernel/utils.py imports (it needs torch_sparse only for its PyG path)."""


class SparseTensor(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("torch_sparse is not available here; only the DGL path of the harness is exercised")
