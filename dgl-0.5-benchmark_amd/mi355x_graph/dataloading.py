"""dgl.dataloading.GraphDataLoader (main_dgl_molhiv_gcn.py:12,163-165): a torch DataLoader whose
collate function block-diagonally batches (graph, label) samples with dgl.batch."""
import torch
from torch.utils.data import DataLoader

from .graph import DGLGraph
from .transform import batch as batch_graphs


def _collate(samples):
    first = samples[0]
    if isinstance(first, DGLGraph):
        return batch_graphs(samples)
    if isinstance(first, (tuple, list)):
        cols = list(zip(*samples))
        return [_collate(list(c)) for c in cols]
    if isinstance(first, torch.Tensor):
        return torch.stack(samples, 0)
    return torch.utils.data.dataloader.default_collate(samples)


class _Indices(object):
    """A dataset of its own indices: the sampler / batching machinery of torch's DataLoader runs unchanged, and the collate function
    below turns a batch of INDICES into the batched graph straight from the pooled storage."""

    def __init__(self, n):
        self.n = n

    def __getitem__(self, i):
        return int(i)

    def __len__(self):
        return self.n


def _pooled_source(dataset):
    """(pool, labels, index map | None) when `dataset` is a (subset of a) list-style dataset whose graphs live in one
    transform.GraphPool and whose labels are one tensor -- mi355x_graph.datasets.SmallGraphDataset -- else None."""
    base, index = dataset, None
    if hasattr(dataset, "dataset") and hasattr(dataset, "indices"):  # dgl.data.utils.Subset
        base, index = dataset.dataset, torch.as_tensor(dataset.indices, dtype=torch.int64)
    pool, labels, graphs = getattr(base, "pool", None), getattr(base, "labels", None), getattr(base, "graphs", None)
    if pool is None or not isinstance(labels, torch.Tensor) or graphs is None or labels.shape[0] != len(graphs):
        return None
    return pool, labels, index


class GraphDataLoader(DataLoader):
    """dgl.dataloading.GraphDataLoader.  For datasets whose graphs live in one GraphPool the loader iterates over INDICES and collates
    from the pooled arrays (no per-sample __getitem__, no per-graph Python work): same batches, same order for the same seed, as the
    general path -- `tests/test_graph_pool.py` compares them."""

    def __init__(self, dataset, collate_fn=None, **kwargs):
        src = _pooled_source(dataset) if collate_fn is None else None
        if src is None:
            super(GraphDataLoader, self).__init__(dataset, collate_fn=collate_fn or _collate, **kwargs)
            return
        pool, labels, index = src
        import numpy as np

        def collate(ids):
            ids = torch.as_tensor(ids, dtype=torch.int64)
            if index is not None:
                ids = index[ids]
            # (a graph whose fields were reassigned after pooling must not be served from the pool: transform.GraphPool.members_clean)
            return [pool.batch(ids.numpy().astype(np.int64)), labels[ids]]

        if not pool.members_clean():
            super(GraphDataLoader, self).__init__(dataset, collate_fn=_collate, **kwargs)
            return
        self.pooled_dataset = dataset
        super(GraphDataLoader, self).__init__(_Indices(len(dataset)), collate_fn=collate, **kwargs)
