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


class GraphDataLoader(DataLoader):
    def __init__(self, dataset, collate_fn=None, **kwargs):
        super(GraphDataLoader, self).__init__(dataset, collate_fn=collate_fn or _collate, **kwargs)
