import torch

from mi355x_graph.datasets import molhiv_like
from . import mol_encoder  # noqa: F401


class DglGraphPropPredDataset(object):
    def __init__(self, name, root="dataset"):
        self._d = molhiv_like(num_graphs=96, seed=1)
        self.num_tasks = 1
        self.task_type = "binary classification"

    def get_idx_split(self):
        idx = torch.arange(len(self._d))
        return {"train": idx[:64], "valid": idx[64:80], "test": idx[80:]}

    def __getitem__(self, i):
        if isinstance(i, torch.Tensor) and i.dim() > 0:
            return [self[int(j)] for j in i]
        g, y = self._d[int(i)]
        return g, y.view(1).float()

    def __len__(self):
        return len(self._d)


def collate_dgl(samples):
    import dgl
    graphs, labels = map(list, zip(*samples))
    return dgl.batch(graphs), torch.stack(labels)


class Evaluator(object):
    def __init__(self, name):
        self.eval_metric = "rocauc"

    def eval(self, d):
        return {"rocauc": 0.5}
