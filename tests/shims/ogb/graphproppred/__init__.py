import torch

from mi355x_graph.datasets import molhiv_like
from . import mol_encoder  # noqa: F401


class _PpaLike(object):
    """ogbg-ppa layout: no node attributes (feat = zeros, one embedding row), 7 float edge attributes."""

    def __init__(self, num_graphs, seed):
        import numpy as np
        from mi355x_graph.datasets import molecule_like_graph
        from mi355x_graph.graph import graph as make_graph
        rng = np.random.default_rng(seed)
        self.items = []
        for _ in range(num_graphs):
            n = int(rng.integers(20, 60))
            s, d = molecule_like_graph(n, rng)
            g = make_graph((torch.from_numpy(s), torch.from_numpy(d)), num_nodes=n)
            g.ndata["feat"] = torch.zeros(n, dtype=torch.int64)
            g.edata["feat"] = torch.from_numpy(rng.random((len(s), 7), dtype=np.float32))
            self.items.append((g, torch.tensor(int(rng.integers(0, 37)))))

    def __getitem__(self, i):
        return self.items[i]

    def __len__(self):
        return len(self.items)


class DglGraphPropPredDataset(object):
    def __init__(self, name, root="dataset"):
        if name == "ogbg-ppa":
            self._d = _PpaLike(96, seed=2)
            self.task_type = "multiclass classification"
        else:
            self._d = molhiv_like(num_graphs=96, seed=1)
            self.task_type = "binary classification"
        self.num_tasks = 1

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
