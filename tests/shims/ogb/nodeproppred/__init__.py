import torch

from mi355x_graph.datasets import SHAPES, synthetic_edges
import dgl

_TOY = {"ogbn-products": ("products", 3000, 40000), "ogbn-arxiv": ("arxiv", 2000, 9000),
        "ogbn-proteins": ("proteins", 1500, 20000)}


class DglNodePropPredDataset(object):
    def __init__(self, name, root="dataset"):
        from mi355x_graph import diskio
        real = diskio.find_dataset(name)
        if real is not None:  # real OGB files under $MGX_DATA_ROOT
            self.graph, self.labels = real.graph, real.labels.view(-1, 1)
            self.num_classes, self._split = real.num_classes, real.split_idx
            return
        key, n, m = _TOY[name]
        spec = SHAPES[key]
        src, dst = synthetic_edges(n, m, 200, spec["seed"], symmetric=spec["symmetric"])
        g = dgl.graph((src, dst), num_nodes=n)
        gen = torch.Generator().manual_seed(0)
        if key == "proteins":
            g.edata["feat"] = torch.rand(g.number_of_edges(), 8, generator=gen)
            g.ndata["species"] = torch.zeros(n, 1, dtype=torch.int64)
            self.labels = torch.randint(0, 2, (n, 112), generator=gen)
            self.num_tasks = 112
        else:
            g.ndata["feat"] = torch.rand(n, spec["feat"], generator=gen)
            self.labels = torch.randint(0, spec["classes"], (n, 1), generator=gen)
        self.num_classes = spec["classes"]
        self.graph = g
        r = torch.rand(n, generator=gen)
        self._split = {"train": torch.nonzero(r < 0.5).flatten(), "valid": torch.nonzero((r >= 0.5) & (r < 0.7)).flatten(),
                       "test": torch.nonzero(r >= 0.7).flatten()}

    def get_idx_split(self):
        return self._split

    def __getitem__(self, i):
        return self.graph, self.labels

    def __len__(self):
        return 1


class Evaluator(object):
    def __init__(self, name):
        self.name = name

    def eval(self, d):
        y_true, y_pred = d["y_true"], d["y_pred"]
        if self.name == "ogbn-proteins":
            return {"rocauc": 0.5}
        return {"acc": float((y_true == y_pred).float().mean())}
