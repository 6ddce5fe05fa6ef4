import torch.nn as nn


class _Enc(nn.Module):
    def __init__(self, emb_dim, cols):
        super().__init__()
        self.tables = nn.ModuleList([nn.Embedding(16, emb_dim) for _ in range(cols)])

    def forward(self, x):
        out = 0
        for i, t in enumerate(self.tables):
            out = out + t(x[:, i])
        return out


class AtomEncoder(_Enc):
    def __init__(self, emb_dim):
        super().__init__(emb_dim, 9)


class BondEncoder(_Enc):
    def __init__(self, emb_dim):
        super().__init__(emb_dim, 3)
