"""Batched small-graph classification: this backend's launcher for
end_to_end/full_graph/graph_classification/main_dgl_molhiv_gcn.py (BASELINE config 5).

The model restates the script's GCN-with-bond-encoder: per layer `in_degrees()`, a Python UDF message
`norm * relu(src.x + edge.w)` reduced with the builtin fn.sum (main_dgl_molhiv_gcn.py:37-52), AvgPooling
readout (:75,93), batches built by GraphDataLoader/dgl.batch and moved with
`.to(device).int().formats('coo')` (:101).  The UDF path here = two per-edge gathers (g-SDDMM copy
kernels), the UDF in PyTorch, then a copy_e/sum g-SpMM; the readout is mgx_segment_reduce.
The dataset is the seeded molhiv-shaped synthetic stand-in (mi355x_graph.datasets.molhiv_like).
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dgl  # noqa: E402
import dgl.function as fn  # noqa: E402
from dgl.dataloading import GraphDataLoader  # noqa: E402


class CategoricalEncoder(nn.Module):
    """Sum of one embedding table per integer feature column (the role of OGB's AtomEncoder/BondEncoder)."""

    def __init__(self, emb_dim, num_columns, cardinality=16):
        super(CategoricalEncoder, self).__init__()
        self.tables = nn.ModuleList([nn.Embedding(cardinality, emb_dim) for _ in range(num_columns)])
        for t in self.tables:
            nn.init.xavier_uniform_(t.weight.data)

    def forward(self, x):
        out = 0
        for i, t in enumerate(self.tables):
            out = out + t(x[:, i])
        return out


class GCNConv(nn.Module):
    def __init__(self, in_feats, out_feats, bond_columns=3):
        super(GCNConv, self).__init__()
        self.fc = nn.Linear(in_feats, out_feats, bias=False)
        self.root_emb = nn.Embedding(1, in_feats)
        self.bond_encoder = CategoricalEncoder(in_feats, bond_columns)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.fc.weight, gain=nn.init.calculate_gain("relu"))
        self.root_emb.reset_parameters()

    def forward(self, graph, feat, bond):
        graph = graph.local_var()
        x = self.fc(feat)
        deg = graph.in_degrees().float().unsqueeze(1) + 1
        graph.ndata["c"] = deg.pow(-0.5)
        graph.ndata["x"] = x
        graph.edata["w"] = self.bond_encoder(bond)
        graph.update_all(self.message, fn.sum("m", "h"))
        return graph.ndata["h"] + F.relu(x + self.root_emb.weight) * 1. / deg

    def message(self, edges):
        norm = edges.src["c"] * edges.dst["c"]
        return {"m": norm * F.relu(edges.src["x"] + edges.data["w"])}


class GCN(nn.Module):
    def __init__(self, emb_dim, num_classes, num_layers, dropout, atom_columns=9, bond_columns=3):
        super(GCN, self).__init__()
        self.atom_encoder = CategoricalEncoder(emb_dim, atom_columns)
        self.layers = nn.ModuleList([GCNConv(emb_dim, emb_dim, bond_columns) for _ in range(num_layers)])
        self.bns = nn.ModuleList([nn.BatchNorm1d(emb_dim) for _ in range(num_layers - 1)])
        self.dropout = nn.Dropout(p=dropout)
        self.readout = dgl.nn.AvgPooling()
        self.graph_pred_fc = nn.Linear(emb_dim, num_classes, bias=False)

    def forward(self, g, atom, bond):
        x = self.atom_encoder(atom)
        for i, layer in enumerate(self.layers[:-1]):
            x = self.dropout(F.relu(self.bns[i](layer(g, x, bond))))
        x = self.layers[-1](g, x, bond)
        return self.graph_pred_fc(self.readout(g, x))


class GIN(nn.Module):
    """5-layer GIN (BASELINE.json config 5 as worded): h <- MLP((1 + eps) h + sum_{u->v} h_u) with dgl.nn.GINConv, i.e. one
    copy_u/sum g-SpMM per layer on the batched graph; atom encoder, BatchNorm / relu / dropout and the mean readout as in
    the GCN above.  Bond features are not used (plain GIN)."""

    def __init__(self, emb_dim, num_classes, num_layers, dropout, atom_columns=9):
        super(GIN, self).__init__()
        from dgl.nn.pytorch import GINConv
        self.atom_encoder = CategoricalEncoder(emb_dim, atom_columns)
        mlp = lambda: nn.Sequential(nn.Linear(emb_dim, 2 * emb_dim), nn.BatchNorm1d(2 * emb_dim), nn.ReLU(),
                                    nn.Linear(2 * emb_dim, emb_dim))
        self.layers = nn.ModuleList([GINConv(mlp(), "sum", learn_eps=True) for _ in range(num_layers)])
        self.bns = nn.ModuleList([nn.BatchNorm1d(emb_dim) for _ in range(num_layers - 1)])
        self.dropout = nn.Dropout(p=dropout)
        self.readout = dgl.nn.AvgPooling()
        self.graph_pred_fc = nn.Linear(emb_dim, num_classes, bias=False)

    def forward(self, g, atom, bond=None):
        x = self.atom_encoder(atom)
        for i, layer in enumerate(self.layers[:-1]):
            x = self.dropout(F.relu(self.bns[i](layer(g, x))))
        x = self.layers[-1](g, x)
        return self.graph_pred_fc(self.readout(g, x))


def train_epoch(model, device, loader, optimizer, loss_fn):
    model.train()
    loss = None
    for batched_graph, labels in loader:
        batched_graph = batched_graph.to(device).int().formats("coo")  # main_dgl_molhiv_gcn.py:101
        labels = labels.to(device)
        optimizer.zero_grad()
        out = model(batched_graph, batched_graph.ndata["feat"], batched_graph.edata["feat"])
        loss = loss_fn(out.float().view(-1), labels.float().view(-1))
        loss.backward()
        optimizer.step()
    return loss.item()


def main():
    p = argparse.ArgumentParser("molhiv-shaped GCN on the MI355X message-passing backend")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--num_layers", type=int, default=5)
    p.add_argument("--emb_dim", type=int, default=256)
    p.add_argument("--batch_size", type=int, default=256)
    p.add_argument("--num_graphs", type=int, default=32901)
    p.add_argument("--epochs", type=int, default=4)
    p.add_argument("--num_workers", type=int, default=0)
    p.add_argument("--model", default="gcn", choices=["gcn", "gin"],
                   help="gcn: the reference's main_dgl_molhiv_gcn.py model; gin: BASELINE.json's wording of config 5")
    args = p.parse_args()
    from mi355x_graph.datasets import molhiv_like
    device = torch.device("cuda:%d" % args.device)
    data = molhiv_like(args.num_graphs)
    loader = GraphDataLoader(data, batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers)
    model = (GCN if args.model == "gcn" else GIN)(args.emb_dim, 1, args.num_layers, 0.5).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=0.001)
    loss_fn = nn.BCEWithLogitsLoss()
    dur = []
    for epoch in range(1, args.epochs + 1):
        t0 = time.time()
        loss = train_epoch(model, device, loader, opt, loss_fn)
        torch.cuda.synchronize()
        if epoch >= 2:
            dur.append(time.time() - t0)
        print("epoch %d loss %.4f time %.3f" % (epoch, loss, time.time() - t0))
    if dur:
        print("Training time/epoch {:.4f}".format(sum(dur) / len(dur)))


if __name__ == "__main__":
    main()
