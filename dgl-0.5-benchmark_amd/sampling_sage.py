"""Neighbor-sampling GraphSAGE: this backend's launcher for
end_to_end/sampling/node-classification/reddit/ns-sage-dgl.py.

Same model (dglnn.SAGEConv 'mean' stack, ns-sage-dgl.py:21-46), fan-outs 10,25, batch 1000, Adam 3e-3
(:203-213) and loop (:159-169); the graph and features stay on the GPU and MultiLayerNeighborSampler samples
there, so a step is sample -> index features -> 2 block g-SpMMs forward / backward -> Adam.  Epochs 0-4 are
discarded as in the script (:176-177).  Dataset: the seeded reddit-shaped synthetic stand-in.
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
import dgl.nn.pytorch as dglnn  # noqa: E402


class SAGE(nn.Module):
    def __init__(self, in_feats, n_hidden, n_classes, n_layers, activation, dropout):
        super().__init__()
        self.layers = nn.ModuleList()
        self.layers.append(dglnn.SAGEConv(in_feats, n_hidden, "mean"))
        for _ in range(1, n_layers - 1):
            self.layers.append(dglnn.SAGEConv(n_hidden, n_hidden, "mean"))
        self.layers.append(dglnn.SAGEConv(n_hidden, n_classes, "mean"))
        self.dropout = nn.Dropout(dropout)
        self.activation = activation

    def forward(self, blocks, x):
        h = x
        for l, (layer, block) in enumerate(zip(self.layers, blocks)):
            h = layer(block, h)
            if l != len(self.layers) - 1:
                h = self.dropout(self.activation(h))
        return h


def main():
    p = argparse.ArgumentParser("neighbor-sampling SAGE on the MI355X message-passing backend")
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--dataset", default="reddit")
    p.add_argument("--num-epochs", type=int, default=8)
    p.add_argument("--num-hidden", type=int, default=16)
    p.add_argument("--num-layers", type=int, default=2)
    p.add_argument("--fan-out", default="10,25")
    p.add_argument("--batch-size", type=int, default=1000)
    p.add_argument("--lr", type=float, default=0.003)
    p.add_argument("--dropout", type=float, default=0.5)
    p.add_argument("--train-frac", type=float, default=0.66, help="reddit trains on 153,431 of 232,965 nodes")
    args = p.parse_args()
    from mi355x_graph.datasets import NodeData
    device = torch.device("cuda:%d" % args.gpu)
    data = NodeData(args.dataset, device=device)
    g = dgl.add_self_loop(data.graph).int()
    g.create_formats_()
    feats, labels = data.features, data.labels
    gen = torch.Generator().manual_seed(0)
    train_nid = torch.nonzero(torch.rand(g.number_of_nodes(), generator=gen) < args.train_frac).flatten()
    sampler = dgl.dataloading.MultiLayerNeighborSampler([int(f) for f in args.fan_out.split(",")])
    loader = dgl.dataloading.NodeDataLoader(g, train_nid, sampler, batch_size=args.batch_size, shuffle=True, drop_last=False)
    model = SAGE(feats.shape[1], args.num_hidden, data.num_classes, args.num_layers, F.relu, args.dropout).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    loss_fcn = nn.CrossEntropyLoss()
    times = []
    for epoch in range(args.num_epochs):
        tic = time.time()
        seen = 0
        for input_nodes, seeds, blocks in loader:
            batch_pred = model(blocks, feats[input_nodes])
            loss = loss_fcn(batch_pred, labels[seeds])
            opt.zero_grad()
            loss.backward()
            opt.step()
            seen += seeds.numel()
        torch.cuda.synchronize()
        toc = time.time()
        print("Epoch {:03d} | Loss {:.4f} | Epoch Time(s): {:.4f} | {:.0f} seeds/s".format(epoch, loss.item(), toc - tic, seen / (toc - tic)))
        if epoch >= 5:
            times.append(toc - tic)
    if times:
        print("Avg epoch time: {:.4f}".format(sum(times) / len(times)))
        print("Training time/epoch {:.5f}".format(sum(times) / len(times)))  # the line generate_result.py parses


if __name__ == "__main__":
    main()
