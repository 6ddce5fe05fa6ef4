"""Full-graph node-classification workloads: this backend's launcher for the model families of
end_to_end/full_graph/node_classification (the callers of the hot path).

The model classes restate the reference scripts' modules with the same parameters, layer order and
operator calls so that every aggregation goes through update_all()/GATConv exactly as there:
  SAGEConv / GraphSAGE   main_dgl_product_sage.py:15-99 (no BatchNorm), main_dgl_arxiv_sage.py:15-105 (BatchNorm)
  GAT                    main_dgl_reddit_gat.py:14-65 (dgl.nn.pytorch.GATConv stack)
  train step             main_dgl_product_sage.py:101-110  (zero_grad, forward, nll_loss on train_idx,
                         backward, Adam step, loss.item() as the host sync)
Epoch timing follows main_dgl_product_sage.py:175-180 (epochs 1-2 discarded) but reports the
steady-state mean, not the reference's cumulative running mean.
Datasets are the seeded synthetic stand-ins of mi355x_graph.datasets (no network here).
"""
import weakref
import argparse
import os
import sys
import time

# GEMM selections recorded once on MI355X for the dense layers of the products model (experiments/tune_dense.py) are
# loaded, never tuned, here -- hipBLASLt's default heuristics pick 3 ms kernels for the K = 2.45 M weight-gradient
# shapes.  Other shapes fall through to the default.  MGX_BENCH_TUNABLEOP=0 turns it off (same switch as bench.py).
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tunable  # noqa: E402

tunable.setup()

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dgl  # noqa: E402
import dgl.function as fn  # noqa: E402
from dgl.nn.pytorch import GATConv, Linear, BatchNorm1d  # noqa: E402
from mi355x_graph import config, ops  # noqa: E402
from dgl.utils import expand_as_pair  # noqa: E402


class SAGEConv(nn.Module):
    """mean-aggregator GraphSAGE layer: fc_self(h_v) + fc_neigh(mean_{u->v} h_u); aggregation happens
    BEFORE the projection, on the un-projected feature (main_dgl_product_sage.py:61-64)."""

    def __init__(self, in_feats, out_feats, self_bias=False, neigh_bias=True, plain=False):
        super(SAGEConv, self).__init__()
        self._in_src_feats, self._in_dst_feats = expand_as_pair(in_feats)
        # plain: the reference's exact module graph (torch.nn.Linear, separate add); otherwise the same parameters with
        # the dense-side helpers of this package (mgx_xty / mgx_column_sum gradients, accumulate-GEMM)
        self.plain = plain
        lin = nn.Linear if plain else Linear
        self.fc_self = lin(self._in_dst_feats, out_feats, bias=self_bias)
        self.fc_neigh = lin(self._in_src_feats, out_feats, bias=neigh_bias)
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

    def forward(self, graph, feat, cat=None):
        """`cat`: an ops.CatBuffer whose left half is (or will receive) `feat`; the layer then runs as ONE GEMM on [feat | neigh]."""
        if not self.plain and not isinstance(feat, tuple) and self.fc_self.bias is None:
            # the whole layer as one autograd node (ops.SageMeanLayerFn / SageMeanCatFn): same aggregation kernel, the two
            # gradients of `feat` meet inside the reversed aggregation instead of in a separate add pass
            # in_feats >> out_feats (reddit: 602 -> 16): project first, aggregate at the output width (dgl.nn.SAGEConv's lin_before_mp)
            y = ops.sage_static_input_project(graph, feat, self.fc_self.weight, self.fc_neigh.weight, self.fc_neigh.bias, cat)
            if y is None:
                y = ops.sage_project_first(graph, feat, self.fc_self.weight, self.fc_neigh.weight, self.fc_neigh.bias)
            if y is None:
                y = ops.sage_mean_layer(graph, feat, self.fc_self.weight, self.fc_neigh.weight, self.fc_neigh.bias, cat=cat)
            if y is not None:
                return y
        graph = graph.local_var()
        feat_src, feat_dst = feat if isinstance(feat, tuple) else (feat, feat)
        graph.srcdata["h"] = feat_src
        graph.update_all(fn.copy_src("h", "m"), fn.mean("m", "neigh"))
        if not self.plain and self.fc_self.bias is None and config.SAGE_FUSED_ADD:
            # fc_self(h) + fc_neigh(neigh): the second GEMM accumulates into the first's output (no separate add pass)
            return ops.linear_sum(feat_dst, self.fc_self.weight, graph.dstdata["neigh"], self.fc_neigh.weight, self.fc_neigh.bias)
        return self.fc_self(feat_dst) + self.fc_neigh(graph.dstdata["neigh"])


class GraphSAGE(nn.Module):
    """`plain=True` is the reference's module graph verbatim in structure -- torch.nn.Linear, F.relu, nn.Dropout,
    nn.BatchNorm1d, `fc_self(h) + fc_neigh(neigh)` as two GEMMs and an add (main_dgl_product_sage.py:15-99): only the
    aggregation inside update_all() is this package's.  bench.py reports its epoch beside the default's."""

    def __init__(self, in_feats, hidden_feats, out_feats, num_layers, dropout, batch_norm=False, neigh_bias=True, plain=False):
        super(GraphSAGE, self).__init__()
        self.plain = plain
        self.rows_are_distinct = False  # set by a caller whose `rows` is an index without duplicates (nonzero of a mask)
        self._input_cat = None          # (key, ops.CatBuffer) of the first layer: [input features | their aggregation]
        self.layers = nn.ModuleList()
        self.bns = nn.ModuleList()
        dims = [in_feats] + [hidden_feats] * (num_layers - 1) + [out_feats]
        for i in range(num_layers):
            self.layers.append(SAGEConv(dims[i], dims[i + 1], neigh_bias=neigh_bias, plain=plain))
            if batch_norm and i < num_layers - 1:
                # default: nn.BatchNorm1d whose training-mode passes run in the library
                self.bns.append(nn.BatchNorm1d(hidden_feats) if plain else BatchNorm1d(hidden_feats))
        self.dropout = nn.Dropout(p=dropout)

    def reset_parameters(self):
        for layer in self.layers:
            layer.reset_parameters()
        for bn in self.bns:
            bn.reset_parameters()

    def invalidate_input_cache(self):
        """Forget the resident copy of the input features (the next forward copies them again)."""
        if self._input_cat is not None:
            self._input_cat[2].static_key = None

    def forward(self, g, x, rows=None):
        """`rows`: return the log-probabilities of these nodes only.  log_softmax is row-wise, so model(g, x, rows) ==
        model(g, x)[rows] (main_dgl_product_sage.py:105: `model(g, feats)[train_idx]`) without normalising -- forward and
        backward -- the 92 % of the rows the loss never reads."""
        # default model: every layer's input lives in the left half of an [N, 2K] buffer whose right half receives the
        # aggregation, so a layer is ONE GEMM (ops.CatBuffer).  The input features are copied there once (same tensor,
        # unmodified -> kept); hidden activations are written there by relu_dropout directly.
        cat = None
        if not self.plain:
            # keyed on the graph OBJECT (weak reference: an id() can be reused after garbage collection) and the input shape.
            # The buffer skips the copy of an input it already holds -- same tensor object, same version counter -- so the
            # features must not be changed behind autograd's back (`.data` writes, raw kernels): call
            # invalidate_input_cache() after such a write.  A step captured by utils.GraphedStep never contains the copy.
            held = self._input_cat
            if held is None or held[0]() is not g or held[1] != tuple(x.shape):
                c = ops.cat_buffer_for(g, x, x.shape[1])
                self._input_cat = held = (weakref.ref(g), tuple(x.shape), c) if c is not None else None
            cat = None if held is None or not torch.is_grad_enabled() else held[2]
        for i, layer in enumerate(self.layers[:-1]):
            nxt = None
            if not self.plain and not len(self.bns) and self.training and cat is not None and layer.fc_self.bias is None:
                # layer + relu + dropout as one node: the GEMM's epilogue applies the activation and writes the next layer's left half
                nxt = ops.cat_buffer_for(g, x, layer.fc_self.weight.shape[0])
                h = ops.sage_mean_layer_act(g, x, layer.fc_self.weight, layer.fc_neigh.weight, layer.fc_neigh.bias, cat, self.dropout.p,
                                            None if nxt is None else nxt.left)
                if h is not None:
                    x, cat = h, (nxt if nxt is not None and nxt.holds(h) else None)
                    continue
            x = layer(g, x, cat=cat) if not self.plain else layer(g, x)
            if len(self.bns):
                x = self.bns[i](x)
            if self.plain:
                x = self.dropout(F.relu(x))
            else:
                # (a buffer allocated for the fused form above that did not apply is the one this layer's activation goes to)
                cat = (nxt if nxt is not None and nxt.K == x.shape[1] else ops.cat_buffer_for(g, x, x.shape[1])) if self.training else None
                # F.relu + dropout, one pass each way on the device, written into the next layer's left half
                x = ops.relu_dropout(x, self.dropout.p, self.training, out=None if cat is None else cat.left)
                if cat is not None and not cat.holds(x):
                    cat = None
        last = self.layers[-1]
        if rows is not None and self.rows_are_distinct and not self.plain and last.fc_self.bias is None:
            # opt-in (MGX_SAGE_SPARSE_LAST=1): the last layer and the row selection as one node whose backward knows that the
            # output gradient is zero outside the loss rows
            y = ops.sage_mean_layer_rows(g, x, last.fc_self.weight, last.fc_neigh.weight, last.fc_neigh.bias, cat, rows)
            if y is not None:
                return y.log_softmax(dim=-1)
        x = self.layers[-1](g, x, cat=cat) if not self.plain else self.layers[-1](g, x)
        if rows is not None:
            x = ops.select_distinct_rows(x, rows) if self.rows_are_distinct else x[rows]
        return x.log_softmax(dim=-1)


class GAT(nn.Module):
    """Stack of dgl.nn GATConv layers: hidden layers are flattened over heads, the output layer is
    averaged over heads (main_dgl_reddit_gat.py:60-65)."""

    def __init__(self, g, num_layers, in_feats, num_hidden, num_classes, heads, activation=F.elu,
                 feat_drop=0.0, attn_drop=0.0, negative_slope=0.2):
        super(GAT, self).__init__()
        self.num_layers, self.g = num_layers, g
        self.gat_layers = nn.ModuleList()
        self.gat_layers.append(GATConv(in_feats, num_hidden, heads[0], 0., 0., negative_slope, activation=activation))
        for l in range(num_layers - 2):
            self.gat_layers.append(GATConv(num_hidden * heads[l], num_hidden, heads[l + 1], feat_drop, attn_drop,
                                           negative_slope, activation=activation))
        self.gat_layers.append(GATConv(num_hidden * heads[-2], num_classes, heads[-1], feat_drop, attn_drop,
                                       negative_slope, activation=None))

    def reset_parameters(self):
        for layer in self.gat_layers:
            layer.reset_parameters()

    def forward(self, h):
        for l in range(self.num_layers - 1):
            h = self.gat_layers[l](self.g, h).flatten(1)
        return self.gat_layers[-1](self.g, h).mean(1)


def sage_train_step(model, g, feats, labels, train_idx, optimizer):
    """main_dgl_product_sage.py:101-110."""
    model.train()
    optimizer.zero_grad()
    if model.plain:
        loss = F.nll_loss(model(g, feats)[train_idx], labels[train_idx])
    else:  # same numbers: log_softmax on the loss rows only, the mean NLL as a gather + sum
        loss = ops.nll_sum(model(g, feats, rows=train_idx), labels[train_idx]) / train_idx.shape[0]
    loss.backward()
    optimizer.step()
    return loss.item()


def gat_train_step(model, feats, labels, train_mask, optimizer, loss_fcn):
    """main_dgl_reddit_gat.py:155-168."""
    model.train()
    logits = model(feats)
    loss = loss_fcn(logits[train_mask], labels[train_mask])
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.item()


GAT_CONFIGS = {
    # the reference scripts' defaults: layers, hidden, heads (hidden layers / output layer), feat_drop = attn_drop
    "reddit": dict(num_layers=3, hidden=16, heads=1, out_heads=1, dropout=0.18074706609292976),   # main_dgl_reddit_gat.py:87-96,145-147
    "arxiv": dict(num_layers=3, hidden=16, heads=4, out_heads=4, dropout=0.18074706609292976),    # main_dgl_arxiv_gat.py:102-111,139-141
    "cora": dict(num_layers=3, hidden=8, heads=8, out_heads=1, dropout=0.6),                      # main_dgl_citation_gat.py:87-96,146-148
    "pubmed": dict(num_layers=3, hidden=8, heads=8, out_heads=1, dropout=0.6),
    "reddit-small": dict(num_layers=2, hidden=16, heads=8, out_heads=1, dropout=0.0),             # BASELINE config 3 (2-layer, 8 heads)
}

SAGE_CONFIGS = {
    # name: dataset, layers, hidden, dropout, lr, batch_norm, bidirect, neigh_bias   (script defaults)
    "products": dict(dataset="products", num_layers=3, hidden=64, dropout=0.5, lr=0.01, batch_norm=False,
                     bidirect=False, neigh_bias=True),   # main_dgl_product_sage.py:136-143
    "arxiv": dict(dataset="arxiv", num_layers=3, hidden=256, dropout=0.5, lr=0.01, batch_norm=True,
                  bidirect=True, neigh_bias=False),      # main_dgl_arxiv_sage.py:141-148,162
    "cora": dict(dataset="cora", num_layers=2, hidden=16, dropout=0.5, lr=0.01, batch_norm=False,
                 bidirect=False, neigh_bias=True),       # main_dgl_citation_sage.py:100-101,139
    "pubmed": dict(dataset="pubmed", num_layers=2, hidden=16, dropout=0.5, lr=0.01, batch_norm=False,
                   bidirect=False, neigh_bias=True),
    "reddit": dict(dataset="reddit", num_layers=2, hidden=16, dropout=0.5, lr=0.01, batch_norm=False,
                   bidirect=False, neigh_bias=True),    # main_dgl_reddit_sage.py:100-101,139 (E = 114.6 M, D = 602 then 16)
    "reddit-small": dict(dataset="reddit-small", num_layers=2, hidden=16, dropout=0.5, lr=0.01, batch_norm=False,
                         bidirect=False, neigh_bias=True),
}


def build_sage(name, device, scale=1.0):
    from mi355x_graph.datasets import NodeData
    cfg = SAGE_CONFIGS[name]
    data = NodeData(cfg["dataset"], device=device, scale=scale)
    g = data.graph
    if cfg["bidirect"]:
        feats_keep = g.ndata["feat"]
        g = dgl.to_bidirected(g)
        g.ndata["feat"] = feats_keep
    g = g.int().formats(["csr", "csc"]).to(device)
    model = GraphSAGE(data.features.shape[1], cfg["hidden"], data.num_classes, cfg["num_layers"], cfg["dropout"],
                      cfg["batch_norm"], cfg["neigh_bias"], plain=os.environ.get("MGX_PLAIN_MODEL", "0") == "1").to(device)
    train_idx = torch.nonzero(data.train_mask).flatten()
    model.rows_are_distinct = True  # nonzero() of a mask
    opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"])
    return cfg, data, g, model, train_idx, opt


def spmm_edges_per_epoch(num_layers, num_edges):
    """L forward SpMMs + (L-1) backward SpMMs (the input features need no gradient)."""
    return (2 * num_layers - 1) * num_edges


def main():
    p = argparse.ArgumentParser("full-graph training on the MI355X message-passing backend")
    p.add_argument("--model", default="sage", choices=["sage", "gat"])
    p.add_argument("--dataset", default="products")
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--scale", type=float, default=1.0)
    p.add_argument("--heads", type=int, default=None, help="heads of the hidden layers (default: the reference script's)")
    p.add_argument("--out-heads", type=int, default=None, help="heads of the output layer (default: the reference script's)")
    p.add_argument("--num-layers", type=int, default=None)
    p.add_argument("--num-hidden", type=int, default=None)
    p.add_argument("--dropout", type=float, default=None, help="GAT: feat_drop = attn_drop (default: the reference script's)")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--hipgraph", action="store_true",
                   help="capture the training step in a HIP graph and replay it (launch-bound small graphs)")
    args = p.parse_args()
    device = torch.device("cuda:%d" % args.device)
    dur = []
    if args.model == "sage":
        cfg, data, g, model, train_idx, opt = build_sage(args.dataset, device, args.scale)
        print(g)
        if args.hipgraph:
            from dgl.utils import GraphedStep
            opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"], capturable=True)
            model.train()
            y_train = data.labels[train_idx]
            # the loss tail of sage_train_step (training rows only, gather + sum): no host synchronisation, capture-safe
            graphed = GraphedStep(lambda: ops.nll_sum(model(g, data.features, rows=train_idx), y_train) / train_idx.shape[0], opt)
        for epoch in range(1, args.epochs + 1):
            t0 = time.time()
            if args.hipgraph:
                loss = graphed().item()
            else:
                loss = sage_train_step(model, g, data.features, data.labels, train_idx, opt)
            if epoch >= 3:
                dur.append(time.time() - t0)
            print("epoch %d loss %.4f time %.4f" % (epoch, loss, time.time() - t0))
        edges = spmm_edges_per_epoch(cfg["num_layers"], g.number_of_edges())
    else:
        from mi355x_graph.datasets import NodeData
        data = NodeData(args.dataset, device=device, scale=args.scale)
        g = dgl.add_self_loop(data.graph).int().to(device)
        gcfg = dict(GAT_CONFIGS.get(args.dataset, GAT_CONFIGS["reddit-small"]))
        for key, val in (("heads", args.heads), ("out_heads", args.out_heads), ("num_layers", args.num_layers),
                         ("hidden", args.num_hidden), ("dropout", args.dropout)):
            if val is not None:
                gcfg[key] = val
        heads = [gcfg["heads"]] * (gcfg["num_layers"] - 1) + [gcfg["out_heads"]]
        print("GAT: %d layers, heads %s, hidden %d, feat_drop = attn_drop = %g" % (gcfg["num_layers"], heads, gcfg["hidden"], gcfg["dropout"]))
        model = GAT(g, gcfg["num_layers"], data.features.shape[1], gcfg["hidden"], data.num_classes, heads,
                    feat_drop=gcfg["dropout"], attn_drop=gcfg["dropout"]).to(device)
        opt = torch.optim.Adam(model.parameters(), lr=0.003, weight_decay=2.4e-5)
        loss_fcn = nn.CrossEntropyLoss()
        if args.hipgraph:
            from dgl.utils import GraphedStep
            opt = torch.optim.Adam(model.parameters(), lr=0.003, weight_decay=2.4e-5, capturable=True)
            train_idx = torch.nonzero(data.train_mask).flatten()  # boolean-mask indexing synchronises; an index does not
            model.train()
            graphed = GraphedStep(lambda: loss_fcn(model(data.features)[train_idx], data.labels[train_idx]), opt)
        for epoch in range(args.epochs):
            t0 = time.time()
            if args.hipgraph:
                loss = graphed().item()
            else:
                loss = gat_train_step(model, data.features, data.labels, data.train_mask, opt, loss_fcn)
            if epoch >= 3:
                dur.append(time.time() - t0)
            print("epoch %d loss %.4f time %.4f" % (epoch, loss, time.time() - t0))
        edges = 0
    if dur:
        mean = sum(dur) / len(dur)
        print("Training time/epoch {:.5f}".format(mean))
        if edges:
            print("aggregated edges/s {:.3e}".format(edges / mean))


if __name__ == "__main__":
    main()
