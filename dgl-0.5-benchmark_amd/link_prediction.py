"""GCMC rating prediction on a MovieLens-shaped heterograph -- the MI355X launcher for the reference's
end_to_end/full_graph/link_prediction/gcmc_dgl (train.py:69-185, model.py:14-409, data.py:245-306), restated on this
repo's `dgl` surface: one `copy_u/sum` g-SpMM per rating relation and direction (HeteroGraphConv), a bilinear decoder whose
scores are `u_dot_v` g-SDDMMs over the user->movie decoder graph.

Synthetic stand-in (no network): users/movies/ratings of the published MovieLens shapes (ml-100k 943 x 1,682 x 100,000;
ml-1m 6,040 x 3,706 x 1,000,209), five rating levels with the ml-1m level frequencies, popularity-skewed endpoints,
identity ("one-hot") node features as with the reference's --use_one_hot_fea.

  PYTHONPATH=dgl-0.5-benchmark_amd python dgl-0.5-benchmark_amd/link_prediction.py --data_name ml-1m --train_max_iter 30
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dgl  # noqa: E402
import dgl.function as fn  # noqa: E402
import dgl.nn.pytorch as dglnn  # noqa: E402

SHAPES = {"ml-100k": (943, 1682, 100000), "ml-1m": (6040, 3706, 1000209)}
LEVEL_P = np.array([0.056, 0.108, 0.261, 0.349, 0.226])


def etype_name(rating):
    return str(rating).replace(".", "_")  # parameter names cannot contain "." (utils.py:81)


class RatingData(object):
    """Train / valid / test rating triples and the graphs built from them (data.py:100-236)."""

    def __init__(self, name, device, seed=7, test_ratio=0.1, valid_ratio=0.1):
        nu, nm, nr = SHAPES[name]
        rng = np.random.default_rng(seed)
        pu = 1.0 / np.arange(1, nu + 1) ** 0.6
        pm = 1.0 / np.arange(1, nm + 1) ** 0.9
        key = np.unique(rng.choice(nu, int(nr * 1.25), p=pu / pu.sum()).astype(np.int64) * nm
                        + rng.choice(nm, int(nr * 1.25), p=pm / pm.sum()))
        key = rng.permutation(key)[:nr]
        users, movies = key // nm, key % nm
        ratings = rng.choice(5, key.shape[0], p=LEVEL_P / LEVEL_P.sum()).astype(np.float32) + 1.0
        self.num_user, self.num_movie = nu, nm
        self.possible_rating_values = np.unique(ratings)
        n_test = int(np.ceil(key.shape[0] * test_ratio))
        n_valid = int(np.ceil((key.shape[0] - n_test) * valid_ratio))
        sl = {"test": slice(0, n_test), "valid": slice(n_test, n_test + n_valid), "train": slice(n_test + n_valid, None)}
        self.device = device
        self.pairs = {k: (users[s], movies[s]) for k, s in sl.items()}
        self.truths = {k: torch.from_numpy(ratings[s]).to(device) for k, s in sl.items()}
        self.labels = {k: torch.from_numpy(np.searchsorted(self.possible_rating_values, ratings[s])).to(device)
                       for k, s in sl.items()}
        self.values = {k: ratings[s] for k, s in sl.items()}
        # encoders: train ratings (valid uses the same graph), test sees train + valid (data.py:186-211)
        self.train_enc_graph = self._enc_graph(self.pairs["train"], self.values["train"])
        self.valid_enc_graph = self.train_enc_graph
        both = tuple(np.concatenate([self.pairs["train"][i], self.pairs["valid"][i]]) for i in (0, 1))
        self.test_enc_graph = self._enc_graph(both, np.concatenate([self.values["train"], self.values["valid"]]))
        self.dec_graph = {k: dgl.bipartite(self.pairs[k], "user", "rate", "movie", num_nodes=(nu, nm)).int().to(device)
                          for k in sl}

    def _enc_graph(self, pairs, values):
        """One relation per rating level and direction; symmetric normalisers c_i, c_j on the nodes (data.py:245-299)."""
        nu, nm = self.num_user, self.num_movie
        rels = []
        for r in self.possible_rating_values:
            sel = values == r
            u, m = pairs[0][sel], pairs[1][sel]
            rels.append(dgl.bipartite((u, m), "user", etype_name(r), "movie", num_nodes=(nu, nm)))
            rels.append(dgl.bipartite((m, u), "movie", "rev-" + etype_name(r), "user", num_nodes=(nm, nu)))
        g = dgl.hetero_from_relations(rels)
        deg_u = torch.bincount(torch.from_numpy(pairs[0]), minlength=nu).float()
        deg_m = torch.bincount(torch.from_numpy(pairs[1]), minlength=nm).float()

        def norm(d):
            d = d.clone()
            d[d == 0] = float("inf")
            return (1.0 / d.sqrt()).unsqueeze(1)

        g.nodes["user"].data.update({"ci": norm(deg_u), "cj": norm(deg_u)})
        g.nodes["movie"].data.update({"ci": norm(deg_m), "cj": norm(deg_m)})
        return g.int().to(self.device)


class GCMCGraphConv(nn.Module):
    """h_dst = c_i * sum_{src} dropout(c_j) * W[src]  (identity input features: the projection IS the weight rows;
    model.py:14-96)."""

    def __init__(self, in_feats, out_feats, dropout_rate=0.0):
        super(GCMCGraphConv, self).__init__()
        self.weight = nn.Parameter(torch.Tensor(in_feats, out_feats))
        self.dropout = nn.Dropout(dropout_rate)
        nn.init.xavier_uniform_(self.weight)

    def forward(self, graph, feat):
        with graph.local_scope():
            feat = feat[0] if isinstance(feat, tuple) else feat
            h = self.weight if feat is None else feat @ self.weight
            graph.srcdata["h"] = h * self.dropout(graph.srcdata["cj"])
            graph.update_all(fn.copy_u("h", "m"), fn.sum("m", "h"))
            return graph.dstdata["h"] * graph.dstdata["ci"]


class GCMCLayer(nn.Module):
    """model.py:98-271: per-rating convolutions combined by 'sum' or 'stack', then activation, dropout, dense."""

    def __init__(self, rating_vals, user_in, movie_in, msg_units, out_units, dropout_rate, agg, agg_act):
        super(GCMCLayer, self).__init__()
        self.ufc = nn.Linear(msg_units, out_units)
        self.ifc = nn.Linear(msg_units, out_units)
        if agg == "stack":
            assert msg_units % len(rating_vals) == 0
            msg_units = msg_units // len(rating_vals)
        convs = {}
        for r in rating_vals:
            convs[etype_name(r)] = GCMCGraphConv(user_in, msg_units, dropout_rate)
            convs["rev-" + etype_name(r)] = GCMCGraphConv(movie_in, msg_units, dropout_rate)
        self.conv = dglnn.HeteroGraphConv(convs, aggregate=agg)
        self.dropout = nn.Dropout(dropout_rate)
        self.agg_act = agg_act
        for p in (self.ufc.weight, self.ifc.weight):
            nn.init.xavier_uniform_(p)

    def forward(self, graph, ufeat=None, ifeat=None):
        out = self.conv(graph, {"user": ufeat, "movie": ifeat})
        u = self.dropout(self.agg_act(out["user"].flatten(1)))
        m = self.dropout(self.agg_act(out["movie"].flatten(1)))
        return self.ufc(u), self.ifc(m)


class BiDecoder(nn.Module):
    """p(M_ij = r) = softmax_r(sum_s a_rs u_i^T P_s v_j); each basis is one u_dot_v g-SDDMM (model.py:273-344)."""

    def __init__(self, in_units, num_classes, num_basis=2):
        super(BiDecoder, self).__init__()
        self.Ps = nn.ParameterList([nn.Parameter(torch.Tensor(in_units, in_units)) for _ in range(num_basis)])
        self.combine_basis = nn.Linear(num_basis, num_classes, bias=False)
        for p in self.parameters():
            nn.init.xavier_uniform_(p)

    def forward(self, graph, ufeat, ifeat):
        with graph.local_scope():
            graph.nodes["movie"].data["h"] = ifeat
            basis_out = []
            for P in self.Ps:
                graph.nodes["user"].data["h"] = ufeat @ P
                graph.apply_edges(fn.u_dot_v("h", "h", "sr"))
                basis_out.append(graph.edata["sr"])
            return self.combine_basis(torch.cat(basis_out, dim=1))


class Net(nn.Module):
    def __init__(self, data, args):
        super(Net, self).__init__()
        self.encoder = GCMCLayer(data.possible_rating_values, data.num_user, data.num_movie, args.gcn_agg_units,
                                 args.gcn_out_units, args.gcn_dropout, args.gcn_agg_accum, nn.LeakyReLU(0.1))
        self.decoder = BiDecoder(args.gcn_out_units, len(data.possible_rating_values), args.gen_r_num_basis_func)

    def forward(self, enc_graph, dec_graph):
        u, m = self.encoder(enc_graph)
        return self.decoder(dec_graph, u, m)


def evaluate(net, data, levels, segment):
    net.eval()
    with torch.no_grad():
        enc = data.valid_enc_graph if segment == "valid" else data.test_enc_graph
        pred = net(enc, data.dec_graph[segment])
    real = (torch.softmax(pred, dim=1) * levels.view(1, -1)).sum(dim=1)
    return float(((real - data.truths[segment]) ** 2).mean().sqrt())


def main():
    p = argparse.ArgumentParser(description="GCMC on a MovieLens-shaped heterograph")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--data_name", default="ml-1m", choices=sorted(SHAPES))
    p.add_argument("--gcn_dropout", type=float, default=0.7)
    p.add_argument("--gcn_agg_units", type=int, default=500)
    p.add_argument("--gcn_agg_accum", default="sum", choices=["sum", "stack"])
    p.add_argument("--gcn_out_units", type=int, default=75)
    p.add_argument("--gen_r_num_basis_func", type=int, default=2)
    p.add_argument("--train_max_iter", type=int, default=30)
    p.add_argument("--train_valid_interval", type=int, default=10)
    p.add_argument("--train_grad_clip", type=float, default=1.0)
    p.add_argument("--train_lr", type=float, default=0.01)
    p.add_argument("--seed", type=int, default=123)
    args = p.parse_args()
    torch.manual_seed(args.seed)
    device = torch.device("cuda:%d" % args.device if args.device >= 0 else "cpu")
    data = RatingData(args.data_name, device)
    print("users %d movies %d train/valid/test ratings %d/%d/%d, %d relations" % (
        data.num_user, data.num_movie, data.truths["train"].shape[0], data.truths["valid"].shape[0],
        data.truths["test"].shape[0], len(data.train_enc_graph.etypes)))
    net = Net(data, args).to(device)
    levels = torch.from_numpy(data.possible_rating_values).to(device)
    loss_fn = nn.CrossEntropyLoss()
    opt = torch.optim.Adam(net.parameters(), lr=args.train_lr)
    dur, best_valid, best_test = [], float("inf"), float("nan")
    for it in range(1, args.train_max_iter + 1):
        if device.type == "cuda":
            torch.cuda.synchronize()
        t0 = time.time()
        net.train()
        pred = net(data.train_enc_graph, data.dec_graph["train"])
        loss = loss_fn(pred, data.labels["train"])
        opt.zero_grad()
        loss.backward()
        nn.utils.clip_grad_norm_(net.parameters(), args.train_grad_clip)
        opt.step()
        loss_v = loss.item()  # host sync, as in train.py:125
        if it > 3:            # the reference discards the first three iterations (train.py:118,131)
            dur.append(time.time() - t0)
        msg = "Iter=%d, loss=%.4f, time=%.4f" % (it, loss_v, np.average(dur) if dur else float("nan"))
        if it % args.train_valid_interval == 0:
            v = evaluate(net, data, levels, "valid")
            msg += ",\tVal RMSE=%.4f" % v
            if v < best_valid:
                best_valid, best_test = v, evaluate(net, data, levels, "test")
                msg += ", Test RMSE=%.4f" % best_test
        print(msg)
    edges = 2 * data.truths["train"].shape[0]
    print("Training time/epoch {:.5f}".format(np.average(dur)))
    print("Best Valid RMSE=%.4f, Best Test RMSE=%.4f; encoder edges/iter %d" % (best_valid, best_test, edges))


if __name__ == "__main__":
    main()
