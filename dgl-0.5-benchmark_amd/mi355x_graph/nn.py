"""dgl.nn.pytorch modules used by the benchmark scripts, written against this backend's ops:
  GATConv    main_dgl_reddit_gat.py:10,31-55 (u_add_v SDDMM + fused edge_softmax + u_mul_e/sum SpMM)
  SAGEConv   main_dgl_arxiv_sage_nn.py:9,27-34
  GraphConv  main_dgl_enzymes_gcn_nn.py:12,29-36
  AvgPooling / SumPooling / MaxPooling   main_dgl_molhiv_gcn.py:75,93
Parameter names, shapes and initialisation follow DGL v0.6.x so state_dicts and the scripts'
reset_parameters() calls (main_dgl_reddit_gat.py:57-59) line up.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ._lib import DGLError
from . import function as fn
from . import config
from . import ops
from .utils import expand_as_pair


def _has_zero_in_degree(graph):
    index = getattr(graph, "_index", None)
    if index is not None and hasattr(index, "has_zero_in_degree"):
        return index.has_zero_in_degree()
    return bool((graph.in_degrees() == 0).any())


class Linear(nn.Linear):
    """torch.nn.Linear (same parameters, init and state_dict) whose bias gradient runs in the library's column-sum kernel."""

    def forward(self, input):
        return ops.linear(input, self.weight, self.bias)


class BatchNorm1d(nn.BatchNorm1d):
    """torch.nn.BatchNorm1d (same parameters, buffers and state_dict) over node features [N, C]: in training mode the
    statistics, the normalisation and their gradients run in the library's column kernels (PyTorch's channels-last
    batch-norm kernels need 0.6-0.9 ms for [169 k, 256]; the data is 173 MB).  Evaluation mode, C > 256 or C % 4 != 0 and
    non-2-D inputs take the PyTorch path."""

    def forward(self, input):
        use_batch = self.training or not self.track_running_stats
        if not use_batch or not ops.batch_norm_supported(input):
            return super(BatchNorm1d, self).forward(input)
        y, mean, var = ops.BatchNormFn.apply(input, self.weight, self.bias, self.eps)
        if self.training and self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked.add_(1)
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                n = input.shape[0]
                self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.mul_(1 - m).add_(var, alpha=m * n / (n - 1))
        return y


class Identity(nn.Module):
    def forward(self, x):
        return x


class GATConv(nn.Module):
    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0., attn_drop=0., negative_slope=0.2,
                 residual=False, activation=None, allow_zero_in_degree=False, bias=True):
        super(GATConv, self).__init__()
        self._num_heads = num_heads
        self._in_src_feats, self._in_dst_feats = expand_as_pair(in_feats)
        self._out_feats = out_feats
        self._allow_zero_in_degree = allow_zero_in_degree
        if isinstance(in_feats, tuple):
            self.fc_src = Linear(self._in_src_feats, out_feats * num_heads, bias=False)
            self.fc_dst = Linear(self._in_dst_feats, out_feats * num_heads, bias=False)
        else:
            self.fc = Linear(self._in_src_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.FloatTensor(size=(1, num_heads, out_feats)))
        self.attn_r = nn.Parameter(torch.FloatTensor(size=(1, num_heads, out_feats)))
        self.feat_drop = nn.Dropout(feat_drop)
        self.attn_drop = nn.Dropout(attn_drop)
        self.leaky_relu = nn.LeakyReLU(negative_slope)
        if bias:
            self.bias = nn.Parameter(torch.FloatTensor(size=(num_heads * out_feats,)))
        else:
            self.register_buffer("bias", None)
        if residual:
            if self._in_dst_feats != out_feats:
                self.res_fc = Linear(self._in_dst_feats, num_heads * out_feats, bias=False)
            else:
                self.res_fc = Identity()
        else:
            self.register_buffer("res_fc", None)
        self.reset_parameters()
        self.activation = activation

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        if hasattr(self, "fc"):
            nn.init.xavier_normal_(self.fc.weight, gain=gain)
        else:
            nn.init.xavier_normal_(self.fc_src.weight, gain=gain)
            nn.init.xavier_normal_(self.fc_dst.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)
        if self.bias is not None:
            nn.init.constant_(self.bias, 0)
        if isinstance(self.res_fc, nn.Linear):
            nn.init.xavier_normal_(self.res_fc.weight, gain=gain)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    def _aggregate_first(self, graph, feat, get_attention):
        """One-head layers that WIDEN (in < out, e.g. 16 hidden -> 41 classes): aggregate the input rows, project afterwards.
        Taken when the fused block exists for the input width and it saves a 16-column gather pass; MGX_GAT_AGG_FIRST=0 never."""
        if (self._num_heads != 1 or isinstance(feat, tuple) or get_attention or not hasattr(self, "fc") or not torch.is_tensor(feat)
                or feat.dim() != 2 or not config.GAT_AGG_FIRST):
            return False
        k, f = self._in_src_feats, self._out_feats
        if k % 4 != 0 or (k + 15) // 16 >= (f + 15) // 16:
            return False
        if self.training and self.attn_drop.p > 0.0 and feat.is_cuda and ops.capture_path():
            return False  # frozen mask under capture: see forward
        return ops.gat_fused_supported(graph, feat.view(feat.shape[0], 1, k))

    def _forward_on_partition(self, graph, feat):
        """A partition of the graph (dist.DistGraph): every owned node has all its in-edges here, its remote in-neighbours'
        rows travel ONCE per layer -- the layer's input or its projection, whichever is narrower (reddit: 602 -> 16: the 16 projected
        columns) -- through the halo exchange (autograd adds the halo rows' gradients back into their owners), and the layer then
        runs on the local block graph [owned | halo] -> owned exactly as on a sampled block."""
        h = self.feat_drop(feat)
        blk = graph._block
        if self._in_src_feats <= self._num_heads * self._out_feats:
            return self.forward(blk, graph.halo_exchange(h), _dropped=True)
        proj = graph.halo_exchange(self.fc(h))                      # [n_own + n_halo, H * F]
        return self.forward(blk, h, _projected=proj)

    def forward(self, graph, feat, get_attention=False, _dropped=False, _projected=None):
        if hasattr(graph, "halo_exchange") and _projected is None and not _dropped:
            if isinstance(feat, tuple) or get_attention or not hasattr(self, "fc"):
                raise DGLError("GATConv on a partitioned graph takes one feature tensor of the owned nodes (no attention output)")
            return self._forward_on_partition(graph, feat)
        with graph.local_scope():
            if not self._allow_zero_in_degree:
                if _has_zero_in_degree(graph):
                    raise DGLError(
                        "There are 0-in-degree nodes in the graph, output for those nodes will be invalid. "
                        "This is harmful for some applications, causing silent performance regression. "
                        "Adding self-loop on the input graph by calling `g = dgl.add_self_loop(g)` will resolve "
                        "the issue. Setting ``allow_zero_in_degree`` to be `True` when constructing this module "
                        "will suppress the check and let the code run.")
            if _projected is None and self._aggregate_first(graph, feat, get_attention):
                # ONE head: a[e] is a scalar, so sum_e a[e] (h[u] W^T) = (sum_e a[e] h[u]) W^T and el = h (W^T attn_l): the
                # fused block runs at the INPUT width and the projection follows it (reddit GAT's 16 -> 41 output layer:
                # three gather walks at 64-byte rows instead of 164-byte ones).  Same function, other fp32 summation order.
                h = feat if _dropped else self.feat_drop(feat)
                w = self.fc.weight                                    # [F, in]
                v = torch.cat([self.attn_l.view(1, -1), self.attn_r.view(1, -1)], 0) @ w  # [2, in]
                lr = h @ v.t()                                        # [N, 2]: el, er
                n_dst = graph.number_of_dst_nodes()
                el, er = lr[:, 0:1].unsqueeze(-1), lr[:n_dst, 1:2].unsqueeze(-1)
                agg = ops.gat_fused(graph, h.view(h.shape[0], 1, -1), el.contiguous(), er.contiguous(),
                                    self.leaky_relu.negative_slope, self.attn_drop.p, self.training)
                rst = (agg.view(n_dst, -1) @ w.t()).view(n_dst, 1, self._out_feats)
                if self.res_fc is not None:
                    rst = rst + self.res_fc(h[:n_dst]).view(n_dst, -1, self._out_feats)
                if self.bias is not None:
                    rst = ops.bias_add(rst, self.bias)
                if self.activation:
                    rst = self.activation(rst)
                return rst
            if _projected is not None:  # a partition: `feat` = the owned rows after feat_drop, `_projected` = fc of [owned | halo]
                h_dst = feat
                feat_src = _projected.view(-1, self._num_heads, self._out_feats)
                feat_dst = feat_src[:graph.number_of_dst_nodes()]
            elif isinstance(feat, tuple):
                h_src = self.feat_drop(feat[0])
                h_dst = self.feat_drop(feat[1])
                if not hasattr(self, "fc_src"):
                    feat_src = self.fc(h_src).view(-1, self._num_heads, self._out_feats)
                    feat_dst = self.fc(h_dst).view(-1, self._num_heads, self._out_feats)
                else:
                    feat_src = self.fc_src(h_src).view(-1, self._num_heads, self._out_feats)
                    feat_dst = self.fc_dst(h_dst).view(-1, self._num_heads, self._out_feats)
            else:
                h_src = h_dst = feat if _dropped else self.feat_drop(feat)
                feat_src = feat_dst = self.fc(h_src).view(-1, self._num_heads, self._out_feats)
                if graph.is_block:
                    feat_dst = feat_src[:graph.number_of_dst_nodes()]
                    h_dst = h_dst[:graph.number_of_dst_nodes()]
            if ops.head_dot_supported(feat_src):  # one pass over feat instead of product + last-dim reduction
                if feat_dst is feat_src:
                    el, er = ops.head_dot(feat_src, self.attn_l, self.attn_r)
                else:
                    el, er = ops.head_dot(feat_src, self.attn_l), ops.head_dot(feat_dst, self.attn_r)
                el, er = el.unsqueeze(-1), er.unsqueeze(-1)
            else:
                el = (feat_src * self.attn_l).sum(dim=-1).unsqueeze(-1)
                er = (feat_dst * self.attn_r).sum(dim=-1).unsqueeze(-1)
            rst = None
            # Under HIP-graph capture the fused block's attn_drop seed would be a frozen launch argument -- every replay the
            # same mask -- so a dropping layer takes the unfused path there (nn.Dropout is capture-safe); ADVICE r02.
            frozen_mask = self.training and self.attn_drop.p > 0.0 and feat_src.is_cuda and ops.capture_path()
            if not get_attention and not frozen_mask and ops.gat_fused_supported(graph, feat_src):
                # the whole block u_add_v -> leaky_relu -> edge_softmax -> attn_drop -> u_mul_e/sum without any E x H tensor
                rst = ops.gat_fused(graph, feat_src, el, er, self.leaky_relu.negative_slope, self.attn_drop.p, self.training,
                                    attn_l=self.attn_l)  # el = (feat_src * attn_l).sum(-1), whichever branch above formed it
            # e and a are internal to the module: keep them in in-CSR (destination-major) edge order so that
            # u_add_v, edge_softmax and u_mul_e/sum stream them instead of gathering by edge id
            cidx, perm = graph._index.canonical() if rst is None else (None, None)
            if rst is not None:
                pass
            elif int(el.shape[1]) <= 64:  # fused u_add_v -> leaky_relu -> edge_softmax: the logits are never materialised
                a = self.attn_drop(ops.gat_attention(cidx, el, er, self.leaky_relu.negative_slope))
            else:
                e = self.leaky_relu(ops.gsddmm(cidx, "add", el, er, "u", "v"))
                a = self.attn_drop(ops.edge_softmax(cidx, e))
            if rst is None:
                rst = ops.gspmm(cidx, "mul", "sum", feat_src, a)
            if self.res_fc is not None:
                resval = self.res_fc(h_dst).view(h_dst.shape[0], -1, self._out_feats)
                rst = rst + resval
            if self.bias is not None:
                rst = ops.bias_add(rst, self.bias)
            if self.activation:
                rst = self.activation(rst)
            if get_attention:
                if perm is not None:  # back to the caller's edge-id order
                    a_user = torch.empty_like(a)
                    a_user[perm.long()] = a
                    a = a_user
                return rst, a
            return rst


class SAGEConv(nn.Module):
    def __init__(self, in_feats, out_feats, aggregator_type, feat_drop=0., bias=True, norm=None, activation=None):
        super(SAGEConv, self).__init__()
        if aggregator_type not in ("mean", "gcn", "pool"):
            raise DGLError("Unsupported aggregator type %r on this backend (mean, gcn, pool)" % (aggregator_type,))
        self._in_src_feats, self._in_dst_feats = expand_as_pair(in_feats)
        self._out_feats = out_feats
        self._aggre_type = aggregator_type
        self.norm = norm
        self.feat_drop = nn.Dropout(feat_drop)
        self.activation = activation
        if aggregator_type == "pool":
            self.fc_pool = Linear(self._in_src_feats, self._in_src_feats)
        if aggregator_type != "gcn":
            self.fc_self = Linear(self._in_dst_feats, out_feats, bias=bias)
        self.fc_neigh = Linear(self._in_src_feats, out_feats, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        if self._aggre_type == "pool":
            nn.init.xavier_uniform_(self.fc_pool.weight, gain=gain)
        if self._aggre_type != "gcn":
            nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

    def forward(self, graph, feat):
        with graph.local_scope():
            if isinstance(feat, tuple):
                feat_src = self.feat_drop(feat[0])
                feat_dst = self.feat_drop(feat[1])
            else:
                feat_src = feat_dst = self.feat_drop(feat)
                if graph.is_block:
                    feat_dst = feat_src[:graph.number_of_dst_nodes()]
            h_self = feat_dst
            if graph.number_of_edges() == 0:
                graph.dstdata["neigh"] = torch.zeros(feat_dst.shape[0], self._in_src_feats).to(feat_dst)
            # aggregate after the projection when that moves fewer bytes (DGL's lin_before_mp)
            lin_before_mp = self._in_src_feats > self._out_feats
            if self._aggre_type == "mean":
                graph.srcdata["h"] = self.fc_neigh(feat_src) if lin_before_mp else feat_src
                graph.update_all(fn.copy_src("h", "m"), fn.mean("m", "neigh"))
                h_neigh = graph.dstdata["neigh"]
                if not lin_before_mp:
                    h_neigh = self.fc_neigh(h_neigh)
            elif self._aggre_type == "gcn":
                graph.srcdata["h"] = feat_src
                graph.dstdata["h"] = feat_dst
                graph.update_all(fn.copy_src("h", "m"), fn.sum("m", "neigh"))
                degs = graph.in_degrees().to(feat_dst)
                h_neigh = (graph.dstdata["neigh"] + graph.dstdata["h"]) / (degs.unsqueeze(-1) + 1)
                h_neigh = self.fc_neigh(h_neigh)
            else:  # pool
                graph.srcdata["h"] = F.relu(self.fc_pool(feat_src))
                graph.update_all(fn.copy_src("h", "m"), fn.max("m", "neigh"))
                h_neigh = self.fc_neigh(graph.dstdata["neigh"])
            rst = h_neigh if self._aggre_type == "gcn" else self.fc_self(h_self) + h_neigh
            if self.activation is not None:
                rst = self.activation(rst)
            if self.norm is not None:
                rst = self.norm(rst)
            return rst


class GraphConv(nn.Module):
    def __init__(self, in_feats, out_feats, norm="both", weight=True, bias=True, activation=None,
                 allow_zero_in_degree=False):
        super(GraphConv, self).__init__()
        if norm not in ("none", "both", "right"):
            raise DGLError('Invalid norm value. Must be either "none", "both" or "right". But got "%s".' % norm)
        self._in_feats, self._out_feats, self._norm = in_feats, out_feats, norm
        self._allow_zero_in_degree = allow_zero_in_degree
        if weight:
            self.weight = nn.Parameter(torch.Tensor(in_feats, out_feats))
        else:
            self.register_parameter("weight", None)
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_feats))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()
        self._activation = activation

    def reset_parameters(self):
        if self.weight is not None:
            nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    def forward(self, graph, feat, weight=None):
        with graph.local_scope():
            if not self._allow_zero_in_degree and _has_zero_in_degree(graph):
                raise DGLError("There are 0-in-degree nodes in the graph, output for those nodes will be invalid. "
                               "Adding self-loop on the input graph by calling `g = dgl.add_self_loop(g)` will "
                               "resolve the issue. Setting ``allow_zero_in_degree`` to be `True` when constructing "
                               "this module will suppress the check and let the code run.")
            feat_src, feat_dst = expand_as_pair(feat, graph)
            if self._norm == "both":
                degs = graph.out_degrees().to(feat_src).clamp(min=1)
                norm = torch.pow(degs, -0.5)
                feat_src = feat_src * norm.view((-1,) + (1,) * (feat_src.dim() - 1))
            if weight is not None and self.weight is not None:
                raise DGLError("External weight is provided while at the same time the module has defined its own "
                               "weight parameter. Please create the module with flag weight=False.")
            weight = self.weight if weight is None else weight
            if self._in_feats > self._out_feats:
                if weight is not None:
                    feat_src = torch.matmul(feat_src, weight)
                graph.srcdata["h"] = feat_src
                graph.update_all(fn.copy_src(src="h", out="m"), fn.sum(msg="m", out="h"))
                rst = graph.dstdata["h"]
            else:
                graph.srcdata["h"] = feat_src
                graph.update_all(fn.copy_src(src="h", out="m"), fn.sum(msg="m", out="h"))
                rst = graph.dstdata["h"]
                if weight is not None:
                    rst = torch.matmul(rst, weight)
            if self._norm != "none":
                degs = graph.in_degrees().to(feat_dst).clamp(min=1)
                norm = torch.pow(degs, -0.5) if self._norm == "both" else 1.0 / degs
                rst = rst * norm.view((-1,) + (1,) * (feat_dst.dim() - 1))
            if self.bias is not None:
                rst = ops.bias_add(rst, self.bias)
            if self._activation is not None:
                rst = self._activation(rst)
            return rst


class GINConv(nn.Module):
    """dgl.nn.pytorch.GINConv (UPSTREAM): h_v = f((1 + eps) h_v + AGG_{u->v} h_u), AGG = sum | mean | max -- a builtin
    copy_u aggregation (one g-SpMM) followed by `apply_func`.  BASELINE.json names "5-layer GIN on ogbg-molhiv" for the
    batched many-small-graphs path; the reference's own molhiv script is the GCN of graph_classification.py."""

    def __init__(self, apply_func=None, aggregator_type="sum", init_eps=0, learn_eps=False):
        super(GINConv, self).__init__()
        if aggregator_type not in ("sum", "max", "mean"):
            raise KeyError("Aggregator type {} not recognized.".format(aggregator_type))
        self.apply_func = apply_func
        self._reducer = getattr(fn, aggregator_type)
        if learn_eps:
            self.eps = nn.Parameter(torch.FloatTensor([init_eps]))
        else:
            self.register_buffer("eps", torch.FloatTensor([init_eps]))

    def forward(self, graph, feat, edge_weight=None):
        with graph.local_scope():
            feat_src, feat_dst = expand_as_pair(feat, graph)
            graph.srcdata["h"] = feat_src
            if edge_weight is not None:
                graph.edata["_edge_weight"] = edge_weight
                graph.update_all(fn.u_mul_e("h", "_edge_weight", "m"), self._reducer("m", "neigh"))
            else:
                graph.update_all(fn.copy_u("h", "m"), self._reducer("m", "neigh"))
            rst = (1 + self.eps) * feat_dst + graph.dstdata["neigh"]
            if self.apply_func is not None:
                rst = self.apply_func(rst)
            return rst


def _hetero_aggregate(name):
    if callable(name):
        return name
    if name == "stack":
        return lambda xs, dsttype: torch.stack(xs, dim=1) if xs else None
    fns = {"sum": lambda t: t.sum(0), "mean": lambda t: t.mean(0), "max": lambda t: t.max(0)[0], "min": lambda t: t.min(0)[0]}
    if name not in fns:
        raise DGLError('Invalid cross type aggregator. Must be one of "sum", "max", "min", "mean" or "stack". But got "%s"' % name)
    return lambda xs, dsttype: fns[name](torch.stack(xs, dim=0)) if xs else None


class HeteroGraphConv(nn.Module):
    """dglnn.HeteroGraphConv (gcmc_dgl/model.py:205): one sub-module per relation, each run on the relation's own
    graph view (its own in-CSR and execution plan), results combined per destination type by `aggregate`."""

    def __init__(self, mods, aggregate="sum"):
        super(HeteroGraphConv, self).__init__()
        self.mods = nn.ModuleDict(mods)
        for _, v in self.mods.items():  # isolated destinations are expected inside a single relation
            set_allow = getattr(v, "set_allow_zero_in_degree", None)
            if callable(set_allow):
                set_allow(True)
        self.agg_fn = _hetero_aggregate(aggregate)

    def forward(self, g, inputs, mod_args=None, mod_kwargs=None):
        mod_args = mod_args or {}
        mod_kwargs = mod_kwargs or {}
        outputs = {nty: [] for nty in g.dsttypes}
        if isinstance(inputs, tuple):
            src_inputs, dst_inputs = inputs
        elif g.is_block:
            src_inputs = inputs
            dst_inputs = {k: v[:g.number_of_dst_nodes(k)] for k, v in inputs.items()}
        else:
            src_inputs = dst_inputs = inputs
        for stype, etype, dtype in g.canonical_etypes:
            rel_graph = g[stype, etype, dtype]
            if rel_graph.number_of_edges() == 0:
                continue
            if stype not in src_inputs or dtype not in dst_inputs:
                continue
            dstdata = self.mods[etype](rel_graph, (src_inputs[stype], dst_inputs[dtype]), *mod_args.get(etype, ()),
                                       **mod_kwargs.get(etype, {}))
            outputs[dtype].append(dstdata)
        rsts = {}
        for nty, alist in outputs.items():
            if len(alist) != 0:
                rsts[nty] = self.agg_fn(alist, nty)
        return rsts


class _Pooling(nn.Module):
    _op = "sum"

    def forward(self, graph, feat):
        return ops.segment_reduce(graph.batch_num_nodes(), feat, self._op, total=graph.number_of_nodes())


class SumPooling(_Pooling):
    _op = "sum"


class AvgPooling(_Pooling):
    """Per-graph mean of node features (main_dgl_molhiv_gcn.py:75,93)."""
    _op = "mean"


class MaxPooling(_Pooling):
    _op = "max"
