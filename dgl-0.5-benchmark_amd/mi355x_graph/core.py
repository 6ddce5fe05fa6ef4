"""update_all / apply_edges dispatch: builtin x builtin -> one g-SpMM, builtin message -> g-SDDMM,
UDF message + builtin reduce -> per-edge gathers, the UDF, then a copy_e g-SpMM.

Reference call sites: main_dgl_product_sage.py:62 (copy_src/mean), main_dgl_proteins_rgcn_for.py:52
(u_mul_e/mean), main_dgl_molhiv_gcn.py:46,50-52 (UDF message + fn.sum), gcmc_dgl/model.py:342
(apply_edges(fn.u_dot_v)).
"""
import torch

from ._lib import DGLError
from . import function as fn
from . import ops


class _LazyGather(object):
    """edges.src / edges.dst view: gathers a node feature onto the edges on first access."""

    def __init__(self, g, frame, target):
        self._g, self._frame, self._target, self._cache = g, frame, target, {}

    def __getitem__(self, key):
        if key not in self._cache:
            x = self._frame[key]
            if self._target == "u":
                self._cache[key] = ops.gsddmm(self._g, "copy_lhs", x, None, "u", "v")
            else:
                self._cache[key] = ops.gsddmm(self._g, "copy_rhs", None, x, "u", "v")
        return self._cache[key]

    def __contains__(self, key):
        return key in self._frame

    def keys(self):
        return self._frame.keys()


class EdgeBatch(object):
    """Argument of a message UDF: edges.src[k], edges.dst[k], edges.data[k] (main_dgl_molhiv_gcn.py:50-52)."""

    def __init__(self, g):
        self._g = g
        self.src = _LazyGather(g, g.srcdata, "u")
        self.dst = _LazyGather(g, g.dstdata, "v")
        self.data = g.edata

    def edges(self):
        s, d = self._g.edges()
        return s, d, torch.arange(s.shape[0], dtype=s.dtype, device=s.device)

    def batch_size(self):
        return self._g.number_of_edges()

    def __len__(self):
        return self.batch_size()


class NodeBatch(object):
    def __init__(self, g, data):
        self._g, self.data = g, data

    def nodes(self):
        return self._g.dstnodes()

    def batch_size(self):
        return self._g.number_of_dst_nodes()


def _field(frame, key, what):
    if key not in frame:
        raise DGLError("Cannot find field %r in the %s features" % (key, what))
    return frame[key]


def _operand(g, target, field):
    if target == "u":
        return _field(g.srcdata, field, "source node")
    if target == "v":
        return _field(g.dstdata, field, "destination node")
    return _field(g.edata, field, "edge")


def _eval_message(g, mfunc):
    """Materialise a builtin message on the edges (g-SDDMM)."""
    if isinstance(mfunc, fn.CopyMessageFunction):
        x = _operand(g, mfunc.target, mfunc.in_field)
        if mfunc.target == "e":
            return x
        return ops.gsddmm(g, "copy_lhs", x, None, mfunc.target, "v")
    lhs = _operand(g, mfunc.lhs, mfunc.lhs_field)
    rhs = _operand(g, mfunc.rhs, mfunc.rhs_field)
    return ops.gsddmm(g, mfunc.binary_op, lhs, rhs, mfunc.lhs, mfunc.rhs)


def _builtin_spmm(g, mfunc, rfunc):
    """builtin message x builtin reduce -> one g-SpMM when the message only reads u and e."""
    red = rfunc.name
    if isinstance(mfunc, fn.CopyMessageFunction):
        x = _operand(g, mfunc.target, mfunc.in_field)
        if mfunc.target == "u":
            return ops.gspmm(g, "copy_lhs", red, x, None)
        if mfunc.target == "e":
            return ops.gspmm(g, "copy_rhs", red, None, x)
    elif mfunc.binary_op != "dot":
        if mfunc.lhs == "u" and mfunc.rhs == "e":
            return ops.gspmm(g, mfunc.binary_op, red, _operand(g, "u", mfunc.lhs_field), _operand(g, "e", mfunc.rhs_field))
        if mfunc.lhs == "e" and mfunc.rhs == "u" and mfunc.binary_op in ("add", "mul"):
            return ops.gspmm(g, mfunc.binary_op, red, _operand(g, "u", mfunc.rhs_field), _operand(g, "e", mfunc.lhs_field))
    # message reads the destination side (or is a dot): materialise it, then reduce the edge tensor
    return ops.gspmm(g, "copy_rhs", red, None, _eval_message(g, mfunc))


def update_all(g, message_func, reduce_func, apply_node_func=None):
    if g.number_of_edges() == 0 and not isinstance(message_func, fn.BuiltinFunction):
        return
    if not isinstance(reduce_func, fn.SimpleReduceFunction):
        raise DGLError("update_all: only builtin reduce functions (fn.sum/mean/max/min) are supported; "
                       "got %r" % (reduce_func,))
    if isinstance(message_func, fn.BuiltinFunction):
        if isinstance(message_func, fn.BuiltinFunction) and reduce_func.msg_field != message_func.out_field:
            raise DGLError("Cannot find message field %r produced by %s" % (reduce_func.msg_field, message_func.name))
        out = _builtin_spmm(g, message_func, reduce_func)
    else:
        msgs = message_func(EdgeBatch(g))
        if not isinstance(msgs, dict) or reduce_func.msg_field not in msgs:
            raise DGLError("message UDF must return a dict containing field %r" % (reduce_func.msg_field,))
        out = ops.gspmm(g, "copy_rhs", reduce_func.name, None, msgs[reduce_func.msg_field])
    g.dstdata[reduce_func.out_field] = out
    if apply_node_func is not None:
        ret = apply_node_func(NodeBatch(g, g.dstdata))
        for k, v in ret.items():
            g.dstdata[k] = v


def apply_edges(g, func):
    if isinstance(func, fn.BuiltinFunction):
        g.edata[func.out_field] = _eval_message(g, func)
        return
    ret = func(EdgeBatch(g))
    if not isinstance(ret, dict):
        raise DGLError("edge UDF must return a dict of edge features")
    for k, v in ret.items():
        g.edata[k] = v
