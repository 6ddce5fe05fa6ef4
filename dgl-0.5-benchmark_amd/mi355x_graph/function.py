"""dgl.function builtins: message and reduce descriptors.

Used as fn.copy_src('h','m'), fn.mean('m','neigh') (main_dgl_product_sage.py:62),
fn.u_mul_e('feat','weight','m') (main_dgl_proteins_rgcn_for.py:52), fn.u_dot_v('h','h','sr')
(link_prediction/gcmc_dgl/model.py:342), fn.copy_u / fn.sum (main_dgl_enzymes_gcn.py:37).
A descriptor only names fields; core.update_all / core.apply_edges map it to one g-SpMM / g-SDDMM.
"""
import sys

__all__ = ["copy_u", "copy_src", "copy_e", "copy_edge", "sum", "mean", "max", "min",
           "BinaryMessageFunction", "CopyMessageFunction", "SimpleReduceFunction"]

_TARGET_CODE = {"u": "u", "v": "v", "e": "e"}


class BuiltinFunction(object):
    pass


class BinaryMessageFunction(BuiltinFunction):
    """`out = lhs_target[lhs_field] <op> rhs_target[rhs_field]` per edge."""

    def __init__(self, binary_op, lhs, rhs, lhs_field, rhs_field, out_field):
        self.binary_op, self.lhs, self.rhs = binary_op, lhs, rhs
        self.lhs_field, self.rhs_field, self.out_field = lhs_field, rhs_field, out_field

    @property
    def name(self):
        return "%s_%s_%s" % (self.lhs, self.binary_op, self.rhs)


class CopyMessageFunction(BuiltinFunction):
    def __init__(self, target, in_field, out_field):
        self.target, self.in_field, self.out_field = target, in_field, out_field

    @property
    def name(self):
        return "copy_%s" % self.target


class SimpleReduceFunction(BuiltinFunction):
    def __init__(self, name, msg_field, out_field):
        self.name, self.msg_field, self.out_field = name, msg_field, out_field


def copy_u(u, out):
    return CopyMessageFunction("u", u, out)


def copy_src(src, out):
    """Deprecated alias of copy_u used throughout the scripts."""
    return copy_u(src, out)


def copy_e(e, out):
    return CopyMessageFunction("e", e, out)


def copy_edge(edge, out):
    return copy_e(edge, out)


def sum(msg, out):  # noqa: A001 - DGL's name
    return SimpleReduceFunction("sum", msg, out)


def mean(msg, out):
    return SimpleReduceFunction("mean", msg, out)


def max(msg, out):  # noqa: A001
    return SimpleReduceFunction("max", msg, out)


def min(msg, out):  # noqa: A001
    return SimpleReduceFunction("min", msg, out)


def _make_binary(lhs, op, rhs):
    def func(lhs_field, rhs_field, out):
        return BinaryMessageFunction(op, lhs, rhs, lhs_field, rhs_field, out)
    func.__name__ = "%s_%s_%s" % (lhs, op, rhs)
    func.__doc__ = "Builtin message function: %s[lhs_field] %s %s[rhs_field]." % (lhs, op, rhs)
    return func


_mod = sys.modules[__name__]
for _l in ("u", "v", "e"):
    for _r in ("u", "v", "e"):
        if _l == _r:
            continue
        for _op in ("add", "sub", "mul", "div", "dot"):
            _f = _make_binary(_l, _op, _r)
            setattr(_mod, _f.__name__, _f)
            __all__.append(_f.__name__)
# src_mul_edge style aliases
src_mul_edge = getattr(_mod, "u_mul_e")
