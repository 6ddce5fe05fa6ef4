"""dgl.utils helpers used by the scripts (expand_as_pair, main_dgl_product_sage.py:10,21)."""


def expand_as_pair(input_, g=None):
    if isinstance(input_, tuple):
        return input_
    if g is not None and getattr(g, "is_block", False):
        if isinstance(input_, dict):
            raise TypeError("heterograph inputs are not supported by this backend")
        return input_, input_[:g.number_of_dst_nodes()]
    return input_, input_
