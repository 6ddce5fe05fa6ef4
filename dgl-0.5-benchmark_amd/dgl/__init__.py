"""Drop-in `dgl` import for the benchmark scripts (import dgl / dgl.function / dgl.ops / dgl.nn...),
backed by mi355x_graph (HIP, gfx950).  Only the symbols the hot path's callers touch exist
(SURVEY Appendix B); everything computes on MI355X through include/mi355x_graph.h.
Put the directory that contains this package on PYTHONPATH to use it."""
from mi355x_graph import DGLError, DGLGraph, DGLHeteroGraph, graph, create_block, ALL  # noqa: F401
from mi355x_graph import heterograph, bipartite, hetero_from_relations  # noqa: F401
from mi355x_graph.transform import (to_bidirected, add_self_loop, remove_self_loop, add_reverse_edges,  # noqa: F401
                                    reverse, from_networkx, from_scipy, batch, unbatch)
from mi355x_graph import function, ops  # noqa: F401
from . import nn, data, dataloading, utils, sampling, transform, backend  # noqa: F401
from mi355x_graph.sampling import to_block, NID, EID  # noqa: F401
from mi355x_graph.ops import edge_softmax  # noqa: F401

__version__ = "0.6.1+mi355x"
