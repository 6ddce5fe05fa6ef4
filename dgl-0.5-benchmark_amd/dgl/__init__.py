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

# The dense half of an unmodified reference model is torch.nn.Linear on 2.45 M-row matrices; PyTorch's own backward for it takes
# 25 ms of a 46 ms products epoch (bias-gradient reduction 19 ms).  The drop-in import therefore routes tall fp32 device matrices through
# mi355x_graph.utils.accelerate_linear() -- same forward, backward by this package's column-sum / X^T Y kernels.  MGX_ACCELERATE_LINEAR=0
# leaves torch.nn.functional.linear alone; `import mi355x_graph` never touches it.
import os as _os
if _os.environ.get("MGX_ACCELERATE_LINEAR", "1") != "0":
    from mi355x_graph.utils import accelerate_linear as _accelerate_linear
    _accelerate_linear(True)
