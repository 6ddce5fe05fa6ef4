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
# 25 ms of a 46 ms products epoch (bias-gradient reduction 19 ms).  mi355x_graph.utils.accelerate_linear() routes tall fp32 device
# matrices through this package's column-sum / X^T Y kernels in the backward (same forward).  It replaces torch.nn.functional.linear
# process-wide, so it is OPT-IN: a launcher calls it, or the user of an unmodified script sets MGX_ACCELERATE_LINEAR=1 -- the import
# then says what it did, once, on stderr.  Without either, `import dgl` leaves PyTorch alone.
import os as _os
if _os.environ.get("MGX_ACCELERATE_LINEAR", "0") == "1":
    import sys as _sys
    from mi355x_graph.utils import accelerate_linear as _accelerate_linear
    if not _accelerate_linear(True):
        _sys.stderr.write("dgl (mi355x_graph): MGX_ACCELERATE_LINEAR=1 -- torch.nn.functional.linear now takes this package's backward "
                          "for fp32 device matrices of >= 65536 rows (mi355x_graph.utils.accelerate_linear(False) undoes it)\n")
