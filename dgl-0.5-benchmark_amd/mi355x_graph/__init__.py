"""mi355x_graph -- MI355X-native sparse message passing behind DGL's operator surface.

Host-side mirror of the reference interface for the hot path (update_all / apply_edges /
dgl.ops.gspmm / gsddmm / edge_softmax); all arithmetic runs in csrc/libmi355x_graph.so (HIP, gfx950)
through the C ABI of include/mi355x_graph.h.
"""
from ._lib import DGLError, LIB_PATH  # noqa: F401
from . import function, ops, sparse  # noqa: F401
from .graph import DGLGraph, GraphIndex, graph, create_block, ALL  # noqa: F401
from .heterograph import DGLHeteroGraph, heterograph, bipartite, hetero_from_relations  # noqa: F401

__version__ = "0.1.0"

# CPU (OpenMP) variants (csrc/cpu_ops.cpp, include/mi355x_graph_cpu.h): opt-in, never a fallback -- see cpu_backend.py
from .cpu_backend import enable_cpu_backend, cpu_backend_enabled  # noqa: E402,F401
import os as _os  # noqa: E402
if _os.environ.get("MGX_CPU_BACKEND", "0") == "1":
    enable_cpu_backend(True)
