from mi355x_graph.nn import GATConv, SAGEConv, GraphConv, AvgPooling, SumPooling, MaxPooling, Linear, HeteroGraphConv, GINConv, BatchNorm1d  # noqa: F401
from mi355x_graph.ops import edge_softmax  # noqa: F401
from . import conv, glob  # noqa: F401
