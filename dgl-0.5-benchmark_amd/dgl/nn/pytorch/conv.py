from mi355x_graph.nn import GATConv, SAGEConv, GraphConv  # noqa: F401
