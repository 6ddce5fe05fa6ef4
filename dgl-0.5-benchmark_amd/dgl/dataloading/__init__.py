from mi355x_graph.dataloading import GraphDataLoader  # noqa: F401
