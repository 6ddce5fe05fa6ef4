from mi355x_graph.sampling import sample_neighbors  # noqa: F401
