from mi355x_graph.utils import expand_as_pair  # noqa: F401
