from mi355x_graph.utils import expand_as_pair, GraphedStep  # noqa: F401
