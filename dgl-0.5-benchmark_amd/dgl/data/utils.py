from mi355x_graph.datasets import Subset  # noqa: F401
