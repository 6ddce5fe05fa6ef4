from mi355x_graph.nn import AvgPooling, SumPooling, MaxPooling  # noqa: F401
