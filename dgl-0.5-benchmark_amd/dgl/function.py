from mi355x_graph.function import *  # noqa: F401,F403
from mi355x_graph.function import __all__  # noqa: F401
