from mi355x_graph.ops import *  # noqa: F401,F403
from mi355x_graph.ops import __all__  # noqa: F401
