from mi355x_graph.datasets import (load_data, RedditDataset, CoraGraphDataset, CiteseerGraphDataset,  # noqa: F401
                                   PubmedGraphDataset, LegacyTUDataset)
from . import utils  # noqa: F401
