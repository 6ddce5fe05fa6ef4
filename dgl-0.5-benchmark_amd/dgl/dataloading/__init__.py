from mi355x_graph.dataloading import GraphDataLoader  # noqa: F401
from mi355x_graph.sampling import (MultiLayerNeighborSampler, MultiLayerFullNeighborSampler,  # noqa: F401
                                   NodeDataLoader)
