from mi355x_graph.sampling import sample_neighbors  # noqa: F401
# cluster-gat/dgl/main.py:88-89 reaches the block sampler and its loader through dgl.sampling (the 0.5 pre-release spelling)
from mi355x_graph.sampling import MultiLayerNeighborSampler, NodeDataLoader  # noqa: F401,E402
