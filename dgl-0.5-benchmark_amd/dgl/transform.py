from mi355x_graph.transform import (metis_partition, metis_partition_assignment, to_bidirected, add_self_loop,  # noqa: F401
                                    remove_self_loop, add_reverse_edges, reverse)
