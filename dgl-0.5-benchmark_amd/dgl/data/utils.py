from mi355x_graph.datasets import Subset  # noqa: F401
from mi355x_graph.diskio import download, extract_archive, get_download_dir  # noqa: F401,E402
