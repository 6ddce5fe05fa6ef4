"""On-disk dataset formats: when the real files are supplied, they replace the synthetic
stand-ins of datasets.py, which makes the accuracy columns of the reference README checkable.

  DGL RedditDataset layout   <root>/reddit/reddit_data.npz  (feature, label, node_types: 1 train / 2 val / 3 test)
                             <root>/reddit/reddit_graph.npz (scipy.sparse.save_npz COO/CSR adjacency)
                             -- what dgl.data.RedditDataset() downloads (kernel/utils.py:52, main_dgl_reddit_sage.py:188)
  OGB node-property layout   <root>/ogbn_<name>/raw/{edge.csv.gz, node-feat.csv.gz, node-label.csv.gz, num-node-list.csv.gz}
                             <root>/ogbn_<name>/split/<scheme>/{train,valid,test}.csv.gz
                             -- what DglNodePropPredDataset(name='ogbn-products') unpacks (main_dgl_product_sage.py:152-156)
  plain npz                  keys src, dst (or edge_index [2,E]), num_nodes, feat, label and train_idx/valid_idx/test_idx
                             or train_mask/val_mask/test_mask

Pure file parsing (numpy / pandas / scipy) on the host; nothing here touches the message-passing library.
Set MGX_DATA_ROOT=<root> to make dgl.data.load_data / RedditDataset and the ogb test shim pick the files up.
"""
import glob
import os

import numpy as np
import torch

from .graph import graph as make_graph


class DiskNodeData(object):
    """Same attribute surface as datasets.NodeData (features / labels / masks / graph / num_classes)."""

    def __init__(self, name, src, dst, num_nodes, feat, label, train_idx, val_idx, test_idx):
        self.name, self.num_nodes = name, int(num_nodes)
        self.features = torch.as_tensor(np.ascontiguousarray(feat), dtype=torch.float32)
        label = np.asarray(label)
        self.labels = torch.as_tensor(label.reshape(label.shape[0], -1)[:, 0] if label.ndim > 1 else label).long()
        self.num_classes = self.num_labels = int(self.labels.max().item()) + 1 if self.labels.numel() else 0
        self.graph = make_graph((torch.as_tensor(np.asarray(src)).long(), torch.as_tensor(np.asarray(dst)).long()),
                                num_nodes=self.num_nodes)

        def mask(idx):
            m = torch.zeros(self.num_nodes, dtype=torch.bool)
            m[torch.as_tensor(np.asarray(idx)).long()] = True
            return m

        self.train_mask, self.val_mask, self.test_mask = mask(train_idx), mask(val_idx), mask(test_idx)
        self.split_idx = {"train": torch.as_tensor(np.asarray(train_idx)).long(), "valid": torch.as_tensor(np.asarray(val_idx)).long(),
                          "test": torch.as_tensor(np.asarray(test_idx)).long()}
        g = self.graph
        g.ndata["feat"], g.ndata["label"] = self.features, self.labels
        g.ndata["train_mask"], g.ndata["val_mask"], g.ndata["test_mask"] = self.train_mask, self.val_mask, self.test_mask

    def __getitem__(self, i):
        assert i == 0
        return self.graph

    def __len__(self):
        return 1


def load_reddit_npz(folder):
    import scipy.sparse as sp
    data = np.load(os.path.join(folder, "reddit_data.npz"))
    adj = sp.load_npz(os.path.join(folder, "reddit_graph.npz")).tocoo()
    types = data["node_types"]
    return DiskNodeData("reddit", adj.row, adj.col, adj.shape[0], data["feature"], data["label"],
                        np.nonzero(types == 1)[0], np.nonzero(types == 2)[0], np.nonzero(types == 3)[0])


def _read_csv(path, dtype):
    import pandas as pd
    return pd.read_csv(path, compression="gzip" if path.endswith(".gz") else None, header=None).values.astype(dtype)


def load_ogb_raw(folder, bidirected=False):
    """OGB raw CSV layout.  `bidirected`: ogbn-products ships one direction per undirected edge and OGB's DGL
    loader adds the reverse (README.md:25 counts 61.86 M, the DGL graph holds 123.7 M)."""
    raw = os.path.join(folder, "raw")
    edge = _read_csv(os.path.join(raw, "edge.csv.gz"), np.int64)
    feat = _read_csv(os.path.join(raw, "node-feat.csv.gz"), np.float32)
    label = _read_csv(os.path.join(raw, "node-label.csv.gz"), np.float32)
    n = int(_read_csv(os.path.join(raw, "num-node-list.csv.gz"), np.int64)[0, 0])
    split_dirs = sorted(glob.glob(os.path.join(folder, "split", "*")))
    if not split_dirs:
        raise FileNotFoundError("no split/<scheme> directory under %s" % folder)
    sp_dir = split_dirs[0]
    tr, va, te = [_read_csv(os.path.join(sp_dir, k + ".csv.gz"), np.int64)[:, 0] for k in ("train", "valid", "test")]
    src, dst = edge[:, 0], edge[:, 1]
    if bidirected:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    return DiskNodeData(os.path.basename(folder), src, dst, n, feat, np.nan_to_num(label).astype(np.int64), tr, va, te)


def load_npz(path):
    z = np.load(path)
    if "edge_index" in z:
        src, dst = z["edge_index"][0], z["edge_index"][1]
    else:
        src, dst = z["src"], z["dst"]
    n = int(z["num_nodes"]) if "num_nodes" in z else int(max(src.max(), dst.max())) + 1

    def idx(key_idx, key_mask):
        if key_idx in z:
            return z[key_idx]
        if key_mask in z:
            return np.nonzero(z[key_mask])[0]
        return np.zeros(0, np.int64)

    return DiskNodeData(os.path.splitext(os.path.basename(path))[0], src, dst, n, z["feat"], z["label"],
                        idx("train_idx", "train_mask"), idx("valid_idx", "val_mask"), idx("test_idx", "test_mask"))


def find_dataset(name):
    """Looks under $MGX_DATA_ROOT for `name` in any supported layout; None when nothing is there."""
    root = os.environ.get("MGX_DATA_ROOT")
    if not root:
        return None
    key = name.lower().replace("-", "_")
    if key.startswith("reddit") and os.path.exists(os.path.join(root, "reddit", "reddit_data.npz")):
        return load_reddit_npz(os.path.join(root, "reddit"))
    for cand in (key, "ogbn_" + key):
        folder = os.path.join(root, cand)
        if os.path.exists(os.path.join(folder, "raw", "edge.csv.gz")):
            return load_ogb_raw(folder, bidirected=(cand.endswith("products") or cand.endswith("proteins")))
        if os.path.exists(folder + ".npz"):
            return load_npz(folder + ".npz")
    return None


# ----------------------------------------------------------------------------- dgl.data.utils file helpers
def get_download_dir():
    """dgl.data.utils.get_download_dir (gcmc_dgl/data.py:19): $DGL_DOWNLOAD_DIR, else $MGX_DATA_ROOT, else ~/.dgl."""
    d = os.environ.get("DGL_DOWNLOAD_DIR") or os.environ.get("MGX_DATA_ROOT") or os.path.join(os.path.expanduser("~"), ".dgl")
    os.makedirs(d, exist_ok=True)
    return d


def download(url, path=None, overwrite=False, sha1_hash=None, retries=5, verify_ssl=True, log=True):
    """dgl.data.utils.download.  The GPU boxes have no network: an archive (or its extracted directory) that is already in
    place is accepted, anything else is an error that says where to put the file."""
    fname = path if path is not None else url.split("/")[-1]
    if os.path.isdir(fname):
        fname = os.path.join(fname, url.split("/")[-1])
    extracted = fname[:-4] if fname.endswith(".zip") else fname
    if os.path.exists(fname) or os.path.isdir(extracted):
        return fname
    raise IOError("cannot download %s: no network on this machine; place the archive at %s or its extracted contents at %s"
                  % (url, fname, extracted))


def extract_archive(file, target_dir, overwrite=False):
    """dgl.data.utils.extract_archive: .zip / .tar(.gz) / .gz; a target directory that already holds files is kept."""
    if os.path.isdir(target_dir) and os.listdir(target_dir) and not overwrite:
        return
    if not os.path.exists(file):
        raise IOError("archive %s not found and %s is empty" % (file, target_dir))
    os.makedirs(target_dir, exist_ok=True)
    if file.endswith(".zip"):
        import zipfile
        with zipfile.ZipFile(file) as z:
            z.extractall(target_dir)
    elif file.endswith((".tar.gz", ".tgz", ".tar")):
        import tarfile
        with tarfile.open(file) as t:
            t.extractall(target_dir)
    elif file.endswith(".gz"):
        import gzip
        import shutil
        with gzip.open(file, "rb") as src, open(os.path.join(target_dir, os.path.basename(file)[:-3]), "wb") as dst:
            shutil.copyfileobj(src, dst)
    else:
        raise IOError("Unrecognized file type: " + file)
