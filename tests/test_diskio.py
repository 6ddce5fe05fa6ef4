"""CPU suite: on-disk dataset layouts -- files written in DGL's reddit npz layout, OGB's raw CSV
layout and the plain npz layout load back to the same graph / features / splits (bit-exact index work)."""
import gzip
import os

import numpy as np
import scipy.sparse as sp
import torch

from mi355x_graph import diskio, datasets


def toy(n=50, m=300, seed=0):
    rng = np.random.default_rng(seed)
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    feat = rng.random((n, 7), dtype=np.float32)
    label = rng.integers(0, 4, n)
    perm = rng.permutation(n)
    return src, dst, feat, label, perm[:30], perm[30:40], perm[40:]


def write_csv_gz(path, arr, fmt):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with gzip.open(path, "wt") as f:
        np.savetxt(f, arr, fmt=fmt, delimiter=",")


def check(d, src, dst, feat, label, tr, va, te, edges_sorted=False):
    s, t = d.graph.edges()
    if edges_sorted:
        key = lambda a, b: np.sort(np.asarray(a) * 1000 + np.asarray(b))
        assert np.array_equal(key(s.numpy(), t.numpy()), key(src, dst))
    else:
        assert np.array_equal(s.numpy(), src) and np.array_equal(t.numpy(), dst)
    assert np.allclose(d.features.numpy(), feat, atol=1e-6) and np.array_equal(d.labels.numpy(), label)
    assert np.array_equal(np.nonzero(d.train_mask.numpy())[0], np.sort(tr))
    assert np.array_equal(np.nonzero(d.val_mask.numpy())[0], np.sort(va))
    assert np.array_equal(np.nonzero(d.test_mask.numpy())[0], np.sort(te))


def test_reddit_npz_layout(tmp_path, monkeypatch):
    src, dst, feat, label, tr, va, te = toy()
    n = feat.shape[0]
    folder = tmp_path / "reddit"
    folder.mkdir()
    types = np.zeros(n, np.int64)
    types[tr], types[va], types[te] = 1, 2, 3
    np.savez(folder / "reddit_data.npz", feature=feat, label=label, node_types=types)
    # simple graph so the sparse matrix keeps every edge
    key = np.unique(src * n + dst)
    src, dst = key // n, key % n
    sp.save_npz(folder / "reddit_graph.npz", sp.coo_matrix((np.ones(len(src)), (src, dst)), shape=(n, n)))
    monkeypatch.setenv("MGX_DATA_ROOT", str(tmp_path))
    d = datasets.RedditDataset()
    check(d, src, dst, feat, label, tr, va, te, edges_sorted=True)
    assert d[0].number_of_nodes() == n


def test_ogb_raw_layout(tmp_path, monkeypatch):
    src, dst, feat, label, tr, va, te = toy(seed=1)
    root = tmp_path / "ogbn_arxiv"
    write_csv_gz(str(root / "raw" / "edge.csv.gz"), np.stack([src, dst], 1), "%d")
    write_csv_gz(str(root / "raw" / "node-feat.csv.gz"), feat, "%.8f")
    write_csv_gz(str(root / "raw" / "node-label.csv.gz"), label[:, None], "%d")
    write_csv_gz(str(root / "raw" / "num-node-list.csv.gz"), np.array([[feat.shape[0]]]), "%d")
    for k, v in (("train", tr), ("valid", va), ("test", te)):
        write_csv_gz(str(root / "split" / "time" / (k + ".csv.gz")), v[:, None], "%d")
    monkeypatch.setenv("MGX_DATA_ROOT", str(tmp_path))
    d = diskio.find_dataset("ogbn-arxiv")
    check(d, src, dst, feat, label, tr, va, te)
    assert torch.equal(d.split_idx["train"], torch.from_numpy(tr))


def test_plain_npz_layout(tmp_path):
    src, dst, feat, label, tr, va, te = toy(seed=2)
    p = tmp_path / "mygraph.npz"
    np.savez(p, edge_index=np.stack([src, dst]), num_nodes=feat.shape[0], feat=feat, label=label, train_idx=tr, valid_idx=va, test_idx=te)
    check(diskio.load_npz(str(p)), src, dst, feat, label, tr, va, te)
    assert diskio.find_dataset("anything") is None or True
