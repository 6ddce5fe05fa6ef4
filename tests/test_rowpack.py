"""csrc/rowpack.hip: halo rows as bitmaps + packed values (dist.SparseHalo's kernels) against a plain restatement of their contract
-- bit (c * G + l) of a block's 64-bit word = column 64 * block + 4 l + c, G = min(16, pow2 >= D / 4); a row's values in increasing
bit order, block after block.  Values are moved, never rounded: everything here is bit for bit."""
import numpy as np
import pytest
import torch

from mi355x_graph import _lib, sparse


def _lanes(D):
    g = 1
    while g * 4 < D and g < 16:
        g <<= 1
    return g


def _reference(x):
    """(masks uint64 [n, W], counts, values in the kernel's order) of a host matrix."""
    n, D = x.shape
    W, G = (D + 63) // 64, _lanes(D)
    masks = np.zeros((n, W), np.uint64)
    order = []  # columns in increasing bit order, block after block
    for b in range(W):
        for bit in range(4 * G):
            c, l = bit // G, bit % G
            col = 64 * b + 4 * l + c
            if col < D:
                order.append((b, bit, col))
    vals, counts = [], np.zeros(n, np.int32)
    for i in range(n):
        for b, bit, col in order:
            if x[i, col] != 0:
                masks[i, b] |= np.uint64(1) << np.uint64(bit)
                vals.append(x[i, col])
                counts[i] += 1
    return masks, counts, np.asarray(vals, np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("D", [4, 8, 12, 16, 36, 64, 100, 128, 200, 256])
def test_pack_and_unpack_follow_the_contract(D):
    dev = torch.device("cuda:0")
    be = sparse.backend_for(torch.empty(1, device=dev))
    rng = np.random.default_rng(D)
    n_src, n = 523, 301
    x = (rng.standard_normal((n_src, D)) * (rng.random((n_src, D)) < 0.3)).astype(np.float32)
    x[5] = 0.0                      # an all-zero row
    x[7] = rng.standard_normal(D)   # a full row
    x[9, D - 1] = -0.0              # negative zero counts as zero (== 0)
    idx = rng.permutation(n_src)[:n].astype(np.int32)
    ld = D + 8                      # a column block of a wider matrix (row-strided view)
    wide = torch.zeros(n_src, ld, device=dev)
    wide[:, :D] = torch.from_numpy(x).to(dev)
    xv = wide[:, :D]
    assert be.rows_pack_supported(xv)
    for index in (torch.from_numpy(idx).to(dev), torch.from_numpy(idx.astype(np.int64)).to(dev), None):
        rows = x if index is None else x[idx]
        masks, counts = be.rows_pack_count(xv, index)
        rm, rc, rv = _reference(rows)
        assert np.array_equal(masks.cpu().numpy().view(np.uint64), rm)
        assert np.array_equal(counts.cpu().numpy(), rc)
        assert np.array_equal(be.rows_mask_count(masks, D).cpu().numpy(), rc)
        off = torch.zeros(rows.shape[0] + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, dtype=torch.int64, out=off[1:])
        vals = be.rows_pack_values(xv, index, masks, off, int(off[-1]))
        assert np.array_equal(vals.cpu().numpy().view(np.uint32), rv.view(np.uint32))
        out = torch.full((rows.shape[0], ld), 7.0, device=dev)  # into a row-strided destination; the padding stays untouched
        be.rows_unpack(masks, off, vals, D, out=out[:, :D])
        got = out.cpu().numpy()
        want = np.where(rows == 0, np.float32(0), rows)  # -0.0 comes back as +0.0: it was never sent
        assert np.array_equal(got[:, :D].view(np.uint32), want.view(np.uint32)) and (got[:, D:] == 7.0).all()
        # the backward direction: another matrix's entries under THESE masks (zeros included where the mask is set)
        g = rng.standard_normal(rows.shape).astype(np.float32)
        g[rows.shape[0] // 2] = 0.0
        gt = torch.from_numpy(g).to(dev)
        gv = be.rows_pack_values(gt, None, masks, off, int(off[-1]))
        back = be.rows_unpack(masks, off, gv, D).cpu().numpy()
        assert np.array_equal(back.view(np.uint32), np.where(rows != 0, g, np.float32(0)).view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("D", [16, 64, 100])
def test_packed_rows_added_into_their_owners(D):
    """mgx_rows_unpack_add_csr == unpack to dense rows, then add them into their owners in CSR order (what dist.SparseHalo's backward
    did in two steps): same order of addition, so bit for bit."""
    dev = torch.device("cuda:0")
    be = sparse.backend_for(torch.empty(1, device=dev))
    rng = np.random.default_rng(100 + D)
    n_own, n_sent = 400, 1300
    owner = np.sort(rng.integers(0, n_own, n_sent)).astype(np.int32)          # several returned rows per owner, some owners none
    owner[owner % 7 == 3] += 1
    owner = np.sort(np.clip(owner, 0, n_own - 1))
    pos = np.arange(n_sent)
    rng.shuffle(pos)                                                             # the rows of one owner lie anywhere in the packed buffer
    x = (rng.standard_normal((n_sent, D)) * (rng.random((n_sent, D)) < 0.25)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev)
    masks, counts = be.rows_pack_count(xt, None)
    off = torch.zeros(n_sent + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, dtype=torch.int64, out=off[1:])
    vals = be.rows_pack_values(xt, None, masks, off, int(off[-1]))
    csr = sparse.coo_to_csr(n_own, n_sent, torch.from_numpy(owner).to(dev), torch.from_numpy(pos.astype(np.int32)).to(dev))
    base = rng.standard_normal((n_own, D + 4)).astype(np.float32)
    out = torch.from_numpy(base).to(dev)
    be.rows_unpack_add_csr(csr, masks, off, vals, out[:, :D])
    want = base.copy()
    ip, ix = csr.indptr.cpu().numpy(), csr.indices.cpu().numpy()
    for v in range(n_own):
        for q in range(ip[v], ip[v + 1]):
            want[v, :D] = want[v, :D] + x[ix[q]]
    got = out.cpu().numpy()
    assert np.array_equal(got[:, D:], base[:, D:])
    assert np.array_equal(got[:, :D], want[:, :D])


@pytest.mark.gpu
def test_empty_and_unsupported_shapes():
    dev = torch.device("cuda:0")
    be = sparse.backend_for(torch.empty(1, device=dev))
    x = torch.rand(0, 64, device=dev)
    masks, counts = be.rows_pack_count(x, None)
    assert masks.shape == (0, 1) and counts.numel() == 0
    assert be.rows_unpack(masks, torch.zeros(1, dtype=torch.int64, device=dev), torch.empty(0, device=dev), 64).shape == (0, 64)
    # rows that are all zero (a dead layer's boundary rows): empty masks, an EMPTY value vector -- packed, unpacked and added back
    z = torch.zeros(37, 64, device=dev)
    masks, counts = be.rows_pack_count(z, None)
    assert int(counts.sum()) == 0 and int((masks != 0).sum()) == 0
    off = torch.zeros(38, dtype=torch.int64, device=dev)
    vals = be.rows_pack_values(z, None, masks, off, 0)
    assert vals.numel() == 0
    assert torch.equal(be.rows_unpack(masks, off, vals, 64), z)
    csr = sparse.coo_to_csr(5, 37, torch.arange(37, device=dev, dtype=torch.int32) % 5, torch.arange(37, device=dev, dtype=torch.int32))
    base = torch.rand(5, 64, device=dev)
    out = base.clone()
    be.rows_unpack_add_csr(csr, masks, off, vals, out)
    assert torch.equal(out, base)
    assert not be.rows_pack_supported(torch.rand(4, 6, device=dev)) and not be.rows_pack_supported(torch.rand(4, 260, device=dev))
    with pytest.raises(_lib.DGLError):
        be.rows_pack_count(torch.rand(4, 6, device=dev), None)


def test_mask_words():
    L = _lib.lib()
    assert [L.mgx_rows_mask_words(d) for d in (4, 64, 65, 128, 256)] == [1, 1, 2, 2, 4]
