"""CPU suite: the C-ABI library loads and exports every symbol include/mi355x_graph.h declares.
No compute calls (there is no GPU here) except the host-pointer COO->CSR, which must be bit-exact."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import mi355x_graph as mg
from mi355x_graph import _lib


def header_symbols():
    text = open(_lib.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mgx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = header_symbols()
    assert len(syms) >= 15
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(handle, s), "libmi355x_graph.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, "ctypes SIGNATURES and the header disagree"
    assert _lib.lib().mgx_abi_version() == 35


def test_bad_arguments_raise_not_crash():
    L = _lib.lib()
    st = L.mgx_spmm_csr(None, None, 0, 0, None, None, 1, 1, 1, None, None, None, None, None, None, None, None, 0, None)
    assert st == 1 and b"csr is NULL" in L.mgx_last_error()
    with pytest.raises(mg.DGLError):
        _lib.check(st)
    c = _lib.MgxCsr(1, 1, 0, None, None, None, 16, 0)
    assert L.mgx_spmm_csr(ctypes.byref(c), None, 0, 0, None, None, 1, 1, 1, None, None, None, None, None, None, None, None, 0, None) == 1
    assert L.mgx_sddmm_coo(1, 1, 1, None, None, 32, 99, None, None, 0, 2, 1, 1, 1, 1, None, None, None, None) == 1


@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
def test_host_coo_to_csr_bit_exact(oracle, dtype):
    rng = np.random.default_rng(0)
    for n, nnz in [(1, 0), (5, 3), (1000, 200000), (300, 70000)]:
        row = rng.integers(0, n, nnz)
        col = rng.integers(0, n, nnz)
        v = mg.sparse.coo_to_csr_host(n, n, torch.from_numpy(row).to(dtype), torch.from_numpy(col).to(dtype))
        ip, ix, ei = oracle.coo_to_csr(n, row, col)
        assert np.array_equal(v.indptr.numpy(), ip) and np.array_equal(v.indices.numpy(), ix)
        assert np.array_equal(v.eids.numpy(), ei)


def test_cpu_message_passing_fails_loudly():
    g = mg.graph((torch.tensor([0, 1]), torch.tensor([1, 0])))
    with pytest.raises(mg.DGLError, match="MI355X"):
        mg.ops.gspmm(g, "copy_lhs", "sum", torch.rand(2, 4), None)
