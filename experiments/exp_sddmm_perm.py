"""g-SDDMM `add` on graphs whose edge list is in RANDOM order (the stand-ins) -- the COO walk in edge-id order
against the lean kernel over the in-CSR's order with the edge id as the output row (mgx_sddmm_coo_perm) and the older generic CSR
body (mgx_sddmm_csr).  Same outputs bit for bit.  One process per MGX_SDDMM_WALK setting is NOT needed: the backend is called directly.

  python experiments/exp_sddmm_perm.py > gpurun_out/r05_sddmm_perm.txt
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import torch  # noqa: E402

import kernel_bench as kb  # noqa: E402
from mi355x_graph import _lib, sparse  # noqa: E402
from mi355x_graph._lib import OP, TARGET  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


for name in ("reddit", "reddit-small", "proteins"):
    g = kb.get_graph(name, dev, 1.0).int().to(dev)
    idx = g._index
    src, dst = idx.coo()
    csc = idx.csc()
    n_src, n_dst, nnz = idx.num_src, idx.num_dst, idx.num_edges()
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    print("# ---- %s: N = %d, E = %d, edge list in generated (random) order" % (name, n_dst, nnz))
    print("%5s %12s %12s %12s   %s" % ("D", "coo ms", "csr-order ms", "old csr ms", "frac of 8 TB/s: coo / csr-order"))
    for D in (16, 64, 128):
        if nnz * D * 4 > 40e9:
            continue
        u = torch.rand(n_src, D, device=dev)
        v = torch.rand(n_dst, D, device=dev)
        out0 = torch.empty(nnz, D, device=dev)
        out1 = torch.empty(nnz, D, device=dev)
        out2 = torch.empty(nnz, D, device=dev)

        def coo():
            _lib.check(L.mgx_sddmm_coo(n_src, n_dst, nnz, ptr(src), ptr(dst), 32, OP["add"], ptr(u), ptr(v), TARGET["u"], TARGET["v"], D, D, D, 1,
                                       None, None, ptr(out0), stream))

        def perm():
            _lib.check(L.mgx_sddmm_coo_perm(n_src, n_dst, nnz, ptr(csc.indices), ptr(csc.row_of_position()), ptr(csc.eids), 32, OP["add"],
                                            ptr(u), ptr(v), TARGET["u"], TARGET["v"], D, ptr(out1), stream))

        plan = csc.plan()

        def old():
            _lib.check(L.mgx_sddmm_csr(ctypes.byref(csc.c_struct()), None if plan is None else ctypes.byref(plan.c_struct()), OP["add"], ptr(u),
                                       ptr(v), TARGET["u"], TARGET["v"], D, D, D, 1, None, None, ptr(out2), stream))
        t0, _, _ = kb.time_op(coo)
        t1, _, _ = kb.time_op(perm)
        t2, _, _ = kb.time_op(old)
        same1, same2 = torch.equal(out0, out1), torch.equal(out0, out2)
        if not (same1 and same2):
            bad = (out0 != out1).any(1)
            print("!! D = %d: csr-order walk %s (%d rows differ, first %s), old csr body %s" % (D, "same" if same1 else "DIFFERS", int(bad.sum()),
                  bad.nonzero()[:4].flatten().tolist(), "same" if same2 else "DIFFERS"))
        b = kb.sddmm_bytes(n_src, n_dst, nnz, D, "add")
        print("%5d %12.3f %12.3f %12.3f   %.3f / %.3f" % (D, t0 * 1e3, t1 * 1e3, t2 * 1e3, b / t0 / 8e12, b / t1 / 8e12))
        del u, v, out0, out1, out2
    del g, idx, src, dst, csc
    torch.cuda.empty_cache()
