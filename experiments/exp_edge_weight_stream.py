"""Round 4: the backward's 1/deg factor as a per-EDGE weight streamed with the ids (u_mul_e with (E, 1) weights in CSR order) against
the stand-alone N x 64 scaling pass + copy_u (what the epoch runs), products D = 64, accumulate form.  (src_scale -- a per-edge GATHER
of the factor -- was +0.86 ms: profiles/r04_src_scale.txt.)"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dgl-0.5-benchmark_amd"))
import torch  # noqa: E402
import kernel_bench  # noqa: E402
from mi355x_graph import _lib, sparse  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=12):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]


g = kernel_bench.get_graph("products", dev, 1.0).int()
csr = g._index.csr()          # the reversed graph's in-CSR: rows = sources of the forward graph
csc = g._index.csc()
n, D = g.num_nodes(), 64
gen = torch.Generator(device=dev).manual_seed(1)
dn = torch.rand(n, D, device=dev, generator=gen)
acc = torch.zeros(n, D, device=dev)
inv = csc.inv_degrees()
# weight of reversed-CSR position p: 1 / deg(neighbour) -- by EDGE ID, as the operator addresses edge features
w = torch.empty(csr.nnz, 1, device=dev)
eids = csr.eids.long() if csr.eids is not None else torch.arange(csr.nnz, device=dev)
w[eids, 0] = inv[csr.indices.long()]
scaled = torch.empty_like(dn)


def two_steps():
    torch.mul(dn, inv.view(-1, 1), out=scaled)
    sparse.gspmm_raw(csr, "copy_lhs", "sum", scaled, None, accumulate_into=acc)


t_pass = timed(lambda: torch.mul(dn, inv.view(-1, 1), out=scaled))
t_two = timed(two_steps)
k_two = _lib.lib().mgx_last_spmm_kernel().decode()
t_mul = timed(lambda: sparse.gspmm_raw(csr, "mul", "sum", dn, w, accumulate_into=acc))
k_mul = _lib.lib().mgx_last_spmm_kernel().decode()
a1 = torch.zeros(n, D, device=dev)
a2 = torch.zeros(n, D, device=dev)
torch.mul(dn, inv.view(-1, 1), out=scaled)
sparse.gspmm_raw(csr, "copy_lhs", "sum", scaled, None, accumulate_into=a1)
sparse.gspmm_raw(csr, "mul", "sum", dn, w, accumulate_into=a2)
# the same with the weights in CSR POSITION order (a view without edge ids: position p reads w[p] -- a sequential stream beside the ids)
pos = sparse.CsrView(csr.num_rows, csr.num_cols, csr.indptr, csr.indices, None)
pos._plan, pos._row_order = csr.plan(), csr._row_order
w_pos = inv[csr.indices.long()].view(-1, 1).contiguous()
t_pos = timed(lambda: sparse.gspmm_raw(pos, "mul", "sum", dn, w_pos, accumulate_into=acc))
k_pos = _lib.lib().mgx_last_spmm_kernel().decode()
a3 = torch.zeros(n, D, device=dev)
sparse.gspmm_raw(pos, "mul", "sum", dn, w_pos, accumulate_into=a3)
print("weights in position order (%s) %.4f ms; max rel diff %.2e" % (k_pos, t_pos, float((a1 - a3).abs().max() / a1.abs().max())))
print("scaling pass %.4f ms; pass + copy_u/sum (%s) %.4f ms; u_mul_e/sum with (E,1) weights (%s) %.4f ms; max rel diff %.2e"
      % (t_pass, k_two, t_two, k_mul, t_mul, float((a1 - a2).abs().max() / a1.abs().max())))
