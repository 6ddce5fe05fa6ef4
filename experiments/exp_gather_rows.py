"""Round 4: pack of a partition's boundary rows -- torch.index_select against mgx_gather_rows_strided (rows of 64 / 100 floats out of a
[n, 2K] buffer's left half and out of a dense matrix)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dgl-0.5-benchmark_amd"))
import torch  # noqa: E402
from mi355x_graph import sparse  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(4):
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


gen = torch.Generator(device=dev).manual_seed(0)
for n_own, n_send, K in ((310000, 997000, 64), (1231086, 1032722, 64), (310000, 997000, 100)):
    buf = torch.rand(n_own, 2 * K, device=dev, generator=gen)
    dense = buf[:, :K].contiguous()
    idx = torch.randint(0, n_own, (n_send,), device=dev, generator=gen, dtype=torch.int32)
    idx64 = idx.long()
    be = sparse.backend_for(buf)
    for name, x in (("left half of [n, 2K]", buf[:, :K]), ("dense [n, K]", dense)):
        a = torch.index_select(x, 0, idx64)
        b = be.gather_rows(x, idx)
        assert torch.equal(a, b)
        print("%8d rows of %3d floats out of %8d, %-22s index_select %.4f ms   mgx_gather_rows_strided %.4f ms" % (
            n_send, K, n_own, name, timed(lambda: torch.index_select(x, 0, idx64)), timed(lambda: be.gather_rows(x, idx))))
