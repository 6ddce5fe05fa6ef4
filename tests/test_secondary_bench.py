"""secondary_bench.py (the `secondary` block of the bench line): the record aggregation and SURVEY 8d's byte formulas on fake
HIP-event records -- the timed legs themselves need the GPU and run inside bench.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dgl-0.5-benchmark_amd"))
import secondary_bench as sb  # noqa: E402


class _Ev(object):
    def __init__(self, t):
        self.t = t

    def elapsed_time(self, other):
        return other.t - self.t


def _rec(ms, **kw):
    kw["start"], kw["end"] = _Ev(0.0), _Ev(ms)
    return kw


def test_dominant_call_and_survey_8d_bytes():
    N, E, D = 169343, 2315598, 256
    recs = [_rec(0.25, op="copy_lhs", reduce="mean", out_len=D, n_rows=N, n_cols=N, nnz=E, accumulate=False) for _ in range(6)]
    recs += [_rec(0.05, op="copy_lhs", reduce="mean", out_len=40, n_rows=N, n_cols=N, nnz=E, accumulate=False) for _ in range(3)]
    recs += [_rec(9.0, kernel="segment_reduce", reduce="mean", segments=4, rows=10, D=3)]  # never the "dominant" hot-path call
    r = sb.dominant_roofline(recs, steps=3)
    algo = 4 * (N + 1) + 4 * E + 4 * N * D + 4 * N * D  # SURVEY 8d worked number: 356.8 MB
    assert r["algorithmic_bytes_per_launch"] == algo and abs(algo / 1e6 - 356.1) < 1.0
    assert r["launches_timed"] == 6 and r["launches_per_step"] == 2.0 and abs(r["avg_launch_ms"] - 0.25) < 1e-9
    assert r["rows"] == N and r["nnz"] == E and "D=256" in r["kernel"]
    assert abs(r["achieved"] - algo / 0.25e-3 / 1e9) < 0.1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    assert abs(r["share_of_hot_path_device_time"] - 1.5 / 1.65) < 1e-3


def test_gat_and_sddmm_and_copy_e_formulas():
    n, E, H, F = 1000, 50000, 8, 16
    idx, nd, nh = 4 * (n + 1) + 4 * E, 4 * n * H * F, 4 * n * H
    fwd = sb._bytes_gat({"kernel": "gat_fwd", "n_src": n, "n_dst": n, "nnz": E, "H": H, "F": F})
    assert fwd == idx + 2 * nd + 6 * nh
    both = sb._bytes_gat({"kernel": "gat_bwd", "n_src": n, "n_dst": n, "nnz": E, "H": H, "F": F, "source_walk": True})
    dst_only = sb._bytes_gat({"kernel": "gat_bwd", "n_src": n, "n_dst": n, "nnz": E, "H": H, "F": F, "source_walk": False})
    assert dst_only == idx + 3 * nd + 7 * nh and both - dst_only == idx + 3 * nd + 6 * nh
    assert sb._bytes_sddmm({"nnz": E, "n_src": n, "n_dst": n, "l_len": 64, "r_len": 64, "out_len": 64, "targets": "uv"}) == 8 * E + 8 * n * 64 + 4 * E * 64
    assert sb._bytes_spmm({"op": "copy_rhs", "n_rows": n, "n_cols": n, "nnz": E, "out_len": 256}) == 4 * (n + 1) + 4 * E + 4 * E * 256 + 4 * n * 256
    assert sb.dominant_roofline([], 1) is None


def test_batches_of_different_sizes_fall_into_one_family():
    """molhiv: every batch is another graph; the family is (kernel, op, width) and the bytes are summed launch by launch."""
    recs = [_rec(0.01, op="copy_rhs", reduce="sum", out_len=256, n_rows=6000 + 10 * i, n_cols=6000 + 10 * i, nnz=13000 + 7 * i, accumulate=False)
            for i in range(10)]
    r = sb.dominant_roofline(recs, steps=1)
    want = sum(sb._bytes_spmm(x) for x in recs)
    assert r["launches_timed"] == 10 and r["algorithmic_bytes_per_launch"] == want // 10
    assert abs(r["achieved"] - want / 0.1e-3 / 1e9) < 0.1
