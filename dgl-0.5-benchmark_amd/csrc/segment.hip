// segment.hip -- per-segment readout (dgl.nn.AvgPooling / segment_reduce, main_dgl_molhiv_gcn.py:75,93).
// A segment reduction is a copy_rhs g-SpMM whose CSR is (offsets, identity): rows are contiguous
// slices of x, so the row-segmented wavefront kernels of spmm.hip stream them with 16-byte loads.
#include "common.h"

extern "C" int32_t mgx_segment_reduce(int64_t num_segments, const int64_t* offsets, int64_t D, int32_t reduce,
                                      const float* x, float* out, int64_t* arg, void* stream) {
  using namespace mgx;
  MGX_CHECK_ARG(num_segments >= 0 && D >= 0, "mgx_segment_reduce: negative sizes");
  MGX_CHECK_ARG(num_segments == 0 || offsets != nullptr, "mgx_segment_reduce: offsets is NULL");
  MGX_CHECK_ARG(reduce >= MGX_REDUCE_SUM && reduce <= MGX_REDUCE_MEAN, "mgx_segment_reduce: unsupported reduce op %d", reduce);
  if (num_segments == 0 || D == 0) return MGX_OK;
  mgx_csr csr;
  csr.num_rows = num_segments;
  csr.num_cols = 0;
  csr.indptr = offsets;
  csr.indices = nullptr;
  csr.eids = nullptr;
  csr.idx_bits = 64;
  csr.reserved = 0;
  // nnz drives only the SPLIT heuristic; segments are usually much longer than 64/G rows.
  csr.nnz = num_segments * 64;
  return mgx_spmm_csr(&csr, nullptr, MGX_OP_COPY_RHS, reduce, nullptr, x, 0, D, D, nullptr, nullptr, nullptr, nullptr,
                      out, nullptr, arg, nullptr, 0, stream);
}

// ------------------------------------------------------------------ column sum (one segment spanning every row)
// The bias gradient of the dense layers around every aggregation: out[c] = sum_r x[r, c].  A segment reduce with a single
// 2.4M-row segment has no row parallelism, so it is done in two stages: every workgroup adds a slab of rows (threads laid
// out as [rows_per_pass][C], so a pass reads rows_per_pass*C consecutive floats), the per-slab partials are added in slab
// order by one wave per column -- fixed order, no atomics, deterministic.
namespace mgx {

constexpr int kColSumSlabs = 2048;

__global__ __launch_bounds__(kBlock) void column_sum_partial_kernel(int64_t n, int C, const float* x, float* part) {
  __shared__ float lds[kBlock];
  const int rpp = kBlock / C;  // C <= kBlock
  const int rl = threadIdx.x / C, c = threadIdx.x % C;
  const bool active = rl < rpp;
  const int64_t slab = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * slab, r1 = (r0 + slab < n) ? r0 + slab : n;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (active) {
    int64_t r = r0 + rl;
    for (; r + 3 * rpp < r1; r += 4 * rpp) {  // four independent loads in flight
      a0 += x[r * C + c];
      a1 += x[(r + rpp) * C + c];
      a2 += x[(r + 2 * rpp) * C + c];
      a3 += x[(r + 3 * rpp) * C + c];
    }
    for (; r < r1; r += rpp) a0 += x[r * C + c];
  }
  lds[threadIdx.x] = active ? (a0 + a1) + (a2 + a3) : 0.f;
  __syncthreads();
  if (threadIdx.x < C) {
    float s = 0.f;
    for (int q = 0; q < rpp; ++q) s += lds[q * C + threadIdx.x];
    part[(int64_t)blockIdx.x * C + threadIdx.x] = s;
  }
}

__global__ __launch_bounds__(kBlock) void column_sum_finish_kernel(int slabs, int C, const float* part, float* out) {
  const int c = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int lane = threadIdx.x % kWave;
  if (c >= C) return;
  float s = 0.f;
  for (int q = lane; q < slabs; q += kWave) s += part[(int64_t)q * C + c];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, kWave);
  if (lane == 0) out[c] = s;
}

}  // namespace mgx

extern "C" int64_t mgx_column_sum_workspace(int64_t C) { return (int64_t)mgx::kColSumSlabs * C * (int64_t)sizeof(float); }

extern "C" int32_t mgx_column_sum(int64_t n, int64_t C, const float* x, float* out, void* workspace, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && C >= 1, "mgx_column_sum: bad sizes");
  MGX_CHECK_ARG(out != nullptr, "mgx_column_sum: out is NULL");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    MGX_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)C * sizeof(float), s));
    return MGX_OK;
  }
  MGX_CHECK_ARG(x != nullptr && workspace != nullptr, "mgx_column_sum: NULL pointer");
  const int64_t cw = C;
  if (C > kBlock) MGX_UNSUPPORTED("mgx_column_sum: C = %lld > %d columns", (long long)C, kBlock);
  hipLaunchKernelGGL(column_sum_partial_kernel, dim3(kColSumSlabs), dim3(kBlock), 0, s, n, (int)cw, x, (float*)workspace);
  MGX_CHECK_LAUNCH();
  hipLaunchKernelGGL(column_sum_finish_kernel, dim3((unsigned)((cw + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, s,
                     kColSumSlabs, (int)cw, (const float*)workspace, out);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

// ------------------------------------------------------------------ per-column statistics pairs + per-column affine map
// BatchNorm1d over the node dimension (main_dgl_arxiv_sage.py:70-77: bn -> relu -> dropout between aggregations) is two
// column reductions and one element-wise map each way; PyTorch's channels-last kernels take 0.6-0.9 ms for [169 k, 256]
// (173 MB: 0.04 ms at the HBM rate).  Same two-stage scheme as the column sum, two accumulators per column:
//   mode 0: (sum a, sum a*a)          forward statistics
//   mode 1: (sum a, sum a*b)          backward: (sum dy, sum dy*x)
//   mode 2: (sum (a-p), sum (a-p)^2)  p[c] = a[0, c]: statistics relative to the first row.  E[x^2] - mean^2 from plain
//   mode 3: (sum a, sum a*(b-p))      p[c] = b[0, c]  fp32 sums cancels catastrophically when |mean| >> std (mean 1e3,
//                                     std 1e-1 loses every significant bit of the variance); relative to ANY sample the
//                                     terms are O(std), which is what Welford's update achieves in torch.nn.BatchNorm1d.
namespace mgx {

__global__ __launch_bounds__(kBlock) void column_pair_partial_kernel(int64_t n, int C, int mode, const float* a, const float* b,
                                                                     float* part0, float* part1) {
  __shared__ float lds0[kBlock], lds1[kBlock];
  const int rpp = kBlock / C;
  const int rl = threadIdx.x / C, c = threadIdx.x % C;
  const bool active = rl < rpp;
  const int64_t slab = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * slab, r1 = (r0 + slab < n) ? r0 + slab : n;
  float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
  if (active) {
    const bool two = mode & 1;  // second operand is b
    const float pa = mode == 2 ? a[c] : 0.f, pb = mode == 3 ? b[c] : 0.f;  // pivot: first row of the shifted operand
    int64_t r = r0 + rl;
    for (; r + rpp < r1; r += 2 * rpp) {  // two independent rows in flight
      const float x0 = a[r * C + c] - pa, x1 = a[(r + rpp) * C + c] - pa;
      const float y0 = two ? b[r * C + c] - pb : x0, y1 = two ? b[(r + rpp) * C + c] - pb : x1;
      s0 += x0; s1 += x1;
      t0 += x0 * y0; t1 += x1 * y1;
    }
    for (; r < r1; r += rpp) {
      const float x0 = a[r * C + c] - pa;
      s0 += x0;
      t0 += x0 * (two ? b[r * C + c] - pb : x0);
    }
  }
  lds0[threadIdx.x] = active ? s0 + s1 : 0.f;
  lds1[threadIdx.x] = active ? t0 + t1 : 0.f;
  __syncthreads();
  if (threadIdx.x < C) {
    float u = 0.f, v = 0.f;
    for (int q = 0; q < rpp; ++q) {
      u += lds0[q * C + threadIdx.x];
      v += lds1[q * C + threadIdx.x];
    }
    part0[(int64_t)blockIdx.x * C + threadIdx.x] = u;
    part1[(int64_t)blockIdx.x * C + threadIdx.x] = v;
  }
}

// out[r, c] = a[r, c] * A[c] + (b ? b[r, c] * B[c] : 0) + Cc[c]; C % 4 == 0, 16-byte accesses
__global__ __launch_bounds__(kBlock) void column_affine_kernel(int64_t n4, int C4, const v4f* __restrict__ a, const v4f* __restrict__ b,
                                                               const v4f* __restrict__ A, const v4f* __restrict__ B,
                                                               const v4f* __restrict__ Cc, v4f* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const int c = (int)(i % C4);
    v4f v = a[i] * A[c] + Cc[c];
    if (b) v += b[i] * B[c];
    out[i] = v;
  }
}

}  // namespace mgx

extern "C" int32_t mgx_column_pair_sums(int64_t n, int64_t C, int32_t mode, const float* a, const float* b, float* out0, float* out1,
                                        void* workspace, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && C >= 1 && mode >= 0 && mode <= 3, "mgx_column_pair_sums: bad arguments");
  MGX_CHECK_ARG(out0 && out1, "mgx_column_pair_sums: out is NULL");
  if (C > kBlock) MGX_UNSUPPORTED("mgx_column_pair_sums: C = %lld > %d columns", (long long)C, kBlock);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    MGX_CHECK_HIP(hipMemsetAsync(out0, 0, (size_t)C * sizeof(float), s));
    MGX_CHECK_HIP(hipMemsetAsync(out1, 0, (size_t)C * sizeof(float), s));
    return MGX_OK;
  }
  MGX_CHECK_ARG(a && workspace && (!(mode & 1) || b), "mgx_column_pair_sums: NULL pointer");
  float* p0 = (float*)workspace;
  float* p1 = p0 + (int64_t)kColSumSlabs * C;
  hipLaunchKernelGGL(column_pair_partial_kernel, dim3(kColSumSlabs), dim3(kBlock), 0, s, n, (int)C, mode, a, b, p0, p1);
  MGX_CHECK_LAUNCH();
  const dim3 fgrid((unsigned)((C + kWavesPerBlock - 1) / kWavesPerBlock));
  hipLaunchKernelGGL(column_sum_finish_kernel, fgrid, dim3(kBlock), 0, s, kColSumSlabs, (int)C, (const float*)p0, out0);
  MGX_CHECK_LAUNCH();
  hipLaunchKernelGGL(column_sum_finish_kernel, fgrid, dim3(kBlock), 0, s, kColSumSlabs, (int)C, (const float*)p1, out1);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int32_t mgx_column_affine(int64_t n, int64_t C, const float* a, const float* b, const float* A, const float* B,
                                     const float* Cc, float* out, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && C >= 4 && C % 4 == 0, "mgx_column_affine: C must be a positive multiple of 4 (got %lld)", (long long)C);
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(a && A && Cc && out && (!b || B), "mgx_column_affine: NULL pointer");
  MGX_CHECK_ARG((uintptr_t)a % 16 == 0 && (uintptr_t)out % 16 == 0 && (!b || (uintptr_t)b % 16 == 0) && (uintptr_t)A % 16 == 0 &&
                (uintptr_t)Cc % 16 == 0 && (!B || (uintptr_t)B % 16 == 0), "mgx_column_affine: pointers must be 16-byte aligned");
  const int64_t n4 = n * (C / 4);
  int64_t blocks = (n4 + kBlock - 1) / kBlock;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(column_affine_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, n4, (int)(C / 4), (const v4f*)a,
                     (const v4f*)b, (const v4f*)A, (const v4f*)B, (const v4f*)Cc, (v4f*)out);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
