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
