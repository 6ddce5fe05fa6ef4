// headdot.hip -- GAT attention terms: el[n,h] = <feat[n,h,:], attn[h,:]> and their backward.
//
// GATConv (main_dgl_reddit_gat.py:10; UPSTREAM dgl.nn.pytorch.GATConv) computes `(feat * attn_l).sum(-1)` with an
// element-wise product and a last-dim reduction, and its backward with a product and a column reduction over all N
// rows; on MI355X PyTorch's reduce kernels run those at ~200 GB/s (27 % of a reddit 8-head epoch).  Here:
//   fwd: one pass over feat [N, H*F]: lanes along the H*F row (float4), the F/4 lanes of a head combine by xor-shuffle;
//        both attention vectors (attn_l, attn_r) are applied in the same pass when they share `feat`.
//   bwd: d_feat[n,h,f] = d_el[n,h]*attn_l[h,f] (+ d_er*attn_r) -- element-wise, written once;
//        d_attn[h,f]   = sum_n d_el[n,h]*feat[n,h,f] -- two-stage column reduction: every workgroup reduces a slab of rows
//        into partial[block, H*F], a second launch adds the partials in block order (deterministic, no atomics).
#include "common.h"

namespace mgx {

constexpr int kHdBlocks = 1024;  // slabs of the column reduction

// one thread per (row, 4 consecutive features); G = C/4 lanes per row, F/4 lanes per head (both powers of two)
template <int LPH /* lanes per head = F/4 */>
__global__ __launch_bounds__(kBlock) void head_dot_fwd_kernel(int64_t n, int C, const float* X, const float* A, const float* B,
                                                              float* outA, float* outB) {
  const int lanes_per_row = C / 4;
  const int H = lanes_per_row / LPH;
  const int64_t total = n * lanes_per_row;
  // a head's LPH lanes are an aligned lane group (LPH | 64, LPH | lanes_per_row): it is live or dead as a whole
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t row = t / lanes_per_row;
    const int c4 = (int)(t % lanes_per_row);
    const v4f x = *reinterpret_cast<const v4f*>(X + row * C + c4 * 4);
    const v4f a = *reinterpret_cast<const v4f*>(A + c4 * 4);
    float sa = x.x * a.x + x.y * a.y + x.z * a.z + x.w * a.w;
    float sb = 0.f;
    if (B) {
      const v4f b = *reinterpret_cast<const v4f*>(B + c4 * 4);
      sb = x.x * b.x + x.y * b.y + x.z * b.z + x.w * b.w;
    }
#pragma unroll
    for (int off = 1; off < LPH; off <<= 1) {
      sa += __shfl_xor(sa, off, kWave);
      sb += __shfl_xor(sb, off, kWave);
    }
    if ((c4 % LPH) == 0) {
      outA[row * H + c4 / LPH] = sa;
      if (B) outB[row * H + c4 / LPH] = sb;
    }
  }
}

// any F <= 64 (e.g. the 41-class output layer): LPH = next power of two >= F lanes per head, one feature per lane
template <int LPH>
__global__ __launch_bounds__(kBlock) void head_dot_fwd_scalar_kernel(int64_t n, int H, int F, const float* X, const float* A,
                                                                     const float* B, float* outA, float* outB) {
  const int64_t total = n * H * LPH;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t rh = t / LPH;  // row * H + head
    const int f = (int)(t % LPH), h = (int)(rh % H);
    float sa = 0.f, sb = 0.f;
    if (f < F) {
      const float x = X[rh * F + f];
      sa = x * A[h * F + f];
      if (B) sb = x * B[h * F + f];
    }
#pragma unroll
    for (int off = 1; off < LPH; off <<= 1) {
      sa += __shfl_xor(sa, off, kWave);
      sb += __shfl_xor(sb, off, kWave);
    }
    if (f == 0) {
      outA[rh] = sa;
      if (B) outB[rh] = sb;
    }
  }
}

// d_feat = dA[n,h]*A[h,f] (+ dB[n,h]*B[h,f]) and per-slab column partials of dA[n,h]*X[n,c] (and dB*X)
__global__ __launch_bounds__(kBlock) void head_dot_bwd_kernel(int64_t n, int C, int F, const float* X, const float* A,
                                                              const float* B, const float* dA, const float* dB, float* dX,
                                                              float* partA, float* partB) {
  extern __shared__ float lds[];  // [rows_per_pass][C] x 2
  const int H = C / F;
  const int rpp = kBlock / C > 0 ? kBlock / C : 1;  // rows handled per pass by one workgroup (C <= 256)
  const int rl = threadIdx.x / C, c = threadIdx.x % C;
  const bool tactive = threadIdx.x < rpp * C;
  const int h = c / F;
  const int64_t slab = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * slab, r1 = (r0 + slab < n) ? r0 + slab : n;
  const float a = tactive ? A[c] : 0.f, b = (tactive && B) ? B[c] : 0.f;
  float accA = 0.f, accB = 0.f;
  if (tactive) {
    for (int64_t r = r0 + rl; r < r1; r += rpp) {
      const float x = X[r * C + c];
      const float ga = dA[r * H + h];
      float g = ga * a;
      accA += ga * x;
      if (B) {
        const float gb = dB[r * H + h];
        g += gb * b;
        accB += gb * x;
      }
      if (dX) dX[r * C + c] = g;
    }
    lds[rl * C + c] = accA;
    lds[(rpp + rl) * C + c] = accB;
  }
  __syncthreads();
  if (threadIdx.x < C) {
    float sa = 0.f, sb = 0.f;
    for (int q = 0; q < rpp; ++q) {
      sa += lds[q * C + threadIdx.x];
      sb += lds[(rpp + q) * C + threadIdx.x];
    }
    partA[(int64_t)blockIdx.x * C + threadIdx.x] = sa;
    if (B) partB[(int64_t)blockIdx.x * C + threadIdx.x] = sb;
  }
}

// one wave per column: lane l adds partials l, l+64, ... in order, then a fixed xor tree -> deterministic
__global__ __launch_bounds__(kBlock) void head_dot_finish_kernel(int blocks, int C, const float* partA, const float* partB,
                                                                 float* gA, float* gB) {
  const int c = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int lane = threadIdx.x % kWave;
  if (c >= C) return;
  float sa = 0.f, sb = 0.f;
  for (int q = lane; q < blocks; q += kWave) {
    sa += partA[(int64_t)q * C + c];
    if (partB) sb += partB[(int64_t)q * C + c];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sa += __shfl_xor(sa, off, kWave);
    sb += __shfl_xor(sb, off, kWave);
  }
  if (lane == 0) {
    gA[c] = sa;
    if (gB) gB[c] = sb;
  }
}

static bool head_dot_vector_path(int64_t F) {
  const int64_t lph = F / 4;
  return F % 4 == 0 && (lph & (lph - 1)) == 0 && lph <= 64;
}
static bool head_dot_supported(int64_t H, int64_t F) { return H * F <= 256 && (head_dot_vector_path(F) || F <= 64); }

}  // namespace mgx

extern "C" int32_t mgx_head_dot_fwd(int64_t n, int64_t H, int64_t F, const float* feat, const float* attn_a,
                                    const float* attn_b, float* out_a, float* out_b, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && H >= 1 && F >= 1, "mgx_head_dot_fwd: bad sizes");
  if (!head_dot_supported(H, F)) MGX_UNSUPPORTED("mgx_head_dot: needs H*F <= 256 and F <= 64 or F in {128, 256} (got H=%lld F=%lld)", (long long)H, (long long)F);
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(feat && attn_a && out_a && (!attn_b || out_b), "mgx_head_dot_fwd: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  const int C = (int)(H * F);
  if (head_dot_vector_path(F)) {
    MGX_CHECK_ARG((uintptr_t)feat % 16 == 0 && (uintptr_t)attn_a % 16 == 0 && (!attn_b || (uintptr_t)attn_b % 16 == 0), "mgx_head_dot_fwd: pointers must be 16-byte aligned");
    const int64_t total = n * (C / 4);
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    switch ((int)(F / 4)) {
#define MGX_HD(L) case L: hipLaunchKernelGGL((head_dot_fwd_kernel<L>), dim3((unsigned)blocks), dim3(kBlock), 0, s, n, C, feat, attn_a, attn_b, out_a, out_b); break;
      MGX_HD(1) MGX_HD(2) MGX_HD(4) MGX_HD(8) MGX_HD(16) MGX_HD(32) MGX_HD(64)
#undef MGX_HD
    }
  } else {
    const int lph = 1 << ilog2_ceil(F);
    const int64_t total = n * H * lph;
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    switch (lph) {
#define MGX_HD(L) case L: hipLaunchKernelGGL((head_dot_fwd_scalar_kernel<L>), dim3((unsigned)blocks), dim3(kBlock), 0, s, n, (int)H, (int)F, feat, attn_a, attn_b, out_a, out_b); break;
      MGX_HD(1) MGX_HD(2) MGX_HD(4) MGX_HD(8) MGX_HD(16) MGX_HD(32) MGX_HD(64)
#undef MGX_HD
    }
  }
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int64_t mgx_head_dot_bwd_workspace(int64_t H, int64_t F) { return 2 * (int64_t)mgx::kHdBlocks * H * F * (int64_t)sizeof(float); }

extern "C" int32_t mgx_head_dot_bwd(int64_t n, int64_t H, int64_t F, const float* feat, const float* attn_a,
                                    const float* attn_b, const float* d_out_a, const float* d_out_b, float* d_feat,
                                    float* d_attn_a, float* d_attn_b, void* workspace, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && H >= 1 && F >= 1, "mgx_head_dot_bwd: bad sizes");
  if (!head_dot_supported(H, F)) MGX_UNSUPPORTED("mgx_head_dot: needs H*F <= 256 and F <= 64 or F in {128, 256} (got H=%lld F=%lld)", (long long)H, (long long)F);
  if (n == 0) {
    MGX_CHECK_HIP(hipMemsetAsync(d_attn_a, 0, (size_t)(H * F) * sizeof(float), (hipStream_t)stream));
    if (d_attn_b) MGX_CHECK_HIP(hipMemsetAsync(d_attn_b, 0, (size_t)(H * F) * sizeof(float), (hipStream_t)stream));
    return MGX_OK;
  }
  MGX_CHECK_ARG(feat && attn_a && d_out_a && d_attn_a && workspace && (!attn_b || (d_out_b && d_attn_b)), "mgx_head_dot_bwd: NULL pointer");
  const int C = (int)(H * F);
  hipStream_t s = (hipStream_t)stream;
  float* partA = (float*)workspace;
  float* partB = attn_b ? partA + (int64_t)kHdBlocks * C : nullptr;
  const int rpp = kBlock / C > 0 ? kBlock / C : 1;
  hipLaunchKernelGGL(head_dot_bwd_kernel, dim3(kHdBlocks), dim3(kBlock), (size_t)(2 * rpp * C) * sizeof(float), s, n, C, (int)F, feat,
                     attn_a, attn_b, d_out_a, d_out_b, d_feat, partA, partB);
  MGX_CHECK_LAUNCH();
  hipLaunchKernelGGL(head_dot_finish_kernel, dim3((unsigned)((C + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, s, kHdBlocks, C,
                     (const float*)partA, (const float*)partB, d_attn_a, d_attn_b);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
