// xty.hip -- C[M, K] = A^T B for A [n, M], B [n, K] row-major with n in the millions and M, K <= 64 / 128 (one tile; up to
// 256 x 1024 as a grid of such tiles in one launch): the weight
// gradient dW = dY^T X of the dense layer that follows every aggregation (SAGEConv's fc_self / fc_neigh,
// main_dgl_product_sage.py:31-33,64).  rocBLAS / hipBLASLt run these tall-skinny reductions (K-dim = 2.45 M) at 0.5-1.0 ms;
// the operands only need to be streamed once (1.6 GB at 64 x 100), so the bound is HBM.
//
// fp32 MFMA 16x16x4: the "k" of the instruction runs over data rows.  Lane l supplies A[row0 + l/16][m0 + l%16] and
// B[row0 + l/16][k0 + l%16] -- both row-major, i.e. operands are loaded straight from global memory in the layout the
// instruction wants, no LDS staging and no transpose; a wave keeps the whole (M/16) x (K/16) grid of 16x16 accumulators
// (<= 4 x 8 x 4 = 128 VGPRs) and walks 16 rows per trip.  Every wave writes its partial tile; a second launch adds the
// partials in wave order (deterministic, no atomics).
#include "common.h"

namespace mgx {

constexpr int kXtyWaves = 2048;  // most partial tiles (workspace size); fewer for shorter inputs, see xty_waves()

// One partial tile (<= 32 KB written, then read back) per wave: keep >= 256 rows per wave so that the tiles stay a small
// part of the traffic, between 256 waves (one per CU) and kXtyWaves; a multiple of 64 for the finish kernel's 16 slices.
static int xty_waves(int64_t n) {
  int64_t w = (n / 256 + 63) / 64 * 64;
  if (w < 256) w = 256;
  if (w > kXtyWaves) w = kXtyWaves;
  return (int)w;
}

template <int MT, int KT>
__global__ __launch_bounds__(kBlock) void xty_partial_kernel(int64_t n, int M, int K, const float* __restrict__ A, int64_t lda,
                                                             const float* __restrict__ B, int64_t ldb, float* __restrict__ part) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t gw = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int c = lane % 16, q = lane / 16;
  v4f acc[MT][KT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[i][j] = (v4f)(0.f);
  bool am[MT], bm[KT];
#pragma unroll
  for (int i = 0; i < MT; ++i) am[i] = i * 16 + c < M;
#pragma unroll
  for (int j = 0; j < KT; ++j) bm[j] = j * 16 + c < K;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * 16;
  for (int64_t r0 = gw * 16; r0 < n; r0 += stride) {
    float a[4][MT], b[4][KT];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int64_t row = r0 + s * 4 + q;
      const bool ok = row < n;
#pragma unroll
      for (int i = 0; i < MT; ++i) a[s][i] = (ok && am[i]) ? A[row * lda + i * 16 + c] : 0.f;
#pragma unroll
      for (int j = 0; j < KT; ++j) b[s][j] = (ok && bm[j]) ? B[row * ldb + j * 16 + c] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
  }
  // D[4*(lane/16) + r][lane%16] of every tile -> partial [MT*16][KT*16]
  float* p = part + gw * (int64_t)(MT * 16) * (KT * 16);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) p[(i * 16 + 4 * q + r) * (KT * 16) + j * 16 + c] = acc[i][j][r];
}

// Round 4: the same partial kernel with 16-BYTE operand loads.  The MFMA wants lane (c, q) to supply A[row + q][m = c] of a 16-column
// tile -- a dword per lane, i.e. 256 bytes per load instruction for 16 rows.  But which 16 columns form a "tile" is only a labelling of the
// OUTPUT: a lane that loads the float4 A[row][4c .. 4c+3] holds one column of FOUR tiles (tile t = component t, its 16 columns are
// 4c + t), so a 64-column operand arrives in one 16-byte load per lane and row, and the epilogue writes accumulator (tile t, index c) to
// column 4c + t.  Per operand: VG groups of 64 columns by float4 loads (VG * 4 tiles) + ST tiles of 16 columns by dword loads for the
// remainder (K = 72 = one group + one tile) or for an operand whose row stride is not a multiple of 4 floats (the 47-column output
// gradient).  Same accumulators, same partial layout, same finish kernel; fp32 MFMA, sums in another (fixed) order.
template <int AVG, int AST, int BVG, int BST>
__global__ __launch_bounds__(kBlock) void xty_partial_v4_kernel(int64_t n, int M, int K, int mt16, int kt16, const float* __restrict__ A,
                                                                int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                                float* __restrict__ part, float* __restrict__ part_sum) {
  // part_sum (may be NULL): the COLUMN SUMS of A as well -- the bias gradient that goes with this weight gradient -- as the product with
  // one more B column of ones: four MFMAs per 16 rows instead of another pass over A (mgx_column_sum: 0.08 - 0.11 ms at N = 2.45 M)
  constexpr int AT = AVG * 4 + AST, BT = BVG * 4 + BST;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t gw = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int c = lane % 16, q = lane / 16;
  v4f acc[AT][BT];
  v4f accs[AT > 0 ? AT : 1];
#pragma unroll
  for (int i = 0; i < AT; ++i) {
    accs[i] = (v4f)(0.f);
#pragma unroll
    for (int j = 0; j < BT; ++j) acc[i][j] = (v4f)(0.f);
  }
  const float one = c == 0 ? 1.f : 0.f;  // B column 0 of the extra tile is all ones, the other fifteen zero
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * 16;
  for (int64_t r0 = gw * 16; r0 < n; r0 += stride) {
    float a[4][AT > 0 ? AT : 1], b[4][BT > 0 ? BT : 1];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int64_t row = r0 + s * 4 + q;
      const bool ok = row < n;
#pragma unroll
      for (int g = 0; g < AVG; ++g) {
        const v4f v = (ok && g * 64 + 4 * c + 3 < M) ? *reinterpret_cast<const v4f*>(A + row * lda + g * 64 + 4 * c) : (v4f)(0.f);
        a[s][g * 4 + 0] = v.x; a[s][g * 4 + 1] = v.y; a[s][g * 4 + 2] = v.z; a[s][g * 4 + 3] = v.w;
      }
#pragma unroll
      for (int i = 0; i < AST; ++i) a[s][AVG * 4 + i] = (ok && AVG * 64 + i * 16 + c < M) ? A[row * lda + AVG * 64 + i * 16 + c] : 0.f;
#pragma unroll
      for (int g = 0; g < BVG; ++g) {
        const v4f v = (ok && g * 64 + 4 * c + 3 < K) ? *reinterpret_cast<const v4f*>(B + row * ldb + g * 64 + 4 * c) : (v4f)(0.f);
        b[s][g * 4 + 0] = v.x; b[s][g * 4 + 1] = v.y; b[s][g * 4 + 2] = v.z; b[s][g * 4 + 3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < BST; ++j) b[s][BVG * 4 + j] = (ok && BVG * 64 + j * 16 + c < K) ? B[row * ldb + BVG * 64 + j * 16 + c] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int j = 0; j < BT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
    if (part_sum) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < AT; ++i) accs[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][i], (r0 + s * 4 + q < n) ? one : 0.f, accs[i], 0, 0, 0);
    }
  }
  if (part_sum && c == 0) {  // lane (0, q) holds the sums of the real columns behind (tile i, index 4 q + r): one [mt16] vector per wave
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mi = 4 * q + r;
        const int m = i < AVG * 4 ? (i / 4) * 64 + 4 * mi + (i % 4) : AVG * 64 + (i - AVG * 4) * 16 + mi;
        if (m < mt16) part_sum[gw * (int64_t)mt16 + m] = accs[i][r];
      }
  }
  // accumulator element r of tile (i, j) in lane (c, q) is C[m][k] with m / k the REAL columns behind (tile i, index 4q + r) / (tile j, index c)
  float* p = part + gw * (int64_t)mt16 * kt16;
#pragma unroll
  for (int i = 0; i < AT; ++i)
#pragma unroll
    for (int j = 0; j < BT; ++j) {
      const int k = j < BVG * 4 ? (j / 4) * 64 + 4 * c + (j % 4) : BVG * 64 + (j - BVG * 4) * 16 + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mi = 4 * q + r;
        const int m = i < AVG * 4 ? (i / 4) * 64 + 4 * mi + (i % 4) : AVG * 64 + (i - AVG * 4) * 16 + mi;
        if (m < mt16 && k < kt16) p[m * kt16 + k] = acc[i][j][r];
      }
    }
}

template <int AVG, int AST>
static bool launch_xty_v4_b(int bvg, int bst, int waves, int64_t n, int M, int K, int mt16, int kt16, const float* A, int64_t lda,
                            const float* B, int64_t ldb, float* part, float* part_sum, hipStream_t s) {
#define MGX_XTY4(G, T) if (bvg == G && bst == T) { hipLaunchKernelGGL((xty_partial_v4_kernel<AVG, AST, G, T>), dim3(waves / kWavesPerBlock), dim3(kBlock), 0, s, n, M, K, mt16, kt16, A, lda, B, ldb, part, part_sum); return true; }
  MGX_XTY4(2, 0) MGX_XTY4(1, 0) MGX_XTY4(1, 1) MGX_XTY4(1, 2) MGX_XTY4(1, 3) MGX_XTY4(1, 4)
#undef MGX_XTY4
  return false;
}

// 16-byte loads for the operand(s) that allow it; false = use the dword kernels above
static bool launch_xty_v4(int mt, int kt, int waves, int64_t n, int M, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                          float* part, float* part_sum, hipStream_t s) {
  const bool bvec = ldb % 4 == 0 && (uintptr_t)B % 16 == 0 && K % 4 == 0 && K >= 64;
  if (!bvec) return false;
  const int bvg = K >= 128 ? 2 : 1, bst = (K - bvg * 64 + 15) / 16;
  const bool avec = lda % 4 == 0 && (uintptr_t)A % 16 == 0 && M == 64;
  const int mt16 = mt * 16, kt16 = kt * 16;
  if (avec) return launch_xty_v4_b<1, 0>(bvg, bst, waves, n, M, K, mt16, kt16, A, lda, B, ldb, part, part_sum, s);
  switch (mt) {
    case 1: return launch_xty_v4_b<0, 1>(bvg, bst, waves, n, M, K, mt16, kt16, A, lda, B, ldb, part, part_sum, s);
    case 2: return launch_xty_v4_b<0, 2>(bvg, bst, waves, n, M, K, mt16, kt16, A, lda, B, ldb, part, part_sum, s);
    case 3: return launch_xty_v4_b<0, 3>(bvg, bst, waves, n, M, K, mt16, kt16, A, lda, B, ldb, part, part_sum, s);
    default: return launch_xty_v4_b<0, 4>(bvg, bst, waves, n, M, K, mt16, kt16, A, lda, B, ldb, part, part_sum, s);
  }
}

// 16 output elements x 16 slices of the partial list per workgroup; slices combined in slice order
__global__ __launch_bounds__(kBlock) void xty_finish_kernel(int M, int K, int ldm /* KT*16 */, int tile /* MT*16*KT*16 */, int waves,
                                                            const float* __restrict__ part, float* __restrict__ out, int64_t ldc) {
  __shared__ float red[16][17];
  const int e = threadIdx.x % 16, sl = threadIdx.x / 16;
  const int idx = blockIdx.x * 16 + e;  // over M*K
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (idx < M * K) {
    const int m = idx / K, k = idx % K;
    const float* p = part + (int64_t)m * ldm + k;
    const int per = waves / 16;
    for (int w = sl * per; w < (sl + 1) * per; w += 4) {
      s0 += p[(int64_t)w * tile];
      s1 += p[(int64_t)(w + 1) * tile];
      s2 += p[(int64_t)(w + 2) * tile];
      s3 += p[(int64_t)(w + 3) * tile];
    }
  }
  red[sl][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && idx < M * K) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[i][e];
    out[(int64_t)(idx / K) * ldc + idx % K] = s;
  }
}

// Outputs wider than one 64 x 128 tile (arxiv's 256 x 512, GAT's 128 x 602, the stacked 64 x 200 of the one-GEMM SAGE layer):
// ONE launch, blockIdx.y = output tile.  The waves of different tiles walk the same data rows at the same time, so an operand
// row is read from HBM once and from L2 by the other tiles, and 4 waves per SIMD are resident -- against one launch per tile,
// each streaming both operands again with less than one wave per SIMD.  About kXtyGridTotal waves in all (4 per SIMD at 128
// VGPRs) shared evenly by the tiles; partials per (tile, wave).
constexpr int kXtyGridTotal = 4096;
constexpr int kXtyTileM = 64, kXtyTileK = 128;

__global__ __launch_bounds__(kBlock) void xty_partial_grid_kernel(int64_t n, int M, int K, int tiles_k, const float* __restrict__ A,
                                                                  int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                                  float* __restrict__ part) {
  constexpr int MT = 4, KT = 8;
  const int tile = blockIdx.y;
  const int m0 = (tile / tiles_k) * kXtyTileM, k0 = (tile % tiles_k) * kXtyTileK;
  const int Mt = M - m0 < kXtyTileM ? M - m0 : kXtyTileM, Kt = K - k0 < kXtyTileK ? K - k0 : kXtyTileK;
  A += m0;
  B += k0;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t gw = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int c = lane % 16, q = lane / 16;
  v4f acc[MT][KT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j) acc[i][j] = (v4f)(0.f);
  bool am[MT], bm[KT];
#pragma unroll
  for (int i = 0; i < MT; ++i) am[i] = i * 16 + c < Mt;
#pragma unroll
  for (int j = 0; j < KT; ++j) bm[j] = j * 16 + c < Kt;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * 16;
  for (int64_t r0 = gw * 16; r0 < n; r0 += stride) {
    float a[4][MT], b[4][KT];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int64_t row = r0 + s * 4 + q;
      const bool ok = row < n;
#pragma unroll
      for (int i = 0; i < MT; ++i) a[s][i] = (ok && am[i]) ? A[row * lda + i * 16 + c] : 0.f;
#pragma unroll
      for (int j = 0; j < KT; ++j) b[s][j] = (ok && bm[j]) ? B[row * ldb + j * 16 + c] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
  }
  const int64_t wpt = (int64_t)gridDim.x * kWavesPerBlock;  // waves per tile
  float* p = part + ((int64_t)tile * wpt + gw) * (int64_t)(kXtyTileM * kXtyTileK);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) p[(i * 16 + 4 * q + r) * kXtyTileK + j * 16 + c] = acc[i][j][r];
}

// blockIdx.y = tile; blockIdx.x over the 64 x 128 elements of a tile, 16 per workgroup; slices combined in slice order
__global__ __launch_bounds__(kBlock) void xty_finish_grid_kernel(int M, int K, int tiles_k, int wpt, const float* __restrict__ part,
                                                                 float* __restrict__ out, int64_t ldc) {
  __shared__ float red[16][17];
  const int tile = blockIdx.y;
  const int m0 = (tile / tiles_k) * kXtyTileM, k0 = (tile % tiles_k) * kXtyTileK;
  const int e = threadIdx.x % 16, sl = threadIdx.x / 16;
  const int idx = blockIdx.x * 16 + e;  // over the tile's 64 x 128 elements
  const int m = idx / kXtyTileK, k = idx % kXtyTileK;
  const bool live = m0 + m < M && k0 + k < K;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (live) {
    const int tsz = kXtyTileM * kXtyTileK;
    const float* p = part + (int64_t)tile * wpt * tsz + idx;
    const int per = wpt / 16;
    for (int w = sl * per; w < (sl + 1) * per; w += 4) {
      s0 += p[(int64_t)w * tsz];
      s1 += p[(int64_t)(w + 1) * tsz];
      s2 += p[(int64_t)(w + 2) * tsz];
      s3 += p[(int64_t)(w + 3) * tsz];
    }
  }
  red[sl][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && live) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[i][e];
    out[(int64_t)(m0 + m) * ldc + k0 + k] = s;
  }
}

template <int MT>
static bool launch_xty_kt(int kt, int waves, int64_t n, int M, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                          float* part, hipStream_t s) {
  switch (kt) {
#define MGX_XTY(J) case J: hipLaunchKernelGGL((xty_partial_kernel<MT, J>), dim3(waves / kWavesPerBlock), dim3(kBlock), 0, s, n, M, K, A, lda, B, ldb, part); return true;
    MGX_XTY(1) MGX_XTY(2) MGX_XTY(3) MGX_XTY(4) MGX_XTY(5) MGX_XTY(6) MGX_XTY(7) MGX_XTY(8)
#undef MGX_XTY
    default: return false;
  }
}

}  // namespace mgx

namespace mgx {
// column sums: the [waves][mt16] partial vectors added in wave order, one wave per output column
__global__ __launch_bounds__(kBlock) void xty_colsum_finish_kernel(int M, int mt16, int waves, const float* __restrict__ part_sum,
                                                                   float* __restrict__ colsum) {
  const int m = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int lane = threadIdx.x % kWave;
  if (m >= M) return;
  float s = 0.f;
  for (int w = lane; w < waves; w += kWave) s += part_sum[(int64_t)w * mt16 + m];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, kWave);
  if (lane == 0) colsum[m] = s;
}
}  // namespace mgx

extern "C" int64_t mgx_xty_workspace(int64_t M, int64_t K) {
  const int64_t mt = (M + 15) / 16, kt = (K + 15) / 16;
  if (M < 1 || K < 1) return -1;
  if (mt > 4 || kt > 8) {  // several 64 x 128 tiles in one launch
    if (M > 256 || K > 1024) return -1;
    return (int64_t)mgx::kXtyGridTotal * mgx::kXtyTileM * mgx::kXtyTileK * (int64_t)sizeof(float);  // tiles x waves-per-tile <= this many
  }
  return (int64_t)mgx::kXtyWaves * (mt * 16 * kt * 16 + mt * 16) * (int64_t)sizeof(float);  // partial tiles + partial column sums
}

static int32_t xty_impl(int64_t n, int64_t M, int64_t K, const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldc,
                        float* colsum, void* workspace, void* stream);

extern "C" int32_t mgx_xty(int64_t n, int64_t M, int64_t K, const float* a, int64_t lda, const float* b, int64_t ldb, float* out,
                           int64_t ldc, void* workspace, void* stream) {
  return xty_impl(n, M, K, a, lda, b, ldb, out, ldc, nullptr, workspace, stream);
}

extern "C" int32_t mgx_xty_colsum(int64_t n, int64_t M, int64_t K, const float* a, int64_t lda, const float* b, int64_t ldb, float* out,
                                  int64_t ldc, float* colsum, void* workspace, void* stream) {
  if (!colsum) return xty_impl(n, M, K, a, lda, b, ldb, out, ldc, nullptr, workspace, stream);
  return xty_impl(n, M, K, a, lda, b, ldb, out, ldc, colsum, workspace, stream);
}

static int32_t xty_impl(int64_t n, int64_t M, int64_t K, const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldc,
                        float* colsum, void* workspace, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && M >= 1 && K >= 1, "mgx_xty: bad sizes");
  const int mt = (int)((M + 15) / 16), kt = (int)((K + 15) / 16);
  if (M > 256 || K > 1024) MGX_UNSUPPORTED("mgx_xty: needs M <= 256 and K <= 1024 (got %lld x %lld)", (long long)M, (long long)K);
  MGX_CHECK_ARG(out != nullptr, "mgx_xty: out is NULL");
  MGX_CHECK_ARG(lda >= M && ldb >= K && ldc >= K, "mgx_xty: leading dimensions smaller than the tile (lda %lld, ldb %lld, ldc %lld)",
                (long long)lda, (long long)ldb, (long long)ldc);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    MGX_CHECK_HIP(hipMemset2DAsync(out, (size_t)ldc * sizeof(float), 0, (size_t)K * sizeof(float), (size_t)M, s));
    if (colsum) MGX_CHECK_HIP(hipMemsetAsync(colsum, 0, (size_t)M * sizeof(float), s));
    return MGX_OK;
  }
  MGX_CHECK_ARG(a && b && workspace, "mgx_xty: NULL pointer");
  float* part = (float*)workspace;
  if (mt > 4 || kt > 8) {
    if (colsum) MGX_UNSUPPORTED("mgx_xty_colsum: one 64 x 128 tile only");
    const int tiles_m = (int)((M + kXtyTileM - 1) / kXtyTileM), tiles_k = (int)((K + kXtyTileK - 1) / kXtyTileK);
    // waves per tile: an even share of kXtyGridTotal, a multiple of 64 (the finish kernel's 16 slices x 4), at least 64 and at
    // most what gives every wave 256 rows
    int wpt = kXtyGridTotal / (tiles_m * tiles_k) / 64 * 64;
    const int64_t by_rows = n / 128 / 64 * 64;  // >= 128 rows per wave: short inputs (a batch of small graphs) take few waves
    if (wpt > by_rows) wpt = (int)by_rows;
    if (wpt < 64) wpt = 64;
    hipLaunchKernelGGL(xty_partial_grid_kernel, dim3(wpt / kWavesPerBlock, tiles_m * tiles_k), dim3(kBlock), 0, s, n, (int)M,
                       (int)K, tiles_k, a, lda, b, ldb, part);
    MGX_CHECK_LAUNCH();
    hipLaunchKernelGGL(xty_finish_grid_kernel, dim3(kXtyTileM * kXtyTileK / 16, tiles_m * tiles_k), dim3(kBlock), 0, s, (int)M, (int)K,
                       tiles_k, wpt, (const float*)part, out, ldc);
    MGX_CHECK_LAUNCH();
    return MGX_OK;
  }
  const int waves = xty_waves(n);
  float* part_sum = colsum ? part + (int64_t)kXtyWaves * mt * 16 * kt * 16 : nullptr;
  bool ok = launch_xty_v4(mt, kt, waves, n, (int)M, (int)K, a, lda, b, ldb, part, part_sum, s);
  if (!ok && colsum) MGX_UNSUPPORTED("mgx_xty_colsum: needs the 16-byte-load form (B: K %% 4 == 0, K >= 64, aligned)");
  if (!ok) switch (mt) {
    case 1: ok = launch_xty_kt<1>(kt, waves, n, (int)M, (int)K, a, lda, b, ldb, part, s); break;
    case 2: ok = launch_xty_kt<2>(kt, waves, n, (int)M, (int)K, a, lda, b, ldb, part, s); break;
    case 3: ok = launch_xty_kt<3>(kt, waves, n, (int)M, (int)K, a, lda, b, ldb, part, s); break;
    default: ok = launch_xty_kt<4>(kt, waves, n, (int)M, (int)K, a, lda, b, ldb, part, s); break;
  }
  MGX_CHECK_ARG(ok, "mgx_xty: no kernel for this tile shape");
  MGX_CHECK_LAUNCH();
  hipLaunchKernelGGL(xty_finish_kernel, dim3((unsigned)((M * K + 15) / 16)), dim3(kBlock), 0, s, (int)M, (int)K, kt * 16,
                     mt * 16 * kt * 16, waves, (const float*)part, out, ldc);
  MGX_CHECK_LAUNCH();
  if (colsum) {
    hipLaunchKernelGGL(xty_colsum_finish_kernel, dim3((unsigned)((M + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, s, (int)M,
                       mt * 16, waves, (const float*)part_sum, colsum);
    MGX_CHECK_LAUNCH();
  }
  return MGX_OK;
}
