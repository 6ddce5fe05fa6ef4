// rowsgemm.hip -- C[n, M] = A[n, K] x B[K, M] (+ bias[M]) (x row_scale[n] on the columns from `scale_from` on) for A with millions
// of rows and SMALL K, M: the dense projections either side of every aggregation of a full-graph layer --
//   forward  [h | mean_agg(h)] (N x 2K') x [W_self | W_neigh]^T        (SAGEConv: main_dgl_product_sage.py:23-24,64)
//   backward d[h | neigh] = dY (N x out) x [W_self | W_neigh], the `neigh` half times 1 / deg (the backward of fn.mean, :62)
// The library GEMMs run these shapes at 3.4 - 4.9 TB/s of the operands they must stream and the 1 / deg factor needs another
// read-modify-write of N x K' floats; the bound is HBM (A read once, C written once; B is 32 KB).
//
// One wave owns 16 rows at a time and ALL M columns: fp32 MFMA 16x16x4, i = data row, j = output column, k = the K dimension.
// Which four k an instruction consumes is only a labelling as long as A and B agree: lane (c = l % 16, q = l / 16) loads the float4
// A[row0 + c][16 jj + 4 q .. + 3] -- 64 contiguous bytes per row and load instruction -- and step (jj, t) pairs component t with
// B[16 jj + 4 q + t][m].  B is staged ONCE per workgroup in LDS in exactly the order the lanes read it (one ds_read_b128 per (jj, column
// tile), conflict-free), zero-padded beyond K and M.  Accumulator element r of column tile mt is C[row0 + 4 q + r][16 mt + c]; bias,
// the row factor and the bounds are applied there.  WHICH 16 columns a tile holds is a labelling too: tile mt, index c stands for column
// 64 (mt / 4) + 4 c + mt % 4, so a lane ends up with four CONSECUTIVE columns of a row in four accumulators and writes them as one float4
// -- 256 contiguous bytes per row and store instruction (tile-major columns gave 64-byte pieces: 2.0 TB/s of writes).  K % 4 != 0 or an unaligned A (the 47-column output gradient) takes the DWORD
// form: lane (c, q) loads A[row0 + c][4 s + q], B staged to match.  No atomics, fixed summation order.
#include "common.h"
#include "slots.h"

namespace mgx {

constexpr int kRowsGemmLdsFloats = 16384;  // 64 KB: K_pad x M_pad of B
constexpr int kRgBlock = 512;              // 8 waves share one stage of B: twice the waves per LDS byte of a 256-thread group (the fp32
constexpr int kRgWaves = kRgBlock / kWave;  // MFMAs of a tile are ~3/4 of its memory time: other waves' loads must run under them)

__device__ __forceinline__ int rows_gemm_col(int mt, int c) { return 64 * (mt / 4) + 4 * c + (mt % 4); }

// relu + inverted dropout in the epilogue (ACT): the activation between two layers (main_dgl_product_sage.py:93-95) written straight
// into its destination with the 4 mask bits per float4, exactly as mgx_relu_dropout_fwd_strided would from the stored GEMM result --
// same random stream (splitmix64 of seed and DENSE float4 index), same arithmetic, so the same bits -- without writing and
// re-reading the N x M pre-activation.
struct RowsGemmAct {
  uint8_t* mask;        // [n * M / 4]
  uint32_t drop_below;  // keep when the 32 random bits are >= this
  float scale;          // 1 / (1 - p)
  uint64_t seed, offset;
  uint32_t* slots;      // optional (M == 64): the activation's rows as 128-byte slots as well (slots.h) -- the next layer's aggregation
                        // gathers those instead of the rows; the epilogue holds a row exactly as the stand-alone pack pass reads it
  unsigned long long* overflow;  // optional: += rows with more than 24 non-zeros
};

// KS = k-steps of 4 (K_pad / 4; a multiple of 4 in the float4 form), MT = column tiles of 16 (a multiple of 4: groups of 64 columns)
template <int KS, int MT, bool V4, bool ACT>
__global__ __launch_bounds__(kRgBlock) void rows_gemm_kernel(int64_t n, int K, int M, const float* __restrict__ A, int64_t lda,
                                                           const float* __restrict__ B, int64_t ldb, int b_transposed,
                                                           const float* __restrict__ bias, const float* __restrict__ row_scale,
                                                           int scale_from, float* __restrict__ C, int64_t ldc, RowsGemmAct act,
                                                           float* __restrict__ C2, int64_t ldc2, int split_col) {
  __shared__ float Bl[KS * MT * 64];
  __shared__ uint32_t slot_stage[ACT ? kRgWaves * 4 * 32 : 1];
  unsigned long long over_rows = 0;
  const int lane = threadIdx.x & (kWave - 1);
  const int c = lane % 16, q = lane / 16;
  // ---- stage B: element (step s, tile mt, lane) = B[k(s, q)][16 mt + c]; float4 form: s = 4 jj + t, k = 16 jj + 4 q + t, stored [jj][mt][lane][t]
  for (int idx = threadIdx.x; idx < KS * MT * 64; idx += kRgBlock) {
    int s, mt, ln;
    if (V4) {
      const int t = idx % 4, rest = idx / 4;
      ln = rest % 64;
      mt = (rest / 64) % MT;
      s = (rest / 64 / MT) * 4 + t;
    } else {
      ln = idx % 64;
      mt = (idx / 64) % MT;
      s = idx / 64 / MT;
    }
    const int k = V4 ? 16 * (s / 4) + 4 * (ln / 16) + (s % 4) : 4 * s + ln / 16;
    const int m = rows_gemm_col(mt, ln % 16);
    float v = 0.f;
    if (k < K && m < M) v = b_transposed ? B[(int64_t)m * ldb + k] : B[(int64_t)k * ldb + m];
    Bl[idx] = v;
  }
  __syncthreads();
  float bv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) bv[mt] = (bias && rows_gemm_col(mt, c) < M) ? bias[rows_gemm_col(mt, c)] : 0.f;
  // float4 stores; columns from split_col on (a multiple of 4) go to a matrix of their own (C2): the two halves of d[h | neigh] are
  // then separate compact matrices -- the reversed aggregation gathers from one and accumulates into the other (rows that share
  // 512-byte blocks with the rows being gathered cost it 2.60 against 2.38 ms)
  const bool cvec = ldc % 4 == 0 && ((uintptr_t)C % 16 == 0) && M % 4 == 0 && (!C2 || (ldc2 % 4 == 0 && (uintptr_t)C2 % 16 == 0));

  const int64_t gw = (int64_t)blockIdx.x * kRgWaves + threadIdx.x / kWave;
  const int64_t stride = (int64_t)gridDim.x * kRgWaves * 16;
  // operand fragment of one 16-row tile: a[s] pairs with step s of the stage (float4 form: a[4 jj + t] = component t of chunk jj)
  auto load_a = [&](int64_t r0, float (&a)[KS]) {
    const int64_t arow = r0 + c;
    const bool aok = arow < n;
    if (V4) {
#pragma unroll
      for (int jj = 0; jj < KS / 4; ++jj) {
        const int k0 = 16 * jj + 4 * q;
        const v4f v = (aok && k0 < K) ? *reinterpret_cast<const v4f*>(A + arow * lda + k0) : (v4f)(0.f);
        a[4 * jj + 0] = v.x; a[4 * jj + 1] = v.y; a[4 * jj + 2] = v.z; a[4 * jj + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s) a[s] = (aok && 4 * s + q < K) ? A[arow * lda + 4 * s + q] : 0.f;
    }
  };
  float a[KS], an[KS];
  int64_t r0 = gw * 16;
  if (r0 < n) load_a(r0, a);
  for (; r0 < n; r0 += stride) {
    if (r0 + stride < n) load_a(r0 + stride, an);  // the next tile's rows travel under this tile's MFMAs and stores
    v4f acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (v4f)(0.f);
    // B stays in LDS: an address the compiler cannot prove loop-invariant keeps it from hoisting the KS x MT fragments into registers
    // (199 - 276 VGPRs, two waves per SIMD or spills, against ~60 with one ds_read per MFMA group)
    int lo = lane;
    asm volatile("" : "+v"(lo));
    if (V4) {
#pragma unroll
      for (int jj = 0; jj < KS / 4; ++jj) {
        v4f b[MT];  // the MT column tiles of this k-chunk first, then MT INDEPENDENT accumulators between two uses of the same one
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b[mt] = *reinterpret_cast<const v4f*>(&Bl[((jj * MT + mt) * 64 + lo) * 4]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * jj + 0], b[mt].x, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * jj + 1], b[mt].y, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * jj + 2], b[mt].z, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * jj + 3], b[mt].w, acc[mt], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], Bl[(s * MT + mt) * 64 + lo], acc[mt], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) a[s] = an[s];
    // ---- epilogue: accumulator element r of tile mt = C[r0 + 4 q + r][rows_gemm_col(mt, c)]; the four tiles of a group = four consecutive columns
    float rs[4] = {1.f, 1.f, 1.f, 1.f};
    if (row_scale) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[r] = (r0 + 4 * q + r < n) ? row_scale[r0 + 4 * q + r] : 1.f;
    }
#pragma unroll
    for (int g = 0; g < MT / 4; ++g) {
      const int m0 = 64 * g + 4 * c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = r0 + 4 * q + r;
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          v[t] = acc[4 * g + t][r] + bv[4 * g + t];
          if (row_scale && m0 + t >= scale_from) v[t] *= rs[r];
        }
        if (ACT) {  // (the host checked: float4 stores possible, M % 4 == 0)
          v4f kept = (v4f)(0.f);
          if (row < n && m0 + 3 < M) {
            const uint64_t i = (uint64_t)row * (uint64_t)(M / 4) + (uint64_t)(m0 / 4);
            const uint64_t q0 = splitmix64(act.seed ^ ((act.offset + i) * 2));
            const uint64_t q1 = splitmix64(act.seed ^ ((act.offset + i) * 2 + 1));
            const bool k0 = (uint32_t)q0 >= act.drop_below && v[0] > 0.f, k1 = (uint32_t)(q0 >> 32) >= act.drop_below && v[1] > 0.f;
            const bool k2 = (uint32_t)q1 >= act.drop_below && v[2] > 0.f, k3 = (uint32_t)(q1 >> 32) >= act.drop_below && v[3] > 0.f;
            v4f o;
            o.x = k0 ? v[0] * act.scale : 0.f;
            o.y = k1 ? v[1] * act.scale : 0.f;
            o.z = k2 ? v[2] * act.scale : 0.f;
            o.w = k3 ? v[3] * act.scale : 0.f;
            *reinterpret_cast<v4f*>(C + row * ldc + m0) = o;
            act.mask[i] = (uint8_t)((k0 ? 1 : 0) | (k1 ? 2 : 0) | (k2 ? 4 : 0) | (k3 ? 8 : 0));
            kept = o;
          }
          if (act.slots) {  // wave-uniform; M == 64: lane (q, c) holds columns 4 c .. 4 c + 3 of row r0 + 4 q + r
            const bool over = slot_pack_rows4(kept, q, c, slot_stage + (threadIdx.x / kWave) * 128, row < n ? act.slots + row * 32 : nullptr);
            over_rows += (unsigned long long)__popcll(__ballot(over && c == 0 && row < n));
          }
        } else if (row < n) {
          float* dst = (C2 && m0 >= split_col) ? C2 + row * ldc2 + (m0 - split_col) : C + row * ldc + m0;
          if (cvec && m0 + 3 < M) {
            v4f o; o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
            *reinterpret_cast<v4f*>(dst) = o;
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (m0 + t < M) dst[t] = v[t];
          }
        }
      }
    }
  }
  if (ACT && act.overflow && lane == 0 && over_rows) atomicAdd(act.overflow, over_rows);
}

template <int KS, int MT, bool V4>
static void launch_rows_gemm(int64_t n, int K, int M, const float* A, int64_t lda, const float* B, int64_t ldb, int bt, const float* bias,
                             const float* rs, int scale_from, float* C, int64_t ldc, const RowsGemmAct* act, float* C2, int64_t ldc2,
                             int split_col, hipStream_t s) {
  // waves: one per 16 rows up to 6 per SIMD of the chip (the stage of B is paid once per workgroup)
  int64_t blocks = (n + 16 * kRgWaves - 1) / (16 * kRgWaves);
  const int64_t cap = 256 * 3;
  if (blocks > cap) blocks = cap;
  if (act) {
    hipLaunchKernelGGL((rows_gemm_kernel<KS, MT, V4, true>), dim3((unsigned)blocks), dim3(kRgBlock), 0, s, n, K, M, A, lda, B, ldb, bt, bias,
                       rs, scale_from, C, ldc, *act, C2, ldc2, split_col);
  } else {
    hipLaunchKernelGGL((rows_gemm_kernel<KS, MT, V4, false>), dim3((unsigned)blocks), dim3(kRgBlock), 0, s, n, K, M, A, lda, B, ldb, bt, bias,
                       rs, scale_from, C, ldc, RowsGemmAct{}, C2, ldc2, split_col);
  }
}

// shape -> kernel; false when none is built for it
static bool rows_gemm_dispatch(bool dry, int64_t n, int64_t K, int64_t M, const float* a, int64_t lda, const float* b, int64_t ldb, int bt,
                               const float* bias, const float* rs, int64_t scale_from, float* c, int64_t ldc, const RowsGemmAct* act,
                               hipStream_t s, float* c2 = nullptr, int64_t ldc2 = 0, int64_t split_col = 0) {
  const bool v4 = K % 4 == 0 && lda % 4 == 0 && (dry || (uintptr_t)a % 16 == 0);
  const int mt = (int)((M + 63) / 64) * 4;  // column tiles in groups of four (64 columns)
  const int ks = v4 ? (int)((K + 15) / 16) * 4 : (int)((K + 3) / 4);
  if (ks * mt * 64 > kRowsGemmLdsFloats) return false;
  bool ok = false;
#define MGX_RG(KS_, MT_, V4_)                                                                                                             \
  if (!ok && ks == KS_ && mt == MT_ && v4 == V4_) {                                                                                       \
    if (!dry) launch_rows_gemm<KS_, MT_, V4_>(n, (int)K, (int)M, a, lda, b, ldb, bt, bias, rs, (int)scale_from, c, ldc, act, c2, ldc2,     \
                                              (int)split_col, s);                                                                         \
    ok = true;                                                                                                                            \
  }
  // float4 form: K = 32 .. 208, M <= 64 / 128 -- the SAGE layers of the products model and of 64-wide models
  MGX_RG(8, 4, true) MGX_RG(8, 8, true) MGX_RG(12, 4, true) MGX_RG(12, 8, true) MGX_RG(16, 4, true) MGX_RG(16, 8, true)
  MGX_RG(32, 4, true) MGX_RG(32, 8, true) MGX_RG(52, 4, true)
  // dword form: K = 37 .. 48 (the 47-class output gradient of products, 41 of reddit, 40 of arxiv)
  MGX_RG(12, 8, false) MGX_RG(12, 4, false) MGX_RG(11, 8, false) MGX_RG(11, 4, false) MGX_RG(10, 8, false) MGX_RG(10, 4, false)
#undef MGX_RG
  return ok;
}

}  // namespace mgx

extern "C" int32_t mgx_rows_gemm(int64_t n, int64_t K, int64_t M, const float* a, int64_t lda, const float* b, int64_t ldb,
                                 int32_t b_transposed, const float* bias, const float* row_scale, int64_t scale_from, float* c,
                                 int64_t ldc, float* c2, int64_t ldc2, int64_t split_col, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && K >= 1 && M >= 1, "mgx_rows_gemm: bad sizes");
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(a && b && c, "mgx_rows_gemm: NULL pointer");
  MGX_CHECK_ARG(lda >= K && ldc >= (c2 ? split_col : M) && ldb >= (b_transposed ? K : M), "mgx_rows_gemm: leading dimensions smaller than the operands");
  MGX_CHECK_ARG(!c2 || (split_col > 0 && split_col < M && split_col % 4 == 0 && ldc2 >= M - split_col),
                "mgx_rows_gemm: a second output needs 0 < split_col < M, split_col % 4 == 0, ldc2 >= M - split_col");
  MGX_CHECK_ARG(scale_from >= 0 && scale_from <= M, "mgx_rows_gemm: scale_from outside [0, M]");
  if (!rows_gemm_dispatch(false, n, K, M, a, lda, b, ldb, b_transposed, bias, row_scale, scale_from, c, ldc, nullptr, (hipStream_t)stream, c2, ldc2,
                          split_col))
    MGX_UNSUPPORTED("mgx_rows_gemm: no kernel for K = %lld, M = %lld", (long long)K, (long long)M);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int32_t mgx_rows_gemm_supported(int64_t K, int64_t M, int64_t lda) {
  return mgx::rows_gemm_dispatch(true, 0, K, M, nullptr, lda, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr) ? 1 : 0;
}

extern "C" int32_t mgx_rows_gemm_relu_dropout(int64_t n, int64_t K, int64_t M, const float* a, int64_t lda, const float* b, int64_t ldb,
                                              int32_t b_transposed, const float* bias, float p, uint64_t seed, uint64_t offset, float* y,
                                              int64_t ldy, uint8_t* mask, void* slots, int64_t* overflow_rows, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && K >= 1 && M >= 1 && M % 4 == 0, "mgx_rows_gemm_relu_dropout: bad sizes (M must be a multiple of 4)");
  MGX_CHECK_ARG(p >= 0.f && p < 1.f, "mgx_rows_gemm_relu_dropout: p must be in [0, 1)");
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(a && b && y && mask, "mgx_rows_gemm_relu_dropout: NULL pointer");
  MGX_CHECK_ARG(lda >= K && ldy >= M && ldy % 4 == 0 && (uintptr_t)y % 16 == 0 && ldb >= (b_transposed ? K : M),
                "mgx_rows_gemm_relu_dropout: leading dimensions / alignment (y: 16 bytes, ldy % 4 == 0)");
  const double thr = (double)p * 4294967296.0;
  RowsGemmAct act;
  act.mask = mask;
  act.drop_below = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr;
  act.scale = 1.f / (1.f - p);
  act.seed = seed;
  act.offset = offset;
  MGX_CHECK_ARG(!slots || (M == 64 && (uintptr_t)slots % 16 == 0), "mgx_rows_gemm_relu_dropout: slots are for M == 64 (16-byte aligned)");
  act.slots = (uint32_t*)slots;
  act.overflow = slots ? (unsigned long long*)overflow_rows : nullptr;
  if (!rows_gemm_dispatch(false, n, K, M, a, lda, b, ldb, b_transposed, bias, nullptr, 0, y, ldy, &act, (hipStream_t)stream))
    MGX_UNSUPPORTED("mgx_rows_gemm_relu_dropout: no kernel for K = %lld, M = %lld", (long long)K, (long long)M);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
