// softmax.hip -- fused edge-softmax (norm_by='dst') forward/backward for gfx950 (MI355X).
//
// Replaces dgl.nn.functional.edge_softmax as reached through GATConv
// (main_dgl_reddit_gat.py:31-55).  DGL 0.6 composes it from copy_rhs/max SpMM, sub SDDMM + exp,
// copy_rhs/sum SpMM, div SDDMM -- four launches and five E*H-sized round trips.  Here one
// row-segmented kernel: a wavefront owns a destination row, lanes are laid out as
// (edge slot, head), the row's logits are read ONCE into registers when in-degree*LH <= 64*kCache
// (re-read from L2 otherwise), max and sum are combined by xor-shuffles, one write.
#include <math.h>

#include "common.h"

namespace mgx {

constexpr int kCache = 4;
constexpr int kSmRows = 64;

template <typename Idx>
struct SoftmaxArgs {
  const Idx* indptr;
  const Idx* eids;
  const float* x;   // fwd: z        bwd: a
  const float* y;   // fwd: unused   bwd: da
  float* out;       // fwd: a        bwd: dz
  int64_t n_rows;
  int64_t nblocks;
  int H;
};

// LH = lanes per edge (pow2 >= H); 64/LH edges per step.
template <typename Idx, int LH, bool BWD>
__global__ __launch_bounds__(kBlock) void edge_softmax_kernel(const SoftmaxArgs<Idx> a) {
  constexpr int EPI = kWave / LH;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int j = lane / LH, h = lane % LH;
  const bool hactive = h < a.H;
  const int64_t H = a.H;
  const int64_t row_base = xcd_remap(blockIdx.x, a.nblocks) * kSmRows;
  for (int r = wave; r < kSmRows; r += kWavesPerBlock) {
    const int64_t row = row_base + r;
    if (row >= a.n_rows) break;
    const int64_t beg = (int64_t)a.indptr[row], end = (int64_t)a.indptr[row + 1];
    if (beg == end) continue;
    int64_t es[kCache];
    float xs[kCache];
    float red = BWD ? 0.f : -INFINITY;
#pragma unroll
    for (int c = 0; c < kCache; ++c) {
      const int64_t p = beg + j + (int64_t)c * EPI;
      const bool ok = p < end && hactive;
      es[c] = ok ? (a.eids ? (int64_t)a.eids[p] : p) : -1;
    }
    if (!BWD) {
#pragma unroll
      for (int c = 0; c < kCache; ++c) {
        xs[c] = es[c] >= 0 ? a.x[es[c] * H + h] : -INFINITY;
        red = fmaxf(red, xs[c]);
      }
      for (int64_t p = beg + j + (int64_t)kCache * EPI; p < end; p += EPI)
        if (hactive) red = fmaxf(red, a.x[(a.eids ? (int64_t)a.eids[p] : p) * H + h]);
#pragma unroll
      for (int off = LH; off < kWave; off <<= 1) red = fmaxf(red, __shfl_xor(red, off, kWave));
      const float m = red;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < kCache; ++c) {
        xs[c] = es[c] >= 0 ? expf(xs[c] - m) : 0.f;
        s += xs[c];
      }
      for (int64_t p = beg + j + (int64_t)kCache * EPI; p < end; p += EPI)
        if (hactive) s += expf(a.x[(a.eids ? (int64_t)a.eids[p] : p) * H + h] - m);
#pragma unroll
      for (int off = LH; off < kWave; off <<= 1) s += __shfl_xor(s, off, kWave);
#pragma unroll
      for (int c = 0; c < kCache; ++c)
        if (es[c] >= 0) a.out[es[c] * H + h] = xs[c] / s;
      for (int64_t p = beg + j + (int64_t)kCache * EPI; p < end; p += EPI)
        if (hactive) {
          const int64_t e = a.eids ? (int64_t)a.eids[p] : p;
          a.out[e * H + h] = expf(a.x[e * H + h] - m) / s;
        }
    } else {
      float ys[kCache];
#pragma unroll
      for (int c = 0; c < kCache; ++c) {
        xs[c] = es[c] >= 0 ? a.x[es[c] * H + h] : 0.f;  // a
        ys[c] = es[c] >= 0 ? a.y[es[c] * H + h] : 0.f;  // da
        red += xs[c] * ys[c];
      }
      for (int64_t p = beg + j + (int64_t)kCache * EPI; p < end; p += EPI)
        if (hactive) {
          const int64_t e = a.eids ? (int64_t)a.eids[p] : p;
          red += a.x[e * H + h] * a.y[e * H + h];
        }
#pragma unroll
      for (int off = LH; off < kWave; off <<= 1) red += __shfl_xor(red, off, kWave);
#pragma unroll
      for (int c = 0; c < kCache; ++c)
        if (es[c] >= 0) a.out[es[c] * H + h] = xs[c] * ys[c] - xs[c] * red;
      for (int64_t p = beg + j + (int64_t)kCache * EPI; p < end; p += EPI)
        if (hactive) {
          const int64_t e = a.eids ? (int64_t)a.eids[p] : p;
          const float av = a.x[e * H + h];
          a.out[e * H + h] = av * a.y[e * H + h] - av * red;
        }
    }
  }
}

template <typename Idx, bool BWD>
static int32_t softmax_launch(const mgx_csr* csr, int64_t H, const float* x, const float* y, float* out,
                              hipStream_t s) {
  if (csr->num_rows == 0 || csr->nnz == 0 || H == 0) return MGX_OK;
  SoftmaxArgs<Idx> a;
  a.indptr = (const Idx*)csr->indptr; a.eids = (const Idx*)csr->eids; a.x = x; a.y = y; a.out = out;
  a.n_rows = csr->num_rows; a.H = (int)H;
  a.nblocks = round_up((csr->num_rows + kSmRows - 1) / kSmRows, kXcds);
  MGX_CHECK_ARG(a.nblocks < (int64_t(1) << 31), "mgx_edge_softmax: too many rows");
  int LH = 1;
  while (LH < H) LH <<= 1;
  dim3 grid((unsigned)a.nblocks), block(kBlock);
  switch (LH) {
    case 1: hipLaunchKernelGGL((edge_softmax_kernel<Idx, 1, BWD>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((edge_softmax_kernel<Idx, 2, BWD>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((edge_softmax_kernel<Idx, 4, BWD>), grid, block, 0, s, a); break;
    case 8: hipLaunchKernelGGL((edge_softmax_kernel<Idx, 8, BWD>), grid, block, 0, s, a); break;
    case 16: hipLaunchKernelGGL((edge_softmax_kernel<Idx, 16, BWD>), grid, block, 0, s, a); break;
    case 32: hipLaunchKernelGGL((edge_softmax_kernel<Idx, 32, BWD>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((edge_softmax_kernel<Idx, 64, BWD>), grid, block, 0, s, a); break;
  }
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

static int32_t softmax_check(const mgx_csr* csr, int64_t H) {
  MGX_CHECK_ARG(csr != nullptr, "mgx_edge_softmax: csr is NULL");
  MGX_CHECK_ARG(csr->idx_bits == 32 || csr->idx_bits == 64, "mgx_edge_softmax: idx_bits must be 32 or 64");
  MGX_CHECK_ARG(H >= 0, "mgx_edge_softmax: negative H");
  if (H > kWave) MGX_UNSUPPORTED("mgx_edge_softmax: H = %lld > 64 heads is not supported", (long long)H);
  MGX_CHECK_ARG(csr->num_rows == 0 || csr->indptr, "mgx_edge_softmax: indptr is NULL");
  return MGX_OK;
}

}  // namespace mgx

extern "C" int32_t mgx_edge_softmax_fwd(const mgx_csr* csr, int64_t H, const float* z, float* a, void* stream) {
  using namespace mgx;
  int32_t st = softmax_check(csr, H);
  if (st != MGX_OK) return st;
  MGX_CHECK_ARG(csr->nnz == 0 || H == 0 || (z && a), "mgx_edge_softmax_fwd: z/a is NULL");
  if (csr->idx_bits == 32) return softmax_launch<int32_t, false>(csr, H, z, nullptr, a, (hipStream_t)stream);
  return softmax_launch<int64_t, false>(csr, H, z, nullptr, a, (hipStream_t)stream);
}

extern "C" int32_t mgx_edge_softmax_bwd(const mgx_csr* csr, int64_t H, const float* a, const float* da, float* dz,
                                        void* stream) {
  using namespace mgx;
  int32_t st = softmax_check(csr, H);
  if (st != MGX_OK) return st;
  MGX_CHECK_ARG(csr->nnz == 0 || H == 0 || (a && da && dz), "mgx_edge_softmax_bwd: a/da/dz is NULL");
  if (csr->idx_bits == 32) return softmax_launch<int32_t, true>(csr, H, a, da, dz, (hipStream_t)stream);
  return softmax_launch<int64_t, true>(csr, H, a, da, dz, (hipStream_t)stream);
}
