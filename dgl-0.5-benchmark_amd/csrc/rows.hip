// rows.hip -- boundary-row pack / unpack for the multi-GPU halo exchange (new capability; the
// reference is single-GPU).  gather_rows packs X[send_idx] into the contiguous RCCL all_to_all
// send buffer; scatter_add_rows adds received gradient rows into their owners.  `idx` must be
// unique within one scatter call (the halo plan guarantees it), so no atomics are needed and the
// result is deterministic.  Lanes run along the feature dimension with 16-byte accesses.
#include "common.h"

namespace mgx {

template <typename Idx, int VEC, bool SCATTER>
__global__ __launch_bounds__(kBlock) void rows_kernel(int64_t n, const Idx* idx, int64_t D, const float* in,
                                                      float* out) {
  typedef typename VecT<VEC>::type V;
  const int64_t chunks = D / VEC;
  const int64_t total = n * chunks;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t i = t / chunks, c = (t % chunks) * VEC;
    const int64_t r = (int64_t)idx[i];
    if (SCATTER) {
      V* dst = reinterpret_cast<V*>(out + r * D + c);
      *dst = *dst + *reinterpret_cast<const V*>(in + i * D + c);
    } else {
      *reinterpret_cast<V*>(out + i * D + c) = *reinterpret_cast<const V*>(in + r * D + c);
    }
  }
}

template <typename Idx, bool SCATTER>
static int32_t rows_launch(int64_t n, const void* idx, int64_t D, const float* in, float* out, hipStream_t s) {
  if (n == 0 || D == 0) return MGX_OK;
  const bool al16 = (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0;
  const int vec = (D % 4 == 0 && al16) ? 4 : 1;
  const int64_t total = n * (D / vec);
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (vec == 4) hipLaunchKernelGGL((rows_kernel<Idx, 4, SCATTER>), dim3((unsigned)blocks), dim3(kBlock), 0, s, n, (const Idx*)idx, D, in, out);
  else hipLaunchKernelGGL((rows_kernel<Idx, 1, SCATTER>), dim3((unsigned)blocks), dim3(kBlock), 0, s, n, (const Idx*)idx, D, in, out);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

}  // namespace mgx

namespace mgx {
// Row gather with row strides (round 4): the pack of a partition's boundary rows reads the left half of a layer's [h | neigh]
// buffer in place.  A lane group of G lanes (G * 16 bytes >= a row pass) owns a row; every lane keeps U rows in flight (one index
// load + one 16-byte gather each) and writes with non-temporal stores: the packed buffer is read next by the NIC / the peer, not
// by this CU.  ~1 M rows of 256 bytes: torch.index_select 208 us, this kernel: see docs/LOG_r04.md.
template <typename Idx, int G>
__global__ __launch_bounds__(kBlock) void gather_rows_strided_kernel(int64_t n, const Idx* __restrict__ idx, int D, int64_t ldx,
                                                                     const float* __restrict__ x, int64_t ldo, float* __restrict__ out) {
  constexpr int U = 4;
  constexpr int RPB = kBlock / G;  // rows per block-instruction
  const int l = threadIdx.x % G, sub = threadIdx.x / G;
  const int c = (blockIdx.y * G + l) * 4;
  if (c >= D) return;
  for (int64_t base = (int64_t)blockIdx.x * (RPB * U); base < n; base += (int64_t)gridDim.x * (RPB * U)) {
    v4f v[U];
    int64_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      row[u] = base + u * RPB + sub;
      const int64_t r = row[u] < n ? (int64_t)idx[row[u]] : 0;
      v[u] = *reinterpret_cast<const v4f*>(x + r * ldx + c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (row[u] < n) __builtin_nontemporal_store(v[u], reinterpret_cast<v4f*>(out + row[u] * ldo + c));
  }
}

template <typename Idx>
static int32_t gather_rows_strided_launch(int64_t n, const void* idx, int64_t D, const float* x, int64_t ldx, float* out, int64_t ldo,
                                          hipStream_t s) {
  int G = 1;
  while (G * 4 < D && G < 64) G <<= 1;
  const unsigned passes = (unsigned)((D / 4 + G - 1) / G);
  const int rpb = kBlock / G * 4;
  int64_t blocks = (n + rpb - 1) / rpb;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const dim3 grid((unsigned)blocks, passes);
#define MGX_GR(GG) hipLaunchKernelGGL((gather_rows_strided_kernel<Idx, GG>), grid, dim3(kBlock), 0, s, n, (const Idx*)idx, (int)D, ldx, x, ldo, out)
  switch (G) {
    case 1: MGX_GR(1); break;
    case 2: MGX_GR(2); break;
    case 4: MGX_GR(4); break;
    case 8: MGX_GR(8); break;
    case 16: MGX_GR(16); break;
    case 32: MGX_GR(32); break;
    default: MGX_GR(64); break;
  }
#undef MGX_GR
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
}  // namespace mgx

extern "C" int32_t mgx_gather_rows_strided(int64_t n, const void* idx, int32_t idx_bits, int64_t D, const float* x, int64_t x_stride,
                                           float* out, int64_t out_stride, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(idx_bits == 32 || idx_bits == 64, "mgx_gather_rows_strided: idx_bits must be 32 or 64");
  MGX_CHECK_ARG(n >= 0 && D >= 0 && x_stride >= D && out_stride >= D, "mgx_gather_rows_strided: negative sizes or strides below D");
  if (n == 0 || D == 0) return MGX_OK;
  MGX_CHECK_ARG(idx && x && out, "mgx_gather_rows_strided: NULL pointer");
  if (D % 4 || x_stride % 4 || out_stride % 4 || (uintptr_t)x % 16 || (uintptr_t)out % 16 || D > (int64_t(1) << 20))
    MGX_UNSUPPORTED("mgx_gather_rows_strided: D and both strides must be multiples of 4 floats, pointers 16-byte aligned");
  if (idx_bits == 32) return gather_rows_strided_launch<int32_t>(n, idx, D, x, x_stride, out, out_stride, (hipStream_t)stream);
  return gather_rows_strided_launch<int64_t>(n, idx, D, x, x_stride, out, out_stride, (hipStream_t)stream);
}

extern "C" int32_t mgx_gather_rows(int64_t n, const void* idx, int32_t idx_bits, int64_t D, const float* x, float* out,
                                   void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(idx_bits == 32 || idx_bits == 64, "mgx_gather_rows: idx_bits must be 32 or 64");
  MGX_CHECK_ARG(n >= 0 && D >= 0, "mgx_gather_rows: negative sizes");
  MGX_CHECK_ARG(n == 0 || D == 0 || (idx && x && out), "mgx_gather_rows: NULL pointer");
  if (n > 0 && D > 0 && D % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 && D <= (int64_t(1) << 20))
    return mgx_gather_rows_strided(n, idx, idx_bits, D, x, D, out, D, stream);
  if (idx_bits == 32) return rows_launch<int32_t, false>(n, idx, D, x, out, (hipStream_t)stream);
  return rows_launch<int64_t, false>(n, idx, D, x, out, (hipStream_t)stream);
}

extern "C" int32_t mgx_scatter_add_rows(int64_t n, const void* idx, int32_t idx_bits, int64_t D, const float* in,
                                        float* x, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(idx_bits == 32 || idx_bits == 64, "mgx_scatter_add_rows: idx_bits must be 32 or 64");
  MGX_CHECK_ARG(n >= 0 && D >= 0, "mgx_scatter_add_rows: negative sizes");
  MGX_CHECK_ARG(n == 0 || D == 0 || (idx && x && in), "mgx_scatter_add_rows: NULL pointer");
  if (idx_bits == 32) return rows_launch<int32_t, true>(n, idx, D, in, x, (hipStream_t)stream);
  return rows_launch<int64_t, true>(n, idx, D, in, x, (hipStream_t)stream);
}
