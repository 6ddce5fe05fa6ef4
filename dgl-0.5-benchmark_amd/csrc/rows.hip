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

extern "C" int32_t mgx_gather_rows(int64_t n, const void* idx, int32_t idx_bits, int64_t D, const float* x, float* out,
                                   void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(idx_bits == 32 || idx_bits == 64, "mgx_gather_rows: idx_bits must be 32 or 64");
  MGX_CHECK_ARG(n >= 0 && D >= 0, "mgx_gather_rows: negative sizes");
  MGX_CHECK_ARG(n == 0 || D == 0 || (idx && x && out), "mgx_gather_rows: NULL pointer");
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
