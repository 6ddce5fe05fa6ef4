// elementwise.hip -- relu followed by (inverted) dropout in one pass each way.
//
// Between two aggregations GraphSAGE applies `x = F.relu(x); x = dropout(x)` (main_dgl_product_sage.py:93-95) on an
// [N, hidden] activation: four PyTorch kernels per layer and direction-pair moving 38 bytes per element (relu r4 w4,
// dropout r4 w4 + mask, masked_scale r4 r1 w4, threshold_backward r4 r4 w4).  Fused: forward reads x, writes y and 4 mask
// bits per float4; backward reads dy and the bits, writes dx -- 16.5 bytes per element.  The mask holds relu AND keep, so
// backward needs neither x nor y.  Random bits: counter-based (splitmix64 of seed and element index), reproducible for a
// given (seed, offset) whatever the launch shape.
#include "common.h"
#include "slots.h"

namespace mgx {

// Rows of c4 float4s; x rows are ldx4 float4s apart, y rows ldy4 (dense: c4 = n4, one row).  The mask and the random stream
// are indexed by the DENSE element number, so a strided call draws the same mask as the dense one.
__global__ __launch_bounds__(kBlock) void relu_dropout_fwd_kernel(int64_t n4, const v4f* __restrict__ x, v4f* __restrict__ y,
                                                                  uint8_t* __restrict__ mask, uint32_t drop_below, float scale,
                                                                  uint64_t seed, uint64_t offset, int64_t c4, int64_t ldx4, int64_t ldy4,
                                                                  const uint64_t* __restrict__ counter) {
  // counter (device memory, may be NULL): the call's position in the random stream is read at RUN time -- a HIP graph that
  // replays this launch draws a new mask every replay once the graph also increments the counter
  if (counter) offset = (*counter * 0x9E3779B97F4A7C15ull) & 0x7FFFFFFFFFFFFFFFull;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / c4, c = i - r * c4;
    const v4f v = __builtin_nontemporal_load(&x[r * ldx4 + c]);
    const uint64_t r0 = splitmix64(seed ^ ((offset + (uint64_t)i) * 2));
    const uint64_t r1 = splitmix64(seed ^ ((offset + (uint64_t)i) * 2 + 1));
    const bool k0 = (uint32_t)r0 >= drop_below && v.x > 0.f, k1 = (uint32_t)(r0 >> 32) >= drop_below && v.y > 0.f;
    const bool k2 = (uint32_t)r1 >= drop_below && v.z > 0.f, k3 = (uint32_t)(r1 >> 32) >= drop_below && v.w > 0.f;
    v4f o;
    o.x = k0 ? v.x * scale : 0.f;
    o.y = k1 ? v.y * scale : 0.f;
    o.z = k2 ? v.z * scale : 0.f;
    o.w = k3 ? v.w * scale : 0.f;
    y[r * ldy4 + c] = o;
    mask[i] = (uint8_t)((k0 ? 1 : 0) | (k1 ? 2 : 0) | (k2 ? 4 : 0) | (k3 ? 8 : 0));
  }
}

__global__ __launch_bounds__(kBlock) void relu_dropout_bwd_kernel(int64_t n4, const v4f* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                                  v4f* __restrict__ dx, float scale, int64_t c4, int64_t lddy4,
                                                                  int64_t lddx4) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / c4, c = i - r * c4;
    const v4f g = __builtin_nontemporal_load(&dy[r * lddy4 + c]);
    const uint8_t m = mask[i];
    v4f o;
    o.x = (m & 1) ? g.x * scale : 0.f;
    o.y = (m & 2) ? g.y * scale : 0.f;
    o.z = (m & 4) ? g.z * scale : 0.f;
    o.w = (m & 8) ? g.w * scale : 0.f;
    dx[r * lddx4 + c] = o;
  }
}

// The same for rows of exactly 64 columns, and the result's rows -- times row_scale[r] when given -- as 128-byte slots as well
// (slots.h): the gradient behind relu + dropout is as sparse as the activation was, and the reversed aggregation that follows gathers
// the slots (mgx_spmm_copy_u_slots).  A wave holds four rows exactly as the stand-alone pack pass reads them.
__global__ __launch_bounds__(kBlock) void relu_dropout_bwd_slots_kernel(int64_t rows, const v4f* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                                        v4f* __restrict__ dx, float scale, int64_t lddy4, int64_t lddx4,
                                                                        const float* __restrict__ row_scale, uint32_t* __restrict__ slots,
                                                                        unsigned long long* __restrict__ overflow) {
  __shared__ uint32_t stage_all[kWavesPerBlock][4 * 32];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int sub = lane >> 4, l = lane & 15;
  const int64_t rows_per_pass = (int64_t)gridDim.x * kWavesPerBlock * 4;
  unsigned long long over_rows = 0;
  for (int64_t base = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * 4; base < rows; base += rows_per_pass) {
    const int64_t r = base + sub;
    v4f o = (v4f)(0.f);
    if (r < rows) {
      const v4f g = __builtin_nontemporal_load(&dy[r * lddy4 + l]);
      const uint8_t m = mask[r * 16 + l];
      o.x = (m & 1) ? g.x * scale : 0.f;
      o.y = (m & 2) ? g.y * scale : 0.f;
      o.z = (m & 4) ? g.z * scale : 0.f;
      o.w = (m & 8) ? g.w * scale : 0.f;
      dx[r * lddx4 + l] = o;
      if (row_scale) o *= row_scale[r];
    }
    const bool over = slot_pack_rows4(o, sub, l, stage_all[wave], r < rows ? slots + r * 32 : nullptr);
    over_rows += (unsigned long long)__popcll(__ballot(over && l == 0 && r < rows));
  }
  if (overflow && lane == 0 && over_rows) atomicAdd(overflow, over_rows);
}

static unsigned ew_grid(int64_t n4) {
  int64_t b = (n4 + kBlock - 1) / kBlock;
  if (b > 256 * 32) b = 256 * 32;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace mgx

extern "C" int32_t mgx_relu_dropout_fwd(int64_t n, const float* x, float p, uint64_t seed, uint64_t offset, float* y,
                                        uint8_t* mask, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && n % 4 == 0, "mgx_relu_dropout_fwd: the element count must be a multiple of 4 (got %lld)", (long long)n);
  MGX_CHECK_ARG(p >= 0.f && p < 1.f, "mgx_relu_dropout_fwd: p must be in [0, 1)");
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(x && y && mask && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0, "mgx_relu_dropout_fwd: NULL or unaligned pointer");
  const double thr = (double)p * 4294967296.0;
  const uint32_t drop_below = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr;
  hipLaunchKernelGGL(relu_dropout_fwd_kernel, dim3(ew_grid(n / 4)), dim3(kBlock), 0, (hipStream_t)stream, n / 4, (const v4f*)x, (v4f*)y,
                     mask, drop_below, 1.f / (1.f - p), seed, offset, n / 4, n / 4, n / 4, (const uint64_t*)nullptr);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int32_t mgx_relu_dropout_bwd(int64_t n, const float* dy, const uint8_t* mask, float p, float* dx, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && n % 4 == 0, "mgx_relu_dropout_bwd: the element count must be a multiple of 4");
  MGX_CHECK_ARG(p >= 0.f && p < 1.f, "mgx_relu_dropout_bwd: p must be in [0, 1)");
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(dy && dx && mask && (uintptr_t)dy % 16 == 0 && (uintptr_t)dx % 16 == 0, "mgx_relu_dropout_bwd: NULL or unaligned pointer");
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(ew_grid(n / 4)), dim3(kBlock), 0, (hipStream_t)stream, n / 4, (const v4f*)dy, mask,
                     (v4f*)dx, 1.f / (1.f - p), n / 4, n / 4, n / 4);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

// Row-strided forms: x / y (dy / dx) are [rows, cols] views whose rows are *_stride floats apart -- column blocks of wider
// matrices (the activation written straight into the left half of the next layer's [h | neigh] GEMM operand).  cols and the
// strides multiples of 4, 16-byte aligned pointers; the mask is dense, [rows * cols / 4].
extern "C" int32_t mgx_relu_dropout_fwd_strided(int64_t rows, int64_t cols, const float* x, int64_t x_stride, float p, uint64_t seed,
                                                uint64_t offset, float* y, int64_t y_stride, uint8_t* mask, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(rows >= 0 && cols >= 0 && cols % 4 == 0 && x_stride % 4 == 0 && y_stride % 4 == 0 && x_stride >= cols && y_stride >= cols,
                "mgx_relu_dropout_fwd_strided: cols and strides must be multiples of 4, strides >= cols");
  MGX_CHECK_ARG(p >= 0.f && p < 1.f, "mgx_relu_dropout_fwd_strided: p must be in [0, 1)");
  const int64_t n = rows * cols;
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(x && y && mask && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0, "mgx_relu_dropout_fwd_strided: NULL or unaligned pointer");
  const double thr = (double)p * 4294967296.0;
  const uint32_t drop_below = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr;
  hipLaunchKernelGGL(relu_dropout_fwd_kernel, dim3(ew_grid(n / 4)), dim3(kBlock), 0, (hipStream_t)stream, n / 4, (const v4f*)x, (v4f*)y,
                     mask, drop_below, 1.f / (1.f - p), seed, offset, cols / 4, x_stride / 4, y_stride / 4, (const uint64_t*)nullptr);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int32_t mgx_relu_dropout_bwd_strided(int64_t rows, int64_t cols, const float* dy, int64_t dy_stride, const uint8_t* mask,
                                                float p, float* dx, int64_t dx_stride, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(rows >= 0 && cols >= 0 && cols % 4 == 0 && dy_stride % 4 == 0 && dx_stride % 4 == 0 && dy_stride >= cols && dx_stride >= cols,
                "mgx_relu_dropout_bwd_strided: cols and strides must be multiples of 4, strides >= cols");
  MGX_CHECK_ARG(p >= 0.f && p < 1.f, "mgx_relu_dropout_bwd_strided: p must be in [0, 1)");
  const int64_t n = rows * cols;
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(dy && dx && mask && (uintptr_t)dy % 16 == 0 && (uintptr_t)dx % 16 == 0, "mgx_relu_dropout_bwd_strided: NULL or unaligned pointer");
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(ew_grid(n / 4)), dim3(kBlock), 0, (hipStream_t)stream, n / 4, (const v4f*)dy, mask,
                     (v4f*)dx, 1.f / (1.f - p), cols / 4, dy_stride / 4, dx_stride / 4);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int32_t mgx_relu_dropout_bwd_slots(int64_t rows, const float* dy, int64_t dy_stride, const uint8_t* mask, float p, float* dx,
                                              int64_t dx_stride, const float* row_scale, void* slots, int64_t* overflow_rows, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(rows >= 0 && dy_stride % 4 == 0 && dx_stride % 4 == 0 && dy_stride >= 64 && dx_stride >= 64,
                "mgx_relu_dropout_bwd_slots: rows of 64 columns, strides multiples of 4 and >= 64");
  MGX_CHECK_ARG(p >= 0.f && p < 1.f, "mgx_relu_dropout_bwd_slots: p must be in [0, 1)");
  if (rows == 0) return MGX_OK;
  MGX_CHECK_ARG(dy && dx && mask && slots && (uintptr_t)dy % 16 == 0 && (uintptr_t)dx % 16 == 0 && (uintptr_t)slots % 16 == 0,
                "mgx_relu_dropout_bwd_slots: NULL or unaligned pointer");
  int64_t blocks = (rows + 4 * kWavesPerBlock - 1) / (4 * kWavesPerBlock);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(relu_dropout_bwd_slots_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, rows, (const v4f*)dy, mask,
                     (v4f*)dx, 1.f / (1.f - p), dy_stride / 4, dx_stride / 4, row_scale, (uint32_t*)slots, (unsigned long long*)overflow_rows);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

// As mgx_relu_dropout_fwd_strided, but the offset into the random stream is `*counter * 0x9E3779B97F4A7C15 mod 2^63`, read on the
// device when the launch runs: captured in a HIP graph together with an increment of the counter, every replay draws a new mask
// (the host-side offset of the plain entry points is a launch argument and would be frozen by the capture).
extern "C" int32_t mgx_relu_dropout_fwd_counter(int64_t rows, int64_t cols, const float* x, int64_t x_stride, float p, uint64_t seed,
                                                const uint64_t* counter, float* y, int64_t y_stride, uint8_t* mask, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(rows >= 0 && cols >= 0 && cols % 4 == 0 && x_stride % 4 == 0 && y_stride % 4 == 0 && x_stride >= cols && y_stride >= cols,
                "mgx_relu_dropout_fwd_counter: cols and strides must be multiples of 4, strides >= cols");
  MGX_CHECK_ARG(p >= 0.f && p < 1.f, "mgx_relu_dropout_fwd_counter: p must be in [0, 1)");
  MGX_CHECK_ARG(counter != nullptr, "mgx_relu_dropout_fwd_counter: counter is NULL");
  const int64_t n = rows * cols;
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(x && y && mask && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0, "mgx_relu_dropout_fwd_counter: NULL or unaligned pointer");
  const double thr = (double)p * 4294967296.0;
  const uint32_t drop_below = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr;
  hipLaunchKernelGGL(relu_dropout_fwd_kernel, dim3(ew_grid(n / 4)), dim3(kBlock), 0, (hipStream_t)stream, n / 4, (const v4f*)x, (v4f*)y,
                     mask, drop_below, 1.f / (1.f - p), seed, (uint64_t)0, cols / 4, x_stride / 4, y_stride / 4, counter);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
