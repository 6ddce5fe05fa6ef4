// common.h -- shared host/device helpers of the MI355X graph library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mi355x_graph.h"

namespace mgx {

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kBlock = 256;        // 4 waves / workgroup: one per SIMD
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kXcds = 8;           // MI355X: 8 XCDs, each with its own 4 MiB L2

// (The C library reads NO environment variables: every choice between kernels is made from the arguments of a call.  The A/B switches
// of rounds 1 - 4 went with the variants that lost -- docs/LOG_r0*.md has their numbers.)

void set_error(const char* fmt, ...);
void note_spmm_kernel(const char* name);  // mgx_last_spmm_kernel(): which kernel family a g-SpMM call was routed to

#define MGX_CHECK_ARG(cond, ...)                    \
  do {                                              \
    if (!(cond)) {                                  \
      ::mgx::set_error(__VA_ARGS__);                \
      return MGX_ERR_INVALID_ARGUMENT;              \
    }                                               \
  } while (0)

#define MGX_UNSUPPORTED(...)                        \
  do {                                              \
    ::mgx::set_error(__VA_ARGS__);                  \
    return MGX_ERR_UNSUPPORTED;                     \
  } while (0)

#define MGX_CHECK_HIP(expr)                                                         \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) {                                                         \
      ::mgx::set_error("HIP error %d (%s) at %s:%d", (int)e_, hipGetErrorString(e_), \
                       __FILE__, __LINE__);                                         \
      return MGX_ERR_HIP;                                                           \
    }                                                                               \
  } while (0)

// First statement of every device entry point: drops a stale (sticky) error left by the CALLER's earlier HIP calls so
// that MGX_CHECK_LAUNCH reports only this library's launches.
#define MGX_ENTER() (void)hipGetLastError()

// Called after every kernel launch: reports launch-configuration errors without synchronising.
#define MGX_CHECK_LAUNCH() MGX_CHECK_HIP(hipGetLastError())

// counter-based random bits of relu+dropout (elementwise.hip, rowsgemm.hip): a function of (seed, element index) only
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef v2f type; };
template <> struct VecT<4> { typedef v4f type; };

// Blocks b and b+8 are observed to land on the same XCD (round-robin dispatch).  Give each XCD a
// CONTIGUOUS range of logical blocks so that neighbouring destination rows -- which share
// source rows on graphs with locality -- hit the same 4 MiB L2.  Speed only, never correctness.
// `nb` must be a multiple of kXcds.
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t nb) {
  return (b % kXcds) * (nb / kXcds) + (b / kXcds);
}

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int ilog2_ceil(int64_t x) { int l = 0; while ((int64_t(1) << l) < x) ++l; return l; }

// Per-XCD stretches of a schedule of `n_items` work items, `rpb` items per workgroup.  Block b serves XCD b % 8 (observed
// round-robin placement; speed only) and takes the (b / 8)-th group of rpb items of that XCD's stretch.  Stretch boundaries:
// the plan's edge-balanced ones (read from device memory) or equal item counts (`per` items each).
struct XcdRanges {
  const int64_t* start_dev;  // [9] or NULL
  int64_t per;               // equal split: items per XCD (a multiple of rpb)
  int64_t n_items;
};
static inline int64_t xcd_ranges(const mgx_spmm_plan* plan, int64_t n_items, int rpb, XcdRanges& r) {
  const bool have = plan && plan->xcd_item_start_dev && plan->xcd_item_start[kXcds] == n_items && plan->xcd_item_start[0] == 0;
  r.per = round_up((n_items + kXcds - 1) / kXcds, rpb);
  r.n_items = n_items;
  r.start_dev = have ? plan->xcd_item_start_dev : nullptr;
  int64_t most = (r.per + rpb - 1) / rpb;
  if (have) {
    most = 0;
    for (int x = 0; x < kXcds; ++x) {
      const int64_t len = plan->xcd_item_start[x + 1] - plan->xcd_item_start[x];
      const int64_t blocks = ((len > 0 ? len : 0) + rpb - 1) / rpb;
      if (blocks > most) most = blocks;
    }
  }
  return (most > 0 ? most : 1) * kXcds;  // grid size
}
__device__ __forceinline__ void xcd_stretch(const XcdRanges& r, int64_t& first, int64_t& stop) {
  const int xcd = blockIdx.x % kXcds;
  if (r.start_dev) {
    first = r.start_dev[xcd];
    stop = r.start_dev[xcd + 1];
  } else {
    first = (int64_t)xcd * r.per;
    stop = first + r.per;
    if (first > r.n_items) first = r.n_items;
    if (stop > r.n_items) stop = r.n_items;
  }
}

template <typename T>
__device__ __forceinline__ T wave_shfl_xor(T v, int mask) { return __shfl_xor(v, mask, kWave); }

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vec_shfl_xor(typename VecT<VEC>::type v, int mask);
template <> __device__ __forceinline__ float vec_shfl_xor<1>(float v, int mask) { return __shfl_xor(v, mask, kWave); }
template <> __device__ __forceinline__ v2f vec_shfl_xor<2>(v2f v, int mask) {
  v2f r; r.x = __shfl_xor(v.x, mask, kWave); r.y = __shfl_xor(v.y, mask, kWave); return r;
}
template <> __device__ __forceinline__ v4f vec_shfl_xor<4>(v4f v, int mask) {
  v4f r;
  r.x = __shfl_xor(v.x, mask, kWave); r.y = __shfl_xor(v.y, mask, kWave);
  r.z = __shfl_xor(v.z, mask, kWave); r.w = __shfl_xor(v.w, mask, kWave);
  return r;
}

// ---- cross-lane sums without the LDS pipe ------------------------------------------------------------------------------------
// __shfl_xor lowers to ds_bpermute_b32 on gfx950 (an LDS-pipe instruction, ~100 cycles of latency in a dependent chain).  Inside a
// 16-lane DPP row the same sums are VALU moves: quad_perm swaps for distances 1 and 2, half-row / row mirrors for 4 and 8.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// x summed over the aligned group of L consecutive lanes (L a power of two), in every lane of the group
template <int L>
__device__ __forceinline__ float lanes_sum(float x) {
  if (L >= 2) x += dpp_mov<0xB1>(x);    // quad_perm [1, 0, 3, 2]
  if (L >= 4) x += dpp_mov<0x4E>(x);    // quad_perm [2, 3, 0, 1]
  if (L >= 8) x += dpp_mov<0x141>(x);   // row_half_mirror: every lane of a quad holds the quad's sum, the mirrored lane is in the other quad
  if (L >= 16) x += dpp_mov<0x140>(x);  // row_mirror: ... in the other half row
#pragma unroll
  for (int off = 16; off < L; off <<= 1) x += __shfl_xor(x, off, kWave);
  return x;
}
}  // namespace mgx
