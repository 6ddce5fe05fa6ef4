// rowpack.hip -- halo rows as bitmaps + packed values (round 5; the multi-GPU exchange of mi355x_graph/dist.py).
//
// The hidden layers of the reference's GraphSAGE feed relu + dropout(0.5) outputs into the next aggregation
// (main_dgl_product_sage.py:93-96): ~75 % of a boundary row is exact zeros.  A row therefore crosses xGMI as a 64-bit mask per 64
// columns plus its non-zero values (72 bytes on average instead of 256 at D = 64), and the gradient of a halo row comes back as
// the values under that SAME mask only (64 bytes, no mask): the owner multiplies whatever arrives at a zero position by relu's /
// dropout's zero anyway, so nothing that is dropped was ever used.  Exact in fp32: values are moved, never rounded.
//
// Bit order.  Lanes run along the feature dimension with 16-byte accesses as everywhere in this library: lane l of the G = min(16,
// D / 4) lanes of a row block holds columns 4 l .. 4 l + 3 of a 64-column block.  Bit (c * G + l) of the block's mask stands for
// column 4 l + c -- the order four wave ballots deliver -- and the packed values of a row are stored in increasing bit order,
// block after block.  Pack and unpack agree on it; nobody else reads the packed form.
//
//   mgx_rows_pack_count    masks + per-row counts of x[idx[i], :] != 0
//   mgx_rows_mask_count    per-row counts of given masks (the receiving side)
//   mgx_rows_pack_values   values of x[idx[i], :] under given masks, row i from offsets[i] on
//   mgx_rows_unpack        dense rows (zeros elsewhere) from masks + offsets + values
// All streaming passes at one 16-byte access per lane; rows of 4 .. 256 columns, D % 4 == 0.
#include "common.h"

namespace mgx {
namespace {

constexpr int kMaxBlocks = 4;  // 64-column blocks per row: D <= 256

// lanes per row block and rows per wave for a row width
struct RowShape {
  int G;        // lanes per 64-column block (16, or D / 4 rounded up to a power of two when D < 64)
  int nblk;     // ceil(D / 64)
};
static inline RowShape row_shape(int64_t D) {
  RowShape s;
  s.nblk = (int)((D + 63) / 64);
  int g = 1;
  while (g * 4 < D && g < 16) g <<= 1;
  s.G = g;
  return s;
}

// 16 c + l bit of the wave-wide ballots belongs to (row group `sub`, lane l, component c): the mask of MY row group's block
template <int G>
__device__ __forceinline__ uint64_t group_mask(const uint64_t (&b)[4], int sub) {
  constexpr uint64_t ones = G >= 64 ? ~0ull : ((1ull << G) - 1ull);
  uint64_t m = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) m |= ((b[c] >> (sub * G)) & ones) << (c * G);
  return m;
}

// MODE 0: masks + counts from x != 0.   MODE 1: values of x under given masks.
template <typename Idx, int G, int MODE>
__global__ __launch_bounds__(kBlock) void rows_pack_kernel(int64_t n, const Idx* __restrict__ idx, int D, int nblk, const float* __restrict__ x,
                                                           int64_t ldx, uint64_t* __restrict__ masks, int32_t* __restrict__ counts,
                                                           const int64_t* __restrict__ offsets, float* __restrict__ values) {
  constexpr int RPW = kWave / G;  // rows per wave-instruction
  const int lane = threadIdx.x & (kWave - 1);
  const int l = lane % G, sub = lane / G;
  const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t base = wave_id * RPW; base < n; base += n_waves * RPW) {  // wave-uniform trip count (ballots inside)
    const int64_t i = base + sub;
    const bool live = i < n;
    const int64_t r = live ? (idx ? (int64_t)idx[i] : i) : 0;
    int total = 0;
    int64_t off = (MODE == 1 && live) ? offsets[i] : 0;
    for (int blk = 0; blk < nblk; ++blk) {
      const int c0 = blk * 64 + l * 4;
      const bool have = live && c0 < D;
      v4f v = (v4f)(0.f);
      if (have) v = *reinterpret_cast<const v4f*>(x + r * ldx + c0);
      if (MODE == 0) {
        uint64_t b[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = __ballot(have && v[c] != 0.f);
        const uint64_t m = group_mask<G>(b, sub);
        if (live && l == 0) masks[i * nblk + blk] = m;
        total += __popcll(m);
      } else {
        const uint64_t m = live ? masks[i * nblk + blk] : 0ull;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int bit = c * G + l;
          if ((m >> bit) & 1ull) values[off + __popcll(m & ((1ull << bit) - 1ull))] = v[c];
        }
        off += __popcll(m);
      }
    }
    if (MODE == 0 && live && l == 0) counts[i] = total;
  }
}

template <int G>
__global__ __launch_bounds__(kBlock) void rows_unpack_kernel(int64_t n, int D, int nblk, const uint64_t* __restrict__ masks,
                                                             const int64_t* __restrict__ offsets, const float* __restrict__ values,
                                                             float* __restrict__ out, int64_t ldo) {
  constexpr int RPW = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int l = lane % G, sub = lane / G;
  const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t base = wave_id * RPW; base < n; base += n_waves * RPW) {
    const int64_t i = base + sub;
    if (i >= n) continue;
    int64_t off = offsets[i];
    for (int blk = 0; blk < nblk; ++blk) {
      const int c0 = blk * 64 + l * 4;
      const uint64_t m = masks[i * nblk + blk];
      if (c0 < D) {
        v4f v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int bit = c * G + l;
          v[c] = ((m >> bit) & 1ull) ? values[off + __popcll(m & ((1ull << bit) - 1ull))] : 0.f;
        }
        __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(out + i * ldo + c0));
      }
      off += __popcll(m);
    }
  }
}

__global__ __launch_bounds__(kBlock) void rows_mask_count_kernel(int64_t n, int nblk, const uint64_t* __restrict__ masks, int32_t* __restrict__ counts) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    int t = 0;
    for (int b = 0; b < nblk; ++b) t += __popcll(masks[i * nblk + b]);
    counts[i] = t;
  }
}

static int64_t stream_blocks(int64_t n, int rows_per_wave) {
  int64_t blocks = (n + (int64_t)rows_per_wave * kWavesPerBlock - 1) / ((int64_t)rows_per_wave * kWavesPerBlock);
  if (blocks > 256 * 32) blocks = 256 * 32;
  return blocks < 1 ? 1 : blocks;
}

static int32_t check_shape(const char* what, int64_t n, int64_t D, int64_t ld, const void* p) {
  MGX_CHECK_ARG(n >= 0 && D > 0 && ld >= D, "%s: negative sizes or a row stride below D", what);
  if (D % 4 || D > 64 * kMaxBlocks || ld % 4 || (uintptr_t)p % 16)
    MGX_UNSUPPORTED("%s: rows of 4 .. %d columns with D %% 4 == 0, strides in multiples of 4 floats, 16-byte aligned pointers", what, 64 * kMaxBlocks);
  return MGX_OK;
}

template <typename Idx, int MODE>
static int32_t pack_launch(int64_t n, const void* idx, int64_t D, const float* x, int64_t ldx, uint64_t* masks, int32_t* counts,
                           const int64_t* offsets, float* values, hipStream_t s) {
  const RowShape sh = row_shape(D);
  const dim3 grid((unsigned)stream_blocks(n, kWave / sh.G)), block(kBlock);
#define MGX_PK(GG) hipLaunchKernelGGL((rows_pack_kernel<Idx, GG, MODE>), grid, block, 0, s, n, (const Idx*)idx, (int)D, sh.nblk, x, ldx, masks, counts, offsets, values)
  switch (sh.G) {
    case 1: MGX_PK(1); break;
    case 2: MGX_PK(2); break;
    case 4: MGX_PK(4); break;
    case 8: MGX_PK(8); break;
    default: MGX_PK(16); break;
  }
#undef MGX_PK
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

}  // namespace
}  // namespace mgx

extern "C" int64_t mgx_rows_mask_words(int64_t D) { return D > 0 ? (D + 63) / 64 : 0; }

extern "C" int32_t mgx_rows_pack_count(int64_t n, const void* idx, int32_t idx_bits, int64_t D, const float* x, int64_t x_stride,
                                       uint64_t* masks, int32_t* counts, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(idx == nullptr || idx_bits == 32 || idx_bits == 64, "mgx_rows_pack_count: idx_bits must be 32 or 64");
  if (int32_t st = check_shape("mgx_rows_pack_count", n, D, x_stride, x)) return st;
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(x && masks && counts, "mgx_rows_pack_count: NULL pointer");
  if (idx && idx_bits == 64) return pack_launch<int64_t, 0>(n, idx, D, x, x_stride, masks, counts, nullptr, nullptr, (hipStream_t)stream);
  return pack_launch<int32_t, 0>(n, idx, D, x, x_stride, masks, counts, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int32_t mgx_rows_mask_count(int64_t n, int64_t D, const uint64_t* masks, int32_t* counts, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && D > 0, "mgx_rows_mask_count: negative sizes");
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(masks && counts, "mgx_rows_mask_count: NULL pointer");
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(rows_mask_count_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, n, (int)((D + 63) / 64), masks, counts);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int32_t mgx_rows_pack_values(int64_t n, const void* idx, int32_t idx_bits, int64_t D, const float* x, int64_t x_stride,
                                        const uint64_t* masks, const int64_t* offsets, float* values, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(idx == nullptr || idx_bits == 32 || idx_bits == 64, "mgx_rows_pack_values: idx_bits must be 32 or 64");
  if (int32_t st = check_shape("mgx_rows_pack_values", n, D, x_stride, x)) return st;
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(x && masks && offsets && values, "mgx_rows_pack_values: NULL pointer");
  uint64_t* m = const_cast<uint64_t*>(masks);
  if (idx && idx_bits == 64) return pack_launch<int64_t, 1>(n, idx, D, x, x_stride, m, nullptr, offsets, values, (hipStream_t)stream);
  return pack_launch<int32_t, 1>(n, idx, D, x, x_stride, m, nullptr, offsets, values, (hipStream_t)stream);
}

extern "C" int32_t mgx_rows_unpack(int64_t n, int64_t D, const uint64_t* masks, const int64_t* offsets, const float* values, float* out,
                                   int64_t out_stride, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  if (int32_t st = check_shape("mgx_rows_unpack", n, D, out_stride, out)) return st;
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(masks && offsets && out, "mgx_rows_unpack: NULL pointer");  // values may be NULL when every mask is empty
  const RowShape sh = row_shape(D);
  const dim3 grid((unsigned)stream_blocks(n, kWave / sh.G)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
#define MGX_UP(GG) hipLaunchKernelGGL((rows_unpack_kernel<GG>), grid, block, 0, s, n, (int)D, sh.nblk, masks, offsets, values, out, out_stride)
  switch (sh.G) {
    case 1: MGX_UP(1); break;
    case 2: MGX_UP(2); break;
    case 4: MGX_UP(4); break;
    case 8: MGX_UP(8); break;
    default: MGX_UP(16); break;
  }
#undef MGX_UP
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// mgx_rows_unpack_add_csr: out[v, :] += sum over the entries p of row v of a CSR of unpack(masks[p], values[offsets[p] ..]) -- the
// returned halo-row gradients added into their owners straight from the packed form (dist.SparseHalo.finish_back + the copy_u over
// return_csr it fed: one dense [send rows, D] matrix written and read less per backward exchange).  A lane group owns an output row and
// walks its entries in CSR order (= peer order: deterministic), rows without entries are left alone.
namespace mgx {
namespace {
// One 64-column block per row (D <= 64): the entries' metadata is fetched by the lane group TOGETHER -- lane q of the group loads position,
// offset and mask of the row's q-th entry -- and handed round with ds_bpermute, so that a row of k entries costs two dependent round trips
// + k independent value gathers (issued four at a time) instead of 3 k dependent ones.  Entries are still added in CSR order.
template <int G>
__global__ __launch_bounds__(kBlock) void rows_unpack_add_csr1_kernel(int64_t n, const int32_t* __restrict__ indptr, const int32_t* __restrict__ pos,
                                                                      int D, const uint64_t* __restrict__ masks,
                                                                      const int64_t* __restrict__ offsets, const float* __restrict__ values,
                                                                      float* __restrict__ out, int64_t ldo) {
  constexpr int RPW = kWave / G;
  constexpr int U = 4;
  const int lane = threadIdx.x & (kWave - 1);
  const int l = lane % G, sub = lane / G, gbase = sub * G;
  const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  const int c0 = l * 4;
  const bool cact = c0 < D;
  for (int64_t base = wave_id * RPW; base < n; base += n_waves * RPW) {  // wave-uniform trip count: the cross-lane reads below need every lane
    const int64_t v = base + sub;
    const bool live = v < n;
    const int32_t beg = live ? indptr[v] : 0, end = live ? indptr[v + 1] : 0;
    v4f acc = (v4f)(0.f);
    if (live && cact && end > beg) acc = *reinterpret_cast<const v4f*>(out + v * ldo + c0);
    for (int32_t chunk = 0;; chunk += G) {  // G entries of every row of the wave at a time
      const int cnt = (end - beg - chunk) < G ? (end - beg - chunk) : G;  // this row's entries in the chunk (may be <= 0)
      if (__builtin_amdgcn_ballot_w64(cnt > 0) == 0) break;
      int32_t p = 0;
      uint32_t off_lo = 0, m_lo = 0, m_hi = 0;
      if (l < cnt) {
        p = pos[beg + chunk + l];
        const int64_t o = offsets[p];  // below 2^32 values per message (host check)
        const uint64_t m = masks[p];
        off_lo = (uint32_t)o; m_lo = (uint32_t)m; m_hi = (uint32_t)(m >> 32);
      }
      const int maxcnt = __builtin_amdgcn_readfirstlane(__reduce_max_sync(~0ull, cnt > 0 ? cnt : 0));
      for (int q = 0; q < maxcnt; q += U) {
        float got[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int bi = (gbase + ((q + u) < G ? (q + u) : 0)) * 4;
          const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute(bi, (int)off_lo);
          const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(bi, (int)m_hi) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(bi, (int)m_lo);
          const bool have = (q + u) < cnt && cact;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int bit = c * G + l;
            got[u][c] = (have && ((m >> bit) & 1ull)) ? values[o + __popcll(m & ((1ull << bit) - 1ull))] : 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] += got[u][c];  // (adding 0.f for an absent entry changes nothing: x + 0 == x, and -0 never occurs in acc + 0)
      }
    }
    if (live && cact && end > beg) *reinterpret_cast<v4f*>(out + v * ldo + c0) = acc;
  }
}

template <int G>
__global__ __launch_bounds__(kBlock) void rows_unpack_add_csr_kernel(int64_t n, const int32_t* __restrict__ indptr, const int32_t* __restrict__ pos,
                                                                     int D, int nblk, const uint64_t* __restrict__ masks,
                                                                     const int64_t* __restrict__ offsets, const float* __restrict__ values,
                                                                     float* __restrict__ out, int64_t ldo) {
  constexpr int RPW = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int l = lane % G, sub = lane / G;
  const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t base = wave_id * RPW; base < n; base += n_waves * RPW) {
    const int64_t v = base + sub;
    if (v >= n) continue;
    const int32_t beg = indptr[v], end = indptr[v + 1];
    if (beg == end) continue;
    for (int blk = 0; blk < nblk; ++blk) {
      const int c0 = blk * 64 + l * 4;
      if (c0 >= D) continue;
      v4f acc = *reinterpret_cast<const v4f*>(out + v * ldo + c0);
      for (int32_t q = beg; q < end; ++q) {
        const int64_t p = pos[q];
        int64_t off = offsets[p];
        for (int b = 0; b < blk; ++b) off += __popcll(masks[p * nblk + b]);
        const uint64_t m = masks[p * nblk + blk];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int bit = c * G + l;
          if ((m >> bit) & 1ull) acc[c] += values[off + __popcll(m & ((1ull << bit) - 1ull))];
        }
      }
      *reinterpret_cast<v4f*>(out + v * ldo + c0) = acc;
    }
  }
}
}  // namespace
}  // namespace mgx

extern "C" int32_t mgx_rows_unpack_add_csr(int64_t n, const int32_t* indptr, const int32_t* positions, int64_t D, const uint64_t* masks,
                                           const int64_t* offsets, const float* values, float* out, int64_t out_stride, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  if (int32_t st = check_shape("mgx_rows_unpack_add_csr", n, D, out_stride, out)) return st;
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(indptr && masks && offsets && out, "mgx_rows_unpack_add_csr: NULL pointer");
  const RowShape sh = row_shape(D);
  const dim3 grid((unsigned)stream_blocks(n, kWave / sh.G)), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
  if (sh.nblk == 1 && sh.G >= 2) {  // one block per row: the lane group fetches its entries' metadata together
#define MGX_UA1(GG) hipLaunchKernelGGL((rows_unpack_add_csr1_kernel<GG>), grid, block, 0, s, n, indptr, positions, (int)D, masks, offsets, values, out, out_stride)
    switch (sh.G) {
      case 2: MGX_UA1(2); break;
      case 4: MGX_UA1(4); break;
      case 8: MGX_UA1(8); break;
      default: MGX_UA1(16); break;
    }
#undef MGX_UA1
    MGX_CHECK_LAUNCH();
    return MGX_OK;
  }
#define MGX_UA(GG) hipLaunchKernelGGL((rows_unpack_add_csr_kernel<GG>), grid, block, 0, s, n, indptr, positions, (int)D, sh.nblk, masks, offsets, values, out, out_stride)
  switch (sh.G) {
    case 1: MGX_UA(1); break;
    case 2: MGX_UA(2); break;
    case 4: MGX_UA(4); break;
    case 8: MGX_UA(8); break;
    default: MGX_UA(16); break;
  }
#undef MGX_UA
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
