// spmm_tile.hip -- LDS-staged g-SpMM (copy_u / sum | mean) for DENSE neighbourhoods on gfx950 (MI355X).
//
// Same operator as spmm.hip's row-per-wave kernel -- what DGL's _CAPI_DGLKernelSpMM executes for kernel/dgl-new.py:20 and
// update_all(fn.copy_src, fn.mean) at main_dgl_reddit_sage.py:73-80 -- for graphs with hundreds of in-edges per node (reddit
// 492, proteins 597), where that kernel is pinned at the L2 -> CU gather rate (every edge re-fetches its source row from L2:
// 17-19 TB/s, 3-5 % of the HBM roofline) although destination rows scheduled next to each other share most of their sources.
//
// One 1024-thread workgroup (16 waves) owns a TILE of R = consumers * NACC * 4 work items of the schedule and walks the
// tables of an mgx_tile_plan (mi355x_graph/tileplan.py):
//   * LOADER waves (NL of the 16) gather the tile's staged sources chunk by chunk (127 rows + one all-zero row, 64 columns
//     = 32 KiB) from L2 into a 4-deep LDS ring with LDS-DMA (global_load_lds_dwordx4: per-lane source address = a row
//     gather, no VGPR round trip); they run up to three chunks ahead of the consumers, paced by ONE workgroup barrier per chunk
//     and counted s_waitcnt vmcnt (the DMA of chunk c+1, c+2 stay in flight while chunk c is handed over);
//   * CONSUMER waves own NACC rows per 16-lane group (lanes along the feature dimension, float4 each; 4 groups per wave),
//     one accumulator register set per row, statically indexed.  They walk their per-(chunk, wave) STREAM: a step is one LDS
//     slot per lane group (1 byte each), rows padded to 4-step supersteps with the zero slot, so the inner loop is one
//     4-byte stream load (prefetched two supersteps ahead), 4 x (v_bfe, ds_read_b128) and 4 x add -- no graph arrays, no
//     per-edge branches, every source row read from L2 once per tile instead of once per edge;
//   * edges whose source occurs once in the tile (nothing to re-use) follow in a DIRECT stream of source ids that the consumer
//     waves gather from global memory into the same accumulators (16-byte gathers, four in flight);
//   * epilogue: mean / dst_scale / accumulate, one 256-byte store per row; hub chunks write partial rows for
//     spmm_hub_fixup_kernel.  No atomics: fixed summation order for a given plan.
// Rows wider than 64 columns run as column passes (grid.y) over the same streams.
#include "common.h"

namespace mgx {

constexpr int kTileThreads = 1024;
constexpr int kTileWaves = kTileThreads / kWave;
constexpr int kChunkSlots = 128;                        // LDS rows per chunk (127 staged + the zero row)
constexpr int kPassCols = 64;                           // columns per pass: 16 lanes x float4
constexpr int kSlotBytes = kPassCols * 4;               // 256
constexpr int kChunkBytes = kChunkSlots * kSlotBytes;   // 32 KiB
constexpr int kRing = 4;                                // chunks resident in LDS
constexpr int kDmaPerChunk = kChunkBytes / 1024;        // LDS-DMA wave-instructions (1 KiB each) per chunk
constexpr int kNoItem = INT32_MIN;

struct TileArgs {
  const float* x;
  float* out;
  float* partial;
  const float* dst_scale;
  const int32_t* indptr;  // mean: full in-degree of a row
  const int32_t* tile_chunk_ptr;
  const int32_t* chunk_ids;
  const int32_t* lds_off;
  const uint16_t* lds_cnt;
  const uint32_t* lds_stream;
  const int32_t* dir_off;
  const int32_t* dir_cnt;
  const int32_t* dir_stream;
  const int32_t* tile_item;
  const float* zero_row;
  int num_tiles, tiles_per_xcd;
  int D, lds, ldo;  // columns, row strides (floats) of x and out
  int mean, accum;
};

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Plan tables are read-only for the whole launch: reading them through the CONSTANT address space lets a wave-uniform address
// become a scalar load (s_load_*, lgkmcnt) -- a vector load would share the in-order vmcnt queue with the LDS-DMA / the stream
// prefetches and every wait for it would drain them.
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* as_const(const T* p) {
  return (const __attribute__((address_space(4))) T*)p;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- loader waves ---------------------------------------------------------------------------------------------------------
template <int NL>
__device__ __forceinline__ void tile_loader(const TileArgs& a, char* ring, int wave, int lane, int cbeg, int n, int col0) {
  constexpr int PER = kDmaPerChunk / NL;  // DMA instructions per loader wave per chunk
  const int g = lane >> 4, l = lane & 15;
  const int col = col0 + l * 4;
  const bool cvalid = col < a.D;
  const char* zrow = reinterpret_cast<const char*>(a.zero_row + l * 4);
  const char* xcol = reinterpret_cast<const char*>(a.x + col);
  const uint32_t rowbytes = (uint32_t)a.lds * 4u;
  // The source ids reach the lanes through the SCALAR path (uniform address -> s_load_dwordx4, lgkmcnt): an ordinary vector
  // load would sit on the same in-order counter as the DMA and every wait for it would drain the chunks in flight.
  auto issue = [&](int k) {  // chunk k of this tile -> ring slot k % kRing; this wave's rows [wave * PER * 4, +PER * 4)
    const auto* ids = as_const(reinterpret_cast<const v4i*>(a.chunk_ids + (int64_t)(cbeg + k) * kChunkSlots + wave * (PER * 4)));
    char* dst = ring + (k % kRing) * kChunkBytes + wave * (PER * 1024);
    v4i q[PER];  // rows 4 i .. 4 i + 3 of this wave's share: one per lane group; all requested before the first is used
#pragma unroll
    for (int i = 0; i < PER; ++i) q[i] = ids[i];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int id = q[i].x;
      id = g == 1 ? q[i].y : id;
      id = g == 2 ? q[i].z : id;
      id = g == 3 ? q[i].w : id;
      const uint32_t off = (id >= 0 && cvalid) ? (uint32_t)id * rowbytes : 0u;  // the gathered matrix is below 4 GiB (host check)
      const char* src = (id >= 0 && cvalid) ? xcol + off : zrow;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dst + i * 1024), 16, 0, 0);
    }
  };
  const int pre = n < kRing - 1 ? n : kRing - 1;
  for (int k = 0; k < pre; ++k) issue(k);
  for (int c = 0; c < n; ++c) {
    const int rem = n - 1 - c;  // chunks issued after c that may stay in flight
    if (rem >= 2) wait_vmcnt<2 * PER>();
    else if (rem == 1) wait_vmcnt<PER>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // chunk c has landed; every consumer has finished chunk c - 1
    asm volatile("" ::: "memory");
    if (c + kRing - 1 < n) issue(c + kRing - 1);  // into the slot chunk c - 1 occupied
  }
}

// ---- consumer waves -------------------------------------------------------------------------------------------------------
template <int NACC, int NL>
__device__ __forceinline__ void tile_consumer(const TileArgs& a, const char* ring, int cw, int lane, int tile, int cbeg, int n, int col0) {
  constexpr int NC = kTileWaves - NL;
  constexpr int R = NC * NACC * 4;
  const int g = lane >> 4, l = lane & 15;
  const int col = col0 + l * 4;
  const bool cvalid = col < a.D;
  v4f acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = (v4f)(0.f);

  // ---- staged part: chunks from LDS.  Stream layout [superstep][lane group][4 steps]: a lane reads ONE dword per superstep (its
  // group's four slot bytes; 16 lanes share the address, a wave reads 16 contiguous bytes), two supersteps ahead of its use --
  // the stream is contiguous over the rows (j) and chunks of a wave and padded behind its end, so the prefetch never branches.
  if (n > 0) {
    const auto* off = as_const(a.lds_off);
    const auto* cnt = as_const(reinterpret_cast<const v4u*>(a.lds_cnt));
    int64_t k = (int64_t)cbeg * NC + cw;
    const uint32_t* sp = a.lds_stream + (int64_t)off[k] * 4 + g;
    uint32_t w0 = sp[0], w1 = sp[4];
    sp += 8;
    for (int c = 0; c < n; ++c) {
      const v4u cq = cnt[k];
      const uint32_t cnts[4] = {cq.x, cq.y, cq.z, cq.w};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every LDS read of the previous chunk has returned
      __builtin_amdgcn_s_barrier();                         // chunk c is in its ring slot
      asm volatile("" ::: "memory");
      const uint32_t lrow = (uint32_t)(c % kRing) * kChunkBytes + (uint32_t)l * 16u;
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        const int nss = (int)((cnts[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
        for (int s = 0; s < nss; ++s) {
          const uint32_t w = w0;
          w0 = w1;
          w1 = *sp;
          sp += 4;
          v4f v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t slot = (w >> (8 * u)) & 0xffu;
            v[u] = *reinterpret_cast<const v4f*>(ring + ((slot << 8) + lrow));
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += v[u];
        }
      }
      k += NC;
    }
  }

  // ---- direct part: sources used once in this tile, gathered from global memory.  Stream layout [superstep][lane group][4 ids].
  {
    const int64_t k = (int64_t)tile * NC + cw;
    const int so = as_const(a.dir_off)[k], se = as_const(a.dir_off)[k + 1];  // supersteps
    if (se > so) {
      const v4u c0 = as_const(reinterpret_cast<const v4u*>(a.dir_cnt))[k * 2];
      const v4u c1 = as_const(reinterpret_cast<const v4u*>(a.dir_cnt))[k * 2 + 1];
      const uint32_t cnts[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
      const v4i* dp = reinterpret_cast<const v4i*>(a.dir_stream) + (int64_t)so * 4 + g;
      v4i i0 = dp[0], i1 = dp[4];
      dp += 8;
      const uint32_t rowbytes = (uint32_t)a.lds * 4u;
      const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)(cvalid ? col : 0) * 4u;  // idle lanes re-read column 0 (never stored)
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        const int nss = (int)cnts[j];
        for (int s = 0; s < nss; ++s) {
          const v4i id = i0;
          i0 = i1;
          i1 = *dp;
          dp += 4;
          v4f v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t o = id[u] >= 0 ? (uint32_t)id[u] * rowbytes : 0u;  // padding re-reads row 0 (valid memory)
            v[u] = *reinterpret_cast<const v4f*>(xb + o);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += id[u] >= 0 ? v[u] : (v4f)(0.f);
        }
      }
    }
  }

  // ---- epilogue: one row per lane group and accumulator
  if (!cvalid) return;
#pragma unroll
  for (int j = 0; j < NACC; ++j) {
    const int item = a.tile_item[(int64_t)tile * R + (cw * NACC + j) * 4 + g];
    if (item == kNoItem) continue;
    v4f r = acc[j];
    if (item >= 0) {
      if (a.mean) {
        const int deg = a.indptr[item + 1] - a.indptr[item];
        r = r / (float)(deg > 1 ? deg : 1);
      }
      if (a.dst_scale) r = r * a.dst_scale[item];
      float* op = a.out + (int64_t)item * a.ldo + col;
      if (a.accum) r += *reinterpret_cast<const v4f*>(op);
      __builtin_nontemporal_store(r, reinterpret_cast<v4f*>(op));
    } else {  // chunk of a split row: summed (and scaled) by spmm_hub_fixup_kernel in slot order
      *reinterpret_cast<v4f*>(a.partial + (int64_t)(-(item + 1)) * a.D + col) = r;
    }
  }
}

template <int NACC, int NL>
__global__ __launch_bounds__(kTileThreads) void spmm_tile_kernel(const TileArgs a) {
  __shared__ __attribute__((aligned(1024))) char ring[kRing * kChunkBytes];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & (kWave - 1);
  // block b serves XCD b % 8 (observed round-robin placement; speed only): consecutive tiles of the schedule -- which share
  // sources -- stay on one XCD's L2
  const int tile = (int)(blockIdx.x % kXcds) * a.tiles_per_xcd + (int)(blockIdx.x / kXcds);
  if (tile >= a.num_tiles || (int)(blockIdx.x / kXcds) >= a.tiles_per_xcd) return;
  const int cbeg = a.tile_chunk_ptr[tile];
  const int n = a.tile_chunk_ptr[tile + 1] - cbeg;
  const int col0 = blockIdx.y * kPassCols;
  if (wave < NL) tile_loader<NL>(a, ring, wave, lane, cbeg, n, col0);
  else tile_consumer<NACC, NL>(a, ring, wave - NL, lane, tile, cbeg, n, col0);
}

template <int NL>
static bool launch_tile(int nacc, const TileArgs& a, dim3 grid, hipStream_t s) {
  switch (nacc) {
    case 4: hipLaunchKernelGGL((spmm_tile_kernel<4, NL>), grid, dim3(kTileThreads), 0, s, a); return true;
    case 5: hipLaunchKernelGGL((spmm_tile_kernel<5, NL>), grid, dim3(kTileThreads), 0, s, a); return true;
    case 6: hipLaunchKernelGGL((spmm_tile_kernel<6, NL>), grid, dim3(kTileThreads), 0, s, a); return true;
    case 8: hipLaunchKernelGGL((spmm_tile_kernel<8, NL>), grid, dim3(kTileThreads), 0, s, a); return true;
    default: return false;
  }
}

// defined in spmm.hip
int32_t spmm_hub_fixup_launch(const mgx_csr* csr, const mgx_spmm_plan* plan, const float* partial, const float* dst_scale, float* out,
                              int D, int mean, int accum, int ldo, hipStream_t s);

}  // namespace mgx

extern "C" int32_t mgx_spmm_tile_copy_u(const mgx_csr* csr, const mgx_spmm_plan* plan, const mgx_tile_plan* tp, int32_t reduce,
                                        const float* ufeat, int64_t D, int64_t u_stride, const float* dst_scale, float* out,
                                        int64_t out_stride, float* partial_ws, int32_t flags, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(csr != nullptr && tp != nullptr, "mgx_spmm_tile_copy_u: NULL csr / tile plan");
  MGX_CHECK_ARG(reduce == MGX_REDUCE_SUM || reduce == MGX_REDUCE_MEAN, "mgx_spmm_tile_copy_u: SUM or MEAN only, got %d", reduce);
  if (csr->idx_bits != 32) MGX_UNSUPPORTED("mgx_spmm_tile_copy_u: int32 graphs only");
  if (D <= 0 || D % 4 != 0 || u_stride % 4 != 0 || out_stride % 4 != 0 || (uintptr_t)ufeat % 16 != 0 || (uintptr_t)out % 16 != 0)
    MGX_UNSUPPORTED("mgx_spmm_tile_copy_u: D and both strides must be multiples of 4 and the operands 16-byte aligned");
  MGX_CHECK_ARG(u_stride >= D && out_stride >= D, "mgx_spmm_tile_copy_u: strides must be >= D");
  if (csr->num_cols * u_stride * 4 >= (int64_t(1) << 32)) MGX_UNSUPPORTED("mgx_spmm_tile_copy_u: gathered matrix must be below 4 GiB");
  MGX_CHECK_ARG(tp->consumers + tp->loaders == kTileWaves && (tp->loaders == 2 || tp->loaders == 4),
                "mgx_spmm_tile_copy_u: tile plan built for %d + %d waves", tp->consumers, tp->loaders);
  MGX_CHECK_ARG(tp->num_tiles >= 0 && tp->num_tiles < (int64_t(1) << 28), "mgx_spmm_tile_copy_u: bad tile count");
  if (tp->num_tiles == 0 || csr->num_rows == 0) return MGX_OK;
  MGX_CHECK_ARG(tp->tile_chunk_ptr && tp->lds_off && tp->dir_off && tp->dir_cnt && tp->tile_item && tp->zero_row,
                "mgx_spmm_tile_copy_u: tile plan tables are NULL");
  MGX_CHECK_ARG(tp->num_chunks == 0 || (tp->chunk_ids && tp->lds_cnt), "mgx_spmm_tile_copy_u: chunk tables are NULL");
  MGX_CHECK_ARG(ufeat != nullptr && out != nullptr && csr->indptr != nullptr, "mgx_spmm_tile_copy_u: NULL operand");
  const bool hubs = plan && plan->num_hubs > 0;
  MGX_CHECK_ARG(!hubs || partial_ws, "mgx_spmm_tile_copy_u: the plan has split rows but no partial workspace");
  TileArgs a;
  a.x = ufeat; a.out = out; a.partial = partial_ws; a.dst_scale = dst_scale; a.indptr = (const int32_t*)csr->indptr;
  a.tile_chunk_ptr = tp->tile_chunk_ptr; a.chunk_ids = tp->chunk_ids; a.lds_off = tp->lds_off; a.lds_cnt = tp->lds_cnt;
  a.lds_stream = tp->lds_stream; a.dir_off = tp->dir_off; a.dir_cnt = tp->dir_cnt; a.dir_stream = tp->dir_stream;
  a.tile_item = tp->tile_item; a.zero_row = tp->zero_row;
  a.num_tiles = (int)tp->num_tiles;
  a.tiles_per_xcd = (int)((tp->num_tiles + kXcds - 1) / kXcds);
  a.D = (int)D; a.lds = (int)u_stride; a.ldo = (int)out_stride;
  a.mean = reduce == MGX_REDUCE_MEAN; a.accum = (flags & MGX_SPMM_ACCUMULATE) ? 1 : 0;
  const dim3 grid((unsigned)(a.tiles_per_xcd * kXcds), (unsigned)((D + kPassCols - 1) / kPassCols));
  hipStream_t s = (hipStream_t)stream;
  const bool ok = tp->loaders == 2 ? launch_tile<2>(tp->nacc, a, grid, s) : launch_tile<4>(tp->nacc, a, grid, s);
  if (!ok) MGX_UNSUPPORTED("mgx_spmm_tile_copy_u: no kernel for %d rows per lane group", tp->nacc);
  MGX_CHECK_LAUNCH();
  if (hubs) return spmm_hub_fixup_launch(csr, plan, partial_ws, dst_scale, out, a.D, a.mean, a.accum, a.ldo, s);
  return MGX_OK;
}
