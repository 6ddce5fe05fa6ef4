// spmm_tile.hip -- LDS-staged g-SpMM (copy_u / sum | mean) for DENSE neighbourhoods on gfx950 (MI355X).
//
// Same operator as spmm.hip's row-per-wave kernel -- what DGL's _CAPI_DGLKernelSpMM executes for kernel/dgl-new.py:20 and
// update_all(fn.copy_src, fn.mean) at main_dgl_reddit_sage.py:73-80 -- for graphs with hundreds of in-edges per node (reddit
// 492, proteins 597), where that kernel is pinned at the L2 -> CU gather rate (every edge re-fetches its source row from L2:
// 17-19 TB/s, 3-5 % of the HBM roofline) although destination rows scheduled next to each other share most of their sources.
//
// One workgroup of W waves (16: one per CU with a 4-chunk ring; or 8: TWO per CU with a 2-chunk ring each, so that one tile's
// barrier waits, direct part and epilogue overlap the other's LDS work) owns a TILE of R = consumers * NACC * 4 work items of
// the schedule and walks the tables of an mgx_tile_plan (mi355x_graph/tileplan.py):
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
#include <stdlib.h>

#include "tile_common.h"

namespace mgx {

constexpr int kChunkSlots = 128;                        // LDS rows per chunk (127 staged + the zero row)
constexpr int kPassCols = 64;                           // columns per pass: 16 lanes x float4
constexpr int kSlotBytes = kPassCols * 4;               // 256
constexpr int kChunkBytes = kChunkSlots * kSlotBytes;   // 32 KiB
constexpr int kDmaPerChunk = kChunkBytes / 1024;        // LDS-DMA wave-instructions (1 KiB each) per chunk
constexpr int kStreamRingBytes = 1024;                  // per wave: 4 windows of 16 supersteps x 16 bytes of the staged part's stream

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
  const int32_t* tile_order;  // optional: dispatch slot -> tile (longest tiles of an XCD's stretch first)
  const float* zero_row;
  int num_tiles, tiles_per_xcd;
  int D, lds, ldo;  // columns, row strides (floats) of x and out
  int mean, accum;
};


// ---- loader waves ---------------------------------------------------------------------------------------------------------
template <int W, int NL, int RING>
__device__ __forceinline__ void tile_loader(const TileArgs& a, char* ring, int wave, int lane, int tile, int cbeg, int n, int col0) {
  constexpr int PER = kDmaPerChunk / NL;  // DMA instructions per loader wave per chunk
  static_assert((RING - 2) * PER <= 48, "the counted vmcnt wait must stay below the 6-bit counter");
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
    char* dst = ring + (k % RING) * kChunkBytes + wave * (PER * 1024);
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
  const int pre = n < RING - 1 ? n : RING - 1;
  for (int k = 0; k < pre; ++k) issue(k);
  for (int c = 0; c < n; ++c) {
    const int rem = n - 1 - c;  // chunks issued after c that may stay in flight
    if (RING >= 4 && rem >= 2) wait_vmcnt<(RING >= 4 ? 2 : 0) * PER>();  // chunks c + 1, c + 2 may stay in flight
    else if (RING >= 3 && rem >= 1) wait_vmcnt<(RING >= 3 ? 1 : 0) * PER>();
    else wait_vmcnt<0>();
    // chunk c has landed; every consumer has finished chunk c - 1.  (Per-slot FULL / FREE words in LDS instead of this barrier were built
    // and measured in round 4 -- slower at every geometry, docs/LOG_r04.md section 8 -- and are gone.)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (c + RING - 1 < n) issue(c + RING - 1);  // into the slot chunk c - 1 occupied
  }
  // the consumers' (n + 1)-th barrier (before the direct part reuses the ring): the loaders take part in it, as in gat_tile.inc,
  // instead of relying on exited waves being dropped from the workgroup's barrier count
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
}

// ---- consumer waves -------------------------------------------------------------------------------------------------------
template <int W, int NL, int NACC, int RING>
__device__ __forceinline__ void tile_consumer(const TileArgs& a, char* ring, int cw, int lane, int tile, int cbeg, int n, int col0) {
  constexpr int NC = W - NL;
  constexpr int R = NC * NACC * 4;
  const int g = lane >> 4, l = lane & 15;
  const int col = col0 + l * 4;
  const bool cvalid = col < a.D;
  v4f acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = (v4f)(0.f);

  // The per-wave STREAMS are read once from HBM, 16 bytes (LDS part) or 64 bytes (direct part) per superstep and wave: a
  // register prefetch two supersteps ahead exposes an HBM miss per cache line (measured: 3.7 us per chunk of 12 supersteps).
  // Instead every consumer wave DMAs its own stream into a private LDS ring, a WINDOW of 16 supersteps per wave-instruction
  // (one coalesced 256-byte / 1-KiB read), two windows = 32 supersteps ahead of its use; the words then come from LDS.  The
  // window test is the only branch of the inner loop and has no register results (a load into registers under that branch
  // made hipcc wait for it at the loop's back edge).  Window k is complete when at most ONE younger DMA is outstanding.
  // ---- staged part: chunks from LDS.  Stream layout [superstep][lane group][4 steps]: one dword per lane group and superstep.
  auto staged_part = [&]() {
  if (n > 0) {
    const auto* cnt = as_const(reinterpret_cast<const v4u*>(a.lds_cnt));  // 16 uint16 = two quads per (chunk, wave)
    int64_t k = (int64_t)cbeg * NC + cw;
    const uint32_t* gs = a.lds_stream + (int64_t)as_const(a.lds_off)[(int64_t)tile * NC + cw] * 4 + lane;
    char* swin = ring + RING * kChunkBytes + cw * kStreamRingBytes;  // 4 windows x 256 bytes
    auto dma = [&](int wi) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gs + (int64_t)wi * 64), (lds_ptr_t)(swin + (wi & 3) * 256), 4, 0, 0);
    };
    dma(0);
    dma(1);
    wait_vmcnt<1>();
    dma(2);
    int ss = 0;      // supersteps of this wave's stream consumed so far
    int ready = 16;  // supersteps [0, ready) are in the LDS ring
    auto word = [&](int i) -> uint32_t { return *reinterpret_cast<const uint32_t*>(swin + (i & 63) * 16 + g * 4); };
    auto open_window = [&](int upto) {  // make the stream words up to superstep `upto` readable
      if (upto >= ready) {
        wait_vmcnt<1>();
        dma((ready >> 4) + 2);
        ready += 16;
      }
    };
    uint32_t w0 = word(0), w1 = word(1);  // the next two supersteps
    for (int c = 0; c < n; ++c) {
      const v4u cq = cnt[2 * k];
      const v4u cr = NACC > 8 ? cnt[2 * k + 1] : (v4u)(0u);
      const uint32_t cnts[8] = {cq.x, cq.y, cq.z, cq.w, cr.x, cr.y, cr.z, cr.w};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every LDS read of the previous chunk has returned
      __builtin_amdgcn_s_barrier();                         // chunk c is in its ring slot
      asm volatile("" ::: "memory");
      const uint32_t lrow = (uint32_t)(c % RING) * kChunkBytes + (uint32_t)l * 16u;
      auto gather4 = [&](uint32_t w, v4f (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const v4f*>(ring + ((((w >> (8 * u)) & 0xffu) << 8) + lrow));
      };
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        const int nss = (int)((cnts[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
        int s = 0;
        for (; s + 2 <= nss; s += 2) {  // two supersteps = 8 LDS rows in flight per lane group
          open_window(ss + 3);
          const uint32_t n0 = word(ss + 2), n1 = word(ss + 3);
          v4f va[4], vb[4];
          gather4(w0, va);
          gather4(w1, vb);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += va[u];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += vb[u];
          w0 = n0;
          w1 = n1;
          ss += 2;
        }
        if (s < nss) {
          open_window(ss + 2);
          const uint32_t n1 = word(ss + 2);
          v4f va[4];
          gather4(w0, va);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += va[u];
          w0 = w1;
          w1 = n1;
          ss += 1;
        }
      }
      k += NC;
    }
  }

  };

  // ---- direct part: sources used once in this tile, gathered from global memory.  Stream layout [superstep][lane group][4 ids]
  // = 64 bytes per superstep; its windows (1 KiB = 16 supersteps) live where the chunk ring was: every consumer has left the
  // staged part (barrier, which the loader waves join before they exit).
  auto direct_part = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int64_t k = (int64_t)tile * NC + cw;
    const int so = as_const(a.dir_off)[k], se = as_const(a.dir_off)[k + 1];  // supersteps
    if (se > so) {
      const auto* dcnt = as_const(reinterpret_cast<const v4u*>(a.dir_cnt)) + k * 4;  // 16 int32 = four quads per (tile, wave)
      const v4u c0 = dcnt[0];
      const v4u c1 = NACC > 4 ? dcnt[1] : (v4u)(0u);
      const v4u c2 = NACC > 8 ? dcnt[2] : (v4u)(0u);
      const v4u c3 = NACC > 12 ? dcnt[3] : (v4u)(0u);
      const uint32_t cnts[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
      const v4i* gd = reinterpret_cast<const v4i*>(a.dir_stream) + (int64_t)so * 4 + lane;
      char* dwin = ring + cw * 4096;  // 4 windows x 1 KiB
      auto dma = [&](int wi) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gd + (int64_t)wi * 64), (lds_ptr_t)(dwin + (wi & 3) * 1024), 16, 0, 0);
      };
      dma(0);
      dma(1);
      wait_vmcnt<1>();
      dma(2);
      int ss = 0, ready = 16;
      auto ids = [&](int i) -> v4i { return *reinterpret_cast<const v4i*>(dwin + (i & 63) * 64 + g * 16); };
      auto open_window = [&](int upto) {
        if (upto >= ready) {
          wait_vmcnt<1>();  // the row gathers of the previous supersteps have been consumed: only window DMAs are in flight
          dma((ready >> 4) + 2);
          ready += 16;
        }
      };
      v4i i0 = ids(0), i1 = ids(1);
      const uint32_t rowbytes = (uint32_t)a.lds * 4u;
      const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)(cvalid ? col : 0) * 4u;  // idle lanes re-read column 0 (never stored)
      auto gather4 = [&](const v4i& id, v4f (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t o = id[u] >= 0 ? (uint32_t)id[u] * rowbytes : 0u;  // padding re-reads row 0 (valid memory)
          v[u] = *reinterpret_cast<const v4f*>(xb + o);
        }
      };
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        const int nss = (int)cnts[j];
        int s = 0;
        for (; s + 2 <= nss; s += 2) {
          open_window(ss + 3);
          const v4i n0 = ids(ss + 2), n1 = ids(ss + 3);
          v4f va[4], vb[4];
          gather4(i0, va);
          gather4(i1, vb);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += i0[u] >= 0 ? va[u] : (v4f)(0.f);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += i1[u] >= 0 ? vb[u] : (v4f)(0.f);
          i0 = n0;
          i1 = n1;
          ss += 2;
        }
        if (s < nss) {
          open_window(ss + 2);
          const v4i n1 = ids(ss + 2);
          v4f va[4];
          gather4(i0, va);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += i0[u] >= 0 ? va[u] : (v4f)(0.f);
          i0 = i1;
          i1 = n1;
          ss += 1;
        }
      }
    }
  };

  // (Measured and dropped: tiles alternating the ORDER of the two parts by dispatch round, with the direct part's windows in a
  // private ring, so that half of the resident workgroups stream rows through the fabric while the other half keeps the LDS
  // pipes busy -- reddit shape D = 64: 1.15 -> 1.27 ms; the direct part of a tile took 260 k cycles instead of 134-191 k.)
  staged_part();
  direct_part();

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

template <int W, int NL, int NACC, int RING>
__global__ __launch_bounds__(W * kWave, 4) void spmm_tile_kernel(const TileArgs a) {
  // W = 16: one workgroup per CU (4 waves per SIMD); W = 8: two per CU -- either way 128 registers per lane
  __shared__ __attribute__((aligned(1024))) char ring[RING * kChunkBytes + W * kStreamRingBytes];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & (kWave - 1);
  // block b serves XCD b % 8 (observed round-robin placement; speed only): consecutive tiles of the schedule -- which share
  // sources -- stay on one XCD's L2
  const int slot = (int)(blockIdx.x % kXcds) * a.tiles_per_xcd + (int)(blockIdx.x / kXcds);
  if (slot >= a.num_tiles) return;
  const int tile = a.tile_order ? a.tile_order[slot] : slot;
  const int cbeg = a.tile_chunk_ptr[tile];
  const int n = a.tile_chunk_ptr[tile + 1] - cbeg;
  const int col0 = blockIdx.y * kPassCols;
  if (wave < NL) tile_loader<W, NL, RING>(a, ring, wave, lane, tile, cbeg, n, col0);
  else tile_consumer<W, NL, NACC, RING>(a, ring, wave - NL, lane, tile, cbeg, n, col0);
}

// ---- narrow rows ------------------------------------------------------------------------------------------------------------
// spmm_tile_narrow_kernel<LG, NACC>: the same tile walk for rows of 32 (LG = 3: 8 lanes per row) or 16 columns (LG = 2: 4 lanes) per
// pass -- on dense graphs the row kernel is REQUEST bound at these widths (one L2 request per edge whatever the row's bytes,
// profiles/r02_reddit_l2_requests.txt), and the tile turns re-used sources into LDS reads: a wave-instruction covers 8 / 16 rows,
// a chunk holds 255 sources (32 / 16 KiB).  8-wave workgroups, two per CU: wave 0 loads (2-chunk ring; the chunk's 256 source ids
// arrive by ONE DMA into LDS a chunk ahead, so the row gathers need no scalar-load chain), waves 1-7 consume.  Streams as above
// ([superstep][lane group][4 steps]), through 1-KiB windows (32 / 16 supersteps) in a private 2-window ring per wave.
template <int LG, int RING>
struct NarrowGeo {
  static constexpr int G = 1 << LG;                 // lanes per row
  static constexpr int NBG = kWave / G;             // rows per wave-instruction
  static constexpr int SLOT = 16 * G;               // bytes per LDS row
  static constexpr int CS = 256;                    // slots per chunk (255 + the zero row): one-byte stream entries
  static constexpr int CHB = CS * SLOT;             // chunk bytes
  static constexpr int PER = CHB / 1024;            // row DMAs per chunk
  static constexpr int WS = 256 / NBG;              // supersteps per 1-KiB stream window
  static constexpr int WSD = 64 / NBG;              // supersteps per 1-KiB window of the direct stream
  static constexpr int kWaves = 8, kConsumers = 7;
  static constexpr int kIdsOff = RING * CHB;        // LDS layout: [RING chunks][RING x 1 KiB ids][7 x 2 KiB stream windows]
  static constexpr int kStreamOff = kIdsOff + RING * 1024;
  static constexpr int kLdsBytes = kStreamOff + kConsumers * 2048;
  // the loader's queue in steady state: [rows(c) ids(c + RING - 1)] [rows(c + 1) ids(c + RING)] ...: rows(c) and the ids the next
  // gather needs have landed once at most this many DMAs are outstanding
  static constexpr int kAhead = (RING - 2) * (PER + 1);
  static_assert(kAhead <= 63, "vmcnt is a 6-bit counter");
};

template <int LG, int NACC, int RING>
__global__ __launch_bounds__(512, 4) void spmm_tile_narrow_kernel(const TileArgs a) {
  typedef NarrowGeo<LG, RING> Geo;
  constexpr int G = Geo::G, NBG = Geo::NBG, NC = Geo::kConsumers;
  __shared__ __attribute__((aligned(1024))) char ring[Geo::kLdsBytes];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & (kWave - 1);
  const int slotid = (int)(blockIdx.x % kXcds) * a.tiles_per_xcd + (int)(blockIdx.x / kXcds);
  if (slotid >= a.num_tiles) return;
  const int tile = a.tile_order ? a.tile_order[slotid] : slotid;
  const int cbeg = a.tile_chunk_ptr[tile];
  const int n = a.tile_chunk_ptr[tile + 1] - cbeg;
  const int col0 = blockIdx.y * (G * 4);
  const int g = lane >> LG, l = lane & (G - 1);
  const int col = col0 + l * 4;
  const bool cvalid = col < a.D;
  const uint32_t rowbytes = (uint32_t)a.lds * 4u;

  if (wave == 0) {  // ---- the loader wave
    const char* zrow = reinterpret_cast<const char*>(a.zero_row + l * 4);
    const char* xcol = reinterpret_cast<const char*>(a.x + col);
    auto issue_ids = [&](int k) {  // the 256 source ids of chunk k: one 1-KiB DMA
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a.chunk_ids + (int64_t)(cbeg + k) * Geo::CS + lane * 4),
                                       (lds_ptr_t)(ring + Geo::kIdsOff + (k % RING) * 1024), 16, 0, 0);
    };
    auto issue_rows = [&](int k) {  // chunk k -> ring slot k % RING; its ids are in LDS (landed: the wait before the barrier)
      const int* ids = reinterpret_cast<const int*>(ring + Geo::kIdsOff + (k % RING) * 1024) + g;
      char* dst = ring + (k % RING) * Geo::CHB;
      int idv[Geo::PER];
#pragma unroll
      for (int i = 0; i < Geo::PER; ++i) idv[i] = ids[i * NBG];
#pragma unroll
      for (int i = 0; i < Geo::PER; ++i) {
        const bool live = idv[i] >= 0 && cvalid;
        const char* src = live ? xcol + (uint32_t)idv[i] * rowbytes : zrow;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dst + i * 1024), 16, 0, 0);
      }
    };
    // ids run RING - 1 chunks ahead of the rows, the rows RING - 1 chunks ahead of the consumers
    for (int k = 0; k < RING - 1 && k < n; ++k) issue_ids(k);
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int k = 0; k < RING - 1; ++k) {
      if (k < n) issue_rows(k);
      if (k + RING - 1 < n) issue_ids(k + RING - 1);
    }
    for (int c = 0; c < n; ++c) {
      if (c + 2 * RING - 2 <= n) wait_vmcnt<Geo::kAhead>();  // chunk c and the ids of chunk c + RING - 1 have landed
      else wait_vmcnt<0>();                                   // (the queue is shorter at the end of the tile)
      __builtin_amdgcn_s_barrier();                           // ... and every consumer has finished chunk c - 1
      asm volatile("" ::: "memory");
      if (c + RING - 1 < n) issue_rows(c + RING - 1);
      if (c + 2 * RING - 2 < n) issue_ids(c + 2 * RING - 2);
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // the consumers' barrier before the direct part (see tile_loader)
    return;
  }

  // ---- consumer waves
  const int cw = wave - 1;
  v4f acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = (v4f)(0.f);

  if (n > 0) {
    const auto* cnt = as_const(reinterpret_cast<const v4u*>(a.lds_cnt));
    int64_t k = (int64_t)cbeg * NC + cw;
    const uint32_t* gs = a.lds_stream + (int64_t)as_const(a.lds_off)[(int64_t)tile * NC + cw] * NBG + lane * 4;
    char* swin = ring + Geo::kStreamOff + cw * 2048;  // 2 windows x 1 KiB
    auto dma = [&](int wi) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gs + (int64_t)wi * 256), (lds_ptr_t)(swin + (wi & 1) * 1024), 16, 0, 0);
    };
    dma(0);
    dma(1);
    wait_vmcnt<0>();
    int ss = 0;                       // supersteps consumed
    int landed = 2 * Geo::WS;         // supersteps [0, landed) are readable
    int issue_at = Geo::WS, nextw = 2;  // window nextw may overwrite its ring slot once superstep issue_at is the oldest still to be read
    auto word = [&](int i) -> uint32_t {
      return *reinterpret_cast<const uint32_t*>(swin + ((i / Geo::WS) & 1) * 1024 + (i % Geo::WS) * (NBG * 4) + g * 4);
    };
    auto prepare = [&](int lo, int hi) {  // the words of supersteps lo .. hi are about to be read
      if (hi >= landed) {
        wait_vmcnt<0>();
        landed = nextw * Geo::WS;
      }
      if (lo >= issue_at) {
        dma(nextw);
        ++nextw;
        issue_at += Geo::WS;
      }
    };
    uint32_t w0 = word(0), w1 = word(1);
    for (int c = 0; c < n; ++c) {
      const v4u cq = cnt[2 * k];
      const v4u cr = NACC > 8 ? cnt[2 * k + 1] : (v4u)(0u);
      const uint32_t cnts[8] = {cq.x, cq.y, cq.z, cq.w, cr.x, cr.y, cr.z, cr.w};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const uint32_t lrow = (uint32_t)(c % RING) * Geo::CHB + (uint32_t)l * 16u;
      auto gather4 = [&](uint32_t w, v4f (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const v4f*>(ring + ((((w >> (8 * u)) & 0xffu) << (LG + 4)) + lrow));
      };
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        const int nss = (int)((cnts[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
        int s = 0;
        for (; s + 2 <= nss; s += 2) {
          prepare(ss + 2, ss + 3);
          const uint32_t n0 = word(ss + 2), n1 = word(ss + 3);
          v4f va[4], vb[4];
          gather4(w0, va);
          gather4(w1, vb);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += va[u];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += vb[u];
          w0 = n0;
          w1 = n1;
          ss += 2;
        }
        if (s < nss) {
          prepare(ss + 2, ss + 2);
          const uint32_t n1 = word(ss + 2);
          v4f va[4];
          gather4(w0, va);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += va[u];
          w0 = w1;
          w1 = n1;
          ss += 1;
        }
      }
      k += NC;
    }
  }

  // ---- direct part: its stream windows (1 KiB = WSD supersteps, ring of 4) live where the chunks were
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    const int64_t k = (int64_t)tile * NC + cw;
    const int so = as_const(a.dir_off)[k], se = as_const(a.dir_off)[k + 1];
    if (se > so) {
      const auto* dcnt = as_const(reinterpret_cast<const v4u*>(a.dir_cnt)) + k * 4;
      const v4u c0 = dcnt[0];
      const v4u c1 = NACC > 4 ? dcnt[1] : (v4u)(0u);
      const v4u c2 = NACC > 8 ? dcnt[2] : (v4u)(0u);
      const uint32_t cnts[12] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w};
      const v4i* gd = reinterpret_cast<const v4i*>(a.dir_stream) + (int64_t)so * NBG + lane;
      char* dwin = ring + cw * 4096;
      auto dma = [&](int wi) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gd + (int64_t)wi * 64), (lds_ptr_t)(dwin + (wi & 3) * 1024), 16, 0, 0);
      };
      dma(0);
      dma(1);
      wait_vmcnt<1>();
      dma(2);
      int ss = 0, ready = Geo::WSD;
      auto ids = [&](int i) -> v4i { return *reinterpret_cast<const v4i*>(dwin + (i % (4 * Geo::WSD)) * (NBG * 16) + g * 16); };
      auto open_window = [&](int upto) {
        if (upto >= ready) {
          wait_vmcnt<1>();
          dma(ready / Geo::WSD + 2);
          ready += Geo::WSD;
        }
      };
      v4i i0 = ids(0), i1 = ids(1);
      const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)(cvalid ? col : 0) * 4u;
      auto gather4 = [&](const v4i& id, v4f (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const v4f*>(xb + (id[u] >= 0 ? (uint32_t)id[u] * rowbytes : 0u));
      };
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        const int nss = (int)cnts[j];
        int s = 0;
        for (; s + 2 <= nss; s += 2) {
          open_window(ss + 3);
          const v4i n0 = ids(ss + 2), n1 = ids(ss + 3);
          v4f va[4], vb[4];
          gather4(i0, va);
          gather4(i1, vb);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += i0[u] >= 0 ? va[u] : (v4f)(0.f);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += i1[u] >= 0 ? vb[u] : (v4f)(0.f);
          i0 = n0;
          i1 = n1;
          ss += 2;
        }
        if (s < nss) {
          open_window(ss + 2);
          const v4i n1 = ids(ss + 2);
          v4f va[4];
          gather4(i0, va);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[j] += i0[u] >= 0 ? va[u] : (v4f)(0.f);
          i0 = i1;
          i1 = n1;
          ss += 1;
        }
      }
    }
  }

  if (!cvalid) return;
#pragma unroll
  for (int j = 0; j < NACC; ++j) {
    const int item = a.tile_item[(int64_t)tile * (NC * NACC * NBG) + (cw * NACC + j) * NBG + g];
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
    } else {
      *reinterpret_cast<v4f*>(a.partial + (int64_t)(-(item + 1)) * a.D + col) = r;
    }
  }
}

#define MGX_TILE_LAUNCH(W_, NL_, NACC_, RING_)                                                            \
  do {                                                                                                   \
    hipLaunchKernelGGL((spmm_tile_kernel<W_, NL_, NACC_, RING_>), grid, dim3(W_ * kWave), 0, s, a);      \
    return true;                                                                                         \
  } while (0)

// (waves, loaders, rows per lane group) the library is built for
static bool launch_tile(int waves, int loaders, int nacc, const TileArgs& a, dim3 grid, hipStream_t s) {
  if (waves == 16 && loaders == 2) {
    if (nacc == 4) MGX_TILE_LAUNCH(16, 2, 4, 4);
    if (nacc == 5) MGX_TILE_LAUNCH(16, 2, 5, 4);
    if (nacc == 6) MGX_TILE_LAUNCH(16, 2, 6, 4);
    if (nacc == 8) MGX_TILE_LAUNCH(16, 2, 8, 4);
  } else if (waves == 16 && loaders == 4) {
    if (nacc == 4) MGX_TILE_LAUNCH(16, 4, 4, 4);
    if (nacc == 6) MGX_TILE_LAUNCH(16, 4, 6, 4);
    if (nacc == 8) MGX_TILE_LAUNCH(16, 4, 8, 4);
  } else if (waves == 8 && loaders == 1) {
    if (nacc == 4) MGX_TILE_LAUNCH(8, 1, 4, 2);
    if (nacc == 5) MGX_TILE_LAUNCH(8, 1, 5, 2);
    if (nacc == 6) MGX_TILE_LAUNCH(8, 1, 6, 2);
    if (nacc == 8) MGX_TILE_LAUNCH(8, 1, 8, 2);
    if (nacc == 10) MGX_TILE_LAUNCH(8, 1, 10, 2);
    if (nacc == 12) MGX_TILE_LAUNCH(8, 1, 12, 2);
  }
  return false;
}
#undef MGX_TILE_LAUNCH

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
  MGX_CHECK_ARG(tp->consumers >= 1 && tp->loaders >= 1 && (tp->consumers + tp->loaders == 16 || tp->consumers + tp->loaders == 8),
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
  a.tile_item = tp->tile_item; a.zero_row = tp->zero_row; a.tile_order = tp->tile_order;
  a.num_tiles = (int)tp->num_tiles;
  a.tiles_per_xcd = (int)((tp->num_tiles + kXcds - 1) / kXcds);
  a.D = (int)D; a.lds = (int)u_stride; a.ldo = (int)out_stride;
  a.mean = reduce == MGX_REDUCE_MEAN; a.accum = (flags & MGX_SPMM_ACCUMULATE) ? 1 : 0;
  const dim3 grid((unsigned)(a.tiles_per_xcd * kXcds), (unsigned)((D + kPassCols - 1) / kPassCols));
  hipStream_t s = (hipStream_t)stream;
  const int lg = tp->lanes_log2 == 0 ? 4 : tp->lanes_log2;
  if (lg != 4) {  // narrow rows: 8-wave workgroups (1 loader + 7 consumers), 32 / 16 columns per pass
    MGX_CHECK_ARG((lg == 3 || lg == 2) && tp->consumers == 7 && tp->loaders == 1, "mgx_spmm_tile_copy_u: narrow tile plans are 7 + 1 waves");
    const dim3 ngrid(grid.x, (unsigned)((D + (4 << lg) - 1) / (4 << lg)));
    const dim3 nblock(512);
    bool ok = true;
#define MGX_NARROW(LG_, NACC_) \
  if (lg == LG_ && tp->nacc == NACC_) { \
    hipLaunchKernelGGL((spmm_tile_narrow_kernel<LG_, NACC_, 2>), ngrid, nblock, 0, s, a); \
  } else
    MGX_NARROW(3, 3) MGX_NARROW(3, 4) MGX_NARROW(3, 5) MGX_NARROW(3, 6) MGX_NARROW(3, 8)
    MGX_NARROW(2, 2) MGX_NARROW(2, 3) MGX_NARROW(2, 4) MGX_NARROW(2, 6) MGX_NARROW(2, 8)
#undef MGX_NARROW
      ok = false;
    if (!ok) MGX_UNSUPPORTED("mgx_spmm_tile_copy_u: no narrow kernel for %d rows per lane group", tp->nacc);
  } else if (!launch_tile(tp->consumers + tp->loaders, tp->loaders, tp->nacc, a, grid, s)) {
    MGX_UNSUPPORTED("mgx_spmm_tile_copy_u: no kernel for %d consumer + %d loader waves with %d rows per lane group", tp->consumers,
                    tp->loaders, tp->nacc);
  }
  MGX_CHECK_LAUNCH();
  note_spmm_kernel("tile");
  if (hubs) return spmm_hub_fixup_launch(csr, plan, partial_ws, dst_scale, out, a.D, a.mean, a.accum, a.ldo, s);
  return MGX_OK;
}

