// spmm.hip -- g-SpMM for gfx950 (MI355X): CSR row-segmented wavefront reductions.
//
// Replaces what DGL's _CAPI_DGLKernelSpMM executes for the reference call sites
//   kernel/dgl-new.py:20, main_dgl_product_sage.py:62, main_dgl_reddit_gat.py:10 (GATConv),
//   main_dgl_proteins_rgcn_for.py:52, main_dgl_molhiv_gcn.py:46.
//
// Design (HBM/gather-bound integer+fp32 work; no MFMA):
//   * one 64-lane wavefront walks one destination row (SPLIT) or 64/G rows (one per lane group);
//   * lanes run ALONG THE FEATURE DIMENSION: a group of G lanes covers one neighbour's feature
//     row with 16-byte (VEC=4) loads, so each gathered row is read as whole 64..1024-byte
//     contiguous segments; 64/G neighbour rows are fetched by ONE wave-instruction and UNROLL
//     instructions are kept in flight (8..64 rows = 4..8 KiB per wave) to cover HBM latency;
//   * fp32 accumulation in registers, cross-group combine by xor-shuffles, one coalesced store;
//     no atomics => bitwise reproducible for a given graph;
//   * blockIdx is remapped so each XCD (own L2) owns a contiguous range of destination rows.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "slots.h"

namespace mgx {

enum { MODE_COPY_LHS = 0, MODE_MUL_EDGE = 1, MODE_COPY_RHS = 2 };

constexpr int kRowsPerBlock = 64;

template <typename Idx>
struct SpmmFastArgs {
  const Idx* indptr;
  const Idx* indices;
  const Idx* eids;
  const float* src;        // gathered matrix: U (copy_lhs, mul) or E (copy_rhs)
  const float* w;          // MODE_MUL_EDGE: [num_edges, H] weights addressed by edge id
  const float* src_scale;  // optional, per gathered row
  const float* dst_scale;  // optional, per output row
  const uint32_t* src_bits; // optional (lean copy_lhs kernel): bit u set = gathered row u may be non-zero; clear rows are skipped
  float* out;
  // optional schedule (mgx_spmm_plan): work items instead of natural row order
  const int32_t* item_row;  // >= 0: row written directly; < 0: partial slot -(v+1)
  const Idx* item_beg;
  const Idx* item_end;
  float* partial;           // [num_slots, D] partial sums of split (hub) rows
  int64_t n_items;  // == n_rows without a plan
  int64_t src_rows; // rows of the gathered matrix (num_cols, or nnz for copy_rhs)
  int64_t n_rows;
  XcdRanges xcd;    // lean kernel: item stretch of every XCD (edge balanced when the plan says so)
  const mgx_spmm_plan* plan;  // host side only
  int64_t nblocks;  // logical blocks, multiple of kXcds
  int rpb;          // work items per workgroup
  int D;            // elements per feature row
  int H;            // heads of w
  int F;            // D / H
  int mean;
  int accum;  // out += result (MGX_SPMM_ACCUMULATE)
  int ragged; // D % 4 != 0 handled with 16-byte gathers (RAGGED kernel)
  int short_rows;  // MGX_SPMM_SHORT_ROWS: the caller vouches for short, even work items -> one item per lane group (spmm_rowgroup32_kernel);
                   // 2: ... and this is the head of a two-part plan (no item above 32 edges)
  int lds;    // row stride of `src` in floats (>= D; == D unless mgx_spmm_copy_u_strided)
  int ldo;    // row stride of `out` in floats
};

// Work items per workgroup.  Workgroups are dispatched in blockIdx order, so a SMALL value makes
// the rows in flight on one XCD a tight window of the schedule (256 resident groups x rpb rows)
// and balances skewed rows better; 16 measured best on MI355X (64: +5..50 %, 256: +20..150 %; re-checked
// under two-part plans: profiles/r04_rest_launch_sweep.txt).
constexpr int kItemsPerBlock = 16;

template <int G, bool SPLIT>
struct Unroll {
  static constexpr int NB = kWave / G;
  static constexpr int value = SPLIT ? (NB >= 32 ? 1 : (NB >= 8 ? 2 : (NB >= 2 ? 4 : 8))) : (G >= 8 ? 4 : 2);
};

template <typename Idx, int VEC, int G, int MODE, bool SPLIT>
__global__ __launch_bounds__(kBlock) void spmm_fast_kernel(const SpmmFastArgs<Idx> a) {
  typedef typename VecT<VEC>::type V;
  constexpr int NB = kWave / G;
  constexpr int UNROLL = Unroll<G, SPLIT>::value;
  constexpr int ROWS_PER_STEP = SPLIT ? 1 : NB;
  constexpr int STEP = SPLIT ? NB : 1;

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int sub = lane / G;
  const int l = lane % G;
  const int f = (blockIdx.y * G + l) * VEC;
  const bool factive = f < a.D;
  const int head = (MODE == MODE_MUL_EDGE && factive) ? f / a.F : 0;
  const int64_t item_base = xcd_remap(blockIdx.x, a.nblocks) * a.rpb;
  const int64_t D = a.D;

  for (int r = wave * ROWS_PER_STEP; r < a.rpb; r += kWavesPerBlock * ROWS_PER_STEP) {
    if (item_base + r >= a.n_items) break;  // wave-uniform
    const int64_t item = item_base + r + (SPLIT ? 0 : sub);
    const bool iactive = item < a.n_items;
    int64_t row = 0, beg = 0, end = 0;
    if (iactive) {
      if (a.item_row) {
        row = (int64_t)a.item_row[item];
        beg = (int64_t)a.item_beg[item];
        end = (int64_t)a.item_end[item];
      } else {
        row = item;
        beg = (int64_t)a.indptr[item];
        end = (int64_t)a.indptr[item + 1];
      }
    }
    V acc = (V)(0.f);
    for (int64_t p = beg + (SPLIT ? sub : 0); p < end; p += (int64_t)STEP * UNROLL) {
      int64_t nbr[UNROLL];
      float wgt[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t q = p + (int64_t)u * STEP;
        nbr[u] = -1;
        wgt[u] = 1.f;
        if (q < end) {
          if (MODE == MODE_COPY_RHS) {
            nbr[u] = a.eids ? (int64_t)a.eids[q] : q;
          } else {
            nbr[u] = (int64_t)a.indices[q];
            if (MODE == MODE_MUL_EDGE) {
              const int64_t e = a.eids ? (int64_t)a.eids[q] : q;
              wgt[u] = a.w[e * a.H + head];
            }
          }
        }
      }
      if (a.src_scale) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
          if (nbr[u] >= 0) wgt[u] *= a.src_scale[nbr[u]];
      }
      V val[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        val[u] = (V)(0.f);
        if (nbr[u] >= 0 && factive) val[u] = *reinterpret_cast<const V*>(a.src + nbr[u] * D + f);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if (MODE == MODE_MUL_EDGE) acc += val[u] * wgt[u];
        else if (a.src_scale) acc += val[u] * wgt[u];
        else acc += val[u];
      }
    }
    if (SPLIT) {
#pragma unroll
      for (int off = G; off < kWave; off <<= 1) acc += vec_shfl_xor<VEC>(acc, off);
    }
    if (iactive && factive && (!SPLIT || sub == 0)) {
      if (row >= 0) {
        if (a.mean) {
          const int64_t deg = end - beg;
          acc = acc / (float)(deg > 1 ? deg : 1);
        }
        if (a.dst_scale) acc = acc * a.dst_scale[row];
        if (a.accum) acc += *reinterpret_cast<const V*>(a.out + row * D + f);
        *reinterpret_cast<V*>(a.out + row * D + f) = acc;
      } else {  // chunk of a split row: the epilogue runs in spmm_hub_fixup_kernel
        *reinterpret_cast<V*>(a.partial + (-(row + 1)) * D + f) = acc;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row-per-wave kernel, software pipelined (the SPLIT schedule for graphs whose rows feed all lane
// groups).  The dependent chain  item -> indptr -> indices -> feature rows  is what bounds a
// naive row loop (~7 serialized memory round trips per 50-edge row), so here:
//   * the up-to-64 neighbour ids of a row are fetched by ONE coalesced load (lane j <- edge j) and
//     handed to the lane groups with ds_bpermute (__shfl), never re-loaded per gather step;
//   * the NEXT item's bounds and neighbour ids are requested before the current row's gathers
//     are issued, so they travel under the row loads;
//   * per-edge scalars (edge id, edge weight with one head, source scale) ride along the same way.
template <int G>
struct RowwaveUnroll {
  static constexpr int NB = kWave / G;
  static constexpr int value = NB >= 16 ? 1 : (NB >= 8 ? 2 : (NB >= 2 ? 4 : 8));
};

template <typename Idx, int VEC, int G, int MODE, int U = RowwaveUnroll<G>::value>
__global__ __launch_bounds__(kBlock) void spmm_rowwave_kernel(const SpmmFastArgs<Idx> a) {
  typedef typename VecT<VEC>::type V;
  constexpr int NB = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int sub = lane / G;
  const int l = lane % G;
  const int f = (blockIdx.y * G + l) * VEC;
  const bool factive = f < a.D;
  const int head = (MODE == MODE_MUL_EDGE && factive) ? f / a.F : 0;
  const int64_t item_base = xcd_remap(blockIdx.x, a.nblocks) * a.rpb;
  const int64_t D = a.D;
  const bool w_per_edge = MODE == MODE_MUL_EDGE && a.H == 1;  // weight can ride with the ids

  auto load_meta = [&](int64_t item, int64_t& row, int64_t& beg, int64_t& end) {
    if (a.item_row) {
      row = (int64_t)a.item_row[item];
      beg = (int64_t)a.item_beg[item];
      end = (int64_t)a.item_end[item];
    } else {
      row = item;
      beg = (int64_t)a.indptr[item];
      end = (int64_t)a.indptr[item + 1];
    }
  };
  // per-lane slice of a row's edge list starting at `base`: gather id, edge id, scalar factor
  auto load_ids = [&](int64_t base, int64_t end, Idx& gid, Idx& eid, float& sc) {
    const int64_t q = base + lane;
    gid = (Idx)-1;
    eid = (Idx)-1;
    sc = 1.f;
    if (q < end) {
      if (MODE == MODE_COPY_RHS) {
        gid = a.eids ? a.eids[q] : (Idx)q;
      } else {
        gid = a.indices[q];
        if (MODE == MODE_MUL_EDGE) {
          eid = a.eids ? a.eids[q] : (Idx)q;
          if (w_per_edge) sc = a.w[(int64_t)eid];
        }
        if (a.src_scale) sc *= a.src_scale[(int64_t)gid];
      }
    }
  };

  int r = wave;
  if (r >= a.rpb || item_base + r >= a.n_items) return;
  int64_t row, beg, end;
  load_meta(item_base + r, row, beg, end);
  Idx gid, eid;
  float sc;
  load_ids(beg, end, gid, eid, sc);
  const bool scaled = MODE == MODE_MUL_EDGE || a.src_scale != nullptr;

  for (;;) {
    const int rn = r + kWavesPerBlock;
    const bool has_next = rn < a.rpb && item_base + rn < a.n_items;
    int64_t nrow = 0, nbeg = 0, nend = 0;
    Idx ngid = (Idx)-1, neid = (Idx)-1;
    float nsc = 1.f;
    if (has_next) {
      load_meta(item_base + rn, nrow, nbeg, nend);
      load_ids(nbeg, nend, ngid, neid, nsc);
    }
    V acc = (V)(0.f);
    for (int64_t base = beg; base < end; base += kWave) {
      if (base != beg) load_ids(base, end, gid, eid, sc);
      const int cnt = (int)((end - base) < kWave ? (end - base) : kWave);
      for (int k = 0; k < cnt; k += NB * U) {
        int64_t nbr[U];
        float wgt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = k + u * NB + sub;
          const Idx g = __shfl(gid, j & (kWave - 1), kWave);
          nbr[u] = j < cnt ? (int64_t)g : -1;
          wgt[u] = 1.f;
          if (scaled) {
            wgt[u] = __shfl(sc, j & (kWave - 1), kWave);
            if (MODE == MODE_MUL_EDGE && !w_per_edge) {
              const Idx e = __shfl(eid, j & (kWave - 1), kWave);
              if (j < cnt) wgt[u] *= a.w[(int64_t)e * a.H + head];
            }
          }
        }
        V val[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          val[u] = (V)(0.f);
          if (nbr[u] >= 0 && factive) val[u] = *reinterpret_cast<const V*>(a.src + nbr[u] * D + f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (scaled) acc += val[u] * wgt[u];
          else acc += val[u];
        }
      }
    }
#pragma unroll
    for (int off = G; off < kWave; off <<= 1) acc += vec_shfl_xor<VEC>(acc, off);
    if (factive && sub == 0) {
      if (row >= 0) {
        if (a.mean) {
          const int64_t deg = end - beg;
          acc = acc / (float)(deg > 1 ? deg : 1);
        }
        if (a.dst_scale) acc = acc * a.dst_scale[row];
        if (a.accum) acc += *reinterpret_cast<const V*>(a.out + row * D + f);
        *reinterpret_cast<V*>(a.out + row * D + f) = acc;
      } else {
        *reinterpret_cast<V*>(a.partial + (-(row + 1)) * D + f) = acc;
      }
    }
    if (!has_next) break;
    r = rn;
    row = nrow; beg = nbeg; end = nend;
    gid = ngid; eid = neid; sc = nsc;
  }
}

// ---------------------------------------------------------------------------------------------
// spmm_rowwave32_kernel: the row-per-wave kernel re-written for instruction count (int32 ids, gathered
// matrix < 4 GiB).  rocprofv3 showed the first version spent ~1500 VALU instructions per wave-row batch --
// 64-bit multiply-add per gather address, a compare + exec-mask branch around every load -- i.e. it was
// VALU-issue bound (≈1.6 ms of a 2.75 ms launch), not memory bound.  Here:
//   * lane j turns its neighbour id into a 32-bit BYTE OFFSET once (one v_mul per 64 edges); the offset,
//     not the id, travels through ds_bpermute, and the load uses the scalar-base + 32-bit-voffset form;
//   * steps whose 64/G*U edges are all valid run unpredicated; only the last step of a chunk clamps
//     its lane index and masks the loaded value;
//   * MODE / weight handling / feature-lane masking are template parameters, so the loop has no
//     uniform branches.
// WMODE: 0 = plain sum, 1 = one scalar per edge (u_mul_e with one head and/or src_scale), 2 = weight per
// (edge, head) loaded after the edge id arrives.  LANEMASK: D is not a multiple of the lane-group width.
// RAGGED (VEC = 4, D % 4 != 0, e.g. the 41-class output layer of the reddit GAT): rows are still gathered 16 bytes per lane
// (dword-aligned accesses); the lane that owns the last 1-3 columns loads the LAST FOUR floats of the row instead -- no
// read past the row, its own columns are the tail components of that window -- and only the epilogue distinguishes it.
// (8 waves per SIMD -- 64 VGPRs, two spilled -- was measured and changes nothing: profiles/r02_spmm_variants.txt.)
template <int VEC, int G, int MODE, int WMODE, bool LANEMASK, bool RAGGED = false, bool MASKED = false>
__global__ __launch_bounds__(kBlock) void spmm_rowwave32_kernel(const SpmmFastArgs<int32_t> a) {
  typedef typename VecT<VEC>::type VA;
  typedef VA VU __attribute__((aligned(4)));  // RAGGED: gathers / stores are only dword-aligned
  typedef typename std::conditional<RAGGED, VU, VA>::type V;
  constexpr int NB = kWave / G;
  constexpr int U = RowwaveUnroll<G>::value;
  constexpr int STEP = NB * U;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int sub = lane / G;
  const int l = lane % G;
  const int f = (blockIdx.y * G + l) * VEC;
  const bool factive = LANEMASK ? (f < a.D) : true;
  const int head = (WMODE == 2 && factive) ? f / a.F : 0;
  // block b serves XCD b % 8 (observed round-robin placement; speed only): its (b / 8)-th group of rpb items of that XCD's stretch
  int64_t item_base, item_stop;
  xcd_stretch(a.xcd, item_base, item_stop);
  item_base += (int64_t)(blockIdx.x / kXcds) * a.rpb;
  const uint32_t rowbytes = (uint32_t)a.lds * 4u;
  const int nvalid = RAGGED ? (a.D - f < VEC ? a.D - f : VEC) : VEC;  // columns this lane owns (RAGGED: the last lane < 4)
  const bool tail = RAGGED && factive && nvalid < VEC;
  // idle feature lanes re-read the first column of this pass (lane 0's cache line; never stored); the tail lane reads the
  // row's last four floats
  const uint32_t f4 = !factive ? (uint32_t)(blockIdx.y * G * VEC) * 4u : (tail ? (uint32_t)(a.D - VEC) * 4u : (uint32_t)f * 4u);
  const char* __restrict__ srcb = reinterpret_cast<const char*>(a.src);
  const int bidx0 = sub * 4;  // byte index of this lane group's first edge for ds_bpermute

  auto load_meta = [&](int64_t item, int64_t& row, int32_t& beg, int32_t& end) {
    if (a.item_row) {
      row = (int64_t)a.item_row[item];
      beg = a.item_beg[item];
      end = a.item_end[item];
    } else {
      row = item;
      beg = a.indptr[item];
      end = a.indptr[item + 1];
    }
  };
  // lane j's slice of the edge list: byte offset of the gathered row, edge id, per-edge scalar
  // `cnt`: edges of this 64-edge chunk to gather.  MASKED (row-sparse gathered matrix, e.g. the gradient of a loss taken on
  // 8 % of the nodes): load_ids only fetches the ids (`gid`, -1 past the end); the flag word of every source row is fetched
  // by mask_fetch once the ids are there, and mask_compact drops the lanes whose row is all-zero and compacts the live
  // offsets to lanes 0..cnt-1 (stable partition through ds_permute) -- three stages so that the NEXT item's flag gather can
  // be issued behind the current item's row gathers and be consumed only after its epilogue.
  auto load_ids = [&](int32_t base, int32_t end, uint32_t& goff, int32_t& eid, float& sc, int& cnt, int32_t& gidx) {
    const int32_t q = base + lane;
    goff = 0;
    eid = 0;
    sc = 1.f;
    gidx = -1;
    if (q < end) {
      int32_t gid;
      if (MODE == MODE_COPY_RHS) {
        gid = a.eids ? __builtin_nontemporal_load(&a.eids[q]) : q;
      } else {
        gid = __builtin_nontemporal_load(&a.indices[q]);  // streamed once: keep L2 for the gathered rows
        if (MODE == MODE_MUL_EDGE) {
          eid = a.eids ? __builtin_nontemporal_load(&a.eids[q]) : q;
          if (WMODE == 1) sc = a.w[eid];
        }
        if (WMODE == 1 && a.src_scale) sc *= a.src_scale[gid];
      }
      goff = (uint32_t)gid * rowbytes;
      gidx = gid;
    }
    cnt = (end - base) < kWave ? (end - base) : kWave;
  };
  auto mask_fetch = [&](int32_t gidx) -> uint32_t {
    return gidx >= 0 ? a.src_bits[(uint32_t)gidx >> 5] : 0u;
  };
  auto mask_compact = [&](uint32_t word, int32_t gidx, uint32_t& goff, int& cnt) {
    const bool live = gidx >= 0 && ((word >> ((uint32_t)gidx & 31u)) & 1u) != 0u;
    const uint64_t b = __ballot(live);
    const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
    const int nlive = __popcll(b);
    const int dest = live ? below : nlive + (lane - below);  // a permutation of 0..63: live lanes first, order kept
    goff = (uint32_t)__builtin_amdgcn_ds_permute(dest * 4, (int)goff);
    cnt = nlive;
  };

  // gathers of one step (STEP edges starting at edge k of the current 64-edge chunk); lanes past the end of
  // the chunk re-read edge 0 (valid memory, same cache line as a live lane) and are zeroed in consume_step
  auto issue_step = [&](int k, int cnt, uint32_t goff, int32_t eid, float sc, V (&val)[U], float (&wgt)[U]) {
    const bool full = k + STEP <= cnt;  // wave-uniform
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int bi = bidx0 + (k + u * NB) * 4;
      if (!full) bi = (k + u * NB + sub < cnt) ? bi : 0;
      const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute(bi, (int)goff) + f4;
      wgt[u] = 1.f;
      if (WMODE == 1) wgt[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(bi, __float_as_int(sc)));
      if (WMODE == 2) {
        const int e = __builtin_amdgcn_ds_bpermute(bi, eid);
        wgt[u] = a.w[(int64_t)e * a.H + head];
      }
      val[u] = *reinterpret_cast<const V*>(srcb + off);
    }
  };
  auto consume_step = [&](int k, int cnt, const V (&val)[U], const float (&wgt)[U], V& acc) {
    if (k + STEP <= cnt) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (WMODE != 0) acc += val[u] * wgt[u];
        else acc += val[u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const V vv = (k + u * NB + sub < cnt) ? val[u] : (V)(0.f);
        if (WMODE != 0) acc += vv * wgt[u];
        else acc += vv;
      }
    }
  };

  int r = wave;
  if (r >= a.rpb || item_base + r >= item_stop) return;
  const bool early_accum = a.accum && !a.mean && a.dst_scale == nullptr;  // wave-uniform: see the accumulator's initial value below
  int64_t row;
  int32_t beg, end;
  load_meta(item_base + r, row, beg, end);
  uint32_t goff;
  int32_t eid;
  float sc;
  int cnt0 = 0;
  int32_t gidx = -1;
  load_ids(beg, end, goff, eid, sc, cnt0, gidx);
  if (MASKED) mask_compact(mask_fetch(gidx), gidx, goff, cnt0);

  for (;;) {
    const int rn = r + kWavesPerBlock;
    const bool has_next = rn < a.rpb && item_base + rn < item_stop;
    int64_t nrow = 0;
    int32_t nbeg = 0, nend = 0, neid = 0;
    uint32_t ngoff = 0;
    float nsc = 1.f;
    int ncnt0 = 0;
    int32_t ngidx = -1;
    if (has_next) {
      load_meta(item_base + rn, nrow, nbeg, nend);
      load_ids(nbeg, nend, ngoff, neid, nsc, ncnt0, ngidx);
    }
    // out += result (MGX_SPMM_ACCUMULATE: the backward aggregation of the one-node SAGE layer): the row to add to is requested
    // BEFORE the item's gathers, not in the epilogue, where its round trip was paid per item with nothing left to hide it behind
    // (products D = 64: the accumulating launch ran 2.63 ms against 2.26 ms for the plain one for 0.11 ms worth of bytes)
    // -- as the INITIAL value of lane group 0's accumulator (no register of its own: 65 VGPRs, 7 waves per SIMD as before), which is
    // exact when nothing scales the sum afterwards (no mean, no dst_scale: the backward launches); scaled sums add it in the epilogue.
    V acc = (V)(0.f);
    if (early_accum && row >= 0 && factive && sub == 0 && !tail) acc = *reinterpret_cast<const V*>(a.out + row * (int64_t)a.ldo + f);
    for (int32_t cbase = beg; cbase < end; cbase += kWave) {
      int cnt = cnt0;
      if (cbase != beg) {
        load_ids(cbase, end, goff, eid, sc, cnt, gidx);
        if (MASKED) mask_compact(mask_fetch(gidx), gidx, goff, cnt);
      }
      if (MASKED && cnt == 0) continue;  // every source row of this chunk is zero
      // Two value buffers in ping-pong: the gathers of step k+1 are issued before step k is summed, so
      // consecutive steps of a row overlap instead of paying one memory round trip each.
      V va[U], vb[U];
      float wa[U], wb[U];
      issue_step(0, cnt, goff, eid, sc, va, wa);
      int k = STEP;
      for (;;) {
        if (k >= cnt) { consume_step(k - STEP, cnt, va, wa, acc); break; }
        issue_step(k, cnt, goff, eid, sc, vb, wb);
        consume_step(k - STEP, cnt, va, wa, acc);
        k += STEP;
        if (k >= cnt) { consume_step(k - STEP, cnt, vb, wb, acc); break; }
        issue_step(k, cnt, goff, eid, sc, va, wa);
        consume_step(k - STEP, cnt, vb, wb, acc);
        k += STEP;
      }
    }
    // the next item's flag words: its ids arrived during the gathers above; this load travels under the epilogue below
    uint32_t nword = 0;
    if (MASKED && has_next) nword = mask_fetch(ngidx);
#pragma unroll
    for (int off = G; off < kWave; off <<= 1) acc += vec_shfl_xor<VEC>(acc, off);
    if (factive && sub == 0) {
      if (row >= 0) {
        if (a.mean) {
          const int deg = end - beg;
          acc = acc / (float)(deg > 1 ? deg : 1);
        }
        if (a.dst_scale) acc = acc * a.dst_scale[row];
      }
      if (!tail) {
        if (row >= 0) {  // separate pointers: the output store keeps its non-temporal hint
          float* op = a.out + row * (int64_t)a.ldo + f;
          if (a.accum && !early_accum) acc += *reinterpret_cast<const V*>(op);
          __builtin_nontemporal_store((VA)acc, reinterpret_cast<V*>(op));
        } else {
          *reinterpret_cast<V*>(a.partial + (-(row + 1)) * (int64_t)a.D + f) = acc;
        }
      } else {  // own columns f .. f+nvalid-1 are components VEC-nvalid .. VEC-1 of the window
        float* op = (row >= 0 ? a.out + row * (int64_t)a.ldo : a.partial + (-(row + 1)) * (int64_t)a.D) + f;
        const float* av = reinterpret_cast<const float*>(&acc);
        for (int j = 0; j < nvalid; ++j) {
          float v = av[VEC - nvalid + j];
          if (row >= 0 && a.accum) v += op[j];
          op[j] = v;
        }
      }
    }
    if (!has_next) break;
    r = rn;
    row = nrow; beg = nbeg; end = nend;
    goff = ngoff; eid = neid; sc = nsc;
    cnt0 = ncnt0;
    if (MASKED) mask_compact(nword, ngidx, goff, cnt0);
  }
}

// ---------------------------------------------------------------------------------------------
// spmm_rowgroup32_kernel (round 4): SHORT rows -- one work item per LANE GROUP, 64 / G items per wave at a time.
// The row-per-wave kernel above spends one dependent chain (item -> ids -> gathers -> store, ~6 us from beyond L2) per row
// whatever its length up to 16 edges; the CSRs of a partition's halo (mi355x_graph/dist.py: 3.4 edges per row backward, 11
// forward, profiles/r04_scale_model.txt), of arxiv-shaped graphs (6.9) and of batched molecules (2.2) are all of that kind and
// run latency-bound on it.  Here a lane group of G lanes owns an item: its lanes load the item's first G ids with one
// coalesced access, the ids travel inside the group through ds_bpermute, and the group gathers its row's neighbours U at a
// time with 16-byte lanes along the feature dimension -- 64 / G chains per wave instead of one, and no cross-group reduction
// at the end.  Terms are added in storage order (exactly the CPU oracle's order for rows that are not split).
// COPY_LHS / COPY_RHS, sum / mean, dst_scale, accumulate, row strides, plan items with partial slots; int32 ids, gathered
// matrix below 4 GiB, D % 4 == 0.  The next batch's bounds and ids are requested before the current batch's gathers.
// LONG = false: the head of a two-part plan, whose items are short by contract -- the whole-wave path for long items is compiled out
// (registers: see launch_rowgroup32); a longer item would still be summed correctly, by its lane group alone.
template <int VEC, int G, int MODE, bool LANEMASK, bool LONG>
__global__ __launch_bounds__(kBlock) void spmm_rowgroup32_kernel(const SpmmFastArgs<int32_t> a) {
  typedef typename VecT<VEC>::type V;
  constexpr int NB = kWave / G;
  constexpr int U = 4;
  constexpr int kLong = 32;  // edges a lane group walks alone (8 steps of 4 gathers); longer items: the whole wave, see below
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int sub = lane / G;
  const int l = lane % G;
  const int f = (blockIdx.y * G + l) * VEC;
  const bool factive = LANEMASK ? (f < a.D) : true;
  int64_t item_base, item_stop;
  xcd_stretch(a.xcd, item_base, item_stop);
  item_base += (int64_t)(blockIdx.x / kXcds) * a.rpb;
  const uint32_t rowbytes = (uint32_t)a.lds * 4u;
  const uint32_t f4 = factive ? (uint32_t)f * 4u : (uint32_t)(blockIdx.y * G * VEC) * 4u;  // idle lanes re-read column 0 of the pass
  const char* __restrict__ srcb = reinterpret_cast<const char*>(a.src);
  const int gbase = sub * G;

  auto load_meta = [&](int r, bool& ok, int64_t& row, int32_t& beg, int32_t& end) {
    const int64_t item = item_base + r;
    ok = r < a.rpb && item < item_stop;
    row = 0; beg = 0; end = 0;
    if (ok) {
      if (a.item_row) {
        row = (int64_t)a.item_row[item];
        beg = a.item_beg[item];
        end = a.item_end[item];
      } else {
        row = item;
        beg = a.indptr[item];
        end = a.indptr[item + 1];
      }
    }
  };
  auto load_ids = [&](int32_t base, int32_t end) -> uint32_t {  // lane l of the group: byte offset of neighbour base + l
    const int32_t q = base + l;
    uint32_t goff = 0;
    if (q < end) {
      int32_t gid;
      if (MODE == MODE_COPY_RHS) gid = a.eids ? __builtin_nontemporal_load(&a.eids[q]) : q;
      else gid = __builtin_nontemporal_load(&a.indices[q]);
      goff = (uint32_t)gid * rowbytes;
    }
    return goff;
  };

  int r = wave * NB + sub;  // this group's item inside the workgroup's stretch; batches of kWavesPerBlock * NB items
  bool ok;
  int64_t row;
  int32_t beg, end;
  load_meta(r, ok, row, beg, end);
  uint32_t goff = load_ids(beg, end);
  for (;;) {
    const int rn = r + kWavesPerBlock * NB;
    const bool more = __builtin_amdgcn_readfirstlane(rn - sub) < a.rpb && item_base + (rn - sub) < item_stop;  // wave-uniform
    bool nok = false;
    int64_t nrow = 0;
    int32_t nbeg = 0, nend = 0;
    uint32_t ngoff = 0;
    if (more) {
      load_meta(rn, nok, nrow, nbeg, nend);
      ngoff = load_ids(nbeg, nend);
    }
    V acc = (V)(0.f);
    // items longer than kLong edges are NOT walked by their lane group alone (one group gathering 256 edges, 4 in flight, is 64
    // dependent round trips -- the tail of the whole launch on a skewed graph): the wave takes them together below
    const bool deferred = LONG && ok && (end - beg) > kLong;
    const int32_t gend = deferred ? beg : end;
    int32_t cbase = beg;
    for (;;) {  // chunks of G edges; wave-uniform trip count (the longest item of the batch), per-group predicates inside
      const int cnt = (gend - cbase) < G ? (gend - cbase) : G;
      for (int k = 0; k < G; k += U) {
        if (__builtin_amdgcn_ballot_w64(k < cnt) == 0) break;
        V val[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = k + u < cnt;
          const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute((gbase + (live ? k + u : 0)) * 4, (int)goff) + f4;
          val[u] = (V)(0.f);
          if (live) val[u] = *reinterpret_cast<const V*>(srcb + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += val[u];
      }
      cbase += G;
      if (__builtin_amdgcn_ballot_w64(cbase < gend) == 0) break;
      goff = load_ids(cbase, gend);
    }
    // ---- long items of this batch, one after the other, by the WHOLE wave: 64 ids per coalesced load, lane group `sub` gathers edges
    // k + u * NB + sub, the groups' partial sums meet in an xor-shuffle tree, the owner group keeps the result
    uint64_t dmask = LONG ? __builtin_amdgcn_ballot_w64(deferred) : 0;
    while (LONG && dmask) {
      const int src_lane = __builtin_amdgcn_readfirstlane(__builtin_ctzll(dmask));
      const int gsel = src_lane / G;
      const int32_t cb = __builtin_amdgcn_readlane(beg, src_lane), ce = __builtin_amdgcn_readlane(end, src_lane);
      V part = (V)(0.f);
      for (int32_t base = cb; base < ce; base += kWave) {
        const int32_t q = base + lane;
        uint32_t go = 0;
        if (q < ce) {
          int32_t gid;
          if (MODE == MODE_COPY_RHS) gid = a.eids ? __builtin_nontemporal_load(&a.eids[q]) : q;
          else gid = __builtin_nontemporal_load(&a.indices[q]);
          go = (uint32_t)gid * rowbytes;
        }
        const int ccnt = (ce - base) < kWave ? (ce - base) : kWave;
        for (int k = 0; k < ccnt; k += NB * U) {
          V val[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int e = k + u * NB + sub;
            const bool live = e < ccnt;
            const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute((live ? e : 0) * 4, (int)go) + f4;
            val[u] = (V)(0.f);
            if (live) val[u] = *reinterpret_cast<const V*>(srcb + off);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) part += val[u];
        }
      }
#pragma unroll
      for (int off = G; off < kWave; off <<= 1) part += vec_shfl_xor<VEC>(part, off);
      if (sub == gsel) acc = part;  // (its own accumulator is still zero: the item was deferred)
      dmask &= ~((G >= 64 ? ~0ull : ((1ull << G) - 1ull)) << (gsel * G));
    }
    if (ok && factive) {
      if (row >= 0) {
        if (a.mean) {
          const int deg = end - beg;
          acc = acc / (float)(deg > 1 ? deg : 1);
        }
        if (a.dst_scale) acc = acc * a.dst_scale[row];
        float* op = a.out + row * (int64_t)a.ldo + f;
        if (a.accum) acc += *reinterpret_cast<const V*>(op);
        __builtin_nontemporal_store(acc, reinterpret_cast<V*>(op));
      } else {
        *reinterpret_cast<V*>(a.partial + (-(row + 1)) * (int64_t)a.D + f) = acc;
      }
    }
    if (!more) break;
    r = rn;
    ok = nok; row = nrow; beg = nbeg; end = nend; goff = ngoff;
  }
}

template <int VEC, int G, int MODE>
static bool launch_rowgroup32(SpmmFastArgs<int32_t> a, int64_t nnz, hipStream_t s) {
  // Taken when the CALLER vouches for short, even work items (MGX_SPMM_SHORT_ROWS in `flags`).  An average-only rule decided here made
  // the arxiv-shaped graph (6.9 in-edges on average, power-law tail, a 13 k-edge hub = 52 consecutive 256-edge chunks) TWICE AS SLOW
  // at D = 4 .. 32 (81 -> 180 us) while uniform short rows gain 2 - 3.5x (profiles/r04_rowgroup.txt, r04_rowgroup_skew.txt); the host
  // layer decides per CSR and lane-group count from the item lengths (mi355x_graph/sparse.py: CsrView.short_rows).
  constexpr int NB = kWave / G;
  if constexpr (NB < 2 || VEC != 4 || MODE == MODE_MUL_EDGE) {
    return false;
  } else {
    if (!a.short_rows || a.ragged || a.src_scale || a.src_bits || a.D % 4 != 0) return false;
    if (a.src_rows * (int64_t)a.lds * 4 >= (int64_t(1) << 32)) return false;
    (void)nnz;
    note_spmm_kernel("rowgroup32");
    a.rpb = 2 * kWavesPerBlock * NB;  // two batches per wave: the second's ids travel under the first's gathers
    const dim3 grid((unsigned)xcd_ranges(a.plan, a.n_items, a.rpb, a.xcd), (unsigned)((a.D + G * VEC - 1) / (G * VEC)));
    if (a.short_rows == 2) {  // head of a two-part plan: every item short
      if (a.D % (G * VEC) != 0) hipLaunchKernelGGL((spmm_rowgroup32_kernel<VEC, G, MODE, true, false>), grid, dim3(kBlock), 0, s, a);
      else hipLaunchKernelGGL((spmm_rowgroup32_kernel<VEC, G, MODE, false, false>), grid, dim3(kBlock), 0, s, a);
    } else {
      if (a.D % (G * VEC) != 0) hipLaunchKernelGGL((spmm_rowgroup32_kernel<VEC, G, MODE, true, true>), grid, dim3(kBlock), 0, s, a);
      else hipLaunchKernelGGL((spmm_rowgroup32_kernel<VEC, G, MODE, false, true>), grid, dim3(kBlock), 0, s, a);
    }
    return true;
  }
}
template <int VEC, int G, int MODE>
static bool launch_rowgroup32(SpmmFastArgs<int64_t>, int64_t, hipStream_t) { return false; }

// Sums the partial slots of every split (hub) row in slot order -- deterministic -- and applies
// the mean / dst_scale epilogue.  One wave per hub row, lanes along the feature dimension.
template <typename Idx>
__global__ __launch_bounds__(kBlock) void spmm_hub_fixup_kernel(const Idx* indptr, const int32_t* hub_row,
                                                                const int32_t* hub_slot_ptr, int64_t num_hubs,
                                                                const float* partial, const float* dst_scale,
                                                                float* out, int D, int mean, int accum, int ldo) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t h = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  if (h >= num_hubs) return;
  const int64_t row = hub_row[h];
  const int s0 = hub_slot_ptr[h], s1 = hub_slot_ptr[h + 1];
  float scale = 1.f;
  for (int k = lane; k < D; k += kWave) {
    float acc = 0.f;
    int s = s0;
    for (; s + 4 <= s1; s += 4) {  // four slots in flight, added in slot order (the same bits as one at a time)
      const float a0 = partial[(int64_t)s * D + k], a1 = partial[(int64_t)(s + 1) * D + k];
      const float a2 = partial[(int64_t)(s + 2) * D + k], a3 = partial[(int64_t)(s + 3) * D + k];
      acc += a0; acc += a1; acc += a2; acc += a3;
    }
    for (; s < s1; ++s) acc += partial[(int64_t)s * D + k];
    if (mean) {
      const int64_t deg = (int64_t)indptr[row + 1] - (int64_t)indptr[row];
      acc = acc / (float)(deg > 1 ? deg : 1);
    }
    if (dst_scale) acc *= dst_scale[row];
    if (accum) acc += out[row * ldo + k];
    out[row * ldo + k] = acc * scale;
  }
}

// used by spmm_tile.hip: the same fix-up pass after the tile kernel
int32_t spmm_hub_fixup_launch(const mgx_csr* csr, const mgx_spmm_plan* plan, const float* partial, const float* dst_scale, float* out,
                              int D, int mean, int accum, int ldo, hipStream_t s) {
  hipLaunchKernelGGL((spmm_hub_fixup_kernel<int32_t>), dim3((unsigned)((plan->num_hubs + kWavesPerBlock - 1) / kWavesPerBlock)),
                     dim3(kBlock), 0, s, (const int32_t*)csr->indptr, plan->hub_row, plan->hub_slot_ptr, plan->num_hubs, partial,
                     dst_scale, out, D, mean, accum, ldo);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

// ---------------------------------------------------------------------------------------------
// Generic kernel: any op x any reduce x arbitrary broadcast tables.  One wave per row, lanes over
// output elements, serial over the row's edges (storage order => same order as the CPU oracle).
template <typename Idx>
struct SpmmGenericArgs {
  const Idx* indptr;
  const Idx* indices;
  const Idx* eids;
  const float* U;
  const float* E;
  const int64_t* u_off;
  const int64_t* e_off;
  const float* src_scale;
  const float* dst_scale;
  float* out;
  Idx* arg_u;
  Idx* arg_e;
  int64_t n_rows;
  int64_t nblocks;
  int64_t u_len, e_len, out_len;
  int op, reduce, accum;
};

__device__ __forceinline__ float apply_op(int op, float l, float r) {
  switch (op) {
    case MGX_OP_ADD: return l + r;
    case MGX_OP_SUB: return l - r;
    case MGX_OP_MUL: return l * r;
    case MGX_OP_DIV: return l / r;
    case MGX_OP_COPY_LHS: return l;
    default: return r;
  }
}

template <typename Idx>
__global__ __launch_bounds__(kBlock) void spmm_generic_kernel(const SpmmGenericArgs<Idx> a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int64_t row_base = xcd_remap(blockIdx.x, a.nblocks) * kRowsPerBlock;
  for (int r = wave; r < kRowsPerBlock; r += kWavesPerBlock) {
    const int64_t row = row_base + r;
    if (row >= a.n_rows) break;
    const int64_t beg = (int64_t)a.indptr[row], end = (int64_t)a.indptr[row + 1];
    for (int64_t k = lane; k < a.out_len; k += kWave) {
      // NULL table: identity, or head-wise broadcast when the operand row is shorter than the output row
      const int64_t uo = !a.U ? 0 : a.u_off ? a.u_off[k] : (a.u_len == a.out_len ? k : k / (a.out_len / a.u_len));
      const int64_t eo = !a.E ? 0 : a.e_off ? a.e_off[k] : (a.e_len == a.out_len ? k : k / (a.out_len / a.e_len));
      float acc = a.reduce == MGX_REDUCE_MAX ? -INFINITY : (a.reduce == MGX_REDUCE_MIN ? INFINITY : 0.f);
      int64_t bu = -1, be = -1;
      for (int64_t p = beg; p < end; ++p) {
        const int64_t u = a.indices ? (int64_t)a.indices[p] : 0;
        const int64_t e = a.eids ? (int64_t)a.eids[p] : p;
        float lv = a.U ? a.U[u * a.u_len + uo] : 0.f;
        if (a.src_scale) lv *= a.src_scale[u];
        const float rv = a.E ? a.E[e * a.e_len + eo] : 0.f;
        const float val = apply_op(a.op, lv, rv);
        if (a.reduce == MGX_REDUCE_MAX) {
          if (val > acc) { acc = val; bu = u; be = e; }
        } else if (a.reduce == MGX_REDUCE_MIN) {
          if (val < acc) { acc = val; bu = u; be = e; }
        } else {
          acc += val;
        }
      }
      if (a.reduce == MGX_REDUCE_MAX || a.reduce == MGX_REDUCE_MIN) {
        if (beg == end) acc = 0.f;
        if (a.arg_u) a.arg_u[row * a.out_len + k] = (Idx)bu;
        if (a.arg_e) a.arg_e[row * a.out_len + k] = (Idx)be;
      } else if (a.reduce == MGX_REDUCE_MEAN) {
        const int64_t deg = end - beg;
        acc = acc / (float)(deg > 1 ? deg : 1);
      }
      if (a.dst_scale) acc *= a.dst_scale[row];
      if (a.accum) acc += a.out[row * a.out_len + k];
      a.out[row * a.out_len + k] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// int32 graphs whose gathered matrix is addressable with 32-bit byte offsets take the lean kernel
template <int VEC, int G, int MODE>
static bool launch_rowwave32(const SpmmFastArgs<int32_t>& a, int64_t src_rows, dim3 grid, hipStream_t s) {
  if (src_rows * (int64_t)a.lds * 4 >= (int64_t(1) << 32)) return false;
  if (a.ragged) {  // VEC == 4, D % 4 != 0, one weight per edge at most (launch_fast checked eligibility)
    if (VEC == 4 && MODE != MODE_COPY_RHS) {
      if (MODE == MODE_MUL_EDGE || a.src_scale) hipLaunchKernelGGL((spmm_rowwave32_kernel<VEC, G, MODE, 1, true, true>), grid, dim3(kBlock), 0, s, a);
      else hipLaunchKernelGGL((spmm_rowwave32_kernel<VEC, G, MODE, 0, true, true>), grid, dim3(kBlock), 0, s, a);
    }
    return true;
  }
  const bool lanemask = a.D % (G * VEC) != 0;
  int wmode = 0;
  if (MODE == MODE_MUL_EDGE) wmode = a.H == 1 ? 1 : 2;
  else if (a.src_scale) wmode = 1;
  if (MODE == MODE_COPY_LHS && a.src_bits && wmode == 0) {  // row-sparse gathered matrix: skip rows flagged all-zero
    if (lanemask) hipLaunchKernelGGL((spmm_rowwave32_kernel<VEC, G, MODE, 0, true, false, true>), grid, dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((spmm_rowwave32_kernel<VEC, G, MODE, 0, false, false, true>), grid, dim3(kBlock), 0, s, a);
    return true;
  }
  if (MODE == MODE_MUL_EDGE && a.H > 1 && a.src_scale) return false;  // not needed by any caller
#define MGX_RW32(W, LM) hipLaunchKernelGGL((spmm_rowwave32_kernel<VEC, G, MODE, W, LM>), grid, dim3(kBlock), 0, s, a)
  if (wmode == 0) { if (lanemask) MGX_RW32(0, true); else MGX_RW32(0, false); }
  else if (wmode == 1) { if (lanemask) MGX_RW32(1, true); else MGX_RW32(1, false); }
  else { if (lanemask) MGX_RW32(2, true); else MGX_RW32(2, false); }
#undef MGX_RW32
  return true;
}
template <int VEC, int G, int MODE>
static bool launch_rowwave32(const SpmmFastArgs<int64_t>&, int64_t, dim3, hipStream_t) { return false; }

template <typename Idx, int VEC, int G, int MODE>
static void launch_fast_g(SpmmFastArgs<Idx> a, bool split, int64_t nnz, hipStream_t s) {
  constexpr int NB = kWave / G;
  if (a.ragged) split = true;  // only the lean row-per-wave kernel implements the ragged 16-byte tail window
  if (a.lds != a.D || a.ldo != a.D) split = true;  // ... and row strides (spmm_fast_kernel ignores lds / ldo)
  if (launch_rowgroup32<VEC, G, MODE>(a, nnz, s)) return;  // short rows: one item per lane group (lean, int32)
  a.rpb = kItemsPerBlock;
  if (!split && a.rpb < kWavesPerBlock * NB) a.rpb = kWavesPerBlock * NB;  // one item per lane group
  a.nblocks = round_up((a.n_items + a.rpb - 1) / a.rpb, kXcds);
  dim3 grid((unsigned)a.nblocks, (unsigned)((a.D + G * VEC - 1) / (G * VEC)));
  {  // lean kernel: per-XCD stretches of the schedule
    const dim3 lgrid((unsigned)xcd_ranges(a.plan, a.n_items, a.rpb, a.xcd), grid.y);
    if (split && launch_rowwave32<VEC, G, MODE>(a, a.src_rows, lgrid, s)) { note_spmm_kernel("rowwave32"); return; }
  }
  note_spmm_kernel(split ? "rowwave" : "fast");
  if (split) hipLaunchKernelGGL((spmm_rowwave_kernel<Idx, VEC, G, MODE>), grid, dim3(kBlock), 0, s, a);
  else hipLaunchKernelGGL((spmm_fast_kernel<Idx, VEC, G, MODE, false>), grid, dim3(kBlock), 0, s, a);
}

template <typename Idx, int VEC, int MODE>
static void launch_fast_v(const SpmmFastArgs<Idx>& a, int64_t nnz, hipStream_t s) {
  const int lanes = (a.D + VEC - 1) / VEC;
  int G = 1;
  while (G < lanes && G < kWave) G <<= 1;
  // Rows wider than 32 lanes are aggregated in column passes of 32 lanes (grid.y): a wave whose 64 lanes all follow ONE edge
  // issues one gather per edge, two lane groups of 32 issue one per two edges -- measured at D = 256 / 512 / 608, G = 64 -> 32:
  // reddit-shaped (492 in-edges per node) 7.24 -> 6.12, 14.45 -> 12.17, 18.19 -> 15.13 ms; proteins-shaped 5.12 -> 4.19 ms at
  // D = 256; products-shaped 10.37 -> 9.91 ms at D = 256 (experiments/exp_wide_ragged.py).  The index stream is re-read per
  // pass (4 bytes per edge against 512 gathered).  A second pass that owns only a few columns costs most of a full one
  // (reddit-shaped, D = 132 / 144 / 160 / 192: one 64-lane pass 4.24 / 4.07 / 4.12 / 5.08 ms, 32 + remainder 5.64 / 5.26 / 4.88 /
  // 4.96 ms), so rows of 33..44 lanes keep the single 64-lane pass.
  const double avg_deg = a.n_rows > 0 ? (double)nnz / (double)a.n_rows : 0.0;
  // short rows: the per-row cost of a second pass outweighs it (arxiv-shaped, 6.9 in-edges per node, D = 256: 0.202 -> 0.221 ms)
  if (MODE == MODE_COPY_LHS && G > 32 && lanes > 44 && avg_deg >= 24.0) G = 32;
  const int NB = kWave / G;
  // One row per wave (lane groups share the row's edges) or one row per lane group.  The lean int32 kernel is faster
  // than the row-per-group kernel at every degree measured (arxiv-shaped, avg in-degree 6.9: D = 64 187 -> 108 us, D = 8
  // 148 -> 82 us; cora / pubmed 40-54 -> 22-26 us), so it is always taken when eligible; the 64-bit kernels keep the old
  // rule (rows long enough to feed all NB lane groups).
  const bool lean = sizeof(Idx) == 4 && a.src_rows * (int64_t)a.lds * 4 < (int64_t(1) << 32);
  const double split_factor = lean ? 0.0 : 2.0;
  const bool split = (NB == 1) || (avg_deg >= split_factor * NB);
  switch (G) {
    case 1: launch_fast_g<Idx, VEC, 1, MODE>(a, split, nnz, s); break;
    case 2: launch_fast_g<Idx, VEC, 2, MODE>(a, split, nnz, s); break;
    case 4: launch_fast_g<Idx, VEC, 4, MODE>(a, split, nnz, s); break;
    case 8: launch_fast_g<Idx, VEC, 8, MODE>(a, split, nnz, s); break;
    case 16: launch_fast_g<Idx, VEC, 16, MODE>(a, split, nnz, s); break;
    case 32: launch_fast_g<Idx, VEC, 32, MODE>(a, split, nnz, s); break;
    default: launch_fast_g<Idx, VEC, 64, MODE>(a, split, nnz, s); break;
  }
}

template <typename Idx, int MODE>
static void launch_fast(const SpmmFastArgs<Idx>& a, int64_t nnz, hipStream_t s) {
  const bool al16 = ((uintptr_t)a.src % 16 == 0) && ((uintptr_t)a.out % 16 == 0);
  const bool al8 = ((uintptr_t)a.src % 8 == 0) && ((uintptr_t)a.out % 8 == 0);
  // MODE_MUL_EDGE needs every lane's VEC features inside one head: F % VEC == 0.
  const int vec_ok = (MODE == MODE_MUL_EDGE) ? a.F : a.D;
  // odd widths (41 classes): 16-byte gathers with a ragged last lane, lean int32 kernel only, one head
  const bool lean = sizeof(Idx) == 4 && a.src_rows * (int64_t)a.lds * 4 < (int64_t(1) << 32);
  if (lean && MODE != MODE_COPY_RHS && a.D % 4 != 0 && a.D > 4 && (a.D <= 256 || MODE == MODE_COPY_LHS) && a.H == 1 &&
      (uintptr_t)a.src % 4 == 0) {
    SpmmFastArgs<Idx> b = a;
    b.ragged = 1;
    launch_fast_v<Idx, 4, MODE>(b, nnz, s);
    return;
  }
  if (a.D % 4 == 0 && vec_ok % 4 == 0 && al16) launch_fast_v<Idx, 4, MODE>(a, nnz, s);
  else if (a.D % 2 == 0 && vec_ok % 2 == 0 && al8) launch_fast_v<Idx, 2, MODE>(a, nnz, s);
  else launch_fast_v<Idx, 1, MODE>(a, nnz, s);
}

#include "spmm_slots.inc"
#include "spmm_tail.inc"
static bool try_spmm_edge_tail(const SpmmFastArgs<int32_t>& a, const void* tail, hipStream_t s) {
  if (!tail) return false;
  launch_spmm_edge_tail(a, tail, s);
  return true;
}
static bool try_spmm_edge_tail(const SpmmFastArgs<int64_t>&, const void*, hipStream_t) { return false; }
static bool try_spmm_slots(const SpmmFastArgs<int32_t>& a, const void* slots, hipStream_t s) {
  if (!slots || !spmm_slots_eligible(a)) return false;
  // short items whose rows carry no factor keep the dense lane-group kernel (products head: 0.32 - 0.33 ms against 0.37 - 0.38 ms on
  // the slot form: a 3-edge row is a dependent chain either way); with a factor the slots are the only operand that has it applied
  if (a.short_rows && !a.src_scale) return false;
  launch_spmm_slots(a, slots, s);
  return true;
}
static bool try_spmm_slots(const SpmmFastArgs<int64_t>&, const void*, hipStream_t) { return false; }

constexpr int kSpmmPartialPlan = 1 << 30;  // internal flag bit: the plan is one part of a two-part plan and need not cover every row

template <typename Idx>
static int32_t spmm_impl(const mgx_csr* csr, const mgx_spmm_plan* plan, float* partial_ws, int flag_bits, int32_t op, int32_t reduce, const float* U, const float* E,
                         int64_t u_len, int64_t e_len, int64_t out_len, const int64_t* u_off,
                         const int64_t* e_off, const float* src_scale, const float* dst_scale, float* out,
                         void* arg_u, void* arg_e, hipStream_t s, const uint32_t* src_bits = nullptr, int64_t u_stride = 0,
                         int64_t out_stride = 0, const void* slots = nullptr, const void* edge_tail = nullptr) {
  const int64_t n_rows = csr->num_rows;
  const int accumulate = (flag_bits & MGX_SPMM_ACCUMULATE) ? 1 : 0;
  if (n_rows == 0 || out_len == 0) return MGX_OK;
  if (plan && plan->rest) {
    // A plan in two parts (mgx_spmm_plan::rest): the SHORT direct items here, on the lane-group kernel; everything else -- items above
    // 32 edges and the chunks of split rows with their hub tables -- in `rest`, on the wave-per-item kernel.  Rows are independent
    // and every row belongs to exactly one part, so the two calls are the whole operation (accumulate, mean and dst_scale included).
    MGX_CHECK_ARG((flag_bits & MGX_SPMM_SHORT_ROWS) && (reduce == MGX_REDUCE_SUM || reduce == MGX_REDUCE_MEAN) &&
                      (op == MGX_OP_COPY_LHS || op == MGX_OP_COPY_RHS) && !u_off && !e_off && !plan->rest->rest && plan->num_slots == 0,
                  "mgx_spmm_csr: a plan with a `rest` part is for MGX_SPMM_SHORT_ROWS copy_u / copy_e sums (and holds no split rows itself)");
    mgx_spmm_plan head = *plan;
    head.rest = nullptr;
    int32_t st = spmm_impl<Idx>(csr, &head, partial_ws, flag_bits | kSpmmPartialPlan, op, reduce, U, E, u_len, e_len, out_len, u_off, e_off,
                                src_scale, dst_scale, out, arg_u, arg_e, s, src_bits, u_stride, out_stride, slots, edge_tail);
    if (st != MGX_OK) return st;
    const char* head_kernel = mgx_last_spmm_kernel();
    st = spmm_impl<Idx>(csr, plan->rest, partial_ws, (flag_bits & ~MGX_SPMM_SHORT_ROWS) | kSpmmPartialPlan, op, reduce, U, E, u_len, e_len,
                        out_len, u_off, e_off, src_scale, dst_scale, out, arg_u, arg_e, s, src_bits, u_stride, out_stride, slots, edge_tail);
    if (slots && st == MGX_OK) return st;  // (the name of the family that walked `rest` -- "slots" or not -- is what a caller of the packed form asks for)
    note_spmm_kernel(head_kernel);  // the family that walked the short items (the bulk of such a plan)
    return st;
  }
  const int64_t nblocks = round_up((n_rows + kRowsPerBlock - 1) / kRowsPerBlock, kXcds);
  MGX_CHECK_ARG(nblocks < (int64_t(1) << 31), "mgx_spmm_csr: too many rows (%lld)", (long long)n_rows);
  if (plan) {
    if ((flag_bits & kSpmmPartialPlan) && plan->num_items == 0) return MGX_OK;
    MGX_CHECK_ARG((plan->num_items >= n_rows || (flag_bits & kSpmmPartialPlan)) && plan->item_row && plan->item_beg && plan->item_end,
                  "mgx_spmm_csr: malformed plan");
    MGX_CHECK_ARG(plan->num_slots == 0 || (partial_ws && plan->hub_row && plan->hub_slot_ptr),
                  "mgx_spmm_csr: plan has split rows but no partial workspace / hub tables");
    MGX_CHECK_ARG(n_rows < (int64_t(1) << 31), "mgx_spmm_csr: plans need fewer than 2^31 rows");
  }
  const bool summing = reduce == MGX_REDUCE_SUM || reduce == MGX_REDUCE_MEAN;
  const bool no_bcast = !u_off && !e_off;

  // ---- fast paths ------------------------------------------------------------------------
  if (summing && out_len < (int64_t(1) << 30)) {
    SpmmFastArgs<Idx> a;
    a.indptr = (const Idx*)csr->indptr; a.indices = (const Idx*)csr->indices; a.eids = (const Idx*)csr->eids;
    a.w = nullptr; a.src_scale = src_scale; a.dst_scale = dst_scale; a.out = out;
    a.src_bits = src_bits;
    a.n_rows = n_rows; a.nblocks = nblocks; a.D = (int)out_len; a.H = 1; a.F = (int)out_len;
    a.mean = reduce == MGX_REDUCE_MEAN;
    a.accum = accumulate;
    a.short_rows = (flag_bits & MGX_SPMM_SHORT_ROWS) ? ((flag_bits & kSpmmPartialPlan) ? 2 : 1) : 0;
    a.ragged = 0;
    a.lds = u_stride ? (int)u_stride : (int)out_len;
    a.ldo = out_stride ? (int)out_stride : (int)out_len;
    a.item_row = nullptr; a.item_beg = nullptr; a.item_end = nullptr; a.partial = nullptr; a.n_items = n_rows;
    a.plan = plan;
    if (plan) {
      a.item_row = plan->item_row; a.item_beg = (const Idx*)plan->item_beg; a.item_end = (const Idx*)plan->item_end;
      a.partial = partial_ws; a.n_items = plan->num_items;
    }
    MGX_CHECK_ARG(a.n_items / 4 < (int64_t(1) << 31) - 16, "mgx_spmm_csr: too many work items");
    auto fixup = [&]() -> int32_t {
      if (plan && plan->num_hubs > 0) {
        hipLaunchKernelGGL((spmm_hub_fixup_kernel<Idx>), dim3((unsigned)((plan->num_hubs + kWavesPerBlock - 1) / kWavesPerBlock)),
                           dim3(kBlock), 0, s, a.indptr, plan->hub_row, plan->hub_slot_ptr, plan->num_hubs,
                           (const float*)partial_ws, dst_scale, out, a.D, a.mean, a.accum, a.ldo);
        MGX_CHECK_LAUNCH();
      }
      return MGX_OK;
    };
    if (op == MGX_OP_COPY_LHS && no_bcast && u_len == out_len) {
      a.src = U; a.src_rows = csr->num_cols;
      if (a.lds != a.D || a.ldo != a.D) {  // strided rows: the lean row-per-wave kernel with 16-byte lanes only
        const bool ok = sizeof(Idx) == 4 && a.D % 4 == 0 && a.lds % 4 == 0 && a.ldo % 4 == 0 && a.lds >= a.D && a.ldo >= a.D &&
                        (uintptr_t)U % 16 == 0 && (uintptr_t)out % 16 == 0 && a.src_rows * (int64_t)a.lds * 4 < (int64_t(1) << 32) &&
                        !src_bits;
        if (!ok) MGX_UNSUPPORTED("mgx_spmm_copy_u_strided: needs int32 ids, D and both strides multiples of 4, 16-byte aligned "
                                 "pointers and a gathered matrix under 4 GiB");
      }
      if (try_spmm_edge_tail(a, edge_tail, s)) {  // a constant 100-column matrix as [rows, 96] + its last four columns along the edge list
        MGX_CHECK_LAUNCH();
        return fixup();
      }
      if (try_spmm_slots(a, slots, s)) {  // mostly-zero rows as 128-byte slots (spmm_slots.inc): the wave-per-item part only
        MGX_CHECK_LAUNCH();
        return fixup();
      }
      launch_fast<Idx, MODE_COPY_LHS>(a, csr->nnz, s);
      MGX_CHECK_LAUNCH();
      return fixup();
    }
    if (op == MGX_OP_COPY_RHS && no_bcast && e_len == out_len && !src_scale) {
      a.src = E; a.src_rows = csr->nnz;
      launch_fast<Idx, MODE_COPY_RHS>(a, csr->nnz, s);
      MGX_CHECK_LAUNCH();
      return fixup();
    }
    // u_add_e with full-width operands: sum_e (U[u] + E[e]) = (copy_u sum) + (copy_e sum) -- two launches of the fast kernels,
    // the second accumulating into the first's output, instead of the generic one-wave-per-row kernel (128 -> ~25 us on a
    // molhiv-sized batch).  mean and dst_scale distribute over the sum; src_scale belongs to the U term only.
    if (op == MGX_OP_ADD && no_bcast && u_len == out_len && e_len == out_len && U && E) {
      a.src = U; a.src_rows = csr->num_cols;
      launch_fast<Idx, MODE_COPY_LHS>(a, csr->nnz, s);
      MGX_CHECK_LAUNCH();
      int32_t st = fixup();
      if (st != MGX_OK) return st;
      SpmmFastArgs<Idx> b = a;
      b.src = E; b.src_rows = csr->nnz; b.src_scale = nullptr; b.accum = 1;
      launch_fast<Idx, MODE_COPY_RHS>(b, csr->nnz, s);
      MGX_CHECK_LAUNCH();
      if (plan && plan->num_hubs > 0) {
        hipLaunchKernelGGL((spmm_hub_fixup_kernel<Idx>), dim3((unsigned)((plan->num_hubs + kWavesPerBlock - 1) / kWavesPerBlock)),
                           dim3(kBlock), 0, s, a.indptr, plan->hub_row, plan->hub_slot_ptr, plan->num_hubs,
                           (const float*)partial_ws, dst_scale, out, a.D, a.mean, 1, a.ldo);
        MGX_CHECK_LAUNCH();
      }
      return MGX_OK;
    }
    // u_mul_e with one weight per (edge, head): U (N,H,F) x E (E,H,1); also (N,D) x (E,1).
    // ABI rule: a NULL offset table with e_len < out_len means head-wise broadcast, k -> k / (out_len/e_len).
    if (op == MGX_OP_MUL && u_len == out_len && no_bcast && e_len >= 1 && out_len % e_len == 0) {
      a.src = U; a.w = E; a.H = (int)e_len; a.F = (int)(out_len / e_len); a.src_rows = csr->num_cols;
      launch_fast<Idx, MODE_MUL_EDGE>(a, csr->nnz, s);
      MGX_CHECK_LAUNCH();
      return fixup();
    }
  }

  // ---- generic ---------------------------------------------------------------------------
  SpmmGenericArgs<Idx> g;
  g.indptr = (const Idx*)csr->indptr; g.indices = (const Idx*)csr->indices; g.eids = (const Idx*)csr->eids;
  g.U = (op == MGX_OP_COPY_RHS) ? nullptr : U;
  g.E = (op == MGX_OP_COPY_LHS) ? nullptr : E;
  g.u_off = u_off; g.e_off = e_off; g.src_scale = src_scale; g.dst_scale = dst_scale; g.out = out;
  g.arg_u = (Idx*)arg_u; g.arg_e = (Idx*)arg_e; g.n_rows = n_rows; g.nblocks = nblocks;
  g.u_len = u_len; g.e_len = e_len; g.out_len = out_len; g.op = op; g.reduce = reduce; g.accum = accumulate;
  note_spmm_kernel("generic");
  hipLaunchKernelGGL((spmm_generic_kernel<Idx>), dim3((unsigned)nblocks), dim3(kBlock), 0, s, g);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

// bits[r / 32] bit (r % 32) = row r of x [n, D] has a non-zero element.  Lanes along the row (16-byte loads), 64 / G rows
// per wave-instruction; a wave owns 64 consecutive rows = two output words.
template <int G>
__global__ __launch_bounds__(kBlock) void row_nonzero_bits_kernel(int64_t n, int D, const float* __restrict__ x, uint32_t* __restrict__ bits) {
  constexpr int NB = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int sub = lane / G, l = lane % G;
  const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int64_t r0 = w * kWave;
  if (r0 >= n) return;
  uint64_t word = 0;
  for (int k = 0; k < kWave; k += NB) {
    const int64_t r = r0 + k + sub;
    bool nz = false;
    if (r < n) {
      for (int f = l * 4; f < D; f += G * 4) {
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(x + r * D + f));
        nz = nz || v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f;
      }
    }
    const uint64_t b = __ballot(nz);
#pragma unroll
    for (int g = 0; g < NB; ++g) {
      const uint64_t gm = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << (g * G);
      if (b & gm) word |= 1ull << (k + g);
    }
  }
  if (lane == 0) {
    bits[2 * w] = (uint32_t)word;
    if (r0 + 32 < n) bits[2 * w + 1] = (uint32_t)(word >> 32);
  }
}

}  // namespace mgx

extern "C" int32_t mgx_row_nonzero_bits(int64_t n, int64_t D, const float* x, uint32_t* bits, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && D >= 4 && D % 4 == 0, "mgx_row_nonzero_bits: D must be a positive multiple of 4 (got %lld)", (long long)D);
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(x && bits && (uintptr_t)x % 16 == 0, "mgx_row_nonzero_bits: NULL or unaligned pointer");
  const dim3 grid((unsigned)((n + (int64_t)kWave * kWavesPerBlock - 1) / ((int64_t)kWave * kWavesPerBlock))), block(kBlock);
  hipStream_t s = (hipStream_t)stream;
  int G = 1;
  while (G * 4 < D && G < kWave) G <<= 1;
  switch (G) {
    case 1: hipLaunchKernelGGL((row_nonzero_bits_kernel<1>), grid, block, 0, s, n, (int)D, x, bits); break;
    case 2: hipLaunchKernelGGL((row_nonzero_bits_kernel<2>), grid, block, 0, s, n, (int)D, x, bits); break;
    case 4: hipLaunchKernelGGL((row_nonzero_bits_kernel<4>), grid, block, 0, s, n, (int)D, x, bits); break;
    case 8: hipLaunchKernelGGL((row_nonzero_bits_kernel<8>), grid, block, 0, s, n, (int)D, x, bits); break;
    case 16: hipLaunchKernelGGL((row_nonzero_bits_kernel<16>), grid, block, 0, s, n, (int)D, x, bits); break;
    case 32: hipLaunchKernelGGL((row_nonzero_bits_kernel<32>), grid, block, 0, s, n, (int)D, x, bits); break;
    default: hipLaunchKernelGGL((row_nonzero_bits_kernel<64>), grid, block, 0, s, n, (int)D, x, bits); break;
  }
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int32_t mgx_spmm_copy_u_masked(const mgx_csr* csr, const mgx_spmm_plan* plan, int32_t reduce, const float* ufeat,
                                          int64_t D, const uint32_t* src_bits, const float* dst_scale, float* out,
                                          float* partial_ws, int32_t flags, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(csr != nullptr && src_bits != nullptr, "mgx_spmm_copy_u_masked: NULL pointer");
  MGX_CHECK_ARG(reduce == MGX_REDUCE_SUM || reduce == MGX_REDUCE_MEAN, "mgx_spmm_copy_u_masked: SUM or MEAN only");
  if (csr->idx_bits != 32 || D % 4 != 0 || csr->num_cols * D * 4 >= (int64_t(1) << 32) || (uintptr_t)ufeat % 16 || (uintptr_t)out % 16)
    MGX_UNSUPPORTED("mgx_spmm_copy_u_masked: needs 32-bit indices, D %% 4 == 0, 16-byte aligned operands below 4 GiB");
  MGX_CHECK_ARG(csr->num_rows == 0 || csr->indptr, "mgx_spmm_copy_u_masked: indptr is NULL");
  MGX_CHECK_ARG(csr->nnz == 0 || (csr->indices && ufeat), "mgx_spmm_copy_u_masked: indices / ufeat is NULL");
  MGX_CHECK_ARG(out != nullptr || csr->num_rows == 0 || D == 0, "mgx_spmm_copy_u_masked: out is NULL");
  return spmm_impl<int32_t>(csr, plan, partial_ws, flags & MGX_SPMM_ACCUMULATE, MGX_OP_COPY_LHS, reduce, ufeat, nullptr, D, 0, D,
                            nullptr, nullptr, nullptr, dst_scale, out, nullptr, nullptr, (hipStream_t)stream, src_bits);
}

extern "C" int32_t mgx_spmm_copy_u_strided(const mgx_csr* csr, const mgx_spmm_plan* plan, int32_t reduce, const float* ufeat,
                                           int64_t D, int64_t u_stride, const float* dst_scale, float* out, int64_t out_stride,
                                           float* partial_ws, int32_t flags, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(csr != nullptr, "mgx_spmm_copy_u_strided: csr is NULL");
  MGX_CHECK_ARG(csr->num_rows >= 0 && csr->nnz >= 0, "mgx_spmm_copy_u_strided: negative sizes");
  MGX_CHECK_ARG(csr->num_rows == 0 || csr->indptr != nullptr, "mgx_spmm_copy_u_strided: indptr is NULL");
  MGX_CHECK_ARG(csr->nnz == 0 || csr->indices != nullptr, "mgx_spmm_copy_u_strided: indices is NULL");
  MGX_CHECK_ARG(reduce == MGX_REDUCE_SUM || reduce == MGX_REDUCE_MEAN, "mgx_spmm_copy_u_strided: SUM or MEAN only, got %d", reduce);
  MGX_CHECK_ARG(D >= 0 && u_stride >= D && out_stride >= D, "mgx_spmm_copy_u_strided: strides must be >= D");
  MGX_CHECK_ARG((ufeat != nullptr || csr->num_cols == 0) && (out != nullptr || csr->num_rows == 0 || D == 0),
                "mgx_spmm_copy_u_strided: ufeat / out is NULL");
  if (csr->idx_bits != 32) MGX_UNSUPPORTED("mgx_spmm_copy_u_strided: int32 graphs only");
  return spmm_impl<int32_t>(csr, plan, partial_ws, flags & (MGX_SPMM_ACCUMULATE | MGX_SPMM_SHORT_ROWS), MGX_OP_COPY_LHS, reduce, ufeat, nullptr, D, 0,
                            D, nullptr, nullptr, nullptr, dst_scale, out, nullptr, nullptr, (hipStream_t)stream, nullptr, u_stride,
                            out_stride);
}

extern "C" int32_t mgx_edge_tail_fill(const mgx_csr* csr, const float* x, int64_t x_stride, float* edge_tail, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(csr != nullptr && csr->nnz >= 0 && x_stride >= 100, "mgx_edge_tail_fill: csr is NULL, or a row stride below 100");
  if (csr->idx_bits != 32 || (uintptr_t)edge_tail % 16 != 0) MGX_UNSUPPORTED("mgx_edge_tail_fill: int32 graphs, a 16-byte aligned tail");
  if (csr->nnz == 0) return MGX_OK;
  MGX_CHECK_ARG(csr->indices && x && edge_tail, "mgx_edge_tail_fill: NULL pointer");
  int64_t blocks = (csr->nnz + kBlock - 1) / kBlock;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(edge_tail_fill_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, csr->nnz, (const int32_t*)csr->indices, x,
                     x_stride, (v4f*)edge_tail);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int32_t mgx_spmm_copy_u_edge_tail(const mgx_csr* csr, const mgx_spmm_plan* plan, int32_t reduce, const float* block_a,
                                             const float* edge_tail, const float* dst_scale, float* out, int64_t out_stride, float* partial_ws,
                                             int32_t flags, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(csr != nullptr, "mgx_spmm_copy_u_edge_tail: csr is NULL");
  MGX_CHECK_ARG(csr->num_rows >= 0 && csr->nnz >= 0, "mgx_spmm_copy_u_edge_tail: negative sizes");
  MGX_CHECK_ARG(csr->num_rows == 0 || csr->indptr != nullptr, "mgx_spmm_copy_u_edge_tail: indptr is NULL");
  MGX_CHECK_ARG(csr->nnz == 0 || (csr->indices != nullptr && block_a != nullptr && edge_tail != nullptr), "mgx_spmm_copy_u_edge_tail: NULL pointer");
  MGX_CHECK_ARG(reduce == MGX_REDUCE_SUM || reduce == MGX_REDUCE_MEAN, "mgx_spmm_copy_u_edge_tail: SUM or MEAN only, got %d", reduce);
  MGX_CHECK_ARG(out_stride >= 100 && (out != nullptr || csr->num_rows == 0), "mgx_spmm_copy_u_edge_tail: out is NULL or its stride below 100");
  if (csr->idx_bits != 32 || out_stride % 4 != 0 || (uintptr_t)block_a % 128 != 0 || (uintptr_t)edge_tail % 16 != 0 || (uintptr_t)out % 16 != 0 ||
      csr->num_cols * (int64_t)384 >= (int64_t(1) << 32) || (flags & MGX_SPMM_SHORT_ROWS) || (plan && plan->rest))
    MGX_UNSUPPORTED("mgx_spmm_copy_u_edge_tail: int32 graphs, block_a 128-byte aligned and below 4 GiB, 16-byte aligned tail / output, one schedule");
  // (u_stride = 100: the logical width; the kernel addresses block_a by its own 96-column stride)
  return spmm_impl<int32_t>(csr, plan, partial_ws, flags & MGX_SPMM_ACCUMULATE, MGX_OP_COPY_LHS, reduce, block_a, nullptr, 100, 0, 100, nullptr,
                            nullptr, nullptr, dst_scale, out, nullptr, nullptr, (hipStream_t)stream, nullptr, 100, out_stride, nullptr, edge_tail);
}

extern "C" int32_t mgx_rows_slots_pack(int64_t n, int64_t D, const float* x, int64_t x_stride, const float* row_scale, void* slots,
                                       int64_t* overflow_rows, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(n >= 0 && x_stride >= D, "mgx_rows_slots_pack: negative size or stride below D");
  if (D != 64 || x_stride % 4 != 0 || (uintptr_t)x % 16 != 0 || (uintptr_t)slots % 16 != 0)
    MGX_UNSUPPORTED("mgx_rows_slots_pack: rows of exactly 64 columns, stride a multiple of 4, 16-byte aligned pointers");
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(x && slots, "mgx_rows_slots_pack: NULL pointer");
  int64_t blocks = (n + 4 * kWavesPerBlock - 1) / (4 * kWavesPerBlock);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rows_slots_pack_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, n, x, x_stride, row_scale,
                     (uint32_t*)slots, (unsigned long long*)overflow_rows);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int32_t mgx_spmm_copy_u_slots(const mgx_csr* csr, const mgx_spmm_plan* plan, int32_t reduce, const float* ufeat, int64_t D,
                                         int64_t u_stride, const void* slots, const float* src_scale, const float* dst_scale, float* out,
                                         int64_t out_stride, float* partial_ws, int32_t flags, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(csr != nullptr, "mgx_spmm_copy_u_slots: csr is NULL");
  MGX_CHECK_ARG(csr->num_rows >= 0 && csr->nnz >= 0, "mgx_spmm_copy_u_slots: negative sizes");
  MGX_CHECK_ARG(csr->num_rows == 0 || csr->indptr != nullptr, "mgx_spmm_copy_u_slots: indptr is NULL");
  MGX_CHECK_ARG(csr->nnz == 0 || csr->indices != nullptr, "mgx_spmm_copy_u_slots: indices is NULL");
  MGX_CHECK_ARG(reduce == MGX_REDUCE_SUM || reduce == MGX_REDUCE_MEAN, "mgx_spmm_copy_u_slots: SUM or MEAN only, got %d", reduce);
  MGX_CHECK_ARG(u_stride >= D && out_stride >= D, "mgx_spmm_copy_u_slots: strides must be >= D");
  MGX_CHECK_ARG((ufeat != nullptr && slots != nullptr) || csr->num_cols == 0, "mgx_spmm_copy_u_slots: ufeat / slots is NULL");
  MGX_CHECK_ARG(out != nullptr || csr->num_rows == 0, "mgx_spmm_copy_u_slots: out is NULL");
  if (csr->idx_bits != 32 || D != 64 || u_stride % 4 != 0 || out_stride % 4 != 0 || (uintptr_t)ufeat % 16 != 0 || (uintptr_t)out % 16 != 0 ||
      (uintptr_t)slots % 16 != 0 || csr->num_cols * (int64_t)128 >= (int64_t(1) << 32) || csr->num_cols * u_stride * 4 >= (int64_t(1) << 32))
    MGX_UNSUPPORTED("mgx_spmm_copy_u_slots: int32 graphs, 64 columns, strides multiples of 4, 16-byte aligned operands below 4 GiB");
  // u_stride / out_stride equal to D are passed as such: spmm_impl treats 0 as "contiguous"
  return spmm_impl<int32_t>(csr, plan, partial_ws, flags & (MGX_SPMM_ACCUMULATE | MGX_SPMM_SHORT_ROWS), MGX_OP_COPY_LHS, reduce, ufeat, nullptr, D, 0,
                            D, nullptr, nullptr, src_scale, dst_scale, out, nullptr, nullptr, (hipStream_t)stream, nullptr, u_stride,
                            out_stride, slots);
}

extern "C" int32_t mgx_spmm_csr(const mgx_csr* csr, const mgx_spmm_plan* plan, int32_t op, int32_t reduce,
                                const float* ufeat, const float* efeat, int64_t u_len, int64_t e_len,
                                int64_t out_len, const int64_t* u_off, const int64_t* e_off, const float* src_scale,
                                const float* dst_scale, float* out, void* arg_u, void* arg_e, float* partial_ws,
                                int32_t flags, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(csr != nullptr, "mgx_spmm_csr: csr is NULL");
  MGX_CHECK_ARG(csr->idx_bits == 32 || csr->idx_bits == 64, "mgx_spmm_csr: idx_bits must be 32 or 64, got %d", csr->idx_bits);
  MGX_CHECK_ARG(csr->num_rows >= 0 && csr->nnz >= 0, "mgx_spmm_csr: negative sizes");
  MGX_CHECK_ARG(csr->num_rows == 0 || csr->indptr != nullptr, "mgx_spmm_csr: indptr is NULL");
  MGX_CHECK_ARG(csr->nnz == 0 || csr->indices != nullptr || op == MGX_OP_COPY_RHS, "mgx_spmm_csr: indices is NULL");
  MGX_CHECK_ARG(op == MGX_OP_ADD || op == MGX_OP_MUL || op == MGX_OP_COPY_LHS || op == MGX_OP_COPY_RHS ||
                    op == MGX_OP_SUB || op == MGX_OP_DIV,
                "mgx_spmm_csr: unsupported binary op %d", op);
  MGX_CHECK_ARG(reduce >= MGX_REDUCE_SUM && reduce <= MGX_REDUCE_MEAN, "mgx_spmm_csr: unsupported reduce op %d", reduce);
  MGX_CHECK_ARG(op == MGX_OP_COPY_RHS || ufeat != nullptr || csr->num_cols == 0, "mgx_spmm_csr: op needs ufeat");
  MGX_CHECK_ARG(op == MGX_OP_COPY_LHS || efeat != nullptr || csr->nnz == 0, "mgx_spmm_csr: op needs efeat");
  MGX_CHECK_ARG(out != nullptr || csr->num_rows == 0 || out_len == 0, "mgx_spmm_csr: out is NULL");
  MGX_CHECK_ARG(out_len >= 0 && u_len >= 0 && e_len >= 0, "mgx_spmm_csr: negative feature length");
  const bool cmp = reduce == MGX_REDUCE_MAX || reduce == MGX_REDUCE_MIN;
  MGX_CHECK_ARG(!cmp || (!src_scale && !dst_scale), "mgx_spmm_csr: src/dst scale only with SUM/MEAN");
  const int accumulate = (flags & MGX_SPMM_ACCUMULATE) ? 1 : 0;
  MGX_CHECK_ARG(!accumulate || !cmp, "mgx_spmm_csr: MGX_SPMM_ACCUMULATE only with SUM/MEAN");
  hipStream_t s = (hipStream_t)stream;
  if (csr->idx_bits == 32)
    return spmm_impl<int32_t>(csr, plan, partial_ws, flags & (MGX_SPMM_ACCUMULATE | MGX_SPMM_SHORT_ROWS), op, reduce, ufeat, efeat, u_len, e_len, out_len, u_off, e_off, src_scale,
                              dst_scale, out, arg_u, arg_e, s);
  return spmm_impl<int64_t>(csr, plan, partial_ws, flags & MGX_SPMM_ACCUMULATE, op, reduce, ufeat, efeat, u_len, e_len, out_len, u_off, e_off, src_scale,
                            dst_scale, out, arg_u, arg_e, s);
}
