// gatfused.hip -- one GAT layer's message passing without any E-sized tensor (gfx950 / MI355X).
//
// Replaces what dgl.nn.pytorch.GATConv.forward runs between the projection and the bias
// (main_dgl_reddit_gat.py:10,31-55; UPSTREAM module): apply_edges(fn.u_add_v) -> leaky_relu -> edge_softmax ->
// attn_drop -> update_all(fn.u_mul_e, fn.sum), and its backward.  DGL 0.6 launches ~7 kernels forward and ~12 backward
// and writes / re-reads six E x H tensors; round 1 of this library fused the logit chain but still wrote the attention
// `a` (E x H), its gradient and the logit gradient.  Here the attention weight of an edge is REBUILT in registers
// wherever it is needed from four per-node numbers (er, row max m, 1 / row sum, t = <out, d out>):
//
//   forward   gat_fused_kernel<FWD>     ONE walk of the in-CSR with an online softmax: every lane keeps a running maximum m and
//                                       sum s of exp(z - m), z = leaky_relu(el[u] + er[v]), and rescales its accumulator when m
//                                       grows; out[v,h,:] = sum_e keep(e,h)/(1-p) exp(z - m) feat[u,h,:] / s,
//                                       nstat[v,h] = (er, m, 1/s, .) is written for the backward
//   backward  gat_fused_kernel<BWD_DST> walks the in-CSR again (gathers feat[u]): d a = <feat[u,h,:], d out[v,h,:]>,
//                                       d z = a (d a - t) leaky_relu'(.), d er[v,h] = sum_e d z; also writes t[v,h]
//             gat_fused_kernel<BWD_SRC> walks the out-CSR (gathers d out[v] and nstat[v]): d feat[u,h,:] = sum_e a' d out[v,h,:]
//                                       and d el[u,h] = sum_e d z
// using  sum_e a' d a = <out[v,h,:], d out[v,h,:]>  (out IS that weighted sum), so the softmax backward needs no pass
// of its own.  attn_drop is a counter-based mask keyed by (seed, edge id, head): every kernel regenerates the same bit (gat_hash).
//
// All three gather kernels are laid out like the summing g-SpMM (spmm.hip): one wave per work item of the
// mgx_spmm_plan, lanes ALONG the H*F feature row with 16-byte loads, 64/G neighbour rows per wave-instruction, ids
// handed out by ds_bpermute as 32-bit byte offsets; hub rows are split by the plan, their partial sums are combined in
// slot order by gat_rows_fixup_kernel (no atomics: deterministic).  HBM/gather-bound fp32 work, no MFMA.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "tile_common.h"

namespace mgx {

// attn_drop: one bit per (seed, key): key = (edge id, head) in the row kernels, (source, destination) in the tile walks.  A 32-bit
// multiply-xorshift mixer (lowbias32's constants; the second key word and the seed's high half enter between its two rounds):
// two 32-bit multiplies per bit -- the splitmix64 finaliser used before costs two 64-bit multiplies, eight quarter-rate 32-bit ones
// on CDNA, per edge and lane (reddit GAT walks with attn_drop: 1.08 / 0.97 / 1.02 -> see profiles/r03_gat_reddit_kernel_stats.txt).
__device__ __forceinline__ uint32_t gat_hash(uint64_t seed, uint32_t klo, uint32_t khi) {
  uint32_t x = klo ^ (uint32_t)seed;
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x ^= khi + (uint32_t)(seed >> 32);
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// Sum over the LPH lanes of a head (LPH a power of two, the lanes aligned to LPH): the xor butterfly of __shfl_xor, with its first
// four steps as DPP moves instead of ds_bpermute -- swap neighbours, swap pairs (quad_perm), then mirror a half row and a row
// (every lane of a quad / half row already holds the same partial sum, so the mirrored partner's value IS the xor partner's).
// Bitwise the same sums; no LDS-pipe instruction for heads of up to 16 lanes.
template <int CTRL>
__device__ __forceinline__ float gat_dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int LPH>
__device__ __forceinline__ float head_sum(float x) {
  if (LPH >= 2) x += gat_dpp<0xB1>(x);    // quad_perm [1, 0, 3, 2]
  if (LPH >= 4) x += gat_dpp<0x4E>(x);    // quad_perm [2, 3, 0, 1]
  if (LPH >= 8) x += gat_dpp<0x141>(x);   // row_half_mirror
  if (LPH >= 16) x += gat_dpp<0x140>(x);  // row_mirror
#pragma unroll
  for (int off = 16; off < LPH; off <<= 1) x += __shfl_xor(x, off, kWave);
  return x;
}

struct GatArgs {
  const int32_t* indptr;
  const int32_t* indices;
  const int32_t* eids;      // CSR position -> edge id (NULL: identity); keys the dropout mask
  const int32_t* item_row;  // plan (all NULL: one item per row)
  const int32_t* item_beg;
  const int32_t* item_end;
  const int32_t* item_node;
  int64_t n_items;
  XcdRanges xcd;    // item stretch of every XCD (edge balanced when the plan says so)
  int64_t nblocks;
  int rpb;
  int H, F, D;
  float slope;
  uint32_t drop_below;  // keep when (uint32)hash >= drop_below
  float keep_scale;     // 1 / (1 - p)
  uint64_t seed;
  const float* gat;     // gathered matrix [*, D]: feat (FWD, BWD_DST) or d out (BWD_SRC)
  const float* el;      // [num_src, H]
  const float* attn;    // ELK kernels: attn_l [H, F] -- el[u,h] = <gat[u,h,:], attn[h,:]> is formed from the gathered row instead
  const float* er;      // [num_dst, H]  (stats kernel)
  const float* nstat;   // [num_dst, H, 4] = (er, m, 1/s, t)
  float* nstat_w;
  const float* rowa;    // row-constant operand: d out[v] (BWD_DST), feat[u] (BWD_SRC)
  const float* rowb;    // BWD_DST: out[v]
  float* out;           // FWD: out; BWD_SRC: d feat
  float* out_h;         // BWD_DST: d er; BWD_SRC: d el   [rows, H]
  float* partial;       // [slots, D]
  float* partial_h;     // [slots, H] (stats: [slots, 2H])
  int gat_ld;           // floats between consecutive rows of `gat` (D unless packed)
  int small_ld;         // floats between consecutive nodes of the small gathered array (el: H, nstat: 4H unless packed)
};

// ---------------------------------------------------------------------------------------------- hub rows of the forward
// A hub row's chunks each leave (running max m_c, sum s_c) per head in partial_h [slot, 2H] and an accumulator relative to
// m_c in partial [slot, D]; merged here in slot order: m = max m_c, s = sum s_c e^(m_c - m), out = sum acc_c e^(m_c - m) / s.
__global__ __launch_bounds__(kBlock) void gat_online_fixup_kernel(const int32_t* hub_row, const int32_t* hub_slot_ptr, int64_t n_hubs,
                                                                  int H, int F, const float* er, const float* partial,
                                                                  const float* partial_h, float* out, float* nstat) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t hb = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  if (hb >= n_hubs) return;
  const int64_t row = hub_row[hb];
  const int s0 = hub_slot_ptr[hb], s1 = hub_slot_ptr[hb + 1];
  const int D = H * F;
  for (int k = lane; k < D; k += kWave) {
    const int h = k / F;
    float m = -INFINITY;
    for (int s = s0; s < s1; ++s) m = fmaxf(m, partial_h[(int64_t)s * 2 * H + h]);
    float sum = 0.f, acc = 0.f;
    for (int s = s0; s < s1; ++s) {
      const float mc = partial_h[(int64_t)s * 2 * H + h];
      if (mc > -INFINITY) {
        const float fct = __expf(mc - m);
        sum += partial_h[(int64_t)s * 2 * H + H + h] * fct;
        acc += partial[(int64_t)s * D + k] * fct;
      }
    }
    const float is = sum > 0.f ? 1.f / sum : 0.f;
    out[row * D + k] = acc * is;
    if (k % F == 0) {
      v4f st;
      st.x = er[row * H + h];
      st.y = m > -INFINITY ? m : 0.f;
      st.z = is;
      st.w = 0.f;
      *reinterpret_cast<v4f*>(nstat + (row * H + h) * 4) = st;
    }
  }
}

// out[hub_row[h], k] = sum over the hub's slots, in slot order, of partial[slot, k]
__global__ __launch_bounds__(kBlock) void gat_rows_fixup_kernel(const int32_t* hub_row, const int32_t* hub_slot_ptr, int64_t n_hubs,
                                                                int L, const float* partial, float* out) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t h = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  if (h >= n_hubs) return;
  const int64_t row = hub_row[h];
  const int s0 = hub_slot_ptr[h], s1 = hub_slot_ptr[h + 1];
  for (int k = lane; k < L; k += kWave) {
    float acc = 0.f;
    for (int s = s0; s < s1; ++s) acc += partial[(int64_t)s * L + k];
    out[row * L + k] = acc;
  }
}

// ---------------------------------------------------------------------------------------------- the gather kernels
enum { GAT_FWD = 0, GAT_BWD_DST = 1, GAT_BWD_SRC = 2 };

template <int G>
struct GatUnroll {
  static constexpr int NB = kWave / G;
  static constexpr int value = NB >= 16 ? 1 : (NB >= 8 ? 2 : 4);
};

// G lanes cover one H*F row (16 bytes each); LPH = F / 4 lanes share a head.
// RAGGED (one head, F % 4 != 0, e.g. the 41-class output layer of main_dgl_reddit_gat.py): rows are still moved 16 bytes per
// lane with dword-aligned accesses; the lane that owns the last 1-3 columns works on the row's LAST four floats instead --
// nothing is read past a row -- with the components that belong to its neighbour zeroed in the row-constant operand (dot
// products) and left out of the store (the same windows as spmm.hip / sddmm.hip use).
// ELK ("el in kernel", several heads per row, unpacked operands: the 8 x 16 layers of BASELINE config 3): the attention term
// el[u,h] = <feat[u,h,:], attn_l[h,:]> of a gathered source is formed from the row the lane group already holds -- 4 multiply-adds
// and log2(LPH) lane swaps per edge instead of a fifth L2 request per edge.  The source walk (BWD_SRC) forms its own el[u] from
// the row-constant feat[u] the same way, so every walk rebuilds the same logit bit for bit.
template <int G, int LPH, int MODE, bool DROP, bool RAGGED = false, bool ELK = false>
__global__ __launch_bounds__(kBlock) void gat_fused_kernel(const GatArgs a) {
  typedef v4f v4u __attribute__((aligned(4)));
  typedef typename std::conditional<RAGGED, v4u, v4f>::type V4;
  constexpr int NB = kWave / G;
  constexpr int U = GatUnroll<G>::value;
  constexpr int STEP = NB * U;
  // heads of >= 4 lanes, four edges per batch: 8 x 16, 4 x 16, 4 x 64, the 41-column layer (whose destination walk measured slower this way)
  constexpr bool QUAD = LPH >= 4 && U == 4 && (LPH == 4 || MODE != GAT_BWD_DST);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int sub = lane / G, l = lane % G;
  const int D = a.D, H = a.H;
  const int f = l * 4;
  const bool fact = f < D;
  const int head = fact ? l / LPH : 0;
  const uint32_t rowbytes = (uint32_t)a.gat_ld * 4u;
  const int nvalid = RAGGED ? (D - f < 4 ? D - f : 4) : 4;   // columns this lane owns (RAGGED: the last active lane < 4)
  const bool tail = RAGGED && fact && nvalid < 4;
  const int fw = tail ? D - 4 : f;                            // first column of this lane's 16-byte window
  const uint32_t f4 = fact ? (uint32_t)fw * 4u : 0u;  // idle feature lanes re-read the row start, never stored
  auto clip = [&](v4f v) -> v4f {  // the window components that belong to the previous lane count for nothing
    if (tail) {
      if (nvalid < 4) v.x = 0.f;
      if (nvalid < 3) v.y = 0.f;
      if (nvalid < 2) v.z = 0.f;
    }
    return v;
  };
  v4f al = (v4f)(0.f);
  if (ELK && fact) al = *reinterpret_cast<const v4f*>(a.attn + f);  // attn_l[head, (l % LPH) * 4 ..]: flat index = this lane's column
  auto head_dot = [&](const v4f& v) -> float {  // <row[h,:], attn_l[h,:]> over the LPH lanes of this lane's head
    return head_sum<LPH>(v.x * al.x + v.y * al.y + v.z * al.z + v.w * al.w);
  };
  const uint32_t hbytes = (uint32_t)a.small_ld * 4u;  // per-node stride of the small per-head array (el: H floats, nstat: 4H; packed: the row stride)
  const uint32_t h4 = (uint32_t)head * (MODE == GAT_BWD_SRC ? 16u : 4u);
  const char* __restrict__ gatb = reinterpret_cast<const char*>(a.gat);
  const char* __restrict__ smallb = reinterpret_cast<const char*>(MODE == GAT_BWD_SRC ? a.nstat : a.el);
  int64_t item_base, item_stop;
  xcd_stretch(a.xcd, item_base, item_stop);
  item_base += (int64_t)(blockIdx.x / kXcds) * a.rpb;

  for (int r = wave; r < a.rpb; r += kWavesPerBlock) {
    const int64_t item = item_base + r;
    if (item >= item_stop) break;
    int64_t row, irow;
    int32_t beg, end;
    if (a.item_row) {
      irow = a.item_row[item];
      row = a.item_node[item];
      beg = a.item_beg[item];
      end = a.item_end[item];
    } else {
      irow = row = item;
      beg = a.indptr[item];
      end = a.indptr[item + 1];
    }
    // ---- row constants
    v4f ra = (v4f)(0.f);
    float c_er = 0.f, c_m = 0.f, c_is = 0.f, c_t = 0.f, c_el = 0.f;
    // (QUAD: an idle feature lane of a ragged row shares its quad with active lanes and does the scalar work of one of the quad's
    // four edges, so it needs the row's constants too -- head 0's, a valid address)
    if (MODE == GAT_BWD_DST) {
      if (fact || QUAD) {
        const v4f st = *reinterpret_cast<const v4f*>(a.nstat + (row * H + head) * 4);
        c_er = st.x; c_m = st.y; c_is = st.z;
      }
    }
    if (MODE == GAT_FWD && (fact || QUAD)) c_er = a.er[row * H + head];
    float run_m = -INFINITY, run_s = 0.f;  // FWD: online softmax state of this lane's head (per lane group)
    if (MODE == GAT_BWD_DST) {
      v4f ov = (v4f)(0.f);
      if (fact) {
        ra = clip((v4f)*reinterpret_cast<const V4*>(a.rowa + row * D + fw));  // d out[v]
        ov = (v4f)*reinterpret_cast<const V4*>(a.rowb + row * D + fw);        // out[v]
      }
      const float t = head_sum<LPH>(ra.x * ov.x + ra.y * ov.y + ra.z * ov.z + ra.w * ov.w);  // idle lanes take part with zeros
      c_t = t;
      // every chunk of a hub row writes the same value
      if (fact && (l % LPH) == 0 && sub == 0) a.nstat_w[(row * H + head) * 4 + 3] = t;
    }
    if (MODE == GAT_BWD_SRC && fact) ra = clip((v4f)*reinterpret_cast<const V4*>(a.rowa + row * D + fw));  // feat[u]
    if (MODE == GAT_BWD_SRC && !ELK && (fact || QUAD)) c_el = a.el[row * H + head];
    if (MODE == GAT_BWD_SRC && ELK) c_el = head_dot(ra);  // every lane takes part in the swaps (idle lanes hold zeros)
    v4f acc = (v4f)(0.f);
    float hacc = 0.f;  // d er (BWD_DST) / d el (BWD_SRC) of this lane's head

    for (int32_t cbase = beg; cbase < end; cbase += kWave) {
      const int32_t q = cbase + lane;
      uint32_t goff = 0, soff = 0;
      int32_t eid = 0;
      if (q < end) {
        const int32_t gid = __builtin_nontemporal_load(&a.indices[q]);
        goff = (uint32_t)gid * rowbytes;
        soff = (uint32_t)gid * hbytes;
        if (DROP) eid = a.eids ? __builtin_nontemporal_load(&a.eids[q]) : q;
      }
      const int cnt = (end - cbase) < kWave ? (end - cbase) : kWave;
      for (int k = 0; k < cnt; k += STEP) {
        if (QUAD) {
          // Heads of four or more lanes with four edges in flight per lane group: lane q of every DPP quad does the SCALAR work of
          // edge q alone -- its logit, its exp, its mask bit, the gather of its record -- instead of every lane doing all four
          // (the quads of a wider head each do the same four); maxima / sums travel by quad_perm swaps, the four weights by
          // quad_perm broadcasts into fused multiply-adds (as gat_tile.inc).
          const int qv = l & 3;
          v4f val[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int jdx = k + u * NB + sub;
            const int bi = (jdx < cnt ? jdx : 0) * 4;
            val[u] = (v4f)*reinterpret_cast<const V4*>(gatb + (uint32_t)__builtin_amdgcn_ds_bpermute(bi, (int)goff) + f4);
          }
          const int jown = k + qv * NB + sub;
          const bool live = jown < cnt;
          const int bown = (live ? jown : 0) * 4;
          v4f smo = (v4f)(0.f);
          if (MODE == GAT_BWD_SRC) smo = *reinterpret_cast<const v4f*>(smallb + (uint32_t)__builtin_amdgcn_ds_bpermute(bown, (int)soff) + h4);
          else if (!ELK) smo.x = *reinterpret_cast<const float*>(smallb + (uint32_t)__builtin_amdgcn_ds_bpermute(bown, (int)soff) + h4);
          float keep = 1.f;
          if (DROP) keep = gat_hash(a.seed, (uint32_t)__builtin_amdgcn_ds_bpermute(bown, eid), (uint32_t)head) >= a.drop_below ? a.keep_scale : 0.f;
          if (ELK && MODE != GAT_BWD_SRC) {
            float e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) e[u] = head_dot(val[u]);
            smo.x = qv == 0 ? e[0] : (qv == 1 ? e[1] : (qv == 2 ? e[2] : e[3]));
          }
          const float t = MODE == GAT_BWD_SRC ? c_el + smo.x : smo.x + c_er;
          const float z = t > 0.f ? t : t * a.slope;
          if (MODE == GAT_FWD) {
            float m4 = live ? z : -INFINITY;
            m4 = fmaxf(m4, gat_dpp<0xB1>(m4));
            m4 = fmaxf(m4, gat_dpp<0x4E>(m4));
            const float mn = fmaxf(run_m, m4);
            const float rescale = run_m > -INFINITY ? __expf(run_m - mn) : 0.f;
            const float pe = live ? __expf(z - mn) : 0.f;
            const float pw = pe * keep;
            v4f pa = val[0] * gat_dpp<0x00>(pw);
            pa = __builtin_elementwise_fma(val[1], (v4f)(gat_dpp<0x55>(pw)), pa);
            pa = __builtin_elementwise_fma(val[2], (v4f)(gat_dpp<0xAA>(pw)), pa);
            pa = __builtin_elementwise_fma(val[3], (v4f)(gat_dpp<0xFF>(pw)), pa);
            run_s = __builtin_fmaf(run_s, rescale, head_sum<4>(pe));
            acc = __builtin_elementwise_fma(acc, (v4f)(rescale), pa);
            run_m = mn;
          } else {
            const float mm = MODE == GAT_BWD_DST ? c_m : smo.y;
            const float is = MODE == GAT_BWD_DST ? c_is : smo.z;
            const float tt = MODE == GAT_BWD_DST ? c_t : smo.w;
            const float av = live ? __expf(z - mm) * is : 0.f;
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
              d[u] = head_sum<LPH>(__builtin_fmaf(val[u].w, ra.w, __builtin_fmaf(val[u].z, ra.z, __builtin_fmaf(val[u].y, ra.y, val[u].x * ra.x))));
            const float dot = qv == 0 ? d[0] : (qv == 1 ? d[1] : (qv == 2 ? d[2] : d[3]));
            hacc += av * (dot * keep - tt) * (t > 0.f ? 1.f : a.slope);  // this lane's edge only: summed over the quad below
            if (MODE == GAT_BWD_SRC) {
              const float w = av * keep;
              acc = __builtin_elementwise_fma(val[0], (v4f)(gat_dpp<0x00>(w)), acc);
              acc = __builtin_elementwise_fma(val[1], (v4f)(gat_dpp<0x55>(w)), acc);
              acc = __builtin_elementwise_fma(val[2], (v4f)(gat_dpp<0xAA>(w)), acc);
              acc = __builtin_elementwise_fma(val[3], (v4f)(gat_dpp<0xFF>(w)), acc);
            }
          }
          continue;
        }
        v4f val[U];
        v4f sm[U];
        int32_t ee[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int jdx = k + u * NB + sub;
          const int bi = (jdx < cnt ? jdx : 0) * 4;  // lanes past the end re-read edge 0 (valid memory), weight zeroed below
          const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute(bi, (int)goff) + f4;
          const uint32_t so = (uint32_t)__builtin_amdgcn_ds_bpermute(bi, (int)soff) + h4;
          ee[u] = DROP ? __builtin_amdgcn_ds_bpermute(bi, eid) : 0;
          val[u] = (v4f)*reinterpret_cast<const V4*>(gatb + off);
          if (MODE == GAT_BWD_SRC) sm[u] = *reinterpret_cast<const v4f*>(smallb + so);
          else if (!ELK) sm[u].x = *reinterpret_cast<const float*>(smallb + so);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = k + u * NB + sub < cnt;
          float keep = 1.f;
          if (DROP) {
            keep = gat_hash(a.seed, (uint32_t)ee[u], (uint32_t)head) >= a.drop_below ? a.keep_scale : 0.f;
          }
          if (ELK && MODE != GAT_BWD_SRC) sm[u].x = head_dot(val[u]);
          if (MODE == GAT_FWD) {
            const float t = sm[u].x + c_er;
            const float z = t > 0.f ? t : t * a.slope;
            const float mn = live ? fmaxf(run_m, z) : run_m;
            const float rescale = run_m > -INFINITY ? __expf(run_m - mn) : 0.f;   // 1 while the maximum stands
            const float pe = live ? __expf(z - mn) : 0.f;
            run_s = run_s * rescale + pe;
            acc = acc * rescale + val[u] * (pe * keep);
            run_m = mn;
          } else {
            const float t = MODE == GAT_BWD_DST ? sm[u].x + c_er : c_el + sm[u].x;
            const float mm = MODE == GAT_BWD_DST ? c_m : sm[u].y;
            const float is = MODE == GAT_BWD_DST ? c_is : sm[u].z;
            const float tt = MODE == GAT_BWD_DST ? c_t : sm[u].w;
            const float z = t > 0.f ? t : t * a.slope;
            const float av = __expf(z - mm) * is;
            const float dot = head_sum<LPH>(val[u].x * ra.x + val[u].y * ra.y + val[u].z * ra.z + val[u].w * ra.w);
            float de = av * (dot * keep - tt) * (t > 0.f ? 1.f : a.slope);
            de = live ? de : 0.f;
            hacc += de;
            if (MODE == GAT_BWD_SRC) {
              const float w = live ? av * keep : 0.f;
              acc += val[u] * w;
            }
          }
        }
      }
    }
    // ---- combine the lane groups, write the row (or the partial slot of a hub chunk)
    if (MODE == GAT_FWD) {
#pragma unroll
      for (int off = G; off < kWave; off <<= 1) {
        const float mo = __shfl_xor(run_m, off, kWave), so = __shfl_xor(run_s, off, kWave);
        const v4f ao = vec_shfl_xor<4>(acc, off);
        const float mn = fmaxf(run_m, mo);
        const float f1 = run_m > -INFINITY ? __expf(run_m - mn) : 0.f, f2 = mo > -INFINITY ? __expf(mo - mn) : 0.f;
        run_s = run_s * f1 + so * f2;
        acc = acc * f1 + ao * f2;
        run_m = mn;
      }
      if (irow >= 0) {
        const float is = run_s > 0.f ? 1.f / run_s : 0.f;  // rows without in-edges aggregate to 0
        acc = acc * is;
        if (fact && sub == 0 && (l % LPH) == 0) {
          v4f st;
          st.x = c_er;
          st.y = run_m > -INFINITY ? run_m : 0.f;
          st.z = is;
          st.w = 0.f;
          *reinterpret_cast<v4f*>(a.nstat_w + (row * H + head) * 4) = st;
        }
      } else if (fact && sub == 0 && (l % LPH) == 0) {  // hub chunk: statistics travel with the unnormalised partial row
        const int64_t slot = -(irow + 1);
        a.partial_h[slot * 2 * H + head] = run_m;
        a.partial_h[slot * 2 * H + H + head] = run_s;
      }
    } else if (MODE != GAT_BWD_DST) {
#pragma unroll
      for (int off = G; off < kWave; off <<= 1) acc += vec_shfl_xor<4>(acc, off);
    }
    if (MODE != GAT_FWD) {
      if (QUAD) hacc = head_sum<4>(hacc);  // every lane of the head accumulated its own edges' terms
#pragma unroll
      for (int off = G; off < kWave; off <<= 1) hacc += __shfl_xor(hacc, off, kWave);
    }
    if (fact && sub == 0) {
      if (MODE != GAT_BWD_DST) {
        float* op = irow >= 0 ? a.out + row * (int64_t)D + f : a.partial + (-(irow + 1)) * (int64_t)D + f;
        if (!tail) {
          *reinterpret_cast<V4*>(op) = acc;
        } else {  // own columns f .. f+nvalid-1 are components 4-nvalid .. 3 of the window
          const float* av = reinterpret_cast<const float*>(&acc);
          for (int j = 0; j < nvalid; ++j) op[j] = av[4 - nvalid + j];
        }
      }
      if (MODE != GAT_FWD && (l % LPH) == 0) {
        float* op = irow >= 0 ? a.out_h + row * (int64_t)H + head : a.partial_h + (-(irow + 1)) * (int64_t)H + head;
        *op = hacc;
      }
    }
  }
}

template <int G, int LPH, int MODE>
static void gat_launch_drop(const GatArgs& a, bool drop, hipStream_t s) {
  const dim3 grid((unsigned)a.nblocks), block(kBlock);
  if (G >= 16 && LPH >= 2 && LPH < G && a.attn) {  // el formed in the kernel (set by the entry points for unpacked multi-head rows)
    constexpr int GE = G >= 16 ? G : 16, LE = (LPH >= 2 && LPH < G) ? LPH : 2;
    if (drop) hipLaunchKernelGGL((gat_fused_kernel<GE, LE, MODE, true, false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gat_fused_kernel<GE, LE, MODE, false, false, true>), grid, block, 0, s, a);
    return;
  }
  if (drop) hipLaunchKernelGGL((gat_fused_kernel<G, LPH, MODE, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((gat_fused_kernel<G, LPH, MODE, false>), grid, block, 0, s, a);
}

template <int G, int MODE>
static void gat_launch_ragged(const GatArgs& a, bool drop, hipStream_t s) {  // one head: the whole lane group is the head
  const dim3 grid((unsigned)a.nblocks), block(kBlock);
  if (drop) hipLaunchKernelGGL((gat_fused_kernel<G, G, MODE, true, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((gat_fused_kernel<G, G, MODE, false, true>), grid, block, 0, s, a);
}

template <int G, int MODE>
static bool gat_launch_lph(const GatArgs& a, int lph, bool drop, hipStream_t s) {
  switch (lph) {
#define MGX_GAT_LPH(L) case L: if (L <= G) { gat_launch_drop<G, (L <= G ? L : G), MODE>(a, drop, s); return true; } return false;
    MGX_GAT_LPH(1) MGX_GAT_LPH(2) MGX_GAT_LPH(4) MGX_GAT_LPH(8) MGX_GAT_LPH(16) MGX_GAT_LPH(32) MGX_GAT_LPH(64)
#undef MGX_GAT_LPH
    default: return false;
  }
}

template <int MODE>
static bool gat_launch(const GatArgs& a, bool drop, hipStream_t s) {
  int G = 1;
  while (G * 4 < a.D) G <<= 1;
  if (a.F % 4 != 0 || ((a.F / 4) & (a.F / 4 - 1)) != 0) {  // gat_check admitted it: H == 1, 4 < F <= 256 (whole lane group = the head)
    switch (G) {
      case 2: gat_launch_ragged<2, MODE>(a, drop, s); return true;
      case 4: gat_launch_ragged<4, MODE>(a, drop, s); return true;
      case 8: gat_launch_ragged<8, MODE>(a, drop, s); return true;
      case 16: gat_launch_ragged<16, MODE>(a, drop, s); return true;
      case 32: gat_launch_ragged<32, MODE>(a, drop, s); return true;
      case 64: gat_launch_ragged<64, MODE>(a, drop, s); return true;
      default: return false;
    }
  }
  const int lph = a.F / 4;
  switch (G) {
    case 1: return gat_launch_lph<1, MODE>(a, lph, drop, s);
    case 2: return gat_launch_lph<2, MODE>(a, lph, drop, s);
    case 4: return gat_launch_lph<4, MODE>(a, lph, drop, s);
    case 8: return gat_launch_lph<8, MODE>(a, lph, drop, s);
    case 16: return gat_launch_lph<16, MODE>(a, lph, drop, s);
    case 32: return gat_launch_lph<32, MODE>(a, lph, drop, s);
    case 64: return gat_launch_lph<64, MODE>(a, lph, drop, s);
    default: return false;
  }
}

static int gat_rows_per_block() { return 16; }  // work items per workgroup: the g-SpMM's measured choice (spmm.hip)

static int32_t gat_check(const mgx_csr* csr, const mgx_spmm_plan* plan, int64_t H, int64_t F, int64_t gathered_rows, float p,
                         const char* who) {
  MGX_CHECK_ARG(csr != nullptr, "%s: csr is NULL", who);
  if (csr->idx_bits != 32) MGX_UNSUPPORTED("%s: 32-bit graph indices only (got %d)", who, csr->idx_bits);
  MGX_CHECK_ARG(H >= 1 && F >= 1, "%s: H and F must be positive", who);
  const bool ragged_ok = H == 1 && F > 4 && F <= 256;  // one head of any width: the ragged-window kernels
  if (!ragged_ok && (F % 4 != 0 || ((F / 4) & (F / 4 - 1)) != 0 || H * F > 256))
    MGX_UNSUPPORTED("%s: needs F in {4, 8, 16, ..., 256} and H*F <= 256, or one head with 4 < F <= 256 (got H = %lld, F = %lld)", who,
                    (long long)H, (long long)F);
  if (gathered_rows * H * F * 4 >= (int64_t(1) << 32) || gathered_rows * H * 16 >= (int64_t(1) << 32) || csr->nnz >= (int64_t(1) << 31))
    MGX_UNSUPPORTED("%s: operands beyond 32-bit byte offsets", who);
  MGX_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout probability %g outside [0, 1)", who, (double)p);
  MGX_CHECK_ARG(csr->num_rows == 0 || csr->indptr, "%s: indptr is NULL", who);
  MGX_CHECK_ARG(csr->nnz == 0 || csr->indices, "%s: indices is NULL", who);
  if (plan) {
    MGX_CHECK_ARG(plan->item_row && plan->item_beg && plan->item_end && plan->item_node && plan->num_items >= csr->num_rows,
                  "%s: malformed plan", who);
    MGX_CHECK_ARG(plan->num_slots == 0 || (plan->hub_row && plan->hub_slot_ptr), "%s: plan has split rows but no hub tables", who);
  }
  return MGX_OK;
}

static void gat_fill(GatArgs& a, const mgx_csr* csr, const mgx_spmm_plan* plan, int64_t H, int64_t F, float slope, float p,
                     uint64_t seed) {
  memset(&a, 0, sizeof(a));
  a.indptr = (const int32_t*)csr->indptr; a.indices = (const int32_t*)csr->indices; a.eids = (const int32_t*)csr->eids;
  a.n_items = csr->num_rows;
  if (plan) {
    a.item_row = plan->item_row; a.item_beg = (const int32_t*)plan->item_beg; a.item_end = (const int32_t*)plan->item_end;
    a.item_node = plan->item_node; a.n_items = plan->num_items;
  }
  a.rpb = gat_rows_per_block();
  a.nblocks = xcd_ranges(plan, a.n_items, a.rpb, a.xcd);
  a.H = (int)H; a.F = (int)F; a.D = (int)(H * F);
  a.gat_ld = a.D; a.small_ld = 0;  // small_ld: set by the caller (H for el, 4H for nstat, the packed stride otherwise)
  a.slope = slope; a.seed = seed;
  a.keep_scale = 1.f / (1.f - p);
  double thr = (double)p * 4294967296.0;
  a.drop_below = thr >= 4294967295.0 ? 4294967295u : (uint32_t)thr;
}

// Packed gather operand: row r = [rows[r, 0..D) | pad to a multiple of 4 | small[r, 0..S) | pad to whole 128-byte lines].
// A narrow layer (reddit GAT: D = 16, one head) gathers a 64-byte feature row AND a 4-byte attention term per edge -- two L2
// requests of one line each, and the kernels are bound by requests; packed, both arrive in ONE line.
// Returns the packed row stride in floats, or 0 when packing would not save a line per edge or costs too much footprint.
static int gat_pack_ld(int64_t D, int64_t S) {
  const int64_t off = round_up(D, 4);
  const int64_t ld = round_up(off + S, 32);
  // not when the padding would more than double the bytes a node occupies in L2 (8 features + 1 term in a 128-byte row)
  return (ld <= round_up(D, 32) && ld <= 2 * (D + S)) ? (int)ld : 0;
}

__global__ __launch_bounds__(kBlock) void gat_pack_kernel(int64_t n, int D, int S, int ld, int off, const float* __restrict__ rows,
                                                          const float* __restrict__ small, float* __restrict__ out) {
  const int64_t total = n * (int64_t)ld;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / ld;
    const int c = (int)(i - r * ld);
    float v = 0.f;
    if (c < D) v = rows[r * D + c];
    else if (c >= off && c < off + S) v = small[r * S + (c - off)];
    out[i] = v;
  }
}

static void gat_pack(int64_t n, int D, int S, int ld, const float* rows, const float* small, float* out, hipStream_t s) {
  int64_t b = (n * (int64_t)ld + kBlock - 1) / kBlock;
  if (b > 256 * 64) b = 256 * 64;
  hipLaunchKernelGGL(gat_pack_kernel, dim3((unsigned)(b < 1 ? 1 : b)), dim3(kBlock), 0, s, n, D, S, ld, (int)round_up(D, 4), rows, small, out);
}

static void gat_fixup(const mgx_spmm_plan* plan, int L, const float* partial, float* out, hipStream_t s) {
  hipLaunchKernelGGL(gat_rows_fixup_kernel, dim3((unsigned)((plan->num_hubs + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, s,
                     plan->hub_row, plan->hub_slot_ptr, plan->num_hubs, L, partial, out);
}

}  // namespace mgx

#include "gat_tile.inc"

extern "C" int64_t mgx_gat_fused_workspace(const mgx_spmm_plan* plan, int64_t H, int64_t F) {
  const int64_t slots = plan ? plan->num_slots : 0;
  const int64_t per = H * F + 2 * H;  // [slots, D] partial rows + [slots, 2H] chunk statistics (forward) / [slots, H] (backward)
  return slots * per * (int64_t)sizeof(float);
}

extern "C" int64_t mgx_gat_fused_pack_workspace(int64_t num_src, int64_t num_dst, int64_t H, int64_t F) {
  using namespace mgx;
  const int la = gat_pack_ld(H * F, H), lb = gat_pack_ld(H * F, 4 * H);
  const int64_t rows = num_src > num_dst ? num_src : num_dst;
  const int64_t ld = la > lb ? la : lb;
  if (ld == 0 || rows * ld * 4 >= (int64_t(1) << 32)) return 0;
  return rows * ld * (int64_t)sizeof(float);
}

static bool gat_el_in_kernel(int64_t H, int64_t F, const float* attn_l) {
  // heads of up to 16 columns only: the reduction over a head's lanes costs log2(F / 4) lane swaps per edge -- measured: 8 x 16
  // (reddit-small, BASELINE config 3) epoch 5.61 -> 5.51 ms, but 4 x 64 (arxiv) 2.79 -> 2.89 ms with the swaps of 16-lane heads
  return attn_l != nullptr && H > 1 && F % 4 == 0 && F >= 8 && F <= 16 && H * F >= 64 && (uintptr_t)attn_l % 16 == 0;
}

extern "C" int32_t mgx_gat_fused_fwd(const mgx_csr* csr, const mgx_spmm_plan* plan, int64_t H, int64_t F, const float* feat,
                                     const float* el, const float* attn_l, const float* er, float negative_slope, float drop_p,
                                     uint64_t seed, float* out, float* nstat, void* workspace, void* pack_ws, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  int32_t st = gat_check(csr, plan, H, F, csr ? csr->num_cols : 0, drop_p, "mgx_gat_fused_fwd");
  if (st != MGX_OK) return st;
  if (csr->num_rows == 0) return MGX_OK;
  MGX_CHECK_ARG(el && er && out && nstat && (feat || csr->nnz == 0), "mgx_gat_fused_fwd: NULL pointer");
  MGX_CHECK_ARG((uintptr_t)feat % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)nstat % 16 == 0, "mgx_gat_fused_fwd: pointers must be 16-byte aligned");
  const bool hubs = plan && plan->num_slots > 0;
  MGX_CHECK_ARG(!hubs || workspace, "mgx_gat_fused_fwd: plan has split rows but no workspace");
  hipStream_t s = (hipStream_t)stream;
  GatArgs a;
  gat_fill(a, csr, plan, H, F, negative_slope, drop_p, seed);
  a.el = el; a.er = er; a.nstat_w = nstat;
  a.gat = feat; a.nstat = nstat; a.out = out;
  a.small_ld = (int)H;
  const int pld = pack_ws ? gat_pack_ld(H * F, H) : 0;
  if (pld && csr->num_cols * (int64_t)pld * 4 < (int64_t(1) << 32)) {  // [feat | el] rows: one line per gathered source
    gat_pack(csr->num_cols, a.D, (int)H, pld, feat, el, (float*)pack_ws, s);
    MGX_CHECK_LAUNCH();
    a.gat = (const float*)pack_ws; a.el = (const float*)pack_ws + round_up(a.D, 4);
    a.gat_ld = pld; a.small_ld = pld;
  } else if (gat_el_in_kernel(H, F, attn_l)) {
    a.attn = attn_l;  // unpacked multi-head rows: el from the gathered row
  }
  float* ws = (float*)workspace;  // [slots, D] partial rows, then [slots, 2H] chunk statistics
  a.partial = ws;
  a.partial_h = hubs ? ws + plan->num_slots * (int64_t)a.D : nullptr;
  if (!gat_launch<GAT_FWD>(a, drop_p > 0.f, s)) MGX_UNSUPPORTED("mgx_gat_fused_fwd: unsupported head layout H = %lld, F = %lld", (long long)H, (long long)F);
  MGX_CHECK_LAUNCH();
  if (hubs) {
    hipLaunchKernelGGL(gat_online_fixup_kernel, dim3((unsigned)((plan->num_hubs + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, s,
                       plan->hub_row, plan->hub_slot_ptr, plan->num_hubs, (int)H, (int)F, er, (const float*)a.partial,
                       (const float*)a.partial_h, out, nstat);
    MGX_CHECK_LAUNCH();
  }
  return MGX_OK;
}

extern "C" int32_t mgx_gat_fused_bwd(const mgx_csr* csc, const mgx_spmm_plan* csc_plan, const mgx_csr* csr, const mgx_spmm_plan* csr_plan,
                                     int64_t H, int64_t F, const float* feat, const float* el, const float* attn_l, float negative_slope,
                                     float drop_p, uint64_t seed, const float* out, const float* d_out, float* nstat, float* d_feat,
                                     float* d_el, float* d_er, void* workspace, void* pack_ws, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  int32_t st = gat_check(csc, csc_plan, H, F, csc ? csc->num_cols : 0, drop_p, "mgx_gat_fused_bwd");
  if (st != MGX_OK) return st;
  st = gat_check(csr, csr_plan, H, F, csr ? csr->num_cols : 0, drop_p, "mgx_gat_fused_bwd");
  if (st != MGX_OK) return st;
  MGX_CHECK_ARG(csc->num_rows == csr->num_cols && csc->num_cols == csr->num_rows && csc->nnz == csr->nnz,
                "mgx_gat_fused_bwd: the two CSRs are not transposes of each other");
  if (csc->num_rows == 0 || csr->num_rows == 0) return MGX_OK;
  MGX_CHECK_ARG(feat && el && out && d_out && nstat && d_er && (d_feat || !d_el || true), "mgx_gat_fused_bwd: NULL pointer");
  MGX_CHECK_ARG((uintptr_t)feat % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)d_out % 16 == 0 && (uintptr_t)nstat % 16 == 0 &&
                (!d_feat || (uintptr_t)d_feat % 16 == 0), "mgx_gat_fused_bwd: pointers must be 16-byte aligned");
  const bool hubs_dst = csc_plan && csc_plan->num_slots > 0, hubs_src = csr_plan && csr_plan->num_slots > 0;
  MGX_CHECK_ARG(!(hubs_dst || hubs_src) || workspace, "mgx_gat_fused_bwd: plan has split rows but no workspace");
  hipStream_t s = (hipStream_t)stream;
  const int D = (int)(H * F);
  {  // destination side: t[v,h] and d er
    GatArgs a;
    gat_fill(a, csc, csc_plan, H, F, negative_slope, drop_p, seed);
    a.gat = feat; a.el = el; a.nstat = nstat; a.nstat_w = nstat; a.rowa = d_out; a.rowb = out; a.out_h = d_er;
    a.small_ld = (int)H;
    const int pld = pack_ws ? gat_pack_ld(H * F, H) : 0;
    if (pld && csc->num_cols * (int64_t)pld * 4 < (int64_t(1) << 32)) {  // [feat | el] rows, as in the forward
      gat_pack(csc->num_cols, D, (int)H, pld, feat, el, (float*)pack_ws, s);
      MGX_CHECK_LAUNCH();
      a.gat = (const float*)pack_ws; a.el = (const float*)pack_ws + round_up(D, 4);
      a.gat_ld = pld; a.small_ld = pld;
    } else if (gat_el_in_kernel(H, F, attn_l)) {
      a.attn = attn_l;
    }
    a.partial_h = (float*)workspace;
    if (!gat_launch<GAT_BWD_DST>(a, drop_p > 0.f, s)) MGX_UNSUPPORTED("mgx_gat_fused_bwd: unsupported head layout");
    MGX_CHECK_LAUNCH();
    if (hubs_dst) {
      gat_fixup(csc_plan, (int)H, (const float*)workspace, d_er, s);
      MGX_CHECK_LAUNCH();
    }
  }
  if (d_feat || d_el) {  // source side: d feat and d el (needs t of every destination: after the launch above, same stream)
    MGX_CHECK_ARG(d_feat && d_el, "mgx_gat_fused_bwd: d_feat and d_el come together");
    GatArgs a;
    gat_fill(a, csr, csr_plan, H, F, negative_slope, drop_p, seed);
    a.gat = d_out; a.el = el; a.nstat = nstat; a.rowa = feat; a.out = d_feat; a.out_h = d_el;
    {  // the logit's el[u] must be the number the other two walks used: formed in the kernel exactly when they formed it
      const int pld_el = pack_ws ? gat_pack_ld(H * F, H) : 0;
      const bool packed_el = pld_el && csc->num_cols * (int64_t)pld_el * 4 < (int64_t(1) << 32);
      if (!packed_el && gat_el_in_kernel(H, F, attn_l)) a.attn = attn_l;
    }
    a.small_ld = 4 * (int)H;
    const int pld = pack_ws ? gat_pack_ld(H * F, 4 * H) : 0;
    if (pld && csr->num_cols * (int64_t)pld * 4 < (int64_t(1) << 32)) {  // [d out | (er, m, 1/s, t)] rows: t was written just above
      gat_pack(csr->num_cols, D, 4 * (int)H, pld, d_out, nstat, (float*)pack_ws, s);
      MGX_CHECK_LAUNCH();
      a.gat = (const float*)pack_ws; a.nstat = (const float*)pack_ws + round_up(D, 4);
      a.gat_ld = pld; a.small_ld = pld;
    }
    float* ws = (float*)workspace;
    a.partial = ws;
    a.partial_h = ws ? ws + (csr_plan ? csr_plan->num_slots : 0) * (int64_t)D : nullptr;
    if (!gat_launch<GAT_BWD_SRC>(a, drop_p > 0.f, s)) MGX_UNSUPPORTED("mgx_gat_fused_bwd: unsupported head layout");
    MGX_CHECK_LAUNCH();
    if (hubs_src) {
      gat_fixup(csr_plan, D, a.partial, d_feat, s);
      gat_fixup(csr_plan, (int)H, a.partial_h, d_el, s);
      MGX_CHECK_LAUNCH();
    }
  }
  return MGX_OK;
}

// ---- tile forms (gat_tile.inc): one head of 4 .. 16 columns on a graph with a tile plan that carries edge ids and node ids.
// `plan` is the tile plan's BASE plan (its hub tables and slots size the workspace: mgx_gat_fused_workspace(plan, 1, F)).
extern "C" int32_t mgx_gat_tile_fwd(const mgx_csr* csr, const mgx_spmm_plan* plan, const mgx_tile_plan* tp, int64_t H, int64_t F,
                                    const float* feat, const float* el, const float* er, float negative_slope, float drop_p,
                                    uint64_t seed, float* out, float* nstat, void* workspace, void* pack_ws, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  int32_t st = gat_check(csr, plan, H, F, csr ? csr->num_cols : 0, drop_p, "mgx_gat_tile_fwd");
  if (st != MGX_OK) return st;
  if (!gat_tile_shape_ok(tp, H, F)) MGX_UNSUPPORTED("mgx_gat_tile_fwd: one head of 4 .. 16 columns and a 4-lane tile plan with node ids");
  MGX_CHECK_ARG(drop_p == 0.f || tp->lds_stream16, "mgx_gat_tile_fwd: attn_drop needs a tile plan with the (slot, rank) stream");
  if (csr->num_cols >= (int64_t(1) << 24)) MGX_UNSUPPORTED("mgx_gat_tile_fwd: sources beyond 24 bits");
  if (csr->num_rows == 0 || tp->num_tiles == 0) return MGX_OK;
  MGX_CHECK_ARG(el && er && out && nstat && feat, "mgx_gat_tile_fwd: NULL pointer");
  MGX_CHECK_ARG((uintptr_t)feat % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)nstat % 16 == 0, "mgx_gat_tile_fwd: pointers must be 16-byte aligned");
  const bool hubs = plan && plan->num_slots > 0;
  MGX_CHECK_ARG(!hubs || workspace, "mgx_gat_tile_fwd: plan has split rows but no workspace");
  hipStream_t s = (hipStream_t)stream;
  GatTileArgs a;
  gat_tile_fill(a, tp, F, negative_slope, drop_p, seed);
  a.gat = feat; a.small = el; a.er = er; a.nstat_w = nstat; a.out = out;
  const int pld = pack_ws ? gat_pack_ld(F, 1) : 0;
  if (pld && csr->num_cols * (int64_t)pld * 4 < (int64_t(1) << 32)) {  // [feat | el] rows: a source's row and record share a line
    gat_pack(csr->num_cols, (int)F, 1, pld, feat, el, (float*)pack_ws, s);
    MGX_CHECK_LAUNCH();
    a.gat = (const float*)pack_ws; a.small = (const float*)pack_ws + round_up(F, 4);
    a.gat_ld = pld; a.small_ld = pld;
  }
  float* ws = (float*)workspace;
  a.partial = ws;
  a.partial_h = hubs ? ws + plan->num_slots * F : nullptr;
  if (!gat_tile_launch<GAT_FWD>(a, tp->nacc, drop_p > 0.f, s)) MGX_UNSUPPORTED("mgx_gat_tile_fwd: no kernel for %d rows per lane group", tp->nacc);
  MGX_CHECK_LAUNCH();
  if (hubs) {
    hipLaunchKernelGGL(gat_online_fixup_kernel, dim3((unsigned)((plan->num_hubs + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, s,
                       plan->hub_row, plan->hub_slot_ptr, plan->num_hubs, 1, (int)F, er, (const float*)a.partial,
                       (const float*)a.partial_h, out, nstat);
    MGX_CHECK_LAUNCH();
  }
  return MGX_OK;
}

extern "C" int32_t mgx_gat_tile_bwd(const mgx_csr* csc, const mgx_spmm_plan* csc_plan, const mgx_tile_plan* csc_tp, const mgx_csr* csr,
                                    const mgx_spmm_plan* csr_plan, const mgx_tile_plan* csr_tp, int64_t H, int64_t F, const float* feat,
                                    const float* el, float negative_slope, float drop_p, uint64_t seed, const float* out,
                                    const float* d_out, float* nstat, float* d_feat, float* d_el, float* d_er, void* workspace,
                                    void* pack_ws, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  int32_t st = gat_check(csc, csc_plan, H, F, csc ? csc->num_cols : 0, drop_p, "mgx_gat_tile_bwd");
  if (st != MGX_OK) return st;
  st = gat_check(csr, csr_plan, H, F, csr ? csr->num_cols : 0, drop_p, "mgx_gat_tile_bwd");
  if (st != MGX_OK) return st;
  if (!gat_tile_shape_ok(csc_tp, H, F) || !gat_tile_shape_ok(csr_tp, H, F))
    MGX_UNSUPPORTED("mgx_gat_tile_bwd: one head of 4 .. 16 columns and 4-lane tile plans with node ids on both CSRs");
  MGX_CHECK_ARG(drop_p == 0.f || (csc_tp->lds_stream16 && csr_tp->lds_stream16),
                "mgx_gat_tile_bwd: attn_drop needs tile plans with the (slot, rank) stream");
  if (csc->num_cols >= (int64_t(1) << 24) || csr->num_cols >= (int64_t(1) << 24)) MGX_UNSUPPORTED("mgx_gat_tile_bwd: nodes beyond 24 bits");
  MGX_CHECK_ARG(csc->num_rows == csr->num_cols && csc->num_cols == csr->num_rows && csc->nnz == csr->nnz,
                "mgx_gat_tile_bwd: the two CSRs are not transposes of each other");
  if (csc->num_rows == 0 || csr->num_rows == 0) return MGX_OK;
  MGX_CHECK_ARG(feat && el && out && d_out && nstat && d_er, "mgx_gat_tile_bwd: NULL pointer");
  MGX_CHECK_ARG((uintptr_t)feat % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)d_out % 16 == 0 && (uintptr_t)nstat % 16 == 0 &&
                (!d_feat || (uintptr_t)d_feat % 16 == 0), "mgx_gat_tile_bwd: pointers must be 16-byte aligned");
  const bool hubs_dst = csc_plan && csc_plan->num_slots > 0, hubs_src = csr_plan && csr_plan->num_slots > 0;
  MGX_CHECK_ARG(!(hubs_dst || hubs_src) || workspace, "mgx_gat_tile_bwd: plan has split rows but no workspace");
  hipStream_t s = (hipStream_t)stream;
  if (csc_tp->num_tiles > 0) {  // destination side: t[v] and d er
    GatTileArgs a;
    gat_tile_fill(a, csc_tp, F, negative_slope, drop_p, seed);
    a.gat = feat; a.small = el; a.nstat = nstat; a.nstat_w = nstat; a.rowa = d_out; a.rowb = out; a.out_h = d_er;
    const int pld = pack_ws ? gat_pack_ld(F, 1) : 0;
    if (pld && csc->num_cols * (int64_t)pld * 4 < (int64_t(1) << 32)) {
      gat_pack(csc->num_cols, (int)F, 1, pld, feat, el, (float*)pack_ws, s);
      MGX_CHECK_LAUNCH();
      a.gat = (const float*)pack_ws; a.small = (const float*)pack_ws + round_up(F, 4);
      a.gat_ld = pld; a.small_ld = pld;
    }
    a.partial_h = (float*)workspace;
    if (!gat_tile_launch<GAT_BWD_DST>(a, csc_tp->nacc, drop_p > 0.f, s)) MGX_UNSUPPORTED("mgx_gat_tile_bwd: no kernel for this tile plan");
    MGX_CHECK_LAUNCH();
    if (hubs_dst) {
      gat_fixup(csc_plan, 1, (const float*)workspace, d_er, s);
      MGX_CHECK_LAUNCH();
    }
  }
  if ((d_feat || d_el) && csr_tp->num_tiles > 0) {  // source side: d feat and d el (t of every destination was written just above)
    MGX_CHECK_ARG(d_feat && d_el, "mgx_gat_tile_bwd: d_feat and d_el come together");
    GatTileArgs a;
    gat_tile_fill(a, csr_tp, F, negative_slope, drop_p, seed);
    a.gat = d_out; a.small = nstat; a.small_ld = 4; a.rowa = feat; a.el_row = el; a.out = d_feat; a.out_h = d_el;
    const int pld = pack_ws ? gat_pack_ld(F, 4) : 0;
    if (pld && csr->num_cols * (int64_t)pld * 4 < (int64_t(1) << 32)) {  // [d out | (er, m, 1/s, t)] rows
      gat_pack(csr->num_cols, (int)F, 4, pld, d_out, nstat, (float*)pack_ws, s);
      MGX_CHECK_LAUNCH();
      a.gat = (const float*)pack_ws; a.small = (const float*)pack_ws + round_up(F, 4);
      a.gat_ld = pld; a.small_ld = pld;
    }
    float* ws = (float*)workspace;
    a.partial = ws;
    a.partial_h = ws ? ws + (csr_plan ? csr_plan->num_slots : 0) * F : nullptr;
    if (!gat_tile_launch<GAT_BWD_SRC>(a, csr_tp->nacc, drop_p > 0.f, s)) MGX_UNSUPPORTED("mgx_gat_tile_bwd: no kernel for this tile plan");
    MGX_CHECK_LAUNCH();
    if (hubs_src) {
      gat_fixup(csr_plan, (int)F, a.partial, d_feat, s);
      gat_fixup(csr_plan, 1, a.partial_h, d_el, s);
      MGX_CHECK_LAUNCH();
    }
  }
  return MGX_OK;
}
