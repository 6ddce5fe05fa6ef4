// sddmm.hip -- g-SDDMM for gfx950 (MI355X).
//
// Replaces what DGL's _CAPI_DGLKernelSDDMM executes for kernel/dgl-new.py:39 (dgl.ops.gsddmm),
// apply_edges(fn.u_add_v) inside GATConv (main_dgl_reddit_gat.py:10), fn.u_dot_v
// (link_prediction/gcmc_dgl/model.py:342) and the per-edge gathers of UDF messages
// (edges.src/edges.dst, main_dgl_molhiv_gcn.py:50-52).  Dense op definitions: kernel/utils.py:8-16.
//
// Edge-parallel, lanes along the feature dimension: a group of G lanes owns one edge and moves
// its feature row with 16-byte accesses; 64/G edges per wave-instruction, kUn instructions in
// flight.  Two edge walks share the bodies:
//   COO: a wave owns kCooChunks*64 consecutive edge ids; output rows are written contiguously;
//   CSR: a wave owns one work item (destination row or chunk of a hub row) of the in-CSR --
//        used when COO is not materialised (formats(['csr','csc'])) and for graphs whose edge ids
//        ARE the CSR positions (GraphIndex.canonical()), where it also writes contiguously.
// In both, the ids of 64 edges arrive by ONE coalesced load (lane j <- edge j) and are handed to
// the lane groups with ds_bpermute; the next 64 are requested before the current gathers issue.
// `dot` keeps kUn edges in flight and reduces inside the lane group with xor-shuffles.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace mgx {

template <typename Idx>
struct SddmmArgs {
  // COO form
  const Idx* src;
  const Idx* dst;
  const Idx* perm;  // optional (lean COO walk only): output row of the q-th WALKED edge -- an edge list walked in another order than its ids
  // CSR form (in-CSR: row = dst) + optional schedule
  const Idx* indptr;
  const Idx* indices;
  const Idx* eids;
  const int32_t* item_node;  // [n_items] destination row of each work item (NULL: item == row)
  const Idx* item_beg;
  const Idx* item_end;
  int64_t n_items;
  int64_t n_rows;
  int64_t n_cols;
  int64_t nblocks;
  int64_t nnz;
  const float* L;
  const float* R;
  const int64_t* l_off;
  const int64_t* r_off;
  float* out;
  int64_t l_len, r_len, out_len, reduce_size;
  int op, lhs_target, rhs_target;
  int coo_chunk;   // COO walk: edges per chunk (<= 64) ...
  int coo_chunks;  // ... and chunks per wave, chosen on the host so that small edge lists still fill the chip
};

__device__ __forceinline__ int64_t pick_target(int t, int64_t u, int64_t e, int64_t v) {
  return t == MGX_TARGET_U ? u : (t == MGX_TARGET_V ? v : e);
}

template <typename V>
__device__ __forceinline__ V sddmm_op(int op, V l, V r) {
  switch (op) {
    case MGX_OP_ADD: return l + r;
    case MGX_OP_SUB: return l - r;
    case MGX_OP_MUL: return l * r;
    case MGX_OP_DIV: return l / r;
    case MGX_OP_COPY_LHS: return l;
    default: return r;
  }
}

constexpr int kUn = 4;          // edges in flight per lane group
constexpr int kCooChunks = 8;   // COO: at most this many 64-edge chunks per wave (large edge lists)
constexpr int kCooTargetWaves = 4096;  // small edge lists: shrink the per-wave share until about this many waves exist
constexpr int kCsrItems = 16;   // CSR: work items per workgroup

// element-wise body: kUn edges (u, v, e), e < 0 = idle.  DIRECT: operands used as-is (VEC-wide).
template <typename Idx, int VEC, int G, bool DIRECT>
__device__ __forceinline__ void sddmm_edges(const SddmmArgs<Idx>& a, const int64_t (&u)[kUn], const int64_t (&v)[kUn],
                                            const int64_t (&e)[kUn], int kc) {
  typedef typename VecT<VEC>::type V;
  const bool kactive = kc < a.out_len;
  int64_t lo = kc, ro = kc;
  if (!DIRECT && kactive) {
    lo = !a.L ? 0 : a.l_off ? a.l_off[kc] : (a.l_len == a.out_len ? kc : kc / (a.out_len / a.l_len));
    ro = !a.R ? 0 : a.r_off ? a.r_off[kc] : (a.r_len == a.out_len ? kc : kc / (a.out_len / a.r_len));
  }
  V lv[kUn], rv[kUn];
#pragma unroll
  for (int i = 0; i < kUn; ++i) {
    lv[i] = (V)(0.f);
    rv[i] = (V)(0.f);
    if (e[i] >= 0 && kactive) {
      if (a.L) lv[i] = *reinterpret_cast<const V*>(a.L + pick_target(a.lhs_target, u[i], e[i], v[i]) * a.l_len + lo);
      if (a.R) rv[i] = *reinterpret_cast<const V*>(a.R + pick_target(a.rhs_target, u[i], e[i], v[i]) * a.r_len + ro);
    }
  }
#pragma unroll
  for (int i = 0; i < kUn; ++i)
    if (e[i] >= 0 && kactive) {
      V* op = reinterpret_cast<V*>(a.out + e[i] * a.out_len + kc);
      // rows of >= 32 bytes: the E x D output is written once and not re-read here -- stream it past L2 (reddit-shape
      // u_add_v, D = 128: 20.0 -> 15.5 ms); narrower rows share cache lines between edges and stay on the normal path
      if (VEC * G >= 8) __builtin_nontemporal_store(sddmm_op<V>(a.op, lv[i], rv[i]), op);
      else *op = sddmm_op<V>(a.op, lv[i], rv[i]);
    }
}

// dot body: out[e,k] = sum_j L[t_l(e), lo(k)*RS + j] * R[t_r(e), ro(k)*RS + j]; G lanes per edge along j.
// All lanes of the wave execute this together (idle edges join the shuffles).
template <typename Idx, int VEC, int G>
__device__ __forceinline__ void sddmm_dot_edges(const SddmmArgs<Idx>& a, const int64_t (&u)[kUn], const int64_t (&v)[kUn],
                                                const int64_t (&e)[kUn], int l) {
  typedef typename VecT<VEC>::type V;
  const int64_t RS = a.reduce_size;
  for (int64_t k = 0; k < a.out_len; ++k) {
    const int64_t lo = a.l_off ? a.l_off[k] : (a.l_len == a.out_len * RS ? k : k / (a.out_len * RS / a.l_len));
    const int64_t ro = a.r_off ? a.r_off[k] : (a.r_len == a.out_len * RS ? k : k / (a.out_len * RS / a.r_len));
    float acc[kUn];
#pragma unroll
    for (int i = 0; i < kUn; ++i) acc[i] = 0.f;
    for (int64_t j = (int64_t)l * VEC; j < RS; j += G * VEC) {  // usually one trip
      V x[kUn], y[kUn];
#pragma unroll
      for (int i = 0; i < kUn; ++i) {
        x[i] = (V)(0.f);
        y[i] = (V)(0.f);
        if (e[i] >= 0) {
          x[i] = *reinterpret_cast<const V*>(a.L + pick_target(a.lhs_target, u[i], e[i], v[i]) * a.l_len + lo * RS + j);
          y[i] = *reinterpret_cast<const V*>(a.R + pick_target(a.rhs_target, u[i], e[i], v[i]) * a.r_len + ro * RS + j);
        }
      }
#pragma unroll
      for (int i = 0; i < kUn; ++i) {
        const V pr = x[i] * y[i];
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[i] += ((const float*)&pr)[c];
      }
    }
#pragma unroll
    for (int i = 0; i < kUn; ++i) acc[i] = lanes_sum<G>(acc[i]);
#pragma unroll
    for (int i = 0; i < kUn; ++i)
      if (e[i] >= 0 && l == 0) a.out[e[i] * a.out_len + k] = acc[i];
  }
}

template <typename Idx, int VEC, int G, bool DIRECT, bool DOT>
__global__ __launch_bounds__(kBlock) void sddmm_coo_kernel(const SddmmArgs<Idx> a) {
  constexpr int NB = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int sub = lane / G, l = lane % G;
  const int kc = (blockIdx.y * G + l) * VEC;
  const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int chunk = a.coo_chunk, nchunks = a.coo_chunks;
  const int64_t e0 = wave_id * ((int64_t)chunk * nchunks);
  if (e0 >= a.nnz) return;
  const bool need_u = a.lhs_target == MGX_TARGET_U || a.rhs_target == MGX_TARGET_U;
  const bool need_v = a.lhs_target == MGX_TARGET_V || a.rhs_target == MGX_TARGET_V;
  auto load_ids = [&](int64_t base, Idx& mu, Idx& mv) {
    const int64_t q = base + lane;
    const bool in = lane < chunk && q < a.nnz;
    mu = (need_u && in) ? __builtin_nontemporal_load(&a.src[q]) : (Idx)0;  // id streams: read once
    mv = (need_v && in) ? __builtin_nontemporal_load(&a.dst[q]) : (Idx)0;
  };
  Idx mu, mv;
  load_ids(e0, mu, mv);
  for (int c = 0; c < nchunks; ++c) {
    const int64_t base = e0 + (int64_t)c * chunk;
    if (base >= a.nnz) break;  // wave-uniform
    Idx nu = 0, nv = 0;
    if (c + 1 < nchunks) load_ids(base + chunk, nu, nv);
    const int cnt = (int)((a.nnz - base) < chunk ? (a.nnz - base) : chunk);
    for (int k = 0; k < cnt; k += NB * kUn) {  // wave-uniform trip count
      int64_t u[kUn], v[kUn], e[kUn];
#pragma unroll
      for (int i = 0; i < kUn; ++i) {
        const int j = k + i * NB + sub;
        u[i] = (int64_t)__shfl(mu, j & (kWave - 1), kWave);
        v[i] = (int64_t)__shfl(mv, j & (kWave - 1), kWave);
        e[i] = j < cnt ? base + j : -1;
      }
      if (DOT) sddmm_dot_edges<Idx, VEC, G>(a, u, v, e, l);
      else sddmm_edges<Idx, VEC, G, DIRECT>(a, u, v, e, kc);
    }
    mu = nu;
    mv = nv;
  }
}

// Lean COO walk (element-wise ops, both operands as wide as the output, int32 ids, operands addressable with 32-bit byte
// offsets): the same edge-parallel layout as sddmm_coo_kernel, but the lane that loaded an edge's ids turns them into the BYTE
// OFFSETS of the two operand rows once, the lane groups receive offsets through ds_bpermute and address
// `base (SGPR pair) + offset (one VGPR)` -- no 64-bit multiply per gather -- the output row of edge j of a chunk is
// `chunk base (SGPR pair) + j * rowbytes`, and the gathers of step k+1 are issued before step k is combined and stored.
// OPS: 0 = both operands, 1 = lhs only (copy_lhs), 2 = rhs only (copy_rhs).
// PERM (round 5): the edge list is walked in an order of its own -- the in-CSR's, i.e. sorted by destination, where one operand row of
// consecutive edges is the same row and the other has the g-SpMM's locality -- and `perm[q]` names the output row (the edge id) of the
// q-th walked edge: whole output rows are scattered instead of streamed (profiles/r05_sddmm_perm.txt).
template <int VEC, int G, int OPS, bool PERM = false>
__global__ __launch_bounds__(kBlock) void sddmm_coo32_kernel(const SddmmArgs<int32_t> a) {
  typedef typename VecT<VEC>::type V;
  constexpr int NB = kWave / G;
  constexpr int U = 2;
  constexpr int STEP = NB * U;
  const int lane = threadIdx.x & (kWave - 1);
  const int sub = lane / G, l = lane % G;
  const int kc = (blockIdx.y * G + l) * VEC;
  const bool kactive = kc < a.out_len;
  const uint32_t kc4 = kactive ? (uint32_t)kc * 4u : 0u;  // idle feature lanes re-read column 0 and store nothing
  const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int chunk = a.coo_chunk, nchunks = a.coo_chunks;
  const int64_t e0 = wave_id * ((int64_t)chunk * nchunks);
  if (e0 >= a.nnz) return;
  const uint32_t rowbytes = (uint32_t)a.out_len * 4u;
  const bool lu = a.lhs_target == MGX_TARGET_U, ru = a.rhs_target == MGX_TARGET_U;
  const bool need_u = (OPS != 2 && lu) || (OPS != 1 && ru);
  const bool need_v = (OPS != 2 && !lu) || (OPS != 1 && !ru);
  const char* __restrict__ Lb = reinterpret_cast<const char*>(a.L);
  const char* __restrict__ Rb = reinterpret_cast<const char*>(a.R);
  const int op = a.op;
  auto load_ids = [&](int64_t base, uint32_t& lo, uint32_t& ro, int32_t& po) {
    const int64_t q = base + lane;
    const bool in = lane < chunk && q < a.nnz;
    const int32_t mu = (need_u && in) ? __builtin_nontemporal_load(&a.src[q]) : 0;  // id streams: read once
    const int32_t mv = (need_v && in) ? __builtin_nontemporal_load(&a.dst[q]) : 0;
    lo = (uint32_t)(lu ? mu : mv) * rowbytes;
    ro = (uint32_t)(ru ? mu : mv) * rowbytes;
    po = (PERM && in) ? __builtin_nontemporal_load(&a.perm[q]) : 0;
  };
  uint32_t lo, ro;
  int32_t po;
  load_ids(e0, lo, ro, po);
  char* __restrict__ outb = reinterpret_cast<char*>(a.out);
  for (int c = 0; c < nchunks; ++c) {
    const int64_t base = e0 + (int64_t)c * chunk;
    if (base >= a.nnz) break;  // wave-uniform
    uint32_t nlo = 0, nro = 0;
    int32_t npo = 0;
    if (c + 1 < nchunks) load_ids(base + chunk, nlo, nro, npo);
    const int cnt = (int)((a.nnz - base) < chunk ? (a.nnz - base) : chunk);
    char* __restrict__ ob = reinterpret_cast<char*>(a.out + base * a.out_len);
    auto issue = [&](int k, V (&lv)[U], V (&rv)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = k + u * NB + sub;
        const int bi = (j < cnt ? j : 0) * 4;  // lanes past the end re-read edge 0 of the chunk (valid memory) and store nothing
        if (OPS != 2) lv[u] = *reinterpret_cast<const V*>(Lb + ((uint32_t)__builtin_amdgcn_ds_bpermute(bi, (int)lo) + kc4));
        if (OPS != 1) rv[u] = *reinterpret_cast<const V*>(Rb + ((uint32_t)__builtin_amdgcn_ds_bpermute(bi, (int)ro) + kc4));
      }
    };
    auto finish = [&](int k, const V (&lv)[U], const V (&rv)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = k + u * NB + sub;
        // (the cross-lane read stays OUTSIDE the branch below: ds_bpermute returns nothing useful from a lane that is switched off)
        const uint32_t pe = PERM ? (uint32_t)__builtin_amdgcn_ds_bpermute((j < cnt ? j : 0) * 4, po) : 0u;
        if (j < cnt && kactive) {
          const V val = OPS == 1 ? lv[u] : (OPS == 2 ? rv[u] : sddmm_op<V>(op, lv[u], rv[u]));
          V* o;
          if (PERM) {  // the edge's own output row: 64-bit offset (E x D x 4 bytes may pass 4 GiB)
            o = reinterpret_cast<V*>(outb + ((uint64_t)pe * rowbytes + kc4));
          } else {
            o = reinterpret_cast<V*>(ob + ((uint32_t)j * rowbytes + kc4));
          }
          // rows of >= 32 bytes: the E x D output is written once and not re-read here -- stream it past L2
          if (VEC * G >= 8) __builtin_nontemporal_store(val, o);
          else *o = val;
        }
      }
    };
    V la[U], ra[U], lb[U], rb[U];
    issue(0, la, ra);
    int k = STEP;
    for (;;) {
      if (k >= cnt) { finish(k - STEP, la, ra); break; }
      issue(k, lb, rb);
      finish(k - STEP, la, ra);
      k += STEP;
      if (k >= cnt) { finish(k - STEP, lb, rb); break; }
      issue(k, la, ra);
      finish(k - STEP, lb, rb);
      k += STEP;
    }
    lo = nlo;
    ro = nro;
    po = npo;
  }
}

template <typename Idx, int VEC, int G, bool DIRECT, bool DOT>
__global__ __launch_bounds__(kBlock) void sddmm_csr_kernel(const SddmmArgs<Idx> a) {
  constexpr int NB = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int sub = lane / G, l = lane % G;
  const int kc = (blockIdx.y * G + l) * VEC;
  const int64_t item_base = xcd_remap(blockIdx.x, a.nblocks) * kCsrItems;
  for (int r = wave; r < kCsrItems; r += kWavesPerBlock) {
    const int64_t item = item_base + r;
    if (item >= a.n_items) break;
    int64_t row, beg, end;
    if (a.item_node) {
      row = (int64_t)a.item_node[item];
      beg = (int64_t)a.item_beg[item];
      end = (int64_t)a.item_end[item];
    } else {
      row = item;
      beg = (int64_t)a.indptr[item];
      end = (int64_t)a.indptr[item + 1];
    }
    auto load_ids = [&](int64_t base, Idx& mu, Idx& me) {
      const int64_t q = base + lane;
      mu = q < end ? __builtin_nontemporal_load(&a.indices[q]) : (Idx)0;
      me = q < end ? (a.eids ? __builtin_nontemporal_load(&a.eids[q]) : (Idx)q) : (Idx)0;
    };
    Idx mu = 0, me = 0;
    if (beg < end) load_ids(beg, mu, me);
    for (int64_t base = beg; base < end; base += kWave) {
      Idx nu = 0, ne = 0;
      if (base + kWave < end) load_ids(base + kWave, nu, ne);
      const int cnt = (int)((end - base) < kWave ? (end - base) : kWave);
      for (int k = 0; k < cnt; k += NB * kUn) {
        int64_t u[kUn], v[kUn], e[kUn];
#pragma unroll
        for (int i = 0; i < kUn; ++i) {
          const int j = k + i * NB + sub;
          u[i] = (int64_t)__shfl(mu, j & (kWave - 1), kWave);
          const int64_t ee = (int64_t)__shfl(me, j & (kWave - 1), kWave);
          e[i] = j < cnt ? ee : -1;
          v[i] = row;
        }
        if (DOT) sddmm_dot_edges<Idx, VEC, G>(a, u, v, e, l);
        else sddmm_edges<Idx, VEC, G, DIRECT>(a, u, v, e, kc);
      }
      mu = nu;
      me = ne;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Head-wise dot on the in-CSR with one operand constant along the row: out[e, h] = <gat[indices[p], h, :], rowc[row, h, :]>
// -- the edge gradient of u_mul_e/sum (GATConv backward: X[u] . dZ[v] per head) and u_dot_v on a CSR-only graph.
// Same shape as the summing g-SpMM: lanes along the WHOLE feature row (G lanes x 16 bytes, all heads at once), the
// row-constant operand is loaded once per work item, lane j turns its neighbour id into a 32-bit byte offset once per 64
// edges and hands it over with ds_bpermute, 64/G edges per gather instruction and kUn of them in flight; the F/4 lanes of
// a head combine by xor-shuffle.  (The generic body re-gathers a 4*F-byte slice per head and reloads rowc for every edge.)
// RAGGED (one head, F % 4 != 0, e.g. 41 classes): the lane that owns the last 1-3 columns loads the LAST four floats of both
// rows (dword-aligned 16-byte accesses, nothing read past a row) and zeroes the overlapping components of the row-constant
// operand once per work item, so the overlap contributes nothing to the dot product.
template <int G, int LPH, bool RAGGED = false>
__global__ __launch_bounds__(kBlock) void sddmm_csr_headdot_kernel(const SddmmArgs<int32_t> a, const float* gat, const float* rowc) {
  typedef v4f v4u __attribute__((aligned(4)));
  typedef typename std::conditional<RAGGED, v4u, v4f>::type V4;
  constexpr int NB = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int sub = lane / G, l = lane % G;
  const int D = (int)a.l_len, H = (int)a.out_len;
  const int f = l * 4;
  const bool factive = f < D;
  const bool writer = factive && (l % LPH) == 0;
  const int head = l / LPH;
  const uint32_t rowbytes = (uint32_t)D * 4u;
  const int nvalid = RAGGED ? (D - f < 4 ? D - f : 4) : 4;
  const bool tail = RAGGED && factive && nvalid < 4;
  const int fw = tail ? D - 4 : f;                         // first column of this lane's 16-byte window
  const uint32_t f4 = factive ? (uint32_t)fw * 4u : 0u;    // idle feature lanes re-read the row start; never stored
  const char* __restrict__ gatb = reinterpret_cast<const char*>(gat);
  const int64_t item_base = xcd_remap(blockIdx.x, a.nblocks) * kCsrItems;
  for (int r = wave; r < kCsrItems; r += kWavesPerBlock) {
    const int64_t item = item_base + r;
    if (item >= a.n_items) break;
    int64_t row;
    int32_t beg, end;
    if (a.item_node) {
      row = (int64_t)a.item_node[item];
      beg = a.item_beg[item];
      end = a.item_end[item];
    } else {
      row = item;
      beg = a.indptr[item];
      end = a.indptr[item + 1];
    }
    if (beg >= end) continue;
    v4f rv = factive ? (v4f)*reinterpret_cast<const V4*>(rowc + row * D + fw) : (v4f)(0.f);
    if (tail) {  // components that belong to the previous lane's columns
      if (nvalid < 4) rv.x = 0.f;
      if (nvalid < 3) rv.y = 0.f;
      if (nvalid < 2) rv.z = 0.f;
    }
    auto load_ids = [&](int32_t base, uint32_t& goff, int32_t& me) {
      const int32_t q = base + lane;
      goff = 0;
      me = 0;
      if (q < end) {
        goff = (uint32_t)__builtin_nontemporal_load(&a.indices[q]) * rowbytes;
        me = a.eids ? __builtin_nontemporal_load(&a.eids[q]) : q;
      }
    };
    uint32_t goff, ngoff = 0;
    int32_t me, nme = 0;
    load_ids(beg, goff, me);
    for (int32_t base = beg; base < end; base += kWave) {
      if (base + kWave < end) load_ids(base + kWave, ngoff, nme);
      const int cnt = (end - base) < kWave ? (end - base) : kWave;
      for (int k = 0; k < cnt; k += NB * kUn) {
        v4f x[kUn];
        int32_t ee[kUn];
#pragma unroll
        for (int i = 0; i < kUn; ++i) {
          const int j = k + i * NB + sub;
          const int bi = (j < cnt ? j : 0) * 4;  // lanes past the end re-read edge 0 (valid memory), never stored
          const uint32_t off = (uint32_t)__builtin_amdgcn_ds_bpermute(bi, (int)goff) + f4;
          ee[i] = __builtin_amdgcn_ds_bpermute(bi, me);
          x[i] = *reinterpret_cast<const V4*>(gatb + off);
        }
#pragma unroll
        for (int i = 0; i < kUn; ++i) {
          const float p = lanes_sum<LPH>(x[i].x * rv.x + x[i].y * rv.y + x[i].z * rv.z + x[i].w * rv.w);
          if (writer && k + i * NB + sub < cnt) a.out[(int64_t)ee[i] * H + head] = p;
        }
      }
      goff = ngoff;
      me = nme;
    }
  }
}

template <int G>
static bool launch_headdot_lph(const SddmmArgs<int32_t>& a, const float* gat, const float* rowc, int lph, bool ragged, hipStream_t s) {
  const dim3 grid((unsigned)a.nblocks), block(kBlock);
  if (ragged) {  // one head: the whole lane group reduces
    if (G > 16) return false;
    hipLaunchKernelGGL((sddmm_csr_headdot_kernel<G, (G <= 16 ? G : 16), true>), grid, block, 0, s, a, gat, rowc);
    return true;
  }
  switch (lph) {
#define MGX_HDOT(L) case L: if (L <= G) { hipLaunchKernelGGL((sddmm_csr_headdot_kernel<G, (L <= G ? L : G)>), grid, block, 0, s, a, gat, rowc); return true; } return false;
    MGX_HDOT(1) MGX_HDOT(2) MGX_HDOT(4) MGX_HDOT(8) MGX_HDOT(16)
#undef MGX_HDOT
    default: return false;
  }
}

// Returns true when the specialised kernel was launched.
static bool try_headdot(const SddmmArgs<int32_t>& a, hipStream_t s) {
  const int64_t RS = a.reduce_size, D = a.out_len * RS;
  const bool uv = a.lhs_target == MGX_TARGET_U && a.rhs_target == MGX_TARGET_V;
  const bool vu = a.lhs_target == MGX_TARGET_V && a.rhs_target == MGX_TARGET_U;
  if (!(uv || vu) || a.l_off || a.r_off || a.l_len != D || a.r_len != D || D > 256) return false;
  const bool ragged = RS % 4 != 0;
  const int64_t lph = RS / 4;
  if (ragged) {
    if (a.out_len != 1 || D <= 4 || D > 64 || (uintptr_t)a.L % 4 || (uintptr_t)a.R % 4) return false;
  } else {
    if ((lph & (lph - 1)) != 0 || lph > 16) return false;
    if ((uintptr_t)a.L % 16 || (uintptr_t)a.R % 16) return false;
  }
  if (a.n_cols * D * 4 >= (int64_t(1) << 32) || a.nnz >= (int64_t(1) << 31)) return false;  // 32-bit byte offsets
  const float* gat = uv ? a.L : a.R;
  const float* rowc = uv ? a.R : a.L;
  int G = 1;
  while (G * 4 < D) G <<= 1;
  switch (G) {
    case 1: return launch_headdot_lph<1>(a, gat, rowc, (int)lph, ragged, s);
    case 2: return launch_headdot_lph<2>(a, gat, rowc, (int)lph, ragged, s);
    case 4: return launch_headdot_lph<4>(a, gat, rowc, (int)lph, ragged, s);
    case 8: return launch_headdot_lph<8>(a, gat, rowc, (int)lph, ragged, s);
    case 16: return launch_headdot_lph<16>(a, gat, rowc, (int)lph, ragged, s);
    case 32: return launch_headdot_lph<32>(a, gat, rowc, (int)lph, ragged, s);
    default: return launch_headdot_lph<64>(a, gat, rowc, (int)lph, ragged, s);
  }
}
template <typename Idx> static bool try_headdot_any(const SddmmArgs<Idx>&, hipStream_t) { return false; }
template <> bool try_headdot_any<int32_t>(const SddmmArgs<int32_t>& a, hipStream_t s) {
  return try_headdot(a, s);
}

// int32 COO graphs with element-wise DIRECT operands on node targets, each addressable with 32-bit byte offsets
template <int VEC, int G, bool ELIGIBLE>
static bool launch_coo32(const SddmmArgs<int32_t>& a, dim3 grid, hipStream_t s) {
  if (!ELIGIBLE) return false;
  auto rows_of = [&](int t) -> int64_t { return t == MGX_TARGET_U ? a.n_cols : (t == MGX_TARGET_V ? a.n_rows : -1); };
  const int64_t lim = int64_t(1) << 32;
  if (a.L && (rows_of(a.lhs_target) <= 0 || rows_of(a.lhs_target) * a.out_len * 4 >= lim)) return false;
  if (a.R && (rows_of(a.rhs_target) <= 0 || rows_of(a.rhs_target) * a.out_len * 4 >= lim)) return false;
  if (a.nnz >= (int64_t(1) << 31)) return false;
  if (a.perm) {
    if (a.L && a.R) hipLaunchKernelGGL((sddmm_coo32_kernel<VEC, G, 0, true>), grid, dim3(kBlock), 0, s, a);
    else if (a.L) hipLaunchKernelGGL((sddmm_coo32_kernel<VEC, G, 1, true>), grid, dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((sddmm_coo32_kernel<VEC, G, 2, true>), grid, dim3(kBlock), 0, s, a);
    return true;
  }
  if (a.L && a.R) hipLaunchKernelGGL((sddmm_coo32_kernel<VEC, G, 0>), grid, dim3(kBlock), 0, s, a);
  else if (a.L) hipLaunchKernelGGL((sddmm_coo32_kernel<VEC, G, 1>), grid, dim3(kBlock), 0, s, a);
  else hipLaunchKernelGGL((sddmm_coo32_kernel<VEC, G, 2>), grid, dim3(kBlock), 0, s, a);
  return true;
}
template <int VEC, int G, bool ELIGIBLE>
static bool launch_coo32(const SddmmArgs<int64_t>&, dim3, hipStream_t) { return false; }

template <typename Idx, int VEC, int G, bool CSR, bool DIRECT, bool DOT>
static void launch_one(const SddmmArgs<Idx>& a, hipStream_t s) {
  const unsigned gy = DOT ? 1u : (unsigned)((a.out_len + G * VEC - 1) / (G * VEC));
  if (CSR) {
    hipLaunchKernelGGL((sddmm_csr_kernel<Idx, VEC, G, DIRECT, DOT>), dim3((unsigned)a.nblocks, gy), dim3(kBlock), 0, s, a);
  } else {
    // a wave's share of the edge list: 512 edges on big graphs; on small ones (a batch of molecules: 14 k edges at D = 256
    // ran on 27 waves, each a serial chain of 128 dependent gathers = 262 us) halved until ~kCooTargetWaves waves exist,
    // but never below one fully unrolled step of the lane layout
    SddmmArgs<Idx> b = a;
    constexpr int kMinShare = (kWave / G) * kUn;
    int64_t share = (int64_t)kWave * kCooChunks;
    while (share > kMinShare && (a.nnz + share - 1) / share < kCooTargetWaves) share >>= 1;
    b.coo_chunk = share < kWave ? (int)share : kWave;
    b.coo_chunks = share < kWave ? 1 : (int)(share / kWave);
    const int64_t per_block = (int64_t)kWavesPerBlock * share;
    const int64_t nb = (a.nnz + per_block - 1) / per_block;
    if (launch_coo32<VEC, G, DIRECT && !DOT>(b, dim3((unsigned)nb, gy), s)) return;
    hipLaunchKernelGGL((sddmm_coo_kernel<Idx, VEC, G, DIRECT, DOT>), dim3((unsigned)nb, gy), dim3(kBlock), 0, s, b);
  }
}

template <typename Idx, int VEC, bool CSR, bool DIRECT, bool DOT>
static void launch_g(const SddmmArgs<Idx>& a, hipStream_t s) {
  const int64_t lanes = ((DOT ? a.reduce_size : a.out_len) + VEC - 1) / VEC;
  int G = 1;
  while (G < lanes && G < kWave) G <<= 1;
  // dot: wider reduce dims loop inside a 16-lane group -- 4 edges per wave-instruction and 16 in flight
  // hide the gather latency better than one edge per 64 lanes (reddit GAT, RS = 41: 11.5 -> see profiles)
  if (DOT && G > 16) G = 16;
  switch (G) {
    case 1: launch_one<Idx, VEC, 1, CSR, DIRECT, DOT>(a, s); break;
    case 2: launch_one<Idx, VEC, 2, CSR, DIRECT, DOT>(a, s); break;
    case 4: launch_one<Idx, VEC, 4, CSR, DIRECT, DOT>(a, s); break;
    case 8: launch_one<Idx, VEC, 8, CSR, DIRECT, DOT>(a, s); break;
    case 16: launch_one<Idx, VEC, 16, CSR, DIRECT, DOT>(a, s); break;
    case 32: launch_one<Idx, VEC, 32, CSR, DIRECT, DOT>(a, s); break;
    default: launch_one<Idx, VEC, 64, CSR, DIRECT, DOT>(a, s); break;
  }
}

template <typename Idx, bool CSR>
static int32_t sddmm_impl(SddmmArgs<Idx>& a, hipStream_t s) {
  if (a.nnz == 0 || a.out_len == 0) return MGX_OK;
  if (CSR) {
    a.nblocks = round_up((a.n_items + kCsrItems - 1) / kCsrItems, kXcds);
    MGX_CHECK_ARG(a.nblocks < (int64_t(1) << 31), "mgx_sddmm: too many rows");
  }
  auto aligned = [&](int bytes) {
    return (!a.L || (uintptr_t)a.L % bytes == 0) && (!a.R || (uintptr_t)a.R % bytes == 0) &&
           (uintptr_t)a.out % bytes == 0;
  };
  if (a.op == MGX_OP_DOT) {
    const int64_t RS = a.reduce_size;
    MGX_CHECK_ARG(RS >= 1, "mgx_sddmm: dot needs reduce_size >= 1");
    MGX_CHECK_ARG(a.L && a.R, "mgx_sddmm: dot needs both operands");
    if (CSR && try_headdot_any<Idx>(a, s)) {
      MGX_CHECK_LAUNCH();
      return MGX_OK;
    }
    if (RS % 4 == 0 && a.l_len % 4 == 0 && a.r_len % 4 == 0 && aligned(16)) launch_g<Idx, 4, CSR, true, true>(a, s);
    else if (RS % 2 == 0 && a.l_len % 2 == 0 && a.r_len % 2 == 0 && aligned(8)) launch_g<Idx, 2, CSR, true, true>(a, s);
    else launch_g<Idx, 1, CSR, true, true>(a, s);
    MGX_CHECK_LAUNCH();
    return MGX_OK;
  }
  if (a.op == MGX_OP_COPY_LHS) a.R = nullptr;
  if (a.op == MGX_OP_COPY_RHS) a.L = nullptr;
  const bool direct = !a.l_off && !a.r_off && (!a.L || a.l_len == a.out_len) && (!a.R || a.r_len == a.out_len);
  if (direct) {
    if (a.out_len % 4 == 0 && aligned(16)) launch_g<Idx, 4, CSR, true, false>(a, s);
    else if (a.out_len % 2 == 0 && aligned(8)) launch_g<Idx, 2, CSR, true, false>(a, s);
    else launch_g<Idx, 1, CSR, true, false>(a, s);
  } else {
    launch_g<Idx, 1, CSR, false, false>(a, s);
  }
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

static int32_t check_common(int32_t op, const float* lhs, const float* rhs, int32_t lt, int32_t rt, int64_t l_len,
                            int64_t r_len, int64_t out_len, float* out, int64_t nnz) {
  MGX_CHECK_ARG(op >= MGX_OP_ADD && op <= MGX_OP_DOT, "mgx_sddmm: unsupported binary op %d", op);
  MGX_CHECK_ARG(lt >= MGX_TARGET_U && lt <= MGX_TARGET_V && rt >= MGX_TARGET_U && rt <= MGX_TARGET_V,
                "mgx_sddmm: bad target (%d, %d)", lt, rt);
  MGX_CHECK_ARG(op == MGX_OP_COPY_RHS || lhs != nullptr || nnz == 0, "mgx_sddmm: op needs lhs");
  MGX_CHECK_ARG(op == MGX_OP_COPY_LHS || rhs != nullptr || nnz == 0, "mgx_sddmm: op needs rhs");
  MGX_CHECK_ARG(l_len >= 0 && r_len >= 0 && out_len >= 0, "mgx_sddmm: negative feature length");
  MGX_CHECK_ARG(out != nullptr || nnz == 0 || out_len == 0, "mgx_sddmm: out is NULL");
  return MGX_OK;
}

template <typename Idx>
static int32_t run_coo(int64_t num_src, int64_t num_dst, int64_t nnz, const void* src, const void* dst, int32_t op, const float* lhs, const float* rhs,
                       int32_t lt, int32_t rt, int64_t l_len, int64_t r_len, int64_t out_len, int64_t reduce_size,
                       const int64_t* l_off, const int64_t* r_off, float* out, hipStream_t s, const void* perm = nullptr) {
  SddmmArgs<Idx> a{};
  a.perm = (const Idx*)perm;
  a.src = (const Idx*)src; a.dst = (const Idx*)dst; a.nnz = nnz; a.L = lhs; a.R = rhs;
  a.n_cols = num_src; a.n_rows = num_dst;  // rows of a U- / V-target operand (the lean kernel's 32-bit offsets)
  a.l_off = l_off; a.r_off = r_off; a.out = out; a.l_len = l_len; a.r_len = r_len; a.out_len = out_len;
  a.reduce_size = reduce_size; a.op = op; a.lhs_target = lt; a.rhs_target = rt;
  return sddmm_impl<Idx, false>(a, s);
}

template <typename Idx>
static int32_t run_csr(const mgx_csr* csr, const mgx_spmm_plan* plan, int32_t op, const float* lhs, const float* rhs,
                       int32_t lt, int32_t rt, int64_t l_len, int64_t r_len, int64_t out_len, int64_t reduce_size,
                       const int64_t* l_off, const int64_t* r_off, float* out, hipStream_t s) {
  SddmmArgs<Idx> a{};
  a.indptr = (const Idx*)csr->indptr; a.indices = (const Idx*)csr->indices; a.eids = (const Idx*)csr->eids;
  a.n_rows = csr->num_rows; a.n_cols = csr->num_cols; a.n_items = csr->num_rows; a.nnz = csr->nnz; a.L = lhs; a.R = rhs;
  a.l_off = l_off; a.r_off = r_off; a.out = out; a.l_len = l_len; a.r_len = r_len; a.out_len = out_len;
  a.reduce_size = reduce_size; a.op = op; a.lhs_target = lt; a.rhs_target = rt;
  if (plan && plan->item_node) {
    MGX_CHECK_ARG(plan->item_beg && plan->item_end && plan->num_items >= csr->num_rows, "mgx_sddmm_csr: malformed plan");
    a.item_node = plan->item_node; a.item_beg = (const Idx*)plan->item_beg; a.item_end = (const Idx*)plan->item_end;
    a.n_items = plan->num_items;
  }
  return sddmm_impl<Idx, true>(a, s);
}

}  // namespace mgx

extern "C" int32_t mgx_sddmm_coo(int64_t num_src, int64_t num_dst, int64_t nnz, const void* src, const void* dst,
                                 int32_t idx_bits, int32_t op, const float* lhs, const float* rhs,
                                 int32_t lhs_target, int32_t rhs_target, int64_t l_len, int64_t r_len,
                                 int64_t out_len, int64_t reduce_size, const int64_t* l_off, const int64_t* r_off,
                                 float* out, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(idx_bits == 32 || idx_bits == 64, "mgx_sddmm_coo: idx_bits must be 32 or 64, got %d", idx_bits);
  MGX_CHECK_ARG(nnz >= 0, "mgx_sddmm_coo: negative nnz");
  MGX_CHECK_ARG(nnz == 0 || (src && dst), "mgx_sddmm_coo: src/dst is NULL");
  int32_t st = check_common(op, lhs, rhs, lhs_target, rhs_target, l_len, r_len, out_len, out, nnz);
  if (st != MGX_OK) return st;
  MGX_CHECK_ARG(nnz / (kWavesPerBlock * 4) < (int64_t(1) << 31) - 2, "mgx_sddmm_coo: nnz too large");
  if (idx_bits == 32)
    return run_coo<int32_t>(num_src, num_dst, nnz, src, dst, op, lhs, rhs, lhs_target, rhs_target, l_len, r_len, out_len, reduce_size,
                            l_off, r_off, out, (hipStream_t)stream);
  return run_coo<int64_t>(num_src, num_dst, nnz, src, dst, op, lhs, rhs, lhs_target, rhs_target, l_len, r_len, out_len, reduce_size,
                          l_off, r_off, out, (hipStream_t)stream);
}

// The lean COO walk over an edge list in ANOTHER order than the edge ids (perm[q] = edge id = output row of the q-th walked edge).
// Only what sddmm_coo32_kernel takes: int32 ids, an element-wise op on node-target operands as wide as the output, 32-bit byte offsets.
extern "C" int32_t mgx_sddmm_coo_perm(int64_t num_src, int64_t num_dst, int64_t nnz, const void* src, const void* dst, const void* perm,
                                      int32_t idx_bits, int32_t op, const float* lhs, const float* rhs, int32_t lhs_target,
                                      int32_t rhs_target, int64_t feat_len, float* out, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(nnz >= 0 && feat_len >= 0, "mgx_sddmm_coo_perm: negative sizes");
  MGX_CHECK_ARG(nnz == 0 || (src && dst && perm), "mgx_sddmm_coo_perm: src / dst / perm is NULL");
  int32_t st = check_common(op, lhs, rhs, lhs_target, rhs_target, feat_len, feat_len, feat_len, out, nnz);
  if (st != MGX_OK) return st;
  const int64_t lim = int64_t(1) << 32;
  auto rows_of = [&](int t) -> int64_t { return t == MGX_TARGET_U ? num_src : (t == MGX_TARGET_V ? num_dst : -1); };
  const bool need_l = op != MGX_OP_COPY_RHS, need_r = op != MGX_OP_COPY_LHS;
  if (idx_bits != 32 || op == MGX_OP_DOT || nnz >= (int64_t(1) << 31) ||
      (need_l && (rows_of(lhs_target) <= 0 || rows_of(lhs_target) * feat_len * 4 >= lim)) ||
      (need_r && (rows_of(rhs_target) <= 0 || rows_of(rhs_target) * feat_len * 4 >= lim)))
    MGX_UNSUPPORTED("mgx_sddmm_coo_perm: int32 ids, an element-wise op on u / v operands below 4 GiB only (use mgx_sddmm_coo)");
  return run_coo<int32_t>(num_src, num_dst, nnz, src, dst, op, lhs, rhs, lhs_target, rhs_target, feat_len, feat_len, feat_len, 1, nullptr,
                          nullptr, out, (hipStream_t)stream, perm);
}

extern "C" int32_t mgx_sddmm_csr(const mgx_csr* csr, const mgx_spmm_plan* plan, int32_t op, const float* lhs,
                                 const float* rhs, int32_t lhs_target, int32_t rhs_target, int64_t l_len, int64_t r_len,
                                 int64_t out_len, int64_t reduce_size, const int64_t* l_off, const int64_t* r_off,
                                 float* out, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(csr != nullptr, "mgx_sddmm_csr: csr is NULL");
  MGX_CHECK_ARG(csr->idx_bits == 32 || csr->idx_bits == 64, "mgx_sddmm_csr: idx_bits must be 32 or 64");
  MGX_CHECK_ARG(csr->nnz == 0 || (csr->indptr && csr->indices), "mgx_sddmm_csr: indptr/indices is NULL");
  int32_t st = check_common(op, lhs, rhs, lhs_target, rhs_target, l_len, r_len, out_len, out, csr->nnz);
  if (st != MGX_OK) return st;
  if (csr->idx_bits == 32)
    return run_csr<int32_t>(csr, plan, op, lhs, rhs, lhs_target, rhs_target, l_len, r_len, out_len, reduce_size, l_off,
                            r_off, out, (hipStream_t)stream);
  return run_csr<int64_t>(csr, plan, op, lhs, rhs, lhs_target, rhs_target, l_len, r_len, out_len, reduce_size, l_off,
                          r_off, out, (hipStream_t)stream);
}
