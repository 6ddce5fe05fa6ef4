// sddmm.hip -- g-SDDMM for gfx950 (MI355X).
//
// Replaces what DGL's _CAPI_DGLKernelSDDMM executes for kernel/dgl-new.py:39 (dgl.ops.gsddmm),
// apply_edges(fn.u_add_v) inside GATConv (main_dgl_reddit_gat.py:10), fn.u_dot_v
// (link_prediction/gcmc_dgl/model.py:342) and the per-edge gathers of UDF messages
// (edges.src/edges.dst, main_dgl_molhiv_gcn.py:50-52).  Dense op definitions: kernel/utils.py:8-16.
//
// Edge-parallel, lanes along the feature dimension: a group of G lanes owns one edge and moves
// its feature row with 16-byte accesses; 64/G edges per wave-instruction, UN instructions in
// flight.  Output rows are written contiguously in edge-id order (COO form) -- the E*D*4-byte
// output stream dominates the traffic.  `dot` reduces inside the lane group with xor-shuffles.
#include "common.h"

namespace mgx {

template <typename Idx>
struct SddmmArgs {
  // COO form
  const Idx* src;
  const Idx* dst;
  // CSR form (in-CSR: row = dst)
  const Idx* indptr;
  const Idx* indices;
  const Idx* eids;
  int64_t n_rows;
  int64_t nblocks;
  int64_t nnz;
  const float* L;
  const float* R;
  const int64_t* l_off;
  const int64_t* r_off;
  float* out;
  int64_t l_len, r_len, out_len, reduce_size;
  int op, lhs_target, rhs_target;
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

constexpr int kSddmmUnroll = 4;
constexpr int kSddmmIters = 4;  // COO: each wave owns NB * UN * ITERS consecutive edges

// One edge handled by one lane group: element-wise ops.
// DIRECT: both operands are used as-is (no broadcast), enabling VEC-wide accesses.
template <typename Idx, int VEC, int G, bool CSR, bool DIRECT>
__device__ __forceinline__ void sddmm_edges(const SddmmArgs<Idx>& a, const int64_t (&u)[kSddmmUnroll],
                                            const int64_t (&v)[kSddmmUnroll], const int64_t (&e)[kSddmmUnroll],
                                            int kc) {
  typedef typename VecT<VEC>::type V;
  const bool kactive = kc < a.out_len;
  int64_t lo = kc, ro = kc;
  if (!DIRECT && kactive) {
    lo = !a.L ? 0 : a.l_off ? a.l_off[kc] : (a.l_len == a.out_len ? kc : kc / (a.out_len / a.l_len));
    ro = !a.R ? 0 : a.r_off ? a.r_off[kc] : (a.r_len == a.out_len ? kc : kc / (a.out_len / a.r_len));
  }
  V lv[kSddmmUnroll], rv[kSddmmUnroll];
#pragma unroll
  for (int i = 0; i < kSddmmUnroll; ++i) {
    lv[i] = (V)(0.f);
    rv[i] = (V)(0.f);
    if (e[i] >= 0 && kactive) {
      if (a.L) lv[i] = *reinterpret_cast<const V*>(a.L + pick_target(a.lhs_target, u[i], e[i], v[i]) * a.l_len + lo);
      if (a.R) rv[i] = *reinterpret_cast<const V*>(a.R + pick_target(a.rhs_target, u[i], e[i], v[i]) * a.r_len + ro);
    }
  }
#pragma unroll
  for (int i = 0; i < kSddmmUnroll; ++i)
    if (e[i] >= 0 && kactive) *reinterpret_cast<V*>(a.out + e[i] * a.out_len + kc) = sddmm_op<V>(a.op, lv[i], rv[i]);
}

// COO form, software pipelined like spmm_rowwave_kernel: a wave owns kCooChunks * 64 consecutive edges;
// the (src, dst) ids of 64 edges arrive by ONE coalesced load each (lane j <- edge j) and are handed to the
// lane groups with ds_bpermute; the next 64 ids are requested before the current gathers are issued.
constexpr int kCooChunks = 8;

template <typename Idx, int VEC, int G, bool DIRECT>
__global__ __launch_bounds__(kBlock) void sddmm_coo_kernel(const SddmmArgs<Idx> a) {
  constexpr int NB = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int sub = lane / G, l = lane % G;
  const int kc = (blockIdx.y * G + l) * VEC;
  const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int64_t e0 = wave_id * (kWave * kCooChunks);
  if (e0 >= a.nnz) return;
  const bool need_u = a.lhs_target == MGX_TARGET_U || a.rhs_target == MGX_TARGET_U;
  const bool need_v = a.lhs_target == MGX_TARGET_V || a.rhs_target == MGX_TARGET_V;
  auto load_ids = [&](int64_t base, Idx& mu, Idx& mv) {
    const int64_t q = base + lane;
    mu = (need_u && q < a.nnz) ? a.src[q] : (Idx)0;
    mv = (need_v && q < a.nnz) ? a.dst[q] : (Idx)0;
  };
  Idx mu, mv;
  load_ids(e0, mu, mv);
  for (int c = 0; c < kCooChunks; ++c) {
    const int64_t base = e0 + (int64_t)c * kWave;
    if (base >= a.nnz) break;  // wave-uniform
    Idx nu = 0, nv = 0;
    if (c + 1 < kCooChunks) load_ids(base + kWave, nu, nv);
    const int cnt = (int)((a.nnz - base) < kWave ? (a.nnz - base) : kWave);
    for (int k = 0; k < cnt; k += NB * kSddmmUnroll) {
      int64_t u[kSddmmUnroll], v[kSddmmUnroll], e[kSddmmUnroll];
#pragma unroll
      for (int i = 0; i < kSddmmUnroll; ++i) {
        const int j = k + i * NB + sub;
        u[i] = (int64_t)__shfl(mu, j & (kWave - 1), kWave);
        v[i] = (int64_t)__shfl(mv, j & (kWave - 1), kWave);
        e[i] = j < cnt ? base + j : -1;
      }
      sddmm_edges<Idx, VEC, G, false, DIRECT>(a, u, v, e, kc);
    }
    mu = nu;
    mv = nv;
  }
}

template <typename Idx, int VEC, int G, bool DIRECT>
__global__ __launch_bounds__(kBlock) void sddmm_csr_kernel(const SddmmArgs<Idx> a) {
  constexpr int NB = kWave / G;
  constexpr int kRows = 64;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int sub = lane / G, l = lane % G;
  const int kc = (blockIdx.y * G + l) * VEC;
  const int64_t row_base = xcd_remap(blockIdx.x, a.nblocks) * kRows;
  for (int r = wave; r < kRows; r += kWavesPerBlock) {
    const int64_t row = row_base + r;
    if (row >= a.n_rows) break;
    const int64_t beg = (int64_t)a.indptr[row], end = (int64_t)a.indptr[row + 1];
    for (int64_t p = beg + sub; p < end; p += (int64_t)NB * kSddmmUnroll) {
      int64_t u[kSddmmUnroll], v[kSddmmUnroll], e[kSddmmUnroll];
#pragma unroll
      for (int i = 0; i < kSddmmUnroll; ++i) {
        const int64_t q = p + (int64_t)i * NB;
        const bool ok = q < end;
        e[i] = ok ? (a.eids ? (int64_t)a.eids[q] : q) : -1;
        u[i] = ok ? (int64_t)a.indices[q] : 0;
        v[i] = row;
      }
      sddmm_edges<Idx, VEC, G, true, DIRECT>(a, u, v, e, kc);
    }
  }
}

// dot: out[e,k] = sum_j L[t_l(e), lo(k)*RS + j] * R[t_r(e), ro(k)*RS + j]; G lanes per edge along j.
template <typename Idx, int VEC, int G, bool CSR>
__global__ __launch_bounds__(kBlock) void sddmm_dot_kernel(const SddmmArgs<Idx> a) {
  typedef typename VecT<VEC>::type V;
  constexpr int NB = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int sub = lane / G, l = lane % G;
  const int64_t RS = a.reduce_size;

  auto do_edge = [&](int64_t u, int64_t v, int64_t e) {
    // all lanes of a group share (u,v,e); e < 0 => idle group (still joins the shuffles)
    const int64_t li = pick_target(a.lhs_target, u, e, v), ri = pick_target(a.rhs_target, u, e, v);
    for (int64_t k = 0; k < a.out_len; ++k) {
      const int64_t lo = a.l_off ? a.l_off[k] : (a.l_len == a.out_len * RS ? k : k / (a.out_len * RS / a.l_len));
      const int64_t ro = a.r_off ? a.r_off[k] : (a.r_len == a.out_len * RS ? k : k / (a.out_len * RS / a.r_len));
      float acc = 0.f;
      if (e >= 0) {
        const float* lp = a.L + li * a.l_len + lo * RS;
        const float* rp = a.R + ri * a.r_len + ro * RS;
        for (int64_t j = (int64_t)l * VEC; j < RS; j += G * VEC) {
          const V x = *reinterpret_cast<const V*>(lp + j);
          const V y = *reinterpret_cast<const V*>(rp + j);
          const V pr = x * y;
          if (VEC == 1) acc += ((const float*)&pr)[0];
          else
            for (int c = 0; c < VEC; ++c) acc += ((const float*)&pr)[c];
        }
      }
#pragma unroll
      for (int off = 1; off < G; off <<= 1) acc += __shfl_xor(acc, off, kWave);
      if (e >= 0 && l == 0) a.out[e * a.out_len + k] = acc;
    }
  };

  if (!CSR) {
    const int64_t wave_id = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    constexpr int kEdgesPerWave = NB * kSddmmUnroll * kSddmmIters;
    const int64_t e0 = wave_id * kEdgesPerWave;
    const bool need_u = a.lhs_target == MGX_TARGET_U || a.rhs_target == MGX_TARGET_U;
    const bool need_v = a.lhs_target == MGX_TARGET_V || a.rhs_target == MGX_TARGET_V;
    for (int it = 0; it < kSddmmUnroll * kSddmmIters; ++it) {
      if (e0 + (int64_t)it * NB >= a.nnz) break;  // wave-uniform
      const int64_t q = e0 + (int64_t)it * NB + sub;
      const bool ok = q < a.nnz;
      do_edge((ok && need_u) ? (int64_t)a.src[q] : 0, (ok && need_v) ? (int64_t)a.dst[q] : 0, ok ? q : -1);
    }
  } else {
    constexpr int kRows = 64;
    const int64_t row_base = xcd_remap(blockIdx.x, a.nblocks) * kRows;
    for (int r = wave; r < kRows; r += kWavesPerBlock) {
      const int64_t row = row_base + r;
      if (row >= a.n_rows) break;
      const int64_t beg = (int64_t)a.indptr[row], end = (int64_t)a.indptr[row + 1];
      for (int64_t p0 = beg; p0 < end; p0 += NB) {  // wave-uniform trip count
        const int64_t q = p0 + sub;
        const bool ok = q < end;
        do_edge(ok ? (int64_t)a.indices[q] : 0, row, ok ? (a.eids ? (int64_t)a.eids[q] : q) : -1);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <typename Idx, int VEC, int G, bool CSR, bool DIRECT>
static void launch_ew(const SddmmArgs<Idx>& a, hipStream_t s) {
  const unsigned gy = (unsigned)((a.out_len + G * VEC - 1) / (G * VEC));
  if (CSR) {
    hipLaunchKernelGGL((sddmm_csr_kernel<Idx, VEC, G, DIRECT>), dim3((unsigned)a.nblocks, gy), dim3(kBlock), 0, s, a);
  } else {
    const int64_t per_block = (int64_t)kWavesPerBlock * kWave * kCooChunks;
    const int64_t nb = (a.nnz + per_block - 1) / per_block;
    hipLaunchKernelGGL((sddmm_coo_kernel<Idx, VEC, G, DIRECT>), dim3((unsigned)nb, gy), dim3(kBlock), 0, s, a);
  }
}

template <typename Idx, int VEC, bool CSR, bool DIRECT>
static void launch_ew_g(const SddmmArgs<Idx>& a, hipStream_t s) {
  const int64_t lanes = (a.out_len + VEC - 1) / VEC;
  int G = 1;
  while (G < lanes && G < kWave) G <<= 1;
  switch (G) {
    case 1: launch_ew<Idx, VEC, 1, CSR, DIRECT>(a, s); break;
    case 2: launch_ew<Idx, VEC, 2, CSR, DIRECT>(a, s); break;
    case 4: launch_ew<Idx, VEC, 4, CSR, DIRECT>(a, s); break;
    case 8: launch_ew<Idx, VEC, 8, CSR, DIRECT>(a, s); break;
    case 16: launch_ew<Idx, VEC, 16, CSR, DIRECT>(a, s); break;
    case 32: launch_ew<Idx, VEC, 32, CSR, DIRECT>(a, s); break;
    default: launch_ew<Idx, VEC, 64, CSR, DIRECT>(a, s); break;
  }
}

template <typename Idx, int VEC, bool CSR>
static void launch_dot_g(const SddmmArgs<Idx>& a, hipStream_t s) {
  const int64_t lanes = (a.reduce_size + VEC - 1) / VEC;
  int G = 1;
  while (G < lanes && G < kWave) G <<= 1;
  auto grid_for = [&](int NB) {
    if (CSR) return dim3((unsigned)a.nblocks);
    const int64_t per_block = (int64_t)kWavesPerBlock * NB * kSddmmUnroll * kSddmmIters;
    return dim3((unsigned)((a.nnz + per_block - 1) / per_block));
  };
#define MGX_DOT_CASE(GG) \
  case GG: hipLaunchKernelGGL((sddmm_dot_kernel<Idx, VEC, GG, CSR>), grid_for(kWave / GG), dim3(kBlock), 0, s, a); break;
  switch (G) {
    MGX_DOT_CASE(1) MGX_DOT_CASE(2) MGX_DOT_CASE(4) MGX_DOT_CASE(8) MGX_DOT_CASE(16) MGX_DOT_CASE(32)
    default: hipLaunchKernelGGL((sddmm_dot_kernel<Idx, VEC, 64, CSR>), grid_for(1), dim3(kBlock), 0, s, a); break;
  }
#undef MGX_DOT_CASE
}

template <typename Idx, bool CSR>
static int32_t sddmm_impl(SddmmArgs<Idx>& a, hipStream_t s) {
  if (a.nnz == 0 || a.out_len == 0) return MGX_OK;
  if (CSR) {
    a.nblocks = round_up((a.n_rows + 63) / 64, kXcds);
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
    if (RS % 4 == 0 && a.l_len % 4 == 0 && a.r_len % 4 == 0 && aligned(16)) launch_dot_g<Idx, 4, CSR>(a, s);
    else if (RS % 2 == 0 && a.l_len % 2 == 0 && a.r_len % 2 == 0 && aligned(8)) launch_dot_g<Idx, 2, CSR>(a, s);
    else launch_dot_g<Idx, 1, CSR>(a, s);
    MGX_CHECK_LAUNCH();
    return MGX_OK;
  }
  if (a.op == MGX_OP_COPY_LHS) a.R = nullptr;
  if (a.op == MGX_OP_COPY_RHS) a.L = nullptr;
  const bool direct = !a.l_off && !a.r_off && (!a.L || a.l_len == a.out_len) && (!a.R || a.r_len == a.out_len);
  if (direct) {
    if (a.out_len % 4 == 0 && aligned(16)) launch_ew_g<Idx, 4, CSR, true>(a, s);
    else if (a.out_len % 2 == 0 && aligned(8)) launch_ew_g<Idx, 2, CSR, true>(a, s);
    else launch_ew_g<Idx, 1, CSR, true>(a, s);
  } else {
    launch_ew_g<Idx, 1, CSR, false>(a, s);
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

}  // namespace mgx

extern "C" int32_t mgx_sddmm_coo(int64_t num_src, int64_t num_dst, int64_t nnz, const void* src, const void* dst,
                                 int32_t idx_bits, int32_t op, const float* lhs, const float* rhs,
                                 int32_t lhs_target, int32_t rhs_target, int64_t l_len, int64_t r_len,
                                 int64_t out_len, int64_t reduce_size, const int64_t* l_off, const int64_t* r_off,
                                 float* out, void* stream) {
  using namespace mgx;
  (void)num_src; (void)num_dst;
  MGX_CHECK_ARG(idx_bits == 32 || idx_bits == 64, "mgx_sddmm_coo: idx_bits must be 32 or 64, got %d", idx_bits);
  MGX_CHECK_ARG(nnz >= 0, "mgx_sddmm_coo: negative nnz");
  MGX_CHECK_ARG(nnz == 0 || (src && dst), "mgx_sddmm_coo: src/dst is NULL");
  int32_t st = check_common(op, lhs, rhs, lhs_target, rhs_target, l_len, r_len, out_len, out, nnz);
  if (st != MGX_OK) return st;
  const int64_t max_edges_per_block = 4 * 64 * kSddmmUnroll * kSddmmIters;
  MGX_CHECK_ARG(nnz / 64 < (int64_t(1) << 31) - max_edges_per_block, "mgx_sddmm_coo: nnz too large");
  if (idx_bits == 32) {
    SddmmArgs<int32_t> a{};
    a.src = (const int32_t*)src; a.dst = (const int32_t*)dst; a.nnz = nnz; a.L = lhs; a.R = rhs;
    a.l_off = l_off; a.r_off = r_off; a.out = out; a.l_len = l_len; a.r_len = r_len; a.out_len = out_len;
    a.reduce_size = reduce_size; a.op = op; a.lhs_target = lhs_target; a.rhs_target = rhs_target;
    return sddmm_impl<int32_t, false>(a, (hipStream_t)stream);
  }
  SddmmArgs<int64_t> a{};
  a.src = (const int64_t*)src; a.dst = (const int64_t*)dst; a.nnz = nnz; a.L = lhs; a.R = rhs;
  a.l_off = l_off; a.r_off = r_off; a.out = out; a.l_len = l_len; a.r_len = r_len; a.out_len = out_len;
  a.reduce_size = reduce_size; a.op = op; a.lhs_target = lhs_target; a.rhs_target = rhs_target;
  return sddmm_impl<int64_t, false>(a, (hipStream_t)stream);
}

extern "C" int32_t mgx_sddmm_csr(const mgx_csr* csr, int32_t op, const float* lhs, const float* rhs,
                                 int32_t lhs_target, int32_t rhs_target, int64_t l_len, int64_t r_len,
                                 int64_t out_len, int64_t reduce_size, const int64_t* l_off, const int64_t* r_off,
                                 float* out, void* stream) {
  using namespace mgx;
  MGX_CHECK_ARG(csr != nullptr, "mgx_sddmm_csr: csr is NULL");
  MGX_CHECK_ARG(csr->idx_bits == 32 || csr->idx_bits == 64, "mgx_sddmm_csr: idx_bits must be 32 or 64");
  MGX_CHECK_ARG(csr->nnz == 0 || (csr->indptr && csr->indices), "mgx_sddmm_csr: indptr/indices is NULL");
  int32_t st = check_common(op, lhs, rhs, lhs_target, rhs_target, l_len, r_len, out_len, out, csr->nnz);
  if (st != MGX_OK) return st;
  if (csr->idx_bits == 32) {
    SddmmArgs<int32_t> a{};
    a.indptr = (const int32_t*)csr->indptr; a.indices = (const int32_t*)csr->indices; a.eids = (const int32_t*)csr->eids;
    a.n_rows = csr->num_rows; a.nnz = csr->nnz; a.L = lhs; a.R = rhs; a.l_off = l_off; a.r_off = r_off; a.out = out;
    a.l_len = l_len; a.r_len = r_len; a.out_len = out_len; a.reduce_size = reduce_size; a.op = op;
    a.lhs_target = lhs_target; a.rhs_target = rhs_target;
    return sddmm_impl<int32_t, true>(a, (hipStream_t)stream);
  }
  SddmmArgs<int64_t> a{};
  a.indptr = (const int64_t*)csr->indptr; a.indices = (const int64_t*)csr->indices; a.eids = (const int64_t*)csr->eids;
  a.n_rows = csr->num_rows; a.nnz = csr->nnz; a.L = lhs; a.R = rhs; a.l_off = l_off; a.r_off = r_off; a.out = out;
  a.l_len = l_len; a.r_len = r_len; a.out_len = out_len; a.reduce_size = reduce_size; a.op = op;
  a.lhs_target = lhs_target; a.rhs_target = rhs_target;
  return sddmm_impl<int64_t, true>(a, (hipStream_t)stream);
}
