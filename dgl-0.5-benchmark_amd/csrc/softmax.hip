// softmax.hip -- fused edge-softmax (norm_by='dst') forward/backward for gfx950 (MI355X).
//
// Replaces dgl.nn.functional.edge_softmax as reached through GATConv
// (main_dgl_reddit_gat.py:31-55).  DGL 0.6 composes it from copy_rhs/max SpMM, sub SDDMM + exp,
// copy_rhs/sum SpMM, div SDDMM -- four launches and five E*H-sized round trips.  Here one
// row-segmented kernel: a wavefront owns a destination row, lanes are laid out as
// (edge slot, head), the row's logits are read ONCE into registers when in-degree*LH <= 64*kCache
// (re-read from L2 otherwise), max and sum are combined by xor-shuffles, one write.
//
// Hub rows: with an mgx_spmm_plan, rows longer than the plan's split threshold are processed as
// chunks -- (1) per-chunk statistics (max, sum-exp | sum a*da), (2) a per-hub combine in slot
// order, (3) a per-chunk normalise -- so a 21k-edge reddit hub no longer serialises on one wave.
#include <math.h>

#include "common.h"

namespace mgx {

constexpr int kCache = 4;
constexpr int kSmRows = 16;  // work items per workgroup (same reasoning as spmm.hip)

template <typename Idx>
struct SoftmaxArgs {
  const Idx* indptr;
  const Idx* eids;
  const float* x;   // fwd: z        bwd: a
  const float* y;   // fwd: unused   bwd: da
  float* out;       // fwd: a        bwd: dz
  // schedule (optional)
  const int32_t* item_row;
  const Idx* item_beg;
  const Idx* item_end;
  const int32_t* slot_item;     // [num_slots] work item of each partial slot
  const int32_t* hub_slot_ptr;  // [num_hubs + 1]
  float* partial;               // [num_slots, 2H]  fwd: (max, sumexp)   bwd: (sum a*da, unused)
  float* hubstat;               // [num_hubs, 2H]
  int64_t n_items;
  int64_t n_slots;
  int64_t n_hubs;
  int64_t n_rows;
  int64_t nblocks;
  int H;
  // fused GAT logits (FUSED kernels): z[e,h] = leaky_relu(el[src(e),h] + er[dst(e),h]) is never materialised
  const Idx* indices;
  const int32_t* item_node;
  const float* el;
  const float* er;
  float slope;
};

enum { SM_FULL = 0, SM_STATS = 1, SM_APPLY = 2 };

// One edge range [beg, end) of one destination row, handled by one wave.
//   SM_FULL : statistics + output for a whole row.
//   SM_STATS: statistics of a chunk -> st0/st1 (fwd: max, sum exp(z - max); bwd: sum a*da, -)
//   SM_APPLY: output of a chunk from the row's combined statistics st0/st1.
template <typename Idx, int LH, bool BWD, int MODE, bool FUSED>
__device__ __forceinline__ void softmax_range(const SoftmaxArgs<Idx>& a, int64_t row, int64_t beg, int64_t end, float& st0,
                                              float& st1) {
  constexpr int EPI = kWave / LH;
  const int lane = threadIdx.x & (kWave - 1);
  const int j = lane / LH, h = lane % LH;
  const bool hactive = h < a.H;
  const int64_t H = a.H;
  const float er_v = (FUSED && hactive) ? a.er[row * H + h] : 0.f;
  // logit of the edge at CSR position p (edge id e): read it, or rebuild it from the two node terms
  auto logit = [&](int64_t p, int64_t e) -> float {
    if (!FUSED) return a.x[e * H + h];
    const float t = a.el[(int64_t)a.indices[p] * H + h] + er_v;
    return t > 0.f ? t : t * a.slope;
  };
  // d leaky_relu / d logit at position p (backward of the fused form)
  auto lrelu_grad = [&](int64_t p) -> float {
    if (!FUSED) return 1.f;
    const float t = a.el[(int64_t)a.indices[p] * H + h] + er_v;
    return t > 0.f ? 1.f : a.slope;
  };
  int64_t es[kCache];
  float xs[kCache];
#pragma unroll
  for (int c = 0; c < kCache; ++c) {
    const int64_t p = beg + j + (int64_t)c * EPI;
    const bool ok = p < end && hactive;
    es[c] = ok ? (a.eids ? (int64_t)a.eids[p] : p) : -1;
  }
  const int64_t tail = beg + j + (int64_t)kCache * EPI;
  if (!BWD) {
    float m = st0, s = st1;
#pragma unroll
    for (int c = 0; c < kCache; ++c) xs[c] = es[c] >= 0 ? logit(beg + j + (int64_t)c * EPI, es[c]) : -INFINITY;
    if (MODE != SM_APPLY) {
      float red = -INFINITY;
#pragma unroll
      for (int c = 0; c < kCache; ++c) red = fmaxf(red, xs[c]);
      for (int64_t p = tail; p < end; p += EPI)
        if (hactive) red = fmaxf(red, logit(p, a.eids ? (int64_t)a.eids[p] : p));
#pragma unroll
      for (int off = LH; off < kWave; off <<= 1) red = fmaxf(red, __shfl_xor(red, off, kWave));
      m = red;
      s = 0.f;
#pragma unroll
      for (int c = 0; c < kCache; ++c) {
        xs[c] = es[c] >= 0 ? expf(xs[c] - m) : 0.f;
        s += xs[c];
      }
      for (int64_t p = tail; p < end; p += EPI)
        if (hactive) s += expf(logit(p, a.eids ? (int64_t)a.eids[p] : p) - m);
#pragma unroll
      for (int off = LH; off < kWave; off <<= 1) s += __shfl_xor(s, off, kWave);
      st0 = m;
      st1 = s;
      if (MODE == SM_STATS) return;
    } else {
#pragma unroll
      for (int c = 0; c < kCache; ++c) xs[c] = es[c] >= 0 ? expf(xs[c] - m) : 0.f;
    }
#pragma unroll
    for (int c = 0; c < kCache; ++c)
      if (es[c] >= 0) a.out[es[c] * H + h] = xs[c] / s;
    for (int64_t p = tail; p < end; p += EPI)
      if (hactive) {
        const int64_t e = a.eids ? (int64_t)a.eids[p] : p;
        a.out[e * H + h] = expf(logit(p, e) - m) / s;
      }
  } else {
    float ys[kCache];
#pragma unroll
    for (int c = 0; c < kCache; ++c) {
      xs[c] = es[c] >= 0 ? a.x[es[c] * H + h] : 0.f;  // a
      ys[c] = es[c] >= 0 ? a.y[es[c] * H + h] : 0.f;  // da
    }
    float red = st0;
    if (MODE != SM_APPLY) {
      red = 0.f;
#pragma unroll
      for (int c = 0; c < kCache; ++c) red += xs[c] * ys[c];
      for (int64_t p = tail; p < end; p += EPI)
        if (hactive) {
          const int64_t e = a.eids ? (int64_t)a.eids[p] : p;
          red += a.x[e * H + h] * a.y[e * H + h];
        }
#pragma unroll
      for (int off = LH; off < kWave; off <<= 1) red += __shfl_xor(red, off, kWave);
      st0 = red;
      if (MODE == SM_STATS) return;
    }
#pragma unroll
    for (int c = 0; c < kCache; ++c)
      if (es[c] >= 0) a.out[es[c] * H + h] = (xs[c] * ys[c] - xs[c] * red) * lrelu_grad(beg + j + (int64_t)c * EPI);
    for (int64_t p = tail; p < end; p += EPI)
      if (hactive) {
        const int64_t e = a.eids ? (int64_t)a.eids[p] : p;
        const float av = a.x[e * H + h];
        a.out[e * H + h] = (av * a.y[e * H + h] - av * red) * lrelu_grad(p);
      }
  }
}

// LH = lanes per edge (pow2 >= H); 64/LH edges per step.
template <typename Idx, int LH, bool BWD, bool FUSED>
__global__ __launch_bounds__(kBlock) void edge_softmax_kernel(const SoftmaxArgs<Idx> a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int h = lane % LH;
  const int64_t item_base = xcd_remap(blockIdx.x, a.nblocks) * kSmRows;
  for (int r = wave; r < kSmRows; r += kWavesPerBlock) {
    const int64_t item = item_base + r;
    if (item >= a.n_items) break;
    int64_t row, beg, end;
    if (a.item_row) {
      row = (int64_t)a.item_row[item];
      beg = (int64_t)a.item_beg[item];
      end = (int64_t)a.item_end[item];
    } else {
      row = item;
      beg = (int64_t)a.indptr[item];
      end = (int64_t)a.indptr[item + 1];
    }
    if (beg == end) continue;
    float s0 = 0.f, s1 = 0.f;
    const int64_t node = a.item_node ? (int64_t)a.item_node[item] : row;  // the destination row, also for hub chunks
    if (row >= 0) {
      softmax_range<Idx, LH, BWD, SM_FULL, FUSED>(a, node, beg, end, s0, s1);
    } else {
      softmax_range<Idx, LH, BWD, SM_STATS, FUSED>(a, node, beg, end, s0, s1);
      const int64_t slot = -(row + 1);
      if (lane < LH && h < a.H) {  // lanes of edge slot 0 hold the combined value for every head
        a.partial[slot * 2 * a.H + h] = s0;
        a.partial[slot * 2 * a.H + a.H + h] = s1;
      }
    }
  }
}

// per hub row: combine the chunk statistics in slot order (deterministic); one thread per (hub, head)
template <bool BWD>
__global__ __launch_bounds__(kBlock) void softmax_hub_combine_kernel(const int32_t* hub_slot_ptr, int64_t n_hubs, int H,
                                                                     const float* partial, float* hubstat) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n_hubs * H) return;
  const int64_t hub = t / H;
  const int h = (int)(t % H);
  const int s_beg = hub_slot_ptr[hub], s_end = hub_slot_ptr[hub + 1];
  if (!BWD) {
    float m = -INFINITY;
    for (int s = s_beg; s < s_end; ++s) m = fmaxf(m, partial[(int64_t)s * 2 * H + h]);
    float sum = 0.f;
    for (int s = s_beg; s < s_end; ++s) {
      const float mc = partial[(int64_t)s * 2 * H + h];
      if (mc > -INFINITY) sum += partial[(int64_t)s * 2 * H + H + h] * expf(mc - m);
    }
    hubstat[hub * 2 * H + h] = m;
    hubstat[hub * 2 * H + H + h] = sum;
  } else {
    float acc = 0.f;
    for (int s = s_beg; s < s_end; ++s) acc += partial[(int64_t)s * 2 * H + h];
    hubstat[hub * 2 * H + h] = acc;
    hubstat[hub * 2 * H + H + h] = 0.f;
  }
}

// one wave per partial slot: normalise its chunk with the hub row's combined statistics
template <typename Idx, int LH, bool BWD, bool FUSED>
__global__ __launch_bounds__(kBlock) void softmax_hub_apply_kernel(const SoftmaxArgs<Idx> a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int h = lane % LH;
  const int64_t slot = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  if (slot >= a.n_slots) return;
  int64_t lo = 0, hi = a.n_hubs;  // last hub whose first slot <= slot
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)a.hub_slot_ptr[mid] <= slot) lo = mid; else hi = mid;
  }
  const int64_t item = a.slot_item[slot];
  const int64_t beg = (int64_t)a.item_beg[item], end = (int64_t)a.item_end[item];
  float s0 = 0.f, s1 = 1.f;
  if (h < a.H) {
    s0 = a.hubstat[lo * 2 * a.H + h];
    s1 = a.hubstat[lo * 2 * a.H + a.H + h];
  }
  if (beg < end) softmax_range<Idx, LH, BWD, SM_APPLY, FUSED>(a, (int64_t)a.item_node[item], beg, end, s0, s1);
}

template <typename Idx, bool BWD, bool FUSED>
static int32_t softmax_launch(const mgx_csr* csr, const mgx_spmm_plan* plan, int64_t H, const float* x, const float* y,
                              float* out, float* ws, const float* el, const float* er, float slope, hipStream_t s) {
  if (csr->num_rows == 0 || csr->nnz == 0 || H == 0) return MGX_OK;
  SoftmaxArgs<Idx> a{};
  a.indices = (const Idx*)csr->indices; a.el = el; a.er = er; a.slope = slope;
  if (FUSED) MGX_CHECK_ARG(el && er && csr->indices, "mgx_gat_attention: el / er / indices is NULL");
  a.indptr = (const Idx*)csr->indptr; a.eids = (const Idx*)csr->eids; a.x = x; a.y = y; a.out = out;
  a.n_rows = csr->num_rows; a.H = (int)H; a.n_items = csr->num_rows;
  if (plan) {
    MGX_CHECK_ARG(plan->item_row && plan->item_beg && plan->item_end && plan->num_items >= csr->num_rows,
                  "mgx_edge_softmax: malformed plan");
    a.item_row = plan->item_row; a.item_beg = (const Idx*)plan->item_beg; a.item_end = (const Idx*)plan->item_end;
    a.n_items = plan->num_items; a.n_slots = plan->num_slots; a.n_hubs = plan->num_hubs;
    a.item_node = plan->item_node;
    MGX_CHECK_ARG(plan->num_slots == 0 || plan->item_node, "mgx_edge_softmax: plan lacks item_node");
    if (plan->num_slots > 0) {
      MGX_CHECK_ARG(ws && plan->slot_item && plan->hub_slot_ptr, "mgx_edge_softmax: plan has split rows but no workspace / slot tables");
      a.slot_item = plan->slot_item; a.hub_slot_ptr = plan->hub_slot_ptr;
      a.partial = ws; a.hubstat = ws + plan->num_slots * 2 * H;
    }
  }
  a.nblocks = round_up((a.n_items + kSmRows - 1) / kSmRows, kXcds);
  MGX_CHECK_ARG(a.nblocks < (int64_t(1) << 31), "mgx_edge_softmax: too many rows");
  int LH = 1;
  while (LH < H) LH <<= 1;
  dim3 grid((unsigned)a.nblocks), block(kBlock);
  dim3 sgrid((unsigned)((a.n_slots + kWavesPerBlock - 1) / kWavesPerBlock));
#define MGX_SM_CASE(L)                                                                                      \
  case L:                                                                                                   \
    hipLaunchKernelGGL((edge_softmax_kernel<Idx, L, BWD, FUSED>), grid, block, 0, s, a);                           \
    if (a.n_slots > 0) {                                                                                    \
      hipLaunchKernelGGL((softmax_hub_combine_kernel<BWD>), dim3((unsigned)((a.n_hubs * H + kBlock - 1) / kBlock)), \
                         block, 0, s, a.hub_slot_ptr, a.n_hubs, a.H, (const float*)a.partial, a.hubstat);  \
      hipLaunchKernelGGL((softmax_hub_apply_kernel<Idx, L, BWD, FUSED>), sgrid, block, 0, s, a);                   \
    }                                                                                                       \
    break;
  switch (LH) {
    MGX_SM_CASE(1) MGX_SM_CASE(2) MGX_SM_CASE(4) MGX_SM_CASE(8) MGX_SM_CASE(16) MGX_SM_CASE(32)
    default:
      hipLaunchKernelGGL((edge_softmax_kernel<Idx, 64, BWD, FUSED>), grid, block, 0, s, a);
      if (a.n_slots > 0) {
        hipLaunchKernelGGL((softmax_hub_combine_kernel<BWD>), dim3((unsigned)((a.n_hubs * H + kBlock - 1) / kBlock)),
                           block, 0, s, a.hub_slot_ptr, a.n_hubs, a.H, (const float*)a.partial, a.hubstat);
        hipLaunchKernelGGL((softmax_hub_apply_kernel<Idx, 64, BWD, FUSED>), sgrid, block, 0, s, a);
      }
      break;
  }
#undef MGX_SM_CASE
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

static int32_t softmax_check(const mgx_csr* csr, int64_t H) {
  MGX_CHECK_ARG(csr != nullptr, "mgx_edge_softmax: csr is NULL");
  MGX_CHECK_ARG(csr->idx_bits == 32 || csr->idx_bits == 64, "mgx_edge_softmax: idx_bits must be 32 or 64");
  MGX_CHECK_ARG(H >= 0, "mgx_edge_softmax: negative H");
  if (H > kWave) MGX_UNSUPPORTED("mgx_edge_softmax: H = %lld > 64 heads is not supported", (long long)H);
  MGX_CHECK_ARG(csr->num_rows == 0 || csr->indptr, "mgx_edge_softmax: indptr is NULL");
  return MGX_OK;
}

}  // namespace mgx

extern "C" int32_t mgx_edge_softmax_fwd(const mgx_csr* csr, const mgx_spmm_plan* plan, int64_t H, const float* z,
                                        float* a, float* ws, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  int32_t st = softmax_check(csr, H);
  if (st != MGX_OK) return st;
  MGX_CHECK_ARG(csr->nnz == 0 || H == 0 || (z && a), "mgx_edge_softmax_fwd: z/a is NULL");
  if (csr->idx_bits == 32) return softmax_launch<int32_t, false, false>(csr, plan, H, z, nullptr, a, ws, nullptr, nullptr, 0.f, (hipStream_t)stream);
  return softmax_launch<int64_t, false, false>(csr, plan, H, z, nullptr, a, ws, nullptr, nullptr, 0.f, (hipStream_t)stream);
}

extern "C" int32_t mgx_edge_softmax_bwd(const mgx_csr* csr, const mgx_spmm_plan* plan, int64_t H, const float* a,
                                        const float* da, float* dz, float* ws, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  int32_t st = softmax_check(csr, H);
  if (st != MGX_OK) return st;
  MGX_CHECK_ARG(csr->nnz == 0 || H == 0 || (a && da && dz), "mgx_edge_softmax_bwd: a/da/dz is NULL");
  if (csr->idx_bits == 32) return softmax_launch<int32_t, true, false>(csr, plan, H, a, da, dz, ws, nullptr, nullptr, 0.f, (hipStream_t)stream);
  return softmax_launch<int64_t, true, false>(csr, plan, H, a, da, dz, ws, nullptr, nullptr, 0.f, (hipStream_t)stream);
}

// GAT attention with the logits fused in: a = softmax_{e->v}( leaky_relu(el[u] + er[v]) ), the composition
// apply_edges(fn.u_add_v) -> leaky_relu -> edge_softmax of GATConv (main_dgl_reddit_gat.py:31-55) in one kernel; the
// E x H logit tensor is never written.  el: [num_cols, H], er: [num_rows, H], a: [nnz, H] by edge id.
extern "C" int32_t mgx_gat_attention_fwd(const mgx_csr* csr, const mgx_spmm_plan* plan, int64_t H, const float* el,
                                         const float* er, float negative_slope, float* a, float* ws, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  int32_t st = softmax_check(csr, H);
  if (st != MGX_OK) return st;
  MGX_CHECK_ARG(csr->nnz == 0 || H == 0 || a, "mgx_gat_attention_fwd: a is NULL");
  if (csr->idx_bits == 32)
    return softmax_launch<int32_t, false, true>(csr, plan, H, nullptr, nullptr, a, ws, el, er, negative_slope, (hipStream_t)stream);
  return softmax_launch<int64_t, false, true>(csr, plan, H, nullptr, nullptr, a, ws, el, er, negative_slope, (hipStream_t)stream);
}

// backward of the above w.r.t. the pre-activation logit: de = (a*da - a*sum(a*da)) * leaky_relu'(el[u] + er[v]);
// the caller reduces de over out-edges / in-edges (copy_e g-SpMM) to get d el / d er.
extern "C" int32_t mgx_gat_attention_bwd(const mgx_csr* csr, const mgx_spmm_plan* plan, int64_t H, const float* el,
                                         const float* er, float negative_slope, const float* a, const float* da,
                                         float* de, float* ws, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  int32_t st = softmax_check(csr, H);
  if (st != MGX_OK) return st;
  MGX_CHECK_ARG(csr->nnz == 0 || H == 0 || (a && da && de), "mgx_gat_attention_bwd: a/da/de is NULL");
  if (csr->idx_bits == 32)
    return softmax_launch<int32_t, true, true>(csr, plan, H, a, da, de, ws, el, er, negative_slope, (hipStream_t)stream);
  return softmax_launch<int64_t, true, true>(csr, plan, H, a, da, de, ws, el, er, negative_slope, (hipStream_t)stream);
}
