// cpu_ops.cpp -- CPU (OpenMP) variants of the hot-path entry points (include/mi355x_graph_cpu.h).
//
// What the reference reaches with `--gpu -1` (kernel/dgl-new.py:55-58) and on machines without a GPU
// (main_dgl_product_sage.py:149): g-SpMM, g-SDDMM, edge softmax, segment reduce and the integer format work, on host
// pointers.  Product code of its own -- the checker under oracle/ is never linked or called from here.
//
// Shape of the code: one generic row kernel per primitive, templated on the index width, with the feature loop innermost so that
// the compiler vectorises it; rows (edges for the COO walk) are the OpenMP dimension, static schedule in chunks so that
// neighbouring rows -- which share source rows on graphs with locality -- stay on one core's cache.  Terms of a row are combined
// in storage order: the result is independent of the thread count and equals DGL's CPU kernels' order.
#include <math.h>
#include <omp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <vector>

#include "../../include/mi355x_graph_cpu.h"

namespace {

thread_local char g_err[512] = "";

int32_t fail(int32_t code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define CPU_CHECK(cond, ...) \
  do {                       \
    if (!(cond)) return fail(MGX_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
  } while (0)

struct Lookup {  // element k of the output reads element at(k) of an operand row
  const int64_t* table;
  int64_t len, out_len;
  inline int64_t at(int64_t k) const {
    if (table) return table[k];
    if (len == out_len) return k;
    return len > 0 ? k / (out_len / len) : 0;  // head-wise broadcast: (N, H, F) x (E, H, 1)
  }
};

inline float combine(int op, float l, float r) {
  switch (op) {
    case MGX_OP_ADD: return l + r;
    case MGX_OP_SUB: return l - r;
    case MGX_OP_MUL: return l * r;
    case MGX_OP_DIV: return l / r;
    case MGX_OP_COPY_LHS: return l;
    default: return r;
  }
}

int check_csr(const mgx_csr* c, const char* who) {
  if (!c) return fail(MGX_ERR_INVALID_ARGUMENT, "%s: csr is NULL", who);
  if (c->idx_bits != 32 && c->idx_bits != 64) return fail(MGX_ERR_INVALID_ARGUMENT, "%s: idx_bits must be 32 or 64", who);
  if (c->num_rows < 0 || c->nnz < 0) return fail(MGX_ERR_INVALID_ARGUMENT, "%s: negative sizes", who);
  if (!c->indptr || (c->nnz > 0 && !c->indices)) return fail(MGX_ERR_INVALID_ARGUMENT, "%s: indptr / indices is NULL", who);
  return MGX_OK;
}

// ------------------------------------------------------------------------------------------------ g-SpMM
template <typename Idx>
int32_t spmm_rows(const mgx_csr* csr, int op, int reduce, const float* U, const float* E, int64_t u_len, int64_t e_len, int64_t D,
                  const int64_t* u_off, const int64_t* e_off, const float* src_scale, const float* dst_scale, float* out, Idx* arg_u,
                  Idx* arg_e, bool accumulate) {
  const Idx* indptr = (const Idx*)csr->indptr;
  const Idx* indices = (const Idx*)csr->indices;
  const Idx* eids = (const Idx*)csr->eids;
  const int64_t n = csr->num_rows;
  const Lookup lu{u_off, u_len, D}, le{e_off, e_len, D};
  const bool summing = reduce == MGX_REDUCE_SUM || reduce == MGX_REDUCE_MEAN;
  const bool direct_u = U && !u_off && u_len == D, direct_e = E && !e_off && e_len == D;
#pragma omp parallel
  {
    std::vector<float> acc((size_t)D);
    std::vector<Idx> bu((size_t)(summing ? 0 : D)), be((size_t)(summing ? 0 : D));
#pragma omp for schedule(static, 64)
    for (int64_t v = 0; v < n; ++v) {
      const int64_t beg = (int64_t)indptr[v], end = (int64_t)indptr[v + 1];
      float* o = out + v * D;
      if (summing) {
        std::fill(acc.begin(), acc.end(), 0.f);
        for (int64_t p = beg; p < end; ++p) {
          const int64_t u = indices ? (int64_t)indices[p] : 0;
          const int64_t e = eids ? (int64_t)eids[p] : p;
          const float* ur = U ? U + u * u_len : nullptr;
          const float* er = E ? E + e * e_len : nullptr;
          const float s = (src_scale && U) ? src_scale[u] : 1.f;
          if (op == MGX_OP_COPY_LHS && direct_u) {
            if (src_scale) for (int64_t k = 0; k < D; ++k) acc[k] += ur[k] * s;
            else for (int64_t k = 0; k < D; ++k) acc[k] += ur[k];
          } else if (op == MGX_OP_COPY_RHS && direct_e) {
            for (int64_t k = 0; k < D; ++k) acc[k] += er[k];
          } else {
            for (int64_t k = 0; k < D; ++k) {
              const float l = ur ? ur[lu.at(k)] * s : 0.f;
              const float r = er ? er[le.at(k)] : 0.f;
              acc[k] += combine(op, l, r);
            }
          }
        }
        float scale = dst_scale ? dst_scale[v] : 1.f;
        const bool mean = reduce == MGX_REDUCE_MEAN;
        const float inv = mean ? 1.f / (float)(end - beg > 1 ? end - beg : 1) : 1.f;
        for (int64_t k = 0; k < D; ++k) {
          float r = acc[k];
          if (mean) r = r * inv;
          if (dst_scale) r = r * scale;
          o[k] = accumulate ? o[k] + r : r;
        }
      } else {
        const bool is_max = reduce == MGX_REDUCE_MAX;
        std::fill(acc.begin(), acc.end(), is_max ? -std::numeric_limits<float>::infinity() : std::numeric_limits<float>::infinity());
        std::fill(bu.begin(), bu.end(), (Idx)-1);
        std::fill(be.begin(), be.end(), (Idx)-1);
        for (int64_t p = beg; p < end; ++p) {
          const int64_t u = indices ? (int64_t)indices[p] : 0;
          const int64_t e = eids ? (int64_t)eids[p] : p;
          const float* ur = U ? U + u * u_len : nullptr;
          const float* er = E ? E + e * e_len : nullptr;
          const float s = (src_scale && U) ? src_scale[u] : 1.f;
          for (int64_t k = 0; k < D; ++k) {
            const float l = ur ? ur[lu.at(k)] * s : 0.f;
            const float r = er ? er[le.at(k)] : 0.f;
            const float val = combine(op, l, r);
            if (is_max ? val > acc[k] : val < acc[k]) {  // the first extremum in storage order wins
              acc[k] = val;
              bu[k] = (Idx)u;
              be[k] = (Idx)e;
            }
          }
        }
        for (int64_t k = 0; k < D; ++k) {
          float r = beg == end ? 0.f : acc[k];  // empty rows give 0, arg -1
          if (dst_scale) r *= dst_scale[v];
          o[k] = r;
          if (arg_u) arg_u[v * D + k] = bu[k];
          if (arg_e) arg_e[v * D + k] = be[k];
        }
      }
    }
  }
  return MGX_OK;
}

// ------------------------------------------------------------------------------------------------ g-SDDMM
template <typename Idx>
inline void sddmm_edge(int op, const float* L, const float* R, int64_t li, int64_t ri, int64_t l_len, int64_t r_len, int64_t out_len,
                       int64_t rs, const Lookup& lo, const Lookup& ro, float* o) {
  const float* lr = L ? L + li * l_len : nullptr;
  const float* rr = R ? R + ri * r_len : nullptr;
  if (op == MGX_OP_DOT) {
    for (int64_t k = 0; k < out_len; ++k) {
      float acc = 0.f;
      for (int64_t j = 0; j < rs; ++j) acc += lr[lo.at(k * rs + j)] * rr[ro.at(k * rs + j)];
      o[k] = acc;
    }
    return;
  }
  for (int64_t k = 0; k < out_len; ++k) o[k] = combine(op, lr ? lr[lo.at(k)] : 0.f, rr ? rr[ro.at(k)] : 0.f);
}

inline int64_t pick(int target, int64_t u, int64_t e, int64_t v) { return target == MGX_TARGET_U ? u : (target == MGX_TARGET_V ? v : e); }

int32_t check_sddmm(const char* who, int op, const float* lhs, const float* rhs, int lt, int rt, int64_t l_len, int64_t r_len, int64_t out_len,
                    int64_t rs, const float* out, int64_t nnz) {
  CPU_CHECK(op >= MGX_OP_ADD && op <= MGX_OP_DOT, "%s: unknown op %d", who, op);
  CPU_CHECK(lt >= MGX_TARGET_U && lt <= MGX_TARGET_V && rt >= MGX_TARGET_U && rt <= MGX_TARGET_V, "%s: bad target (%d, %d)", who, lt, rt);
  CPU_CHECK(op == MGX_OP_COPY_RHS || lhs || nnz == 0, "%s: op needs lhs", who);
  CPU_CHECK(op == MGX_OP_COPY_LHS || rhs || nnz == 0, "%s: op needs rhs", who);
  CPU_CHECK(l_len >= 0 && r_len >= 0 && out_len >= 0 && (op != MGX_OP_DOT || rs >= 1), "%s: bad feature lengths", who);
  CPU_CHECK(out || nnz == 0 || out_len == 0, "%s: out is NULL", who);
  return MGX_OK;
}

// ------------------------------------------------------------------------------------------------ formats
template <typename Idx>
void counting_sort_by_row(int64_t num_rows, int64_t nnz, const Idx* row, const Idx* col, Idx* indptr, Idx* indices, Idx* eids) {
  // stable: entries of a row keep their order in the edge list (= edge-id order); two passes over the list, one prefix sum
  std::vector<int64_t> cursor((size_t)num_rows + 1, 0);
  for (int64_t e = 0; e < nnz; ++e) ++cursor[(size_t)row[e] + 1];
  for (int64_t r = 0; r < num_rows; ++r) cursor[r + 1] += cursor[r];
  for (int64_t r = 0; r <= num_rows; ++r) indptr[r] = (Idx)cursor[r];
  for (int64_t e = 0; e < nnz; ++e) {
    const int64_t p = cursor[(size_t)row[e]]++;
    indices[p] = col[e];
    eids[p] = (Idx)e;
  }
}

template <typename Idx>
int32_t transpose(const mgx_csr* csr, Idx* tp, Idx* tx, Idx* te) {
  const int64_t n = csr->num_rows, m = csr->num_cols, nnz = csr->nnz;
  const Idx* indptr = (const Idx*)csr->indptr;
  const Idx* indices = (const Idx*)csr->indices;
  const Idx* eids = (const Idx*)csr->eids;
  // the graph's COO in edge-id order, then the same stable sort by the OTHER endpoint: bit-identical to coo_to_csr on that COO
  std::vector<Idx> src((size_t)nnz), dst((size_t)nnz);
  for (int64_t v = 0; v < n; ++v)
    for (int64_t p = (int64_t)indptr[v]; p < (int64_t)indptr[v + 1]; ++p) {
      const int64_t e = eids ? (int64_t)eids[p] : p;
      if (e < 0 || e >= nnz) return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_cpu_csr_transpose: eids is not a permutation of [0, nnz)");
      src[(size_t)e] = indices[p];
      dst[(size_t)e] = (Idx)v;
    }
  counting_sort_by_row<Idx>(m, nnz, src.data(), dst.data(), tp, tx, te);
  return MGX_OK;
}

}  // namespace

extern "C" const char* mgx_cpu_last_error(void) { return g_err; }
extern "C" int32_t mgx_cpu_num_threads(void) { return (int32_t)omp_get_max_threads(); }
extern "C" void mgx_cpu_set_num_threads(int32_t n) { if (n > 0) omp_set_num_threads(n); }

extern "C" int32_t mgx_cpu_spmm_csr(const mgx_csr* csr, const mgx_spmm_plan*, int32_t op, int32_t reduce, const float* ufeat,
                                    const float* efeat, int64_t u_len, int64_t e_len, int64_t out_len, const int64_t* u_off,
                                    const int64_t* e_off, const float* src_scale, const float* dst_scale, float* out, void* arg_u,
                                    void* arg_e, float*, int32_t flags, void*) {
  if (int st = check_csr(csr, "mgx_cpu_spmm_csr")) return st;
  CPU_CHECK(op == MGX_OP_ADD || op == MGX_OP_MUL || op == MGX_OP_COPY_LHS || op == MGX_OP_COPY_RHS,
            "mgx_cpu_spmm_csr: op must be ADD, MUL, COPY_LHS or COPY_RHS (callers rewrite SUB / DIV), got %d", op);
  CPU_CHECK(reduce >= MGX_REDUCE_SUM && reduce <= MGX_REDUCE_MEAN, "mgx_cpu_spmm_csr: unknown reduce %d", reduce);
  CPU_CHECK(out_len >= 0 && u_len >= 0 && e_len >= 0, "mgx_cpu_spmm_csr: negative feature length");
  if (csr->num_rows == 0 || out_len == 0) return MGX_OK;
  CPU_CHECK(out, "mgx_cpu_spmm_csr: out is NULL");
  CPU_CHECK(op == MGX_OP_COPY_RHS || ufeat || csr->nnz == 0, "mgx_cpu_spmm_csr: op needs ufeat");
  CPU_CHECK(op == MGX_OP_COPY_LHS || efeat || csr->nnz == 0, "mgx_cpu_spmm_csr: op needs efeat");
  const bool summing = reduce == MGX_REDUCE_SUM || reduce == MGX_REDUCE_MEAN;
  const bool accumulate = (flags & MGX_SPMM_ACCUMULATE) != 0;
  CPU_CHECK(!accumulate || summing, "mgx_cpu_spmm_csr: MGX_SPMM_ACCUMULATE is for SUM / MEAN");
  const float* U = op == MGX_OP_COPY_RHS ? nullptr : ufeat;
  const float* E = op == MGX_OP_COPY_LHS ? nullptr : efeat;
  if (csr->idx_bits == 32)
    return spmm_rows<int32_t>(csr, op, reduce, U, E, u_len, e_len, out_len, u_off, e_off, src_scale, dst_scale, out, (int32_t*)arg_u,
                              (int32_t*)arg_e, accumulate);
  return spmm_rows<int64_t>(csr, op, reduce, U, E, u_len, e_len, out_len, u_off, e_off, src_scale, dst_scale, out, (int64_t*)arg_u,
                            (int64_t*)arg_e, accumulate);
}

extern "C" int32_t mgx_cpu_sddmm_coo(int64_t, int64_t, int64_t nnz, const void* src, const void* dst, int32_t idx_bits, int32_t op,
                                     const float* lhs, const float* rhs, int32_t lt, int32_t rt, int64_t l_len, int64_t r_len,
                                     int64_t out_len, int64_t reduce_size, const int64_t* l_off, const int64_t* r_off, float* out, void*) {
  CPU_CHECK(idx_bits == 32 || idx_bits == 64, "mgx_cpu_sddmm_coo: idx_bits must be 32 or 64");
  CPU_CHECK(nnz >= 0 && (nnz == 0 || (src && dst)), "mgx_cpu_sddmm_coo: src / dst is NULL");
  if (int st = check_sddmm("mgx_cpu_sddmm_coo", op, lhs, rhs, lt, rt, l_len, r_len, out_len, reduce_size, out, nnz)) return st;
  const int64_t rs = op == MGX_OP_DOT ? reduce_size : 1, full = out_len * rs;
  const Lookup lo{l_off, l_len, full}, ro{r_off, r_len, full};
  const float* L = op == MGX_OP_COPY_RHS ? nullptr : lhs;
  const float* R = op == MGX_OP_COPY_LHS ? nullptr : rhs;
#pragma omp parallel for schedule(static, 1024)
  for (int64_t e = 0; e < nnz; ++e) {
    const int64_t u = idx_bits == 32 ? (int64_t)((const int32_t*)src)[e] : ((const int64_t*)src)[e];
    const int64_t v = idx_bits == 32 ? (int64_t)((const int32_t*)dst)[e] : ((const int64_t*)dst)[e];
    sddmm_edge<int64_t>(op, L, R, pick(lt, u, e, v), pick(rt, u, e, v), l_len, r_len, out_len, rs, lo, ro, out + e * out_len);
  }
  return MGX_OK;
}

extern "C" int32_t mgx_cpu_sddmm_csr(const mgx_csr* csr, const mgx_spmm_plan*, int32_t op, const float* lhs, const float* rhs, int32_t lt,
                                     int32_t rt, int64_t l_len, int64_t r_len, int64_t out_len, int64_t reduce_size,
                                     const int64_t* l_off, const int64_t* r_off, float* out, void*) {
  if (int st = check_csr(csr, "mgx_cpu_sddmm_csr")) return st;
  if (int st = check_sddmm("mgx_cpu_sddmm_csr", op, lhs, rhs, lt, rt, l_len, r_len, out_len, reduce_size, out, csr->nnz)) return st;
  const int64_t rs = op == MGX_OP_DOT ? reduce_size : 1, full = out_len * rs;
  const Lookup lo{l_off, l_len, full}, ro{r_off, r_len, full};
  const float* L = op == MGX_OP_COPY_RHS ? nullptr : lhs;
  const float* R = op == MGX_OP_COPY_LHS ? nullptr : rhs;
  const bool w32 = csr->idx_bits == 32;
  const int64_t n = csr->num_rows;
#pragma omp parallel for schedule(static, 64)
  for (int64_t v = 0; v < n; ++v) {
    const int64_t beg = w32 ? (int64_t)((const int32_t*)csr->indptr)[v] : ((const int64_t*)csr->indptr)[v];
    const int64_t end = w32 ? (int64_t)((const int32_t*)csr->indptr)[v + 1] : ((const int64_t*)csr->indptr)[v + 1];
    for (int64_t p = beg; p < end; ++p) {
      const int64_t u = w32 ? (int64_t)((const int32_t*)csr->indices)[p] : ((const int64_t*)csr->indices)[p];
      const int64_t e = !csr->eids ? p : (w32 ? (int64_t)((const int32_t*)csr->eids)[p] : ((const int64_t*)csr->eids)[p]);
      sddmm_edge<int64_t>(op, L, R, pick(lt, u, e, v), pick(rt, u, e, v), l_len, r_len, out_len, rs, lo, ro, out + e * out_len);
    }
  }
  return MGX_OK;
}

extern "C" int32_t mgx_cpu_edge_softmax_fwd(const mgx_csr* csr, const mgx_spmm_plan*, int64_t H, const float* z, float* a, float*, void*) {
  if (int st = check_csr(csr, "mgx_cpu_edge_softmax_fwd")) return st;
  CPU_CHECK(H >= 0 && (csr->nnz == 0 || H == 0 || (z && a)), "mgx_cpu_edge_softmax_fwd: NULL operand");
  const bool w32 = csr->idx_bits == 32;
  const int64_t n = csr->num_rows;
#pragma omp parallel for schedule(static, 64)
  for (int64_t v = 0; v < n; ++v) {
    const int64_t beg = w32 ? (int64_t)((const int32_t*)csr->indptr)[v] : ((const int64_t*)csr->indptr)[v];
    const int64_t end = w32 ? (int64_t)((const int32_t*)csr->indptr)[v + 1] : ((const int64_t*)csr->indptr)[v + 1];
    auto edge = [&](int64_t p) { return !csr->eids ? p : (w32 ? (int64_t)((const int32_t*)csr->eids)[p] : ((const int64_t*)csr->eids)[p]); };
    for (int64_t h = 0; h < H; ++h) {
      float m = -std::numeric_limits<float>::infinity();
      for (int64_t p = beg; p < end; ++p) m = std::max(m, z[edge(p) * H + h]);
      float s = 0.f;
      for (int64_t p = beg; p < end; ++p) {
        const float x = expf(z[edge(p) * H + h] - m);
        a[edge(p) * H + h] = x;
        s += x;
      }
      for (int64_t p = beg; p < end; ++p) a[edge(p) * H + h] /= s;
    }
  }
  return MGX_OK;
}

extern "C" int32_t mgx_cpu_edge_softmax_bwd(const mgx_csr* csr, const mgx_spmm_plan*, int64_t H, const float* a, const float* da, float* dz,
                                            float*, void*) {
  if (int st = check_csr(csr, "mgx_cpu_edge_softmax_bwd")) return st;
  CPU_CHECK(H >= 0 && (csr->nnz == 0 || H == 0 || (a && da && dz)), "mgx_cpu_edge_softmax_bwd: NULL operand");
  const bool w32 = csr->idx_bits == 32;
  const int64_t n = csr->num_rows;
#pragma omp parallel for schedule(static, 64)
  for (int64_t v = 0; v < n; ++v) {
    const int64_t beg = w32 ? (int64_t)((const int32_t*)csr->indptr)[v] : ((const int64_t*)csr->indptr)[v];
    const int64_t end = w32 ? (int64_t)((const int32_t*)csr->indptr)[v + 1] : ((const int64_t*)csr->indptr)[v + 1];
    auto edge = [&](int64_t p) { return !csr->eids ? p : (w32 ? (int64_t)((const int32_t*)csr->eids)[p] : ((const int64_t*)csr->eids)[p]); };
    for (int64_t h = 0; h < H; ++h) {
      float t = 0.f;
      for (int64_t p = beg; p < end; ++p) t += a[edge(p) * H + h] * da[edge(p) * H + h];
      for (int64_t p = beg; p < end; ++p) {
        const int64_t i = edge(p) * H + h;
        dz[i] = a[i] * da[i] - a[i] * t;
      }
    }
  }
  return MGX_OK;
}

extern "C" int32_t mgx_cpu_segment_reduce(int64_t nseg, const int64_t* offsets, int64_t D, int32_t reduce, const float* x, float* out,
                                          int64_t* arg, void*) {
  CPU_CHECK(nseg >= 0 && D >= 0, "mgx_cpu_segment_reduce: negative sizes");
  CPU_CHECK(reduce >= MGX_REDUCE_SUM && reduce <= MGX_REDUCE_MEAN, "mgx_cpu_segment_reduce: unknown reduce %d", reduce);
  if (nseg == 0 || D == 0) return MGX_OK;
  CPU_CHECK(offsets && out, "mgx_cpu_segment_reduce: NULL pointer");
#pragma omp parallel for schedule(static, 16)
  for (int64_t s = 0; s < nseg; ++s) {
    const int64_t beg = offsets[s], end = offsets[s + 1];
    float* o = out + s * D;
    if (reduce == MGX_REDUCE_SUM || reduce == MGX_REDUCE_MEAN) {
      for (int64_t k = 0; k < D; ++k) o[k] = 0.f;
      for (int64_t r = beg; r < end; ++r)
        for (int64_t k = 0; k < D; ++k) o[k] += x[r * D + k];
      if (reduce == MGX_REDUCE_MEAN) {
        const float inv = 1.f / (float)(end - beg > 1 ? end - beg : 1);
        for (int64_t k = 0; k < D; ++k) o[k] *= inv;
      }
    } else {
      const bool is_max = reduce == MGX_REDUCE_MAX;
      for (int64_t k = 0; k < D; ++k) {
        float best = is_max ? -std::numeric_limits<float>::infinity() : std::numeric_limits<float>::infinity();
        int64_t at = -1;
        for (int64_t r = beg; r < end; ++r) {
          const float val = x[r * D + k];
          if (is_max ? val > best : val < best) { best = val; at = r; }
        }
        o[k] = beg == end ? 0.f : best;
        if (arg) arg[s * D + k] = at;
      }
    }
  }
  return MGX_OK;
}

extern "C" int32_t mgx_cpu_coo_to_csr(int64_t num_rows, int64_t nnz, const void* row, const void* col, int32_t idx_bits, void* indptr,
                                      void* indices, void* eids, void*, int64_t, void*) {
  CPU_CHECK(idx_bits == 32 || idx_bits == 64, "mgx_cpu_coo_to_csr: idx_bits must be 32 or 64");
  CPU_CHECK(num_rows >= 0 && nnz >= 0 && indptr && (nnz == 0 || (row && col && indices && eids)), "mgx_cpu_coo_to_csr: bad arguments");
  for (int64_t e = 0; e < nnz; ++e) {
    const int64_t r = idx_bits == 32 ? (int64_t)((const int32_t*)row)[e] : ((const int64_t*)row)[e];
    CPU_CHECK(r >= 0 && r < num_rows, "mgx_cpu_coo_to_csr: row id %lld outside [0, %lld)", (long long)r, (long long)num_rows);
  }
  if (idx_bits == 32) counting_sort_by_row<int32_t>(num_rows, nnz, (const int32_t*)row, (const int32_t*)col, (int32_t*)indptr, (int32_t*)indices, (int32_t*)eids);
  else counting_sort_by_row<int64_t>(num_rows, nnz, (const int64_t*)row, (const int64_t*)col, (int64_t*)indptr, (int64_t*)indices, (int64_t*)eids);
  return MGX_OK;
}

extern "C" int32_t mgx_cpu_csr_transpose(const mgx_csr* csr, void* indptr_t, void* indices_t, void* eids_t, void*, int64_t, void*) {
  if (int st = check_csr(csr, "mgx_cpu_csr_transpose")) return st;
  CPU_CHECK(csr->num_cols >= 0 && indptr_t && (csr->nnz == 0 || (indices_t && eids_t)), "mgx_cpu_csr_transpose: bad arguments");
  if (csr->idx_bits == 32) return transpose<int32_t>(csr, (int32_t*)indptr_t, (int32_t*)indices_t, (int32_t*)eids_t);
  return transpose<int64_t>(csr, (int64_t*)indptr_t, (int64_t*)indices_t, (int64_t*)eids_t);
}

extern "C" int32_t mgx_cpu_csr_degrees(int64_t num_rows, const void* indptr, int32_t idx_bits, void* deg, void*) {
  CPU_CHECK(idx_bits == 32 || idx_bits == 64, "mgx_cpu_csr_degrees: idx_bits must be 32 or 64");
  CPU_CHECK(num_rows >= 0 && (num_rows == 0 || (indptr && deg)), "mgx_cpu_csr_degrees: NULL pointer");
#pragma omp parallel for schedule(static, 4096)
  for (int64_t v = 0; v < num_rows; ++v) {
    if (idx_bits == 32) ((int32_t*)deg)[v] = ((const int32_t*)indptr)[v + 1] - ((const int32_t*)indptr)[v];
    else ((int64_t*)deg)[v] = ((const int64_t*)indptr)[v + 1] - ((const int64_t*)indptr)[v];
  }
  return MGX_OK;
}
