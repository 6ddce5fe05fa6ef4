/*
 * oracle.c -- CPU restatement of the sparse message-passing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports, links or
 * executes this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may use it, and there only as the checker / the timed CPU
 * baseline -- never as the thing shipped.
 *
 * PARITY UNPINNED against the reference: dglai/dgl-0.5-benchmark is a pure
 * Python harness with no tests, no golden vectors and no native code; the
 * arithmetic it times lives in the un-vendored third-party wheel `dgl-cu111`
 * (README.md:6 names DGL v0.6.1; docker/build.dockerfile:14 installs it
 * unpinned).  This file therefore restates DGL v0.6.x's *published* CPU
 * algorithm (src/array/cpu/spmm.h SpMMSumCsr / SpMMCmpCsr, cpu/sddmm.h
 * SDDMMCoo, python/dgl/backend/pytorch/sparse.py EdgeSoftmax, cpu/
 * segment_reduce.h) and is anchored on the reference's own call sites:
 *
 *   kernel/dgl-new.py:20      dgl.ops.gspmm(g, binary_op, reduce_op, nfeat, efeat)
 *   kernel/dgl-new.py:39      dgl.ops.gsddmm(g, op, ufeat, vfeat)
 *   kernel/utils.py:8-16      binary_op_dict: dense definitions of add/sub/mul/div/dot/copy_u/copy_e
 *   end_to_end/full_graph/node_classification/main_dgl_product_sage.py:62
 *                             update_all(fn.copy_src('h','m'), fn.mean('m','neigh'))
 *   .../main_dgl_reddit_gat.py:31-55   GATConv = u_add_v SDDMM + edge_softmax + u_mul_e/sum SpMM
 *   .../main_dgl_proteins_rgcn_for.py:52  u_mul_e / mean with (E,1) weights
 *   .../graph_classification/main_dgl_molhiv_gcn.py:41-52,75  in_degrees, copy_e/sum, AvgPooling
 *
 * What pins it instead (tests/test_oracle.py, tests/golden/): scipy.sparse
 * csr_matrix @ X, torch index_add_/scatter_reduce, numpy stable argsort and
 * hand-computed tiny graphs.
 *
 * Algorithmic contract (same as DGL's CPU kernels):
 *   - in-CSR ("CSC"): row v = destination node, indices[p] = source node,
 *     eids[p] = id of that edge in the original COO (NULL => identity).
 *   - rows are independent (OpenMP `parallel for` when built with -fopenmp);
 *     inside a row, fp32 accumulation is SEQUENTIAL in storage order.
 *   - edge features are addressed by edge id, node features by node id.
 *   - broadcasting is expressed by per-output-element offset tables
 *     (u_off / e_off, NULL => identity), as DGL's BcastOff does.
 *
 * Index type: int32 (every reference script calls g.int(),
 * main_dgl_product_sage.py:158); all element offsets are computed in int64
 * because E*D exceeds 2^31 on ogbn-products (123.7M x 64).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_DIV = 3, OP_COPY_LHS = 4, OP_COPY_RHS = 5, OP_DOT = 6 };
enum { RED_SUM = 0, RED_MAX = 1, RED_MIN = 2 };
enum { TGT_U = 0, TGT_E = 1, TGT_V = 2 };

/* kernel/utils.py:8-16 -- the dense definition of every binary op */
static inline float binop(int op, float l, float r) {
  switch (op) {
    case OP_ADD: return l + r;
    case OP_SUB: return l - r;
    case OP_MUL: return l * r;
    case OP_DIV: return l / r;
    case OP_COPY_LHS: return l;
    case OP_COPY_RHS: return r;
    default: return l * r; /* OP_DOT element product */
  }
}

/* ---------------------------------------------------------------------------
 * COO -> CSR, stable counting sort on `row` (edge-id order is kept inside a
 * row).  For the in-CSR the caller passes row = dst, col = src.
 * Follows DGL COOToCSR semantics: data[] = position -> edge id.
 * Used by: g.formats(['csr','csc']) main_dgl_product_sage.py:158.
 * ------------------------------------------------------------------------- */
void orc_coo_to_csr(int64_t n_rows, int64_t nnz, const int32_t* row, const int32_t* col,
                    int32_t* indptr, int32_t* indices, int32_t* eids) {
  int64_t* cursor = (int64_t*)calloc((size_t)n_rows + 1, sizeof(int64_t));
  for (int64_t e = 0; e < nnz; ++e) cursor[row[e] + 1]++;
  for (int64_t r = 0; r < n_rows; ++r) cursor[r + 1] += cursor[r];
  for (int64_t r = 0; r <= n_rows; ++r) indptr[r] = (int32_t)cursor[r];
  for (int64_t e = 0; e < nnz; ++e) {
    int64_t p = cursor[row[e]]++;
    indices[p] = col[e];
    eids[p] = (int32_t)e;
  }
  free(cursor);
}

/* in_degrees(): main_dgl_molhiv_gcn.py:41 -- indptr[v+1]-indptr[v] of the in-CSR */
void orc_in_degrees(int64_t n_rows, const int32_t* indptr, int32_t* deg) {
  for (int64_t v = 0; v < n_rows; ++v) deg[v] = indptr[v + 1] - indptr[v];
}

/* ---------------------------------------------------------------------------
 * g-SpMM:  out[v,k] = REDUCE_{p in row v} op(U[indices[p], u_off[k]], E[eid(p), e_off[k]])
 * kernel/dgl-new.py:20; main_dgl_product_sage.py:62 (copy_lhs/sum + mean divide
 * done by the caller); main_dgl_reddit_gat.py:31-55 (mul/sum with head broadcast).
 * sum: starts at 0.  max/min: empty rows produce 0 and arg = -1.
 * arg_u / arg_e (may be NULL) receive the winning source node / edge id.
 * ------------------------------------------------------------------------- */
void orc_spmm(int64_t n_rows, const int32_t* indptr, const int32_t* indices, const int32_t* eids,
              int op, int reduce, const float* U, const float* E,
              int64_t u_len, int64_t e_len, int64_t out_len,
              const int64_t* u_off, const int64_t* e_off,
              float* out, int32_t* arg_u, int32_t* arg_e) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t v = 0; v < n_rows; ++v) {
    const int64_t beg = indptr[v], end = indptr[v + 1];
    float* o = out + v * out_len;
    if (reduce == RED_SUM) {
      for (int64_t k = 0; k < out_len; ++k) o[k] = 0.f;
      if (op == OP_COPY_LHS && !u_off) { /* the SAGE fast path, same arithmetic */
        for (int64_t p = beg; p < end; ++p) {
          const float* u = U + (int64_t)indices[p] * u_len;
          for (int64_t k = 0; k < out_len; ++k) o[k] += u[k];
        }
        continue;
      }
      for (int64_t p = beg; p < end; ++p) {
        const int64_t eid = eids ? eids[p] : p;
        const float* u = U ? U + (int64_t)indices[p] * u_len : NULL;
        const float* e = E ? E + eid * e_len : NULL;
        for (int64_t k = 0; k < out_len; ++k) {
          const float l = u ? u[u_off ? u_off[k] : k] : 0.f;
          const float r = e ? e[e_off ? e_off[k] : k] : 0.f;
          o[k] += binop(op, l, r);
        }
      }
    } else {
      for (int64_t k = 0; k < out_len; ++k) {
        float best = (reduce == RED_MAX) ? -INFINITY : INFINITY;
        int32_t bu = -1, be = -1;
        for (int64_t p = beg; p < end; ++p) {
          const int64_t eid = eids ? eids[p] : p;
          const float l = U ? U[(int64_t)indices[p] * u_len + (u_off ? u_off[k] : k)] : 0.f;
          const float r = E ? E[eid * e_len + (e_off ? e_off[k] : k)] : 0.f;
          const float val = binop(op, l, r);
          /* strict comparison: the first extremum in storage order wins */
          if ((reduce == RED_MAX) ? (val > best) : (val < best)) {
            best = val; bu = indices[p]; be = (int32_t)eid;
          }
        }
        if (beg == end) best = 0.f;
        o[k] = best;
        if (arg_u) arg_u[v * out_len + k] = bu;
        if (arg_e) arg_e[v * out_len + k] = be;
      }
    }
  }
}

/* ---------------------------------------------------------------------------
 * g-SDDMM on COO:  out[e,k] = op(L[t_l(e), l_off[k]], R[t_r(e), r_off[k]]);
 * dot reduces `reduce_size` trailing elements.  Output is indexed by EDGE ID.
 * kernel/dgl-new.py:39; kernel/utils.py:8-16; gcmc_dgl/model.py:342 (u_dot_v).
 * ------------------------------------------------------------------------- */
void orc_sddmm(int64_t nnz, const int32_t* src, const int32_t* dst, int op,
               const float* L, const float* R, int lhs_target, int rhs_target,
               int64_t l_len, int64_t r_len, int64_t out_len, int64_t reduce_size,
               const int64_t* l_off, const int64_t* r_off, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < nnz; ++e) {
    const int64_t li = lhs_target == TGT_U ? src[e] : (lhs_target == TGT_V ? dst[e] : e);
    const int64_t ri = rhs_target == TGT_U ? src[e] : (rhs_target == TGT_V ? dst[e] : e);
    const float* l = L ? L + li * l_len : NULL;
    const float* r = R ? R + ri * r_len : NULL;
    float* o = out + e * out_len;
    for (int64_t k = 0; k < out_len; ++k) {
      const int64_t lo = l_off ? l_off[k] : k, ro = r_off ? r_off[k] : k;
      if (op == OP_DOT) {
        float acc = 0.f;
        for (int64_t j = 0; j < reduce_size; ++j) acc += l[lo * reduce_size + j] * r[ro * reduce_size + j];
        o[k] = acc;
      } else {
        o[k] = binop(op, l ? l[lo] : 0.f, r ? r[ro] : 0.f);
      }
    }
  }
}

/* ---------------------------------------------------------------------------
 * edge_softmax (norm_by='dst'), reached through GATConv
 * (main_dgl_reddit_gat.py:31-55).  DGL 0.6 composes it from
 *   m = copy_rhs/max SpMM ; s = exp(z - m[v]) ; d = copy_rhs/sum SpMM ; a = s / d[v]
 * which is what is restated here, per trailing element (head) h.
 * z, a: (E, H) addressed by edge id.
 * ------------------------------------------------------------------------- */
void orc_edge_softmax_fwd(int64_t n_rows, const int32_t* indptr, const int32_t* eids,
                          int64_t H, const float* z, float* a) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t v = 0; v < n_rows; ++v) {
    const int64_t beg = indptr[v], end = indptr[v + 1];
    for (int64_t h = 0; h < H; ++h) {
      float m = -INFINITY;
      for (int64_t p = beg; p < end; ++p) {
        const float x = z[(eids ? eids[p] : p) * H + h];
        if (x > m) m = x;
      }
      float d = 0.f;
      for (int64_t p = beg; p < end; ++p) {
        const int64_t e = eids ? eids[p] : p;
        const float s = expf(z[e * H + h] - m);
        a[e * H + h] = s;
        d += s;
      }
      for (int64_t p = beg; p < end; ++p) a[(eids ? eids[p] : p) * H + h] /= d;
    }
  }
}

/* backward: dz = a*da - a * sum_{e'->v}(a*da)   (DGL EdgeSoftmax.backward) */
void orc_edge_softmax_bwd(int64_t n_rows, const int32_t* indptr, const int32_t* eids,
                          int64_t H, const float* a, const float* da, float* dz) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t v = 0; v < n_rows; ++v) {
    const int64_t beg = indptr[v], end = indptr[v + 1];
    for (int64_t h = 0; h < H; ++h) {
      float acc = 0.f;
      for (int64_t p = beg; p < end; ++p) {
        const int64_t e = eids ? eids[p] : p;
        acc += a[e * H + h] * da[e * H + h];
      }
      for (int64_t p = beg; p < end; ++p) {
        const int64_t e = eids ? eids[p] : p;
        dz[e * H + h] = a[e * H + h] * da[e * H + h] - a[e * H + h] * acc;
      }
    }
  }
}

/* ---------------------------------------------------------------------------
 * segment_reduce: out[s,:] = REDUCE_{i in [off[s], off[s+1])} x[i,:]
 * dgl.nn.AvgPooling (main_dgl_molhiv_gcn.py:75,93) = sum / max(len,1).
 * reduce: RED_SUM / RED_MAX / RED_MIN; mean: reduce = 3.
 * ------------------------------------------------------------------------- */
void orc_segment_reduce(int64_t n_seg, const int64_t* offsets, int64_t D, int reduce,
                        const float* x, float* out, int64_t* arg) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t s = 0; s < n_seg; ++s) {
    const int64_t beg = offsets[s], end = offsets[s + 1];
    for (int64_t k = 0; k < D; ++k) {
      if (reduce == RED_SUM || reduce == 3) {
        float acc = 0.f;
        for (int64_t i = beg; i < end; ++i) acc += x[i * D + k];
        if (reduce == 3) acc /= (float)((end - beg) > 1 ? (end - beg) : 1);
        out[s * D + k] = acc;
      } else {
        float best = (reduce == RED_MAX) ? -INFINITY : INFINITY;
        int64_t bi = -1;
        for (int64_t i = beg; i < end; ++i) {
          const float val = x[i * D + k];
          if ((reduce == RED_MAX) ? (val > best) : (val < best)) { best = val; bi = i; }
        }
        out[s * D + k] = (beg == end) ? 0.f : best;
        if (arg) arg[s * D + k] = bi;
      }
    }
  }
}

/* number of OpenMP threads the library was built to use (1 without -fopenmp) */
#ifdef _OPENMP
#include <omp.h>
int orc_num_threads(void) { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { omp_set_num_threads(n); }
#else
int orc_num_threads(void) { return 1; }
void orc_set_num_threads(int n) { (void)n; }
#endif
