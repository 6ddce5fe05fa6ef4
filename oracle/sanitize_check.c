/*
 * sanitize_check.c -- runs every entry point of oracle.c on ragged random inputs (empty rows, empty graphs, width 1, broadcast
 * offset tables, permuted edge ids) with EXACTLY sized heap buffers, built with -fsanitize=address,undefined: an access outside a
 * buffer, a signed overflow in an offset or a misaligned load stops the program.  TEST INFRASTRUCTURE, like oracle.c itself
 * (tests/test_oracle_sanitize.py builds and runs it; SURVEY.md section 5 lists the sanitizer build of the CPU restatement).
 * Prints "sanitize_check ok <checksum>"; the checksum only keeps the calls from being optimised away.
 */
#include <stdio.h>
#include "oracle.c"

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd(void) {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}
#define ADD(x) do { const double t_ = (double)(x); if (isfinite(t_)) sum += t_; } while (0)
static float rndf(void) { return (float)(rnd() % 2001) / 1000.0f - 1.0f; }
static void* exact(size_t n, size_t sz) { return malloc(n * sz ? n * sz : 1); }  /* zero-sized: 1 byte, any access is out of bounds */
static float* rand_mat(int64_t r, int64_t c) {
  float* m = (float*)exact((size_t)(r * c), sizeof(float));
  for (int64_t i = 0; i < r * c; ++i) m[i] = rndf();
  return m;
}

static double run_case(int64_t n_src, int64_t n_dst, int64_t nnz, int64_t D) {
  double sum = 0.0;
  int32_t* src = (int32_t*)exact((size_t)nnz, 4);
  int32_t* dst = (int32_t*)exact((size_t)nnz, 4);
  for (int64_t e = 0; e < nnz; ++e) {
    src[e] = (int32_t)(rnd() % (uint32_t)n_src);
    /* skewed destinations: a quarter of the rows stay empty, one row is a hub */
    dst[e] = (rnd() % 8 == 0) ? 0 : (int32_t)(rnd() % (uint32_t)(n_dst - n_dst / 4 > 0 ? n_dst - n_dst / 4 : 1));
  }
  int32_t* indptr = (int32_t*)exact((size_t)n_dst + 1, 4);
  int32_t* indices = (int32_t*)exact((size_t)nnz, 4);
  int32_t* eids = (int32_t*)exact((size_t)nnz, 4);
  orc_coo_to_csr(n_dst, nnz, dst, src, indptr, indices, eids);
  int32_t* deg = (int32_t*)exact((size_t)n_dst, 4);
  orc_in_degrees(n_dst, indptr, deg);
  for (int64_t v = 0; v < n_dst; ++v) sum += deg[v];

  float* U = rand_mat(n_src, D);
  float* E = rand_mat(nnz, D);
  float* E1 = rand_mat(nnz, 1);
  float* V = rand_mat(n_dst, D);
  float* out = (float*)exact((size_t)(n_dst * D), 4);
  int32_t* au = (int32_t*)exact((size_t)(n_dst * D), 4);
  int32_t* ae = (int32_t*)exact((size_t)(n_dst * D), 4);
  int64_t* ident = (int64_t*)exact((size_t)D, 8);
  int64_t* zeros = (int64_t*)exact((size_t)D, 8);
  for (int64_t k = 0; k < D; ++k) { ident[k] = k; zeros[k] = 0; }
  for (int op = OP_ADD; op <= OP_COPY_RHS; ++op) {
    for (int red = RED_SUM; red <= RED_MIN; ++red) {
      const float* u = op == OP_COPY_RHS ? NULL : U;
      const float* e = op == OP_COPY_LHS ? NULL : E;
      orc_spmm(n_dst, indptr, indices, eids, op, red, u, e, D, D, D, NULL, NULL, out, red ? au : NULL, red ? ae : NULL);
      for (int64_t i = 0; i < n_dst * D; ++i) ADD(out[i]);
      orc_spmm(n_dst, indptr, indices, NULL, op, red, u, e, D, D, D, ident, ident, out, NULL, NULL);  /* identity edge ids + tables */
      if (op != OP_COPY_LHS && op != OP_COPY_RHS) {  /* (E,1) weights broadcast over the feature dimension (proteins_rgcn_for.py:52) */
        orc_spmm(n_dst, indptr, indices, eids, op, red, U, E1, D, 1, D, ident, zeros, out, red ? au : NULL, red ? ae : NULL);
        for (int64_t i = 0; i < n_dst * D; ++i) ADD(out[i]);
      }
    }
  }
  float* eo = (float*)exact((size_t)(nnz * D), 4);
  float* e1 = (float*)exact((size_t)nnz, 4);
  for (int op = OP_ADD; op <= OP_DIV; ++op) {
    orc_sddmm(nnz, src, dst, op, U, V, TGT_U, TGT_V, D, D, D, 1, NULL, NULL, eo);
    for (int64_t i = 0; i < nnz * D; ++i) ADD(eo[i]);
    orc_sddmm(nnz, src, dst, op, E, V, TGT_E, TGT_V, D, D, D, 1, ident, ident, eo);
  }
  orc_sddmm(nnz, src, dst, OP_DOT, U, V, TGT_U, TGT_V, D, D, 1, D, NULL, NULL, e1);
  for (int64_t i = 0; i < nnz; ++i) ADD(e1[i]);
  orc_sddmm(nnz, src, dst, OP_COPY_LHS, U, NULL, TGT_U, TGT_V, D, D, D, 1, NULL, NULL, eo);
  orc_sddmm(nnz, src, dst, OP_COPY_RHS, NULL, V, TGT_U, TGT_V, D, D, D, 1, NULL, NULL, eo);

  float* a = (float*)exact((size_t)(nnz * D), 4);
  float* dz = (float*)exact((size_t)(nnz * D), 4);
  orc_edge_softmax_fwd(n_dst, indptr, eids, D, E, a);        /* D heads */
  orc_edge_softmax_bwd(n_dst, indptr, eids, D, a, E, dz);
  for (int64_t i = 0; i < nnz * D; ++i) ADD(a[i] + dz[i]);
  orc_edge_softmax_fwd(n_dst, indptr, NULL, D, E, a);

  /* segment reduce over the source rows: ragged segments, some empty */
  const int64_t n_seg = n_src / 3 + 1;
  int64_t* offs = (int64_t*)exact((size_t)n_seg + 1, 8);
  offs[0] = 0;
  for (int64_t s = 0; s < n_seg; ++s) {
    int64_t left = n_src - offs[s];
    int64_t take = (s == n_seg - 1) ? left : (int64_t)(rnd() % 5);
    if (take > left) take = left;
    offs[s + 1] = offs[s] + take;
  }
  float* so = (float*)exact((size_t)(n_seg * D), 4);
  int64_t* sa = (int64_t*)exact((size_t)(n_seg * D), 8);
  for (int red = RED_SUM; red <= RED_MIN; ++red) {
    orc_segment_reduce(n_seg, offs, D, red, U, so, red ? sa : NULL);
    for (int64_t i = 0; i < n_seg * D; ++i) ADD(so[i]);
  }
  free(src); free(dst); free(indptr); free(indices); free(eids); free(deg); free(U); free(E); free(E1); free(V); free(out);
  free(au); free(ae); free(ident); free(zeros); free(eo); free(e1); free(a); free(dz); free(offs); free(so); free(sa);
  return sum;
}

int main(void) {
  double sum = 0.0;
#ifdef SANITIZE_SELFTEST  /* the harness itself: a CSR whose last index points one row beyond U must stop the program */
  {
    int32_t indptr[2] = {0, 1}, indices[1] = {2};
    float* U = rand_mat(2, 4);
    float out[4];
    orc_spmm(1, indptr, indices, NULL, OP_COPY_LHS, RED_SUM, U, NULL, 4, 4, 4, NULL, NULL, out, NULL, NULL);
    sum += out[0];
    free(U);
  }
#endif
  const int64_t shapes[][4] = {{1, 1, 0, 1}, {1, 1, 3, 1}, {7, 5, 40, 1}, {50, 64, 700, 3}, {300, 257, 5000, 8}, {64, 1000, 900, 33}, {2000, 1500, 30000, 16}};
  for (unsigned i = 0; i < sizeof(shapes) / sizeof(shapes[0]); ++i) sum += run_case(shapes[i][0], shapes[i][1], shapes[i][2], shapes[i][3]);
  orc_set_num_threads(2);
  sum += run_case(500, 400, 9000, 5) + orc_num_threads();
  printf("sanitize_check ok %.6e\n", sum);
  return 0;
}
