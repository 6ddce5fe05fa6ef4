/* A plain-C consumer of include/mi355x_graph.h: no Python, no torch.  Builds the in-CSR of a 5-node graph on the
 * device with mgx_coo_to_csr, runs copy_u/mean g-SpMM, u_add_v g-SDDMM and the fused edge softmax, and checks the
 * results against hand-computed values (the same tiny graph as tests/golden/tiny.npz).
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ abi_smoke.c -I../../include -I/opt/rocm/include -L<csrc> -lmi355x_graph -L/opt/rocm/lib -lamdhip64 -lm
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mi355x_graph.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)
#define CHECK_MGX(x) do { int32_t s_ = (x); if (s_ != MGX_OK) { printf("mgx error %d: %s (line %d)\n", s_, mgx_last_error(), __LINE__); return 3; } } while (0)

static void* dev_copy(const void* host, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) return NULL;
  if (bytes && host) hipMemcpy(d, host, bytes, hipMemcpyHostToDevice);
  return d;
}

int main(void) {
  /* edges u->v; node 4 is isolated, (3,3) is a self loop, (1,2) appears twice */
  const int32_t src[8] = {0, 1, 2, 0, 0, 3, 3, 1};
  const int32_t dst[8] = {1, 2, 3, 3, 3, 3, 0, 2};
  const int n = 5, nnz = 8, D = 2;
  float X[10];
  for (int i = 0; i < 10; ++i) X[i] = (float)(i + 1); /* [[1,2],[3,4],[5,6],[7,8],[9,10]] */
  if (mgx_abi_version() < 30) { printf("unexpected ABI version %d\n", mgx_abi_version()); return 1; }

  int32_t *d_src = dev_copy(src, sizeof src), *d_dst = dev_copy(dst, sizeof dst);
  float* d_X = dev_copy(X, sizeof X);
  int32_t *d_indptr = dev_copy(NULL, (n + 1) * 4), *d_indices = dev_copy(NULL, nnz * 4), *d_eids = dev_copy(NULL, nnz * 4);
  int64_t ws_bytes = mgx_coo_to_csr_workspace(n, nnz, 32);
  if (ws_bytes < 0) { printf("workspace query failed: %s\n", mgx_last_error()); return 3; }
  void* d_ws = dev_copy(NULL, (size_t)ws_bytes);
  /* in-CSR: rows = destinations */
  CHECK_MGX(mgx_coo_to_csr(n, nnz, d_dst, d_src, 32, d_indptr, d_indices, d_eids, d_ws, ws_bytes, NULL));
  int32_t indptr[6], indices[8], eids[8];
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(indptr, d_indptr, sizeof indptr, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(indices, d_indices, sizeof indices, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(eids, d_eids, sizeof eids, hipMemcpyDeviceToHost));
  const int32_t want_indptr[6] = {0, 1, 2, 4, 8, 8}, want_indices[8] = {3, 0, 1, 1, 2, 0, 0, 3}, want_eids[8] = {6, 0, 1, 7, 2, 3, 4, 5};
  if (memcmp(indptr, want_indptr, sizeof indptr) || memcmp(indices, want_indices, sizeof indices) || memcmp(eids, want_eids, sizeof eids)) {
    printf("coo_to_csr mismatch\n");
    return 4;
  }

  mgx_csr csr = {n, n, nnz, d_indptr, d_indices, d_eids, 32, 0};
  float* d_out = dev_copy(NULL, n * D * 4);
  CHECK_MGX(mgx_spmm_csr(&csr, NULL, MGX_OP_COPY_LHS, MGX_REDUCE_MEAN, d_X, NULL, D, 0, D, NULL, NULL, NULL, NULL, d_out, NULL,
                         NULL, NULL, 0, NULL));
  float out[10];
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost));
  /* in-neighbours: 0<-{3}, 1<-{0}, 2<-{1,1}, 3<-{2,0,0,3}, 4<-{} ; mean = sum / max(deg,1) */
  const float want[10] = {7, 8, 1, 2, 3, 4, 14.f / 4, 18.f / 4, 0, 0};
  for (int i = 0; i < 10; ++i)
    if (fabsf(out[i] - want[i]) > 1e-6f) { printf("spmm mismatch at %d: %f vs %f\n", i, out[i], want[i]); return 5; }

  /* u_add_v on the first feature column (stride trick: use D = 2 rows, out_len = 2) */
  float* d_e = dev_copy(NULL, nnz * D * 4);
  CHECK_MGX(mgx_sddmm_coo(n, n, nnz, d_src, d_dst, 32, MGX_OP_ADD, d_X, d_X, MGX_TARGET_U, MGX_TARGET_V, D, D, D, 1, NULL, NULL,
                          d_e, NULL));
  float e[16];
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(e, d_e, sizeof e, hipMemcpyDeviceToHost));
  for (int k = 0; k < nnz; ++k)
    for (int j = 0; j < D; ++j)
      if (e[k * D + j] != X[src[k] * D + j] + X[dst[k] * D + j]) { printf("sddmm mismatch at edge %d\n", k); return 6; }

  /* edge softmax over the in-edges of every destination, H = 2 heads */
  float* d_a = dev_copy(NULL, nnz * D * 4);
  CHECK_MGX(mgx_edge_softmax_fwd(&csr, NULL, D, d_e, d_a, NULL, NULL));
  float a[16];
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(a, d_a, sizeof a, hipMemcpyDeviceToHost));
  for (int v = 0; v < n; ++v)
    for (int j = 0; j < D; ++j) {
      float s = 0.f;
      int cnt = 0;
      for (int k = 0; k < nnz; ++k)
        if (dst[k] == v) { s += a[k * D + j]; ++cnt; }
      if (cnt && fabsf(s - 1.f) > 1e-5f) { printf("softmax of node %d does not sum to 1: %f\n", v, s); return 7; }
    }

  /* a two-part plan (mgx_spmm_plan::rest) from a plain-C caller: rows 0, 1, 2 (<= 2 edges) on the lane-group kernel, row 3 (4 edges) and
   * the empty row 4 in `rest`; same result as above */
  {
    const int32_t h_row[3] = {0, 1, 2}, h_beg[3] = {0, 1, 2}, h_end[3] = {1, 2, 4};
    const int32_t r_row[2] = {3, 4}, r_beg[2] = {4, 8}, r_end[2] = {8, 8};
    mgx_spmm_plan rest, head;
    memset(&rest, 0, sizeof rest);
    memset(&head, 0, sizeof head);
    rest.num_items = 2;
    rest.item_row = (const int32_t*)dev_copy(r_row, sizeof r_row);
    rest.item_beg = dev_copy(r_beg, sizeof r_beg);
    rest.item_end = dev_copy(r_end, sizeof r_end);
    rest.item_node = rest.item_row;
    head.num_items = 3;
    head.item_row = (const int32_t*)dev_copy(h_row, sizeof h_row);
    head.item_beg = dev_copy(h_beg, sizeof h_beg);
    head.item_end = dev_copy(h_end, sizeof h_end);
    head.item_node = head.item_row;
    head.rest = &rest;
    float* d_out2 = dev_copy(NULL, n * 4 * 4);
    float X4[20], out4[20];
    for (int i = 0; i < n; ++i) { X4[4 * i] = X[2 * i]; X4[4 * i + 1] = X[2 * i + 1]; X4[4 * i + 2] = -X[2 * i]; X4[4 * i + 3] = 1.f; }
    float* d_X4 = dev_copy(X4, sizeof X4);
    CHECK_MGX(mgx_spmm_csr(&csr, &head, MGX_OP_COPY_LHS, MGX_REDUCE_MEAN, d_X4, NULL, 4, 0, 4, NULL, NULL, NULL, NULL, d_out2, NULL,
                           NULL, NULL, MGX_SPMM_SHORT_ROWS, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(out4, d_out2, sizeof out4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i)
      if (fabsf(out4[4 * i] - want[2 * i]) > 1e-6f || fabsf(out4[4 * i + 1] - want[2 * i + 1]) > 1e-6f ||
          fabsf(out4[4 * i + 2] + want[2 * i]) > 1e-6f || fabsf(out4[4 * i + 3] - (i == 4 ? 0.f : 1.f)) > 1e-6f) {
        printf("two-part plan mismatch at row %d (%s)\n", i, mgx_last_spmm_kernel());
        return 9;
      }
    if (strcmp(mgx_last_spmm_kernel(), "rowgroup32") != 0) { printf("two-part plan took %s\n", mgx_last_spmm_kernel()); return 9; }
  }

  /* the dense helpers of a layer: C = A B + bias with the second half of the columns times a row factor, as two compact matrices;
   * A^T B with the column sums of A */
  {
    enum { R = 40, K = 32, M = 64 };
    static float A[R * K], B[K * M], bias[M], rs[R], C1[R * (M / 2)], C2[R * (M / 2)], W[K * M], sums[K];
    for (int i = 0; i < R * K; ++i) A[i] = (float)((i * 7) % 11) - 5.f;
    for (int i = 0; i < K * M; ++i) B[i] = (float)((i * 5) % 13) - 6.f;
    for (int i = 0; i < M; ++i) bias[i] = (float)i;
    for (int i = 0; i < R; ++i) rs[i] = 1.f / (float)(1 + i % 4);
    float *d_A = dev_copy(A, sizeof A), *d_B = dev_copy(B, sizeof B), *d_b = dev_copy(bias, sizeof bias), *d_rs = dev_copy(rs, sizeof rs);
    float *d_C1 = dev_copy(NULL, sizeof C1), *d_C2 = dev_copy(NULL, sizeof C2);
    if (!mgx_rows_gemm_supported(K, M, K)) { printf("mgx_rows_gemm_supported(32, 64) is 0\n"); return 10; }
    CHECK_MGX(mgx_rows_gemm(R, K, M, d_A, K, d_B, M, 0, d_b, d_rs, M / 2, d_C1, M / 2, d_C2, M / 2, M / 2, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(C1, d_C1, sizeof C1, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(C2, d_C2, sizeof C2, hipMemcpyDeviceToHost));
    for (int r = 0; r < R; ++r)
      for (int m = 0; m < M; ++m) {
        float ref = bias[m];
        for (int k = 0; k < K; ++k) ref += A[r * K + k] * B[k * M + m];   /* small integers: exact in fp32 whatever the order */
        if (m >= M / 2) ref *= rs[r];
        const float got = m < M / 2 ? C1[r * (M / 2) + m] : C2[r * (M / 2) + m - M / 2];
        if (fabsf(got - ref) > 1e-4f * (1.f + fabsf(ref))) { printf("rows_gemm mismatch at (%d, %d): %f vs %f\n", r, m, got, ref); return 10; }
      }
    /* A^T [A | A] (K x 2K = 32 x 64) and the column sums of A */
    static float AA[R * 2 * K];
    for (int r = 0; r < R; ++r)
      for (int k = 0; k < 2 * K; ++k) AA[r * 2 * K + k] = A[r * K + k % K];
    float *d_AA = dev_copy(AA, sizeof AA), *d_W = dev_copy(NULL, sizeof W), *d_s = dev_copy(NULL, sizeof sums);
    const int64_t xws = mgx_xty_workspace(K, 2 * K);
    void* d_xws = dev_copy(NULL, (size_t)xws);
    CHECK_MGX(mgx_xty_colsum(R, K, 2 * K, d_A, K, d_AA, 2 * K, d_W, 2 * K, d_s, d_xws, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(W, d_W, sizeof W, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(sums, d_s, sizeof sums, hipMemcpyDeviceToHost));
    for (int i = 0; i < K; ++i) {
      float cs = 0.f;
      for (int r = 0; r < R; ++r) cs += A[r * K + i];
      if (sums[i] != cs) { printf("xty_colsum: column sum %d is %f, expected %f\n", i, sums[i], cs); return 11; }
      for (int j = 0; j < 2 * K; ++j) {
        float ref = 0.f;
        for (int r = 0; r < R; ++r) ref += A[r * K + i] * AA[r * 2 * K + j];
        if (W[i * 2 * K + j] != ref) { printf("xty mismatch at (%d, %d)\n", i, j); return 11; }
      }
    }
  }

  /* round 5: halo rows as bitmaps + packed values -- unpack(pack(x)) == x bit for bit, rows gathered through an index, a strided source */
  {
    enum { NR = 37, PD = 100, LD = 104, NS = 23 };
    static float P[NR * LD], Q[NS * PD];
    static int32_t idx[NS], cnt[NS];
    static int64_t off[NS + 1];
    static uint64_t msk[NS * 2];
    for (int i = 0; i < NR * LD; ++i) P[i] = ((i * 2654435761u) >> 7) % 4 == 0 ? (float)(i % 97) - 48.f : 0.f;  /* ~25 % non-zero */
    for (int i = 0; i < NS; ++i) idx[i] = (i * 7) % NR;
    if (mgx_rows_mask_words(PD) != 2) { printf("mgx_rows_mask_words(100) != 2\n"); return 12; }
    float* d_P = dev_copy(P, sizeof P);
    int32_t* d_idx = dev_copy(idx, sizeof idx);
    uint64_t* d_msk = dev_copy(NULL, sizeof msk);
    int32_t* d_cnt = dev_copy(NULL, sizeof cnt);
    CHECK_MGX(mgx_rows_pack_count(NS, d_idx, 32, PD, d_P, LD, d_msk, d_cnt, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost));
    off[0] = 0;
    for (int i = 0; i < NS; ++i) {
      int want_n = 0;
      for (int c = 0; c < PD; ++c) want_n += P[idx[i] * LD + c] != 0.f;
      if (cnt[i] != want_n) { printf("rows_pack_count: row %d has %d non-zeros, kernel says %d\n", i, want_n, cnt[i]); return 12; }
      off[i + 1] = off[i] + cnt[i];
    }
    int64_t* d_off = dev_copy(off, sizeof off);
    float* d_val = dev_copy(NULL, (size_t)off[NS] * 4);
    float* d_Q = dev_copy(NULL, sizeof Q);
    CHECK_MGX(mgx_rows_pack_values(NS, d_idx, 32, PD, d_P, LD, d_msk, d_off, d_val, NULL));
    CHECK_MGX(mgx_rows_unpack(NS, PD, d_msk, d_off, d_val, d_Q, PD, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(Q, d_Q, sizeof Q, hipMemcpyDeviceToHost));
    for (int i = 0; i < NS; ++i)
      for (int c = 0; c < PD; ++c)
        if (Q[i * PD + c] != P[idx[i] * LD + c]) { printf("rows pack / unpack mismatch at (%d, %d)\n", i, c); return 12; }
  }

  /* round 5: u_add_v walked in the in-CSR's order with the edge id as the output row == the walk in edge-id order */
  {
    enum { SD = 8 };
    static float U8[5 * SD], a_coo[8 * SD], a_perm[8 * SD];
    for (int i = 0; i < 5 * SD; ++i) U8[i] = (float)(i % 13) * 0.5f;
    /* the CSR as an edge list in ITS order: sources = indices, destinations = the row of every position, perm = eids */
    const int32_t rows_of_pos[8] = {0, 1, 2, 2, 3, 3, 3, 3};
    float* d_U8 = dev_copy(U8, sizeof U8);
    int32_t* d_rows = dev_copy(rows_of_pos, sizeof rows_of_pos);
    float *d_a1 = dev_copy(NULL, sizeof a_coo), *d_a2 = dev_copy(NULL, sizeof a_perm);
    CHECK_MGX(mgx_sddmm_coo(n, n, nnz, d_src, d_dst, 32, MGX_OP_ADD, d_U8, d_U8, MGX_TARGET_U, MGX_TARGET_V, SD, SD, SD, 1, NULL, NULL, d_a1, NULL));
    CHECK_MGX(mgx_sddmm_coo_perm(n, n, nnz, d_indices, d_rows, d_eids, 32, MGX_OP_ADD, d_U8, d_U8, MGX_TARGET_U, MGX_TARGET_V, SD, d_a2, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(a_coo, d_a1, sizeof a_coo, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(a_perm, d_a2, sizeof a_perm, hipMemcpyDeviceToHost));
    for (int e = 0; e < nnz; ++e)
      for (int c = 0; c < SD; ++c) {
        const float ref = U8[src[e] * SD + c] + U8[dst[e] * SD + c];
        if (a_coo[e * SD + c] != ref || a_perm[e * SD + c] != ref) { printf("sddmm_coo_perm mismatch at edge %d\n", e); return 13; }
      }
  }

  /* round 5: mostly-zero rows of 64 columns as 128-byte slots -- the slot aggregation equals the dense one (small integers: exact in any
     order), with a row factor folded into the slots, and a row above 24 non-zeros read from the dense matrix */
  {
    enum { SW = 64 };
    static float X64[5 * SW], sc5[5], o_dense[5 * SW], o_slots[5 * SW];
    static uint32_t slot_words[5 * 32];
    for (int i = 0; i < 5 * SW; ++i) X64[i] = (i * 7 + i / SW) % 5 == 0 ? (float)(i % 11 - 5) : 0.f;   /* ~20 % non-zero */
    for (int c = 0; c < 40; ++c) X64[3 * SW + c] = (float)(c % 7 + 1);                                  /* row 3: 40 non-zeros */
    for (int r = 0; r < 5; ++r) sc5[r] = (float)(1 << r);                                                /* powers of two: exact products */
    float *d_X = dev_copy(X64, sizeof X64), *d_sc = dev_copy(sc5, sizeof sc5);
    float *d_o1 = dev_copy(NULL, sizeof o_dense), *d_o2 = dev_copy(NULL, sizeof o_slots);
    uint32_t* d_slots = dev_copy(NULL, sizeof slot_words);
    int64_t zero64 = 0, over = -1;
    int64_t* d_over = dev_copy(&zero64, sizeof zero64);
    CHECK_MGX(mgx_rows_slots_pack(5, SW, d_X, SW, d_sc, d_slots, d_over, NULL));
    CHECK_MGX(mgx_spmm_copy_u_slots(&csr, NULL, MGX_REDUCE_SUM, d_X, SW, SW, d_slots, d_sc, NULL, d_o2, SW, NULL, 0, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(&over, d_over, sizeof over, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(slot_words, d_slots, sizeof slot_words, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(o_slots, d_o2, sizeof o_slots, hipMemcpyDeviceToHost));
    if (over != 1 || (slot_words[3 * 32] >> 24) != 255u || (slot_words[0] >> 24) != 0u) { printf("rows_slots_pack: overflow row not flagged\n"); return 14; }
    for (int v = 0; v < n; ++v)
      for (int c = 0; c < SW; ++c) {
        float ref = 0.f;
        for (int e = 0; e < nnz; ++e)
          if (dst[e] == v) ref += sc5[src[e]] * X64[src[e] * SW + c];
        if (o_slots[v * SW + c] != ref) { printf("spmm_copy_u_slots mismatch at (%d, %d): %f != %f\n", v, c, o_slots[v * SW + c], ref); return 14; }
      }
    (void)d_o1; (void)o_dense;
  }

  /* error path: bad argument must return a code and a message, not crash */
  if (mgx_spmm_csr(NULL, NULL, 0, 0, NULL, NULL, 1, 1, 1, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL) != MGX_ERR_INVALID_ARGUMENT ||
      strstr(mgx_last_error(), "csr is NULL") == NULL) { printf("error path broken\n"); return 8; }
  printf("abi_smoke ok\n");
  return 0;
}
