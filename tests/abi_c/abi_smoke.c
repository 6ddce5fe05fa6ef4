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
  if (mgx_abi_version() < 8) { printf("unexpected ABI version %d\n", mgx_abi_version()); return 1; }

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

  /* error path: bad argument must return a code and a message, not crash */
  if (mgx_spmm_csr(NULL, NULL, 0, 0, NULL, NULL, 1, 1, 1, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL) != MGX_ERR_INVALID_ARGUMENT ||
      strstr(mgx_last_error(), "csr is NULL") == NULL) { printf("error path broken\n"); return 8; }
  printf("abi_smoke ok\n");
  return 0;
}
