/* mi355x_graph_cpu.h -- CPU (OpenMP) variants of the hot-path entry points, same signatures as include/mi355x_graph.h.
 *
 * "... plus CPU (OpenMP) variants of each with the same signatures".  The reference reaches DGL's CPU kernels with
 * `--gpu -1` (kernel/dgl-new.py:55-58) and when no GPU is visible (main_dgl_product_sage.py:149); BASELINE configs[0] "runs
 * without a GPU".  These are product code (csrc/cpu_ops.cpp -> libmi355x_graph_cpu.so, g++ -fopenmp), written for this
 * library -- not the test oracle, which lives under oracle/ and is only ever the checker.
 *
 * Opt-in: the host layer routes CPU tensors here only after mi355x_graph.enable_cpu_backend() (or MGX_CPU_BACKEND=1); by default
 * message passing on CPU tensors raises DGLError, and nothing EVER falls back from the GPU path to this one.
 *
 * Conventions: pointers are HOST pointers; `stream`, `plan`, `partial_ws` / `ws` are accepted for signature parity and ignored;
 * rows are independent and run in parallel, the terms of one row are combined sequentially in storage order (what DGL's CPU
 * kernels do), so results do not depend on the thread count.  Status codes and mgx_cpu_last_error() as in mi355x_graph.h.
 */
#ifndef MI355X_GRAPH_CPU_H_
#define MI355X_GRAPH_CPU_H_

#include "mi355x_graph.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* mgx_cpu_last_error(void);
int32_t mgx_cpu_num_threads(void);          /* OpenMP threads a call will use */
void mgx_cpu_set_num_threads(int32_t n);

/* mgx_spmm_csr: out[v,k] = dst_scale[v] * REDUCE_{p in row v} op(src_scale[u] * U[u, u_off[k]], E[eid(p), e_off[k]]) */
int32_t mgx_cpu_spmm_csr(const mgx_csr* csr, const mgx_spmm_plan* plan, int32_t op, int32_t reduce,
                         const float* ufeat, const float* efeat, int64_t u_len, int64_t e_len, int64_t out_len,
                         const int64_t* u_off, const int64_t* e_off, const float* src_scale, const float* dst_scale,
                         float* out, void* arg_u, void* arg_e, float* partial_ws, int32_t flags, void* stream);

/* mgx_sddmm_coo / mgx_sddmm_csr: out[e] = op(lhs[t_l(e)], rhs[t_r(e)]) by edge id */
int32_t mgx_cpu_sddmm_coo(int64_t num_src, int64_t num_dst, int64_t nnz, const void* src, const void* dst, int32_t idx_bits,
                          int32_t op, const float* lhs, const float* rhs, int32_t lhs_target, int32_t rhs_target,
                          int64_t l_len, int64_t r_len, int64_t out_len, int64_t reduce_size,
                          const int64_t* l_off, const int64_t* r_off, float* out, void* stream);
int32_t mgx_cpu_sddmm_csr(const mgx_csr* csr, const mgx_spmm_plan* plan, int32_t op, const float* lhs, const float* rhs,
                          int32_t lhs_target, int32_t rhs_target, int64_t l_len, int64_t r_len, int64_t out_len,
                          int64_t reduce_size, const int64_t* l_off, const int64_t* r_off, float* out, void* stream);

/* mgx_edge_softmax_fwd / _bwd: per destination row and head, max -> exp -> sum -> divide */
int32_t mgx_cpu_edge_softmax_fwd(const mgx_csr* csr, const mgx_spmm_plan* plan, int64_t H, const float* z, float* a,
                                 float* ws, void* stream);
int32_t mgx_cpu_edge_softmax_bwd(const mgx_csr* csr, const mgx_spmm_plan* plan, int64_t H, const float* a, const float* da,
                                 float* dz, float* ws, void* stream);

/* mgx_segment_reduce: out[s,:] = REDUCE over rows [offsets[s], offsets[s+1]) of x */
int32_t mgx_cpu_segment_reduce(int64_t num_segments, const int64_t* offsets, int64_t D, int32_t reduce, const float* x,
                               float* out, int64_t* arg, void* stream);

/* formats (integer work, bit-exact with the device versions) */
int32_t mgx_cpu_coo_to_csr(int64_t num_rows, int64_t nnz, const void* row, const void* col, int32_t idx_bits,
                           void* indptr, void* indices, void* eids, void* workspace, int64_t workspace_bytes, void* stream);
int32_t mgx_cpu_csr_transpose(const mgx_csr* csr, void* indptr_t, void* indices_t, void* eids_t,
                              void* workspace, int64_t workspace_bytes, void* stream);
int32_t mgx_cpu_csr_degrees(int64_t num_rows, const void* indptr, int32_t idx_bits, void* deg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_GRAPH_CPU_H_ */
