// segment.hip -- per-segment readout (dgl.nn.AvgPooling / segment_reduce, main_dgl_molhiv_gcn.py:75,93).
// A segment reduction is a copy_rhs g-SpMM whose CSR is (offsets, identity): rows are contiguous
// slices of x, so the row-segmented wavefront kernels of spmm.hip stream them with 16-byte loads.
#include "common.h"

extern "C" int32_t mgx_segment_reduce(int64_t num_segments, const int64_t* offsets, int64_t D, int32_t reduce,
                                      const float* x, float* out, int64_t* arg, void* stream) {
  using namespace mgx;
  MGX_CHECK_ARG(num_segments >= 0 && D >= 0, "mgx_segment_reduce: negative sizes");
  MGX_CHECK_ARG(num_segments == 0 || offsets != nullptr, "mgx_segment_reduce: offsets is NULL");
  MGX_CHECK_ARG(reduce >= MGX_REDUCE_SUM && reduce <= MGX_REDUCE_MEAN, "mgx_segment_reduce: unsupported reduce op %d", reduce);
  if (num_segments == 0 || D == 0) return MGX_OK;
  mgx_csr csr;
  csr.num_rows = num_segments;
  csr.num_cols = 0;
  csr.indptr = offsets;
  csr.indices = nullptr;
  csr.eids = nullptr;
  csr.idx_bits = 64;
  csr.reserved = 0;
  // nnz drives only the SPLIT heuristic; segments are usually much longer than 64/G rows.
  csr.nnz = num_segments * 64;
  return mgx_spmm_csr(&csr, nullptr, MGX_OP_COPY_RHS, reduce, nullptr, x, 0, D, D, nullptr, nullptr, nullptr, nullptr,
                      out, nullptr, arg, nullptr, 0, stream);
}
