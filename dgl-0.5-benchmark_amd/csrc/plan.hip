// plan.hip -- builds the work-item tables of an mgx_spmm_plan on the device (hub-row splitting over an optional row
// order), so that a C / C++ consumer of the ABI gets the schedule without the Python host layer.
//
// Two phases because the table sizes depend on the degrees:
//   mgx_spmm_plan_count  per-row chunk counts -> three exclusive scans (rocPRIM) kept in the workspace; totals
//                        {num_items, num_hubs, num_slots} are written to a device int64[3] for the caller to read;
//   mgx_spmm_plan_fill   writes item_row / item_beg / item_end / item_node / hub_row / hub_slot_ptr / slot_item into
//                        caller-allocated arrays, using the scans left in the SAME workspace.
// Integer work, deterministic (no atomics): items follow the row order, chunks of a row are consecutive, slots are
// numbered in item order -- the layout spmm.hip / softmax.hip / sddmm.hip expect.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace mgx {

static inline size_t align256p(size_t x) { return (x + 255) / 256 * 256; }

template <typename Idx>
__global__ __launch_bounds__(kBlock) void plan_count_kernel(const Idx* indptr, const Idx* order, int64_t n, int64_t split,
                                                            int64_t* nchunk, int64_t* hubflag, int64_t* nslot) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= n; i += (int64_t)gridDim.x * kBlock) {
    int64_t c = 0;
    if (i < n) {
      const int64_t r = order ? (int64_t)order[i] : i;
      const int64_t deg = (int64_t)indptr[r + 1] - (int64_t)indptr[r];
      c = (deg + split - 1) / split;
      if (c < 1) c = 1;
    }
    nchunk[i] = c;  // element n = 0: its exclusive-scan slot receives the total
    hubflag[i] = c > 1 ? 1 : 0;
    nslot[i] = c > 1 ? c : 0;
  }
}

__global__ void plan_totals_kernel(const int64_t* item_off, const int64_t* hub_off, const int64_t* slot_off, int64_t n,
                                   int64_t* totals) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    totals[0] = item_off[n];
    totals[1] = hub_off[n];
    totals[2] = slot_off[n];
  }
}

template <typename Idx>
__global__ __launch_bounds__(kBlock) void plan_fill_kernel(const Idx* indptr, const Idx* order, int64_t n, int64_t split,
                                                           const int64_t* item_off, const int64_t* hub_off,
                                                           const int64_t* slot_off, int32_t* item_row, Idx* item_beg,
                                                           Idx* item_end, int32_t* item_node, int32_t* hub_row,
                                                           int32_t* hub_slot_ptr, int32_t* slot_item) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = order ? (int64_t)order[i] : i;
    const int64_t rb = (int64_t)indptr[r], re = (int64_t)indptr[r + 1];
    const int64_t first = item_off[i], chunks = item_off[i + 1] - first;
    const bool hub = chunks > 1;
    const int64_t s0 = slot_off[i];
    if (hub) {
      const int64_t h = hub_off[i];
      hub_row[h] = (int32_t)r;
      hub_slot_ptr[h] = (int32_t)s0;
      if (hub_off[i + 1] == hub_off[n]) hub_slot_ptr[hub_off[n]] = (int32_t)slot_off[n];  // last hub closes the table
    }
    for (int64_t c = 0; c < chunks; ++c) {
      const int64_t it = first + c;
      const int64_t b = rb + c * split;
      item_beg[it] = (Idx)b;
      item_end[it] = (Idx)(b + split < re ? b + split : re);
      item_node[it] = (int32_t)r;
      if (hub) {
        item_row[it] = (int32_t)(-(s0 + c + 1));
        slot_item[s0 + c] = (int32_t)it;
      } else {
        item_row[it] = (int32_t)r;
      }
    }
  }
}

static hipError_t scan_temp_bytes(int64_t n, size_t* bytes) {
  *bytes = 0;
  return rocprim::exclusive_scan((void*)nullptr, *bytes, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)n,
                                 rocprim::plus<int64_t>(), nullptr, false);
}

struct PlanWs {
  int64_t *nchunk, *hubflag, *nslot, *item_off, *hub_off, *slot_off;
  void* temp;
  size_t temp_bytes;
};

static int32_t carve(void* ws, int64_t ws_bytes, int64_t n, PlanWs* out) {
  size_t temp = 0;
  MGX_CHECK_HIP(scan_temp_bytes(n + 1, &temp));
  const size_t arr = align256p(sizeof(int64_t) * (size_t)(n + 1));
  MGX_CHECK_ARG(ws != nullptr && (size_t)ws_bytes >= 6 * arr + temp, "mgx_spmm_plan: workspace too small (%lld < %lld)",
                (long long)ws_bytes, (long long)(6 * arr + temp));
  char* p = (char*)ws;
  out->nchunk = (int64_t*)p; p += arr;
  out->hubflag = (int64_t*)p; p += arr;
  out->nslot = (int64_t*)p; p += arr;
  out->item_off = (int64_t*)p; p += arr;
  out->hub_off = (int64_t*)p; p += arr;
  out->slot_off = (int64_t*)p; p += arr;
  out->temp = p;
  out->temp_bytes = temp;
  return MGX_OK;
}

static inline unsigned plan_grid(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > 256 * 16) b = 256 * 16;
  return (unsigned)b;
}

}  // namespace mgx

extern "C" int64_t mgx_spmm_plan_workspace(int64_t num_rows) {
  using namespace mgx;
  MGX_ENTER();
  if (num_rows < 0) return -1;
  size_t temp = 0;
  if (scan_temp_bytes(num_rows + 1, &temp) != hipSuccess) {
    set_error("mgx_spmm_plan_workspace: rocPRIM size query failed");
    return -1;
  }
  return (int64_t)(6 * align256p(sizeof(int64_t) * (size_t)(num_rows + 1)) + temp);
}

extern "C" int32_t mgx_spmm_plan_count(const mgx_csr* csr, int64_t split, const void* row_order, int64_t* totals,
                                       void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(csr != nullptr && totals != nullptr, "mgx_spmm_plan_count: NULL argument");
  MGX_CHECK_ARG(csr->idx_bits == 32 || csr->idx_bits == 64, "mgx_spmm_plan_count: idx_bits must be 32 or 64");
  MGX_CHECK_ARG(split >= 1, "mgx_spmm_plan_count: split must be >= 1");
  MGX_CHECK_ARG(csr->num_rows >= 0 && csr->num_rows < (int64_t(1) << 31), "mgx_spmm_plan_count: plans need fewer than 2^31 rows");
  const int64_t n = csr->num_rows;
  PlanWs w;
  int32_t st = carve(workspace, workspace_bytes, n, &w);
  if (st != MGX_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  if (csr->idx_bits == 32)
    hipLaunchKernelGGL((plan_count_kernel<int32_t>), dim3(plan_grid(n + 1)), dim3(kBlock), 0, s, (const int32_t*)csr->indptr,
                       (const int32_t*)row_order, n, split, w.nchunk, w.hubflag, w.nslot);
  else
    hipLaunchKernelGGL((plan_count_kernel<int64_t>), dim3(plan_grid(n + 1)), dim3(kBlock), 0, s, (const int64_t*)csr->indptr,
                       (const int64_t*)row_order, n, split, w.nchunk, w.hubflag, w.nslot);
  MGX_CHECK_LAUNCH();
  size_t tb = w.temp_bytes;
  MGX_CHECK_HIP(rocprim::exclusive_scan(w.temp, tb, (const int64_t*)w.nchunk, w.item_off, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(), s, false));
  MGX_CHECK_HIP(rocprim::exclusive_scan(w.temp, tb, (const int64_t*)w.hubflag, w.hub_off, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(), s, false));
  MGX_CHECK_HIP(rocprim::exclusive_scan(w.temp, tb, (const int64_t*)w.nslot, w.slot_off, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(), s, false));
  hipLaunchKernelGGL(plan_totals_kernel, dim3(1), dim3(64), 0, s, (const int64_t*)w.item_off, (const int64_t*)w.hub_off,
                     (const int64_t*)w.slot_off, n, totals);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int32_t mgx_spmm_plan_fill(const mgx_csr* csr, int64_t split, const void* row_order, int32_t* item_row,
                                      void* item_beg, void* item_end, int32_t* item_node, int32_t* hub_row,
                                      int32_t* hub_slot_ptr, int32_t* slot_item, void* workspace, int64_t workspace_bytes,
                                      void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(csr != nullptr, "mgx_spmm_plan_fill: csr is NULL");
  MGX_CHECK_ARG(csr->idx_bits == 32 || csr->idx_bits == 64, "mgx_spmm_plan_fill: idx_bits must be 32 or 64");
  MGX_CHECK_ARG(split >= 1, "mgx_spmm_plan_fill: split must be >= 1");
  const int64_t n = csr->num_rows;
  if (n == 0) return MGX_OK;
  MGX_CHECK_ARG(item_row && item_beg && item_end && item_node, "mgx_spmm_plan_fill: item tables are NULL");
  PlanWs w;
  int32_t st = carve(workspace, workspace_bytes, n, &w);
  if (st != MGX_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  if (csr->idx_bits == 32)
    hipLaunchKernelGGL((plan_fill_kernel<int32_t>), dim3(plan_grid(n)), dim3(kBlock), 0, s, (const int32_t*)csr->indptr,
                       (const int32_t*)row_order, n, split, (const int64_t*)w.item_off, (const int64_t*)w.hub_off,
                       (const int64_t*)w.slot_off, item_row, (int32_t*)item_beg, (int32_t*)item_end, item_node, hub_row,
                       hub_slot_ptr, slot_item);
  else
    hipLaunchKernelGGL((plan_fill_kernel<int64_t>), dim3(plan_grid(n)), dim3(kBlock), 0, s, (const int64_t*)csr->indptr,
                       (const int64_t*)row_order, n, split, (const int64_t*)w.item_off, (const int64_t*)w.hub_off,
                       (const int64_t*)w.slot_off, item_row, (int64_t*)item_beg, (int64_t*)item_end, item_node, hub_row,
                       hub_slot_ptr, slot_item);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
