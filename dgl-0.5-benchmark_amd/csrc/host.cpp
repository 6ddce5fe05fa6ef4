// host.cpp -- host-side pieces of the C ABI: error text, device query, CPU-pointer COO->CSR.
//
// mgx_coo_to_csr_host serves graphs prepared on the CPU before .to(device)
// (g.int().formats(['csr','csc']).to(device), main_dgl_product_sage.py:158): a stable counting
// sort, parallel over row ranges so the 123.7M-edge products graph converts in well under a second
// per format on the host cores.
#include <stdarg.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "common.h"

namespace mgx {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

template <typename Idx>
static void coo_to_csr_host_impl(int64_t n_rows, int64_t nnz, const Idx* row, const Idx* col, Idx* indptr,
                                 Idx* indices, Idx* eids) {
  // Threads own contiguous EDGE ranges; per-thread row histograms make the placement stable:
  // edge e of thread t lands after all edges of the same row owned by threads < t.
  int T = (int)std::min<int64_t>(std::max(1u, std::thread::hardware_concurrency()), 16);
  if (nnz < (1 << 16) || (int64_t)T * n_rows > (int64_t(1) << 28)) T = 1;
  std::vector<std::vector<int64_t>> hist(T, std::vector<int64_t>((size_t)n_rows + 1, 0));
  auto range = [&](int t) { return std::make_pair(nnz * t / T, nnz * (t + 1) / T); };
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        auto r = range(t);
        auto& h = hist[t];
        for (int64_t e = r.first; e < r.second; ++e) h[(size_t)row[e]]++;
      });
    for (auto& x : th) x.join();
  }
  int64_t run = 0;
  for (int64_t r = 0; r < n_rows; ++r) {
    indptr[r] = (Idx)run;
    for (int t = 0; t < T; ++t) {
      const int64_t c = hist[t][(size_t)r];
      hist[t][(size_t)r] = run;  // becomes thread t's cursor for row r
      run += c;
    }
  }
  indptr[n_rows] = (Idx)run;
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        auto r = range(t);
        auto& cur = hist[t];
        for (int64_t e = r.first; e < r.second; ++e) {
          const int64_t p = cur[(size_t)row[e]]++;
          indices[p] = col[e];
          eids[p] = (Idx)e;
        }
      });
    for (auto& x : th) x.join();
  }
}
}  // namespace mgx

extern "C" const char* mgx_last_error(void) { return mgx::g_err; }

extern "C" int32_t mgx_abi_version(void) { return 35; }

namespace mgx {
static thread_local const char* g_last_spmm_kernel = "";
void note_spmm_kernel(const char* name) { g_last_spmm_kernel = name; }
}  // namespace mgx
extern "C" const char* mgx_last_spmm_kernel(void) { return mgx::g_last_spmm_kernel; }

extern "C" int32_t mgx_device_info(int32_t* num_cus, int32_t* lds_bytes_per_cu, char* arch_name, int32_t arch_name_len) {
  using namespace mgx;
  int dev = 0;
  MGX_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  MGX_CHECK_HIP(hipGetDeviceProperties(&p, dev));
  if (num_cus) *num_cus = p.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int32_t)p.maxSharedMemoryPerMultiProcessor;
  if (arch_name && arch_name_len > 0) {
    strncpy(arch_name, p.gcnArchName, (size_t)arch_name_len - 1);
    arch_name[arch_name_len - 1] = 0;
  }
  return MGX_OK;
}

extern "C" int32_t mgx_coo_to_csr_host(int64_t num_rows, int64_t nnz, const void* row, const void* col, int32_t idx_bits,
                                       void* indptr, void* indices, void* eids) {
  using namespace mgx;
  MGX_CHECK_ARG(idx_bits == 32 || idx_bits == 64, "mgx_coo_to_csr_host: idx_bits must be 32 or 64");
  MGX_CHECK_ARG(num_rows >= 0 && nnz >= 0, "mgx_coo_to_csr_host: negative sizes");
  MGX_CHECK_ARG(indptr != nullptr, "mgx_coo_to_csr_host: indptr is NULL");
  MGX_CHECK_ARG(nnz == 0 || (row && col && indices && eids), "mgx_coo_to_csr_host: NULL pointer");
  MGX_CHECK_ARG(idx_bits == 64 || (nnz < (int64_t(1) << 31) && num_rows < (int64_t(1) << 31)), "mgx_coo_to_csr_host: sizes overflow int32");
  if (idx_bits == 32)
    coo_to_csr_host_impl<int32_t>(num_rows, nnz, (const int32_t*)row, (const int32_t*)col, (int32_t*)indptr, (int32_t*)indices, (int32_t*)eids);
  else
    coo_to_csr_host_impl<int64_t>(num_rows, nnz, (const int64_t*)row, (const int64_t*)col, (int64_t*)indptr, (int64_t*)indices, (int64_t*)eids);
  return MGX_OK;
}
