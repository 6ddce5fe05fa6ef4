// sample.hip -- uniform neighbor sampling without replacement on the device.
//
// Replaces what dgl.dataloading.MultiLayerNeighborSampler / dgl.sampling.sample_neighbors do in CPU worker
// processes for end_to_end/sampling/node-classification/reddit/ns-sage-dgl.py:132-141: for every seed
// (destination node) keep all in-edges when in-degree <= fanout, else draw `fanout` distinct in-edges
// uniformly.  One thread per seed runs Floyd's subset-sampling algorithm over CSR positions with a
// counter-based generator (no state, reproducible for a given rng_seed), sorts the <= 64 picks so the
// block keeps CSR order, and writes source ids and edge ids at the caller-computed offsets.  Integer work,
// no atomics; the caller sizes the outputs from min(in_degree, fanout).
#include "common.h"

namespace mgx {

constexpr int kMaxFanout = 64;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

template <typename Idx>
__global__ __launch_bounds__(kBlock) void sample_neighbors_kernel(const Idx* indptr, const Idx* indices, const Idx* eids,
                                                                  const Idx* seeds, int64_t num_seeds, int fanout,
                                                                  uint64_t rng_seed, const int64_t* out_offsets,
                                                                  Idx* out_src, Idx* out_eid) {
  const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (s >= num_seeds) return;
  const int64_t v = (int64_t)seeds[s];
  const int64_t beg = (int64_t)indptr[v];
  const int64_t deg = (int64_t)indptr[v + 1] - beg;
  int64_t o = out_offsets[s];
  if (deg <= fanout) {
    for (int64_t p = beg; p < beg + deg; ++p, ++o) {
      out_src[o] = indices[p];
      out_eid[o] = eids ? eids[p] : (Idx)p;
    }
    return;
  }
  int64_t pick[kMaxFanout];
  int n = 0;
  for (int64_t j = deg - fanout; j < deg; ++j) {  // Floyd: a uniform `fanout`-subset of [0, deg)
    const uint64_t r = mix64(rng_seed ^ mix64((uint64_t)s * 0x100000001B3ull + (uint64_t)(j - (deg - fanout))));
    int64_t t = (int64_t)(((unsigned __int128)r * (uint64_t)(j + 1)) >> 64);  // uniform in [0, j]
    bool seen = false;
    for (int i = 0; i < n; ++i) seen |= pick[i] == t;
    if (seen) t = j;
    int i = n++;  // insertion keeps the picks ascending (CSR order)
    while (i > 0 && pick[i - 1] > t) { pick[i] = pick[i - 1]; --i; }
    pick[i] = t;
  }
  for (int i = 0; i < n; ++i, ++o) {
    const int64_t p = beg + pick[i];
    out_src[o] = indices[p];
    out_eid[o] = eids ? eids[p] : (Idx)p;
  }
}

}  // namespace mgx

extern "C" int32_t mgx_sample_neighbors(const mgx_csr* csr, int64_t num_seeds, const void* seeds, int32_t fanout,
                                        uint64_t rng_seed, const int64_t* out_offsets, void* out_src, void* out_eid,
                                        void* stream) {
  using namespace mgx;
  MGX_ENTER();
  MGX_CHECK_ARG(csr != nullptr, "mgx_sample_neighbors: csr is NULL");
  MGX_CHECK_ARG(csr->idx_bits == 32 || csr->idx_bits == 64, "mgx_sample_neighbors: idx_bits must be 32 or 64");
  MGX_CHECK_ARG(num_seeds >= 0, "mgx_sample_neighbors: negative num_seeds");
  MGX_CHECK_ARG(fanout >= 1 && fanout <= kMaxFanout, "mgx_sample_neighbors: fanout must be in [1, %d], got %d", kMaxFanout, fanout);
  if (num_seeds == 0) return MGX_OK;
  MGX_CHECK_ARG(csr->indptr && seeds && out_offsets, "mgx_sample_neighbors: NULL pointer");
  MGX_CHECK_ARG(csr->nnz == 0 || (csr->indices && out_src && out_eid), "mgx_sample_neighbors: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((num_seeds + kBlock - 1) / kBlock));
  if (csr->idx_bits == 32)
    hipLaunchKernelGGL((sample_neighbors_kernel<int32_t>), grid, dim3(kBlock), 0, s, (const int32_t*)csr->indptr,
                       (const int32_t*)csr->indices, (const int32_t*)csr->eids, (const int32_t*)seeds, num_seeds, fanout,
                       rng_seed, out_offsets, (int32_t*)out_src, (int32_t*)out_eid);
  else
    hipLaunchKernelGGL((sample_neighbors_kernel<int64_t>), grid, dim3(kBlock), 0, s, (const int64_t*)csr->indptr,
                       (const int64_t*)csr->indices, (const int64_t*)csr->eids, (const int64_t*)seeds, num_seeds, fanout,
                       rng_seed, out_offsets, (int64_t*)out_src, (int64_t*)out_eid);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
