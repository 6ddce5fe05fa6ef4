// tile_common.h -- what the LDS-staged tile kernels (spmm_tile.hip, gat_tile.hip) share: address-space casts for LDS-DMA and for
// scalar loads of the plan tables, counted waits.
#pragma once
#include "common.h"

namespace mgx {

constexpr int kNoItem = INT32_MIN;  // tile_item of a position without a work item

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Plan tables are read-only for the whole launch: reading them through the CONSTANT address space lets a wave-uniform address
// become a scalar load (s_load_*, lgkmcnt) -- a vector load would share the in-order vmcnt queue with the LDS-DMA / the stream
// prefetches and every wait for it would drain them.
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* as_const(const T* p) {
  return (const __attribute__((address_space(4))) T*)p;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace mgx
