// slots.h -- the 128-byte SLOT of a mostly-zero row of 64 columns (include/mi355x_graph.h, mgx_rows_slots_pack): how a wave that holds
// four rows as 16 lanes x float4 each turns them into slots.  Shared by the stand-alone pack pass (spmm_slots.inc) and the epilogue
// of the layer GEMM that produces such rows (rowsgemm.hip), so that both write the same bytes.
#pragma once
#include "common.h"

namespace mgx {

constexpr int kSlotBytes = 128;
constexpr int kSlotValues = 24;

// Lane (sub = lane / 16, l = lane % 16) holds columns 4 l .. 4 l + 3 of the wave's row `sub` in v (zeros where the row does not exist).
// `stage`: 4 x 32 uint32 of LDS owned by this wave.  Writes the row's slot -- 8 x { meta, v0, v1, v2 }, the non-zeros in increasing
// (component, lane) order, three per group, more than 24: flag 255 in every meta -- to slot_row (NULL: no row) with one 8-byte store
// per lane; returns true in every lane of an overflow row.  Must be called by all 64 lanes.
__device__ __forceinline__ bool slot_pack_rows4(const v4f v, int sub, int l, uint32_t* stage, uint32_t* slot_row) {
  uint32_t* st = stage + sub * 32;
  uint64_t mask = 0;  // bit 16 c + l of MY row: component c of lane l is non-zero
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint64_t b = __ballot(v[c] != 0.f);
    mask |= ((b >> (sub * 16)) & 0xffffull) << (16 * c);
  }
  const bool over = __popcll(mask) > kSlotValues;
  // the slot starts as "no value anywhere": lane l owns words 2 l, 2 l + 1 (word 4 i = the meta of group i)
  st[2 * l] = (l & 1) ? 0u : (0x00404040u | (over ? 0xff000000u : 0u));
  st[2 * l + 1] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (!over) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (v[c] != 0.f) {
        const int pos = __popcll(mask & ((1ull << (16 * c + l)) - 1ull));
        const int i = pos / 3, k = pos - 3 * i;
        reinterpret_cast<uint8_t*>(st)[16 * i + k] = (uint8_t)(4 * l + c);
        st[4 * i + 1 + k] = __float_as_uint(v[c]);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (slot_row) {
    uint2 w;
    w.x = st[2 * l];
    w.y = st[2 * l + 1];
    *reinterpret_cast<uint2*>(slot_row + 2 * l) = w;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return over;
}

}  // namespace mgx
