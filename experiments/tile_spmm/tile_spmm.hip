// EXPERIMENT (not part of the library): g-SpMM copy_u/sum for DENSE neighbourhoods (reddit: 492 in-edges per node, proteins:
// 597) where the row-per-wave kernel is bound by L2 gather bandwidth (14-17 TB/s).  A workgroup owns a TILE of 64 destination
// rows that the schedule placed next to each other, stages the tile's DISTINCT source rows through LDS 64 at a time (each is
// read from L2 once per tile instead of once per edge: 2.9x fewer gathered bytes on the reddit-shaped graph at 64 rows,
// experiments/exp_tile_reuse.py) and accumulates every edge from LDS into registers: wave w owns rows 16w..16w+15 of the tile,
// one accumulator register set per row, statically indexed.
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int kR = 64;   // destination rows per tile
constexpr int kRW = 16;  // rows per wave (4 waves)
constexpr int kC = 64;   // source rows staged per chunk

struct TileArgs {
  const float* x;
  float* out;
  const int* item_tile;       // [I] work item = (tile, chunks [beg, end), whole tile?)
  const int* item_beg;
  const int* item_end;
  const uint8_t* item_whole;
  const int* chunk_src;       // [NCH * kC] source row per LDS slot, -1 = empty
  const uint16_t* counts;     // [NCH * kR] entries per (chunk, local row)
  const int* ent_ptr;         // [NCH * 4 + 1] first entry of (chunk, wave)
  const uint8_t* ent;         // [E] LDS slot per edge, ordered by (chunk, wave, local row)
  const int* tile_row;        // [T * kR] output row of (tile, local row), -1 = none
  const float* row_scale;     // optional per-output-row factor (mean)
  int64_t D;
  int I, items_per_xcd, dc;   // dc: columns per pass (multiple of VEC, <= 64 * VEC)
};

template <int VEC> struct Vec;
template <> struct Vec<1> { typedef float type; };
template <> struct Vec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int VEC>
__global__ __launch_bounds__(256) void spmm_tile_kernel(TileArgs a) {
  typedef typename Vec<VEC>::type V;
  typedef float V2 __attribute__((ext_vector_type(2)));
  extern __shared__ float lds[];
  const int bid = blockIdx.x;
  const int item = (bid % 8) * a.items_per_xcd + bid / 8;
  if (item >= a.I) return;
  const int tile = a.item_tile[item];
  const bool whole = a.item_whole[item] != 0;
  const int col0 = blockIdx.y * a.dc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dc = a.dc;
  const int64_t D = a.D;
  // staging map: float2 units, UP (power of two >= dc / 2) units per row
  constexpr int UP = 32 * VEC;            // dc <= 64 * VEC  ->  dc / 2 <= 32 * VEC
  constexpr int RPP = 256 / UP;           // rows per pass of the whole block
  const int cu = threadIdx.x % UP, r0 = threadIdx.x / UP;
  const int scol = col0 + cu * 2;
  const bool sactive = cu * 2 < dc && scol < D;
  const bool cactive = lane * VEC < dc;   // compute lanes
  V acc[kRW];
#pragma unroll
  for (int j = 0; j < kRW; ++j) acc[j] = V(0.f);

  const int cbeg = a.item_beg[item], cend = a.item_end[item];
  for (int chunk = cbeg; chunk < cend; ++chunk) {
    // ---- stage kC source rows x dc columns
    const int* cs = a.chunk_src + (int64_t)chunk * kC;
    if (sactive) {
#pragma unroll 8
      for (int r = r0; r < kC; r += RPP) {
        const int s = cs[r];
        V2 v = V2(0.f);
        if (s >= 0) v = *reinterpret_cast<const V2*>(a.x + (int64_t)s * D + scol);
        *reinterpret_cast<V2*>(lds + r * dc + cu * 2) = v;
      }
    }
    __syncthreads();
    // ---- accumulate this wave's rows
    const uint4* hp = reinterpret_cast<const uint4*>(a.counts + (int64_t)chunk * kR + wave * kRW);
    const uint4 h0 = hp[0], h1 = hp[1];
    const uint32_t hw[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    int e = a.ent_ptr[chunk * 4 + wave];
    const int eend = a.ent_ptr[chunk * 4 + wave + 1];
    int myent = (e + lane < eend) ? (int)a.ent[e + lane] : 0;
    int k = 0;
    auto next = [&]() -> int {
      if (k == 64) { e += 64; myent = (e + lane < eend) ? (int)a.ent[e + lane] : 0; k = 0; }
      return __builtin_amdgcn_readlane(myent, k++);
    };
    const float* lrow = lds + lane * VEC;
#pragma unroll
    for (int j = 0; j < kRW; ++j) {
      const int c = (hw[j >> 1] >> ((j & 1) * 16)) & 65535;
      int i = 0;
      for (; i + 1 < c; i += 2) {
        const int s0 = next(), s1 = next();
        if (cactive) {
          const V v0 = *reinterpret_cast<const V*>(lrow + s0 * dc);
          const V v1 = *reinterpret_cast<const V*>(lrow + s1 * dc);
          acc[j] += v0;
          acc[j] += v1;
        }
      }
      if (i < c) {
        const int s0 = next();
        if (cactive) acc[j] += *reinterpret_cast<const V*>(lrow + s0 * dc);
      }
    }
    __syncthreads();
  }
  // ---- write the tile's rows
  if (!cactive) return;
#pragma unroll
  for (int j = 0; j < kRW; ++j) {
    const int row = a.tile_row[tile * kR + wave * kRW + j];
    if (row < 0) continue;
    V v = acc[j];
    if (a.row_scale) v *= a.row_scale[row];
    float* o = a.out + (int64_t)row * D + col0 + lane * VEC;
    const float* pv = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int q = 0; q < VEC; ++q)
      if (col0 + lane * VEC + q < D) {
        if (whole) o[q] = pv[q];
        else atomicAdd(o + q, pv[q]);  // a tile cut into several items (hub-heavy): out was zeroed by the caller
      }
  }
}

extern "C" int tile_spmm(const float* x, float* out, const int* item_tile, const int* item_beg, const int* item_end,
                         const uint8_t* item_whole, const int* chunk_src, const uint16_t* counts,
                         const int* ent_ptr, const uint8_t* ent, const int* tile_row, const float* row_scale, int64_t D, int I,
                         int npass, void* stream) {
  TileArgs a;
  a.x = x; a.out = out; a.item_tile = item_tile; a.item_beg = item_beg; a.item_end = item_end; a.item_whole = item_whole; a.chunk_src = chunk_src; a.counts = counts; a.ent_ptr = ent_ptr;
  a.ent = ent; a.tile_row = tile_row; a.row_scale = row_scale; a.D = D; a.I = I;
  a.items_per_xcd = (I + 7) / 8;
  if (D % 2) return -1;
  int dc = (int)((D + npass - 1) / npass);
  dc = (dc + 3) / 4 * 4;
  if (dc > 256) return -2;
  a.dc = dc;
  const dim3 grid(8 * a.items_per_xcd, (unsigned)((D + dc - 1) / dc));
  const size_t shmem = (size_t)kC * dc * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (dc > 128) hipLaunchKernelGGL(spmm_tile_kernel<4>, grid, dim3(256), shmem, s, a);
  else if (dc > 64) hipLaunchKernelGGL(spmm_tile_kernel<2>, grid, dim3(256), shmem, s, a);
  else hipLaunchKernelGGL(spmm_tile_kernel<1>, grid, dim3(256), shmem, s, a);
  return (int)hipGetLastError();
}
