// packed_proto.hip -- EXPERIMENT (round 5): bound, then prototype, of the "compressed row" gather for the D = 64
// forward aggregations whose input is a relu + dropout output (75 % exact zeros).  NOT part of libmi355x_graph.so.
//
// A source row travels as ONE 128-byte slot instead of 256 bytes (one cache line per edge instead of two):
//   lane i of the 8 lanes that fetch a slot (16 B each) gets { meta, v0, v1, v2 }: meta = c0 | c1 << 8 | c2 << 16 | flag << 24, the columns
//   of its three values (64 = none, value 0), flag = 255 in EVERY lane of a row with more than 24 non-zeros (the dense row is gathered
//   instead).  24 values per slot; Binomial(64, 1/4) exceeds 24 for 0.7 % of the rows.
// Kernel = the row-per-wave schedule of spmm_rowwave32_kernel (plan items, per-XCD stretches, ids as byte offsets through ds_bpermute,
// U gathers in flight), template MODE:
//   0  gather floor: the slots are fetched and summed as they are (meaningless numbers; = what one line per edge costs)
//   1  + the instruction mix of a register-side unpack (VERDICT's (b): four ds_bpermute + prefix pop-counts per edge and lane group),
//      synthetic: same issue slots, meaningless numbers
//   2  the real thing: every lane adds its three values into a per-lane-group accumulator row in LDS (ds_add_f32 at column address;
//      columns of one row are distinct, lane groups own different rows: no two lanes of an instruction meet), the eight rows are summed
//      per output column in the epilogue.  Fixed summation order.
//   3  the same accumulator rows updated by plain LDS read - add - write (three reads, then three writes per lane and source row: the
//      columns of one row are distinct, the padding column 64 is skipped), no atomics
//   4  mode 2 with the padding column skipped (are the 8 lanes of a row that all add 0 to column 64 what serialises the atomics?)
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int kWave = 64, kBlock = 256, kWaves = 4, kXcds = 8;
constexpr int kAccStride = 72;  // floats per lane-group accumulator row: 64 columns + the dummy column 64 (+ pad): bank = (8 g + c) % 64

typedef float v4f __attribute__((ext_vector_type(4)));

struct Args {
  const int32_t* item_row;
  const int32_t* item_beg;
  const int32_t* item_end;
  const int64_t* xcd_start;  // [9]
  const int32_t* indices;
  const char* slots;     // [n_src, 128 B]
  const float* dense;    // [n_src, ld] the unpacked matrix (overflow rows)
  float* out;            // [n_dst, ldo]
  float* partial;        // [slots, 64]
  int rpb, ld, ldo, mean;
};

template <int MODE, int U>
__global__ __launch_bounds__(kBlock) void packed_spmm_kernel(const Args a) {
  __shared__ float acc_all[kWaves][8][kAccStride];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = lane >> 3, gi = lane & 7;
  float* accw = &acc_all[wave][0][0];
  float* accg = accw + grp * kAccStride;
  if (MODE >= 2) {
    for (int i = lane; i < 8 * kAccStride; i += 64) accw[i] = 0.f;
  }
  const int xcd = blockIdx.x % kXcds;
  int64_t item_base = a.xcd_start[xcd], item_stop = a.xcd_start[xcd + 1];
  item_base += (int64_t)(blockIdx.x / kXcds) * a.rpb;
  int r = wave;
  if (r >= a.rpb || item_base + r >= item_stop) return;
  const uint32_t laneoff = (uint32_t)gi * 16u;

  int64_t item = item_base + r;
  int32_t row = a.item_row[item], beg = a.item_beg[item], end = a.item_end[item];
  uint32_t goff = 0;
  {
    const int32_t q = beg + lane;
    if (q < end) goff = (uint32_t)__builtin_nontemporal_load(&a.indices[q]) * 128u;
  }
  for (;;) {
    const int rn = r + kWaves;
    const bool has_next = rn < a.rpb && item_base + rn < item_stop;
    int32_t nrow = 0, nbeg = 0, nend = 0;
    uint32_t ngoff = 0;
    if (has_next) {
      nrow = a.item_row[item_base + rn]; nbeg = a.item_beg[item_base + rn]; nend = a.item_end[item_base + rn];
      const int32_t q = nbeg + lane;
      if (q < nend) ngoff = (uint32_t)__builtin_nontemporal_load(&a.indices[q]) * 128u;
    }
    v4f accv = (v4f)(0.f);
    for (int32_t cbase = beg; cbase < end; cbase += kWave) {
      if (cbase != beg) {
        const int32_t q = cbase + lane;
        goff = 0;
        if (q < end) goff = (uint32_t)__builtin_nontemporal_load(&a.indices[q]) * 128u;
      }
      const int cnt = (end - cbase) < kWave ? (end - cbase) : kWave;
      for (int k = 0; k < cnt; k += 8 * U) {
        v4f val[U];
        uint32_t roff[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = k + u * 8 + grp;
          roff[u] = (uint32_t)__builtin_amdgcn_ds_bpermute((e < cnt ? e : 0) * 4, (int)goff);
          val[u] = *reinterpret_cast<const v4f*>(a.slots + roff[u] + laneoff);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = k + u * 8 + grp < cnt;
          if (MODE == 0) {
            if (live) accv += val[u];
          } else if (MODE == 1) {
            // the issue slots of a register-side unpack: a 64-bit mask per row -> prefix pop-count per lane, then FOUR values fetched
            // from the lanes that hold them and selected by the lane's own four mask bits
            const uint32_t mlo = (uint32_t)__builtin_amdgcn_ds_bpermute((lane & ~7) * 4, __float_as_int(val[u].x));
            const uint32_t mhi = (uint32_t)__builtin_amdgcn_ds_bpermute((lane & ~7) * 4, __float_as_int(val[u].y));
            const uint32_t below = gi < 4 ? (mlo & ((1u << (gi * 8)) - 1u)) : mlo;
            const uint32_t belowh = gi < 4 ? 0u : (mhi & ((1u << ((gi - 4) * 8)) - 1u));
            const int start = __popc(below) + __popc(belowh);
            const uint32_t mine = ((gi < 4 ? mlo : mhi) >> ((gi & 3) * 8)) & 0xffu;
            v4f got;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int idx = start + __popc(mine & ((1u << c) - 1u)) + 2;
              const float fetched = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane & ~7) + ((idx >> 2) & 7)) * 4, __float_as_int(val[u][c])));
              got[c] = ((mine >> c) & 1u) ? fetched : 0.f;
            }
            if (live) accv += got;
          } else if (MODE == 3) {
            const uint32_t meta = (uint32_t)__float_as_int(val[u].x);
            if (live) {
              if ((meta >> 24) != 255u) {
                const uint32_t c0 = meta & 0xffu, c1 = (meta >> 8) & 0xffu, c2 = (meta >> 16) & 0xffu;
                const float a0 = accg[c0], a1 = accg[c1], a2 = accg[c2];
                if (c0 < 64u) accg[c0] = a0 + val[u].y;
                if (c1 < 64u) accg[c1] = a1 + val[u].z;
                if (c2 < 64u) accg[c2] = a2 + val[u].w;
              } else {
                const float* dr = a.dense + (size_t)(roff[u] >> 7) * a.ld + gi * 8;
                const v4f d0 = *reinterpret_cast<const v4f*>(dr), d1 = *reinterpret_cast<const v4f*>(dr + 4);
                v4f* ar = reinterpret_cast<v4f*>(accg + gi * 8);
                ar[0] += d0;
                ar[1] += d1;
              }
            }
          } else if (MODE == 4) {
            const uint32_t meta = (uint32_t)__float_as_int(val[u].x);
            if (live) {
              if ((meta >> 24) != 255u) {
                const uint32_t c0 = meta & 0xffu, c1 = (meta >> 8) & 0xffu, c2 = (meta >> 16) & 0xffu;
                if (c0 < 64u) atomicAdd(&accg[c0], val[u].y);
                if (c1 < 64u) atomicAdd(&accg[c1], val[u].z);
                if (c2 < 64u) atomicAdd(&accg[c2], val[u].w);
              } else {
                const float* dr = a.dense + (size_t)(roff[u] >> 7) * a.ld + gi * 8;
                const v4f d0 = *reinterpret_cast<const v4f*>(dr), d1 = *reinterpret_cast<const v4f*>(dr + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  atomicAdd(&accg[gi * 8 + c], d0[c]);
                  atomicAdd(&accg[gi * 8 + 4 + c], d1[c]);
                }
              }
            }
          } else {
            const uint32_t meta = (uint32_t)__float_as_int(val[u].x);
            if (live) {
              if ((meta >> 24) != 255u) {
                atomicAdd(&accg[meta & 0xffu], val[u].y);
                atomicAdd(&accg[(meta >> 8) & 0xffu], val[u].z);
                atomicAdd(&accg[(meta >> 16) & 0xffu], val[u].w);
              } else {  // more than 24 non-zeros: the dense row, 8 columns per lane
                const float* dr = a.dense + (size_t)(roff[u] >> 7) * a.ld + gi * 8;
                const v4f d0 = *reinterpret_cast<const v4f*>(dr), d1 = *reinterpret_cast<const v4f*>(dr + 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                  atomicAdd(&accg[gi * 8 + c], d0[c]);
                  atomicAdd(&accg[gi * 8 + 4 + c], d1[c]);
                }
              }
            }
          }
        }
      }
    }
    float res;
    if (MODE >= 2) {
      res = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        res += accw[g * kAccStride + lane];
        accw[g * kAccStride + lane] = 0.f;
      }
    } else {
      // 8 lane groups x 4 floats -> a 64-float row (meaningless in these modes): fold the groups with three xor-shuffles
      v4f t = accv;
#pragma unroll
      for (int off = 8; off < 64; off <<= 1) {
        t.x += __shfl_xor(t.x, off, 64); t.y += __shfl_xor(t.y, off, 64); t.z += __shfl_xor(t.z, off, 64); t.w += __shfl_xor(t.w, off, 64);
      }
      res = t[lane & 3] ;
    }
    if (a.mean && row >= 0) {
      const int deg = end - beg;
      res = res / (float)(deg > 1 ? deg : 1);
    }
    if (row >= 0) __builtin_nontemporal_store(res, a.out + (size_t)row * a.ldo + lane);
    else a.partial[(size_t)(-(row + 1)) * 64 + lane] = res;
    if (!has_next) break;
    r = rn; row = nrow; beg = nbeg; end = nend; goff = ngoff;
  }
}

// one thread per row: slot = 8 x { meta, v0, v1, v2 }
__global__ void pack_rows_kernel(int64_t n, const float* x, int ld, uint32_t* slots, unsigned long long* overflow) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const float* xr = x + r * ld;
  uint32_t out[32];
  uint8_t cols[24];
  float vals[24];
  int nnz = 0, total = 0;
  for (int c = 0; c < 64; ++c) {
    const float v = xr[c];
    if (v != 0.f) {
      if (nnz < 24) { cols[nnz] = (uint8_t)c; vals[nnz] = v; ++nnz; }
      ++total;
    }
  }
  const bool over = total > 24;
  for (int i = 0; i < 8; ++i) {
    uint32_t meta = over ? 0xff000000u : 0u;
    for (int k = 0; k < 3; ++k) {
      const int j = i * 3 + k;
      const bool have = !over && j < nnz;
      meta |= (uint32_t)(have ? cols[j] : 64) << (8 * k);
      out[i * 4 + 1 + k] = have ? __float_as_uint(vals[j]) : 0u;
    }
    out[i * 4] = meta;
  }
  for (int i = 0; i < 32; ++i) slots[r * 32 + i] = out[i];
  if (over) atomicAdd(overflow, 1ull);
}
}  // namespace

extern "C" int packed_pack(int64_t n, const float* x, int ld, void* slots, void* overflow_counter, void* stream) {
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, x, ld, (uint32_t*)slots,
                     (unsigned long long*)overflow_counter);
  return (int)hipGetLastError();
}

extern "C" int packed_spmm(int mode, int unroll, int64_t grid, const int32_t* item_row, const int32_t* item_beg, const int32_t* item_end,
                           const int64_t* xcd_start_dev, const int32_t* indices, const void* slots, const float* dense, int ld, float* out,
                           int ldo, float* partial, int rpb, int mean, void* stream) {
  Args a{item_row, item_beg, item_end, xcd_start_dev, indices, (const char*)slots, dense, out, partial, rpb, ld, ldo, mean};
  const dim3 g((unsigned)grid), b(kBlock);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(M, UU) hipLaunchKernelGGL((packed_spmm_kernel<M, UU>), g, b, 0, s, a)
  if (mode == 0 && unroll == 2) LAUNCH(0, 2);
  else if (mode == 0 && unroll == 4) LAUNCH(0, 4);
  else if (mode == 1 && unroll == 2) LAUNCH(1, 2);
  else if (mode == 1 && unroll == 4) LAUNCH(1, 4);
  else if (mode == 2 && unroll == 2) LAUNCH(2, 2);
  else if (mode == 2 && unroll == 4) LAUNCH(2, 4);
  else if (mode == 2 && unroll == 8) LAUNCH(2, 8);
  else if (mode == 3 && unroll == 2) LAUNCH(3, 2);
  else if (mode == 3 && unroll == 4) LAUNCH(3, 4);
  else if (mode == 3 && unroll == 8) LAUNCH(3, 8);
  else if (mode == 4 && unroll == 4) LAUNCH(4, 4);
  else return -1;
#undef LAUNCH
  return (int)hipGetLastError();
}
