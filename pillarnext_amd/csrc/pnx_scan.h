// pnx_scan.h -- two-level exclusive prefix sums over uint32 arrays (device code, included by .hip files).
//   level 1  k_scan_local<MODE>: each 256-thread block scans PNX_SCAN_ITEMS items, writes the exclusive prefix
//            inside the block and the block total
//   level 2  k_scan_blocks: one block turns the totals into exclusive block offsets and publishes the grand total
// A consumer adds both: prefix(i) = blk[i >> PNX_SCAN_SHIFT] + local[i].
#pragma once
#include "pnx_common.h"

namespace {

constexpr int kBlock = 256;

enum { SCAN_POPC = 0, SCAN_IDENT = 1, SCAN_KEPT = 2, SCAN_PLUS1 = 3 };

template <int MODE>
__device__ __forceinline__ uint32_t scan_value(uint32_t v) {
  if (MODE == SCAN_POPC) return (uint32_t)__popc(v);
  if (MODE == SCAN_KEPT) return ((int32_t)v) >= 0 ? 1u : 0u;
  return v;
}

// Level 2 as a device function: one block turns the per-block totals into exclusive offsets (in place) and publishes the grand
// total (also stored at blk[nblk]).  (Folding this into the last block of the level-1 kernel was measured in round 2: the device-scope
// release fence it needs in every block writes the XCD's dirty L2 lines back, +115 us on a 255 us voxelize -- level 2 stays a launch.)
__device__ __forceinline__ void scan_blocks_body(uint32_t* blk, int nblk, int32_t* total_out) {
  __shared__ uint32_t s_wave2[kBlock / 64];
  __shared__ uint32_t s_carry;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += kBlock) {
    const int i = base + t;
    const uint32_t x = i < nblk ? __hip_atomic_load(&blk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    uint32_t inc = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t y = __shfl_up(inc, d);
      if (lane >= d) inc += y;
    }
    if (lane == 63) s_wave2[wave] = inc;
    __syncthreads();
    uint32_t woff = s_carry;
    for (int w = 0; w < wave; w++) woff += s_wave2[w];
    if (i < nblk) blk[i] = woff + inc - x;
    __syncthreads();
    if (t == kBlock - 1) s_carry = woff + inc;
    __syncthreads();
  }
  if (t == 0) {
    blk[nblk] = s_carry;
    if (total_out) *total_out = (int32_t)s_carry;
  }
}

// Level 1: each block scans PNX_SCAN_ITEMS items; out_local = exclusive prefix inside the block.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_scan_local(const uint32_t* __restrict__ in, int64_t n,
                                                       uint32_t* __restrict__ out_local, uint32_t* blk_tot,
                                                       const int32_t* __restrict__ limit = nullptr) {
  __shared__ uint32_t s_wave[kBlock / 64];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int64_t base = (int64_t)blockIdx.x * PNX_SCAN_ITEMS + (int64_t)t * 8;
  uint32_t v[8];
  if (base + 8 <= n) {
    const uint4 a = *reinterpret_cast<const uint4*>(in + base);
    const uint4 b = *reinterpret_cast<const uint4*>(in + base + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = (base + k < n) ? in[base + k] : (MODE == SCAN_KEPT ? 0xFFFFFFFFu : 0u);
  }
  uint32_t sum = 0;
  const int64_t lim = (MODE == SCAN_PLUS1) ? (int64_t)limit[0] : 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    uint32_t x = scan_value<MODE>(v[k]);
    if (MODE == SCAN_PLUS1) x = (base + k < lim) ? v[k] + 1u : 0u;  // items are "extra" counts: value+1 below the limit
    v[k] = sum;
    sum += x;
  }
  uint32_t inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(inc, d);
    if (lane >= d) inc += y;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wave; w++) woff += s_wave[w];
  const uint32_t excl = woff + inc - sum;
  if (base + 8 <= n) {
    *reinterpret_cast<uint4*>(out_local + base) = make_uint4(excl + v[0], excl + v[1], excl + v[2], excl + v[3]);
    *reinterpret_cast<uint4*>(out_local + base + 4) = make_uint4(excl + v[4], excl + v[5], excl + v[6], excl + v[7]);
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (base + k < n) out_local[base + k] = excl + v[k];
  }
  if (t == kBlock - 1) blk_tot[blockIdx.x] = excl + sum;
}

// Level 2: one block turns the per-block totals into exclusive offsets (in place) and publishes the
// grand total (also stored at blk[nblk]).
__global__ __launch_bounds__(kBlock) void k_scan_blocks(uint32_t* blk, int nblk, int32_t* total_out) { scan_blocks_body(blk, nblk, total_out); }

}  // namespace
