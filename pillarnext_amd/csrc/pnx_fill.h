// pnx_fill.h -- zero-fill of the pillar-free cells of the dense NHWC canvas (device code shared by reader.hip's stand-alone
// k_canvas_fill_nhwc and the fused PFN+fill launch of pfn_v3.hip).
//
// Direct mode of the reader (sparse_resnet.py:63-68 .dense() semantics, written BEFORE a dense backbone): the PFN kernel stores
// every pillar's 64 features straight into its cell; this code writes the zeros of all OTHER cells and the occupancy bytes.
// Together they write each canvas byte exactly once.  One call = one 32x32-cell tile; `s_word` is 32 words of LDS.
#pragma once
#include "pnx_common.h"

// NT: nontemporal stores.  Measured (C2 sweep, bf16): 2 GB canvas (8 frames) 356 -> 335 us = 6.0 TB/s; 1 GB (4 frames) 171 -> 189 us,
// where a quarter of the footprint is absorbed by the 256 MB Infinity Cache that the nontemporal hint bypasses -- the host picks.
template <int DT, bool NT>
__device__ __forceinline__ void pnx_fill_tile(const uint32_t* __restrict__ bitmap, const PnxGeomDev& g, void* __restrict__ canvas,
                                              uint8_t* __restrict__ occ, int tile, uint32_t* s_word, int t, int nthreads) {
  constexpr int ESZ = (DT == PNX_F32) ? 4 : 2;
  constexpr int CH = 64 * ESZ / 16;  // 16-byte chunks per cell
  const int tiles_x = (g.gx + 31) >> 5, tiles_y = g.gyp >> 5;
  const int tx = tile % tiles_x;
  tile /= tiles_x;
  const int ty = tile % tiles_y, b = tile / tiles_y;
  const int x0 = tx << 5, y0 = ty << 5;
  if (t < 32) {
    const int xi = x0 + t;
    s_word[t] = xi < g.gx ? bitmap[((b * g.gx + xi) * g.gyp + y0) >> 5] : 0xFFFFFFFFu;
  }
  __syncthreads();
  uint4* out = reinterpret_cast<uint4*>(canvas);
  const int rows = min(32, g.gy - y0);
  // occupancy bytes: 16 cells = one 16-byte store (a byte store per cell from every 8th lane of the loop below cost 60 us of the
  // 400 us at 8 frames); tiles cut by the right edge, an unaligned row pitch or an unaligned buffer keep the per-cell stores
  const bool occ_wide = occ != nullptr && x0 + 32 <= g.gx && (g.gx & 15) == 0 && (reinterpret_cast<uintptr_t>(occ) & 15) == 0;
  if (occ_wide && t < rows * 2) {
    const int yl = t >> 1, half = t & 1;
    uint32_t w4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t v = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) v |= ((s_word[half * 16 + k * 4 + i] >> yl) & 1u) << (8 * i);
      w4[k] = v;
    }
    const int64_t cell = ((int64_t)b * g.gy + (y0 + yl)) * g.gx + x0 + half * 16;
    *reinterpret_cast<uint4*>(occ + cell) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
  }
  for (int idx = t; idx < rows * 32 * CH; idx += nthreads) {
    const int q = idx % CH;
    const int xl = (idx / CH) & 31;
    const int yl = idx / (CH * 32);
    const int xi = x0 + xl;
    if (xi >= g.gx) continue;
    const uint32_t bit = (s_word[xl] >> yl) & 1u;
    const int64_t cell = ((int64_t)b * g.gy + (y0 + yl)) * g.gx + xi;
    if (occ != nullptr && !occ_wide && q == 0) occ[cell] = (uint8_t)bit;
    if (!bit) {
      if (NT) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(u32x4{0u, 0u, 0u, 0u}, reinterpret_cast<u32x4*>(out + cell * CH + q));
      } else {
        out[cell * CH + q] = make_uint4(0, 0, 0, 0);
      }
    }
  }
}

static inline int pnx_fill_tiles(const PnxGeomDev& g) { return ((g.gx + 31) / 32) * (g.gyp / 32) * g.B; }

// A share of the zero-fill carried by extra blocks of another launch.  The fill is HBM-write bound and needs next to no ALU; the
// reader's grouping kernels are latency bound and leave HBM idle, the PFN is MFMA bound -- so blocks [n_main, gridDim) of those
// launches take the tiles [base, base + quota) one by one from `counter` (zeroed with the reader's other counters) while the
// launch's own blocks do their work.  Every tile belongs to exactly one launch: each canvas byte is still written exactly once.
struct PnxFillJob {
  const uint32_t* bitmap;
  void* canvas;
  uint8_t* occ;
  int32_t* counter;
  int base, quota;  // tiles [base, base + quota)
  int dt, nt;       // canvas dtype, nontemporal stores
  int n_main;       // first fill block of the launch (quota == 0: no fill blocks)
};

template <int DT>
__device__ __forceinline__ void pnx_fill_share_dt(const PnxFillJob& j, const PnxGeomDev& g, uint32_t* s_word /* 33 words */, int t, int nthreads) {
  for (;;) {
    if (t == 0) s_word[32] = (uint32_t)atomicAdd(j.counter, 1);
    __syncthreads();
    const int k = (int)s_word[32];
    if (k >= j.quota) break;  // block-uniform
    if (j.nt) pnx_fill_tile<DT, true>(j.bitmap, g, j.canvas, j.occ, j.base + k, s_word, t, nthreads);
    else pnx_fill_tile<DT, false>(j.bitmap, g, j.canvas, j.occ, j.base + k, s_word, t, nthreads);
    __syncthreads();  // s_word is rewritten by the next tile
  }
}
__device__ __forceinline__ void pnx_fill_share(const PnxFillJob& j, const PnxGeomDev& g, uint32_t* s_word, int t, int nthreads) {
  if (j.dt == PNX_F32) pnx_fill_share_dt<PNX_F32>(j, g, s_word, t, nthreads);
  else if (j.dt == PNX_BF16) pnx_fill_share_dt<PNX_BF16>(j, g, s_word, t, nthreads);
  else pnx_fill_share_dt<PNX_F16>(j, g, s_word, t, nthreads);
}

// ---- the same tiles, read from an occupancy BYTE map in canvas order (cell = (b*gy + yi)*gx + xi, one byte per cell, 0 / 1): what
// k_chunk_sort (chunk_sort.hip) leaves behind -- normally straight in the caller's occupancy output, so nothing is written here
// beside the zeros.  `s_row` is 32 words of LDS: bit x of word y = cell (x0 + x, y0 + y) is occupied (or outside the grid).
template <int DT, bool NT>
__device__ __forceinline__ void pnx_fill_tile_bytes(const uint8_t* __restrict__ bytemap, const PnxGeomDev& g, void* __restrict__ canvas, int tile,
                                                    uint32_t* s_row, int t, int nthreads) {
  constexpr int ESZ = (DT == PNX_F32) ? 4 : 2;
  constexpr int CH = 64 * ESZ / 16;  // 16-byte chunks per cell
  const int tiles_x = (g.gx + 31) >> 5, tiles_y = (g.gy + 31) >> 5;
  const int tx = tile % tiles_x;
  tile /= tiles_x;
  const int ty = tile % tiles_y, b = tile / tiles_y;
  const int x0 = tx << 5, y0 = ty << 5;
  const int rows = min(32, g.gy - y0);
  if (t < 256) {  // thread -> (row t >> 3, cells 4 (t & 7) .. + 3); the eight lanes of a row OR their nibbles together
    const int yl = t >> 3, part = t & 7, xs = x0 + 4 * part;
    uint32_t nib = 0xFu;
    if (yl < rows && bytemap == nullptr) {  // unconditional tile (a pre-fill ahead of the grouping kernels): only the grid's edge masks
      nib = 0u;
#pragma unroll
      for (int i = 0; i < 4; i++) nib |= ((xs + i < g.gx) ? 0u : 1u) << i;
    } else if (yl < rows) {
      const int64_t at = ((int64_t)b * g.gy + (y0 + yl)) * g.gx + xs;
      if (xs + 4 <= g.gx && ((reinterpret_cast<uintptr_t>(bytemap) + (uintptr_t)at) & 3) == 0) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(bytemap + at);
        nib = ((v * 0x00204081u) >> 21) & 0xFu;  // 4 bytes (0/1) -> 4 bits
      } else {
        nib = 0u;
#pragma unroll
        for (int i = 0; i < 4; i++) nib |= ((xs + i < g.gx) ? (uint32_t)(bytemap[at + i] & 1u) : 1u) << i;
      }
    }
    uint32_t w = nib << (4 * part);
    w |= (uint32_t)__shfl_xor((int)w, 1);
    w |= (uint32_t)__shfl_xor((int)w, 2);
    w |= (uint32_t)__shfl_xor((int)w, 4);
    if (part == 0) s_row[yl] = w;
  }
  __syncthreads();
  uint4* out = reinterpret_cast<uint4*>(canvas);
  for (int idx = t; idx < rows * 32 * CH; idx += nthreads) {
    const int q = idx % CH;
    const int xl = (idx / CH) & 31;
    const int yl = idx / (CH * 32);
    if ((s_row[yl] >> xl) & 1u) continue;
    const int64_t cell = ((int64_t)b * g.gy + (y0 + yl)) * g.gx + (x0 + xl);
    if (NT) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(u32x4{0u, 0u, 0u, 0u}, reinterpret_cast<u32x4*>(out + cell * CH + q));
    } else {
      out[cell * CH + q] = make_uint4(0, 0, 0, 0);
    }
  }
}

static inline int pnx_fill_tiles_bytes(const PnxGeomDev& g) { return ((g.gx + 31) / 32) * ((g.gy + 31) / 32) * g.B; }

struct PnxByteFillJob {
  const uint8_t* bytemap;  // null: every cell of the tile is zeroed
  void* canvas;
  int32_t* counter;  // zero before the launch
  int tiles, nt;     // tiles [base, base + tiles); nt bit 0: nontemporal stores, bit 1: raised wave priority (k_canvas_fill_bytes)
  int base;
};

// persistent: tiles by ticket, in canvas order (a compact front of neighbouring tiles: 6.5-6.8 TB/s alone against 5.6 for one block
// per tile or a static deal -- tools/microbench/fill_bw.hip)
template <int DT>
__device__ __forceinline__ void pnx_fill_bytes_share(const PnxByteFillJob& j, const PnxGeomDev& g, uint32_t* s_row /* 33 words */, int t, int nthreads) {
  for (;;) {
    if (t == 0) s_row[32] = (uint32_t)atomicAdd(j.counter, 1);
    __syncthreads();
    const int k = (int)s_row[32];
    if (k >= j.tiles) break;  // block-uniform
    if (j.nt & 1) pnx_fill_tile_bytes<DT, true>(j.bytemap, g, j.canvas, j.base + k, s_row, t, nthreads);
    else pnx_fill_tile_bytes<DT, false>(j.bytemap, g, j.canvas, j.base + k, s_row, t, nthreads);
    __syncthreads();  // s_row is rewritten by the next tile
  }
}
