// reader.hip -- point cloud -> pillars -> PFN -> dense BEV canvas, for gfx950 (MI355X).
//
// What it computes (reference: det3d/models/readers/pillar_encoder.py, "pe:" below):
//   PillarNet.forward pe:78-125, PFNLayer.forward pe:35-50 (x2), PillarFeatureNet.forward pe:174-182,
//   and the dense canvas of SparseConvTensor(...).dense() (det3d/models/backbones/sparse_resnet.py:63-68).
//
// How (nothing like the reference's torch.unique + torch_scatter sequence):
//   1. k_keys        one coalesced pass over the raw point buffer: fp32 IEEE (x-min)/vs, range mask,
//                    cell key = (b*gx + xi)*gyp + yi  (gyp = gy rounded up to 32), occupancy BITMAP
//   2. scan          popcount prefix over the bitmap  -> pillar rank of every cell == torch.unique(dim=0)
//                    order, without sorting a single point
//   3. k_rank        per point: rank = prefix + popc(bits below); slot inside the pillar by an integer
//                    atomic; the slot-0 point writes coords[rank]
//   4. scan + k_fill counting sort -> CSR list of point ids per pillar
//   5. k_pfn_*       Linear(10->32)+BN+ReLU, per-pillar max, concat, Linear(64->64)+BN+ReLU, per-pillar max
//   6. k_canvas_*    32x32-cell tiles: the bitmap says which cells are empty (write zeros) and the rank
//                    says which feat_max row to convert and write -- every canvas byte is written exactly
//                    once, in 1 KiB contiguous wave-stores (NHWC), no memset pass.
//
// Determinism: pillar membership, rank, coords and the max-pool do not depend on thread timing.  The
// per-pillar mean is summed in fp64 (exact for LiDAR-range coordinates), so the atomic slot order does
// not show in the results either.
#include <vector>

#include "pnx_common.h"
#include "pnx_scan.h"
#include "pnx_fill.h"
#include "spans.h"

namespace {

using GeomDev = PnxGeomDev;

// ------------------------------------------------------------------------------------------ voxelize
// pe:91-109.  fp32 subtract then IEEE divide (never a reciprocal multiply -- SURVEY H1), compares on the
// float coordinate, truncation, key.  Rows whose batch index is outside [0,B) are dropped.
__global__ __launch_bounds__(kBlock) void k_keys(const float* __restrict__ pts, int64_t n, int stride, GeomDev g,
                                                 int32_t* __restrict__ key_out, uint8_t* __restrict__ bytemap, int32_t* __restrict__ owner) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float* p = pts + i * stride;
  const float bf = p[0], x = p[1], y = p[2];
  const float cx = __fdiv_rn(__fsub_rn(x, g.minx), g.vx);
  const float cy = __fdiv_rn(__fsub_rn(y, g.miny), g.vy);
  bool keep = (cx >= 0.f) && (cx < (float)g.gx) && (cy >= 0.f) && (cy < (float)g.gy);
  keep = keep && (bf > -1.0f) && (bf < (float)g.B);  // (long)b in [0,B): truncation maps (-1,0) to 0
  int32_t key = -1;
  if (keep) {
    const int xi = (int)cx, yi = (int)cy, bi = (int)bf;
    key = (bi * g.gx + xi) * g.gyp + yi;
    // Plain stores instead of atomics (agent-scope atomics run at ~20/ns on the fabric: measured 17 us for 300 k):
    //  - occupancy as one BYTE per cell: every writer stores the same value, byte-granular dirty masks merge across XCDs
    //  - owner[cell] = some point of the cell (last writer wins): that point takes slot 0 without an atomic
    bytemap[key] = 1;
    if (owner != nullptr) owner[key] = (int32_t)i;  // the binned path (reader_bins.h) needs no owner
  }
  key_out[i] = key;
}

// 32 occupancy bytes -> one bitmap word, fused with level 1 of the popcount scan.
__device__ __forceinline__ uint32_t nib4(uint32_t v) { return ((v * 0x00204081u) >> 21) & 0xFu; }  // 4 bytes (0/1) -> 4 bits

__global__ __launch_bounds__(kBlock) void k_pack_scan(const uint8_t* __restrict__ bytemap, int64_t nwords, uint32_t* __restrict__ bitmap,
                                                      uint32_t* __restrict__ out_local, uint32_t* blk_tot, uint2* __restrict__ wcomb) {
  __shared__ uint32_t s_wave[kBlock / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t base = (int64_t)blockIdx.x * PNX_SCAN_ITEMS + (int64_t)t * 8;
  uint32_t w[8], pre[8];
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    uint32_t bits = 0;
    if (base + k < nwords) {
      const uint4* src = reinterpret_cast<const uint4*>(bytemap + (base + k) * 32);
      const uint4 a = src[0], b = src[1];
      bits = nib4(a.x) | (nib4(a.y) << 4) | (nib4(a.z) << 8) | (nib4(a.w) << 12) | (nib4(b.x) << 16) | (nib4(b.y) << 20) |
             (nib4(b.z) << 24) | (nib4(b.w) << 28);
    }
    w[k] = bits;
    pre[k] = sum;
    sum += (uint32_t)__popc(bits);
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
  for (int q = 0; q < wave; q++) woff += s_wave[q];
  const uint32_t excl = woff + inc - sum;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (base + k < nwords) {
      bitmap[base + k] = w[k];
      out_local[base + k] = excl + pre[k];
      if (wcomb != nullptr) wcomb[base + k] = make_uint2(w[k], excl + pre[k]);  // {bits, prefix} in one 8-byte load for the per-point rank
    }
  if (t == kBlock - 1) blk_tot[blockIdx.x] = excl + sum;
}

__device__ __forceinline__ int32_t cell_rank(int32_t key, const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ wpre,
                                             const uint32_t* __restrict__ wblk) {
  const int32_t w = key >> 5;
  const uint32_t bits = bitmap[w];
  return (int32_t)(wblk[w >> PNX_SCAN_SHIFT] + wpre[w] + __popc(bits & ((1u << (key & 31)) - 1u)));
}

__device__ __forceinline__ int32_t cell_rank2(int32_t key, const uint2* __restrict__ wcomb, const uint32_t* __restrict__ wblk) {
  const int32_t w = key >> 5;
  const uint2 c = wcomb[w];
  return (int32_t)(wblk[w >> PNX_SCAN_SHIFT] + c.y + __popc(c.x & ((1u << (key & 31)) - 1u)));
}

// rank of every point (== unq_inv of the reference for kept points), slot inside the pillar, coords.
// The cell's owner point takes slot 0 (and writes coords) without an atomic; the others draw slots 1.. from
// extra[rank].  A pillar therefore holds extra[rank] + 1 points.
__global__ __launch_bounds__(kBlock) void k_rank(const int32_t* __restrict__ key, int64_t n, GeomDev g,
                                                 const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ wpre,
                                                 const uint32_t* __restrict__ wblk, const int32_t* __restrict__ owner,
                                                 int32_t* __restrict__ rank_out, int32_t* __restrict__ slot_out,
                                                 uint32_t* __restrict__ extra, int32_t* __restrict__ coords, int64_t pillar_capacity,
                                                 int32_t* __restrict__ pillar_of_point, int32_t* __restrict__ cell_of_pillar) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t k = key[i];
  int32_t r = -1;
  if (k >= 0) {
    r = cell_rank(k, bitmap, wpre, wblk);
    if (owner[k] == (int32_t)i) {
      slot_out[i] = 0;
      const int yi = k % g.gyp;
      const int t = k / g.gyp;
      const int xi = t % g.gx, bi = t / g.gx;
      cell_of_pillar[r] = (bi * g.gy + yi) * g.gx + xi;  // NHWC canvas cell of the pillar
      if (coords != nullptr && r < pillar_capacity) {
        coords[(int64_t)r * 3 + 0] = bi;  // [b, yi, xi]  (pe:125 swaps x/y)
        coords[(int64_t)r * 3 + 1] = yi;
        coords[(int64_t)r * 3 + 2] = xi;
      }
    } else {
      slot_out[i] = 1 + (int32_t)atomicAdd(&extra[r], 1u);
    }
  }
  rank_out[i] = r;
  if (pillar_of_point) pillar_of_point[i] = r;
}

__device__ __forceinline__ uint32_t pillar_start(int32_t r, const uint32_t* __restrict__ cpre, const uint32_t* __restrict__ cblk) {
  return cblk[r >> PNX_SCAN_SHIFT] + cpre[r];
}

#include "reader_bins.h"

// CSR fill: plist[start(rank) + slot] = point id, plus a 32-byte record per slot [x, y, z, f.. (6 words), idx|rem, rank] so the PFN
// kernel streams its input with coalesced loads instead of chasing plist -> rank -> point row.
__global__ __launch_bounds__(kBlock) void k_fill(const float* __restrict__ pts, int stride, const int32_t* __restrict__ rank,
                                                 const int32_t* __restrict__ slot, int64_t n, const uint32_t* __restrict__ count,
                                                 const uint32_t* __restrict__ cpre, const uint32_t* __restrict__ cblk,
                                                 int32_t* __restrict__ plist, uint32_t* __restrict__ rec) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rank[i];
  if (r < 0) return;
  const uint32_t pos = pillar_start(r, cpre, cblk) + (uint32_t)slot[i];
  plist[pos] = (int32_t)i;
  const float* p = pts + i * stride;
  uint32_t v[8];
#pragma unroll
  for (int k = 0; k < 6; k++) v[k] = (k < stride - 1) ? __float_as_uint(p[1 + k]) : 0u;
  // word 6: position inside the pillar and points still to come (both clamped to 16 bits) -> the PFN kernel knows
  // heads (idx == 0), tails (rem == 0) and pillar sizes without touching any other array
  const uint32_t idx = (uint32_t)slot[i], rem = count[r] - idx;  // count[] = size - 1
  v[6] = min(idx, 0xFFFFu) | (min(rem, 0xFFFFu) << 16);
  v[7] = (uint32_t)r;
  uint4* o = reinterpret_cast<uint4*>(rec + (int64_t)pos * 8);
  o[0] = make_uint4(v[0], v[1], v[2], v[3]);
  o[1] = make_uint4(v[4], v[5], v[6], v[7]);
}

// unq_inv (pe:110): pillar rank of the j-th kept point, j = exclusive count of kept rows before it.
__global__ __launch_bounds__(kBlock) void k_write_inv(const int32_t* __restrict__ rank, int64_t n, const uint32_t* __restrict__ kpre,
                                                      const uint32_t* __restrict__ kblk, int64_t* __restrict__ unq_inv) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rank[i];
  if (r >= 0) unq_inv[kblk[i >> PNX_SCAN_SHIFT] + kpre[i]] = (int64_t)r;
}

// Per-pillar mean of xyz (scatter_mean, pe:113-114): fp64 sum over the CSR list, fp32 divide by count.
__global__ __launch_bounds__(kBlock) void k_pillar_mean(const float* __restrict__ pts, int stride, const int32_t* __restrict__ plist,
                                                        const uint32_t* __restrict__ count, const uint32_t* __restrict__ cpre,
                                                        const uint32_t* __restrict__ cblk, const int32_t* __restrict__ counters,
                                                        float* __restrict__ mean) {
  const int32_t r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= counters[0]) return;
  const uint32_t s = pillar_start(r, cpre, cblk), c = count[r] + 1u;  // count[] holds the non-owner points
  double sx = 0, sy = 0, sz = 0;
  for (uint32_t k = 0; k < c; k++) {
    const float* p = pts + (int64_t)plist[s + k] * stride;
    sx += (double)p[1];
    sy += (double)p[2];
    sz += (double)p[3];
  }
  const float fc = (float)c;
  mean[(int64_t)r * 3 + 0] = __fdiv_rn((float)sx, fc);
  mean[(int64_t)r * 3 + 1] = __fdiv_rn((float)sy, fc);
  mean[(int64_t)r * 3 + 2] = __fdiv_rn((float)sz, fc);
}

// Decorated point features (pe:116-123): [raw F | xyz - mean | xy - pillar centre].
template <int F>
__device__ __forceinline__ void decorate(const float* __restrict__ p, const float mx, const float my, const float mz, const GeomDev& g,
                                         float* f) {
#pragma unroll
  for (int k = 0; k < F; k++) f[k] = p[1 + k];
  const float x = p[1], y = p[2], z = p[3];
  f[F + 0] = __fsub_rn(x, mx);
  f[F + 1] = __fsub_rn(y, my);
  f[F + 2] = __fsub_rn(z, mz);
  const float cx = __fdiv_rn(__fsub_rn(x, g.minx), g.vx);
  const float cy = __fdiv_rn(__fsub_rn(y, g.miny), g.vy);
  const float xi = (float)(int)cx, yi = (float)(int)cy;
  // idx*vs + vs/2 + min, each step rounded (pe:119-120)
  const float ctrx = __fadd_rn(__fadd_rn(__fmul_rn(xi, g.vx), __fdiv_rn(g.vx, 2.0f)), g.minx);
  const float ctry = __fadd_rn(__fadd_rn(__fmul_rn(yi, g.vy), __fdiv_rn(g.vy, 2.0f)), g.miny);
  f[F + 3] = __fsub_rn(x, ctrx);
  f[F + 4] = __fsub_rn(y, ctry);
}

template <int F>
__global__ __launch_bounds__(kBlock) void k_decorate(const float* __restrict__ pts, int64_t n, GeomDev g, const int32_t* __restrict__ rank,
                                                     const uint32_t* __restrict__ kpre, const uint32_t* __restrict__ kblk,
                                                     const float* __restrict__ mean, float* __restrict__ features) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rank[i];
  if (r < 0) return;
  float f[F + 5];
  decorate<F>(pts + i * (F + 1), mean[(int64_t)r * 3], mean[(int64_t)r * 3 + 1], mean[(int64_t)r * 3 + 2], g, f);
  float* o = features + (int64_t)(kblk[i >> PNX_SCAN_SHIFT] + kpre[i]) * (F + 5);
#pragma unroll
  for (int k = 0; k < F + 5; k++) o[k] = f[k];
}

// ------------------------------------------------------------------------------------------ PFN
// folded layout (include/pnx.h): W0' (32 x C0) | s0 (32) | W1' (64 x 64) | s1 (64)
template <int F>
struct Folded {
  static constexpr int C0 = F + 5;
  static constexpr int W0 = 0, S0 = 32 * C0, W1 = S0 + 32, S1 = W1 + 64 * 64;
};

// BatchNorm1d(eval) folded into the bias-free Linear in front of it (pe:32-33,37-38).
__global__ void k_fold_bn(int C0, const float* w0, const float* g0, const float* b0, const float* m0, const float* v0, const float* w1,
                          const float* g1, const float* b1, const float* m1, const float* v1, float eps, float* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int S0 = 32 * C0, W1 = S0 + 32, S1 = W1 + 4096;
  if (t < 32 * C0) {
    const int c = t / C0;
    const float a = __fmul_rn(__fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(v0[c], eps))), g0[c]);
    out[t] = __fmul_rn(w0[t], a);
  }
  if (t < 4096) {
    const int c = t >> 6;
    const float a = __fmul_rn(__fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(v1[c], eps))), g1[c]);
    out[W1 + t] = __fmul_rn(w1[t], a);
  }
  if (t < 32) {
    const float a = __fmul_rn(__fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(v0[t], eps))), g0[t]);
    out[S0 + t] = __fsub_rn(b0[t], __fmul_rn(m0[t], a));
  }
  if (t < 64) {
    const float a = __fmul_rn(__fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(v1[t], eps))), g1[t]);
    out[S1 + t] = __fsub_rn(b1[t], __fmul_rn(m1[t], a));
  }
  // Fragment-ordered copy for the MFMA PFN kernels (pfn_v3.hip, pfn_spans.hip): register j of lane l at FR + j*64 + l.
  //   j 0..5   W0'[col][2j+h]            (A/B fragment of layer 0; column C0 = s0[col] for the constant-1 feature)
  //   j 6      s0[col]
  //   j 7..22  s0[ch(i,h)]               ch(i,h) = (i&3) + 8*(i>>2) + 4h  (accumulator-register channel order)
  //   j 23..54 W1'[col][k(i,h)]          k(i,h) = (i<16 ? ch(i,h) : 32 + ch(i-16,h))
  //   j 55..86 W1'[32+col][k(i,h)]
  //   j 87, 88 s1[col], s1[32+col]
  //   then 64 x 32: s1[ch(i,h)] (i<16) | s1[32+ch(i-16,h)], contiguous per lane (tail-lane epilogue)
  // Second fragment block (pfn_v3.hip, fp16x3 layer 1), register j of lane l at FR + 64*121 + j*64 + l:
  //   j 0..6   layer-0 fragments as j 0..6 above, times 2^PNX_PFN_SU (layer 0 then leaves h0 pre-scaled for the fp16 split)
  //   j 7..70  W1' * 2^PNX_PFN_SW as fp16 hi/lo pairs in the operand order of v_mfma_f32_32x32x16_f16:
  //            j = 7 + ((part*2 + mt)*4 + s)*4 + tq, part 0 = hi (RNE), 1 = lo = fp16(x - hi); row = 32*mt + (l&31); the word packs the
  //            K slots e = 2tq (low half), 2tq+1 of K group kg = l>>5 at K step s, slot (s, kg, e) = the input channel that lane-half kg
  //            holds as its value 8s + e: ch(8s+e, kg) for s < 2 (h0), 32 + ch(8(s-2)+e, kg) for s >= 2 (pillar max)
  const int FR = S1 + 64;
  for (int idx = t; idx < 64 * 71; idx += gridDim.x * blockDim.x) {
    const int j = idx >> 6, l = idx & 63;
    const int col = l & 31, h = l >> 5;
    auto fold0 = [&](int c, int k) { return __fmul_rn(w0[c * C0 + k], __fmul_rn(__fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(v0[c], eps))), g0[c])); };
    auto fold1 = [&](int c, int k) { return __fmul_rn(w1[c * 64 + k], __fmul_rn(__fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(v1[c], eps))), g1[c])); };
    auto shift0 = [&](int c) { return __fsub_rn(b0[c], __fmul_rn(m0[c], __fmul_rn(__fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(v0[c], eps))), g0[c]))); };
    uint32_t word;
    if (j < 7) {
      float v;
      if (j < 6) {
        const int k = 2 * j + h;
        v = k < C0 ? fold0(col, k) : (k == C0 ? shift0(col) : 0.f);
      } else {
        v = shift0(col);
      }
      word = __float_as_uint(__fmul_rn(v, (float)(1 << PNX_PFN_SU)));
    } else {
      const int q = j - 7, tq = q & 3, st = (q >> 2) & 3, mt = (q >> 4) & 1, part = q >> 5;
      uint32_t hw[2];
      for (int e2 = 0; e2 < 2; e2++) {
        const int e = 2 * tq + e2, i8 = 8 * (st & 1) + e;
        const int k = (st < 2 ? 0 : 32) + (i8 & 3) + 8 * (i8 >> 2) + 4 * h;
        const float x = __fmul_rn(fold1(32 * mt + col, k), (float)(1 << PNX_PFN_SW));
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)__fsub_rn(x, (float)hi);
        hw[e2] = (uint32_t)__builtin_bit_cast(unsigned short, part ? lo : hi);
      }
      word = hw[0] | (hw[1] << 16);
    }
    out[FR + 64 * 121 + idx] = __uint_as_float(word);
  }
  for (int idx = t; idx < 64 * 121; idx += gridDim.x * blockDim.x) {
    int j = idx >> 6, l = idx & 63;
    if (idx >= 64 * 89) {  // rows 89..120: s1 in each lane's own channel order, 32 contiguous floats per lane
      l = (idx - 64 * 89) >> 5;
      j = 89 + ((idx - 64 * 89) & 31);
    }
    const int col = l & 31, h = l >> 5;
    auto fold0 = [&](int c, int k) { return __fmul_rn(w0[c * C0 + k], __fmul_rn(__fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(v0[c], eps))), g0[c])); };
    auto fold1 = [&](int c, int k) { return __fmul_rn(w1[c * 64 + k], __fmul_rn(__fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(v1[c], eps))), g1[c])); };
    auto shift0 = [&](int c) { return __fsub_rn(b0[c], __fmul_rn(m0[c], __fmul_rn(__fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(v0[c], eps))), g0[c]))); };
    auto shift1 = [&](int c) { return __fsub_rn(b1[c], __fmul_rn(m1[c], __fmul_rn(__fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(v1[c], eps))), g1[c]))); };
    float v;
    if (j < 6) {
      const int k = 2 * j + h;
      v = k < C0 ? fold0(col, k) : (k == C0 ? shift0(col) : 0.f);  // column C0 multiplies the constant-1 feature
    } else if (j == 6) {
      v = shift0(col);
    } else if (j < 23) {
      const int i = j - 7;
      v = shift0((i & 3) + 8 * (i >> 2) + 4 * h);
    } else if (j < 87) {
      const int i = (j - 23) & 31, ii = i & 15;
      const int k = (i < 16 ? 0 : 32) + (ii & 3) + 8 * (ii >> 2) + 4 * h;
      v = fold1((j < 55 ? 0 : 32) + col, k);
    } else if (j < 89) {
      v = shift1((j == 87 ? 0 : 32) + col);
    } else {
      const int i = j - 89, ii = i & 15;
      v = shift1((i < 16 ? 0 : 32) + (ii & 3) + 8 * (ii >> 2) + 4 * h);
    }
    out[FR + idx] = v;
  }
}

// ------------------------------------------------------------------------------------------ canvas
__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ uint32_t f32_to_f16_rne(float f) {
  const _Float16 hv = (_Float16)f;  // round-to-nearest-even
  return (uint32_t)__builtin_bit_cast(unsigned short, hv);
}
template <int DT>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  if (DT == PNX_BF16) return f32_to_bf16_rne(a) | (f32_to_bf16_rne(b) << 16);
  return f32_to_f16_rne(a) | (f32_to_f16_rne(b) << 16);
}

// NHWC canvas, 64 channels.  One block per 32x32-cell tile.  s_word[x] holds the 32 occupancy bits of
// column x (rows y0..y0+31 -- one aligned bitmap word thanks to the gyp padding), s_pre[x] the rank of its
// first pillar.  Each 16-byte store chunk is zero or 8 (4 for fp32) converted feat_max values.
template <int DT>
__global__ __launch_bounds__(kBlock) void k_canvas_nhwc(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ wpre,
                                                        const uint32_t* __restrict__ wblk, const float* __restrict__ g1, int64_t g1_rows,
                                                        GeomDev g, void* __restrict__ canvas, uint8_t* __restrict__ occ) {
  constexpr int ESZ = (DT == PNX_F32) ? 4 : 2;
  constexpr int CH = 64 * ESZ / 16;  // 16-byte chunks per cell
  constexpr int VPC = 16 / ESZ;      // values per chunk
  __shared__ uint32_t s_word[32], s_pre[32];
  const int tiles_x = (g.gx + 31) >> 5, tiles_y = g.gyp >> 5;
  int tile = blockIdx.x;
  const int tx = tile % tiles_x;
  tile /= tiles_x;
  const int ty = tile % tiles_y, b = tile / tiles_y;
  const int x0 = tx << 5, y0 = ty << 5;
  const int t = threadIdx.x;
  if (t < 32) {
    uint32_t word = 0, pre = 0;
    const int xi = x0 + t;
    if (xi < g.gx) {
      const int32_t w = ((b * g.gx + xi) * g.gyp + y0) >> 5;
      word = bitmap[w];
      pre = wblk[w >> PNX_SCAN_SHIFT] + wpre[w];
    }
    s_word[t] = word;
    s_pre[t] = pre;
  }
  __syncthreads();
  uint4* out = reinterpret_cast<uint4*>(canvas);
  const int rows = min(32, g.gy - y0);
  for (int idx = t; idx < rows * 32 * CH; idx += kBlock) {
    const int q = idx % CH;
    const int xl = (idx / CH) & 31;
    const int yl = idx / (CH * 32);
    const int xi = x0 + xl;
    if (xi >= g.gx) continue;
    const uint32_t word = s_word[xl];
    uint4 v = make_uint4(0, 0, 0, 0);
    if (occ != nullptr && q == 0) occ[((int64_t)b * g.gy + (y0 + yl)) * g.gx + xi] = (uint8_t)((word >> yl) & 1u);
    if ((word >> yl) & 1u) {
      const int64_t r = (int64_t)s_pre[xl] + __popc(word & ((1u << yl) - 1u));
      if (r < g1_rows) {
        const float4* src = reinterpret_cast<const float4*>(g1 + r * 64 + q * VPC);
        if (DT == PNX_F32) {
          const float4 a = src[0];
          v = make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w));
        } else {
          const float4 a = src[0], c = src[1];
          v = make_uint4(pack2<DT>(a.x, a.y), pack2<DT>(a.z, a.w), pack2<DT>(c.x, c.y), pack2<DT>(c.z, c.w));
        }
      }
    }
    out[(((int64_t)b * g.gy + (y0 + yl)) * g.gx + xi) * CH + q] = v;
  }
}

// NCHW canvas (what .dense() returns).  Not the performance layout; same tile scheme, one element per store.
template <int DT>
__global__ __launch_bounds__(kBlock) void k_canvas_nchw(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ wpre,
                                                        const uint32_t* __restrict__ wblk, const float* __restrict__ g1, int64_t g1_rows,
                                                        GeomDev g, void* __restrict__ canvas, uint8_t* __restrict__ occ) {
  __shared__ uint32_t s_word[32], s_pre[32];
  const int tiles_x = (g.gx + 31) >> 5, tiles_y = g.gyp >> 5;
  int tile = blockIdx.x;
  const int tx = tile % tiles_x;
  tile /= tiles_x;
  const int ty = tile % tiles_y, b = tile / tiles_y;
  const int x0 = tx << 5, y0 = ty << 5;
  const int t = threadIdx.x;
  if (t < 32) {
    uint32_t word = 0, pre = 0;
    const int xi = x0 + t;
    if (xi < g.gx) {
      const int32_t w = ((b * g.gx + xi) * g.gyp + y0) >> 5;
      word = bitmap[w];
      pre = wblk[w >> PNX_SCAN_SHIFT] + wpre[w];
    }
    s_word[t] = word;
    s_pre[t] = pre;
  }
  __syncthreads();
  const int rows = min(32, g.gy - y0);
  const int xl = t & 31;
  const int xi = x0 + xl;
  if (xi >= g.gx) return;
  const uint32_t word = s_word[xl];
  for (int yl = t >> 5; yl < rows; yl += kBlock / 32) {
    const bool occ_bit = (word >> yl) & 1u;
    const int64_t r = (int64_t)s_pre[xl] + __popc(word & ((1u << yl) - 1u));
    if (occ != nullptr) occ[((int64_t)b * g.gy + (y0 + yl)) * g.gx + xi] = (uint8_t)occ_bit;
    const bool ok = occ_bit && r < g1_rows;
    for (int c = 0; c < 64; c++) {
      const float v = ok ? g1[r * 64 + c] : 0.f;
      const int64_t o = (((int64_t)b * 64 + c) * g.gy + (y0 + yl)) * g.gx + xi;
      if (DT == PNX_F32)
        reinterpret_cast<float*>(canvas)[o] = v;
      else if (DT == PNX_BF16)
        reinterpret_cast<uint16_t*>(canvas)[o] = (uint16_t)f32_to_bf16_rne(v);
      else
        reinterpret_cast<uint16_t*>(canvas)[o] = (uint16_t)f32_to_f16_rne(v);
    }
  }
}

// Zero-fill + scatter from an explicit (P,64) list and coords (pnx_scatter_canvas).
__global__ __launch_bounds__(kBlock) void k_zero16(uint4* __restrict__ p, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (int64_t)gridDim.x * kBlock) p[i] = make_uint4(0, 0, 0, 0);
}
template <int DT>
__global__ __launch_bounds__(kBlock) void k_scatter_list(const float* __restrict__ feat, const int32_t* __restrict__ coords,
                                                         const int32_t* __restrict__ num_pillars, int64_t cap, int B, int gy, int gx,
                                                         int layout, void* __restrict__ canvas) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t r = idx >> 6;
  const int c = idx & 63;
  const int64_t P = min((int64_t)num_pillars[0], cap);
  if (r >= P) return;
  const int b = coords[r * 3], yi = coords[r * 3 + 1], xi = coords[r * 3 + 2];
  if (b < 0 || b >= B || yi < 0 || yi >= gy || xi < 0 || xi >= gx) return;
  const float v = feat[r * 64 + c];
  const int64_t o = layout == PNX_NHWC ? (((int64_t)b * gy + yi) * gx + xi) * 64 + c : (((int64_t)b * 64 + c) * gy + yi) * gx + xi;
  if (DT == PNX_F32)
    reinterpret_cast<float*>(canvas)[o] = v;
  else if (DT == PNX_BF16)
    reinterpret_cast<uint16_t*>(canvas)[o] = (uint16_t)f32_to_bf16_rne(v);
  else
    reinterpret_cast<uint16_t*>(canvas)[o] = (uint16_t)f32_to_f16_rne(v);
}

// ---- span path helpers (chunk_sort.hip + pfn_spans.hip)
// zeroes the reader's counter block and the occupancy bytes in ONE launch (two hipMemsetAsync calls = two launches + their gap)
__global__ __launch_bounds__(kBlock) void k_clear2(uint4* __restrict__ a, int64_t na16, uint8_t* __restrict__ b, int64_t nb) {
  const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x, gsz = (int64_t)gridDim.x * kBlock;
  for (int64_t i = gid; i < na16; i += gsz) a[i] = make_uint4(0, 0, 0, 0);
  if (nb <= 0) return;
  int64_t head = (int64_t)((16 - (reinterpret_cast<uintptr_t>(b) & 15)) & 15);
  head = head < nb ? head : nb;
  for (int64_t i = gid; i < head; i += gsz) b[i] = 0;
  uint4* bb = reinterpret_cast<uint4*>(b + head);
  const int64_t n16 = (nb - head) >> 4;
  for (int64_t i = gid; i < n16; i += gsz) bb[i] = make_uint4(0, 0, 0, 0);
  for (int64_t i = head + (n16 << 4) + gid; i < nb; i += gsz) b[i] = 0;
}

// the zero-fill tiles of pnx_fill.h read from occupancy bytes, as a kernel of its own (a call without points; experiments)
template <int DT>
__global__ __launch_bounds__(kBlock) void k_canvas_fill_bytes(PnxByteFillJob fj, GeomDev g) {
  __shared__ uint32_t s_row[36];
  if (fj.nt & 2) __builtin_amdgcn_s_setprio(3);  // beside the span kernel: few instructions, all of them stores -- issue them first
  pnx_fill_bytes_share<DT>(fj, g, s_row, threadIdx.x, kBlock);
}

// rank outputs of the span path: pillar rank of every point (== unq_inv of the reference for kept points, pe:110-111)
__global__ __launch_bounds__(kBlock) void k_point_rank(const int32_t* __restrict__ key, int64_t n, const uint2* __restrict__ wcomb,
                                                       const uint32_t* __restrict__ wblk, int32_t* __restrict__ rank_out,
                                                       int32_t* __restrict__ pillar_of_point) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t k = key[i];
  const int32_t r = k >= 0 ? cell_rank2(k, wcomb, wblk) : -1;
  rank_out[i] = r;
  if (pillar_of_point) pillar_of_point[i] = r;
}

// ------------------------------------------------------------------------------------------ host side
// Optional event timing (pnx_profile_begin/end).
constexpr int kEv = 8;
struct Prof {
  bool on = false;
  int cap = 0, n = 0;
  std::vector<hipEvent_t> ev;  // kEv per sample: reader start, canvas start, canvas stop, reader stop, pfn start, pfn stop, voxelize stop
} g_prof;
inline void prof_mark(int which, hipStream_t st) {
  if (g_prof.on && g_prof.n < g_prof.cap) (void)hipEventRecord(g_prof.ev[g_prof.n * kEv + which], st);
}

struct ReaderWs {
  int32_t* counters;  // [0]=P [1]=N'
  int32_t* tick;      // 16 ticket words in separate lines (pfn_v3.hip), zeroed with the counters
  uint32_t *bitmap, *wpre, *wblk;
  uint8_t* bytemap;
  int32_t* owner;
  uint32_t* rec;
  int32_t* biglist;  // pillars with more than 32 points (handled by k_pfn_big)
  int32_t* cell;     // canvas cell of every pillar
  int64_t bigcap;
  size_t zero_bytes, zero_bytes2;  // counters | tick | bytemap [| count] are contiguous: one memset per call
  int32_t *key, *rank, *slot;
  uint32_t *count, *cpre, *cblk;
  int32_t* plist;
  uint32_t *kpre, *kblk;
  float* mean;
  float* g1;
  int64_t nwords, pcap;
  int nblk_w, nblk_c, nblk_k;
  // binned path (reader_bins.h): bins of 2^sh pillars, K1 bins, points handled in `nwg` chunks of `chunk`
  int sh, K1, chunk, nwg, nblk_m;
  int gthreads;  // threads per workgroup of k_bin_count / k_bin_scatter
  int64_t matlen;
  uint32_t *histmat, *hpre, *hblk;
  uint32_t* rec64;               // pillar-sorted decorated records, 64 B per kept point
  uint32_t *pfirst, *pcnt;       // first sorted slot / number of points of every pillar
  uint2* wcomb;                  // {bitmap word, popcount prefix} pairs
  // span path (chunk_sort.hip + pfn_spans.hip, PNX_READER_IMPL=4, default)
  SpanGeom sg;
  uint4* srecs;                  // chunk-sorted 32-byte records
  uint16_t* stab;                // run table
  int32_t* srowframe;
  uint32_t* srowbase;
  int32_t *frame_lo, *frame_hi;
  uint32_t* slab_tot;
  uint2* span_desc;
  int32_t* nspan;
  int32_t* row_of;               // feat_max row per spill id
  uint8_t* cbytes;               // occupancy bytes in canvas order (when the caller passes no occupancy output)
  size_t zero_bytes_span;        // counters | tick | frame_lo | frame_hi
  size_t bytes;
};

// PNX_READER_IMPL: 4 (default) = chunk sort + span PFN (chunk_sort.hip, pfn_spans.hip); 2 = the round-2 pipeline kept as the one
// cross-check: binned grouping (reader_bins.h) + k_bin_sort + k_pfn3 (64-byte pillar-sorted records through HBM, pfn_v3.hip) -- the
// records the fused training passes (pfn_train.hip) consume, and the path of 6 point features / PNX_PFN_F16X3=0.
int reader_impl() {
  const char* e = getenv("PNX_READER_IMPL");
  return e && atoi(e) == 2 ? 2 : 4;
}

int64_t cells_padded(const pnx_geom* g, int32_t batch) {
  const int64_t gyp = (g->gy + 31) / 32 * 32;
  return (int64_t)batch * g->gx * gyp;
}

ReaderWs carve(void* ws, int64_t n, int32_t batch, const pnx_geom* g) {
  ReaderWs w;
  PnxCarver c(ws);
  const int64_t cells = cells_padded(g, batch);
  w.nwords = cells / 32;
  w.pcap = n < (int64_t)batch * g->gx * g->gy ? n : (int64_t)batch * g->gx * g->gy;
  if (w.pcap < 1) w.pcap = 1;
  w.nblk_w = (int)((w.nwords + PNX_SCAN_ITEMS - 1) / PNX_SCAN_ITEMS);
  w.nblk_c = (int)((w.pcap + PNX_SCAN_ITEMS - 1) / PNX_SCAN_ITEMS);
  w.nblk_k = (int)((n + PNX_SCAN_ITEMS - 1) / PNX_SCAN_ITEMS);
  if (w.nblk_k < 1) w.nblk_k = 1;
  w.counters = c.take<int32_t>(64);
  w.tick = c.take<int32_t>(24 * 32);  // 16 window-ticket words (word 0: span tickets of pfn_spans.hip) + 5 fill-share counters, one 128-byte line each
  w.frame_lo = c.take<int32_t>(batch);
  w.frame_hi = c.take<int32_t>(batch);
  w.zero_bytes_span = c.used();
  w.bytemap = c.take<uint8_t>(cells + 64);
  w.zero_bytes2 = c.used();            // binned path: counters | tick | bytemap
  w.count = c.take<uint32_t>(w.pcap + 8);
  w.zero_bytes = c.used();             // round-1 path: + count
  w.owner = c.take<int32_t>(cells + 8);
  w.rec = c.take<uint32_t>((n + 8) * 8);
  w.bigcap = w.pcap + 8;  // pillars of > 32 points and every pillar of a tile that leaves the fp16x3 range (pfn_v3.hip)
  w.biglist = c.take<int32_t>(2 * w.bigcap);  // [0, bigcap): > 32 points (k_bin_sort); [bigcap, 2 bigcap): fp16x3 overflow (k_pfn3)
  w.cell = c.take<int32_t>((w.pcap > 0 ? w.pcap : 1) + 8);
  w.bitmap = c.take<uint32_t>(w.nwords + 8);
  w.wpre = c.take<uint32_t>(w.nwords + 8);
  w.wblk = c.take<uint32_t>(w.nblk_w + 8);
  w.key = c.take<int32_t>(n + 8);
  w.rank = c.take<int32_t>(n + 8);
  w.slot = c.take<int32_t>(n + 8);
  w.cpre = c.take<uint32_t>(w.pcap + 8);
  w.cblk = c.take<uint32_t>(w.nblk_c + 8);
  w.plist = c.take<int32_t>(n + 8);
  w.kpre = c.take<uint32_t>(n + 8);
  w.kblk = c.take<uint32_t>(w.nblk_k + 8);
  w.mean = c.take<float>(w.pcap * 3 + 8);
  w.g1 = c.take<float>(w.pcap * 64 + 8);
  // binned path (k_bin_count / k_bin_scatter / k_bin_sort): bins small enough for k_bin_sort's LDS (<= 2048 pillars), at most ~2400 of them,
  // 512 chunks of points
  {
    w.sh = 8;
    while (w.sh < 11 && ((w.pcap + ((int64_t)1 << w.sh) - 1) >> w.sh) > 2400) w.sh++;
    w.K1 = (int)((w.pcap + ((int64_t)1 << w.sh) - 1) >> w.sh);
    int64_t chunk = (n / 512 + 255) / 256 * 256;
    if (chunk < 2048) chunk = 2048;
    w.chunk = (int)chunk;
    w.gthreads = kBlock;
  }
  w.nwg = (int)((n + w.chunk - 1) / w.chunk);
  if (w.nwg < 1) w.nwg = 1;
  w.matlen = (int64_t)w.K1 * w.nwg;
  w.nblk_m = (int)((w.matlen + PNX_SCAN_ITEMS - 1) / PNX_SCAN_ITEMS);
  w.histmat = c.take<uint32_t>(w.matlen + 8);
  w.hpre = c.take<uint32_t>(w.matlen + 8);
  w.hblk = c.take<uint32_t>(w.nblk_m + 8);
  w.rec64 = c.take<uint32_t>((n + 8) * 16);
  w.pfirst = c.take<uint32_t>(w.pcap + 8);
  w.pcnt = c.take<uint32_t>(w.pcap + 8);
  w.wcomb = c.take<uint2>(w.nwords + 8);
  // span path: chunks of kChunk points, one table row per chunk + a pool that covers the worst case (every chunk holds every frame)
  {
    const int64_t cpf = (int64_t)g->gx * g->gy;
    w.sg.cpf = (int)cpf;
    w.sg.nf = (int)((cpf + kSlabCells - 1) >> kSlabShift);
    w.sg.tabw = (w.sg.nf + 2) & ~1;
    w.sg.nchunks = (int)((n + kChunk - 1) / kChunk);
    w.sg.ovf_cap = w.sg.nchunks * (batch - 1);
    w.sg.B = batch;
    const int q_env = getenv("PNX_SPAN_QUOTA") ? atoi(getenv("PNX_SPAN_QUOTA")) : 0, s_env = getenv("PNX_SPAN_SOLO") ? atoi(getenv("PNX_SPAN_SOLO")) : -1;
    w.sg.quota = q_env >= 32 ? q_env : kSpanQuota;
    w.sg.solo = s_env >= 0 ? s_env : kSpanSolo;
    const int64_t rows = (int64_t)w.sg.nchunks + w.sg.ovf_cap;
    w.srecs = c.take<uint4>(((int64_t)w.sg.nchunks * kChunk + 8) * 2);
    w.stab = c.take<uint16_t>(rows * w.sg.tabw + 8);
    w.srowframe = c.take<int32_t>(rows + 8);
    w.srowbase = c.take<uint32_t>(rows + 8);
    w.slab_tot = c.take<uint32_t>((int64_t)batch * w.sg.nf + 8);
    w.span_desc = c.take<uint2>((int64_t)batch * (w.sg.nf + 1) + 8);
    w.nspan = c.take<int32_t>(batch + 8);
    w.row_of = c.take<int32_t>(w.pcap + 8);
    w.cbytes = c.take<uint8_t>((int64_t)batch * cpf + 64);
  }
  w.bytes = c.used();
  return w;
}

inline int64_t cells_of(const GeomDev& g) { return (int64_t)g.B * g.gx * g.gyp; }

GeomDev make_geom(const pnx_geom* g, int32_t batch) {
  GeomDev d;
  d.minx = g->pc_min[0]; d.miny = g->pc_min[1]; d.minz = g->pc_min[2];
  d.vx = g->voxel[0]; d.vy = g->voxel[1]; d.vz = g->voxel[2];
  d.gx = g->gx; d.gy = g->gy; d.gyp = (g->gy + 31) / 32 * 32;
  d.B = batch;
  return d;
}

inline int nblocks(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

int check_common(const float* points, int64_t n, int32_t stride, int32_t batch, const pnx_geom* g, void* ws, size_t ws_bytes) {
  PNX_REQUIRE(g != nullptr, PNX_ERR_INVALID, "geom is NULL");
  PNX_REQUIRE(n >= 0 && batch >= 1, PNX_ERR_INVALID, "n_points=%lld batch=%d", (long long)n, batch);
  PNX_REQUIRE(n == 0 || points != nullptr, PNX_ERR_INVALID, "points is NULL");
  PNX_REQUIRE(stride >= 4 && stride <= 7, PNX_ERR_UNSUPPORTED, "row_stride %d: kernels are built for 3..6 point features", stride);
  PNX_REQUIRE(g->gx > 0 && g->gy > 0, PNX_ERR_INVALID, "empty grid");
  PNX_REQUIRE(cells_padded(g, batch) < ((int64_t)1 << 31), PNX_ERR_UNSUPPORTED, "batch*gx*gy exceeds the int32 cell key");
  PNX_REQUIRE(n < ((int64_t)1 << 31) - 64, PNX_ERR_UNSUPPORTED, "more than 2^31 points");
  PNX_REQUIRE(ws != nullptr && ((uintptr_t)ws & 255) == 0, PNX_ERR_INVALID, "workspace must be 256-byte aligned");
  const size_t need = pnx_reader_workspace_bytes(n, batch, g);
  PNX_REQUIRE(ws_bytes >= need, PNX_ERR_WORKSPACE, "workspace %zu bytes < %zu needed", ws_bytes, need);
  return PNX_OK;
}

// Steps 1-4: keys, bitmap scan, rank/slots/coords, CSR.  Leaves counters = {P, N'}.
int run_voxelize(const float* points, int64_t n, int32_t stride, const GeomDev& gd, const ReaderWs& w, int32_t* coords,
                 int64_t pillar_capacity, int64_t* unq_inv, int32_t* pillar_of_point, bool need_kept_scan, hipStream_t st) {
  PNX_CHECK_HIP(hipMemsetAsync(w.counters, 0, w.zero_bytes, st));  // counters | count | bytemap
  if (n > 0) {
    k_keys<<<nblocks(n), kBlock, 0, st>>>(points, n, stride, gd, w.key, w.bytemap, w.owner);
    PNX_LAUNCH_CHECK();
  }
  k_pack_scan<<<w.nblk_w, kBlock, 0, st>>>(w.bytemap, w.nwords, w.bitmap, w.wpre, w.wblk, nullptr);
  k_scan_blocks<<<1, kBlock, 0, st>>>(w.wblk, w.nblk_w, w.counters + 0);
  PNX_LAUNCH_CHECK();
  if (n > 0) {
    k_rank<<<nblocks(n), kBlock, 0, st>>>(w.key, n, gd, w.bitmap, w.wpre, w.wblk, w.owner, w.rank, w.slot, w.count, coords,
                                          pillar_capacity, pillar_of_point, w.cell);
    PNX_LAUNCH_CHECK();
  }
  k_scan_local<SCAN_PLUS1><<<w.nblk_c, kBlock, 0, st>>>(w.count, w.pcap, w.cpre, w.cblk, w.counters + 0);
  k_scan_blocks<<<1, kBlock, 0, st>>>(w.cblk, w.nblk_c, w.counters + 1);
  PNX_LAUNCH_CHECK();
  if (n > 0) {
    k_fill<<<nblocks(n), kBlock, 0, st>>>(points, stride, w.rank, w.slot, n, w.count, w.cpre, w.cblk, w.plist, w.rec);
    PNX_LAUNCH_CHECK();
    if (need_kept_scan) {
      k_scan_local<SCAN_KEPT><<<w.nblk_k, kBlock, 0, st>>>(reinterpret_cast<const uint32_t*>(w.key), n, w.kpre, w.kblk);
      k_scan_blocks<<<1, kBlock, 0, st>>>(w.kblk, w.nblk_k, nullptr);
      PNX_LAUNCH_CHECK();
      if (unq_inv) {
        k_write_inv<<<nblocks(n), kBlock, 0, st>>>(w.rank, n, w.kpre, w.kblk, unq_inv);
        PNX_LAUNCH_CHECK();
      }
    }
  }
  return PNX_OK;
}

// Steps 1-4 of the binned path (reader_bins.h): keys, bitmap scan, bin histogram matrix + scan, bin scatter, in-LDS bin sort.
// Leaves counters = {P, N'}, the pillar-sorted decorated records w.rec64, w.pfirst / w.pcnt / w.cell per pillar, w.rank per point.
template <int F>
int launch_bin_sort(const GeomDev& gd, const ReaderWs& w, int32_t* coords, int64_t pillar_capacity, const PnxFillJob& fj, int fill_blocks, hipStream_t st) {
  const size_t lds = bin_sort_lds(w.sh);
  static size_t lds_set = 0;
  if (lds > lds_set) {  // more than the 64 KiB a launch gets by default
    PNX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_sort<F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set = lds;
  }
  const int bc = (int)(w.bigcap > 0x7fffffff ? 0x7fffffff : w.bigcap);
  k_bin_sort<F><<<w.K1 + (fj.quota > 0 ? fill_blocks : 0), kSortBlock, lds, st>>>(w.rec, gd, w.sh, w.nwg, w.matlen, w.hpre, w.hblk, w.counters, w.rec64,
                                                                                 w.pfirst, w.pcnt, w.cell, coords, pillar_capacity, w.biglist, bc,
                                                                                 fj);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

// fill[0..2]: shares of the canvas zero-fill carried by extra blocks of k_bin_count / k_bin_scatter / k_bin_sort (quota 0 = none)
int run_voxelize2(const float* points, int64_t n, int32_t stride, const GeomDev& gd, const ReaderWs& w, int32_t* coords,
                  int64_t pillar_capacity, int64_t* unq_inv, int32_t* pillar_of_point, const PnxFillJob* fill, int fill_blocks, hipStream_t st) {
  PNX_CHECK_HIP(hipMemsetAsync(w.counters, 0, w.zero_bytes2, st));  // counters | tick | bytemap
  if (n > 0) {
    k_keys<<<nblocks(n), kBlock, 0, st>>>(points, n, stride, gd, w.key, w.bytemap, nullptr);
    PNX_LAUNCH_CHECK();
  }
  k_pack_scan<<<w.nblk_w, kBlock, 0, st>>>(w.bytemap, w.nwords, w.bitmap, w.wpre, w.wblk, w.wcomb);
  k_scan_blocks<<<1, kBlock, 0, st>>>(w.wblk, w.nblk_w, w.counters + 0);
  PNX_LAUNCH_CHECK();
  if (n <= 0) return PNX_OK;
  const size_t hl = (size_t)(w.K1 > 64 ? w.K1 : 64) * sizeof(uint32_t);
  PnxFillJob f0 = fill[0], f1 = fill[1], f2 = fill[2];
  f0.n_main = w.nwg, f1.n_main = w.nwg, f2.n_main = w.K1;
  k_bin_count<<<w.nwg + (f0.quota > 0 ? fill_blocks : 0), w.gthreads, hl, st>>>(w.key, n, w.chunk, w.sh, w.K1, w.nwg, w.wcomb, w.wblk, w.rank, pillar_of_point,
                                                                           w.histmat, gd, f0);
  k_scan_local<SCAN_IDENT><<<w.nblk_m, kBlock, 0, st>>>(w.histmat, w.matlen, w.hpre, w.hblk);
  k_scan_blocks<<<1, kBlock, 0, st>>>(w.hblk, w.nblk_m, w.counters + 1);
  k_bin_scatter<<<w.nwg + (f1.quota > 0 ? fill_blocks : 0), w.gthreads, hl, st>>>(points, stride, w.key, w.rank, n, w.chunk, w.sh, w.K1, w.nwg, w.hpre, w.hblk,
                                                                             w.rec, gd, f1);
  PNX_LAUNCH_CHECK();
  int rc = PNX_OK;
  switch (stride - 1) {
    case 3: rc = launch_bin_sort<3>(gd, w, coords, pillar_capacity, f2, fill_blocks / 2, st); break;
    case 4: rc = launch_bin_sort<4>(gd, w, coords, pillar_capacity, f2, fill_blocks / 2, st); break;
    case 5: rc = launch_bin_sort<5>(gd, w, coords, pillar_capacity, f2, fill_blocks / 2, st); break;
    default: rc = launch_bin_sort<6>(gd, w, coords, pillar_capacity, f2, fill_blocks / 2, st); break;
  }
  if (rc != PNX_OK) return rc;
  if (unq_inv) {
    k_scan_local<SCAN_KEPT><<<w.nblk_k, kBlock, 0, st>>>(reinterpret_cast<const uint32_t*>(w.key), n, w.kpre, w.kblk);
    k_scan_blocks<<<1, kBlock, 0, st>>>(w.kblk, w.nblk_k, nullptr);
    k_write_inv<<<nblocks(n), kBlock, 0, st>>>(w.rank, n, w.kpre, w.kblk, unq_inv);
    PNX_LAUNCH_CHECK();
  }
  return PNX_OK;
}

template <int DT>
int launch_canvas(const ReaderWs& w, const float* g1, int64_t g1_rows, const GeomDev& gd, void* canvas, uint8_t* occ, int layout, hipStream_t st) {
  const int tiles = ((gd.gx + 31) / 32) * (gd.gyp / 32) * gd.B;
  if (layout == PNX_NHWC)
    k_canvas_nhwc<DT><<<tiles, kBlock, 0, st>>>(w.bitmap, w.wpre, w.wblk, g1, g1_rows, gd, canvas, occ);
  else
    k_canvas_nchw<DT><<<tiles, kBlock, 0, st>>>(w.bitmap, w.wpre, w.wblk, g1, g1_rows, gd, canvas, occ);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // namespace

// implemented in pfn_v3.hip: PFN over the pillar-sorted records of the binned path, optionally fused with the canvas zero-fill
int pnx_launch_pfn_v3(int F, const uint32_t* rec64, const uint32_t* pfirst, const uint32_t* pcnt, const int32_t* cell_of_pillar,
                      int32_t* counters, int32_t* tick, int32_t* biglist, int64_t bigcap, const float* folded, float* g1, int64_t g1_rows,
                      void* canvas, int canvas_dt, int64_t n_points, int n_fill, const PnxGeomDev& geom, const PnxFillJob& fj, hipStream_t st);

// implemented in pfn_v3.hip: the one-wave-per-pillar kernel for what the span kernel (pfn_spans.hip) spills
int pnx_launch_pfn3_tail(int F, const uint32_t* rec64, const uint32_t* pfirst, const uint32_t* pcnt, const int32_t* cell_of_pillar, int32_t* counters,
                         const int32_t* biglist, int64_t bigcap, const float* folded, float* g1, int64_t g1_rows, void* canvas, int canvas_dt, int blocks,
                         hipStream_t st, const int32_t* row_of = nullptr);

// implemented in pfn_train.hip
int pnx_launch_pfn_train(int F, int pass, const uint32_t* rec64, const uint32_t* pfirst, const uint32_t* pcnt, const int32_t* counters,
                         const float* prm, float* part, const float* G, const float* out_saved, float* out, int64_t out_rows, hipStream_t st);
int pnx_pfn_train_blocks(void);

// implemented in chunk_sort.hip / pfn_spans.hip: the one-pass grouping front end and its consumer (spans.h)
int pnx_launch_chunk_sort(const float* points, int64_t n, int stride, const PnxGeomDev& g, const SpanGeom& sg, uint4* recs, uint16_t* tab,
                          int32_t* rowframe, uint32_t* rowbase, int32_t* counters, int32_t* frame_lo, int32_t* frame_hi, uint8_t* bytemap,
                          uint32_t* slab_tot, uint2* span_desc, int32_t* nspan, hipStream_t st, hipEvent_t sorted);
int pnx_launch_span_pfn(int F, const SpanTables& T, const SpanGeom& sg, int32_t* counters, int32_t* tick, uint32_t* rec64,
                        uint32_t* pfirst, uint32_t* pcnt, int32_t* cell_of_pillar, int32_t* row_of, int32_t* biglist, int64_t bigcap, int64_t idcap,
                        const uint2* wcomb, const uint32_t* wblk, int32_t* coords, int64_t pillar_capacity, const float* folded, float* g1,
                        int64_t g1_rows, void* canvas, int canvas_dt, int canvas_nt, int64_t n_points, const PnxGeomDev& geom, hipStream_t st);

namespace {
// The zero-fill's own stream and its fork / join events: one set per host thread and device (calls of a thread are issued in order, so
// the record / wait pairs of consecutive calls cannot interleave).
struct FillSide {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
int fill_side(FillSide** out) {
  static thread_local FillSide sides[16];
  int dev = 0;
  PNX_CHECK_HIP(hipGetDevice(&dev));
  PNX_REQUIRE(dev >= 0 && dev < 16, PNX_ERR_UNSUPPORTED, "device index %d", dev);
  FillSide& f = sides[dev];
  if (f.stream == nullptr) {
    PNX_CHECK_HIP(hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking));
    PNX_CHECK_HIP(hipEventCreateWithFlags(&f.fork, hipEventDisableTiming));
    PNX_CHECK_HIP(hipEventCreateWithFlags(&f.join, hipEventDisableTiming));
  }
  *out = &f;
  return PNX_OK;
}

// PNX_READER_IMPL=4 (default): clear -> chunk sort -> [slab totals -> span carve -> span grouping + PFN -> tail] with the canvas zero-fill
// BESIDE the bracketed part: a kernel of its own on a second stream, one workgroup per CU, from the moment the occupancy bytes exist
// (the span kernel is built to leave it registers and LDS on every CU: pfn_spans.hip).  The fill has to be on the CUs BEFORE the span
// kernel's workgroups arrive: started behind the carve it is 100 us slower (profiles/r04_reader_ab.txt).  Canvas-only calls (the
// detector's path) never build the key-order bitmap; the rank outputs (feat_max / coords / unq_inv / pillar_of_point, an NCHW canvas)
// add the bitmap chain of the binned pipeline beside it.
int reader_forward_spans(const float* points, int64_t n, int32_t stride, const GeomDev& gd, const ReaderWs& w, const float* pfn_folded, void* canvas,
                         int32_t canvas_dtype, int32_t canvas_layout, uint8_t* occupancy, float* feat_max, int32_t* coords, int64_t pillar_capacity,
                         int64_t* unq_inv, int32_t* pillar_of_point, int32_t* counts, hipStream_t st) {
  const int F = stride - 1;
  const bool direct = canvas != nullptr && canvas_layout == PNX_NHWC;
  const bool ranked = feat_max != nullptr || coords != nullptr || unq_inv != nullptr || pillar_of_point != nullptr || (canvas != nullptr && !direct);
  float* g1 = nullptr;
  int64_t g1_rows = 0;
  if (feat_max != nullptr || (canvas != nullptr && !direct)) {
    g1 = (feat_max && pillar_capacity >= w.pcap) ? feat_max : w.g1;
    g1_rows = (g1 == feat_max) ? pillar_capacity : w.pcap;
  }
  const size_t canvas_bytes = (size_t)gd.B * gd.gx * gd.gy * 64 * (canvas_dtype == PNX_F32 ? 4 : 2);
  const char* nt_env = getenv("PNX_FILL_NT");
  const bool fill_nt = nt_env ? nt_env[0] == '1' : canvas_bytes >= ((size_t)3 << 29);
  uint8_t* cbytes = direct ? (occupancy != nullptr ? occupancy : w.cbytes) : nullptr;
  const int64_t ncb = cbytes != nullptr ? (int64_t)gd.B * w.sg.cpf : 0;
  int rc;
  prof_mark(0, st);
  const int all_tiles = direct ? pnx_fill_tiles_bytes(gd) : 0;
  auto launch_fill_kernel = [&](const PnxByteFillJob& job, int blocks, hipStream_t fs) -> int {
    if (canvas_dtype == PNX_F32) k_canvas_fill_bytes<PNX_F32><<<blocks, kBlock, 0, fs>>>(job, gd);
    else if (canvas_dtype == PNX_BF16) k_canvas_fill_bytes<PNX_BF16><<<blocks, kBlock, 0, fs>>>(job, gd);
    else k_canvas_fill_bytes<PNX_F16><<<blocks, kBlock, 0, fs>>>(job, gd);
    PNX_LAUNCH_CHECK();
    return PNX_OK;
  };
  {
    const int64_t work = (int64_t)(w.zero_bytes_span >> 4) + (ncb >> 4) + 64;
    int blocks = (int)((work + kBlock * 8 - 1) / (kBlock * 8));
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    k_clear2<<<blocks, kBlock, 0, st>>>(reinterpret_cast<uint4*>(w.counters), (int64_t)(w.zero_bytes_span >> 4), cbytes, ncb);
    PNX_LAUNCH_CHECK();
  }
  if (ranked) {  // the key-order bitmap and its popcount prefix: the pillar rank of a cell == torch.unique(dim=0) order (pe:110)
    PNX_CHECK_HIP(hipMemsetAsync(w.bytemap, 0, (size_t)cells_of(gd) + 64, st));
    if (n > 0) {
      k_keys<<<nblocks(n), kBlock, 0, st>>>(points, n, stride, gd, w.key, w.bytemap, nullptr);
      PNX_LAUNCH_CHECK();
    }
    k_pack_scan<<<w.nblk_w, kBlock, 0, st>>>(w.bytemap, w.nwords, w.bitmap, w.wpre, w.wblk, w.wcomb);
    k_scan_blocks<<<1, kBlock, 0, st>>>(w.wblk, w.nblk_w, w.counters + 0);
    PNX_LAUNCH_CHECK();
  }
  const char* fb_env = getenv("PNX_FILL_BLOCKS");
  const int n_fill = all_tiles > 0 ? (fb_env ? atoi(fb_env) : 256) : 0;  // one workgroup per CU (0: timing experiments, the canvas is wrong)
  const bool side = n > 0 && n_fill > 0;
  FillSide* fs = nullptr;
  if (side && (rc = fill_side(&fs)) != PNX_OK) return rc;
  rc = pnx_launch_chunk_sort(points, n, stride, gd, w.sg, w.srecs, w.stab, w.srowframe, w.srowbase, w.counters, w.frame_lo, w.frame_hi, cbytes, w.slab_tot,
                             w.span_desc, w.nspan, st, side ? fs->fork : nullptr);
  if (rc != PNX_OK) return rc;
  prof_mark(6, st);
  PnxByteFillJob fj;
  fj.bytemap = cbytes, fj.canvas = canvas, fj.counter = w.tick + 16 * 32, fj.tiles = all_tiles, fj.nt = fill_nt ? 1 : 0, fj.base = 0;
  SpanTables T;
  T.recs = w.srecs, T.tab = w.stab, T.rowframe = w.srowframe, T.rowbase = w.srowbase, T.frame_lo = w.frame_lo, T.frame_hi = w.frame_hi;
  T.span_desc = w.span_desc, T.nspan = w.nspan;
  prof_mark(4, st);
  prof_mark(1, st);
  if (side) {  // few instructions, all of them stores: the fill's waves issue ahead of the span kernel's (s_setprio in the kernel)
    PnxByteFillJob sj = fj;
    sj.nt |= 2;
    PNX_CHECK_HIP(hipStreamWaitEvent(fs->stream, fs->fork, 0));
    rc = launch_fill_kernel(sj, n_fill, fs->stream);
    if (rc != PNX_OK) return rc;
    PNX_CHECK_HIP(hipEventRecord(fs->join, fs->stream));
  } else if (n_fill > 0) {  // nothing to group: the fill alone
    rc = launch_fill_kernel(fj, n_fill, st);
    if (rc != PNX_OK) return rc;
  }
  // From here on the fill may be writing the caller's canvas on the side stream: EVERY return path joins it back into `st` first, so that whatever
  // the caller does with the canvas next (reuse, free after an error) is ordered behind the fill.
  rc = pnx_launch_span_pfn(F, T, w.sg, w.counters, w.tick, w.rec64, w.pfirst, w.pcnt, w.cell, w.row_of, w.biglist, w.bigcap, w.pcap,
                           ranked ? w.wcomb : nullptr, w.wblk, coords, pillar_capacity, pfn_folded, g1, g1_rows, direct ? canvas : nullptr, canvas_dtype,
                           fill_nt ? 1 : 0, n, gd, st);
  if (rc == PNX_OK && n > 0) {
    const int tb = 128;
    rc = pnx_launch_pfn3_tail(F, w.rec64, w.pfirst, w.pcnt, w.cell, w.counters, w.biglist, w.bigcap, pfn_folded, g1, g1_rows, direct ? canvas : nullptr,
                              canvas_dtype, tb, st, ranked ? w.row_of : nullptr);
  }
  if (side) {
    const hipError_t je = hipStreamWaitEvent(st, fs->join, 0);
    if (rc == PNX_OK && je != hipSuccess) {
      pnx_set_error("hipStreamWaitEvent(join) failed: %s", hipGetErrorString(je));
      return PNX_ERR_HIP;
    }
  }
  if (rc != PNX_OK) return rc;
  prof_mark(5, st);
  prof_mark(2, st);
  if (feat_max && g1 != feat_max) {  // caller's buffer is smaller than the worst case: copy what fits (P is unknown on the host)
    PNX_CHECK_HIP(hipMemcpyAsync(feat_max, g1, (size_t)(pillar_capacity < w.pcap ? pillar_capacity : w.pcap) * 64 * sizeof(float),
                                 hipMemcpyDeviceToDevice, st));
  }
  if (canvas != nullptr && !direct) {
    if (canvas_dtype == PNX_F32) rc = launch_canvas<PNX_F32>(w, g1, g1_rows, gd, canvas, occupancy, canvas_layout, st);
    else if (canvas_dtype == PNX_BF16) rc = launch_canvas<PNX_BF16>(w, g1, g1_rows, gd, canvas, occupancy, canvas_layout, st);
    else rc = launch_canvas<PNX_F16>(w, g1, g1_rows, gd, canvas, occupancy, canvas_layout, st);
    if (rc != PNX_OK) return rc;
  }
  if (n > 0 && (unq_inv != nullptr || pillar_of_point != nullptr)) {
    k_point_rank<<<nblocks(n), kBlock, 0, st>>>(w.key, n, w.wcomb, w.wblk, w.rank, pillar_of_point);
    if (unq_inv) {
      k_scan_local<SCAN_KEPT><<<w.nblk_k, kBlock, 0, st>>>(reinterpret_cast<const uint32_t*>(w.key), n, w.kpre, w.kblk);
      k_scan_blocks<<<1, kBlock, 0, st>>>(w.kblk, w.nblk_k, nullptr);
      k_write_inv<<<nblocks(n), kBlock, 0, st>>>(w.rank, n, w.kpre, w.kblk, unq_inv);
    }
    PNX_LAUNCH_CHECK();
  }
  if (counts) PNX_CHECK_HIP(hipMemcpyAsync(counts, w.counters, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  prof_mark(3, st);
  if (g_prof.on && g_prof.n < g_prof.cap) g_prof.n++;
  return PNX_OK;
}
}  // namespace


extern "C" {

size_t pnx_reader_workspace_bytes(int64_t n_points, int32_t batch, const pnx_geom* g) {
  if (!g || n_points < 0 || batch < 1) return 0;
  return carve(nullptr, n_points, batch, g).bytes;
}

int pnx_pfn_fold_bn(int32_t F, const float* w0, const float* gamma0, const float* beta0, const float* mean0, const float* var0,
                    const float* w1, const float* gamma1, const float* beta1, const float* mean1, const float* var1, float eps,
                    float* folded_out, pnx_stream_t stream) {
  PNX_REQUIRE(F >= 3 && F <= 6, PNX_ERR_UNSUPPORTED, "num_point_features %d not in 3..6", F);
  PNX_REQUIRE(w0 && gamma0 && beta0 && mean0 && var0 && w1 && gamma1 && beta1 && mean1 && var1 && folded_out, PNX_ERR_INVALID,
              "null parameter pointer");
  k_fold_bn<<<32, 256, 0, (hipStream_t)stream>>>(F + 5, w0, gamma0, beta0, mean0, var0, w1, gamma1, beta1, mean1, var1, eps, folded_out);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

int pnx_reader_forward(const float* points, int64_t n, int32_t stride, int32_t batch, const pnx_geom* g, const float* pfn_folded,
                       void* canvas, int32_t canvas_dtype, int32_t canvas_layout, uint8_t* occupancy, float* feat_max, int32_t* coords,
                       int64_t pillar_capacity, int64_t* unq_inv, int32_t* pillar_of_point, int32_t* counts, void* workspace,
                       size_t workspace_bytes, pnx_stream_t stream) {
  int rc = check_common(points, n, stride, batch, g, workspace, workspace_bytes);
  if (rc != PNX_OK) return rc;
  PNX_REQUIRE(pfn_folded != nullptr, PNX_ERR_INVALID, "pfn_folded is NULL");
  PNX_REQUIRE(canvas_dtype >= PNX_F32 && canvas_dtype <= PNX_F16, PNX_ERR_INVALID, "bad canvas_dtype %d", canvas_dtype);
  PNX_REQUIRE(canvas_layout == PNX_NHWC || canvas_layout == PNX_NCHW, PNX_ERR_INVALID, "bad canvas_layout %d", canvas_layout);
  PNX_REQUIRE(canvas == nullptr || ((uintptr_t)canvas & 15) == 0, PNX_ERR_INVALID, "canvas must be 16-byte aligned");
  PNX_REQUIRE(occupancy == nullptr || canvas != nullptr, PNX_ERR_INVALID, "occupancy is produced by the canvas kernel: pass a canvas too");
  PNX_REQUIRE(feat_max == nullptr || ((uintptr_t)feat_max & 15) == 0, PNX_ERR_INVALID, "feat_max must be 16-byte aligned");
  PNX_REQUIRE((feat_max == nullptr && coords == nullptr) || pillar_capacity > 0, PNX_ERR_INVALID, "pillar_capacity must be > 0");
  hipStream_t st = (hipStream_t)stream;
  const ReaderWs w = carve(workspace, n, batch, g);
  const GeomDev gd = make_geom(g, batch);

  const int F = stride - 1;
  {
    const char* h16_env = getenv("PNX_PFN_F16X3");
    const int64_t rows = (int64_t)w.sg.nchunks * batch;
    if (reader_impl() == 4 && F <= 5 && !(h16_env && h16_env[0] == '0') && w.sg.nf <= 32768 && batch <= 1024 && rows < ((int64_t)1 << 30))
      return reader_forward_spans(points, n, stride, gd, w, pfn_folded, canvas, canvas_dtype, canvas_layout, occupancy, feat_max, coords, pillar_capacity,
                                  unq_inv, pillar_of_point, counts, st);
  }
  // ---- the binned pipeline (PNX_READER_IMPL=2; 6 point features; PNX_PFN_F16X3=0): reader_bins.h + pfn_v3.hip
  PNX_REQUIRE(w.K1 <= 16384, PNX_ERR_UNSUPPORTED, "too many points for the binned grouping");
  // Direct mode (NHWC canvas): the PFN kernel stores each pillar straight into its canvas cell and fill blocks write the zeros of every
  // other cell -- no (P,64) fp32 intermediate, every canvas byte written exactly once.
  const bool direct = canvas != nullptr && canvas_layout == PNX_NHWC;
  float* g1 = (feat_max && pillar_capacity >= w.pcap) ? feat_max : w.g1;  // feat_max doubles as the PFN output buffer when it can hold every possible pillar
  if (direct && feat_max == nullptr) g1 = nullptr;
  const int64_t g1_rows = (g1 == feat_max) ? pillar_capacity : w.pcap;
  const size_t canvas_bytes = (size_t)gd.B * gd.gx * gd.gy * 64 * (canvas_dtype == PNX_F32 ? 4 : 2);
  const char* nt_env = getenv("PNX_FILL_NT");
  const bool fill_nt = nt_env ? nt_env[0] == '1' : canvas_bytes >= ((size_t)3 << 29);  // >= 1.5 GiB: far beyond what the Infinity Cache absorbs
  // The zero-fill's 32x32-cell tiles are dealt to the launches that leave HBM idle (pnx_fill.h): extra blocks of k_bin_count /
  // k_bin_scatter / k_bin_sort take `split` percent each (PNX_FILL_SPLIT="a,b,c", default 0,0,24), the PFN launch the rest.
  PnxFillJob fjob[4];
  {
    int split[3];
    pnx_reader_fill_split(split);
    const int tiles = pnx_fill_tiles(gd);
    int base = 0;
    for (int k = 0; k < 4; k++) {
      int q = k < 3 ? (int)((int64_t)tiles * split[k] / 100) : tiles - base;
      if (!direct || n <= 0) q = k < 3 ? 0 : (direct ? tiles : 0);
      if (q > tiles - base) q = tiles - base;
      fjob[k].bitmap = w.bitmap, fjob[k].canvas = canvas, fjob[k].occ = occupancy, fjob[k].counter = w.tick + (16 + k) * 32;
      fjob[k].base = base, fjob[k].quota = q, fjob[k].dt = canvas_dtype, fjob[k].nt = fill_nt ? 1 : 0, fjob[k].n_main = 0;
      base += q;
    }
  }
  const char* fb_env = getenv("PNX_FILL_BLOCKS");
  const int fill_blocks = fb_env ? atoi(fb_env) : 256;
  prof_mark(0, st);
  rc = run_voxelize2(points, n, stride, gd, w, coords, pillar_capacity, unq_inv, pillar_of_point, fjob, fill_blocks, st);
  if (rc != PNX_OK) return rc;
  prof_mark(6, st);
  prof_mark(4, st);
  prof_mark(1, st);
  rc = pnx_launch_pfn_v3(F, w.rec64, w.pfirst, w.pcnt, w.cell, w.counters, w.tick, w.biglist, w.bigcap, pfn_folded, g1, g1_rows,
                         direct ? canvas : nullptr, canvas_dtype, n, direct ? fill_blocks : 0, gd, fjob[3], st);
  if (rc != PNX_OK) return rc;
  prof_mark(5, st);
  prof_mark(2, st);
  if (feat_max && g1 != feat_max) {
    // caller's buffer is smaller than the worst case: copy what fits (P is unknown on the host)
    PNX_CHECK_HIP(hipMemcpyAsync(feat_max, g1, (size_t)(pillar_capacity < w.pcap ? pillar_capacity : w.pcap) * 64 * sizeof(float),
                                 hipMemcpyDeviceToDevice, st));
  }
  if (canvas != nullptr && !direct) {
    if (canvas_dtype == PNX_F32) rc = launch_canvas<PNX_F32>(w, g1, g1_rows, gd, canvas, occupancy, canvas_layout, st);
    else if (canvas_dtype == PNX_BF16) rc = launch_canvas<PNX_BF16>(w, g1, g1_rows, gd, canvas, occupancy, canvas_layout, st);
    else rc = launch_canvas<PNX_F16>(w, g1, g1_rows, gd, canvas, occupancy, canvas_layout, st);
    if (rc != PNX_OK) return rc;
  }
  if (counts) PNX_CHECK_HIP(hipMemcpyAsync(counts, w.counters, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  prof_mark(3, st);
  if (g_prof.on && g_prof.n < g_prof.cap) g_prof.n++;
  return PNX_OK;
}

// ---- training-mode PFN (pfn_train.hip).  Parameter block: W1 (4096) | mu1 is1 gamma1 beta1 m1 m2 (6 x 64) | mu0 is0 gamma0 beta0
// (4 x 32) | W0 (32 x (F+5)).
size_t pnx_pfn_train_param_floats(int32_t F) { return 4096 + 6 * 64 + 4 * 32 + 32 * (size_t)(F + 5); }
size_t pnx_pfn_train_partial_floats(int32_t F, int32_t which) {
  const size_t C0 = (size_t)F + 5, nb = (size_t)pnx_pfn_train_blocks();
  switch (which) {
    case 0: return nb * (C0 + C0 * C0);         // forward pass 0: per block  [F1 | F2]
    case 1: return nb * 4 * 64 * 65;            // forward pass 1: per wave   [64][U2 row | U1]
    case 3: return nb * 4 * 64 * 66;            // backward pass 0: per wave  [64][A row | D1 | D2]
    case 4: return nb * 4 * 32 * (C0 + 2);      // backward pass 1: per wave  [32][B0 row | E1 | E2]
  }
  return 0;
}

int pnx_pfn_forward_train(int32_t pass, const float* points, int64_t n, int32_t stride, int32_t batch, const pnx_geom* g, const float* params,
                          float* partials, float* feat_max, int64_t pillar_capacity, int32_t* coords, int32_t* pillar_of_point, int32_t* counts,
                          void* workspace, size_t workspace_bytes, pnx_stream_t stream) {
  int rc = check_common(points, n, stride, batch, g, workspace, workspace_bytes);
  if (rc != PNX_OK) return rc;
  PNX_REQUIRE(pass >= 0 && pass <= 2, PNX_ERR_INVALID, "pass %d not in 0..2", pass);
  PNX_REQUIRE(pass == 2 || partials != nullptr, PNX_ERR_INVALID, "partials is NULL");
  PNX_REQUIRE(pass == 0 || params != nullptr, PNX_ERR_INVALID, "params is NULL");
  PNX_REQUIRE(pass != 2 || (feat_max != nullptr && pillar_capacity > 0), PNX_ERR_INVALID, "pass 2 needs feat_max");
  hipStream_t st = (hipStream_t)stream;
  const ReaderWs w = carve(workspace, n, batch, g);
  PNX_REQUIRE(w.K1 <= 16384, PNX_ERR_UNSUPPORTED, "too many points for the binned grouping");
  const GeomDev gd = make_geom(g, batch);
  if (pass == 0) {
    PnxFillJob nofill[3] = {};
    rc = run_voxelize2(points, n, stride, gd, w, coords, pillar_capacity, nullptr, pillar_of_point, nofill, 0, st);
    if (rc != PNX_OK) return rc;
    if (counts) PNX_CHECK_HIP(hipMemcpyAsync(counts, w.counters, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  }
  return pnx_launch_pfn_train(stride - 1, pass, w.rec64, w.pfirst, w.pcnt, w.counters, params, partials, nullptr, nullptr, feat_max, pillar_capacity, st);
}

int pnx_pfn_backward(int32_t pass, int64_t n, int32_t stride, int32_t batch, const pnx_geom* g, const float* params, const float* grad_feat_max,
                     const float* feat_max, float* partials, void* workspace, size_t workspace_bytes, pnx_stream_t stream) {
  PNX_REQUIRE(g != nullptr && workspace != nullptr && params && grad_feat_max && feat_max && partials, PNX_ERR_INVALID, "null pointer");
  PNX_REQUIRE(pass == 0 || pass == 1, PNX_ERR_INVALID, "pass %d not in 0..1", pass);
  PNX_REQUIRE(stride >= 4 && stride <= 7, PNX_ERR_UNSUPPORTED, "row_stride %d", stride);
  PNX_REQUIRE(workspace_bytes >= pnx_reader_workspace_bytes(n, batch, g), PNX_ERR_WORKSPACE, "workspace too small");
  const ReaderWs w = carve(workspace, n, batch, g);  // the records written by pnx_pfn_forward_train(pass 0) on the same workspace
  return pnx_launch_pfn_train(stride - 1, pass == 0 ? 3 : 4, w.rec64, w.pfirst, w.pcnt, w.counters, params, partials, grad_feat_max, feat_max, nullptr, 0,
                              (hipStream_t)stream);
}

void pnx_reader_fill_split(int32_t* percent3) {
  percent3[0] = 0, percent3[1] = 0, percent3[2] = reader_impl() == 4 ? 0 : 24;  // round-2 pipeline, measured on C2 / 8 frames (tools/reader_ab.py): only k_bin_sort's share pays
  const char* sp_env = getenv("PNX_FILL_SPLIT");
  if (sp_env) sscanf(sp_env, "%d,%d,%d", &percent3[0], &percent3[1], &percent3[2]);
  for (int k = 0; k < 3; k++) percent3[k] = percent3[k] < 0 ? 0 : (percent3[k] > 100 ? 100 : percent3[k]);
}

int pnx_profile_begin(int32_t max_samples) {
  PNX_REQUIRE(max_samples > 0 && max_samples <= 65536, PNX_ERR_INVALID, "max_samples out of range");
  while ((int)g_prof.ev.size() < max_samples * kEv) {
    hipEvent_t e;
    PNX_CHECK_HIP(hipEventCreate(&e));
    g_prof.ev.push_back(e);
  }
  g_prof.cap = max_samples;
  g_prof.n = 0;
  g_prof.on = true;
  return PNX_OK;
}

static float g_last_pfn_us = 0.f, g_last_vox_us = 0.f;
float pnx_profile_last_pfn_us(void) { return g_last_pfn_us; }
float pnx_profile_last_voxelize_us(void) { return g_last_vox_us; }

int pnx_profile_end(float* reader_us, float* canvas_us, int32_t* samples) {
  g_prof.on = false;
  double r = 0, c = 0, f = 0, v = 0;
  for (int i = 0; i < g_prof.n; i++) {
    float ms = 0;
    PNX_CHECK_HIP(hipEventSynchronize(g_prof.ev[i * kEv + 3]));
    PNX_CHECK_HIP(hipEventElapsedTime(&ms, g_prof.ev[i * kEv + 0], g_prof.ev[i * kEv + 3]));
    r += ms;
    PNX_CHECK_HIP(hipEventElapsedTime(&ms, g_prof.ev[i * kEv + 1], g_prof.ev[i * kEv + 2]));
    c += ms;
    if (hipEventElapsedTime(&ms, g_prof.ev[i * kEv + 4], g_prof.ev[i * kEv + 5]) == hipSuccess) f += ms;
    if (hipEventElapsedTime(&ms, g_prof.ev[i * kEv + 0], g_prof.ev[i * kEv + 6]) == hipSuccess) v += ms;
  }
  if (getenv("PNX_DEBUG") && g_prof.n) fprintf(stderr, "[pnx] profile: pfn kernel %.2f us avg over %d calls\n", f * 1e3 / g_prof.n, g_prof.n);
  const int n = g_prof.n;
  g_last_pfn_us = n ? (float)(f * 1e3 / n) : 0.f;
  g_last_vox_us = n ? (float)(v * 1e3 / n) : 0.f;
  if (reader_us) *reader_us = n ? (float)(r * 1e3 / n) : 0.f;
  if (canvas_us) *canvas_us = n ? (float)(c * 1e3 / n) : 0.f;
  if (samples) *samples = n;
  g_prof.n = 0;
  return PNX_OK;
}

int pnx_voxelize(const float* points, int64_t n, int32_t stride, int32_t batch, const pnx_geom* g, float* features, int32_t* coords,
                 int64_t pillar_capacity, int64_t* unq_inv, int32_t* pillar_of_point, int32_t* counts, void* workspace,
                 size_t workspace_bytes, pnx_stream_t stream) {
  int rc = check_common(points, n, stride, batch, g, workspace, workspace_bytes);
  if (rc != PNX_OK) return rc;
  PNX_REQUIRE(coords == nullptr || pillar_capacity > 0, PNX_ERR_INVALID, "pillar_capacity must be > 0");
  hipStream_t st = (hipStream_t)stream;
  const ReaderWs w = carve(workspace, n, batch, g);
  const GeomDev gd = make_geom(g, batch);
  rc = run_voxelize(points, n, stride, gd, w, coords, pillar_capacity, unq_inv, pillar_of_point, unq_inv != nullptr || features != nullptr, st);
  if (rc != PNX_OK) return rc;
  if (features && n > 0) {
    k_pillar_mean<<<nblocks(w.pcap), kBlock, 0, st>>>(points, stride, w.plist, w.count, w.cpre, w.cblk, w.counters, w.mean);
    switch (stride - 1) {
      case 3: k_decorate<3><<<nblocks(n), kBlock, 0, st>>>(points, n, gd, w.rank, w.kpre, w.kblk, w.mean, features); break;
      case 4: k_decorate<4><<<nblocks(n), kBlock, 0, st>>>(points, n, gd, w.rank, w.kpre, w.kblk, w.mean, features); break;
      case 5: k_decorate<5><<<nblocks(n), kBlock, 0, st>>>(points, n, gd, w.rank, w.kpre, w.kblk, w.mean, features); break;
      default: k_decorate<6><<<nblocks(n), kBlock, 0, st>>>(points, n, gd, w.rank, w.kpre, w.kblk, w.mean, features); break;
    }
    PNX_LAUNCH_CHECK();
  }
  if (counts) PNX_CHECK_HIP(hipMemcpyAsync(counts, w.counters, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return PNX_OK;
}

int pnx_scatter_canvas(const float* feat_max, const int32_t* coords, const int32_t* num_pillars_dev, int64_t pillar_capacity,
                       int32_t batch, int32_t gy, int32_t gx, void* canvas, int32_t canvas_dtype, int32_t canvas_layout,
                       pnx_stream_t stream) {
  PNX_REQUIRE(feat_max && coords && num_pillars_dev && canvas, PNX_ERR_INVALID, "null pointer");
  PNX_REQUIRE(batch >= 1 && gy > 0 && gx > 0 && pillar_capacity >= 0, PNX_ERR_INVALID, "bad sizes");
  PNX_REQUIRE(canvas_dtype >= PNX_F32 && canvas_dtype <= PNX_F16, PNX_ERR_INVALID, "bad canvas_dtype %d", canvas_dtype);
  PNX_REQUIRE(((uintptr_t)canvas & 15) == 0, PNX_ERR_INVALID, "canvas must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t bytes = (int64_t)batch * 64 * gy * gx * (canvas_dtype == PNX_F32 ? 4 : 2);
  k_zero16<<<2048, kBlock, 0, st>>>(reinterpret_cast<uint4*>(canvas), bytes / 16);
  if (pillar_capacity > 0) {
    const int nb = nblocks(pillar_capacity * 64);
    if (canvas_dtype == PNX_F32) k_scatter_list<PNX_F32><<<nb, kBlock, 0, st>>>(feat_max, coords, num_pillars_dev, pillar_capacity, batch, gy, gx, canvas_layout, canvas);
    else if (canvas_dtype == PNX_BF16) k_scatter_list<PNX_BF16><<<nb, kBlock, 0, st>>>(feat_max, coords, num_pillars_dev, pillar_capacity, batch, gy, gx, canvas_layout, canvas);
    else k_scatter_list<PNX_F16><<<nb, kBlock, 0, st>>>(feat_max, coords, num_pillars_dev, pillar_capacity, batch, gy, gx, canvas_layout, canvas);
  }
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // extern "C"
