// conv_wgrad.hip -- weight gradient of the masked stride-1 3x3 convolution of the sparse backbone in training (gfx950).
//
// Reference: det3d/models/utils/sparse_conv.py:16-63 under autograd -- spconv's SubMConv2d / SparseConv2d accumulate the weight gradient
// over the ACTIVE output sites only.  On the masked-dense stand-in (bf16 NHWC maps that are zero at inactive sites):
//     dW[co][ci][ky][kx] = sum over active output sites p of  dY[p][co] * X[p + (ky-1, kx-1)][ci]
// MIOpen's dense wrw kernels run this at ~190 TFLOP/s over every cell of the 1440^2 maps (3.2 ms per 64 -> 64 layer at 4 frames) although
// 70-83 % of the 16-pixel row pieces hold no active output.  Here it is an implicit GEMM with K = pixels:
//   M = 64 output channels, N = 64 input channels (one 64 x 64 block pair of the layer per workgroup row), K = 16 consecutive pixels of a row
//   A = dY^T and B = X are both "pixel-major" operands of v_mfma_f32_32x32x16_bf16 while the maps are channel-major (NHWC): the fragments
//   come from LDS through ds_read_b64_tr_b16, gfx950's transposing read (a 16-lane group hands in a [4 pixels][16 channels] block, 8 bytes
//   per lane, and every lane gets the 4 pixels of ONE channel back) -- two reads per operand and K step, straight from the staged NHWC tile
//   a wave = one 32 x 32 channel block x all 9 taps (9 accumulators, 144 registers); per 16-pixel step 2 reads of dY serve 9 MFMAs,
//   every tap reads its own shifted window of the X halo tile (immediate offsets from one base address)
//   stride 2 (the entry convolutions of stages 1-3): the same kernel on 2 x 32 output tiles, the X window of a tap strided by two pixels
//   (a transposing read takes a free address per lane)
//   K steps without an active output are skipped (row masks built while staging); tiles are dealt to the workgroups STATICALLY and every workgroup
//   writes one fp32 partial, which a second kernel adds up in a fixed order: the result does not depend on timing (MIOpen's wrw adds
//   with global atomics)
// LDS: pixels 144 bytes apart (64 bf16 + 16 bytes): the four pixel rows of a transposing read then fall into disjoint banks.
#include "pnx_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int WG_PS = 144;                                 // bytes per staged pixel
constexpr int WG_LIST_MAX = 2040;                          // non-empty tiles a workgroup lists (more tiles: PNX_ERR_UNSUPPORTED at launch)
constexpr int WG_GROUPS = 512;                             // workgroups per launch (all block pairs together)

// Tile geometry by stride S (1: the submanifold / stride-1 layers; 2: the entry convolutions of stages 1-3, iy = 2 oy + ky - 1):
//   output tile TH rows x 32 pixels, input halo tile XH x XW pixels, 16-byte chunks per thread of the two operands
template <int S>
struct WgGeo {
  static constexpr int TH = S == 1 ? 4 : 2;
  static constexpr int XH = S * (TH - 1) + 3, XW = S * 31 + 3;  // 6 x 34 | 5 x 65
  static constexpr int XBYTES = XH * XW * WG_PS, YBYTES = TH * 32 * WG_PS;
  static constexpr int NY = TH, NX = (XH * XW * 8 + 255) / 256;  // 4 + 7 | 2 + 11
  static constexpr int LDS = XBYTES + YBYTES + (WG_LIST_MAX + 16) * 4;  // 56 000 | 64 272: two workgroups per CU beside 9 x 16 accumulator registers
};

// 8 pixels (k) of one channel for this lane: two transposing reads, 4 pixels each (STEP pixels of the LDS image apart: 4 outputs = 4 S inputs)
template <int OFF, int STEP>
__device__ __forceinline__ bf16x8 tr_frag(const uint8_t* base) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + OFF));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + OFF + STEP * WG_PS));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// One tile's operands on their way from HBM to LDS: NY + NX chunks of 16 bytes per thread, held in registers while the previous tile's K steps run.
template <int S>
struct TileRegs {
  uint4 y[WgGeo<S>::NY], x[WgGeo<S>::NX];
  uint32_t on;  // bit j: the thread's pixel of row j is an active output
};

// H, W: the INPUT map; Ho, Wo: the output map (= H, W at stride 1)
template <int S>
__device__ __forceinline__ void tile_load(TileRegs<S>& R, const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, const uint8_t* __restrict__ mask, int tile,
                                          int tiles_x, int tiles_y, int H, int W, int Ho, int Wo, int cin, int cout, int cb, int ib, int t, bool ld_x = true,
                                          bool ld_y = true) {
  using G = WgGeo<S>;
  const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
  const int x0 = tx << 5, y0 = ty * G::TH;
  R.on = 0u;
#pragma unroll
  for (int j = 0; j < G::NY && ld_y; j++) {  // dY: chunk c = t + 256 j -> row j, pixel (t >> 3), chunk t & 7; zero at inactive outputs
    const int q = t & 7, px = t >> 3, oy = y0 + j, ox = x0 + px;
    R.y[j] = make_uint4(0, 0, 0, 0);
    if (oy < Ho && ox < Wo) {
      const int64_t site = ((int64_t)b * Ho + oy) * Wo + ox;
      if (mask[site] != 0) {
        R.y[j] = *reinterpret_cast<const uint4*>(dy + site * cout + 64 * cb + 8 * q);
        R.on |= 1u << j;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < G::NX && ld_x; j++) {  // X halo: chunk c = t + 256 j of XH x XW x 8; zero outside the image
    const int c = t + 256 * j, q = c & 7, p = c >> 3, r = p / G::XW, px = p - r * G::XW;
    const int iy = S * y0 - 1 + r, ix = S * x0 - 1 + px;
    R.x[j] = make_uint4(0, 0, 0, 0);
    if (c < G::XH * G::XW * 8 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
      R.x[j] = *reinterpret_cast<const uint4*>(x + (((int64_t)b * H + iy) * W + ix) * cin + 64 * ib + 8 * q);
  }
}

template <int S>
__device__ __forceinline__ void tile_store(const TileRegs<S>& R, uint8_t* sx, uint8_t* sy, uint32_t* rowmask, int t, bool st_x = true, bool st_y = true) {
  using G = WgGeo<S>;
#pragma unroll
  for (int j = 0; j < G::NY && st_y; j++) {
    *reinterpret_cast<uint4*>(sy + (j * 32 + (t >> 3)) * WG_PS + 16 * (t & 7)) = R.y[j];
    if ((t & 7) == 0 && ((R.on >> j) & 1u)) atomicOr(&rowmask[j], 1u << (t >> 3));  // row masks of the tile: which 16-pixel pieces hold an active output
  }
#pragma unroll
  for (int j = 0; j < G::NX && st_x; j++) {
    const int c = t + 256 * j;
    if (c < G::XH * G::XW * 8) *reinterpret_cast<uint4*>(sx + (c >> 3) * WG_PS + 16 * (c & 7)) = R.x[j];
  }
}

// X3 (pnx_conv3x3_wgrad_x3: the fp32 weight gradient from the operands' bf16 halves): every listed tile is visited three times, the accumulators running
// through -- X = x_lo, Y = dy_hi;  then X = x_hi (Y stays in LDS);  then Y = dy_lo (X stays) -- four operand tiles from HBM for the three products.
template <int S, bool X3 = false>
__global__ __launch_bounds__(256, 2) void k_wgrad64(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                    float* __restrict__ part, int B, int H, int W, int Ho, int Wo, int cin, int cout, int G,
                                                    const uint16_t* __restrict__ x_lo = nullptr, const uint16_t* __restrict__ dy_lo = nullptr) {
  using Geo = WgGeo<S>;
  constexpr int TH = Geo::TH, XW = Geo::XW;
  extern __shared__ __align__(16) uint8_t s_tile[];  // X halo tile, the dY tile, the workgroup's list of non-empty tiles (no static LDS in front: the base stays 16-byte aligned)
  uint8_t* sx = s_tile;
  uint8_t* sy = s_tile + Geo::XBYTES;
  int32_t* s_list = reinterpret_cast<int32_t*>(s_tile + Geo::XBYTES + Geo::YBYTES);   // [0] count, [1..] tiles in ascending order
  uint32_t* s_rm = reinterpret_cast<uint32_t*>(s_list + WG_LIST_MAX + 8);              // two sets of 4 row masks (tile k uses set k & 1)
  const int t = threadIdx.x, l = t & 63, wv = t >> 6;
  const int nib = cin >> 6;
  const int pair = blockIdx.y, cb = pair / nib, ib = pair - cb * nib;  // 64-channel block of the outputs / of the inputs
  const int mb = wv & 1, nb = wv >> 1;                                 // this wave's 32-channel halves
  const int grp = l >> 4, i16 = l & 15;
  // the lane's corner of the [4 pixels][16 channels] block its group hands to a transposing read: output pixel 8 (grp >> 1) + i16 / 4 of the
  // 16-pixel piece (input pixel S times that), channels 4 (i16 % 4)..
  const int lane_px = 8 * (grp >> 1) + (i16 >> 2), lane_ch = 16 * (grp & 1) + 4 * (i16 & 3);
  const uint8_t* ay = sy + lane_px * WG_PS + (32 * mb + lane_ch) * 2;
  const uint8_t* bx = sx + S * lane_px * WG_PS + (32 * nb + lane_ch) * 2;
  v16f acc[9];
#pragma unroll
  for (int k = 0; k < 9; k++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[k][i] = 0.f;
  const int tiles_x = (Wo + 31) >> 5, tiles_y = (Ho + TH - 1) / TH;
  const int n_tiles = B * tiles_y * tiles_x;
  // ---- this workgroup's tiles (blockIdx.x, + G, ...) that hold an active output, in ascending order: one tile per thread and pass,
  // compacted with ballots (the order, and with it the fp32 sum, is fixed)
  if (t == 0) s_list[0] = 0;
  if (t < 8) s_rm[t] = 0u;
  __syncthreads();
  for (int base = blockIdx.x; base < n_tiles; base += 256 * G) {
    const int tile = base + t * G;
    bool any = false;
    if (tile < n_tiles) {
      const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
      const int x0 = tx << 5, y0 = ty * TH;
      const bool wide = (Wo & 15) == 0 && x0 + 32 <= Wo;
      uint32_t o = 0;
      for (int r = 0; r < TH && y0 + r < Ho; r++) {
        const uint8_t* row = mask + ((int64_t)b * Ho + (y0 + r)) * Wo + x0;
        if (wide) {
          const uint4 a = reinterpret_cast<const uint4*>(row)[0], c = reinterpret_cast<const uint4*>(row)[1];
          o |= a.x | a.y | a.z | a.w | c.x | c.y | c.z | c.w;
        } else {
          for (int k = 0; k < 32 && x0 + k < Wo; k++) o |= row[k];
        }
      }
      any = o != 0;
    }
    const uint64_t bal = __ballot(any);
    int* s_wcnt = s_list + WG_LIST_MAX + 1;
    if (l == 0) s_wcnt[wv] = __popcll(bal);
    __syncthreads();
    int off = s_list[0];
    for (int k = 0; k < wv; k++) off += s_wcnt[k];
    if (any) {
      const int at = off + __popcll(bal & ((1ull << l) - 1ull));
      if (at < WG_LIST_MAX) s_list[1 + at] = tile;
    }
    __syncthreads();
    if (t == 0) s_list[0] += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
  }
  const int n_mine = min(s_list[0], WG_LIST_MAX);
  // ---- software pipeline over the non-empty tiles: the operands of tile k + 1 travel HBM -> registers while the K steps of tile k run
  TileRegs<S> R;
  constexpr int NV = X3 ? 3 : 1;  // visits per tile
  auto visit_load = [&](int v) {
    const int ph = X3 ? v % 3 : 0;
    tile_load<S>(R, X3 && ph == 0 ? x_lo : x, X3 && ph == 2 ? dy_lo : dy, mask, s_list[1 + v / NV], tiles_x, tiles_y, H, W, Ho, Wo, cin, cout, cb, ib, t,
                 !X3 || ph != 2, !X3 || ph != 1);
  };
  if (n_mine > 0) visit_load(0);
  for (int v = 0; v < NV * n_mine; v++) {
    const int k = v / NV, ph = X3 ? v % 3 : 0;
    uint32_t* rm = s_rm + 4 * (k & 1);  // the tile's row masks: set by the visits that stage dY (the same bits each time)
    tile_store<S>(R, sx, sy, rm, t, !X3 || ph != 2, !X3 || ph != 1);
    __syncthreads();
    if (t < 4 && ph == NV - 1) s_rm[4 * ((k + 1) & 1) + t] = 0u;  // the next tile's set: its writers come behind this iteration's last barrier
    if (v + 1 < NV * n_mine) visit_load(v + 1);
#pragma unroll 1
    for (int ks = 0; ks < 2 * TH; ks++) {  // rolled: unrolled, the steps' reads are hoisted and the accumulators spill
      const int r = ks >> 1, hs = ks & 1;
      if (((rm[r] >> (16 * hs)) & 0xFFFFu) == 0u) continue;  // block-uniform: no active output among these 16 pixels
      const bf16x8 a = tr_frag<0, 4>(ay + (r * 32 + 16 * hs) * WG_PS);
      const uint8_t* bb = bx + (S * r * XW + S * 16 * hs) * WG_PS;
#define PNX_WG_TAP(KY, KX) acc[(KY) * 3 + (KX)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, tr_frag<((KY) * XW + (KX)) * WG_PS, 4 * S>(bb), acc[(KY) * 3 + (KX)], 0, 0, 0);
      PNX_WG_TAP(0, 0) PNX_WG_TAP(0, 1) PNX_WG_TAP(0, 2) PNX_WG_TAP(1, 0) PNX_WG_TAP(1, 1) PNX_WG_TAP(1, 2) PNX_WG_TAP(2, 0) PNX_WG_TAP(2, 1) PNX_WG_TAP(2, 2)
#undef PNX_WG_TAP
    }
    __syncthreads();  // the next tile overwrites the LDS image
  }
  // ---- the workgroup's partial: [pair][group][tap][64 co][64 ci] fp32; D[m][n]: m = (i & 3) + 8 (i >> 2) + 4 (l >> 5), n = l & 31
  float* out = part + ((int64_t)pair * G + blockIdx.x) * 9 * 4096;
#pragma unroll
  for (int k = 0; k < 9; k++)
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int m = 32 * mb + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), n = 32 * nb + (l & 31);
      out[k * 4096 + m * 64 + n] = acc[k][i];
    }
}

// dW[co][ci][tap] = sum over the groups in a FIXED order: 8 slices of the groups (g = slice, slice + 8, ...) are added up by 8 threads per
// element and then combined slice 0..7 -- the same association for every launch; one thread per element walking all 512 partials left the
// GPU at two waves per SIMD of dependent adds
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int cin, int cout, int G) {
  __shared__ float s_sum[8][32];
  const int t = threadIdx.x, el = t & 31, sl = t >> 5;
  const int e = blockIdx.x * 32 + el;  // ((pair * 9 + tap) * 64 + m) * 64 + n
  const int nib = cin >> 6, n_pairs = (cout >> 6) * nib;
  const bool live = e < n_pairs * 9 * 4096;
  const int n = e & 63, m = (e >> 6) & 63, tap = (e >> 12) % 9, pair = (e >> 12) / 9;
  float s = 0.f;
  if (live) {
    const float* p = part + (int64_t)pair * G * 9 * 4096 + tap * 4096 + m * 64 + n;
    for (int g = sl; g < G; g += 8) s += p[(int64_t)g * 9 * 4096];
  }
  s_sum[sl][el] = s;
  __syncthreads();
  if (sl == 0 && live) {
    float r = s_sum[0][el];
#pragma unroll
    for (int k = 1; k < 8; k++) r += s_sum[k][el];
    const int co = 64 * (pair / nib) + m, ci = 64 * (pair % nib) + n;
    dw[((int64_t)co * cin + ci) * 9 + tap] = r;
  }
}

// Weights (Cout, Cin, 3, 3) fp32 / bf16 -> the bf16 MFMA-fragment order of csrc/conv3x3.hip ([tap][cin/16][cout/32][lane = kb*32 + n][8]) in one
// launch; transposed: the weights of the data gradient, wt[ci][co][ky][kx] = w[co][ci][2-ky][2-kx] (the kernel then maps ci -> co channels).
// The host statement (ops.conv3x3_pack_weights: cast, permute, copy, cast; flip + transpose + copy in front for the gradient) is 3-6 launches
// per layer and step of a training run.
template <typename T>
__global__ __launch_bounds__(256) void k_pack_w3x3(const T* __restrict__ w, int cout, int cin, int transposed, uint16_t* __restrict__ out) {
  const int M = transposed ? cin : cout, K = transposed ? cout : cin;  // output / input channels of the packed convolution
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 9 * M * K) return;
  const int e = idx & 7, n = (idx >> 3) & 31, kb = (idx >> 8) & 1;
  int r = idx >> 9;
  const int MT = M >> 5, CB = K >> 4;
  const int mt = r % MT;
  r /= MT;
  const int cb = r % CB, tap = r / CB;
  const int o = mt * 32 + n, i = cb * 16 + kb * 8 + e, ky = tap / 3, kx = tap - 3 * ky;
  float v;
  if (transposed) v = (float)w[(((int64_t)i * cin + o) * 3 + (2 - ky)) * 3 + (2 - kx)];
  else v = (float)w[(((int64_t)o * cin + i) * 3 + ky) * 3 + kx];
  uint32_t u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even (finite weights)
  out[idx] = (uint16_t)(u >> 16);
}

int groups_per_pair(int cin, int cout) {
  const int n_pairs = (cin >> 6) * (cout >> 6);
  const int g = WG_GROUPS / n_pairs;
  return g < 1 ? 1 : g;
}

}  // namespace

extern "C" size_t pnx_conv3x3_wgrad_workspace_bytes(int32_t cin, int32_t cout) {
  if (cin < 64 || cout < 64 || (cin & 63) || (cout & 63)) return 0;
  return (size_t)(cin >> 6) * (cout >> 6) * groups_per_pair(cin, cout) * 9 * 4096 * sizeof(float) + 256;
}

template <int S, bool X3 = false>
int launch_wgrad(const void* x, const void* dy, const uint8_t* mask, float* dw, int batch, int h, int w, int cin, int cout, void* workspace, hipStream_t st,
                 const void* x_lo = nullptr, const void* dy_lo = nullptr) {
  using Geo = WgGeo<S>;
  const int ho = (h - 1) / S + 1, wo = (w - 1) / S + 1;
  const int G = groups_per_pair(cin, cout), n_pairs = (cin >> 6) * (cout >> 6);
  const int64_t n_tiles = (int64_t)batch * ((ho + Geo::TH - 1) / Geo::TH) * ((wo + 31) / 32);
  PNX_REQUIRE(n_tiles < 0x7fffffff && (n_tiles + G - 1) / G <= WG_LIST_MAX, PNX_ERR_UNSUPPORTED, "%lld tiles over %d workgroups: more than %d per workgroup",
              (long long)n_tiles, G, WG_LIST_MAX);
  static bool attr_done = false;
  if (!attr_done) {
    PNX_CHECK_HIP(hipFuncSetAttribute((const void*)k_wgrad64<S, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS));
    attr_done = true;
  }
  k_wgrad64<S, X3><<<dim3((unsigned)G, (unsigned)n_pairs), 256, Geo::LDS, st>>>((const uint16_t*)x, (const uint16_t*)dy, mask, (float*)workspace, batch, h, w, ho,
                                                                                 wo, cin, cout, G, (const uint16_t*)x_lo, (const uint16_t*)dy_lo);
  PNX_LAUNCH_CHECK();
  k_wgrad_reduce<<<(unsigned)((n_pairs * 9 * 4096 + 31) / 32), 256, 0, st>>>((const float*)workspace, dw, cin, cout, G);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

extern "C" int pnx_conv3x3_wgrad_bf16(const void* x, const void* dy, const uint8_t* mask, float* dw, int32_t batch, int32_t h, int32_t w, int32_t cin,
                                      int32_t cout, int32_t stride, void* workspace, size_t workspace_bytes, pnx_stream_t stream) {
  PNX_REQUIRE(x && dy && mask && dw && workspace && batch > 0 && h > 0 && w > 0, PNX_ERR_INVALID, "pnx_conv3x3_wgrad_bf16: bad arguments");
  PNX_REQUIRE(stride == 1 || stride == 2, PNX_ERR_UNSUPPORTED, "stride %d", stride);
  PNX_REQUIRE(cin >= 64 && cout >= 64 && (cin & 63) == 0 && (cout & 63) == 0 && cin <= 512 && cout <= 512, PNX_ERR_UNSUPPORTED,
              "weight gradient for %d -> %d channels (multiples of 64 up to 512)", cin, cout);
  PNX_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)workspace) & 15) == 0, PNX_ERR_INVALID, "16-byte alignment required");
  PNX_REQUIRE(workspace_bytes >= pnx_conv3x3_wgrad_workspace_bytes(cin, cout), PNX_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) return launch_wgrad<1>(x, dy, mask, dw, batch, h, w, cin, cout, workspace, st);
  return launch_wgrad<2>(x, dy, mask, dw, batch, h, w, cin, cout, workspace, st);
}

extern "C" int pnx_conv3x3_pack_weights(const void* w, int32_t dtype, int32_t cout, int32_t cin, int32_t transposed, void* wfrag, pnx_stream_t stream) {
  PNX_REQUIRE(w && wfrag && cout >= 32 && cin >= 16 && (cout & 31) == 0 && (cin & 15) == 0 && (!transposed || ((cin & 31) == 0 && (cout & 15) == 0)),
              PNX_ERR_INVALID, "pnx_conv3x3_pack_weights: %d -> %d channels", cin, cout);
  PNX_REQUIRE(dtype == PNX_F32 || dtype == PNX_BF16, PNX_ERR_UNSUPPORTED, "weights must be fp32 or bf16");
  hipStream_t st = (hipStream_t)stream;
  const int n = 9 * cout * cin;
  if (dtype == PNX_F32) k_pack_w3x3<float><<<(n + 255) / 256, 256, 0, st>>>((const float*)w, cout, cin, transposed, (uint16_t*)wfrag);
  else k_pack_w3x3<__bf16><<<(n + 255) / 256, 256, 0, st>>>((const __bf16*)w, cout, cin, transposed, (uint16_t*)wfrag);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

// fp32 weight gradient from the bf16 halves of x and of the upstream gradient (pnx_split_f32): dW = x_hi dY_hi + x_lo dY_hi + x_hi dY_lo in ONE pass over
// the tiles (k_wgrad64<S, true>), fp32 accumulation throughout, deterministic.  Same shapes, mask and workspace as pnx_conv3x3_wgrad_bf16.
extern "C" int pnx_conv3x3_wgrad_x3(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, const uint8_t* mask, float* dw, int32_t batch,
                                    int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride, void* workspace, size_t workspace_bytes,
                                    pnx_stream_t stream) {
  PNX_REQUIRE(x_hi && x_lo && dy_hi && dy_lo && mask && dw && workspace && batch > 0 && h > 0 && w > 0, PNX_ERR_INVALID, "pnx_conv3x3_wgrad_x3: bad arguments");
  PNX_REQUIRE(stride == 1 || stride == 2, PNX_ERR_UNSUPPORTED, "stride %d", stride);
  PNX_REQUIRE(cin >= 64 && cout >= 64 && (cin & 63) == 0 && (cout & 63) == 0 && cin <= 512 && cout <= 512, PNX_ERR_UNSUPPORTED,
              "weight gradient for %d -> %d channels (multiples of 64 up to 512)", cin, cout);
  PNX_REQUIRE((((uintptr_t)x_hi | (uintptr_t)x_lo | (uintptr_t)dy_hi | (uintptr_t)dy_lo | (uintptr_t)workspace) & 15) == 0, PNX_ERR_INVALID,
              "16-byte alignment required");
  PNX_REQUIRE(workspace_bytes >= pnx_conv3x3_wgrad_workspace_bytes(cin, cout), PNX_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) return launch_wgrad<1, true>(x_hi, dy_hi, mask, dw, batch, h, w, cin, cout, workspace, st, x_lo, dy_lo);
  return launch_wgrad<2, true>(x_hi, dy_hi, mask, dw, batch, h, w, cin, cout, workspace, st, x_lo, dy_lo);
}
