// conv3x3.hip -- masked 3x3 convolution for the dense stand-in of the sparse backbone, gfx950.
//
// One kernel = SubMConv2d / SparseConv2d(k=3, stride s, pad 1) + folded BatchNorm + [residual] + ReLU + active-site mask of
// det3d/models/utils/sparse_conv.py:16-63, on bf16 NHWC tensors:
//        y[b, oy, ox, :] = mask[b, oy, ox] * relu( sum_{ky,kx} x[b, oy*s+ky-1, ox*s+kx-1, :] . W[:, ky, kx, :] + bias [+ res] )
//
// Implicit GEMM on v_mfma_f32_32x32x16_bf16 with M = output channels, N = 32 consecutive output pixels of one image row,
// K = 9 taps x CIN.  The B fragment of a lane is 8 consecutive input channels of one pixel = ONE 16-byte NHWC load (from
// L1/L2: every input pixel is reused by 9 taps and 2-4 channel tiles), the A fragments (weights, pre-arranged on the host in
// fragment order) sit in LDS when they fit (64->64: 72 KiB) and are streamed through L2 otherwise.  A wave owns 4 rows x 32
// columns of output pixels (8 accumulators at COUT=64); rows/tiles without a single active site skip their MFMAs -- that is
// where the sparsity of the BEV map pays in a dense layout -- and the epilogue (bias, residual, ReLU, mask, bf16 pack) writes
// each output line once.  MIOpen needs a conv pass plus a separate elementwise pass for the same result.
#include "pnx_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf2f_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf2f_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t f2bf_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}

// wfrag layout: [kstep = tap * (CIN/16) + cb][mtile][lane][8 bf16], lane = kb*32 + n :
//   W[out = mtile*32 + n][ky][kx][cin = cb*16 + 8*kb + e]   (host: pillarnext_amd/ops.py::conv3x3_pack_weights)
template <int CIN, int COUT, int STRIDE, bool W_LDS>
__global__ __launch_bounds__(256, 2) void k_conv3x3(const uint16_t* __restrict__ x, const uint4* __restrict__ wfrag,
                                                 const float* __restrict__ bias, const uint16_t* __restrict__ res,
                                                 const uint8_t* __restrict__ mask, uint16_t* __restrict__ y, int B, int H, int W, int Ho,
                                                 int Wo, int relu) {
  constexpr int CB = CIN / 16, MT = COUT / 32, KSTEPS = 9 * CB;
  constexpr int NT = (MT <= 2) ? 4 : 2;  // rows of 32 pixels per wave: 8 accumulators either way
  extern __shared__ uint4 s_w[];         // KSTEPS * MT * 64 uint4 when W_LDS
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int px = lane & 31, kb = lane >> 5;
  if (W_LDS) {
    for (int i = threadIdx.x; i < KSTEPS * MT * 64; i += 256) s_w[i] = wfrag[i];
    __syncthreads();
  }
  const int tiles_x = (Wo + 31) >> 5, tiles_y = (Ho + NT - 1) / NT;
  const int64_t n_tiles = (int64_t)B * tiles_y * tiles_x;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int tx = (int)(tile % tiles_x);
    const int ty = (int)((tile / tiles_x) % tiles_y);
    const int b = (int)(tile / ((int64_t)tiles_x * tiles_y));
    const int ox = tx * 32 + px, oy0 = ty * NT;
    // ---- active sites of the tile
    bool act[NT];
    bool any_row[NT];
    bool any = false;
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int oy = oy0 + j;
      const bool in = ox < Wo && oy < Ho;
      act[j] = in && (mask == nullptr || mask[((int64_t)b * Ho + oy) * Wo + ox] != 0);
      any_row[j] = __ballot(act[j]) != 0;
      any = any || any_row[j];
    }
    v16f acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[j][m][i] = 0.f;

    if (any) {
      for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const uint16_t* src[NT];
        bool ok[NT];
#pragma unroll
        for (int j = 0; j < NT; j++) {
          const int iy = (oy0 + j) * STRIDE + dy, ix = ox * STRIDE + dx;
          ok[j] = any_row[j] && ox < Wo && (oy0 + j) < Ho && iy >= 0 && iy < H && ix >= 0 && ix < W;
          src[j] = x + (((int64_t)b * H + (ok[j] ? iy : 0)) * W + (ok[j] ? ix : 0)) * CIN + 8 * kb;
        }
#pragma unroll
        for (int cb = 0; cb < CB; cb++) {
          const int ks = tap * CB + cb;
          bf16x8 af[MT];
#pragma unroll
          for (int m = 0; m < MT; m++) {
            const uint4 wq = W_LDS ? s_w[(ks * MT + m) * 64 + lane] : wfrag[(ks * MT + m) * 64 + lane];
            af[m] = __builtin_bit_cast(bf16x8, wq);
          }
#pragma unroll
          for (int j = 0; j < NT; j++) {
            if (!any_row[j]) continue;  // wave-uniform
            uint4 q = make_uint4(0, 0, 0, 0);
            if (ok[j]) q = *reinterpret_cast<const uint4*>(src[j] + cb * 16);
            const bf16x8 bfr = __builtin_bit_cast(bf16x8, q);
#pragma unroll
            for (int m = 0; m < MT; m++) acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr, acc[j][m], 0, 0, 0);
          }
        }
      }
    }
    // ---- epilogue: lane = (pixel px, half kb); register i of tile m = out channel m*32 + (i&3) + 8*(i>>2) + 4*kb
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int oy = oy0 + j;
      if (!(ox < Wo && oy < Ho)) continue;
      const int64_t o = (((int64_t)b * Ho + oy) * Wo + ox) * COUT;
#pragma unroll
      for (int m = 0; m < MT; m++) {
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
          const int c0 = m * 32 + 8 * gq + 4 * kb;
          uint2 p = make_uint2(0, 0);
          if (act[j]) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + c0);
            float v0 = acc[j][m][4 * gq + 0] + bv.x, v1 = acc[j][m][4 * gq + 1] + bv.y;
            float v2 = acc[j][m][4 * gq + 2] + bv.z, v3 = acc[j][m][4 * gq + 3] + bv.w;
            if (res != nullptr) {
              const uint2 r = *reinterpret_cast<const uint2*>(res + o + c0);
              v0 += bf2f_lo(r.x);
              v1 += bf2f_hi(r.x);
              v2 += bf2f_lo(r.y);
              v3 += bf2f_hi(r.y);
            }
            if (relu) {
              v0 = fmaxf(v0, 0.f);
              v1 = fmaxf(v1, 0.f);
              v2 = fmaxf(v2, 0.f);
              v3 = fmaxf(v3, 0.f);
            }
            p.x = f2bf_rne(v0) | (f2bf_rne(v1) << 16);
            p.y = f2bf_rne(v2) | (f2bf_rne(v3) << 16);
          }
          *reinterpret_cast<uint2*>(y + o + c0) = p;
        }
      }
    }
  }
}

// Stride-1 variant with the INPUT tile staged in LDS: a workgroup owns TH x 32 output pixels; the (TH+2) x 34 halo tile of
// 64 input channels (128 B per pixel) is loaded once with coalesced 16-byte loads, XOR-swizzled (chunk ^ (column & 7)) so the
// ds_read_b128 of a B fragment does not pile onto one bank column, and then serves all 9 taps of all 4 waves -- the direct
// kernel above re-reads every pixel line 9 times through L1/L2 (measured: 4.8 GB of L2->L1 traffic per 1440x1440 frame).
// Weights (fragment order, 1 KiB per wave-load, shared by every wave on the chip) come straight from L2.  CIN > 64 is
// processed as successive 64-channel slabs; COUT > 64 as successive 64-channel passes over the same staged tile (the merged
// SepHead convolution 64 -> 384 stages its input once and reuses it six times).
template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void k_conv3x3_lds(const uint16_t* __restrict__ x, const uint4* __restrict__ wfrag,
                                                     const float* __restrict__ bias, const uint16_t* __restrict__ res,
                                                     const uint8_t* __restrict__ mask, uint16_t* __restrict__ y, int B, int H, int W,
                                                     int relu) {
  constexpr int CB = CIN / 16;
  constexpr int MTALL = COUT / 32;           // 32-channel output tiles in total
  constexpr int MT = 2;                      // ... handled two at a time (64 output channels per pass over the staged tile)
  constexpr int NT = 4;                      // rows per wave
  constexpr int TH = 4 * NT;                 // rows per workgroup
  constexpr int HW_ = 34;                    // halo tile width
  constexpr int NSLAB = CIN / 64;            // 64-channel input slabs (LDS holds them all: 78 KiB each -> CIN 64 only; else one at a time)
  __shared__ uint4 s_in[(TH + 2) * HW_ * 8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int px = lane & 31, kb = lane >> 5;
  const int tiles_x = (W + 31) >> 5, tiles_y = (H + TH - 1) / TH;
  const int64_t n_tiles = (int64_t)B * tiles_y * tiles_x;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int tx = (int)(tile % tiles_x);
    const int ty = (int)((tile / tiles_x) % tiles_y);
    const int b = (int)(tile / ((int64_t)tiles_x * tiles_y));
    const int x0 = tx * 32, y0 = ty * TH;
    const int ox = x0 + px, oy0 = y0 + wv * NT;
    bool act[NT], any_row[NT];
    bool any = false;
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int oy = oy0 + j;
      const bool in = ox < W && oy < H;
      act[j] = in && (mask == nullptr || mask[((int64_t)b * H + oy) * W + ox] != 0);
      any_row[j] = __ballot(act[j]) != 0;
      any = any || any_row[j];
    }
    const bool wg_any = __syncthreads_or(any ? 1 : 0) != 0;  // also: everybody is done with the previous tile's LDS

    for (int mg = 0; mg < MTALL; mg += MT) {  // 64 output channels per pass
      v16f acc[NT][MT];
#pragma unroll
      for (int j = 0; j < NT; j++)
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
          for (int i = 0; i < 16; i++) acc[j][m][i] = 0.f;
      if (wg_any) {
        for (int slab = 0; slab < NSLAB; slab++) {
          const int ch0 = slab * 64;
          if (NSLAB > 1 || mg == 0) {  // with a single slab the staged tile serves every output-channel pass
            if (slab > 0 || mg > 0) __syncthreads();
#pragma unroll 5
            for (int idx = threadIdx.x; idx < (TH + 2) * HW_ * 8; idx += 256) {
              const int chunk = idx & 7, pix = idx >> 3;
              const int r = pix / HW_, c = pix - r * HW_;
              const int iy = y0 - 1 + r, ix = x0 - 1 + c;
              uint4 q = make_uint4(0, 0, 0, 0);
              if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                q = *reinterpret_cast<const uint4*>(x + (((int64_t)b * H + iy) * W + ix) * CIN + ch0 + chunk * 8);
              s_in[pix * 8 + (chunk ^ (c & 7))] = q;
            }
            __syncthreads();
          }
          if (any) {
            // weight fragments of one tap (4 k-steps x 2 channel tiles) are fetched one tap ahead of their MFMAs
            uint4 wn[4][MT];
#pragma unroll
            for (int cbl = 0; cbl < 4; cbl++)
#pragma unroll
              for (int m = 0; m < MT; m++) wn[cbl][m] = wfrag[((0 * CB + (ch0 >> 4) + cbl) * MTALL + mg + m) * 64 + lane];
            for (int tap = 0; tap < 9; tap++) {  // not unrolled: a full unroll spills (measured: 484 B scratch, 25 % slower)
              const int dy = tap / 3, dx = tap % 3;  // halo coordinates: +1 already included
              const int c = px + dx;
              uint4 wc[4][MT];
#pragma unroll
              for (int cbl = 0; cbl < 4; cbl++)
#pragma unroll
                for (int m = 0; m < MT; m++) wc[cbl][m] = wn[cbl][m];
              if (tap < 8) {
#pragma unroll
                for (int cbl = 0; cbl < 4; cbl++)
#pragma unroll
                  for (int m = 0; m < MT; m++) wn[cbl][m] = wfrag[(((tap + 1) * CB + (ch0 >> 4) + cbl) * MTALL + mg + m) * 64 + lane];
              }
#pragma unroll
              for (int cbl = 0; cbl < 4; cbl++) {
#pragma unroll
                for (int j = 0; j < NT; j++) {
                  if (!any_row[j]) continue;  // wave-uniform
                  const uint4 q = s_in[((wv * NT + j + dy) * HW_ + c) * 8 + ((cbl * 2 + kb) ^ (c & 7))];
                  const bf16x8 bfr = __builtin_bit_cast(bf16x8, q);
#pragma unroll
                  for (int m = 0; m < MT; m++)
                    acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wc[cbl][m]), bfr, acc[j][m], 0, 0, 0);
                }
              }
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const int oy = oy0 + j;
        if (!(ox < W && oy < H)) continue;
        const int64_t o = (((int64_t)b * H + oy) * W + ox) * COUT;
#pragma unroll
        for (int m = 0; m < MT; m++) {
#pragma unroll
          for (int gq = 0; gq < 4; gq++) {
            const int c0 = (mg + m) * 32 + 8 * gq + 4 * kb;
            uint2 p = make_uint2(0, 0);
            if (act[j]) {
              const float4 bv = *reinterpret_cast<const float4*>(bias + c0);
              float v0 = acc[j][m][4 * gq + 0] + bv.x, v1 = acc[j][m][4 * gq + 1] + bv.y;
              float v2 = acc[j][m][4 * gq + 2] + bv.z, v3 = acc[j][m][4 * gq + 3] + bv.w;
              if (res != nullptr) {
                const uint2 r = *reinterpret_cast<const uint2*>(res + o + c0);
                v0 += bf2f_lo(r.x);
                v1 += bf2f_hi(r.x);
                v2 += bf2f_lo(r.y);
                v3 += bf2f_hi(r.y);
              }
              if (relu) {
                v0 = fmaxf(v0, 0.f);
                v1 = fmaxf(v1, 0.f);
                v2 = fmaxf(v2, 0.f);
                v3 = fmaxf(v3, 0.f);
              }
              p.x = f2bf_rne(v0) | (f2bf_rne(v1) << 16);
              p.y = f2bf_rne(v2) | (f2bf_rne(v3) << 16);
            }
            *reinterpret_cast<uint2*>(y + o + c0) = p;
          }
        }
      }
    }
  }
}

template <int CIN, int COUT>
int launch_lds(const void* x, const void* wfrag, const float* bias, const void* res, const uint8_t* mask, void* y, int B, int H, int W, int relu,
               hipStream_t st) {
  constexpr int TH = 16;
  const int64_t n_tiles = (int64_t)B * ((H + TH - 1) / TH) * ((W + 31) / 32);
  int64_t nb = n_tiles;
  const int64_t cap = 256 * 2;  // resident workgroups (LDS: 78 KiB per workgroup)
  if (nb > cap) nb = cap;
  k_conv3x3_lds<CIN, COUT><<<(unsigned)nb, 256, 0, st>>>((const uint16_t*)x, (const uint4*)wfrag, bias, (const uint16_t*)res, mask, (uint16_t*)y, B,
                                                        H, W, relu);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

template <int CIN, int COUT, int STRIDE>
int launch(const void* x, const void* wfrag, const float* bias, const void* res, const uint8_t* mask, void* y, int B, int H, int W, int Ho,
           int Wo, int relu, hipStream_t st) {
  constexpr size_t wbytes = (size_t)9 * (CIN / 16) * (COUT / 32) * 64 * 16;
  constexpr bool W_LDS = wbytes <= 76 * 1024;
  constexpr int NT = (COUT / 32 <= 2) ? 4 : 2;
  const int64_t n_tiles = (int64_t)B * ((Ho + NT - 1) / NT) * ((Wo + 31) / 32);
  int64_t nb = (n_tiles + 3) / 4;
  if (nb > 512) nb = 512;  // persistent: 2 workgroups per CU
  auto kern = k_conv3x3<CIN, COUT, STRIDE, W_LDS>;
  if (W_LDS) {
    static bool attr_done = false;
    if (!attr_done) {
      PNX_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wbytes));
      attr_done = true;
    }
  }
  kern<<<(unsigned)nb, 256, W_LDS ? wbytes : 0, st>>>((const uint16_t*)x, (const uint4*)wfrag, bias, (const uint16_t*)res, mask, (uint16_t*)y, B, H, W,
                                                    Ho, Wo, relu);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // namespace

extern "C" {

int pnx_conv3x3_bf16(const void* x, const void* wfrag, const float* bias, const void* residual, const uint8_t* mask, void* y, int32_t batch,
                     int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride, int32_t relu, pnx_stream_t stream) {
  PNX_REQUIRE(x && wfrag && bias && y && batch > 0 && h > 0 && w > 0, PNX_ERR_INVALID, "bad arguments");
  PNX_REQUIRE(stride == 1 || stride == 2, PNX_ERR_UNSUPPORTED, "stride %d", stride);
  PNX_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)wfrag | (uintptr_t)bias | (uintptr_t)residual) & 15) == 0, PNX_ERR_INVALID,
              "16-byte alignment required");
  const int ho = (h + 2 - 3) / stride + 1, wo = (w + 2 - 3) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1 && getenv("PNX_CONV_DIRECT") == nullptr) {
    if (cin == 64 && cout == 64) return launch_lds<64, 64>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, st);
    if (cin == 128 && cout == 128) return launch_lds<128, 128>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, st);
    if (cin == 64 && cout == 384) return launch_lds<64, 384>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, st);
    if (cin == 64 && cout == 320) return launch_lds<64, 320>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, st);
    if (cin == 64 && cout == 448) return launch_lds<64, 448>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, st);
  }
#define PNX_CONV_CASE(CI, CO)                                                                                          \
  if (cin == CI && cout == CO) {                                                                                       \
    if (stride == 1) return launch<CI, CO, 1>(x, wfrag, bias, residual, mask, y, batch, h, w, ho, wo, relu, st);         \
    return launch<CI, CO, 2>(x, wfrag, bias, residual, mask, y, batch, h, w, ho, wo, relu, st);                         \
  }
  PNX_CONV_CASE(64, 64)
  PNX_CONV_CASE(64, 128)
  PNX_CONV_CASE(128, 128)
#undef PNX_CONV_CASE
  pnx_set_error("pnx_conv3x3_bf16: no kernel for %d -> %d channels", cin, cout);
  return PNX_ERR_UNSUPPORTED;
}

}  // extern "C"
