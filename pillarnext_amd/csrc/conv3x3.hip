// conv3x3.hip -- masked 3x3 convolution for the dense stand-in of the sparse backbone, gfx950.
//
// One kernel = SubMConv2d / SparseConv2d(k=3, stride s, pad 1) + folded BatchNorm + [residual] + ReLU + active-site mask of
// det3d/models/utils/sparse_conv.py:16-63, on bf16 NHWC tensors:
//        y[b, oy, ox, :] = mask[b, oy, ox] * relu( sum_{ky,kx} x[b, oy*s+ky-1, ox*s+kx-1, :] . W[:, ky, kx, :] + bias [+ res] )
//
// Implicit GEMM on v_mfma_f32_32x32x16_bf16 with M = output channels, N = 32 consecutive output pixels of one image row,
// K = 9 taps x CIN.  The B fragment of a lane is 8 consecutive input channels of one pixel = ONE 16-byte NHWC read, the A
// fragments (weights) are pre-arranged on the host in fragment order.  Kernels in this file:
//   k_conv3x3          direct: B fragments straight from L1/L2 (strided layers; every input line is re-read 9 times)
//   k_conv3x3_pc       (conv_pc.h, round 6; default for 64 -> 64) producer / consumer form: 4 producer waves stage the NEXT tile by LDS-DMA into a ring of
//                      slots while 8 consumer waves (row groups x 32-channel groups) run the taps; one workgroup per CU
//   k_conv3x3_lds      stride 1, 64 input channels: 18x34 halo tile staged once in LDS, 1..7 passes of 64 output channels
//   k_conv3x3_ldsx     stride 1, 128 -> 128 and 256 -> 256: 10x34 tile, 64-channel input slabs through one LDS buffer under live
//                      accumulators, 128 output channels per pass
//   k_sephead_out      block-diagonal 16-output convolution closing the merged SepHead branches (16x16x32 MFMA)
// Rows/tiles without an active site skip their MFMAs (and, with row_dirty, their HBM traffic) -- that is where the sparsity of
// the BEV map pays in a dense layout; bias and residual start the accumulators, the epilogue (ReLU, mask, bf16 pack) writes
// complete 128-byte lines.  MIOpen needs a conv pass plus a separate elementwise pass for the same result.  DESIGN.md section 4
// has the measurements each of these choices came from.
#include "pnx_common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Element type of activations and weights.  This file is compiled TWICE (pillarnext_amd/build.py): as is for bf16 (entry points pnx_*_bf16,
// plus the type-independent tile-list helpers) and with -DPNX_CONV_F16 for IEEE half (pnx_*_f16: BASELINE configs[4], the fp16 Waymo network).
// Everything between the loads and the stores is fp32 either way; the type shows in exactly four places: the MFMA opcode, the pack of the
// epilogue (round to nearest even), the widening of the residual, and the rounding of the lazy head's intermediate.
#ifdef PNX_CONV_F16
typedef _Float16 el8 __attribute__((ext_vector_type(8)));
typedef _Float16 el2 __attribute__((ext_vector_type(2)));
#define PNX_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define PNX_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define PNX_CONV_FN(name) name##_f16
// fp32 bit patterns of the low / high element of a packed pair
__device__ __forceinline__ uint32_t el_lo_bits(uint32_t w) { return __float_as_uint((float)__builtin_bit_cast(el2, w)[0]); }
__device__ __forceinline__ uint32_t el_hi_bits(uint32_t w) { return __float_as_uint((float)__builtin_bit_cast(el2, w)[1]); }
#else
typedef __bf16 el8 __attribute__((ext_vector_type(8)));
typedef __bf16 el2 __attribute__((ext_vector_type(2)));
#define PNX_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define PNX_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define PNX_CONV_FN(name) name##_bf16
__device__ __forceinline__ uint32_t el_lo_bits(uint32_t w) { return w << 16; }
__device__ __forceinline__ uint32_t el_hi_bits(uint32_t w) { return w & 0xffff0000u; }
#endif
// two fp32 -> one packed pair, round to nearest even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32: one instruction per channel pair)
__device__ __forceinline__ uint32_t pack_el(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, el2));
}
// x rounded to the element type and back (what a value becomes when it goes through a stored intermediate)
__device__ __forceinline__ float round_el(float x) { return __uint_as_float(el_lo_bits(pack_el(x, 0.f))); }

// Epilogue.  The MFMA leaves lane (px, kb) with channels {8g + 4kb + i} of pixel px: four 8-byte pieces per 32-channel tile,
// i.e. 32 scattered 8-byte stores per row of 32 pixels.  Measured, that store pattern -- not the MFMAs -- bounded the kernel
// (42-50 % of wave time: the vector-memory path pays per distinct line an instruction touches, here 32 lines for 512 bytes).
// Two steps fix it:
//  (1) pack_tile: v_permlane32_swap trades the odd 4-channel groups between the two half-waves (MI355X guide T21) so each lane
//      holds 8 consecutive channels; residual add / ReLU / bf16 rounding / active-site mask happen here in fp32;
//  (2) store_row64: the four 16-byte slots a lane holds for one pixel (64 output channels = one 128-byte line, lanes kb=0/1
//      holding the even/odd chunks) are transposed across lanes with ds_bpermute (LDS crossbar, no LDS storage): slot index
//      <-> pixel-octet index, the classic rotate / permute / rotate scheme (16 bpermutes + 64 selects per row).  Afterwards
//      store d of lane L is chunk L&7 of pixel 8d + (L>>3): every store instruction writes 8 complete 128-byte lines.
// Both must be called by all 64 lanes.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// `res` (optional): the residual lines of this lane's pixel for the tile, res[t] = channels 16 t + 8 kb + 0..7 -- exactly the 8 consecutive channels the
// lane holds after the swap, added in fp32 before the ReLU.
__device__ __forceinline__ void pack_tile(const v16f& a, bool act, int relu, uint4 (&out)[2], const uint4* res = nullptr) {
#pragma unroll
  for (int t = 0; t < 2; t++) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[8 * t + i]), __float_as_uint(a[8 * t + 4 + i]), false, false);
      v[i] = __uint_as_float(r.x);
      v[4 + i] = __uint_as_float(r.y);
    }
    if (res != nullptr) {
      const uint4 q = res[t];
      v[0] += __uint_as_float(el_lo_bits(q.x)), v[1] += __uint_as_float(el_hi_bits(q.x)), v[2] += __uint_as_float(el_lo_bits(q.y)), v[3] += __uint_as_float(el_hi_bits(q.y));
      v[4] += __uint_as_float(el_lo_bits(q.z)), v[5] += __uint_as_float(el_hi_bits(q.z)), v[6] += __uint_as_float(el_lo_bits(q.w)), v[7] += __uint_as_float(el_hi_bits(q.w));
    }
    if (relu) {
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = fmaxf(v[i], 0.f);
    }
    uint4 p;
    p.x = pack_el(v[0], v[1]);
    p.y = pack_el(v[2], v[3]);
    p.z = pack_el(v[4], v[5]);
    p.w = pack_el(v[6], v[7]);
    if (!act) p = make_uint4(0, 0, 0, 0);
    out[t] = p;
  }
}

__device__ __forceinline__ void cswap(bool c, uint4& x, uint4& y) {
  const uint4 a = x, b = y;
  x.x = c ? b.x : a.x, x.y = c ? b.y : a.y, x.z = c ? b.z : a.z, x.w = c ? b.w : a.w;
  y.x = c ? a.x : b.x, y.y = c ? a.y : b.y, y.z = c ? a.z : b.z, y.w = c ? a.w : b.w;
}

// R[s], s = 2*(tile within the 64-channel group) + t: chunk 2s + kb of pixel px = lane & 31.  On return R[d] of lane L is chunk
// L & 7 of pixel 8d + (L >> 3).
__device__ __forceinline__ void transpose_row64(uint4 (&R)[4], int lane) {
  const int a_src = (lane & 31) >> 3;       // pixel octet of this lane as a source
  const int s_dst = (lane & 7) >> 1;        // slot this lane stores as a destination
  // rotate: U[k] = R[k ^ a_src]
  cswap(a_src & 1, R[0], R[1]);
  cswap(a_src & 1, R[2], R[3]);
  cswap(a_src & 2, R[0], R[2]);
  cswap(a_src & 2, R[1], R[3]);
  // permute: round k fetches slot register k of source lane (octet k ^ s_dst, pixel-in-octet L>>3, half L&1)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int src = (8 * (k ^ s_dst) + (lane >> 3)) + 32 * (lane & 1);
    R[k].x = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)R[k].x);
    R[k].y = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)R[k].y);
    R[k].z = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)R[k].z);
    R[k].w = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)R[k].w);
  }
  // rotate back: D[d] = V[d ^ s_dst]
  cswap(s_dst & 1, R[0], R[1]);
  cswap(s_dst & 1, R[2], R[3]);
  cswap(s_dst & 2, R[0], R[2]);
  cswap(s_dst & 2, R[1], R[3]);
}

// `row` (wave-uniform) points at channel 0 of the 64-channel group for pixel 0 of the 32-pixel row segment; pixels >= n_valid
// are not stored; CSTRIDE = channels per pixel.  Uniform base + 32-bit lane offset: no per-store 64-bit address arithmetic.
template <int CSTRIDE>
__device__ __forceinline__ void store_row64(const uint4 (&D)[4], uint16_t* __restrict__ row, int n_valid, int lane) {
  const uint32_t voff = (uint32_t)((lane >> 3) * CSTRIDE + (lane & 7) * 8) * 2u;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const int P = 8 * d + (lane >> 3);
    if (P < n_valid) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(row) + voff + (uint32_t)(d * 8 * CSTRIDE * 2)) = D[d];
  }
}

// ---- the same for a wave that holds ONE 32-channel tile (the producer / consumer kernel, the 64 -> 128 stride-2 kernel): 64-byte half lines
// R[t], t = 0, 1: chunk 2t + kb (8 channels = 16 bytes) of the wave's 32-channel group for pixel px = lane & 31 (what pack_tile leaves).
// On return R[d] of lane L is chunk L & 3 of pixel 16 d + (L >> 2): one store instruction writes 16 complete 64-byte half lines.
__device__ __forceinline__ void transpose_row32(uint4 (&R)[2], int lane) {
  cswap((lane & 16) != 0, R[0], R[1]);  // rotate by the pixel half of the source: U[k] = R[k ^ (px >> 4)]
  const int T = (lane >> 1) & 1;        // register (t) this lane wants as a destination
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const int src = 16 * (k ^ T) + (lane >> 2) + 32 * (lane & 1);
    R[k].x = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)R[k].x);
    R[k].y = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)R[k].y);
    R[k].z = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)R[k].z);
    R[k].w = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)R[k].w);
  }
  cswap(T != 0, R[0], R[1]);  // round k delivered the piece of store k ^ T
}

// `row` (wave-uniform): channel 0 of the wave's 32-channel group at pixel 0 of the row segment
template <int CSTRIDE>
__device__ __forceinline__ void store_row32(const uint4 (&D)[2], uint16_t* __restrict__ row, int n_valid, int lane) {
  const uint32_t voff = (uint32_t)((lane >> 2) * CSTRIDE + (lane & 3) * 8) * 2u;
#pragma unroll
  for (int d = 0; d < 2; d++) {
    const int P = 16 * d + (lane >> 2);
    if (P < n_valid) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(row) + voff + (uint32_t)(d * 16 * CSTRIDE * 2)) = D[d];
  }
}

// The accumulators start from the (folded-BN) bias: register i of a lane's 32-channel tile is channel (i&3) + 8*(i>>2) + 4*kb.
// fp32 epilogue of the three-product kernels: the accumulator tile as it is, lane (px, kb) writing channels 8g + 4kb + 0..3 of its pixel (the two half-waves
// complete 32 bytes per pixel and g); inactive sites get zeros.  `row`: channel 0 of the tile at pixel 0 of the row segment.
template <int COUT>
__device__ __forceinline__ void store_tile_f32(const v16f& a, bool act, float* __restrict__ row, int n_valid, int px, int kb) {
  if (px >= n_valid) return;
  float* p = row + (int64_t)px * COUT + 4 * kb;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    float4 v = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
    if (!act) v = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(p + 8 * g) = v;
  }
}

__device__ __forceinline__ v16f bias_tile(const float* __restrict__ bias, int cbase, int kb) {
  v16f r;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const float4 q = *reinterpret_cast<const float4*>(bias + cbase + 8 * g + 4 * kb);
    r[4 * g + 0] = q.x, r[4 * g + 1] = q.y, r[4 * g + 2] = q.z, r[4 * g + 3] = q.w;
  }
  return r;
}

// The residual (identity branch of a BasicBlock) is folded into the accumulators BEFORE the MFMAs, like the bias: its lines are
// loaded 16 bytes per lane (the layout pack_tile produces) and moved into MFMA register order by the same half-wave swap, which
// is its own inverse.  The loads overlap the staging of the input tile and the epilogue is left without a single load -- on
// gfx9 a load's s_waitcnt also waits for every OLDER store, so a load between stores serialises on HBM write acknowledgements.
// `res_px` = residual line of this lane's pixel (+ channel base of the 64-channel group); must be called by all 64 lanes.
__device__ __forceinline__ void load_residual(uint4 (&rq)[2][2], const uint16_t* __restrict__ res_px, bool act, int kb) {
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int t = 0; t < 2; t++) {
      rq[m][t] = make_uint4(0, 0, 0, 0);
      if (act) rq[m][t] = *reinterpret_cast<const uint4*>(res_px + m * 32 + 16 * t + 8 * kb);
    }
}
// the same lines from a WAVE-UNIFORM row pointer (pixel 0 of the row segment, channel base of the 64-channel group) + one per-lane byte offset
// ((px * CSTRIDE + 8 kb) * 2, the same for every row): scalar base + 32-bit lane offset + immediate, no 64-bit pointer per row in vector registers
__device__ __forceinline__ void load_residual_u(uint4 (&rq)[2][2], const uint16_t* __restrict__ row, uint32_t lane_off, bool act) {
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int t = 0; t < 2; t++) {
      rq[m][t] = make_uint4(0, 0, 0, 0);
      if (act) rq[m][t] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(row) + lane_off + (uint32_t)((m * 32 + 16 * t) * 2));
    }
}
__device__ __forceinline__ void add_residual(v16f (&acc2)[2], const uint4 (&rq)[2][2]) {
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const uint4 r = rq[m][t];
      const uint32_t lo[4] = {el_lo_bits(r.x), el_hi_bits(r.x), el_lo_bits(r.y), el_hi_bits(r.y)};
      const uint32_t hi[4] = {el_lo_bits(r.z), el_hi_bits(r.z), el_lo_bits(r.w), el_hi_bits(r.w)};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const u32x2 q = __builtin_amdgcn_permlane32_swap(lo[i], hi[i], false, false);
        acc2[m][8 * t + i] += __uint_as_float(q.x);
        acc2[m][8 * t + 4 + i] += __uint_as_float(q.y);
      }
    }
}

// wfrag layout: [kstep = tap * (CIN/16) + cb][mtile][lane][8 bf16], lane = kb*32 + n :
//   W[out = mtile*32 + n][ky][kx][cin = cb*16 + 8*kb + e]   (host: pillarnext_amd/ops.py::conv3x3_pack_weights)
template <int CIN, int COUT, int STRIDE, bool W_LDS>
__global__ __launch_bounds__(256, 2) void k_conv3x3(const uint16_t* __restrict__ x, const uint4* __restrict__ wfrag,
                                                 const float* __restrict__ bias, const uint16_t* __restrict__ res,
                                                 const uint8_t* __restrict__ mask, uint16_t* __restrict__ y, int B, int H, int W, int Ho,
                                                 int Wo, int relu, uint8_t* __restrict__ row_dirty) {
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
    // rows without an active site are written (as zeros) only when the row segment may hold stale data (see pnx.h: row_dirty)
    bool row_store[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int oy = oy0 + j;
      row_store[j] = oy < Ho;
      if (row_dirty != nullptr && oy < Ho) {
        uint8_t* d = row_dirty + ((int64_t)b * Ho + oy) * tiles_x + tx;
        const bool was = __builtin_amdgcn_readfirstlane((int)*d) != 0;
        row_store[j] = any_row[j] || was;
        if (lane == 0 && was != any_row[j]) *d = any_row[j] ? 1 : 0;
      }
    }
    v16f acc[NT][MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
      const v16f bq = bias_tile(bias, m * 32, kb);
#pragma unroll
      for (int j = 0; j < NT; j++) acc[j][m] = bq;
    }

    if (res != nullptr) {
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const bool in = ox < Wo && oy0 + j < Ho;
        const uint16_t* rp = res + (in ? (((int64_t)b * Ho + oy0 + j) * Wo + ox) * COUT : 0);
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += 2) {
          uint4 rq[2][2];
          load_residual(rq, rp + m0 * 32, act[j], kb);
          v16f a2[2] = {acc[j][m0], acc[j][m0 + 1]};
          add_residual(a2, rq);
          acc[j][m0] = a2[0], acc[j][m0 + 1] = a2[1];
        }
      }
    }

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
          el8 af[MT];
#pragma unroll
          for (int m = 0; m < MT; m++) {
            const uint4 wq = W_LDS ? s_w[(ks * MT + m) * 64 + lane] : wfrag[(ks * MT + m) * 64 + lane];
            af[m] = __builtin_bit_cast(el8, wq);
          }
#pragma unroll
          for (int j = 0; j < NT; j++) {
            if (!any_row[j]) continue;  // wave-uniform
            uint4 q = make_uint4(0, 0, 0, 0);
            if (ok[j]) q = *reinterpret_cast<const uint4*>(src[j] + cb * 16);
            const el8 bfr = __builtin_bit_cast(el8, q);
#pragma unroll
            for (int m = 0; m < MT; m++) acc[j][m] = PNX_MFMA32(af[m], bfr, acc[j][m]);
          }
        }
      }
    }
    // ---- epilogue (see pack_tile / store_row64)
    const int n_valid = Wo - tx * 32;  // pixels of this row segment inside the image (>= 32: all)
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const bool row_in = row_store[j];  // wave-uniform
      if (!row_in) continue;
      uint16_t* row = y + (((int64_t)b * Ho + (oy0 + j)) * Wo + tx * 32) * COUT;
#pragma unroll
      for (int m0 = 0; m0 < MT; m0 += 2) {
        uint4 R[4];
#pragma unroll
        for (int m = 0; m < 2; m++) {
          uint4 pk[2];
          pack_tile(acc[j][m0 + m], act[j], relu, pk);
          R[2 * m] = pk[0], R[2 * m + 1] = pk[1];
        }
        transpose_row64(R, lane);
        store_row64<COUT>(R, row + m0 * 32, row_in ? n_valid : 0, lane);
      }
    }
  }
}

// Stride-1, 64-input-channel variant with the INPUT tile staged in LDS: a workgroup owns 16 x 32 output pixels; the 18 x 34 halo tile of 64
// input channels (128 B per pixel) is loaded once with coalesced 16-byte loads (all loads of a batch in flight before the first
// ds_write), XOR-swizzled so that each 16-lane service group of a ds_read_b128 B-fragment read covers all 64 banks, and then
// serves all 9 taps of all 4 waves -- the direct kernel above re-reads every pixel line 9 times through L1/L2 (measured:
// 4.8 GB of L2->L1 traffic per 1440x1440 frame).  Sparsity: the 16 row segments of a tile that hold at least one active site
// are dealt round-robin to the 4 waves (row indices live in SGPRs), each wave runs a branch-free tap loop specialised on its
// row count NR = 1..4 (LDS reads pipeline ahead of the MFMAs), halo rows no active row needs are not staged, and rows without
// an active site are zero-filled without touching the MFMA pipe.  Weights (fragment order, 1 KiB per wave-load, shared by every
// wave on the chip) come straight from L1/L2, one tap ahead of their use.  CIN > 64 is processed as successive 64-channel
// slabs; COUT > 64 as successive 64-channel passes over the same staged tile (the merged SepHead convolution 64 -> 384 stages
// its input once and reuses it six times).
#ifdef PNX_CONV_TIMERS  // section timers (build with PNX_CONV_TIMERS=1 in the environment of build.py)
__device__ unsigned long long g_conv_T[8];
#define CT_DECL unsigned long long ct_T[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tk = __builtin_amdgcn_s_memtime();
#define CT_TOCK(k)                                              \
  {                                                             \
    const unsigned long long _n = __builtin_amdgcn_s_memtime(); \
    ct_T[k] += _n - tk;                                         \
    tk = _n;                                                    \
  }
#define CT_FLUSH                                                                     \
  if ((threadIdx.x & 63) == 0) {                                                     \
    for (int k = 0; k < 8; k++) atomicAdd(&g_conv_T[k], ct_T[k]);                     \
  }
#else
#define CT_DECL
#define CT_TOCK(k)
#define CT_FLUSH
#endif

constexpr int LDS_TH = 16;    // rows per workgroup tile
constexpr int LDS_HW = 34;    // halo tile width
constexpr int LDS_NSTAGE = (LDS_TH + 2) * LDS_HW * 8;  // uint4 per staged tile (64 input channels)

__device__ __forceinline__ int lds_swz(int c) { return (c & 7) ^ ((c >> 3) & 1); }

// Dynamic tile scheduling.  Tiles cost anything between "read 16 mask rows" and "4 rows x 9 taps of MFMAs per wave"; dealing
// them to the persistent workgroups round-robin leaves the slowest workgroup 36 % above the mean on a LiDAR sweep (simulated
// from the occupancy), a shared counter 3 %.  Thread 0 draws the NEXT tile at the top of an iteration (the
// atomic's latency hides behind the mask loads), everybody reads it after the iteration's first barrier; the LDS word is
// double-buffered by iteration parity because an empty tile has no second barrier.  g_tile_ctr[slot] = {next tile - gridDim.x,
// finished workgroups}; the last workgroup to finish re-arms the slot, the host rotates 64 slots so that launches in flight
// never share one.
__device__ unsigned int g_tile_ctr[64][2];
// slot < 0: static round-robin (dense inputs: every tile costs the same and the ticket only adds latency).
__device__ __forceinline__ void sched_draw(unsigned int* s_next, int it, int slot) {
  if (slot >= 0 && threadIdx.x == 0) s_next[it & 1] = atomicAdd(&g_tile_ctr[slot][0], 1u);
}
__device__ __forceinline__ int64_t sched_next(const unsigned int* s_next, int it, int slot, int64_t tile) {
  return slot >= 0 ? (int64_t)s_next[it & 1] + gridDim.x : tile + gridDim.x;
}
// Depth-2 pipeline of the tile loops: iteration `it` works on index A, already knows index B (its mask bytes are requested during
// A) and draws index C.  Indices 0..2G-1 are dealt statically (G = gridDim.x), the tickets continue from 2G.  (Drawing the first B
// by ticket as well put 2 x 512 same-address atomics at the start of every launch: 272 us instead of 228 us for 256 -> 256 at
// 180 x 180, three runs each, no difference at the larger stages.)  With a tile list (pnx_conv_tile_list: the tiles that
// hold an active site or a stale row) an index is a position in the list.
__device__ __forceinline__ int64_t sched_next2(const unsigned int* s_next, int it, int slot, int64_t idx_b) {
  return slot >= 0 ? (int64_t)s_next[it & 1] + 2 * (int64_t)gridDim.x : idx_b + gridDim.x;
}
__device__ __forceinline__ int64_t tile_at(const int32_t* __restrict__ tlist, int64_t idx, int64_t n) {
  if (idx >= n) return -1;
  return tlist != nullptr ? (int64_t)tlist[idx] : idx;
}
__device__ __forceinline__ void sched_done(int slot) {
  if (slot >= 0 && threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&g_tile_ctr[slot][1], 1u) == gridDim.x - 1) {
      g_tile_ctr[slot][0] = 0;
      g_tile_ctr[slot][1] = 0;
      __threadfence();
    }
  }
}

// Stage the (TH+2) x 34 halo tile of 64 input channels [ch0, ch0+64) of image b at tile origin (y0, x0).  Thread t owns slot t&7
// of halo column t>>3 (0..31) in every row: a wave covers 8 pixels x 8 slots = 1 KiB of LDS that is contiguous in lane order, so
// the rows go global -> LDS directly (global_load_lds_dwordx4: no VGPRs, no ds_write pass, ALL rows in flight at once instead of
// two register batches); the XOR swizzle is applied on the SOURCE side -- slot k of column c receives chunk k ^ swz(c) of the
// pixel's 128-byte line, so coalescing is unchanged.  LDS-DMA cannot write zeros: cells that are needed but lie outside the image
// get a plain ds_write.  The two remaining columns (32, 33; 16 slots per row) keep the register path.  `need` = halo rows that
// will be read (bit r); other rows keep stale data.  Callers follow with __syncthreads(), which drains the DMA (vmcnt(0)).
template <int CSTRIDE, int TH = LDS_TH>
__device__ __forceinline__ void stage_tile64(uint4* __restrict__ s_in, const uint16_t* __restrict__ x, int b, int H, int W, int ch0, int y0, int x0,
                                             uint32_t need) {
  constexpr int NROW = TH + 2, NEXTRA = NROW * 16;
  static_assert(NEXTRA <= 512, "two slots of the extra columns per thread");
#ifdef PNX_CONV_DBG_NOSTAGE  // timing experiment: the tile is not staged (the taps run on whatever LDS holds)
  return;
#endif
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint16_t* xt = x + (((int64_t)b * H + (y0 - 1)) * W + (x0 - 1)) * CSTRIDE + ch0;  // element (0, 0) of the halo tile
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // opaque: keeps per-thread staging addresses from being hoisted out of the tile loop (and spilled)
  const int slot = tid & 7, col = tid >> 3, wbase = (tid >> 6) * 8;  // wbase: first column of this wave
  uint32_t rows_in = (1u << NROW) - 1u;  // halo rows inside the image
  if (y0 == 0) rows_in &= ~1u;
  if (y0 - 1 + NROW > H) rows_in &= (1u << (H - (y0 - 1))) - 1u;
  const bool col_ok = (unsigned)(x0 - 1 + col) < (unsigned)W;
  const uint32_t voff = (uint32_t)(col * CSTRIDE + (slot ^ lds_swz(col)) * 8) * 2u;
#pragma unroll
  for (int r = 0; r < NROW; r++) {
    if (!((need >> r) & 1u)) continue;  // wave-uniform
    if (((rows_in >> r) & 1u) && col_ok) {
      __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(xt + (int64_t)r * W * CSTRIDE) + voff),
                                       (lptr_t)(s_in + (r * LDS_HW + wbase) * 8), 16, 0, 0);
    } else {
      s_in[(r * LDS_HW + col) * 8 + slot] = make_uint4(0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int part = 0; part < (NEXTRA + 255) / 256; part++) {
    const int e = part * 256 + tid;  // slot of the two extra columns: row e>>4, column 32 + ((e>>3)&1), chunk e&7
    const int re = e >> 4, ce = 32 + ((e >> 3) & 1);
    if (e < NEXTRA && ((need >> re) & 1u)) {
      uint4 qe = make_uint4(0, 0, 0, 0);
      if (((rows_in >> re) & 1u) && (unsigned)(x0 - 1 + ce) < (unsigned)W)
        qe = *reinterpret_cast<const uint4*>(xt + ((int64_t)re * W + ce) * CSTRIDE + (e & 7) * 8);
      s_in[(re * LDS_HW + ce) * 8 + ((e & 7) ^ lds_swz(ce))] = qe;
    }
  }
}

// One 64-output-channel pass of a wave over its NR rows: 9 taps x 4 k-steps, software-pipelined in registers -- B fragments one
// k-step ahead (LDS), weight fragments one tap ahead (L1/L2; each register pair is refilled for the next tap right after its
// last MFMA of this tap).
// STRIDE 2 (stage_tile64_s2): a halo row is [even input columns 0..31 | odd input columns 32..64] and has S2_RS slots, so that the 32
// lanes of a tap still read 32 consecutive slots: tap column dx of output pixel px is input column 2 px + dx - 1 (relative to the
// tile), i.e. slot px of the even plane for dx = 1 and slot px + dx/2 of the odd plane otherwise; row dy of output row r is halo
// row 2 r + dy (the caller folds 2 r into rbase).
constexpr int S2_RS = 65;
template <int STRIDE>
__device__ __forceinline__ int tap_slot(int px, int dx) {
  return STRIDE == 1 ? px + dx : (dx == 1 ? px : 32 + px + (dx >> 1));
}
// MB = 32-channel output tiles per wave (2: the row-split kernels; 1: the producer / consumer kernel, whose waves split the output channels).
// This is the ROLLED form of rounds 2-5 (9-iteration tap loop, its `tap < 8` tests compile to branches inside the loop): -DPNX_TAPS_ROLLED selects it.
template <int NR, int MTALL, int CB = 4, int STRIDE = 1, int MB = 2>
__device__ __forceinline__ void conv_taps_rolled(v16f (&acc)[NR][MB], const uint4* __restrict__ s_in, const uint4* __restrict__ wfrag, const int (&rbase)[4],
                                          int mg, int px, int kb, int lane, int kstep0 = 0) {
  constexpr int RS = STRIDE == 1 ? LDS_HW : S2_RS;
  uint4 w[4][MB];
#pragma unroll
  for (int cbl = 0; cbl < 4; cbl++)
#pragma unroll
    for (int m = 0; m < MB; m++) w[cbl][m] = wfrag[((kstep0 + cbl) * MTALL + mg + m) * 64 + lane];
#ifdef PNX_CONV_DBG_SAMEW  // timing experiment (tools/conv_ab.sh): every tap re-reads the fragments of tap 0 (L1-resident) -- results are wrong
#define PNX_W_TAP(tn) 0
#else
#define PNX_W_TAP(tn) (tn)
#endif
  uint4 qn[NR];
  {
    const int c0 = tap_slot<STRIDE>(px, 0), sw = lds_swz(c0);
#pragma unroll
    for (int j = 0; j < NR; j++) qn[j] = s_in[rbase[j] + c0 * 8 + (kb ^ sw)];
  }
#pragma unroll 1
  for (int tap = 0; tap < 9; tap++) {  // not unrolled: a full unroll spills
    const int tn = tap + 1;
    const int dyn = tn / 3, dxn = tn - 3 * dyn;  // next tap (halo coordinates: +1 already included)
    const int dy = tap / 3, dx = tap - 3 * dy;
    const int c = tap_slot<STRIDE>(px, dx), cn = tap_slot<STRIDE>(px, dxn);
    const int sw = lds_swz(c), swn = lds_swz(cn);
    const int cbase = (dy * RS + c) * 8, cbasen = (dyn * RS + cn) * 8;
#pragma unroll
    for (int cbl = 0; cbl < 4; cbl++) {
      uint4 qc[NR];
#pragma unroll
      for (int j = 0; j < NR; j++) qc[j] = qn[j];
      if (cbl < 3) {
        const int chunk = ((cbl + 1) * 2 + kb) ^ sw;
#pragma unroll
        for (int j = 0; j < NR; j++) qn[j] = s_in[rbase[j] + cbase + chunk];
      } else if (tap < 8) {
        const int chunk = kb ^ swn;
#pragma unroll
        for (int j = 0; j < NR; j++) qn[j] = s_in[rbase[j] + cbasen + chunk];
      }
#pragma unroll
      for (int j = 0; j < NR; j++) {
        const el8 bfr = __builtin_bit_cast(el8, qc[j]);
#pragma unroll
        for (int m = 0; m < MB; m++)
          acc[j][m] = PNX_MFMA32(__builtin_bit_cast(el8, w[cbl][m]), bfr, acc[j][m]);
      }
      if (tap < 8) {
#pragma unroll
        for (int m = 0; m < MB; m++) w[cbl][m] = wfrag[((PNX_W_TAP(tn) * CB + kstep0 + cbl) * MTALL + mg + m) * 64 + lane];
      }
      __builtin_amdgcn_sched_barrier(0);  // (round 5, profiles/r05_conv_sched.txt: without this fence 128 / 256 channels run 3-4 % slower; s_setprio around the MFMAs: no gain)
    }
  }
}

// The tap loop, fully unrolled (round 6): 36 k-steps (9 taps x 4 chunks of 16 input channels) of one 64-channel slab, software-pipelined in the source --
// B fragments TAPS_PD k-steps ahead (LDS), weight fragments TAPS_WD k-steps ahead (L1 / L2), both in register rings indexed by compile-time constants
// (no copies, no branches); a sched_barrier per k-step keeps hipcc from sinking the loads back to their users.  What made the full unroll spill in
// round 2 was hoisting: 36 x NR slot addresses and 36 weight pointers are loop-invariant over the persistent tile loop, so hipcc computed all of
// them at kernel entry.  Now the weight address is a wave-uniform base (scalar registers) + lane * 16, and the per-lane part of a slot address goes
// through an opaque asm at every k-step.  Measured (profiles/r06_conv_pc_ab.txt): deeper prefetch (2-3 k-steps, 6-8 fragments) is slower -- the loop was
// never latency-bound, see DESIGN.md section 4 on the power ceiling.
#ifndef TAPS_PD
#define TAPS_PD 1
#endif
#ifndef TAPS_WD
#define TAPS_WD 4
#endif
#ifndef TAPS_NOBUFW  // (-DTAPS_NOBUFW: per-lane global loads, the A/B reference: profiles/r06_conv_pc_ab.txt)
#define TAPS_BUFW
#endif
template <int NR, int MTALL, int CB = 4, int STRIDE = 1, int MB = 2, int WD = TAPS_WD>
__device__ __forceinline__ void conv_taps(v16f (&acc)[NR][MB], const uint4* __restrict__ s_in, const uint4* __restrict__ wfrag, const int (&rbase)[4],
                                          int mg, int px, int kb, int lane, int kstep0 = 0) {
#ifdef PNX_TAPS_ROLLED
  conv_taps_rolled<NR, MTALL, CB, STRIDE, MB>(acc, s_in, wfrag, rbase, mg, px, kb, lane, kstep0);
#else
  constexpr int RS = STRIDE == 1 ? LDS_HW : S2_RS;
  constexpr int KS = 36, PD = TAPS_PD;
  // weight fragment (k-step ks -> tap, chunk cbl; output tile m): a wave-uniform base + lane * 16
  const uint4* wu = wfrag + (int64_t)__builtin_amdgcn_readfirstlane(kstep0 * MTALL + mg) * 64;
#ifdef TAPS_BUFW  // weights through a buffer resource: scalar base + scalar offset + lane * 16, one instruction per fragment and no per-lane 64-bit address
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wu), 0, 0x7fffffff, 0x00020000);
  const int wlo = lane * 16;
#endif
  auto wload = [&](int ks, uint4 (&dst)[MB]) {
    const int grp = ks / 3, cbl = grp & 3;
#ifdef PNX_CONV_DBG_SAMEW
    const int tap = 0;
#else
    const int tap = (ks % 3) * 3 + (grp >> 2);  // dy * 3 + dx
#endif
#pragma unroll
    for (int m = 0; m < MB; m++) {
#ifdef TAPS_BUFW
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(wrs, wlo, (((tap * CB + cbl) * MTALL + m) * 64) * 16, 0);
      dst[m] = make_uint4(r.x, r.y, r.z, r.w);
#else
      dst[m] = wu[((tap * CB + cbl) * MTALL + m) * 64 + lane];
#endif
    }
  };
  // k-step order: ks = (dx * 4 + cbl) * 3 + dy.  The three dy taps of a (column dx, chunk cbl) group read the SAME per-lane slots one halo row apart, i.e.
  // the same address registers with another immediate offset: NR address adds per three k-steps instead of per k-step.  The per-lane slot of a group --
  // c * 8 + ((2 cbl + kb) ^ swz(c)), c = tap_slot(px, dx) -- is recomputed from an OPAQUE copy of px at every group (6 VALU per 3 k-steps): written as
  // loop-invariant arrays hipcc kept them (and, before that, all 36 x NR slot addresses) live across the persistent tile loop and spilled.
  int addr[NR];
  auto group_addr = [&](int grp) {
    const int dx = grp >> 2, cbl = grp & 3;
    int p = px;
    asm volatile("" : "+v"(p));
    const int c = tap_slot<STRIDE>(p, dx);
    const int a = c * 8 + ((2 * cbl + kb) ^ lds_swz(c));
#pragma unroll
    for (int j = 0; j < NR; j++) addr[j] = a + rbase[j];
  };
  auto bload = [&](int ks, uint4 (&dst)[NR]) {
    const int dy = ks % 3;
    if (dy == 0) group_addr(ks / 3);
#pragma unroll
    for (int j = 0; j < NR; j++) dst[j] = s_in[addr[j] + dy * (RS * 8)];
  };
  uint4 w[WD][MB];
  uint4 q[PD + 1][NR];
#pragma unroll
  for (int k = 0; k < WD; k++) wload(k, w[k]);
#pragma unroll
  for (int k = 0; k < PD; k++) bload(k, q[k]);
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
    if (ks + PD < KS) bload(ks + PD, q[(ks + PD) % (PD + 1)]);
#pragma unroll
    for (int j = 0; j < NR; j++) {
      const el8 bfr = __builtin_bit_cast(el8, q[ks % (PD + 1)][j]);
#pragma unroll
      for (int m = 0; m < MB; m++) acc[j][m] = PNX_MFMA32(__builtin_bit_cast(el8, w[ks % WD][m]), bfr, acc[j][m]);
    }
    if (ks + WD < KS) wload(ks + WD, w[ks % WD]);
#ifndef TAPS_NOSB  // (measured, profiles/r06_conv_pc_ab.txt: without the fence -1.5 % end to end; an explicit one-load-behind-each-MFMA sched_group_barrier pattern: no difference)
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
#endif
}

// Everything a wave does for its NR active rows once the tile is staged: per 64-channel pass, accumulators = bias (+ residual),
// the taps, and the epilogue.  No barrier inside.
template <int NR, int COUT, bool HAS_RES>
__device__ __forceinline__ void conv_rows(const uint4* __restrict__ s_in, const uint4* __restrict__ wfrag, const float* __restrict__ bias,
                                          const uint16_t* const (&rrow_res)[4], uint32_t res_off, const int (&rbase)[4], const uint32_t (&rmask)[4],
                                          uint16_t* const (&yrow)[4], int n_valid, int relu, int px, int kb, int lane) {
  constexpr int MTALL = COUT / 32;
#pragma unroll 1
  for (int mg = 0; mg < MTALL; mg += 2) {  // 64 output channels per pass over the staged tile
    v16f acc[NR][2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
      const v16f bq = bias_tile(bias, (mg + m) * 32, kb);
#pragma unroll
      for (int j = 0; j < NR; j++) acc[j][m] = bq;
    }
#ifdef PNX_CONV_RES_EARLY  // round 2-4 form: the residual starts the accumulators (64 registers held across the staging: 212 B/lane of scratch)
    if (HAS_RES) {
#pragma unroll
      for (int j = 0; j < NR; j++) {
        uint4 rq[2][2];
        load_residual_u(rq, rrow_res[j], res_off, (rmask[j] >> px) & 1u);
        add_residual(acc[j], rq);
      }
    }
#endif
    // The residual (HAS_RES: one 64-channel pass by construction) joins in the epilogue: the lines of row 0 are requested here and arrive under the
    // tap loop (16 registers), the lines of row j + 1 are requested before row j is packed and stored -- a load is never issued BEHIND a store it
    // then has to wait for (gfx9 vmcnt is in-order), and nothing is held across the staging of the tile.
    uint4 rq[2][2];
#ifndef PNX_CONV_RES_EARLY
    if (HAS_RES) load_residual_u(rq, rrow_res[0], res_off, rmask[0] >> px & 1u);
#endif
    conv_taps<NR, MTALL, 4, 1, 2, COUT == 64 ? 2 : TAPS_WD>(acc, s_in, wfrag, rbase, mg, px, kb, lane);  // 64 -> 64 (the cross-check of k_conv3x3_pc): a two-deep weight ring keeps it at 0 scratch
#pragma unroll
    for (int j = 0; j < NR; j++) {
      const bool act = (rmask[j] >> px) & 1u;
      uint4 rc[2][2];
#ifndef PNX_CONV_RES_EARLY
      if (HAS_RES) {
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
          for (int t = 0; t < 2; t++) rc[m][t] = rq[m][t];
        if (j + 1 < NR) load_residual_u(rq, rrow_res[j + 1], res_off, rmask[j + 1] >> px & 1u);
      }
#endif
      uint4 D[4];
#pragma unroll
      for (int m = 0; m < 2; m++) {
        uint4 pk[2];
#ifndef PNX_CONV_RES_EARLY
        pack_tile(acc[j][m], act, relu, pk, HAS_RES ? rc[m] : nullptr);
#else
        pack_tile(acc[j][m], act, relu, pk);
#endif
        D[2 * m] = pk[0], D[2 * m + 1] = pk[1];
      }
      transpose_row64(D, lane);
      store_row64<COUT>(D, yrow[j] + mg * 32, n_valid, lane);
    }
  }
}

template <int COUT, bool HAS_RES>
__global__ __launch_bounds__(256, 2) void k_conv3x3_lds(const uint16_t* __restrict__ x, const uint4* __restrict__ wfrag,
                                                     const float* __restrict__ bias, const uint16_t* __restrict__ res,
                                                     const uint8_t* __restrict__ mask, uint16_t* __restrict__ y, int B, int H, int W,
                                                     int relu, uint8_t* __restrict__ row_dirty, int slot, const int32_t* __restrict__ tlist,
                                                      const int32_t* __restrict__ tcount) {
  constexpr int CIN = 64;
  constexpr int TH = LDS_TH, HW_ = LDS_HW;
  static_assert(!HAS_RES || COUT == 64, "the residual is folded into the accumulators of a single 64-channel pass");
  __shared__ uint4 s_in[LDS_NSTAGE];
  __shared__ uint32_t s_rowmask2[2 * TH];  // double-buffered by iteration parity: an empty tile has a single barrier (see s_next)
  __shared__ unsigned int s_next[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int px = lane & 31, kb = lane >> 5;
  const int tiles_x = (W + 31) >> 5, tiles_y = (H + TH - 1) / TH;
  const int64_t n_tiles = tlist != nullptr ? (int64_t)__builtin_amdgcn_readfirstlane(*tcount) : (int64_t)B * tiles_y * tiles_x;
  // requests the mask bytes / row_dirty flags of the rows this wave looks at in tile t (t < 0: none)
  auto load_mask = [&](int64_t t, bool (&a)[4], int (&wz)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) a[j] = false, wz[j] = 1;
    if (t < 0) return;
    const int tx = (int)(t % tiles_x);
    const int ty = (int)((t / tiles_x) % tiles_y);
    const int b = (int)(t / ((int64_t)tiles_x * tiles_y));
    const int ox = tx * 32 + px;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int oy = ty * TH + wv * 4 + j;
      a[j] = ox < W && oy < H && (mask == nullptr || mask[((int64_t)b * H + oy) * W + ox] != 0);
      if (row_dirty != nullptr && oy < H) wz[j] = (int)row_dirty[((int64_t)b * H + oy) * tiles_x + tx];
    }
  };
  int64_t idxB = (int64_t)blockIdx.x + gridDim.x, idxC = 0;
  int64_t tileA = tile_at(tlist, blockIdx.x, n_tiles), tileB = tile_at(tlist, idxB, n_tiles), tileC = -1;
  bool aP[4], aN[4];
  int wasP[4], wasN[4];
  load_mask(tileA, aP, wasP);
#pragma unroll
  for (int j = 0; j < 4; j++) aN[j] = false, wasN[j] = 1;
  CT_DECL
  int it = 0;
  for (; tileA >= 0; tileA = tileB, tileB = tileC, idxB = idxC, it++) {
    const int64_t tile = (int64_t)__builtin_amdgcn_readfirstlane((int)tileA);  // uniform by construction: everything derived from it (origins, row pointers) lives in scalar registers
    sched_draw(s_next, it, slot);
    uint32_t* const s_rowmask = s_rowmask2 + (it & 1) * TH;
    const int tx = (int)(tile % tiles_x);
    const int ty = (int)((tile / tiles_x) % tiles_y);
    const int b = (int)(tile / ((int64_t)tiles_x * tiles_y));
    const int x0 = tx * 32, y0 = ty * TH;
    const int ox = x0 + px;
    CT_TOCK(7)
    bool was[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {  // active sites: one 32-bit column mask per row of the tile, from the bytes requested one tile ago
      was[j] = __builtin_amdgcn_readfirstlane(wasP[j]) != 0;
      const uint32_t bal = (uint32_t)__ballot(aP[j]);
      if (lane == 0) s_rowmask[wv * 4 + j] = bal;
    }
    __syncthreads();  // row masks visible; everybody is done reading the previous tile's s_in
    idxC = sched_next2(s_next, it, slot, idxB);
    tileC = tile_at(tlist, idxC, n_tiles);
    load_mask(tileB, aN, wasN);  // consumed at the top of the next iteration
#pragma unroll
    for (int j = 0; j < 4; j++) aP[j] = aN[j], wasP[j] = wasN[j];
    CT_TOCK(0)
    const uint32_t my_rm = s_rowmask[lane & 15];
    const uint32_t am = (uint32_t)__ballot(my_rm != 0) & 0xffffu;  // rows with an active site (wave-uniform, same in all waves)
    // ---- rows without any active site: zero-fill where needed (natural rows of this wave)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int rr = wv * 4 + j, oy = y0 + rr;
      const bool active = (am >> rr) & 1u;
      if (!active && was[j] && oy < H && ox < W) {
        uint4* dst = reinterpret_cast<uint4*>(y + (((int64_t)b * H + oy) * W + ox) * COUT);
#pragma unroll 1
        for (int ch = kb; ch < COUT / 8; ch += 2) dst[ch] = make_uint4(0, 0, 0, 0);
      }
      if (row_dirty != nullptr && oy < H && lane == 0 && was[j] != active) row_dirty[((int64_t)b * H + oy) * tiles_x + tx] = active ? 1 : 0;
    }
    if (am == 0) continue;  // uniform over the workgroup
    // ---- the active rows, dealt round-robin to the waves (wave wv takes the active rows number wv, wv+4, ...)
    int nr = 0;
    int rbase[4], rrow[4];
    {
      uint32_t rest = am;
      for (int k = 0; k < wv && rest; k++) rest &= rest - 1;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool has = rest != 0;
        rrow[j] = has ? __builtin_ctz(rest) : 0;  // dummy rows point at row 0
        nr += has ? 1 : 0;
        for (int k = 0; k < 4 && rest; k++) rest &= rest - 1;
      }
    }
    nr = __builtin_amdgcn_readfirstlane(nr);
    uint32_t rmask[4];
    uint16_t* yrow[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      rrow[j] = __builtin_amdgcn_readfirstlane(rrow[j]);
      rbase[j] = rrow[j] * HW_ * 8;
      rmask[j] = j < nr ? __builtin_amdgcn_readfirstlane(s_rowmask[rrow[j]]) : 0u;
      yrow[j] = y + (((int64_t)b * H + (y0 + rrow[j])) * W + x0) * COUT;
    }
    const uint32_t need = am | (am << 1) | (am << 2);  // halo rows some active row reads
    CT_TOCK(1)
    // ---- residual lines of this lane's pixel in the wave's rows (requested inside conv_rows)
    const uint16_t* rres[4];  // wave-uniform: pixel 0 of the row segment
#pragma unroll
    for (int j = 0; j < 4; j++) rres[j] = HAS_RES ? res + (((int64_t)b * H + (y0 + rrow[j])) * W + x0) * COUT : nullptr;
    const uint32_t res_off = (uint32_t)(px * COUT + 8 * kb) * 2u;
    stage_tile64<CIN>(s_in, x, b, H, W, 0, y0, x0, need);
    CT_TOCK(2)
    __syncthreads();
    CT_TOCK(3)
    const int n_valid = W - x0;
    switch (nr) {  // wave-uniform
      case 1: conv_rows<1, COUT, HAS_RES>(s_in, wfrag, bias, rres, res_off, rbase, rmask, yrow, n_valid, relu, px, kb, lane); break;
      case 2: conv_rows<2, COUT, HAS_RES>(s_in, wfrag, bias, rres, res_off, rbase, rmask, yrow, n_valid, relu, px, kb, lane); break;
      case 3: conv_rows<3, COUT, HAS_RES>(s_in, wfrag, bias, rres, res_off, rbase, rmask, yrow, n_valid, relu, px, kb, lane); break;
      case 4: conv_rows<4, COUT, HAS_RES>(s_in, wfrag, bias, rres, res_off, rbase, rmask, yrow, n_valid, relu, px, kb, lane); break;
      default: break;
    }
    CT_TOCK(4)
  }
  sched_done(slot);
  CT_FLUSH
}

// ---- CIN -> COUT channels in multiples of 64 / 128, stride 1: 8 x 32 pixel tiles, 64-channel input slabs through one LDS buffer
constexpr int L128_TH = 8;
constexpr int L128_NSTAGE = (L128_TH + 2) * LDS_HW * 8;

int next_sched_slot() {
  static unsigned int n = 0;  // one host thread per process drives the launches (pnx.h: not thread-safe)
  return (int)(n++ & 63u);
}

// 128 -> 128 (stage 1) and 256 -> 256 (stages 2 and 3 of the backbone, sparse_resnet.py:50-68).  Per tile, COUT/128 passes (the 4
// waves = 2 row groups x 2 groups of 64 output channels, 4 rows x 64 channels of accumulators each) over CIN/64 input slabs that
// go through the one 43.5 KiB LDS buffer under live accumulators, so 2 workgroups share a CU.  Every (pass, slab) step is
// bracketed by the same two barriers in every wave, whatever its row count (s_barrier counts arrivals, not program counters).
// The residual of a pass is loaded right before it is added: holding the lines from the top of the tile (as the 64-channel
// kernel does) cost 64-128 VGPRs and ~90 spilled registers here -- measured on the stage-1 LiDAR mask, 128 -> 128 with residual:
// 653 us with early loads, 567 us with late ones.
template <int NR, int CIN, int COUT, bool HAS_RES>
__device__ __forceinline__ void conv_rows_x(uint4* __restrict__ s_in, const uint16_t* __restrict__ x, const uint4* __restrict__ wfrag,
                                            const float* __restrict__ bias, const uint16_t* __restrict__ res, const int (&rrow)[4],
                                            const int (&rbase)[4], const uint32_t (&rmask)[4], uint16_t* const (&yrow)[4], int b, int H, int W,
                                            int y0, int x0, uint32_t need, int mg0, int relu, int px, int kb, int lane, int hsel) {
  constexpr int NS = CIN / 64, NRA = NR > 0 ? NR : 1;
  const int ox = x0 + px;
  {  // ONE pass of 128 output channels per call (round 6: a work unit is (tile, pass), see k_conv3x3_ldsx); slab 0 of the tile is already staged
    const int h = hsel;
    const int mg = mg0 + 4 * h;
    v16f acc[NRA][2];
    if (NR > 0) {
#pragma unroll
      for (int m = 0; m < 2; m++) {
        const v16f bq = bias_tile(bias, (mg + m) * 32, kb);
#pragma unroll
        for (int j = 0; j < NR; j++) acc[j][m] = bq;
      }
#ifdef PNX_CONV_RES_EARLY
      if (HAS_RES) {
#pragma unroll
        for (int j = 0; j < NR; j++) {
          uint4 rq[2][2];
          load_residual_u(rq, res + (((int64_t)b * H + (y0 + rrow[j])) * W + x0) * COUT + mg * 32, (uint32_t)(px * COUT + 8 * kb) * 2u, (rmask[j] >> px) & 1u);
          add_residual(acc[j], rq);
        }
      }
#endif
    }
    // residual of this pass: row 0's lines are requested before the LAST input slab's taps and arrive under them, row j + 1's before row j is packed
    // (see conv_rows); nothing is held across a barrier-bracketed slab change
    const uint16_t* rres[NRA];  // wave-uniform: pixel 0 of the row segment, first channel of this wave's 64-channel group
    const uint32_t res_off = (uint32_t)(px * COUT + 8 * kb) * 2u;
    uint4 rq[2][2];
    if (NR > 0 && HAS_RES) {
#pragma unroll
      for (int j = 0; j < NR; j++) rres[j] = res + (((int64_t)b * H + (y0 + rrow[j])) * W + x0) * COUT + mg * 32;
    }
#pragma unroll 1
    for (int sl = 0; sl < NS; sl++) {
      if (sl) {
        __syncthreads();  // previous slab consumed
        stage_tile64<CIN, L128_TH>(s_in, x, b, H, W, 64 * sl, y0, x0, need);
        __syncthreads();
      }
#ifndef PNX_CONV_RES_EARLY
      if (NR > 0 && HAS_RES && sl == NS - 1) load_residual_u(rq, rres[0], res_off, rmask[0] >> px & 1u);
#endif
#ifndef LDSX_RES_WD
#define LDSX_RES_WD 2
#endif
      // with a residual the row-0 lines are in flight under the last slab's taps (16 registers): one weight fragment pair less in the ring keeps the kernel at
      // 256 registers without spilling the tile loop's state (profiles/r06_conv_pc_ab.txt (12))
      if (NR > 0) conv_taps<NRA, COUT / 32, CIN / 16, 1, 2, HAS_RES ? LDSX_RES_WD : TAPS_WD>(acc, s_in, wfrag, rbase, mg, px, kb, lane, 4 * sl);
    }
    if (NR > 0) {
      const int n_valid = W - x0;
#pragma unroll
      for (int j = 0; j < NR; j++) {
        const bool act = (rmask[j] >> px) & 1u;
        uint4 rc[2][2];
#ifndef PNX_CONV_RES_EARLY
        if (HAS_RES) {
#pragma unroll
          for (int m = 0; m < 2; m++)
#pragma unroll
            for (int t = 0; t < 2; t++) rc[m][t] = rq[m][t];
          if (j + 1 < NR) load_residual_u(rq, rres[j + 1], res_off, rmask[j + 1] >> px & 1u);
        }
#endif
        uint4 D[4];
#pragma unroll
        for (int m = 0; m < 2; m++) {
          uint4 pk[2];
#ifndef PNX_CONV_RES_EARLY
          pack_tile(acc[j][m], act, relu, pk, HAS_RES ? rc[m] : nullptr);
#else
          pack_tile(acc[j][m], act, relu, pk);
#endif
          D[2 * m] = pk[0], D[2 * m + 1] = pk[1];
        }
        transpose_row64(D, lane);
        store_row64<COUT>(D, yrow[j] + mg * 32, n_valid, lane);
      }
    }
  }
}

template <int CIN, int COUT, bool HAS_RES>
__global__ __launch_bounds__(256, 2) void k_conv3x3_ldsx(const uint16_t* __restrict__ x, const uint4* __restrict__ wfrag,
                                                      const float* __restrict__ bias, const uint16_t* __restrict__ res,
                                                      const uint8_t* __restrict__ mask, uint16_t* __restrict__ y, int B, int H, int W,
                                                      int relu, uint8_t* __restrict__ row_dirty, int slot, const int32_t* __restrict__ tlist,
                                                      const int32_t* __restrict__ tcount) {
  static_assert(CIN % 64 == 0 && (COUT % 128 == 0 || COUT == 64), "64-channel input slabs, 128-channel output passes (or one of 64)");
  constexpr int TH = L128_TH, HW_ = LDS_HW;
  constexpr int NRG = COUT == 64 ? 4 : 2, NRMAX = TH / NRG;  // row groups (the other waves split the 128 output channels of a pass)
  __shared__ uint4 s_in[L128_NSTAGE];
  __shared__ uint32_t s_rowmask2[2 * TH];  // double-buffered by iteration parity (an empty tile has a single barrier)
  __shared__ unsigned int s_next[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int px = lane & 31, kb = lane >> 5;
  const int rg = wv % NRG, mg0 = 2 * (wv / NRG);  // row group, first 32-channel output tile of this wave within a pass
  const int tiles_x = (W + 31) >> 5, tiles_y = (H + TH - 1) / TH;
  const int n_tiles = tlist != nullptr ? __builtin_amdgcn_readfirstlane(*tcount) : B * tiles_y * tiles_x;  // < 2^31 / NH: 32-bit unit indices keep the tile loop's state in fewer registers
  // Work unit = (tile, pass of 128 output channels) (round 6): every pass re-stages the tile's input slabs anyway, so handing the COUT / 128 passes of a tile to
  // different workgroups costs nothing and doubles the number of units -- at 180 x 180 a 256 -> 256 layer has ~830 active 8 x 32 tiles for 512 workgroups.
  // A unit is coded tile * NH + pass; pass 0 also does the tile's zero-fill / row_dirty bookkeeping.
  constexpr int NH = (COUT + 127) / 128;
  const int n_units = n_tiles * NH;
  auto unit_at = [&](int64_t idx) -> int {
    if (idx >= n_units) return -1;
    const int i = (int)idx;
    const int t = tlist != nullptr ? tlist[i / NH] : i / NH;
    return t * NH + i % NH;
  };
  // requests the mask bytes / row_dirty flags of the rows this wave looks at in tile t (t < 0: none)
  auto load_mask = [&](int t, bool (&a)[2], int (&wz)[2]) {
#pragma unroll
    for (int j = 0; j < 2; j++) a[j] = false, wz[j] = 1;
    if (t < 0) return;
    const int tx = (int)(t % tiles_x);
    const int ty = (int)((t / tiles_x) % tiles_y);
    const int b = (int)(t / ((int64_t)tiles_x * tiles_y));
    const int ox = tx * 32 + px;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int oy = ty * TH + wv * 2 + j;
      a[j] = ox < W && oy < H && (mask == nullptr || mask[((int64_t)b * H + oy) * W + ox] != 0);
      if (row_dirty != nullptr && oy < H) wz[j] = (int)row_dirty[((int64_t)b * H + oy) * tiles_x + tx];
    }
  };
  int64_t idxB = (int64_t)blockIdx.x + gridDim.x, idxC = 0;
  int tileA = unit_at(blockIdx.x), tileB = unit_at(idxB), tileC = -1;
  bool aP[2], aN[2];
  int wasP[2], wasN[2];
  load_mask(tileA >= 0 ? tileA / NH : -1, aP, wasP);
#pragma unroll
  for (int j = 0; j < 2; j++) aN[j] = false, wasN[j] = 1;
  int it = 0;
  for (; tileA >= 0; tileA = tileB, tileB = tileC, idxB = idxC, it++) {
    const int unit = __builtin_amdgcn_readfirstlane(tileA);  // uniform by construction: everything derived from it (origins, row pointers) lives in scalar registers
    const int tile = unit / NH;
    const int hsel = unit - tile * NH;
    sched_draw(s_next, it, slot);
    uint32_t* const s_rowmask = s_rowmask2 + (it & 1) * TH;
    const int tx = (int)(tile % tiles_x);
    const int ty = (int)((tile / tiles_x) % tiles_y);
    const int b = (int)(tile / ((int64_t)tiles_x * tiles_y));
    const int x0 = tx * 32, y0 = ty * TH;
    const int ox = x0 + px;
    bool was[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {  // active sites: one 32-bit column mask per row of the tile, from the bytes requested one tile ago
      was[j] = __builtin_amdgcn_readfirstlane(wasP[j]) != 0;
      const uint32_t bal = (uint32_t)__ballot(aP[j]);
      if (lane == 0) s_rowmask[wv * 2 + j] = bal;
    }
    __syncthreads();  // row masks visible; everybody is done reading the previous tile's s_in
    idxC = sched_next2(s_next, it, slot, idxB);
    tileC = unit_at(idxC);
    load_mask(tileB >= 0 ? tileB / NH : -1, aN, wasN);  // consumed at the top of the next iteration
#pragma unroll
    for (int j = 0; j < 2; j++) aP[j] = aN[j], wasP[j] = wasN[j];
    const uint32_t my_rm = s_rowmask[lane & 7];
    const uint32_t am = (uint32_t)__ballot(my_rm != 0) & 0xffu;
#pragma unroll
    for (int j = 0; j < 2; j++) {  // rows without any active site: zero-fill where needed
      const int rr = wv * 2 + j, oy = y0 + rr;
      const bool active = (am >> rr) & 1u;
      if (hsel != 0) continue;  // the tile's bookkeeping belongs to its pass-0 unit
      if (!active && was[j] && oy < H && ox < W) {
        uint4* dst = reinterpret_cast<uint4*>(y + (((int64_t)b * H + oy) * W + ox) * COUT);
#pragma unroll 1
        for (int ch = kb; ch < COUT / 8; ch += 2) dst[ch] = make_uint4(0, 0, 0, 0);
      }
      if (row_dirty != nullptr && oy < H && lane == 0 && was[j] != active) row_dirty[((int64_t)b * H + oy) * tiles_x + tx] = active ? 1 : 0;
    }
    if (am == 0) continue;  // uniform over the workgroup
    int nr = 0;
    int rbase[4], rrow[4];
    {  // the active rows, dealt round-robin to the 2 row groups
      uint32_t rest = am;
      for (int k = 0; k < rg && rest; k++) rest &= rest - 1;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool has = j < NRMAX && rest != 0;
        rrow[j] = has ? __builtin_ctz(rest) : 0;  // dummy rows point at row 0
        nr += has ? 1 : 0;
        for (int k = 0; k < NRG && rest; k++) rest &= rest - 1;
      }
    }
    nr = __builtin_amdgcn_readfirstlane(nr);
    uint32_t rmask[4];
    uint16_t* yrow[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      rrow[j] = __builtin_amdgcn_readfirstlane(rrow[j]);
      rbase[j] = rrow[j] * HW_ * 8;
      rmask[j] = j < nr ? __builtin_amdgcn_readfirstlane(s_rowmask[rrow[j]]) : 0u;
      yrow[j] = y + (((int64_t)b * H + (y0 + rrow[j])) * W + x0) * COUT;
    }
    const uint32_t need = am | (am << 1) | (am << 2);  // halo rows some active row reads
    stage_tile64<CIN, TH>(s_in, x, b, H, W, 0, y0, x0, need);
    __syncthreads();
#define PNX_ROWS_X(N_) conv_rows_x<(N_ <= NRMAX ? N_ : NRMAX), CIN, COUT, HAS_RES>(s_in, x, wfrag, bias, res, rrow, rbase, rmask, yrow, b, H, W, y0, x0, need, mg0, relu, px, kb, lane, hsel)
    switch (nr) {  // wave-uniform; every case runs the same barriers
      case 0: PNX_ROWS_X(0); break;
      case 1: PNX_ROWS_X(1); break;
      case 2: PNX_ROWS_X(2); break;
      case 3: PNX_ROWS_X(3); break;
      default: PNX_ROWS_X(4); break;
    }
#undef PNX_ROWS_X
  }
  sched_done(slot);
}

template <int CIN, int COUT>
int launch_ldsx(const void* x, const void* wfrag, const float* bias, const void* res, const uint8_t* mask, void* y, int B, int H, int W, int relu,
                uint8_t* row_dirty, const int32_t* tlist, const int32_t* tcount, hipStream_t st) {
  const int slot = mask != nullptr ? next_sched_slot() : -1;
  int64_t nb = (int64_t)B * ((H + L128_TH - 1) / L128_TH) * ((W + 31) / 32) * ((COUT + 127) / 128);  // work units = (tile, pass)
  if (nb > 512) nb = 512;  // resident workgroups: 2 per CU (registers)
  if (res != nullptr)
    k_conv3x3_ldsx<CIN, COUT, true><<<(unsigned)nb, 256, 0, st>>>((const uint16_t*)x, (const uint4*)wfrag, bias, (const uint16_t*)res, mask, (uint16_t*)y,
                                                                 B, H, W, relu, row_dirty, slot, tlist, tcount);
  else
    k_conv3x3_ldsx<CIN, COUT, false><<<(unsigned)nb, 256, 0, st>>>((const uint16_t*)x, (const uint4*)wfrag, bias, nullptr, mask, (uint16_t*)y, B, H, W,
                                                                  relu, row_dirty, slot, tlist, tcount);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

// ---- stride 2 (the entry convolution of backbone stages 1-3: 64 -> 128, 128 -> 256, 256 -> 256; sparse_resnet.py:50-68, SparseConv2d).
// Output tile 4 rows x 32 pixels; its 9 x 65 input halo of one 64-channel slab is staged de-interleaved (even columns, then odd
// columns) so that the tap loop above reads consecutive slots.  74.9 KB of LDS: 2 workgroups per CU.
constexpr int S2_TH = 4;
constexpr int S2_ROWS = 2 * S2_TH + 1;
constexpr int S2_NSTAGE = S2_ROWS * S2_RS * 8;

// Halo rows iy0 + r (r = 0..8, `need` bit r), input columns ix0 + c (c = 0..64) of channels [ch0, ch0 + 64): slot c/2 of the even
// plane for odd c (= even input column 2 x0 + ...: ix0 = 2 x0 - 1 is odd), slot 32 + c/2 of the odd plane for even c.  Thread t
// owns 16-byte chunk t & 7 of plane slot t >> 3; as in stage_tile64 the rows go global -> LDS directly and the XOR swizzle is
// applied on the source side.  Slot 64 (c = 64) keeps the register path.
template <int CSTRIDE>
__device__ __forceinline__ void stage_tile64_s2(uint4* __restrict__ s_in, const uint16_t* __restrict__ x, int b, int H, int W, int ch0, int iy0,
                                                int ix0, uint32_t need) {
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // opaque: keeps the per-thread staging addresses out of the tile loop's live ranges
  const int slot = tid & 7, col = tid >> 3, wbase = (tid >> 6) * 8;
  const uint16_t* xb = x + (int64_t)b * H * W * CSTRIDE + ch0;
#pragma unroll
  for (int pl = 0; pl < 2; pl++) {        // pl 0: even plane (c = 2 col + 1), pl 1: odd plane (c = 2 col)
    const int ls = pl * 32 + col;         // LDS slot of this thread's column
    const int ix = ix0 + 2 * col + (1 - pl);
    const bool col_ok = (unsigned)ix < (unsigned)W;
    const int chunk = slot ^ lds_swz(ls);
#pragma unroll
    for (int r = 0; r < S2_ROWS; r++) {
      if (!((need >> r) & 1u)) continue;  // wave-uniform
      const int iy = iy0 + r;
      if ((unsigned)iy < (unsigned)H && col_ok) {
        __builtin_amdgcn_global_load_lds((gptr_t)(xb + ((int64_t)iy * W + ix) * CSTRIDE + chunk * 8),
                                         (lptr_t)(s_in + (r * S2_RS + pl * 32 + wbase) * 8), 16, 0, 0);
      } else {
        s_in[(r * S2_RS + ls) * 8 + slot] = make_uint4(0, 0, 0, 0);
      }
    }
  }
  if (tid < S2_ROWS * 8) {  // slot 64: input column ix0 + 64
    const int re = tid >> 3, ce = tid & 7;
    if ((need >> re) & 1u) {
      const int iy = iy0 + re, ix = ix0 + 64;
      uint4 qe = make_uint4(0, 0, 0, 0);
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) qe = *reinterpret_cast<const uint4*>(xb + ((int64_t)iy * W + ix) * CSTRIDE + ce * 8);
      s_in[(re * S2_RS + 64) * 8 + (ce ^ lds_swz(64))] = qe;
    }
  }
}

template <int NR, int CIN, int COUT, int MB, bool X3 = false>
__device__ __forceinline__ void conv_rows_s2(uint4* __restrict__ s_in, const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2,
                                             const uint4* __restrict__ wfrag, const uint4* __restrict__ wfrag2, const float* __restrict__ bias, const int (&rbase)[4], const uint32_t (&rmask)[4],
                                             uint16_t* const (&yrow)[4], int b, int H, int W, int iy0, int ix0, int n_valid, uint32_t need, int mg,
                                             int relu, int px, int kb, int lane) {
  constexpr int NS = CIN / 64, NRA = NR > 0 ? NR : 1;
  v16f acc[NRA][MB];
  if (NR > 0) {
#pragma unroll
    for (int m = 0; m < MB; m++) {
      v16f bq;
      if (X3 && bias == nullptr) {
#pragma unroll
        for (int i = 0; i < 16; i++) bq[i] = 0.f;
      } else {
        bq = bias_tile(bias, (mg + m) * 32, kb);
      }
#pragma unroll
      for (int j = 0; j < NR; j++) acc[j][m] = bq;
    }
  }
  // X3 (fp32 product of bf16 pairs, see conv_pc.h): the NS slabs of the high halves (taps with W_hi and with W_lo), then those of the low halves (W_hi)
#pragma unroll 1
  for (int sq = 0; sq < NS * (X3 ? 2 : 1); sq++) {
    const int sl = X3 ? sq % NS : sq;
    if (sq) {
      __syncthreads();  // previous slab consumed
      stage_tile64_s2<CIN>(s_in, X3 && sq >= NS ? x2 : x, b, H, W, 64 * sl, iy0, ix0, need);
      __syncthreads();
    }
    if (NR > 0) {
      conv_taps<NRA, COUT / 32, CIN / 16, 2, MB>(acc, s_in, wfrag, rbase, mg, px, kb, lane, 4 * sl);
      if constexpr (X3) {
        if (sq < NS) conv_taps<NRA, COUT / 32, CIN / 16, 2, MB>(acc, s_in, wfrag2, rbase, mg, px, kb, lane, 4 * sl);
      }
    }
  }
  if (NR > 0) {
#pragma unroll
    for (int j = 0; j < NR; j++) {
      const bool act = (rmask[j] >> px) & 1u;
      if constexpr (X3) {
#pragma unroll
        for (int m = 0; m < MB; m++) store_tile_f32<COUT>(acc[j][m], act, reinterpret_cast<float*>(yrow[j]) + (mg + m) * 32, n_valid, px, kb);
      } else if constexpr (MB == 2) {
        uint4 D[4];
#pragma unroll
        for (int m = 0; m < 2; m++) {
          uint4 pk[2];
          pack_tile(acc[j][m], act, relu, pk);
          D[2 * m] = pk[0], D[2 * m + 1] = pk[1];
        }
        transpose_row64(D, lane);
        store_row64<COUT>(D, yrow[j] + mg * 32, n_valid, lane);
      } else {  // one 32-channel tile per wave: complete 64-byte half lines
        uint4 pk[2];
        pack_tile(acc[j][0], act, relu, pk);
        transpose_row32(pk, lane);
        store_row32<COUT>(pk, yrow[j] + mg * 32, n_valid, lane);
      }
    }
  }
}

// H, W: input; Ho, Wo: output.  The 4 waves are COUT/64 groups of 64 output channels x 4/(COUT/64) row groups.
template <int CIN, int COUT, bool X3 = false>
__global__ __launch_bounds__(256, 2) void k_conv3x3_s2(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2, const uint4* __restrict__ wfrag,
                                                    const uint4* __restrict__ wfrag2, const float* __restrict__ bias,
                                                    const uint8_t* __restrict__ mask, uint16_t* __restrict__ y, int B, int H, int W, int Ho, int Wo,
                                                    int relu, uint8_t* __restrict__ row_dirty, int slot) {
  static_assert(CIN % 64 == 0 && (COUT == 128 || COUT == 256), "64-channel input slabs; 4 groups of 32 or of 64 output channels");
  // Round 6: with 128 output channels the waves were 2 row groups x 2 groups of 64 channels, i.e. at most TWO rows per wave -- one 1 KiB weight fragment from
  // L1 per two MFMAs, which is all of the L1's bandwidth at full MFMA rate (the 759 us outlier of profiles/r06_bench_steady_trace.md).  Now every shape has one
  // row group: four rows per wave, four channel groups of COUT / 4.
  constexpr int TH = S2_TH, MB = COUT / 128, NCG = 4, NRG = 1, NRMAX = TH / NRG;
  constexpr int YS = X3 ? 2 : 1;  // output element in units of uint16_t (X3: fp32)
  extern __shared__ uint4 s_in[];  // S2_NSTAGE
  __shared__ uint32_t s_rowmask2[2 * TH];
  __shared__ unsigned int s_next[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int px = lane & 31, kb = lane >> 5;
  const int rg = wv % NRG, mg = MB * (wv / NRG);
  static_assert(NCG * NRG == 4, "four waves");
  const int tiles_x = (Wo + 31) >> 5, tiles_y = (Ho + TH - 1) / TH;
  const int64_t n_tiles = (int64_t)B * tiles_y * tiles_x;
  int64_t next = 0;
  int it = 0;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile = next, it++) {
    sched_draw(s_next, it, slot);
    uint32_t* const s_rowmask = s_rowmask2 + (it & 1) * TH;
    const int tx = (int)(tile % tiles_x);
    const int ty = (int)((tile / tiles_x) % tiles_y);
    const int b = (int)(tile / ((int64_t)tiles_x * tiles_y));
    const int x0 = tx * 32, y0 = ty * TH;
    const int ox = x0 + px;
    bool was = true;
    {  // active sites: wave wv looks at output row wv of the tile
      const int oy = y0 + wv;
      const bool a = ox < Wo && oy < Ho && (mask == nullptr || mask[((int64_t)b * Ho + oy) * Wo + ox] != 0);
      if (row_dirty != nullptr && oy < Ho) was = __builtin_amdgcn_readfirstlane((int)row_dirty[((int64_t)b * Ho + oy) * tiles_x + tx]) != 0;
      const uint32_t bal = (uint32_t)__ballot(a);
      if (lane == 0) s_rowmask[wv] = bal;
    }
    __syncthreads();  // row masks visible; everybody is done reading the previous tile's s_in
    next = sched_next(s_next, it, slot, tile);
    const uint32_t my_rm = s_rowmask[lane & 3];
    const uint32_t am = (uint32_t)__ballot(my_rm != 0) & 0xfu;
    {  // a row without any active site: zero-fill where needed
      const int oy = y0 + wv;
      const bool active = (am >> wv) & 1u;
      if (!active && was && oy < Ho && ox < Wo) {
        uint4* dst = reinterpret_cast<uint4*>(y + (((int64_t)b * Ho + oy) * Wo + ox) * COUT * YS);
#pragma unroll 1
        for (int ch = kb; ch < COUT * YS / 8; ch += 2) dst[ch] = make_uint4(0, 0, 0, 0);
      }
      if (row_dirty != nullptr && oy < Ho && lane == 0 && was != active) row_dirty[((int64_t)b * Ho + oy) * tiles_x + tx] = active ? 1 : 0;
    }
    if (am == 0) continue;  // uniform over the workgroup
    int nr = 0;
    int rbase[4], rrow[4];
    {  // the active rows, dealt round-robin to the row groups
      uint32_t rest = am;
      for (int k = 0; k < rg && rest; k++) rest &= rest - 1;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool has = j < NRMAX && rest != 0;
        rrow[j] = has ? __builtin_ctz(rest) : 0;  // dummy rows point at row 0
        nr += has ? 1 : 0;
        for (int k = 0; k < NRG && rest; k++) rest &= rest - 1;
      }
    }
    nr = __builtin_amdgcn_readfirstlane(nr);
    uint32_t rmask[4];
    uint16_t* yrow[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      rrow[j] = __builtin_amdgcn_readfirstlane(rrow[j]);
      rbase[j] = 2 * rrow[j] * S2_RS * 8;
      rmask[j] = j < nr ? __builtin_amdgcn_readfirstlane(s_rowmask[rrow[j]]) : 0u;
      yrow[j] = y + (((int64_t)b * Ho + (y0 + rrow[j])) * Wo + x0) * COUT * YS;
    }
    uint32_t need = 0;  // halo rows some active output row reads: rows 2r, 2r+1, 2r+2
#pragma unroll
    for (int r = 0; r < TH; r++)
      if ((am >> r) & 1u) need |= 7u << (2 * r);
    const int iy0 = 2 * y0 - 1, ix0 = 2 * x0 - 1;
    stage_tile64_s2<CIN>(s_in, x, b, H, W, 0, iy0, ix0, need);
    __syncthreads();
#define PNX_ROWS_S2(N_) conv_rows_s2<(N_ <= NRMAX ? N_ : NRMAX), CIN, COUT, MB, X3>(s_in, x, x2, wfrag, wfrag2, bias, rbase, rmask, yrow, b, H, W, iy0, ix0, Wo - x0, need, mg, relu, px, kb, lane)
    switch (nr) {  // wave-uniform; every case runs the same barriers
      case 0: PNX_ROWS_S2(0); break;
      case 1: PNX_ROWS_S2(1); break;
      case 2: PNX_ROWS_S2(2); break;
      case 3: PNX_ROWS_S2(3); break;
      default: PNX_ROWS_S2(4); break;
    }
#undef PNX_ROWS_S2
  }
  sched_done(slot);
}

template <int CIN, int COUT, bool X3 = false>
int launch_s2(const void* x, const void* wfrag, const float* bias, const uint8_t* mask, void* y, int B, int H, int W, int Ho, int Wo, int relu,
              uint8_t* row_dirty, hipStream_t st, const void* x2 = nullptr, const void* wfrag2 = nullptr) {
  const int slot = mask != nullptr ? next_sched_slot() : -1;
  int64_t nb = (int64_t)B * ((Ho + S2_TH - 1) / S2_TH) * ((Wo + 31) / 32);
  if (nb > 512) nb = 512;  // resident workgroups: 2 per CU (LDS and registers)
  auto kern = k_conv3x3_s2<CIN, COUT, X3>;
  constexpr int lds = S2_NSTAGE * 16;
  static bool attr_done = false;
  if (!attr_done) {
    PNX_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  kern<<<(unsigned)nb, 256, lds, st>>>((const uint16_t*)x, (const uint16_t*)x2, (const uint4*)wfrag, (const uint4*)wfrag2, bias, mask, (uint16_t*)y, B, H, W, Ho,
                                       Wo, relu, row_dirty, slot);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

// ---- ConvTranspose2d(64, 64, kernel 2, stride 2) + folded BN + ReLU: the deblock of every SepHead (det3d/models/heads/centerhead.py:17-21).
// Kernel == stride, so the four output parities (ky, kx) are four independent 1x1 convolutions of the input pixel: per 32 input
// pixels 4 x (2 output-channel tiles x 4 k-steps) MFMAs and four 128-byte output lines per pixel, written as complete lines (the
// epilogue above with a pixel stride of two).  Weights (32 KB, fragment order) stay in registers of the persistent waves.
__global__ __launch_bounds__(256, 2) void k_deconv2x2_64(const uint16_t* __restrict__ x, const uint4* __restrict__ wfrag, const float* __restrict__ bias,
                                                      uint16_t* __restrict__ y, int B, int H, int W, int relu) {
  constexpr int C = 64;
  const int lane = threadIdx.x & 63, px = lane & 31, kb = lane >> 5;
  uint4 w[4][4][2];  // [parity][k-step][output-channel tile]
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
      for (int m = 0; m < 2; m++) w[p][ks][m] = wfrag[((p * 4 + ks) * 2 + m) * 64 + lane];
  v16f bq[2];
#pragma unroll
  for (int m = 0; m < 2; m++) bq[m] = bias_tile(bias, m * 32, kb);
  const int segs = (W + 31) >> 5;
  const int64_t n_seg = (int64_t)B * H * segs;
  const int Wo = 2 * W;
  for (int64_t sg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); sg < n_seg; sg += (int64_t)gridDim.x * 4) {
    const int sx = (int)(sg % segs);
    const int64_t row = sg / segs;  // b * H + y
    const int x0 = sx * 32, n_valid = W - x0;
    const bool in = x0 + px < W;
    const uint16_t* src = x + (row * W + (in ? x0 + px : x0)) * C + 8 * kb;
    uint4 q[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) q[ks] = in ? *reinterpret_cast<const uint4*>(src + ks * 16) : make_uint4(0, 0, 0, 0);
    const int b = (int)(row / H), yy = (int)(row - (int64_t)b * H);
#pragma unroll
    for (int p = 0; p < 4; p++) {
      v16f acc[2] = {bq[0], bq[1]};
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const el8 bfr = __builtin_bit_cast(el8, q[ks]);
#pragma unroll
        for (int m = 0; m < 2; m++) acc[m] = PNX_MFMA32(__builtin_bit_cast(el8, w[p][ks][m]), bfr, acc[m]);
      }
      uint4 D[4];
#pragma unroll
      for (int m = 0; m < 2; m++) {
        uint4 pk[2];
        pack_tile(acc[m], true, relu, pk);
        D[2 * m] = pk[0], D[2 * m + 1] = pk[1];
      }
      transpose_row64(D, lane);
      uint16_t* orow = y + ((((int64_t)b * 2 * H + 2 * yy + (p >> 1)) * Wo) + 2 * x0 + (p & 1)) * C;
      store_row64<2 * C>(D, orow, n_valid, lane);  // output pixel of input pixel P: 2 P + kx -> a pixel stride of 2 x 64 channels
    }
  }
}

// ---- final convolution of the merged SepHead branches (det3d/models/heads/centerhead.py:12-59: Conv2d(64, k_j, 3) of every
// branch j of a task).  The first (merged) convolution leaves NBR x 64 channels per pixel; branch j reads only its own 64, so
// the stacked weight (sum k_j <= 16 outputs x NBR*64 inputs) is block diagonal.  HBM-bound by its input (768 B per pixel at
// NBR = 6 against 32 B of output): the 64-channel slabs are staged through LDS one after the other (each pixel is read from
// HBM/L2 once instead of 9 times) and accumulated on v_mfma_f32_16x16x32_bf16 with M = the 16 output channels, N = 16 pixels;
// the D fragment (lane = pixel n + 16 q: channels 4q..4q+3) stores 8 bytes per lane, 512 contiguous bytes per instruction.
// wfrag: [branch][tap][kc][lane = q*16 + o][8]: W[o][branch*64 + kc*32 + q*8 + e][ky][kx] (ops.py::sephead_pack_weights).
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NBR, int TH>
__global__ __launch_bounds__(256) void k_sephead_out(const uint16_t* __restrict__ x, const uint4* __restrict__ wfrag, const float* __restrict__ bias,
                                                     uint16_t* __restrict__ y, int B, int H, int W) {
  constexpr int CIN = NBR * 64, HW_ = LDS_HW, RW = TH / 4;  // RW rows per wave
  __shared__ uint4 s_in[(TH + 2) * LDS_HW * 8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const float4 bq = *reinterpret_cast<const float4*>(bias + 4 * q);
  const int tiles_x = (W + 31) >> 5, tiles_y = (H + TH - 1) / TH;
  const int64_t n_tiles = (int64_t)B * tiles_y * tiles_x;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int tx = (int)(tile % tiles_x);
    const int ty = (int)((tile / tiles_x) % tiles_y);
    const int b = (int)(tile / ((int64_t)tiles_x * tiles_y));
    const int x0 = tx * 32, y0 = ty * TH;
    v4f acc[RW][2];
#pragma unroll
    for (int r = 0; r < RW; r++)
#pragma unroll
      for (int h = 0; h < 2; h++) acc[r][h] = v4f{bq.x, bq.y, bq.z, bq.w};
#pragma unroll 1
    for (int j = 0; j < NBR; j++) {
      __syncthreads();  // everybody is done reading the previous slab
      stage_tile64<CIN, TH>(s_in, x, b, H, W, j * 64, y0, x0, (1u << (TH + 2)) - 1u);
      __syncthreads();
      const uint4* wj = wfrag + (size_t)j * 18 * 64 + lane;
      uint4 wn[2] = {wj[0], wj[64]};
#pragma unroll 1
      for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3, dx = tap - 3 * dy;  // halo coordinates: +1 already included
        const uint4 wc[2] = {wn[0], wn[1]};
        if (tap < 8) wn[0] = wj[(tap + 1) * 128], wn[1] = wj[(tap + 1) * 128 + 64];
#pragma unroll
        for (int kc = 0; kc < 2; kc++) {
          uint4 bf[RW][2];
#pragma unroll
          for (int r = 0; r < RW; r++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const int c = 16 * h + n + dx;
              bf[r][h] = s_in[((wv * RW + r + dy) * HW_ + c) * 8 + ((kc * 4 + q) ^ lds_swz(c))];
            }
#pragma unroll
          for (int r = 0; r < RW; r++)
#pragma unroll
            for (int h = 0; h < 2; h++)
              acc[r][h] = PNX_MFMA16(__builtin_bit_cast(el8, wc[kc]), __builtin_bit_cast(el8, bf[r][h]), acc[r][h]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RW; r++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int oy = y0 + wv * RW + r, ox = x0 + 16 * h + n;
        if (oy < H && ox < W) {
          uint2 p;
          p.x = pack_el(acc[r][h][0], acc[r][h][1]);
          p.y = pack_el(acc[r][h][2], acc[r][h][3]);
          *reinterpret_cast<uint2*>(y + (((int64_t)b * H + oy) * W + ox) * 16 + 4 * q) = p;
        }
      }
  }
}

template <int NBR>
int launch_sephead(const void* x, const void* wfrag, const float* bias, void* y, int B, int H, int W, hipStream_t st) {
  constexpr int TH = 8;  // 8-row tiles: 43.5 KiB of LDS, 3 workgroups per CU (4 and 16 rows measured slower in round 2)
  int64_t nb = (int64_t)B * ((H + TH - 1) / TH) * ((W + 31) / 32);
  if (nb > 768) nb = 768;  // resident workgroups
  k_sephead_out<NBR, TH><<<(unsigned)nb, 256, 0, st>>>((const uint16_t*)x, (const uint4*)wfrag, bias, (uint16_t*)y, B, H, W);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

// ---- tile list: the 32-pixel-wide, `th`-row tiles that hold an active site now or a stale row in one of the persistent output
// buffers (row_dirty).  A 1440 x 1440 sweep leaves 65 % of the 16 x 32 tiles empty, and an empty tile costs the convolution
// kernels a mask read + a barrier (~1.5 us) each.  One THREAD per tile (consecutive lanes read consecutive 32-byte pieces of a mask
// row), the workgroup compacts its hits in LDS and draws ONE range of the list: a wave per tile with an atomic per hit ran 73 us
// on the 32 400 tiles of stage 0 (11 000 same-address atomics at ~13 ns), this form ~10 us.  The list order follows the atomics.
struct DirtySet {
  const uint8_t* p[4];
};
// Round 6: ONE LANE PER ROW of a tile (th <= 16 lanes per tile, 4 tiles per wave) instead of one thread per tile: a thread walked its 16 rows one dependent
// load after the other (38-47 us per stage for 25 MB of mask bytes); now every lane issues its two 16-byte mask loads and its <= 4 flag bytes at once and the
// tile's verdict is an OR across its 16 lanes.
__global__ __launch_bounds__(256) void k_tile_list(const uint8_t* __restrict__ mask, DirtySet ds, int B, int H, int W, int th, int32_t* __restrict__ list,
                                                   int32_t* __restrict__ count) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int tiles_x = (W + 31) >> 5, tiles_y = (H + th - 1) / th;
  const int64_t n_tiles = (int64_t)B * tiles_y * tiles_x;
  const int r = lane & 15;                                       // row of the tile this lane looks at (th <= 16)
  const int64_t tile = ((int64_t)blockIdx.x * 4 + wv) * 4 + (lane >> 4);  // 16 tiles per workgroup
  uint32_t acc = 0;
  if (tile < n_tiles && r < th) {
    const int tx = (int)(tile % tiles_x);
    const int ty = (int)((tile / tiles_x) % tiles_y);
    const int b = (int)(tile / ((int64_t)tiles_x * tiles_y));
    const int x0 = tx * 32, oy = ty * th + r;
    if (oy < H) {
      const uint8_t* row = mask + ((int64_t)b * H + oy) * W + x0;
      if ((W & 15) == 0 && x0 + 32 <= W) {  // two aligned 16-byte loads per row
        const uint4 a = reinterpret_cast<const uint4*>(row)[0], c = reinterpret_cast<const uint4*>(row)[1];
        acc |= a.x | a.y | a.z | a.w | c.x | c.y | c.z | c.w;
      } else {
        for (int k = 0; k < 32 && x0 + k < W; k++) acc |= row[k];
      }
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (ds.p[k] != nullptr) acc |= ds.p[k][((int64_t)b * H + oy) * tiles_x + tx];
    }
  }
  // OR over the 16 lanes of the tile (xor butterfly inside aligned groups of 16)
  acc |= (uint32_t)__shfl_xor((int)acc, 1);
  acc |= (uint32_t)__shfl_xor((int)acc, 2);
  acc |= (uint32_t)__shfl_xor((int)acc, 4);
  acc |= (uint32_t)__shfl_xor((int)acc, 8);
  const bool any = acc != 0 && r == 0 && tile < n_tiles;         // one lane per listed tile
  const unsigned long long bal = __ballot(any);
  if (lane == 0) s_wave[wv] = __popcll(bal);
  __syncthreads();
  if (t == 0) {
    const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = tot > 0 ? atomicAdd(count, tot) : 0;
  }
  __syncthreads();
  if (any) {
    int off = s_base + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wv; w++) off += s_wave[w];
    list[off] = (int32_t)tile;
  }
}

template <int COUT>
int launch_lds(const void* x, const void* wfrag, const float* bias, const void* res, const uint8_t* mask, void* y, int B, int H, int W, int relu,
               uint8_t* row_dirty, const int32_t* tlist, const int32_t* tcount, hipStream_t st) {
  constexpr int TH = LDS_TH;
  const int64_t n_tiles = (int64_t)B * ((H + TH - 1) / TH) * ((W + 31) / 32);
  int64_t nb = n_tiles;
  const int64_t cap = 256 * 2;  // resident workgroups (LDS: 76.5 KiB per workgroup)
  if (nb > cap) nb = cap;
  const int slot = mask != nullptr ? next_sched_slot() : -1;
  if constexpr (COUT == 64) {
    if (res != nullptr) {
      k_conv3x3_lds<COUT, true><<<(unsigned)nb, 256, 0, st>>>((const uint16_t*)x, (const uint4*)wfrag, bias, (const uint16_t*)res, mask, (uint16_t*)y, B,
                                                             H, W, relu, row_dirty, slot, tlist, tcount);
      PNX_LAUNCH_CHECK();
      return PNX_OK;
    }
  } else {
    PNX_REQUIRE(res == nullptr, PNX_ERR_UNSUPPORTED, "residual with %d output channels", COUT);
  }
  k_conv3x3_lds<COUT, false><<<(unsigned)nb, 256, 0, st>>>((const uint16_t*)x, (const uint4*)wfrag, bias, nullptr, mask, (uint16_t*)y, B, H, W, relu,
                                                          row_dirty, slot, tlist, tcount);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

#include "conv_pc.h"

template <int CIN, int COUT, int STRIDE>
int launch(const void* x, const void* wfrag, const float* bias, const void* res, const uint8_t* mask, void* y, int B, int H, int W, int Ho,
           int Wo, int relu, uint8_t* row_dirty, hipStream_t st) {
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
                                                    Ho, Wo, relu, row_dirty);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // namespace

extern "C" {

#if defined(PNX_CONV_TIMERS) && !defined(PNX_CONV_F16)
int pnx_debug_conv_timers(unsigned long long* out) {  // sums over waves of s_memtime ticks per section; resets the counters
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  PNX_CHECK_HIP(hipDeviceSynchronize());
  PNX_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_T), sizeof(z)));
  PNX_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_T), z, sizeof(z)));
  return PNX_OK;
}
#endif

int PNX_CONV_FN(pnx_sephead_out)(const void* x, const void* wfrag, const float* bias, void* y, int32_t batch, int32_t h, int32_t w, int32_t n_branch,
                         pnx_stream_t stream) {
  PNX_REQUIRE(x && wfrag && bias && y && batch > 0 && h > 0 && w > 0, PNX_ERR_INVALID, "bad arguments");
  PNX_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)wfrag | (uintptr_t)bias) & 15) == 0, PNX_ERR_INVALID, "16-byte alignment required");
  hipStream_t st = (hipStream_t)stream;
  switch (n_branch) {
    case 1: return launch_sephead<1>(x, wfrag, bias, y, batch, h, w, st);  // lazy head: the class branch alone ...
    case 2: return launch_sephead<2>(x, wfrag, bias, y, batch, h, w, st);  // ... or iou + class
    case 5: return launch_sephead<5>(x, wfrag, bias, y, batch, h, w, st);
    case 6: return launch_sephead<6>(x, wfrag, bias, y, batch, h, w, st);
    case 7: return launch_sephead<7>(x, wfrag, bias, y, batch, h, w, st);
    default: break;
  }
  pnx_set_error("pnx_sephead_out: no kernel for %d branches", n_branch);
  return PNX_ERR_UNSUPPORTED;
}

int PNX_CONV_FN(pnx_deconv2x2)(const void* x, const void* wfrag, const float* bias, void* y, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                       int32_t relu, pnx_stream_t stream) {
  PNX_REQUIRE(x && wfrag && bias && y && batch > 0 && h > 0 && w > 0, PNX_ERR_INVALID, "bad arguments");
  PNX_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)wfrag | (uintptr_t)bias) & 15) == 0, PNX_ERR_INVALID, "16-byte alignment required");
  if (cin != 64 || cout != 64) {
    pnx_set_error("pnx_deconv2x2: no kernel for %d -> %d channels", cin, cout);
    return PNX_ERR_UNSUPPORTED;
  }
  const int64_t n_seg = (int64_t)batch * h * ((w + 31) / 32);
  int64_t nb = (n_seg + 3) / 4;
  if (nb > 512) nb = 512;  // persistent: the weights are loaded once per wave
  k_deconv2x2_64<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>((const uint16_t*)x, (const uint4*)wfrag, bias, (uint16_t*)y, batch, h, w, relu);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

#ifndef PNX_CONV_F16  // type-independent helpers: in the bf16 object only
int pnx_conv3x3_tile_rows(int32_t cin, int32_t cout, int32_t stride) {
  if (stride != 1) return 0;
  if (cin == 64 && (cout == 64 || cout == 320 || cout == 384 || cout == 448)) return LDS_TH;
  if ((cin == 128 && cout == 128) || (cin == 256 && (cout == 256 || cout == 64))) return L128_TH;
  return 0;
}

static int conv_tile_list_impl(const uint8_t* mask, const uint8_t* const* row_dirty, int32_t n_dirty, int32_t batch, int32_t h, int32_t w, int32_t tile_rows,
                               int32_t* tile_list, int32_t* tile_count, pnx_stream_t stream, bool zero_count) {
  PNX_REQUIRE(mask && tile_list && tile_count && batch > 0 && h > 0 && w > 0 && tile_rows > 0, PNX_ERR_INVALID, "bad arguments");
  PNX_REQUIRE(n_dirty >= 0 && n_dirty <= 4 && (n_dirty == 0 || row_dirty != nullptr), PNX_ERR_INVALID, "0..4 row_dirty arrays");
  DirtySet ds;
  for (int k = 0; k < 4; k++) ds.p[k] = k < n_dirty ? row_dirty[k] : nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (zero_count) PNX_CHECK_HIP(hipMemsetAsync(tile_count, 0, sizeof(int32_t), st));
  const int64_t n_tiles = (int64_t)batch * ((h + tile_rows - 1) / tile_rows) * ((w + 31) / 32);
  PNX_REQUIRE(tile_rows <= 16, PNX_ERR_UNSUPPORTED, "tile_rows %d (at most 16: one lane per row)", tile_rows);
  k_tile_list<<<(unsigned)((n_tiles + 15) / 16), 256, 0, st>>>(mask, ds, batch, h, w, tile_rows, tile_list, tile_count);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}
int pnx_conv_tile_list(const uint8_t* mask, const uint8_t* const* row_dirty, int32_t n_dirty, int32_t batch, int32_t h, int32_t w, int32_t tile_rows,
                       int32_t* tile_list, int32_t* tile_count, pnx_stream_t stream) {
  return conv_tile_list_impl(mask, row_dirty, n_dirty, batch, h, w, tile_rows, tile_list, tile_count, stream, true);
}
// internal (enqueue.hip): *tile_count is already zero -- pnx_enqueue clears the counters of all tile-list entries of a table in one launch
int pnx_conv_tile_list_prezeroed(const uint8_t* mask, const uint8_t* const* row_dirty, int32_t n_dirty, int32_t batch, int32_t h, int32_t w, int32_t tile_rows,
                                 int32_t* tile_list, int32_t* tile_count, pnx_stream_t stream) {
  return conv_tile_list_impl(mask, row_dirty, n_dirty, batch, h, w, tile_rows, tile_list, tile_count, stream, false);
}

#endif

int PNX_CONV_FN(pnx_conv3x3)(const void* x, const void* wfrag, const float* bias, const void* residual, const uint8_t* mask, void* y, int32_t batch,
                     int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride, int32_t relu, uint8_t* row_dirty, const int32_t* tile_list,
                     const int32_t* tile_count, pnx_stream_t stream) {
  PNX_REQUIRE((tile_list == nullptr) == (tile_count == nullptr), PNX_ERR_INVALID, "tile_list and tile_count come together");
  PNX_REQUIRE(tile_list == nullptr || (mask != nullptr && pnx_conv3x3_tile_rows(cin, cout, stride) > 0), PNX_ERR_INVALID,
              "a tile list needs a mask and a kernel that takes one (pnx_conv3x3_tile_rows)");
  PNX_REQUIRE(x && wfrag && bias && y && batch > 0 && h > 0 && w > 0, PNX_ERR_INVALID, "bad arguments");
  PNX_REQUIRE(stride == 1 || stride == 2, PNX_ERR_UNSUPPORTED, "stride %d", stride);
  PNX_REQUIRE(row_dirty == nullptr || mask != nullptr, PNX_ERR_INVALID, "row_dirty needs an active-site mask");
  PNX_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)wfrag | (uintptr_t)bias | (uintptr_t)residual) & 15) == 0, PNX_ERR_INVALID,
              "16-byte alignment required");
  const int ho = (h + 2 - 3) / stride + 1, wo = (w + 2 - 3) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1 && getenv("PNX_CONV_DIRECT") == nullptr) {
    static const int pc_mode = getenv("PNX_CONV_PC") != nullptr ? atoi(getenv("PNX_CONV_PC")) : 1;  // bit 0: 64 -> 64 on the producer / consumer kernel (default), bit 1: 128 -> 128 and 256 -> 256, bit 2: 64 -> 320 / 384 / 448; 0: row-split kernels only (the cross-check)
    if (pc_mode != 0) {
      if (cin == 64 && cout == 64) return launch_pc<64, 64>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
      if (residual == nullptr && (pc_mode & 4)) {  // slower than the multi-pass row-split kernel (profiles/r06_conv_pc_ab.txt): each pass re-reads the B fragments for half the channels
        if (cin == 64 && cout == 320) return launch_pc<64, 320>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
        if (cin == 64 && cout == 384) return launch_pc<64, 384>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
        if (cin == 64 && cout == 448) return launch_pc<64, 448>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
      }
      if (pc_mode & 2) {
        if (cin == 128 && cout == 128) return launch_pc<128, 128>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
        if (cin == 256 && cout == 256) return launch_pc<256, 256>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
      }
    }
    if (cin == 64 && cout == 64) return launch_lds<64>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
    if (cin == 128 && cout == 128) return launch_ldsx<128, 128>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
    if (cin == 256 && cout == 64) return launch_ldsx<256, 64>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
    if (cin == 256 && cout == 256) return launch_ldsx<256, 256>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
    if (cin == 64 && cout == 384) return launch_lds<384>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
    if (cin == 64 && cout == 320) return launch_lds<320>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
    if (cin == 64 && cout == 448) return launch_lds<448>(x, wfrag, bias, residual, mask, y, batch, h, w, relu, row_dirty, tile_list, tile_count, st);
  }
  if (stride == 2 && residual == nullptr && getenv("PNX_CONV_DIRECT") == nullptr) {
    if (cin == 64 && cout == 128) return launch_s2<64, 128>(x, wfrag, bias, mask, y, batch, h, w, ho, wo, relu, row_dirty, st);
    if (cin == 128 && cout == 256) return launch_s2<128, 256>(x, wfrag, bias, mask, y, batch, h, w, ho, wo, relu, row_dirty, st);
    if (cin == 256 && cout == 256) return launch_s2<256, 256>(x, wfrag, bias, mask, y, batch, h, w, ho, wo, relu, row_dirty, st);
  }
#define PNX_CONV_CASE(CI, CO)                                                                                          \
  if (cin == CI && cout == CO) {                                                                                       \
    if (stride == 1) return launch<CI, CO, 1>(x, wfrag, bias, residual, mask, y, batch, h, w, ho, wo, relu, row_dirty, st);         \
    return launch<CI, CO, 2>(x, wfrag, bias, residual, mask, y, batch, h, w, ho, wo, relu, row_dirty, st);                         \
  }
  PNX_CONV_CASE(64, 64)
  PNX_CONV_CASE(64, 128)
  PNX_CONV_CASE(128, 128)
#undef PNX_CONV_CASE
  pnx_set_error("pnx_conv3x3: no kernel for %d -> %d channels", cin, cout);
  return PNX_ERR_UNSUPPORTED;
}

#ifndef PNX_CONV_F16  // the training-only entries exist once, on the bf16 instructions
}  // extern "C"
namespace {
#include "conv_dgrad_s2.h"
}
extern "C" {

// Data gradient of a stride-2 SparseConv2d layer (conv_dgrad_s2.h): g (batch, ho, wo, cout) NHWC bf16, wfrag_t = pnx_conv3x3_pack_weights(transposed = 1) of the
// layer's (cout, cin, 3, 3) weights, mask_in = active sites of the layer's INPUT (batch, h, w); dx (batch, h, w, cin) bf16, zeros at inactive sites, every site written.
int pnx_conv3x3_dgrad_s2_bf16(const void* g, const void* wfrag_t, const uint8_t* mask_in, void* dx, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                              pnx_stream_t stream) {
  PNX_REQUIRE(g && wfrag_t && mask_in && dx && batch > 0 && h > 0 && w > 0, PNX_ERR_INVALID, "pnx_conv3x3_dgrad_s2_bf16: bad arguments");
  PNX_REQUIRE((((uintptr_t)g | (uintptr_t)wfrag_t | (uintptr_t)dx) & 15) == 0, PNX_ERR_INVALID, "16-byte alignment required");
  hipStream_t st = (hipStream_t)stream;
  if (cin == 64 && cout == 128) return launch_dgrad_s2<128, 64, false>(g, nullptr, wfrag_t, nullptr, mask_in, dx, batch, h, w, st);
  if (cin == 128 && cout == 256) return launch_dgrad_s2<256, 128, false>(g, nullptr, wfrag_t, nullptr, mask_in, dx, batch, h, w, st);
  if (cin == 256 && cout == 256) return launch_dgrad_s2<256, 256, false>(g, nullptr, wfrag_t, nullptr, mask_in, dx, batch, h, w, st);
  pnx_set_error("pnx_conv3x3_dgrad_s2_bf16: no kernel for %d -> %d channels", cin, cout);
  return PNX_ERR_UNSUPPORTED;
}

// The same from the bf16 halves of an fp32 gradient and of the fp32 weights (pnx_split_f32), fp32 out: the stride-2 companion of pnx_conv3x3_x3.
int pnx_conv3x3_dgrad_s2_x3(const void* g_hi, const void* g_lo, const void* wfrag_t_hi, const void* wfrag_t_lo, const uint8_t* mask_in, float* dx, int32_t batch,
                            int32_t h, int32_t w, int32_t cin, int32_t cout, pnx_stream_t stream) {
  PNX_REQUIRE(g_hi && g_lo && wfrag_t_hi && wfrag_t_lo && mask_in && dx && batch > 0 && h > 0 && w > 0, PNX_ERR_INVALID, "pnx_conv3x3_dgrad_s2_x3: bad arguments");
  PNX_REQUIRE((((uintptr_t)g_hi | (uintptr_t)g_lo | (uintptr_t)wfrag_t_hi | (uintptr_t)wfrag_t_lo | (uintptr_t)dx) & 15) == 0, PNX_ERR_INVALID,
              "16-byte alignment required");
  hipStream_t st = (hipStream_t)stream;
  if (cin == 64 && cout == 128) return launch_dgrad_s2<128, 64, true>(g_hi, g_lo, wfrag_t_hi, wfrag_t_lo, mask_in, dx, batch, h, w, st);
  if (cin == 128 && cout == 256) return launch_dgrad_s2<256, 128, true>(g_hi, g_lo, wfrag_t_hi, wfrag_t_lo, mask_in, dx, batch, h, w, st);
  if (cin == 256 && cout == 256) return launch_dgrad_s2<256, 256, true>(g_hi, g_lo, wfrag_t_hi, wfrag_t_lo, mask_in, dx, batch, h, w, st);
  pnx_set_error("pnx_conv3x3_dgrad_s2_x3: no kernel for %d -> %d channels", cin, cout);
  return PNX_ERR_UNSUPPORTED;
}

// fp32 convolution out of three bf16 products: x = x_hi + x_lo and W = W_hi + W_lo (bf16 halves of fp32 values, pnx_split_f32; both weight halves in
// pnx_conv3x3_pack_weights order), y = x_hi W_hi + x_hi W_lo + x_lo W_hi accumulated in fp32 inside ONE launch (the low x low term is below the
// halves' own rounding, 2^-17 relative) and written as fp32 NHWC; zeros at inactive sites, every site written.  bias (fp32, may be null) starts the
// accumulators; no activation: this is the convolution of the fp32 training graph (forward, and the stride-1 data gradient with transposed weights).
int pnx_conv3x3_x3(const void* x_hi, const void* x_lo, const void* wfrag_hi, const void* wfrag_lo, const float* bias, const uint8_t* mask, float* y,
                   int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride, pnx_stream_t stream) {
  PNX_REQUIRE(x_hi && x_lo && wfrag_hi && wfrag_lo && y, PNX_ERR_INVALID, "null pointer");
  PNX_REQUIRE(batch > 0 && h > 0 && w > 0, PNX_ERR_INVALID, "bad shape");
  PNX_REQUIRE(stride == 1 || stride == 2, PNX_ERR_UNSUPPORTED, "stride %d", stride);
  PNX_REQUIRE((((uintptr_t)x_hi | (uintptr_t)x_lo | (uintptr_t)y | (uintptr_t)wfrag_hi | (uintptr_t)wfrag_lo | (uintptr_t)bias) & 15) == 0, PNX_ERR_INVALID,
              "16-byte alignment required");
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) {
    if (cin == 64 && cout == 64) return launch_pc_x3<64, 64>(x_hi, x_lo, wfrag_hi, wfrag_lo, bias, mask, y, batch, h, w, st);
    if (cin == 128 && cout == 128) return launch_pc_x3<128, 128>(x_hi, x_lo, wfrag_hi, wfrag_lo, bias, mask, y, batch, h, w, st);
    if (cin == 256 && cout == 256) return launch_pc_x3<256, 256>(x_hi, x_lo, wfrag_hi, wfrag_lo, bias, mask, y, batch, h, w, st);
  } else {
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    if (cin == 64 && cout == 128) return launch_s2<64, 128, true>(x_hi, wfrag_hi, bias, mask, y, batch, h, w, ho, wo, 0, nullptr, st, x_lo, wfrag_lo);
    if (cin == 128 && cout == 256) return launch_s2<128, 256, true>(x_hi, wfrag_hi, bias, mask, y, batch, h, w, ho, wo, 0, nullptr, st, x_lo, wfrag_lo);
    if (cin == 256 && cout == 256) return launch_s2<256, 256, true>(x_hi, wfrag_hi, bias, mask, y, batch, h, w, ho, wo, 0, nullptr, st, x_lo, wfrag_lo);
  }
  pnx_set_error("pnx_conv3x3_x3: no kernel for %d -> %d channels, stride %d", cin, cout, stride);
  return PNX_ERR_UNSUPPORTED;
}
#endif

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------------------
// Lazy SepHead (models.FusedPillarNeXt): the five regression branches of a task (reg, height, dim, rot, vel: conv3x3 64 -> 64 + BN +
// ReLU, then conv3x3 64 -> k each; det3d/models/heads/centerhead.py:12-59) evaluated ONLY at the candidate cells the decoder selected
// (<= pre_max per sample and class, centerhead.py:341-363) instead of at every cell: 5/6 of the head's convolution work and its
// 384-channel intermediate (9.6 GB of HBM traffic per 8-frame step) shrink to ~8 % of the cells.
// One launch covers every task.  One workgroup (8 waves) = 32 consecutive candidates of one (sample, class) list.  The 5 x 5 x 64 input
// patch of every candidate is staged in LDS (swizzled 16-byte chunks); the first convolution is an implicit GEMM on
// v_mfma_f32_32x32x16_bf16 with M = 320 output channels (10 tiles), N = 32 candidates x 9 neighbour positions = 288 pixels (9 tiles),
// K = 9 taps x 64 channels, dealt to the waves as 15 items (branch = 2 M tiles) x (3 N tiles); its outputs are rounded to bf16 like the
// dense kernel's intermediate and contracted on the spot with the second convolution's weights of their branch (the intermediate never
// exists); per-(pixel, branch) partial sums go through LDS and are added in a fixed order: deterministic.
namespace {

constexpr int kLzG = 32, kLzN = kLzG * 9, kLzNT = kLzN / 32;  // candidates, pixels, N tiles per workgroup
constexpr int kLzPatch = kLzG * 25 * 8;                       // uint4 slots of the staged patches
constexpr int kLzW2 = 10 * 9 * 32 * 3;                        // floats of the packed second-convolution weights
constexpr int kLzMaxTasks = 8, kLzMaxClasses = 32;
constexpr int kLzAhead = 2;  // taps of weight-fragment lookahead

struct LazyTaskDev {
  const uint16_t* up;
  const uint4* wfrag;
  const float *b1, *w2c, *b2;
  int h, w;
};
struct LazyArgs {
  LazyTaskDev t[kLzMaxTasks];
  signed char class_task[kLzMaxClasses];
  int nc_total, pre_max, bps;
};

__global__ __launch_bounds__(512) void k_sephead_lazy(const LazyArgs A, const int64_t* __restrict__ local, const int32_t* __restrict__ seg_len,
                                                      float* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char s_lz[];
  uint4* s_patch = reinterpret_cast<uint4*>(s_lz);                                    // [cand][cell 25][chunk 8 ^ swz]
  float* s_w2 = reinterpret_cast<float*>(s_patch + kLzPatch);                         // [M tile 10][pos 9][channel 32][3]
  float* s_part = s_w2 + kLzW2;                                                       // [pixel 288][branch 5][3]
  int* s_cell = reinterpret_cast<int*>(s_part + kLzN * 5 * 3);                        // [cand]: b*H*W + cell, or -1
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, px = lane & 31, kb = lane >> 5;
  const int seg = blockIdx.x / A.bps, c0 = (blockIdx.x - seg * A.bps) * kLzG;
  const int len = seg_len[seg];
  const int rows = min(kLzG, A.pre_max - c0);
  float* const outp = out + ((int64_t)seg * A.pre_max + c0) * 10;
  if (c0 >= len) {  // behind the end of the list: zero rows, no work
    for (int i = t; i < rows * 10; i += 512) outp[i] = 0.f;
    return;
  }
  const LazyTaskDev T = A.t[A.class_task[seg % A.nc_total]];
  CT_DECL
  const int H = T.h, W = T.w, HW = H * W;
  if (t < kLzG) s_cell[t] = (c0 + t < len) ? (int)local[(int64_t)seg * A.pre_max + c0 + t] : -1;
  for (int e = t; e < kLzW2 / 4; e += 512) reinterpret_cast<float4*>(s_w2)[e] = reinterpret_cast<const float4*>(T.w2c)[e];
  __syncthreads();
  CT_TOCK(0)
  // ---- stage the 5 x 5 patches (zeros outside the map / for the slots behind the end of the list): all of a thread's 16-byte
  // gathers are in flight before the first LDS write (one after the other they cost a memory latency each)
  {
    constexpr int NP = (kLzPatch + 511) / 512;
    uint4 v[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const int e = t + k * 512;
      const int q = e & 7, cell = (e >> 3) % 25, c = min(e / 200, kLzG - 1);
      const int lc = s_cell[c];
      v[k] = make_uint4(0, 0, 0, 0);
      if (lc >= 0 && e < kLzPatch) {
        const int b = lc / HW, rem = lc - b * HW;
        const int y = rem / W + cell / 5 - 2, x = rem % W + cell % 5 - 2;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
          v[k] = *reinterpret_cast<const uint4*>(T.up + (((int64_t)b * H + y) * W + x) * 64 + q * 8);
      }
    }
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const int e = t + k * 512;
      const int q = e & 7, cell = (e >> 3) % 25, c = e / 200;
      if (e < kLzPatch) s_patch[(c * 25 + cell) * 8 + (q ^ (cell & 7))] = v[k];
    }
  }
  CT_TOCK(1)
  __syncthreads();
  CT_TOCK(2)
#pragma unroll 1
  for (int item = wv; item < 15; item += 8) {
    const int br = item / 3, g0 = (item - br * 3) * 3;
    int pb[3], wc[3], posv[3];
    bool inside[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int pix = (g0 + j) * 32 + px;
      const int cand = pix / 9, pos = pix - cand * 9;
      posv[j] = pos;
      wc[j] = (pos / 3) * 5 + (pos % 3);  // top-left cell of this pixel's 3 x 3 window inside the 5 x 5 patch
      pb[j] = cand * 25 * 8;
      const int lc = s_cell[cand];
      inside[j] = lc >= 0;
      if (inside[j]) {
        const int rem = lc % HW;
        const int y = rem / W + pos / 3 - 1, x = rem % W + pos % 3 - 1;
        inside[j] = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      }
    }
    float ps[3][3] = {};
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
      const int mt = 2 * br + half;
      v16f acc[3];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const float bq = T.b1[mt * 32 + 8 * (i >> 2) + 4 * kb + (i & 3)];
#pragma unroll
        for (int j = 0; j < 3; j++) acc[j][i] = bq;
      }
      // Weight fragments (L2): kLzAhead taps ahead in straight-line code.  A tap is 12 MFMAs (~400 cycles) per wave; with one tap of lookahead
      // in a rolled loop the k-steps ran at the latency of their fragment loads (the kernel took 620 us for 2.6 M MFMAs = 110 us of pipe time).
      const uint4* wp = T.wfrag + mt * 64 + lane;
      uint4 wq[9][4];
#pragma unroll
      for (int tp = 0; tp < kLzAhead; tp++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) wq[tp][ks] = wp[(tp * 4 + ks) * 640];
#pragma unroll
      for (int tap = 0; tap < 9; tap++) {
        if (tap + kLzAhead < 9) {
          const uint4* wpt = wp;
          asm volatile("" : "+v"(wpt));  // the loads of a later tap are not hoisted to the top (144 live registers)
#pragma unroll
          for (int ks = 0; ks < 4; ks++) wq[tap + kLzAhead][ks] = wpt[((tap + kLzAhead) * 4 + ks) * 640];
        }
        const int toff = (tap / 3) * 5 + (tap % 3);
        int cwj[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
          cwj[j] = wc[j] + toff;
          asm volatile("" : "+v"(cwj[j]));  // recomputed per tap: hoisted out of the loops, the 108 LDS addresses of an item spill
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const int cw = cwj[j];
            const uint4 bq4 = s_patch[pb[j] + cw * 8 + ((ks * 2 + kb) ^ (cw & 7))];
            acc[j] = PNX_MFMA32(__builtin_bit_cast(el8, wq[tap][ks]), __builtin_bit_cast(el8, bq4), acc[j]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // keeps the fragment loads where they are: kLzAhead taps ahead, not all at the top (registers)
      }
      CT_TOCK(3)
      // ---- ReLU, bf16 rounding (the dense kernel's intermediate), contraction with the second convolution
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const float* w2p = s_w2 + ((mt * 9 + posv[j]) * 32 + 4 * kb) * 3;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const float tv = inside[j] ? round_el(fmaxf(acc[j][i], 0.f)) : 0.f;
          const float* w = w2p + (8 * (i >> 2) + (i & 3)) * 3;
          ps[j][0] = __builtin_fmaf(tv, w[0], ps[j][0]);
          ps[j][1] = __builtin_fmaf(tv, w[1], ps[j][1]);
          ps[j][2] = __builtin_fmaf(tv, w[2], ps[j][2]);
        }
      }
      CT_TOCK(4)
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int pix = (g0 + j) * 32 + px;
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const float v = ps[j][q] + __shfl_xor(ps[j][q], 32);  // the other half of the wave holds the other 16 channels of each tile
        if (kb == 0) s_part[(pix * 5 + br) * 3 + q] = v;
      }
    }
  }
  CT_TOCK(5)
  __syncthreads();
  CT_TOCK(6)
  // ---- out[cand][o] = b2[o] + sum over the 9 positions of the output's branch, in a fixed order; rounded to bf16
  for (int e = t; e < rows * 10; e += 512) {
    const int cand = e / 10, o = e - cand * 10;
    const int br = o < 2 ? 0 : (o < 3 ? 1 : (o < 6 ? 2 : (o < 8 ? 3 : 4)));
    const int q = o - (br == 0 ? 0 : (br == 1 ? 2 : (br == 2 ? 3 : (br == 3 ? 6 : 8))));
    float s = 0.f;
    if (s_cell[cand] >= 0) {
      s = T.b2[o];
      for (int pos = 0; pos < 9; pos++) s += s_part[((cand * 9 + pos) * 5 + br) * 3 + q];
      s = round_el(s);
    }
    outp[e] = s;
  }
  CT_TOCK(7)
  CT_FLUSH
}

constexpr size_t kLzLds = (size_t)kLzPatch * 16 + (size_t)kLzW2 * 4 + (size_t)kLzN * 5 * 3 * 4 + kLzG * 4;
static_assert(kLzLds <= 160 * 1024, "k_sephead_lazy: LDS budget");

}  // namespace

extern "C" int PNX_CONV_FN(pnx_sephead_lazy)(const PnxLazyTask* tasks, int32_t n_tasks, const int32_t* class_task, int32_t nc_total, int32_t batch,
                                     const int64_t* local, const int32_t* seg_len, int32_t pre_max, float* out, pnx_stream_t stream) {
  PNX_REQUIRE(tasks && class_task && local && seg_len && out && batch > 0 && pre_max > 0, PNX_ERR_INVALID, "bad arguments");
  PNX_REQUIRE(n_tasks >= 1 && n_tasks <= kLzMaxTasks && nc_total >= 1 && nc_total <= kLzMaxClasses, PNX_ERR_UNSUPPORTED, "at most 8 tasks / 32 classes");
  LazyArgs a;
  for (int i = 0; i < n_tasks; i++) {
    const PnxLazyTask& s = tasks[i];
    PNX_REQUIRE(s.up && s.wfrag1 && s.bias1 && s.w2c && s.bias2 && s.h > 0 && s.w > 0, PNX_ERR_INVALID, "bad task");
    PNX_REQUIRE((int64_t)batch * s.h * s.w < ((int64_t)1 << 31), PNX_ERR_UNSUPPORTED, "map too large for 32-bit cell indices");
    PNX_REQUIRE((((uintptr_t)s.up | (uintptr_t)s.wfrag1 | (uintptr_t)s.w2c) & 15) == 0, PNX_ERR_INVALID, "16-byte alignment required");
    a.t[i] = LazyTaskDev{(const uint16_t*)s.up, (const uint4*)s.wfrag1, s.bias1, s.w2c, s.bias2, s.h, s.w};
  }
  for (int c = 0; c < nc_total; c++) {
    PNX_REQUIRE(class_task[c] >= 0 && class_task[c] < n_tasks, PNX_ERR_INVALID, "class_task out of range");
    a.class_task[c] = (signed char)class_task[c];
  }
  a.nc_total = nc_total, a.pre_max = pre_max, a.bps = (pre_max + kLzG - 1) / kLzG;
  static bool attr = false;
  if (!attr) {
    PNX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sephead_lazy), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLzLds));
    attr = true;
  }
  const unsigned nb = (unsigned)((int64_t)batch * nc_total * a.bps);
  k_sephead_lazy<<<nb, 512, kLzLds, (hipStream_t)stream>>>(a, local, seg_len, out);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}
