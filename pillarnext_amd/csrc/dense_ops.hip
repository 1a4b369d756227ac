// dense_ops.hip -- elementwise epilogue of the masked-dense backbone stand-in, gfx950.
//
// After BatchNorm(eval) is folded into the preceding bias-free convolution, every spconv block of the reference
// (det3d/models/utils/sparse_conv.py:31-36, 52-61) reduces to
//        y = relu(conv(x) + b[c] (+ identity)) * mask[site]
// MIOpen computes conv(x); this kernel does the rest in ONE pass over the NHWC tensor (16-byte vector loads/stores,
// 8 bf16 per lane), instead of the 3-4 separate elementwise passes (BN, add, ReLU, mask) a module-by-module
// PyTorch graph makes.  HBM-bound: 2 (3 with residual) x tensor bytes.
#include "pnx_common.h"

namespace {

__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}

// 16-bit element <-> fp32 for the two canvas dtypes (DT = PNX_BF16 / PNX_F16)
template <int DT>
__device__ __forceinline__ float h2f(uint32_t h) {
  if (DT == PNX_BF16) return bf2f(h);
  return (float)__builtin_bit_cast(_Float16, (unsigned short)h);
}
template <int DT>
__device__ __forceinline__ uint32_t f2h(float f) {
  if (DT == PNX_BF16) return f2bf(f);
  return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f);  // round to nearest even
}

// x, res, out: bf16 NHWC with C channels (C % 8 == 0); bias fp32[C]; mask u8[sites] (or null) ; one thread = 8 channels
// RELU: 0 none, 1 relu(x + b + res), 2 relu(x + b) + res (BasicBlock of the neck: the residual joins after block2's own ReLU)
template <bool HAS_RES, bool HAS_MASK, int RELU, int DT>
__global__ __launch_bounds__(256) void k_bias_act_mask_bf16(const uint4* __restrict__ x, const uint4* __restrict__ res,
                                                            const float* __restrict__ bias, const uint8_t* __restrict__ mask,
                                                            uint4* __restrict__ out, int64_t n_vec, int cvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * 256) {
    const int64_t site = i / cvec;
    const int c0 = (int)(i - site * cvec) * 8;
    uint4 o = make_uint4(0, 0, 0, 0);
    const bool on = !HAS_MASK || mask[site] != 0;
    if (on) {
      const uint4 v = x[i];
      uint4 r = make_uint4(0, 0, 0, 0);
      if (HAS_RES) r = res[i];
      const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
      const uint32_t vw[4] = {v.x, v.y, v.z, v.w}, rw[4] = {r.x, r.y, r.z, r.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      uint32_t ow[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float lo = h2f<DT>(vw[k] & 0xffffu) + bb[2 * k], hi = h2f<DT>(vw[k] >> 16) + bb[2 * k + 1];
        if (RELU == 2) {
          lo = fmaxf(lo, 0.f);
          hi = fmaxf(hi, 0.f);
        }
        if (HAS_RES) {
          lo += h2f<DT>(rw[k] & 0xffffu);
          hi += h2f<DT>(rw[k] >> 16);
        }
        if (RELU == 1) {
          lo = fmaxf(lo, 0.f);
          hi = fmaxf(hi, 0.f);
        }
        ow[k] = f2h<DT>(lo) | (f2h<DT>(hi) << 16);
      }
      o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
    out[i] = o;
  }
}

// out = [relu](sum_k src_k + bias): the five partial results of the folded ASPP neck (models.py) summed in fp32 in ONE pass
// instead of four bf16 read-modify-write adds + an epilogue.  bf16 NHWC, one thread = 8 channels.
struct SumSrcs {
  const uint4* p[8];
};
template <int N, bool RELU, int DT>
__global__ __launch_bounds__(256) void k_sum_bias_act_bf16(SumSrcs srcs, const float* __restrict__ bias, uint4* __restrict__ out, int64_t n_vec,
                                                           int cvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (int64_t)gridDim.x * 256) {
    const int c0 = (int)(i % cvec) * 8;
    const float4 b0 = *reinterpret_cast<const float4*>(bias + c0), b1 = *reinterpret_cast<const float4*>(bias + c0 + 4);
    float a[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    uint4 v[N];
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = srcs.p[k][i];
#pragma unroll
    for (int k = 0; k < N; k++) {
      const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        a[2 * q] += h2f<DT>(w[q] & 0xffffu);
        a[2 * q + 1] += h2f<DT>(w[q] >> 16);
      }
    }
    uint32_t ow[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      float lo = a[2 * q], hi = a[2 * q + 1];
      if (RELU) {
        lo = fmaxf(lo, 0.f);
        hi = fmaxf(hi, 0.f);
      }
      ow[q] = f2h<DT>(lo) | (f2h<DT>(hi) << 16);
    }
    out[i] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
}

// 3x3 / stride-s / pad-1 max-pool of the occupancy mask (the active-site rule of a strided sparse conv).
__global__ __launch_bounds__(256) void k_mask_pool(const uint8_t* __restrict__ in, int B, int H, int W, int stride, uint8_t* __restrict__ out,
                                                   int Ho, int Wo) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * Ho * Wo) return;
  const int xo = (int)(idx % Wo), yo = (int)((idx / Wo) % Ho), b = (int)(idx / ((int64_t)Wo * Ho));
  uint8_t m = 0;
  for (int dy = -1; dy <= 1; dy++) {
    const int y = yo * stride + dy;
    if (y < 0 || y >= H) continue;
    for (int dx = -1; dx <= 1; dx++) {
      const int x = xo * stride + dx;
      if (x < 0 || x >= W) continue;
      m |= in[((int64_t)b * H + y) * W + x];
    }
  }
  out[idx] = m ? 1 : 0;
}

// Four output sites per thread from aligned 32-bit reads (three words per input row instead of up to 36 byte loads): W and Wo
// multiples of 4, stride 1 or 2.  Byte k of a word = column x + k (little endian); mask values may be any non-zero byte.
template <int STRIDE>
__global__ __launch_bounds__(256) void k_mask_pool4(const uint8_t* __restrict__ in, int B, int H, int W, uint8_t* __restrict__ out, int Ho, int Wo) {
  const int wq = Wo >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * Ho * wq) return;
  const int xo = (int)(idx % wq) * 4, yo = (int)((idx / wq) % Ho), b = (int)(idx / ((int64_t)wq * Ho));
  const int x0 = xo * STRIDE;
  uint32_t A = 0, M = 0, C = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; dy++) {
    const int y = yo * STRIDE + dy;
    if (y < 0 || y >= H) continue;
    const uint8_t* row = in + ((int64_t)b * H + y) * W;
    if (x0 >= 4) A |= *reinterpret_cast<const uint32_t*>(row + x0 - 4);
    M |= *reinterpret_cast<const uint32_t*>(row + x0);
    if (x0 + 4 < W) C |= *reinterpret_cast<const uint32_t*>(row + x0 + 4);
  }
  uint32_t r;
  if (STRIDE == 1) {
    r = M | ((M >> 8) | (C << 24)) | ((M << 8) | (A >> 24));
  } else {
    const uint32_t o0 = (A >> 24) | M | (M >> 8);          // byte 0: columns -1, 0, 1
    const uint32_t o1 = (M >> 8) | (M >> 16) | (M >> 24);  // byte 0: columns 1, 2, 3
    const uint32_t o2 = (M >> 24) | C | (C >> 8);          // byte 0: columns 3, 4, 5
    const uint32_t o3 = (C >> 8) | (C >> 16) | (C >> 24);  // byte 0: columns 5, 6, 7
    r = (o0 & 0xffu) | ((o1 & 0xffu) << 8) | ((o2 & 0xffu) << 16) | ((o3 & 0xffu) << 24);
  }
  r |= r >> 4;  // any bit of a byte -> its bit 0
  r |= r >> 2;
  r |= r >> 1;
  *reinterpret_cast<uint32_t*>(out + ((int64_t)b * Ho + yo) * Wo + xo) = r & 0x01010101u;
}

// Sixteen output sites per thread from aligned 16-byte reads (round 6: the 4-byte form above ran the 1440 x 1440 x 12 dilation -- 25 MB in, 25 MB out -- in
// 60 us): W, Wo multiples of 16 (and of 32 on the input side at stride 2), 16-byte aligned maps.
template <int STRIDE>
__global__ __launch_bounds__(256) void k_mask_pool16(const uint8_t* __restrict__ in, int B, int H, int W, uint8_t* __restrict__ out, int Ho, int Wo) {
  const int wq = Wo >> 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * Ho * wq) return;
  const int xo = (int)(idx % wq) * 16, yo = (int)((idx / wq) % Ho), b = (int)(idx / ((int64_t)wq * Ho));
  const int x0 = xo * STRIDE;
  constexpr int ND = 4 * STRIDE;  // input dwords under the output's 16 columns
  uint32_t D[ND + 2];             // D[0]: the dword in front (columns x0 - 4 .. x0 - 1), D[ND + 1]: the dword behind
#pragma unroll
  for (int k = 0; k < ND + 2; k++) D[k] = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; dy++) {
    const int y = yo * STRIDE + dy;
    if (y < 0 || y >= H) continue;
    const uint8_t* row = in + ((int64_t)b * H + y) * W;
    if (x0 >= 4) D[0] |= *reinterpret_cast<const uint32_t*>(row + x0 - 4);
#pragma unroll
    for (int q = 0; q < STRIDE; q++) {
      const uint4 v = *reinterpret_cast<const uint4*>(row + x0 + 16 * q);
      D[1 + 4 * q] |= v.x, D[2 + 4 * q] |= v.y, D[3 + 4 * q] |= v.z, D[4 + 4 * q] |= v.w;
    }
    if (x0 + 16 * STRIDE < W) D[ND + 1] |= *reinterpret_cast<const uint32_t*>(row + x0 + 16 * STRIDE);
  }
  uint32_t R[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    uint32_t r;
    if (STRIDE == 1) {
      const uint32_t A = D[q], M = D[q + 1], C = D[q + 2];
      r = M | ((M >> 8) | (C << 24)) | ((M << 8) | (A >> 24));
    } else {
      const uint32_t A = D[2 * q], M = D[2 * q + 1], C = D[2 * q + 2];
      const uint32_t o0 = (A >> 24) | M | (M >> 8);          // byte 0: columns -1, 0, 1
      const uint32_t o1 = (M >> 8) | (M >> 16) | (M >> 24);  // byte 0: columns 1, 2, 3
      const uint32_t o2 = (M >> 24) | C | (C >> 8);          // byte 0: columns 3, 4, 5
      const uint32_t o3 = (C >> 8) | (C >> 16) | (C >> 24);  // byte 0: columns 5, 6, 7
      r = (o0 & 0xffu) | ((o1 & 0xffu) << 8) | ((o2 & 0xffu) << 16) | ((o3 & 0xffu) << 24);
    }
    r |= r >> 4;  // any bit of a byte -> its bit 0
    r |= r >> 2;
    r |= r >> 1;
    R[q] = r & 0x01010101u;
  }
  *reinterpret_cast<uint4*>(out + ((int64_t)b * Ho + yo) * Wo + xo) = make_uint4(R[0], R[1], R[2], R[3]);
}

// fp32 -> two bf16 halves: hi = RNE(x), lo = RNE(x - hi); x - hi is exact in fp32, so hi + lo carries 16 mantissa bits of x (2^-17 relative).
// 8 values per thread: two 16-byte loads, two 16-byte stores.
__device__ __forceinline__ uint32_t bf16_rne(float v) {
  const uint32_t u = __float_as_uint(v);
  if ((u & 0x7f800000u) == 0x7f800000u) return (u >> 16) | ((u & 0xffffu) != 0 ? 0x40u : 0u);  // inf / nan stay what they are
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// MASKED: x is an NHWC map that is zero at the inactive sites of `mask` (c8 = channels / 8 threads per site): those sites are not read, zeros are written.
template <bool MASKED>
__global__ __launch_bounds__(256) void k_split_f32(const float4* __restrict__ x, uint4* __restrict__ hi, uint4* __restrict__ lo, int64_t n8,
                                                   const uint8_t* __restrict__ mask, int c8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    if (MASKED && mask[i / c8] == 0) {
      hi[i] = make_uint4(0, 0, 0, 0);
      lo[i] = make_uint4(0, 0, 0, 0);
      continue;
    }
    const float4 a = x[2 * i], b = x[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t h[8], l[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      h[k] = bf16_rne(v[k]);
      l[k] = bf16_rne(v[k] - __uint_as_float(h[k] << 16));
    }
    hi[i] = make_uint4(h[0] | h[1] << 16, h[2] | h[3] << 16, h[4] | h[5] << 16, h[6] | h[7] << 16);
    lo[i] = make_uint4(l[0] | l[1] << 16, l[2] | l[3] << 16, l[4] | l[5] << 16, l[6] | l[7] << 16);
  }
}

}  // namespace

extern "C" {

int pnx_split_f32(const float* x, void* hi, void* lo, int64_t n, const uint8_t* mask, int32_t channels, pnx_stream_t stream) {
  PNX_REQUIRE(x && hi && lo && n >= 0 && n % 8 == 0, PNX_ERR_INVALID, "pnx_split_f32: null pointer or a count that is not a multiple of 8");
  PNX_REQUIRE(mask == nullptr || (channels > 0 && channels % 8 == 0 && n % channels == 0), PNX_ERR_INVALID,
              "pnx_split_f32: with a mask, channels must be a multiple of 8 that divides n");
  PNX_REQUIRE((((uintptr_t)x | (uintptr_t)hi | (uintptr_t)lo) & 15) == 0, PNX_ERR_INVALID, "16-byte alignment required");
  if (n == 0) return PNX_OK;
  const int64_t n8 = n / 8;
  int64_t nb = (n8 + 255) / 256;
  if (nb > 256 * 16) nb = 256 * 16;
  if (mask != nullptr) k_split_f32<true><<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>((const float4*)x, (uint4*)hi, (uint4*)lo, n8, mask, channels / 8);
  else k_split_f32<false><<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>((const float4*)x, (uint4*)hi, (uint4*)lo, n8, nullptr, 1);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

int pnx_bias_act_mask(const void* x, const void* residual, const float* bias, const uint8_t* mask, void* out, int64_t sites,
                      int32_t channels, int32_t dtype, int32_t relu, pnx_stream_t stream) {
  PNX_REQUIRE(x && bias && out, PNX_ERR_INVALID, "null pointer");
  PNX_REQUIRE(dtype == PNX_BF16 || dtype == PNX_F16, PNX_ERR_UNSUPPORTED, "pnx_bias_act_mask is built for bf16 / f16 NHWC tensors");
  PNX_REQUIRE(channels > 0 && channels % 8 == 0 && sites >= 0, PNX_ERR_INVALID, "channels must be a positive multiple of 8");
  PNX_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)bias) & 15) == 0, PNX_ERR_INVALID, "16-byte alignment required");
  if (sites == 0) return PNX_OK;
  hipStream_t st = (hipStream_t)stream;
  const int cvec = channels / 8;
  const int64_t n_vec = sites * cvec;
  int64_t nb = (n_vec + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  const uint4 *xv = (const uint4*)x, *rv = (const uint4*)residual;
  uint4* ov = (uint4*)out;
#define PNX_LAUNCH_BAM(R_, M_, A_)                                                                                         \
  {                                                                                                                       \
    if (dtype == PNX_BF16) k_bias_act_mask_bf16<R_, M_, A_, PNX_BF16><<<(unsigned)nb, 256, 0, st>>>(xv, rv, bias, mask, ov, n_vec, cvec); \
    else k_bias_act_mask_bf16<R_, M_, A_, PNX_F16><<<(unsigned)nb, 256, 0, st>>>(xv, rv, bias, mask, ov, n_vec, cvec);     \
  }
  const bool r = residual != nullptr, m = mask != nullptr, a = relu != 0;
  if (relu == 2 && r && m) PNX_LAUNCH_BAM(true, true, 2)
  else if (relu == 2 && r) PNX_LAUNCH_BAM(true, false, 2)
  else if (r && m && a) PNX_LAUNCH_BAM(true, true, 1)
  else if (r && m) PNX_LAUNCH_BAM(true, true, 0)
  else if (r && a) PNX_LAUNCH_BAM(true, false, 1)
  else if (r) PNX_LAUNCH_BAM(true, false, 0)
  else if (m && a) PNX_LAUNCH_BAM(false, true, 1)
  else if (m) PNX_LAUNCH_BAM(false, true, 0)
  else if (a) PNX_LAUNCH_BAM(false, false, 1)
  else PNX_LAUNCH_BAM(false, false, 0)
#undef PNX_LAUNCH_BAM
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

int pnx_mask_pool3(const uint8_t* mask_in, int32_t batch, int32_t h, int32_t w, int32_t stride, uint8_t* mask_out, pnx_stream_t stream) {
  PNX_REQUIRE(mask_in && mask_out && batch > 0 && h > 0 && w > 0 && stride >= 1, PNX_ERR_INVALID, "bad arguments");
  const int ho = (h + 2 - 3) / stride + 1, wo = (w + 2 - 3) / stride + 1;
  const int64_t n = (int64_t)batch * ho * wo;
  if ((stride == 1 || stride == 2) && (w & 15) == 0 && (wo & 15) == 0 && (((uintptr_t)mask_in | (uintptr_t)mask_out) & 15) == 0) {
    const int64_t n16 = n / 16;
    if (stride == 1) k_mask_pool16<1><<<(unsigned)((n16 + 255) / 256), 256, 0, (hipStream_t)stream>>>(mask_in, batch, h, w, mask_out, ho, wo);
    else k_mask_pool16<2><<<(unsigned)((n16 + 255) / 256), 256, 0, (hipStream_t)stream>>>(mask_in, batch, h, w, mask_out, ho, wo);
    PNX_LAUNCH_CHECK();
    return PNX_OK;
  }
  if ((stride == 1 || stride == 2) && (w & 3) == 0 && (wo & 3) == 0 && (((uintptr_t)mask_in | (uintptr_t)mask_out) & 3) == 0) {
    const int64_t n4 = n / 4;
    if (stride == 1) k_mask_pool4<1><<<(unsigned)((n4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(mask_in, batch, h, w, mask_out, ho, wo);
    else k_mask_pool4<2><<<(unsigned)((n4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(mask_in, batch, h, w, mask_out, ho, wo);
    PNX_LAUNCH_CHECK();
    return PNX_OK;
  }
  k_mask_pool<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(mask_in, batch, h, w, stride, mask_out, ho, wo);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

int pnx_sum_bias_act(const void* const* srcs, int32_t n_src, const float* bias, void* out, int64_t sites, int32_t channels, int32_t dtype,
                     int32_t relu, pnx_stream_t stream) {
  PNX_REQUIRE(srcs && bias && out && sites > 0, PNX_ERR_INVALID, "bad arguments");
  PNX_REQUIRE(dtype == PNX_BF16 || dtype == PNX_F16, PNX_ERR_UNSUPPORTED, "dtype %d (bf16 / f16 only)", dtype);
  PNX_REQUIRE(n_src >= 1 && n_src <= 8, PNX_ERR_UNSUPPORTED, "%d summands (1..8)", n_src);
  PNX_REQUIRE(channels > 0 && channels % 8 == 0, PNX_ERR_UNSUPPORTED, "channels %d not a multiple of 8", channels);
  SumSrcs ss;
  for (int k = 0; k < 8; k++) {
    ss.p[k] = k < n_src ? (const uint4*)srcs[k] : nullptr;
    PNX_REQUIRE(k >= n_src || (srcs[k] && ((uintptr_t)srcs[k] & 15) == 0), PNX_ERR_INVALID, "summand %d: null or not 16-byte aligned", k);
  }
  hipStream_t st = (hipStream_t)stream;
  const int cvec = channels / 8;
  const int64_t n_vec = sites * cvec;
  int64_t nb = (n_vec + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
#define PNX_SUM_GO(N_, R_)                                                                                                  \
  {                                                                                                                         \
    if (dtype == PNX_BF16) k_sum_bias_act_bf16<N_, R_, PNX_BF16><<<(unsigned)nb, 256, 0, st>>>(ss, bias, (uint4*)out, n_vec, cvec); \
    else k_sum_bias_act_bf16<N_, R_, PNX_F16><<<(unsigned)nb, 256, 0, st>>>(ss, bias, (uint4*)out, n_vec, cvec);            \
  }
#define PNX_SUM_CASE(N_)                 \
  case N_:                               \
    if (relu) PNX_SUM_GO(N_, true)       \
    else PNX_SUM_GO(N_, false)           \
    break;
  switch (n_src) {
    PNX_SUM_CASE(1) PNX_SUM_CASE(2) PNX_SUM_CASE(3) PNX_SUM_CASE(4) PNX_SUM_CASE(5) PNX_SUM_CASE(6) PNX_SUM_CASE(7) PNX_SUM_CASE(8)
  }
#undef PNX_SUM_CASE
#undef PNX_SUM_GO
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // extern "C"
