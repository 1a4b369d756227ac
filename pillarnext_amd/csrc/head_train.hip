// head_train.hip -- the SepHead's output convolutions in TRAINING (gfx950): nn.Conv2d(64, k, 3, padding 1, bias) with k = 1..4 output channels
// (reg 2, height 1, dim 3, rot 2, vel 2, hm = classes of the task; det3d/models/heads/centerhead.py:31-41), 36 of them per step on six tasks.
//
// Each is 2 * 9 * 64 * k flops per site over a 64-channel map -- 1.2 GFLOP against 133 MB of fp32 input at 4 x 360^2 sites: bandwidth-bound work that a
// matrix-core implicit GEMM with M or N = k <= 4 cannot fill.  MIOpen's kernels for it take 225 us (forward) and 240 us (weight gradient) per layer in fp32,
// 100-180 us in bf16; the map streams from HBM in ~30 us.  Here:
//   forward   16 lanes share a site (4 channels each, one 16-byte load per lane and site = a complete 256-byte line per group); a group walks a strip of 16
//             sites of a row with a sliding window of columns (3 loads per site instead of 9, the 12 loads of four sites requested together: the walk is latency-bound), the k x 9 x 4 weights of its channels stay in registers,
//             and the 16 partial dot products meet in a 4-step butterfly inside the group.  fp32 FMAs in the order tap-major, channel-minor.
//   wgrad     the same walk; every site adds dy[k] * x[tap][c] into k x 9 x 4 accumulators per lane (+ the bias gradient), a wave's four groups are
//             folded with two butterfly steps, a workgroup's four waves through LDS, and every workgroup writes ONE partial; a second kernel adds the partials in a fixed order (deterministic, like
//             csrc/conv_wgrad.hip).
// The data gradient stays on MIOpen (49 us per layer).  Element type of the maps: fp32 (the reference's training precision) or bf16 (autocast); weights,
// bias, accumulation and gradients fp32.
#include "pnx_common.h"

namespace {

constexpr int HT_C = 64;       // input channels
constexpr int HT_STRIP = 16;   // sites per strip
constexpr int HT_WAVES = 2048; // waves per launch (4 per workgroup; one weight-gradient partial per workgroup)

// a lane's 4 channels as they sit in memory (the window keeps them raw: bf16 maps then cost half the registers) and as fp32
template <typename T>
struct HtRaw;
template <>
struct HtRaw<float> {
  typedef float4 type;
  static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ float4 load(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ float4 cvt(const float4& r) { return r; }
};
template <>
struct HtRaw<uint16_t> {
  typedef uint2 type;
  static __device__ __forceinline__ uint2 zero() { return make_uint2(0u, 0u); }
  static __device__ __forceinline__ uint2 load(const uint16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ float4 cvt(const uint2& q) {
    return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u));
  }
};
// sites per iteration of the walk (their columns are requested together): as many as the registers hold
template <typename T, int K>
constexpr int ht_unr() {
  return K <= 2 ? 8 : (sizeof(T) == 2 && K == 3 ? 8 : 4);
}
__device__ __forceinline__ float ht_load1(const float* p) { return *p; }
__device__ __forceinline__ float ht_load1(const uint16_t* p) { return __uint_as_float((uint32_t)*p << 16); }
__device__ __forceinline__ void ht_store1(float* p, float v) { *p = v; }
__device__ __forceinline__ void ht_store1(uint16_t* p, float v) {
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7f800000u) != 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);
  *p = (uint16_t)(u >> 16);
}

// column `cx` of the window of row r: rows r - 1, r, r + 1, this lane's 4 channels; zeros outside the map
template <typename T>
__device__ __forceinline__ void ht_col(typename HtRaw<T>::type (&col)[3], const T* __restrict__ xb, int r, int cx, int H, int W, int ch) {
#pragma unroll
  for (int dy = 0; dy < 3; dy++) {
    const int iy = r - 1 + dy;
    col[dy] = HtRaw<T>::zero();
    if ((unsigned)iy < (unsigned)H && (unsigned)cx < (unsigned)W) col[dy] = HtRaw<T>::load(xb + ((int64_t)iy * W + cx) * HT_C + ch);
  }
}

// unit u of the walk: image b, band of 4 rows, block of 16 columns; group g of the wave takes row 4 band + g
struct HtUnit {
  int b, r, x0;
};
__device__ __forceinline__ HtUnit ht_unit(int64_t u, int bands, int cblocks, int grp) {
  HtUnit t;
  const int cb = (int)(u % cblocks);
  const int band = (int)((u / cblocks) % bands);
  t.b = (int)(u / ((int64_t)cblocks * bands));
  t.r = 4 * band + grp;
  t.x0 = cb * HT_STRIP;
  return t;
}

template <typename T, int K>
__global__ __launch_bounds__(256) void k_smallk_fwd(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y, int B,
                                                    int H, int W) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, j = lane & 15, ch = 4 * j;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  float wr[K][9][4];
#pragma unroll
  for (int k = 0; k < K; k++)
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int i = 0; i < 4; i++) wr[k][t][i] = w[((int64_t)k * HT_C + ch + i) * 9 + t];
  float bk[K];
#pragma unroll
  for (int k = 0; k < K; k++) bk[k] = bias != nullptr ? bias[k] : 0.f;
  const int bands = (H + 3) >> 2, cblocks = (W + HT_STRIP - 1) / HT_STRIP;
  const int64_t n_units = (int64_t)B * bands * cblocks;
  for (int64_t u = wave; u < n_units; u += n_waves) {
    const HtUnit t = ht_unit(u, bands, cblocks, grp);
    if (t.r >= H) continue;  // whole groups leave together: the butterflies below stay inside a group
    const T* xb = x + (int64_t)t.b * H * W * HT_C;
    // window of 6 columns [column][row ky]: columns x - 1, x of the first site, then the 4 new columns of this iteration's 4 sites -- their 12 loads are
    // requested together (the walk is latency-bound: what counts is the number of bytes a wave has in flight)
    constexpr int U = ht_unr<T, K>();
    typename HtRaw<T>::type win[U + 2][3];
    ht_col<T>(win[0], xb, t.r, t.x0 - 1, H, W, ch);
    ht_col<T>(win[1], xb, t.r, t.x0, H, W, ch);
#pragma unroll 1
    for (int i = 0; i < HT_STRIP; i += U) {
      if (t.x0 + i >= W) break;
#pragma unroll
      for (int q = 0; q < U; q++) ht_col<T>(win[2 + q], xb, t.r, t.x0 + i + q + 1, H, W, ch);
#pragma unroll
      for (int q = 0; q < U; q++) {
        const int ox = t.x0 + i + q;
        float s[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
          float a = 0.f;
#pragma unroll
          for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
              const float4 v = HtRaw<T>::cvt(win[q + kx][ky]);
              const float* wq = wr[k][ky * 3 + kx];
              a = fmaf(v.x, wq[0], a), a = fmaf(v.y, wq[1], a), a = fmaf(v.z, wq[2], a), a = fmaf(v.w, wq[3], a);
            }
          s[k] = a;
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
#pragma unroll
          for (int off = 8; off >= 1; off >>= 1) s[k] += __shfl_xor(s[k], off, 16);
        }
        float v = s[0] + bk[0];
#pragma unroll
        for (int k = 1; k < K; k++) v = j == k ? s[k] + bk[k] : v;
        if (j < K && ox < W) ht_store1(y + (((int64_t)t.b * H + t.r) * W + ox) * K + j, v);
      }
#pragma unroll
      for (int ky = 0; ky < 3; ky++) win[0][ky] = win[U][ky], win[1][ky] = win[U + 1][ky];
    }
  }
}

// partial of a workgroup: [K][9][64] weight-gradient sums, then K bias-gradient sums
template <int K>
constexpr int ht_part() {
  return K * 9 * HT_C + 64;
}

template <typename T, int K>
__global__ __launch_bounds__(256, K <= 1 && sizeof(T) == 4 ? 3 : 2) void k_smallk_wgrad(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, int B, int H, int W) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, j = lane & 15, ch = 4 * j;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  float acc[K][9][4], accb[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    accb[k] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[k][t][i] = 0.f;
  }
  const int bands = (H + 3) >> 2, cblocks = (W + HT_STRIP - 1) / HT_STRIP;
  const int64_t n_units = (int64_t)B * bands * cblocks;
  for (int64_t u = wave; u < n_units; u += n_waves) {
    const HtUnit t = ht_unit(u, bands, cblocks, grp);
    if (t.r >= H) continue;
    const T* xb = x + (int64_t)t.b * H * W * HT_C;
    constexpr int U = sizeof(T) == 2 && K <= 2 ? 8 : 4;
    typename HtRaw<T>::type win[U + 2][3];
    ht_col<T>(win[0], xb, t.r, t.x0 - 1, H, W, ch);
    ht_col<T>(win[1], xb, t.r, t.x0, H, W, ch);
#pragma unroll 1
    for (int i = 0; i < HT_STRIP; i += U) {
      if (t.x0 + i >= W) break;
#pragma unroll
      for (int q = 0; q < U; q++) ht_col<T>(win[2 + q], xb, t.r, t.x0 + i + q + 1, H, W, ch);
      const T* gp = dy + (((int64_t)t.b * H + t.r) * W + t.x0 + i) * K;
      float g[U][K];
#pragma unroll
      for (int q = 0; q < U; q++)
#pragma unroll
        for (int k = 0; k < K; k++) g[q][k] = t.x0 + i + q < W ? ht_load1(gp + q * K + k) : 0.f;
#pragma unroll
      for (int q = 0; q < U; q++) {
#pragma unroll
        for (int k = 0; k < K; k++) {
          accb[k] += g[q][k];
#pragma unroll
          for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
              const float4 v = HtRaw<T>::cvt(win[q + kx][ky]);
              float* aq = acc[k][ky * 3 + kx];
              aq[0] = fmaf(g[q][k], v.x, aq[0]), aq[1] = fmaf(g[q][k], v.y, aq[1]), aq[2] = fmaf(g[q][k], v.z, aq[2]), aq[3] = fmaf(g[q][k], v.w, aq[3]);
            }
        }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ky++) win[0][ky] = win[U][ky], win[1][ky] = win[U + 1][ky];
    }
  }
  // the wave's four groups hold the same channels: fold them (two butterfly steps), then the block's four waves through LDS in the order 0, 1, 2, 3:
  // ONE partial per workgroup
  __shared__ float s_fold[3][ht_part<K>()];
  const int wv = threadIdx.x >> 6;
  float* out = part + (int64_t)blockIdx.x * ht_part<K>();
  float bsum[K];
  float4 vsum[K][9];
#pragma unroll
  for (int k = 0; k < K; k++) {
    float b = accb[k];
    b += __shfl_xor(b, 16, 64);
    b += __shfl_xor(b, 32, 64);
    bsum[k] = b;
#pragma unroll
    for (int t = 0; t < 9; t++) {
      float4 v = make_float4(acc[k][t][0], acc[k][t][1], acc[k][t][2], acc[k][t][3]);
      v.x += __shfl_xor(v.x, 16, 64), v.y += __shfl_xor(v.y, 16, 64), v.z += __shfl_xor(v.z, 16, 64), v.w += __shfl_xor(v.w, 16, 64);
      v.x += __shfl_xor(v.x, 32, 64), v.y += __shfl_xor(v.y, 32, 64), v.z += __shfl_xor(v.z, 32, 64), v.w += __shfl_xor(v.w, 32, 64);
      vsum[k][t] = v;
      if (wv > 0 && grp == 0) *reinterpret_cast<float4*>(&s_fold[wv - 1][(k * 9 + t) * HT_C + ch]) = v;
    }
    if (wv > 0 && lane == 0) s_fold[wv - 1][K * 9 * HT_C + k] = b;
  }
  __syncthreads();
  if (wv == 0 && grp == 0) {
#pragma unroll
    for (int k = 0; k < K; k++) {
#pragma unroll
      for (int t = 0; t < 9; t++) {
        float4 v = vsum[k][t];
#pragma unroll
        for (int o = 0; o < 3; o++) {
          const float4 q = *reinterpret_cast<const float4*>(&s_fold[o][(k * 9 + t) * HT_C + ch]);
          v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
        }
        *reinterpret_cast<float4*>(out + (k * 9 + t) * HT_C + ch) = v;
      }
      if (lane == 0) out[K * 9 * HT_C + k] = ((bsum[k] + s_fold[0][K * 9 * HT_C + k]) + s_fold[1][K * 9 * HT_C + k]) + s_fold[2][K * 9 * HT_C + k];
    }
  }
}

// dw[k][c][tap] and dbias[k]: the workgroups' partials added in a FIXED order (32 slices of the partials by 32 threads per element, then slice 0..31)
template <int K>
__global__ __launch_bounds__(1024) void k_smallk_reduce(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ dbias, int n_part) {
  __shared__ float s_sum[32][33];
  const int t = threadIdx.x, el = t & 31, sl = t >> 5;
  const int e = blockIdx.x * 32 + el;  // (k * 9 + tap) * 64 + c, then the K bias sums
  const bool live = e < K * 9 * HT_C + K;
  float s = 0.f;
  if (live) {
    const float* p = part + e;
    for (int g = sl; g < n_part; g += 32) s += p[(int64_t)g * ht_part<K>()];
  }
  s_sum[sl][el] = s;
  __syncthreads();
  if (sl == 0 && live) {
    float r = s_sum[0][el];
#pragma unroll
    for (int k = 1; k < 32; k++) r += s_sum[k][el];
    if (e < K * 9 * HT_C) {
      const int c = e & 63, tap = (e >> 6) % 9, k = (e >> 6) / 9;
      dw[((int64_t)k * HT_C + c) * 9 + tap] = r;
    } else if (dbias != nullptr) {
      dbias[e - K * 9 * HT_C] = r;
    }
  }
}

int64_t ht_blocks(int batch, int h, int w) {
  const int64_t units = (int64_t)batch * ((h + 3) / 4) * ((w + HT_STRIP - 1) / HT_STRIP);
  int64_t nb = (units + 3) / 4;
  if (nb > HT_WAVES / 4) nb = HT_WAVES / 4;
  return nb < 1 ? 1 : nb;
}

template <typename T>
int launch_fwd(const void* x, const float* w, const float* bias, void* y, int batch, int h, int wd, int k, hipStream_t st) {
  const unsigned nb = (unsigned)ht_blocks(batch, h, wd);
  switch (k) {
    case 1: k_smallk_fwd<T, 1><<<nb, 256, 0, st>>>((const T*)x, w, bias, (T*)y, batch, h, wd); break;
    case 2: k_smallk_fwd<T, 2><<<nb, 256, 0, st>>>((const T*)x, w, bias, (T*)y, batch, h, wd); break;
    case 3: k_smallk_fwd<T, 3><<<nb, 256, 0, st>>>((const T*)x, w, bias, (T*)y, batch, h, wd); break;
    default: k_smallk_fwd<T, 4><<<nb, 256, 0, st>>>((const T*)x, w, bias, (T*)y, batch, h, wd); break;
  }
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

template <typename T, int K>
int launch_wgrad_k(const void* x, const void* dy, float* dw, float* dbias, int batch, int h, int wd, float* ws, hipStream_t st) {
  const unsigned nb = (unsigned)ht_blocks(batch, h, wd);
  k_smallk_wgrad<T, K><<<nb, 256, 0, st>>>((const T*)x, (const T*)dy, ws, batch, h, wd);
  PNX_LAUNCH_CHECK();
  k_smallk_reduce<K><<<(K * 9 * HT_C + K + 31) / 32, 1024, 0, st>>>(ws, dw, dbias, (int)nb);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

template <typename T>
int launch_wgrad_t(const void* x, const void* dy, float* dw, float* dbias, int batch, int h, int wd, int k, float* ws, hipStream_t st) {
  switch (k) {
    case 1: return launch_wgrad_k<T, 1>(x, dy, dw, dbias, batch, h, wd, ws, st);
    case 2: return launch_wgrad_k<T, 2>(x, dy, dw, dbias, batch, h, wd, ws, st);
    case 3: return launch_wgrad_k<T, 3>(x, dy, dw, dbias, batch, h, wd, ws, st);
    default: return launch_wgrad_k<T, 4>(x, dy, dw, dbias, batch, h, wd, ws, st);
  }
}

}  // namespace

extern "C" {

int pnx_conv3x3_smallk(const void* x, const float* weight, const float* bias, void* y, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t k, int32_t dtype,
                       pnx_stream_t stream) {
  PNX_REQUIRE(x && weight && y && batch > 0 && h > 0 && w > 0, PNX_ERR_INVALID, "pnx_conv3x3_smallk: bad arguments");
  PNX_REQUIRE(cin == HT_C && k >= 1 && k <= 4, PNX_ERR_UNSUPPORTED, "pnx_conv3x3_smallk: %d -> %d channels (64 -> 1..4)", cin, k);
  PNX_REQUIRE(dtype == PNX_F32 || dtype == PNX_BF16, PNX_ERR_UNSUPPORTED, "pnx_conv3x3_smallk: fp32 or bf16 maps");
  PNX_REQUIRE(((uintptr_t)x & 15) == 0, PNX_ERR_INVALID, "16-byte alignment required");
  if (dtype == PNX_F32) return launch_fwd<float>(x, weight, bias, y, batch, h, w, k, (hipStream_t)stream);
  return launch_fwd<uint16_t>(x, weight, bias, y, batch, h, w, k, (hipStream_t)stream);
}

size_t pnx_conv3x3_smallk_wgrad_workspace_bytes(int32_t k) {
  if (k < 1 || k > 4) return 0;
  return (size_t)(HT_WAVES / 4) * (k * 9 * HT_C + 64) * sizeof(float);
}

int pnx_conv3x3_smallk_wgrad(const void* x, const void* dy, float* dw, float* dbias, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t k, int32_t dtype,
                             void* workspace, size_t workspace_bytes, pnx_stream_t stream) {
  PNX_REQUIRE(x && dy && dw && workspace && batch > 0 && h > 0 && w > 0, PNX_ERR_INVALID, "pnx_conv3x3_smallk_wgrad: bad arguments");
  PNX_REQUIRE(cin == HT_C && k >= 1 && k <= 4, PNX_ERR_UNSUPPORTED, "pnx_conv3x3_smallk_wgrad: %d -> %d channels (64 -> 1..4)", cin, k);
  PNX_REQUIRE(dtype == PNX_F32 || dtype == PNX_BF16, PNX_ERR_UNSUPPORTED, "pnx_conv3x3_smallk_wgrad: fp32 or bf16 maps");
  PNX_REQUIRE((((uintptr_t)x | (uintptr_t)workspace) & 15) == 0, PNX_ERR_INVALID, "16-byte alignment required");
  PNX_REQUIRE(workspace_bytes >= pnx_conv3x3_smallk_wgrad_workspace_bytes(k), PNX_ERR_WORKSPACE, "workspace too small");
  if (dtype == PNX_F32) return launch_wgrad_t<float>(x, dy, dw, dbias, batch, h, w, k, (float*)workspace, (hipStream_t)stream);
  return launch_wgrad_t<uint16_t>(x, dy, dw, dbias, batch, h, w, k, (float*)workspace, (hipStream_t)stream);
}

}  // extern "C"
