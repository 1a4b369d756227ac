// masked_bn.hip -- train-mode BatchNorm over the ACTIVE sites of a dense NHWC map + residual + ReLU + mask, forward and backward, gfx950.
//
// The reference normalises the features of a SparseConvTensor with BatchNorm1d (det3d/models/utils/sparse_conv.py:31-37, 57-60: statistics
// over the active sites only); this package's masked-dense stand-in did that with ~12 PyTorch elementwise / reduction passes over full fp32
// maps in the forward and as many in the backward (models._MaskedBNActFn): 212 of the 341 ms of a C2 x 4-frame bf16 training step, although
// only 17-30 % of the cells are active.  Here:
//   stats       one pass over the active sites: per channel sum x, sum x^2 (fp32 partials per workgroup, summed in fp64 by the caller), count
//   apply       y = relu(x * a + b [+ residual]) at the active sites, zeros elsewhere          (a = gamma * invstd, b = beta - mean * a)
//   bwd_stats   g = gy * [pre > 0] at the active sites (pre recomputed from x): per channel sum g, sum g * xhat
//   bwd_apply   dx = a * (g - mean_g - xhat * mean_gx) at the active sites, zeros elsewhere; dresidual = g
// Inactive sites are never read.  Between stats and apply the caller (models.py) all-reduces the sums under SyncBatchNorm
// (tools/train.py:56) and updates the running statistics.  One thread = 8 consecutive channels of one site (16 bytes of bf16).
#include "pnx_common.h"

namespace {

constexpr int kMbnBlocks = 1024;

template <typename T>
struct Ld8;
template <>
struct Ld8<uint16_t> {  // bf16
  static __device__ __forceinline__ void load(const uint16_t* p, float (&v)[8]) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; k++) v[2 * k] = __uint_as_float(w[k] << 16), v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
  }
  static __device__ __forceinline__ uint32_t rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
  }
  static __device__ __forceinline__ void store(uint16_t* p, const float (&v)[8]) {
    uint4 q;
    q.x = rne(v[0]) | (rne(v[1]) << 16), q.y = rne(v[2]) | (rne(v[3]) << 16), q.z = rne(v[4]) | (rne(v[5]) << 16), q.w = rne(v[6]) | (rne(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = q;
  }
};
template <>
struct Ld8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};

// sums of NV vectors of 8 channels per thread -> partials[block][NV * C (+1: count)]: threads with equal channel chunk are added up
template <int NV>
__device__ __forceinline__ void block_reduce(const float (&acc)[NV][8], float cnt, bool with_cnt, int cvec, int C, float* __restrict__ out) {
  __shared__ float s_red[256 * 8];
  __shared__ float s_cnt[256];
  const int t = threadIdx.x, chunk = t % cvec, rows = 256 / cvec;
  for (int v = 0; v < NV; v++) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; k++) s_red[t * 8 + k] = acc[v][k];
    if (v == 0) s_cnt[t] = cnt;
    __syncthreads();
    if (t < cvec * 8) {  // (chunk c, channel k) = t / 8, t % 8
      const int c = t >> 3, k = t & 7;
      float s = 0.f;
      for (int r = 0; r < rows; r++) s += s_red[(r * cvec + c) * 8 + k];
      out[v * C + c * 8 + k] = s;
    }
    if (v == 0 && with_cnt && t == 0) {
      float s = 0.f;
      for (int r = 0; r < rows; r++) s += s_cnt[r * cvec];  // chunk-0 threads counted the sites
      out[NV * C] = s;
    }
  }
  (void)chunk;
}

template <typename T>
__global__ __launch_bounds__(256) void k_mbn_stats(const T* __restrict__ x, const float* __restrict__ mask, int64_t n, int C, const float* __restrict__ center,
                                                   float* __restrict__ partials) {
  const int cvec = C >> 3, t = threadIdx.x, chunk = t % cvec, rows = 256 / cvec;
  float acc[2][8] = {};
  float cnt = 0.f;
  // sums of (x - center) and (x - center)^2: with the running mean as the centre the one-pass variance E[d^2] - E[d]^2 does not cancel
  // when |mean| >> std (the plain E[x^2] - mean^2 loses the variance's leading digits in fp32 there)
  float ctr[8] = {};
  if (center != nullptr) {
#pragma unroll
    for (int k = 0; k < 8; k++) ctr[k] = center[chunk * 8 + k];
  }
  // four sites of the thread's walk per iteration, their mask words and then their lines requested together (round 6: one site at a time was a chain of two
  // dependent loads per iteration -- 36 us for a 66 MB map); the additions keep the walk's order
  const int64_t stride = (int64_t)gridDim.x * rows;
  for (int64_t s0 = (int64_t)blockIdx.x * rows + t / cvec; s0 < n; s0 += 4 * stride) {
    float m[4], v[4][8];
#pragma unroll
    for (int u = 0; u < 4; u++) m[u] = s0 + u * stride < n ? (mask != nullptr ? mask[s0 + u * stride] : 1.f) : 0.f;
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (m[u] != 0.f) Ld8<T>::load(x + (s0 + u * stride) * C + chunk * 8, v[u]);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (m[u] == 0.f) continue;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float d = v[u][k] - ctr[k];
        acc[0][k] += d, acc[1][k] += d * d;
      }
      if (chunk == 0) cnt += 1.f;
    }
  }
  block_reduce<2>(acc, cnt, true, cvec, C, partials + (size_t)blockIdx.x * (2 * C + 1));
}

// 8 per-channel values of a thread's channel chunk, read once: in the apply passes a thread's chunk never changes (the grid stride is a multiple of the
// chunks per site), and fetching scale / shift / mean / ... per vector was 4-12 extra load instructions for every 16 or 32 bytes of map
__device__ __forceinline__ void ld_par(const float* __restrict__ p, int c0, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p + c0), b = *reinterpret_cast<const float4*>(p + c0 + 4);
  v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
}

template <typename T, bool HAS_RES>
__global__ __launch_bounds__(256) void k_mbn_apply(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ mask, int64_t n, int C,
                                                   const float* __restrict__ scale, const float* __restrict__ shift, int relu, T* __restrict__ y) {
  const int cvec = C >> 3, csh = __builtin_ctz(cvec);
  const int64_t nvec = n * cvec, stride = (int64_t)gridDim.x * 256;
  const int c0 = (int)(threadIdx.x & (cvec - 1)) * 8;  // 256 and the grid stride are multiples of cvec (a power of two <= 32)
  float sc[8], sh[8];
  ld_par(scale, c0, sc), ld_par(shift, c0, sh);
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < nvec; i0 += 2 * stride) {  // two vectors per iteration, their loads requested together
    float m[2], v[2][8], r[2][8] = {};
#pragma unroll
    for (int u = 0; u < 2; u++) m[u] = i0 + u * stride < nvec ? (mask != nullptr ? mask[(i0 + u * stride) >> csh] : 1.f) : 0.f;
#pragma unroll
    for (int u = 0; u < 2; u++)
      if (m[u] != 0.f) {
        const int64_t off = ((i0 + u * stride) >> csh) * C + c0;
        Ld8<T>::load(x + off, v[u]);
        if (HAS_RES) Ld8<T>::load(res + off, r[u]);
      }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (i0 + u * stride >= nvec) continue;
      float o[8] = {};
      if (m[u] != 0.f) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const float p = v[u][k] * sc[k] + sh[k] + r[u][k];
          o[k] = relu ? fmaxf(p, 0.f) : p;
        }
      }
      Ld8<T>::store(y + ((i0 + u * stride) >> csh) * C + c0, o);
    }
  }
}

// the per-channel vectors of the backward passes for one channel chunk
struct BwdPar {
  float sc[8], sh[8], mu[8], is[8];
};
__device__ __forceinline__ void ld_bwd_par(BwdPar& P, const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                                           const float* __restrict__ invstd, int c0) {
  ld_par(scale, c0, P.sc), ld_par(shift, c0, P.sh), ld_par(mean, c0, P.mu), ld_par(invstd, c0, P.is);
}

// g of one (site, chunk): gy gated by the recomputed pre-activation; xh = (x - mean) * invstd
template <typename T, bool HAS_RES>
__device__ __forceinline__ void bwd_terms(const T* __restrict__ gy, const T* __restrict__ x, const T* __restrict__ res, int64_t off, const BwdPar& P, int relu,
                                          float (&g)[8], float (&xh)[8]) {
  float v[8], r[8] = {};
  Ld8<T>::load(gy + off, g);
  Ld8<T>::load(x + off, v);
  if (HAS_RES) Ld8<T>::load(res + off, r);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const float pre = v[k] * P.sc[k] + P.sh[k] + r[k];
    if (relu && !(pre > 0.f)) g[k] = 0.f;
    xh[k] = (v[k] - P.mu[k]) * P.is[k];
  }
}

template <typename T, bool HAS_RES>
__global__ __launch_bounds__(256) void k_mbn_bwd_stats(const T* __restrict__ gy, const T* __restrict__ x, const T* __restrict__ res,
                                                       const float* __restrict__ mask, int64_t n, int C, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       int relu, float* __restrict__ partials) {
  const int cvec = C >> 3, t = threadIdx.x, chunk = t % cvec, rows = 256 / cvec;
  BwdPar P;
  ld_bwd_par(P, scale, shift, mean, invstd, chunk * 8);
  float acc[2][8] = {};
  const int64_t stride = (int64_t)gridDim.x * rows;  // two sites per iteration, their loads requested together (see k_mbn_stats)
  for (int64_t s0 = (int64_t)blockIdx.x * rows + t / cvec; s0 < n; s0 += 2 * stride) {
    float m[2], g[2][8], xh[2][8];
#pragma unroll
    for (int u = 0; u < 2; u++) m[u] = s0 + u * stride < n ? (mask != nullptr ? mask[s0 + u * stride] : 1.f) : 0.f;
#pragma unroll
    for (int u = 0; u < 2; u++)
      if (m[u] != 0.f) bwd_terms<T, HAS_RES>(gy, x, res, (s0 + u * stride) * C + chunk * 8, P, relu, g[u], xh[u]);
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (m[u] == 0.f) continue;
#pragma unroll
      for (int k = 0; k < 8; k++) acc[0][k] += g[u][k], acc[1][k] += g[u][k] * xh[u][k];
    }
  }
  block_reduce<2>(acc, 0.f, false, cvec, C, partials + (size_t)blockIdx.x * (2 * C));
}

template <typename T, bool HAS_RES>
__global__ __launch_bounds__(256) void k_mbn_bwd_apply(const T* __restrict__ gy, const T* __restrict__ x, const T* __restrict__ res,
                                                       const float* __restrict__ mask, int64_t n, int C, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       int relu, const float* __restrict__ mg, const float* __restrict__ mgx, T* __restrict__ dx,
                                                       T* __restrict__ gres) {
  const int cvec = C >> 3, csh = __builtin_ctz(cvec);
  const int64_t nvec = n * cvec, stride = (int64_t)gridDim.x * 256;
  const int c0 = (int)(threadIdx.x & (cvec - 1)) * 8;
  BwdPar P;
  ld_bwd_par(P, scale, shift, mean, invstd, c0);
  float pmg[8], pmgx[8];
  ld_par(mg, c0, pmg), ld_par(mgx, c0, pmgx);
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < nvec; i0 += 2 * stride) {
    float m[2], g[2][8] = {}, xh[2][8] = {};
#pragma unroll
    for (int u = 0; u < 2; u++) m[u] = i0 + u * stride < nvec ? (mask != nullptr ? mask[(i0 + u * stride) >> csh] : 1.f) : 0.f;
#pragma unroll
    for (int u = 0; u < 2; u++)
      if (m[u] != 0.f) bwd_terms<T, HAS_RES>(gy, x, res, ((i0 + u * stride) >> csh) * C + c0, P, relu, g[u], xh[u]);
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (i0 + u * stride >= nvec) continue;
      float d[8] = {};
      if (m[u] != 0.f) {
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = P.sc[k] * (g[u][k] - pmg[k] - xh[u][k] * pmgx[k]);
      }
      const int64_t off = ((i0 + u * stride) >> csh) * C + c0;
      Ld8<T>::store(dx + off, d);
      if (gres != nullptr) Ld8<T>::store(gres + off, g[u]);
    }
  }
}

bool mbn_shape_ok(int C) {
  const int cvec = C >> 3;
  return C % 8 == 0 && cvec >= 1 && cvec <= 32 && (cvec & (cvec - 1)) == 0;
}

}  // namespace

namespace {

// sums[j] = sum over the rows of partials[r][j], in fp64 and a fixed order: 64 slices of the rows by 64 threads per column, then slice 0..63
__global__ __launch_bounds__(256) void k_mbn_reduce(const float* __restrict__ part, int rows, int cols, double* __restrict__ sums) {
  __shared__ double s_sum[64][4];
  const int t = threadIdx.x, el = t & 3, sl = t >> 2;
  const int j = blockIdx.x * 4 + el;
  double s = 0.0;
  if (j < cols) {
    int r = sl;
    for (; r + 192 < rows; r += 256) {  // four independent loads per step
      const float a = part[(int64_t)r * cols + j], b = part[(int64_t)(r + 64) * cols + j], c = part[(int64_t)(r + 128) * cols + j],
                  d = part[(int64_t)(r + 192) * cols + j];
      s += (double)a, s += (double)b, s += (double)c, s += (double)d;
    }
    for (; r < rows; r += 64) s += (double)part[(int64_t)r * cols + j];
  }
  s_sum[sl][el] = s;
  __syncthreads();
  if (sl == 0 && j < cols) {
    double r = s_sum[0][el];
#pragma unroll
    for (int k = 1; k < 64; k++) r += s_sum[k][el];
    sums[j] = r;
  }
}

// what the host did with ~20 small tensor statements per layer: batch statistics from [sum d | sum d^2 | count], the running statistics' update
// (torch's BatchNorm rule: unbiased variance, fp32 buffers), scale / shift for the apply kernel, and the vectors the backward needs
__global__ __launch_bounds__(256) void k_mbn_finalize(const double* __restrict__ sums, int C, const float* __restrict__ center, const float* __restrict__ weight,
                                                      const float* __restrict__ bias, double eps, double momentum, float* __restrict__ running_mean,
                                                      float* __restrict__ running_var, int64_t* __restrict__ nbt, float* __restrict__ mean_o,
                                                      float* __restrict__ invstd_o, float* __restrict__ scale_o, float* __restrict__ shift_o,
                                                      float* __restrict__ cnt_o) {
  const int c = threadIdx.x;
  const double cnt = sums[2 * C] < 1.0 ? 1.0 : sums[2 * C];
  if (c == 0) {
    cnt_o[0] = (float)cnt;
    if (nbt != nullptr) nbt[0] += 1;
  }
  if (c >= C) return;
  const double dmean = sums[c] / cnt;
  double var = sums[C + c] / cnt - dmean * dmean;
  var = var < 0.0 ? 0.0 : var;
  const double mean = (center != nullptr ? (double)center[c] : 0.0) + dmean;
  const double invstd = 1.0 / sqrt(var + eps);
  if (running_mean != nullptr) {
    const float keep = (float)(1.0 - momentum), mom = (float)momentum;
    const double unb = var * cnt / (cnt - 1.0 < 1.0 ? 1.0 : cnt - 1.0);
    running_mean[c] = running_mean[c] * keep + mom * (float)mean;   // center aliases running_mean: read above, written here by the same thread
    running_var[c] = running_var[c] * keep + mom * (float)unb;
  }
  const double w = (double)weight[c], b = (double)bias[c];
  mean_o[c] = (float)mean;
  invstd_o[c] = (float)invstd;
  scale_o[c] = (float)(invstd * w);
  shift_o[c] = (float)(b - mean * invstd * w);
}

// backward: parameter gradients from the LOCAL sums [sum g | sum g xhat], the two means the apply kernel subtracts from the (all-reduced) ones
__global__ __launch_bounds__(256) void k_mbn_bwd_finalize(const double* __restrict__ s_local, const double* __restrict__ s_global, int C,
                                                          const float* __restrict__ cnt, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ mean_g, float* __restrict__ mean_gx) {
  const int c = threadIdx.x;
  if (c >= C) return;
  const double n = (double)cnt[0];
  dbeta[c] = (float)s_local[c];
  dgamma[c] = (float)s_local[C + c];
  mean_g[c] = (float)(s_global[c] / n);
  mean_gx[c] = (float)(s_global[C + c] / n);
}

}  // namespace

extern "C" {

int32_t pnx_masked_bn_blocks(void) { return kMbnBlocks; }

int pnx_masked_bn_reduce(const float* partials, int32_t rows, int32_t cols, double* sums, pnx_stream_t stream) {
  PNX_REQUIRE(partials && sums && rows > 0 && cols > 0, PNX_ERR_INVALID, "pnx_masked_bn_reduce: bad arguments");
  k_mbn_reduce<<<(unsigned)((cols + 3) / 4), 256, 0, (hipStream_t)stream>>>(partials, rows, cols, sums);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

int pnx_masked_bn_finalize(const double* sums, int32_t channels, const float* center, const float* weight, const float* bias, double eps, double momentum,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                           float* count, pnx_stream_t stream) {
  PNX_REQUIRE(sums && weight && bias && mean && invstd && scale && shift && count && channels > 0 && channels <= 256, PNX_ERR_INVALID,
              "pnx_masked_bn_finalize: bad arguments (channels <= 256)");
  PNX_REQUIRE((running_mean == nullptr) == (running_var == nullptr), PNX_ERR_INVALID, "running_mean and running_var come together");
  k_mbn_finalize<<<1, 256, 0, (hipStream_t)stream>>>(sums, channels, center, weight, bias, eps, momentum, running_mean, running_var, num_batches_tracked, mean,
                                                     invstd, scale, shift, count);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

int pnx_masked_bn_bwd_finalize(const double* sums_local, const double* sums_global, int32_t channels, const float* count, float* dgamma, float* dbeta,
                               float* mean_g, float* mean_gx, pnx_stream_t stream) {
  PNX_REQUIRE(sums_local && count && dgamma && dbeta && mean_g && mean_gx && channels > 0 && channels <= 256, PNX_ERR_INVALID,
              "pnx_masked_bn_bwd_finalize: bad arguments (channels <= 256)");
  k_mbn_bwd_finalize<<<1, 256, 0, (hipStream_t)stream>>>(sums_local, sums_global != nullptr ? sums_global : sums_local, channels, count, dgamma, dbeta, mean_g,
                                                         mean_gx);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

#define MBN_COMMON_CHECKS                                                                                                        \
  PNX_REQUIRE(x && n_sites > 0 && mbn_shape_ok(channels), PNX_ERR_INVALID, "bad arguments (channels %d: 8, 16, 32, 64, 128 or 256)", channels); \
  PNX_REQUIRE(dtype == PNX_BF16 || dtype == PNX_F32, PNX_ERR_INVALID, "dtype %d: bf16 or fp32", dtype);                          \
  PNX_REQUIRE(n_sites < ((int64_t)1 << 40), PNX_ERR_INVALID, "too many sites");                                                  \
  hipStream_t st = (hipStream_t)stream;

int pnx_masked_bn_stats(const void* x, int32_t dtype, const float* mask, int64_t n_sites, int32_t channels, const float* center, float* partials,
                        pnx_stream_t stream) {
  MBN_COMMON_CHECKS
  PNX_REQUIRE(partials != nullptr, PNX_ERR_INVALID, "partials is NULL");
  if (dtype == PNX_BF16) k_mbn_stats<uint16_t><<<kMbnBlocks, 256, 0, st>>>((const uint16_t*)x, mask, n_sites, channels, center, partials);
  else k_mbn_stats<float><<<kMbnBlocks, 256, 0, st>>>((const float*)x, mask, n_sites, channels, center, partials);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

int pnx_masked_bn_apply(const void* x, const void* residual, int32_t dtype, const float* mask, int64_t n_sites, int32_t channels, const float* scale,
                        const float* shift, int32_t relu, void* y, pnx_stream_t stream) {
  MBN_COMMON_CHECKS
  PNX_REQUIRE(scale && shift && y, PNX_ERR_INVALID, "null pointer");
  const int64_t nvec = n_sites * (channels / 8);
  const unsigned nb = (unsigned)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
#define GO(T, R) k_mbn_apply<T, R><<<nb, 256, 0, st>>>((const T*)x, (const T*)residual, mask, n_sites, channels, scale, shift, relu, (T*)y)
  if (dtype == PNX_BF16) {
    if (residual) GO(uint16_t, true); else GO(uint16_t, false);
  } else {
    if (residual) GO(float, true); else GO(float, false);
  }
#undef GO
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

int pnx_masked_bn_bwd_stats(const void* gy, const void* x, const void* residual, int32_t dtype, const float* mask, int64_t n_sites, int32_t channels,
                            const float* scale, const float* shift, const float* mean, const float* invstd, int32_t relu, float* partials,
                            pnx_stream_t stream) {
  MBN_COMMON_CHECKS
  PNX_REQUIRE(gy && scale && shift && mean && invstd && partials, PNX_ERR_INVALID, "null pointer");
#define GO(T, R) k_mbn_bwd_stats<T, R><<<kMbnBlocks, 256, 0, st>>>((const T*)gy, (const T*)x, (const T*)residual, mask, n_sites, channels, scale, shift, mean, invstd, relu, partials)
  if (dtype == PNX_BF16) {
    if (residual) GO(uint16_t, true); else GO(uint16_t, false);
  } else {
    if (residual) GO(float, true); else GO(float, false);
  }
#undef GO
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

int pnx_masked_bn_bwd_apply(const void* gy, const void* x, const void* residual, int32_t dtype, const float* mask, int64_t n_sites, int32_t channels,
                            const float* scale, const float* shift, const float* mean, const float* invstd, int32_t relu, const float* mean_g,
                            const float* mean_gx, void* dx, void* dresidual, pnx_stream_t stream) {
  MBN_COMMON_CHECKS
  PNX_REQUIRE(gy && scale && shift && mean && invstd && mean_g && mean_gx && dx, PNX_ERR_INVALID, "null pointer");
  const int64_t nvec = n_sites * (channels / 8);
  const unsigned nb = (unsigned)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
#define GO(T, R) k_mbn_bwd_apply<T, R><<<nb, 256, 0, st>>>((const T*)gy, (const T*)x, (const T*)residual, mask, n_sites, channels, scale, shift, mean, invstd, relu, mean_g, mean_gx, (T*)dx, (T*)dresidual)
  if (dtype == PNX_BF16) {
    if (residual) GO(uint16_t, true); else GO(uint16_t, false);
  } else {
    if (residual) GO(float, true); else GO(float, false);
  }
#undef GO
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // extern "C"
