// pfn_train.hip -- the PFN in TRAINING mode (pillar_encoder.py:35-50 x2 with BatchNorm1d batch statistics, :174-182) and its
// backward, without a single (N',32/64) intermediate in memory: every pass RECOMPUTES the per-point chain from the pillar-sorted
// decorated records of reader_bins.h.  Reference data flow for comparison: Linear -> BN -> ReLU -> scatter_max ->
// gather -> cat -> Linear -> BN -> ReLU -> scatter_max keeps ~8 tensors of (N',64) fp32 alive for autograd (77 MB each per 300 k
// points).
//
//   x0 = W0 f          h0 = relu(BN0(x0))     g0 = max over the pillar of h0      u = [h0, g0]      (f: decorated features, C0 = F+5)
//   x1 = W1 u          h1 = relu(BN1(x1))     out = max over the pillar of h1                       (BN: batch statistics over N' rows)
//
// Passes (host: pillarnext_amd/pfn_train.py; between the passes the host sums the per-wave partials in fp64 and, under SyncBN,
// all-reduces the SAME small vectors the reference's SyncBatchNorm exchanges -- 65 and 129 floats forward, 128 and 64 backward):
//   A  gram0    F1 = sum f, F2 = sum f f^T                       -> mean/var of x0 (x0 is linear in f: S0 = W0 F1, Q0 = diag(W0 F2 W0^T))
//   B  gram1    U1 = sum u, U2 = sum u u^T                       -> mean/var of x1 the same way; U1, U2 are kept for the backward
//   C  output   out (P,64)
//   D  bwd1     dz1 = G routed to the argmax row of every (pillar, channel), masked by ReLU:  D1 = sum dz1, D2 = sum dz1 * xhat1,
//               A = sum dz1^T u          -> dgamma1 = D2, dbeta1 = D1, dW1 = gamma1*invstd1 * (A - D1/N U1^T - D2/N * sum xhat1 u^T)
//   E  bwd0     dx1 = gamma1*invstd1*(dz1 - D1/N - xhat1*D2/N) for EVERY row, du = W1^T dx1, dh0 = du[:32] + (du[32:] summed over the
//               pillar, routed to the argmax row of g0), dz0 = dh0 * (h0 > 0):  E1 = sum dz0, E2 = sum dz0 * xhat0, B0 = sum dz0^T f
//               -> dgamma0 = E2, dbeta0 = E1, dW0 from B0, F1, F2 as above.  No gradient reaches the points (pe:91-123 are index math).
// Max ties: only possible between equal values; at 0 the ReLU gradient is 0, elsewhere the first row in record order takes the
// gradient (torch_scatter's choice there is unspecified as well).
//
// Execution: LANES = CHANNELS.  A wave walks whole pillars (a contiguous range of pillar ranks), one point at a time: the point's
// 12 feature words are wave-uniform scalars (s_load), lane l owns row l of W0 (l & 31) and of W1; u (and dx1) go through 256 bytes
// of the wave's LDS and come back as broadcast reads.  This is a VALU formulation (~100-400 instructions per point and pass): the
// training step is dominated by the dense backbone (profiles/r02_train_step_c2_b4.log), what matters here is that the reader's
// activations and their gradients never touch HBM.
#include "pnx_common.h"

namespace {

// parameter block (floats), built by the host for every pass
struct TP {
  static constexpr int W1 = 0;                                                  // 64 x 64 row-major
  static constexpr int MU1 = 4096, IS1 = MU1 + 64, GA1 = IS1 + 64, BE1 = GA1 + 64;  // mean, 1/sqrt(var+eps), gamma, beta of BN1
  static constexpr int M1 = BE1 + 64, M2 = M1 + 64;                             // D1/N, D2/N (pass E)
  static constexpr int MU0 = M2 + 64, IS0 = MU0 + 32, GA0 = IS0 + 32, BE0 = GA0 + 32;
  static constexpr int W0 = BE0 + 32;                                           // 32 x C0 row-major
};

__device__ __forceinline__ float rec_f(const uint32_t* __restrict__ rp, int k) { return __uint_as_float(rp[8 * (k & 1) + (k >> 1)]); }

enum { MODE_GRAM1 = 1, MODE_OUT = 2, MODE_BWD1 = 3, MODE_BWD0 = 4 };

// per-wave partial layouts (floats per wave)
//   GRAM1: [64 lanes][65]  U2 row l (64) | U1[l]
//   BWD1 : [64 lanes][66]  A row l (64) | D1[l] | D2[l]
//   BWD0 : [32 lanes][C0 + 2]  B0 row l (C0) | E1[l] | E2[l]
template <int C0, int MODE>
__global__ __launch_bounds__(256) void k_pfn_train(const uint32_t* __restrict__ rec, const uint32_t* __restrict__ pfirst,
                                                  const uint32_t* __restrict__ pcnt, const int32_t* __restrict__ counters,
                                                  const float* __restrict__ prm, float* __restrict__ part, const float* __restrict__ G,
                                                  const float* __restrict__ out_saved, float* __restrict__ out, int64_t out_rows) {
  __shared__ __align__(16) float s_x[4][128];  // per wave: u[64] | dx1[64]
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, l5 = l & 31;
  float* su = s_x[wv];
  float* sd = su + 64;
  const int P = counters[0];
  const int nw = gridDim.x * 4, w = blockIdx.x * 4 + wv;
  // a wave's pillars: a contiguous range of ranks holding ~1/nw of the POINTS (round 6: equal pillar counts gave the waves of the dense near range 3-5 x the
  // work of the others, and the pass lasts as long as its slowest wave).  pfirst is ascending: the range starts at the first pillar whose first record is at
  // or behind the wave's share of the records -- the same split for the same cloud, whatever the timing.
  const int64_t n_rec = counters[1];
  auto first_at = [&](int64_t target) -> int {
    int lo = 0, hi = P;  // first r in [0, P] with pfirst[r] >= target
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)pfirst[mid] >= target) hi = mid;
      else lo = mid + 1;
    }
    return lo;
  };
  const int r0 = __builtin_amdgcn_readfirstlane(w == 0 ? 0 : first_at(n_rec * w / nw));
  const int r1 = __builtin_amdgcn_readfirstlane(w + 1 == nw ? P : first_at(n_rec * (w + 1) / nw));

  float w0[C0];
#pragma unroll
  for (int k = 0; k < C0; k++) w0[k] = prm[TP::W0 + l5 * C0 + k];
  const float a0 = prm[TP::GA0 + l5] * prm[TP::IS0 + l5], mu0 = prm[TP::MU0 + l5], is0 = prm[TP::IS0 + l5];
  const float sh0 = prm[TP::BE0 + l5] - mu0 * a0;
  float w1[64];
  if (MODE != MODE_GRAM1) {
#pragma unroll
    for (int k = 0; k < 64; k++) w1[k] = prm[TP::W1 + l * 64 + k];
  }
  const float mu1 = prm[TP::MU1 + l], is1 = prm[TP::IS1 + l], a1 = prm[TP::GA1 + l] * is1;
  const float sh1 = prm[TP::BE1 + l] - mu1 * a1;
  float w1t[64];  // column l of W1 (pass E: du = W1^T dx1)
  float m1 = 0.f, m2 = 0.f;
  if (MODE == MODE_BWD0) {
#pragma unroll
    for (int k = 0; k < 64; k++) w1t[k] = prm[TP::W1 + k * 64 + l];
    m1 = prm[TP::M1 + l];
    m2 = prm[TP::M2 + l];
  }
  float acc[64];
#pragma unroll
  for (int k = 0; k < 64; k++) acc[k] = 0.f;
  float s1 = 0.f, s2 = 0.f;

  auto layer0 = [&](const uint32_t* rp, float& x0) {  // lane l: channel l & 31
    float x = 0.f;
#pragma unroll
    for (int k = 0; k < C0; k++) x = __builtin_fmaf(w0[k], rec_f(rp, k), x);
    x0 = x;
    const float y = __builtin_fmaf(x, a0, sh0);
    return y > 0.f ? y : 0.f;
  };
  auto layer1 = [&](float& x1) {  // su holds u
    float x = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 16; k4++) {
      const float4 uu = *reinterpret_cast<const float4*>(su + 4 * k4);
      x = __builtin_fmaf(w1[4 * k4 + 0], uu.x, x);
      x = __builtin_fmaf(w1[4 * k4 + 1], uu.y, x);
      x = __builtin_fmaf(w1[4 * k4 + 2], uu.z, x);
      x = __builtin_fmaf(w1[4 * k4 + 3], uu.w, x);
    }
    x1 = x;
    const float y = __builtin_fmaf(x, a1, sh1);
    return y > 0.f ? y : 0.f;
  };

  for (int r = r0; r < r1; r++) {
    const uint32_t first = pfirst[r], c = pcnt[r];
    // sweep 1: the pillar's layer-0 maximum
    float g0 = 0.f;
    for (uint32_t t = 0; t < c; t++) {
      float x0;
      g0 = fmaxf(g0, layer0(rec + (size_t)(first + t) * 16, x0));
    }
    if (MODE == MODE_GRAM1) {
      for (uint32_t t = 0; t < c; t++) {
        float x0;
        const float h0 = layer0(rec + (size_t)(first + t) * 16, x0);
        const float u = l < 32 ? h0 : g0;
        __builtin_amdgcn_wave_barrier();
        su[l] = u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k4 = 0; k4 < 16; k4++) {
          const float4 uu = *reinterpret_cast<const float4*>(su + 4 * k4);
          acc[4 * k4 + 0] = __builtin_fmaf(u, uu.x, acc[4 * k4 + 0]);
          acc[4 * k4 + 1] = __builtin_fmaf(u, uu.y, acc[4 * k4 + 1]);
          acc[4 * k4 + 2] = __builtin_fmaf(u, uu.z, acc[4 * k4 + 2]);
          acc[4 * k4 + 3] = __builtin_fmaf(u, uu.w, acc[4 * k4 + 3]);
        }
        s1 += u;
      }
      continue;
    }
    // sweep 2: layer 1 per point
    float omax = 0.f;
    const float gp = (MODE == MODE_BWD1 || MODE == MODE_BWD0) ? G[(int64_t)r * 64 + l] : 0.f;
    const float op = (MODE == MODE_BWD1 || MODE == MODE_BWD0) ? out_saved[(int64_t)r * 64 + l] : 0.f;
    bool taken = false;
    float dg0 = 0.f;  // pass E, lanes >= 32: du summed over the pillar
    for (uint32_t t = 0; t < c; t++) {
      float x0, x1;
      const float h0 = layer0(rec + (size_t)(first + t) * 16, x0);
      const float u = l < 32 ? h0 : g0;
      __builtin_amdgcn_wave_barrier();
      su[l] = u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float h1 = layer1(x1);
      if (MODE == MODE_OUT) {
        omax = fmaxf(omax, h1);
        continue;
      }
      // the gradient of the pillar maximum goes to its first row that attains it (and only through an open ReLU)
      const bool isarg = !taken && h1 == op && h1 > 0.f;
      taken = taken || isarg;
      const float dz = isarg ? gp : 0.f;
      const float xh = (x1 - mu1) * is1;
      if (MODE == MODE_BWD1) {
        s1 += dz;
        s2 = __builtin_fmaf(dz, xh, s2);
        if (__ballot(dz != 0.f) != 0) {
#pragma unroll
          for (int k4 = 0; k4 < 16; k4++) {
            const float4 uu = *reinterpret_cast<const float4*>(su + 4 * k4);
            acc[4 * k4 + 0] = __builtin_fmaf(dz, uu.x, acc[4 * k4 + 0]);
            acc[4 * k4 + 1] = __builtin_fmaf(dz, uu.y, acc[4 * k4 + 1]);
            acc[4 * k4 + 2] = __builtin_fmaf(dz, uu.z, acc[4 * k4 + 2]);
            acc[4 * k4 + 3] = __builtin_fmaf(dz, uu.w, acc[4 * k4 + 3]);
          }
        }
      } else {  // MODE_BWD0, first of two sweeps: the pillar's dg0
        const float dx1 = a1 * (dz - m1 - xh * m2);
        __builtin_amdgcn_wave_barrier();
        sd[l] = dx1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float du = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < 16; k4++) {
          const float4 dd = *reinterpret_cast<const float4*>(sd + 4 * k4);
          du = __builtin_fmaf(w1t[4 * k4 + 0], dd.x, du);
          du = __builtin_fmaf(w1t[4 * k4 + 1], dd.y, du);
          du = __builtin_fmaf(w1t[4 * k4 + 2], dd.z, du);
          du = __builtin_fmaf(w1t[4 * k4 + 3], dd.w, du);
        }
        if (l >= 32) dg0 += du;
      }
    }
    if (MODE == MODE_OUT) {
      if ((int64_t)r < out_rows) out[(int64_t)r * 64 + l] = omax;
      continue;
    }
    if (MODE == MODE_BWD0) {
      // second sweep: dh0 = du[:32] + dg0 routed to the argmax row of g0; dz0, its statistics and dz0^T f
      const float dgl = __shfl(dg0, l5 + 32);  // lane l < 32 takes channel l's pillar sum from lane l + 32
      bool taken0 = false;
      taken = false;
      for (uint32_t t = 0; t < c; t++) {
        const uint32_t* rp = rec + (size_t)(first + t) * 16;
        float x0, x1;
        const float h0 = layer0(rp, x0);
        const float u = l < 32 ? h0 : g0;
        __builtin_amdgcn_wave_barrier();
        su[l] = u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float h1 = layer1(x1);
        const bool isarg = !taken && h1 == op && h1 > 0.f;
        taken = taken || isarg;
        const float dz = isarg ? gp : 0.f;
        const float xh = (x1 - mu1) * is1;
        const float dx1 = a1 * (dz - m1 - xh * m2);
        __builtin_amdgcn_wave_barrier();
        sd[l] = dx1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float du = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < 16; k4++) {
          const float4 dd = *reinterpret_cast<const float4*>(sd + 4 * k4);
          du = __builtin_fmaf(w1t[4 * k4 + 0], dd.x, du);
          du = __builtin_fmaf(w1t[4 * k4 + 1], dd.y, du);
          du = __builtin_fmaf(w1t[4 * k4 + 2], dd.z, du);
          du = __builtin_fmaf(w1t[4 * k4 + 3], dd.w, du);
        }
        // lanes < 32 own layer-0 channel l (lanes >= 32 mirror it and are not written out)
        const float dud = __shfl(du, l5);  // du of channel l & 31 (direct part)
        const bool isarg0 = !taken0 && h0 == g0 && h0 > 0.f;
        taken0 = taken0 || isarg0;
        const float dh0 = dud + (isarg0 ? dgl : 0.f);
        const float dz0 = h0 > 0.f ? dh0 : 0.f;
        s1 += dz0;
        s2 = __builtin_fmaf(dz0, (x0 - mu0) * is0, s2);
#pragma unroll
        for (int k = 0; k < C0; k++) acc[k] = __builtin_fmaf(dz0, rec_f(rp, k), acc[k]);
      }
    }
  }
  // per-wave partials
  if (MODE == MODE_GRAM1) {
    float* o = part + ((int64_t)w * 64 + l) * 65;
#pragma unroll
    for (int k = 0; k < 64; k++) o[k] = acc[k];
    o[64] = s1;
  } else if (MODE == MODE_BWD1) {
    float* o = part + ((int64_t)w * 64 + l) * 66;
#pragma unroll
    for (int k = 0; k < 64; k++) o[k] = acc[k];
    o[64] = s1;
    o[65] = s2;
  } else if (MODE == MODE_BWD0) {
    if (l < 32) {
      float* o = part + ((int64_t)w * 32 + l) * (C0 + 2);
#pragma unroll
      for (int k = 0; k < C0; k++) o[k] = acc[k];
      o[C0] = s1;
      o[C0 + 1] = s2;
    }
  }
}

// pass A: Gram matrix of the decorated features, thread per point.  Partials per thread block: [C0 + C0*C0] floats.
template <int C0>
__global__ __launch_bounds__(256) void k_pfn_gram0(const uint32_t* __restrict__ rec, const int32_t* __restrict__ counters, float* __restrict__ part) {
  __shared__ float s_red[4][C0 + C0 * C0];
  const int n = counters[1];
  float f1[C0], f2[C0 * (C0 + 1) / 2];
#pragma unroll
  for (int k = 0; k < C0; k++) f1[k] = 0.f;
#pragma unroll
  for (int k = 0; k < C0 * (C0 + 1) / 2; k++) f2[k] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint4* q = reinterpret_cast<const uint4*>(rec + i * 16);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    const uint32_t wds[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    float f[C0];
#pragma unroll
    for (int k = 0; k < C0; k++) f[k] = __uint_as_float(wds[8 * (k & 1) + (k >> 1)]);
    int idx = 0;
#pragma unroll
    for (int j = 0; j < C0; j++) {
      f1[j] += f[j];
#pragma unroll
      for (int k = j; k < C0; k++) {
        f2[idx] = __builtin_fmaf(f[j], f[k], f2[idx]);
        idx++;
      }
    }
  }
  // wave reduction, then the block's four waves through LDS
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  auto wsum = [&](float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
  };
#pragma unroll
  for (int j = 0; j < C0; j++) {
    const float v = wsum(f1[j]);
    if (lane == 0) s_red[wv][j] = v;
  }
  {
    int idx = 0;
#pragma unroll
    for (int j = 0; j < C0; j++)
#pragma unroll
      for (int k = j; k < C0; k++) {
        const float v = wsum(f2[idx++]);
        if (lane == 0) {
          s_red[wv][C0 + j * C0 + k] = v;
          s_red[wv][C0 + k * C0 + j] = v;
        }
      }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < C0 + C0 * C0; k += 256)
    part[(int64_t)blockIdx.x * (C0 + C0 * C0) + k] = s_red[0][k] + s_red[1][k] + s_red[2][k] + s_red[3][k];
}

constexpr int kTrainBlocks = 512;  // two blocks per CU (two waves per SIMD hide the LDS round trips); the per-wave partial sums scale with this

template <int C0>
int launch_train(int mode, const uint32_t* rec, const uint32_t* pfirst, const uint32_t* pcnt, const int32_t* counters, const float* prm,
                 float* part, const float* G, const float* out_saved, float* out, int64_t out_rows, hipStream_t st) {
  switch (mode) {
    case 0: k_pfn_gram0<C0><<<kTrainBlocks, 256, 0, st>>>(rec, counters, part); break;
    case MODE_GRAM1: k_pfn_train<C0, MODE_GRAM1><<<kTrainBlocks, 256, 0, st>>>(rec, pfirst, pcnt, counters, prm, part, G, out_saved, out, out_rows); break;
    case MODE_OUT: k_pfn_train<C0, MODE_OUT><<<kTrainBlocks, 256, 0, st>>>(rec, pfirst, pcnt, counters, prm, part, G, out_saved, out, out_rows); break;
    case MODE_BWD1: k_pfn_train<C0, MODE_BWD1><<<kTrainBlocks, 256, 0, st>>>(rec, pfirst, pcnt, counters, prm, part, G, out_saved, out, out_rows); break;
    case MODE_BWD0: k_pfn_train<C0, MODE_BWD0><<<kTrainBlocks, 256, 0, st>>>(rec, pfirst, pcnt, counters, prm, part, G, out_saved, out, out_rows); break;
    default: pnx_set_error("pfn train: bad pass %d", mode); return PNX_ERR_INVALID;
  }
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // namespace

// pass: 0 gram0, 1 gram1, 2 output, 3 backward-1, 4 backward-0 (see the header of this file)
int pnx_launch_pfn_train(int F, int pass, const uint32_t* rec64, const uint32_t* pfirst, const uint32_t* pcnt, const int32_t* counters,
                         const float* prm, float* part, const float* G, const float* out_saved, float* out, int64_t out_rows, hipStream_t st) {
  switch (F) {
    case 3: return launch_train<8>(pass, rec64, pfirst, pcnt, counters, prm, part, G, out_saved, out, out_rows, st);
    case 4: return launch_train<9>(pass, rec64, pfirst, pcnt, counters, prm, part, G, out_saved, out, out_rows, st);
    case 5: return launch_train<10>(pass, rec64, pfirst, pcnt, counters, prm, part, G, out_saved, out, out_rows, st);
    case 6: return launch_train<11>(pass, rec64, pfirst, pcnt, counters, prm, part, G, out_saved, out, out_rows, st);
  }
  pnx_set_error("num_point_features %d not in 3..6", F);
  return PNX_ERR_UNSUPPORTED;
}

int pnx_pfn_train_blocks(void) { return kTrainBlocks; }
