// pfn_mfma.hip -- the PFN (pillar_encoder.py:35-50 x2, :174-182) as a single-pass fp32-MFMA kernel.
//
// Input: the per-slot records built by reader.hip's k_fill.  Points of one pillar occupy contiguous "slots", pillars are
// in torch.unique order; a record is 32 bytes: [x y z f4 f5 f6 | idx,rem | rank] where idx = position inside the pillar
// and rem = points still to come, so heads (idx==0), tails (rem==0) and pillar sizes need no other array.
//
// One wave = one tile of <= 32 points at a time, cut at pillar boundaries (a pillar never straddles tiles; pillars with
// more than 32 points take the k_pfn_big path).  Two lanes share a point: lane = (point, h), h = lane>>5 selecting the
// even/odd K element -- exactly the operand layout of v_mfma_f32_32x32x2_f32 (an exact fp32 fmaf chain at the fp32
// vector rate, MI355X_MICROARCH.md):  A[i = lane&31][k = lane>>5],  B[k = lane>>5][j = lane&31],
// D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
//
//   layer 0   D0 = W0'(32ch x K) * F^T(K x points)   (bias rides on a constant-1 K column): a lane now holds 16 CHANNELS of
//             its point -- which is precisely the B fragment of the next GEMM if K is visited in accumulator-register order
//             (channel (i&3)+8*(i>>2)+4h at step i).  Layer 0 feeds layer 1 with no transpose and no LDS.
//   max       per-pillar max = segmented scan ACROSS LANES with DPP row shifts (row_shr:1,2,4,8 + row_bcast:15), the segment
//             test is `idx >= d`; steps no pillar of the tile needs are skipped by a ballot.  The pillar total is fetched from
//             the tail lane with one ds_bpermute per register (the "max" half of the concat, pe:44,49).
//   layer 1   D1 = W1'(64ch x 64) * U^T: 64 MFMAs, scan again; relu(max(x) + s) == max(relu(x + s)) lets bias and ReLU run on
//             tail lanes only, which store their 32 channels as eight 16-byte pieces of the pillar's feat_max row.
// No LDS, no atomics (except the rare big-pillar list), no cross-wave traffic, deterministic.  BN is pre-folded (k_fold_bn).
#include <vector>

#include "pnx_common.h"
#include "pnx_dppscan.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

#define PNX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ double seg_sum(double v, int idx, int col, const ScanPlan& pl) {
  if (pl.s1) { const double t = dpp_d<DPP_ROW_SHR1, 0xF>(v); v += idx >= 1 ? t : 0.0; }
  if (pl.s2) { const double t = dpp_d<DPP_ROW_SHR2, 0xF>(v); v += idx >= 2 ? t : 0.0; }
  if (pl.s4) { const double t = dpp_d<DPP_ROW_SHR4, 0xF>(v); v += idx >= 4 ? t : 0.0; }
  if (pl.s8) { const double t = dpp_d<DPP_ROW_SHR8, 0xF>(v); v += idx >= 8 ? t : 0.0; }
  if (pl.s1) {
    const double t = dpp_d<DPP_ROW_BCAST15, 0xA>(v);
    v += idx > (col & 15) ? t : 0.0;
  }
  return v;
}
__device__ __forceinline__ float from_lane(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ double from_lane_d(double v, int src_lane) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b & 0xffffffffLL));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Where the reader writes a pillar's 64 features: an fp32 feat_max row and/or the pillar's cell of the dense NHWC canvas.
struct PfnOut {
  float* g1;               // (rows, 64) fp32 or null
  int64_t g1_rows;
  void* canvas;            // NHWC canvas or null
  const int32_t* cell;     // linear cell index (b*gy + yi)*gx + xi of every pillar
  int dt;                  // PNX_F32 / PNX_BF16 / PNX_F16
};
__device__ __forceinline__ uint32_t bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);  // inputs are finite post-ReLU values
  return u >> 16;
}
__device__ __forceinline__ uint32_t f16_rne(float f) {
  const _Float16 hv = (_Float16)f;
  return (uint32_t)__builtin_bit_cast(unsigned short, hv);
}
// four consecutive channels starting at chan0 (multiple of 4) of pillar `r`
__device__ __forceinline__ void store_piece(const PfnOut& o, int r, int64_t cell, int chan0, float4 v) {
  if (o.g1 != nullptr && (int64_t)r < o.g1_rows) *reinterpret_cast<float4*>(o.g1 + (int64_t)r * 64 + chan0) = v;
  if (o.canvas != nullptr) {
    if (o.dt == PNX_F32) {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(o.canvas) + cell * 64 + chan0) = v;
    } else {
      uint2 p;
      if (o.dt == PNX_BF16) {
        p.x = bf16_rne(v.x) | (bf16_rne(v.y) << 16);
        p.y = bf16_rne(v.z) | (bf16_rne(v.w) << 16);
      } else {
        p.x = f16_rne(v.x) | (f16_rne(v.y) << 16);
        p.y = f16_rne(v.z) | (f16_rne(v.w) << 16);
      }
      *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(o.canvas) + cell * 64 + chan0) = p;
    }
  }
}

__device__ __forceinline__ uint32_t pstart(int32_t r, const uint32_t* __restrict__ cpre, const uint32_t* __restrict__ cblk) {
  return cblk[r >> PNX_SCAN_SHIFT] + cpre[r];
}

struct Rec {
  uint4 a, b;
};
__device__ __forceinline__ Rec load_rec(const uint32_t* __restrict__ rec, uint32_t slot) {
  const uint4* p = reinterpret_cast<const uint4*>(rec + (int64_t)slot * 8);
  Rec r;
  r.a = p[0];
  r.b = p[1];
  return r;
}

// Decorated point features (pe:116-123): [raw F | xyz - mean | xy - pillar centre | 1 (bias column) | 0].
template <int F>
__device__ __forceinline__ void decorate_rec(const Rec& q, float mx, float my, float mz, const PnxGeomDev& g, float* f) {
  const float w[6] = {__uint_as_float(q.a.x), __uint_as_float(q.a.y), __uint_as_float(q.a.z),
                      __uint_as_float(q.a.w), __uint_as_float(q.b.x), __uint_as_float(q.b.y)};
#pragma unroll
  for (int k = 0; k < F; k++) f[k] = w[k];
  const float x = w[0], y = w[1], z = w[2];
  f[F + 0] = __fsub_rn(x, mx);
  f[F + 1] = __fsub_rn(y, my);
  f[F + 2] = __fsub_rn(z, mz);
  const float cx = __fdiv_rn(__fsub_rn(x, g.minx), g.vx);
  const float cy = __fdiv_rn(__fsub_rn(y, g.miny), g.vy);
  const float xi = (float)(int)cx, yi = (float)(int)cy;
  const float ctrx = __fadd_rn(__fadd_rn(__fmul_rn(xi, g.vx), __fdiv_rn(g.vx, 2.0f)), g.minx);
  const float ctry = __fadd_rn(__fadd_rn(__fmul_rn(yi, g.vy), __fdiv_rn(g.vy, 2.0f)), g.miny);
  f[F + 3] = __fsub_rn(x, ctrx);
  f[F + 4] = __fsub_rn(y, ctry);
  f[F + 5] = 1.f;
  f[F + 6] = 0.f;
}

// End of the pillar that contains `slot` (= first slot of the next pillar), from the record's idx|rem word; falls back to
// the prefix arrays when the 16-bit fields are saturated (pillars with >= 65535 points).
__device__ __forceinline__ uint32_t pillar_end_at(const uint32_t* __restrict__ rec, int64_t slot, const uint32_t* __restrict__ count,
                                                  const uint32_t* __restrict__ cpre, const uint32_t* __restrict__ cblk, bool* is_head) {
  const uint4 w = *reinterpret_cast<const uint4*>(rec + slot * 8 + 4);  // words 4..7
  const uint32_t idx = w.z & 0xFFFFu, rem = w.z >> 16;
  *is_head = idx == 0;
  if (idx < 0xFFFFu && rem < 0xFFFFu) return (uint32_t)slot + rem + 1u;
  return pstart((int)w.w, cpre, cblk) + count[w.w] + 1u;
}

// A pillar with more than 32 points, processed by the wave that meets it: three sweeps over its tiles (mean; layer-0 max;
// layer 1 + max) with per-lane running maxima, then one all-lane reduction per register.  Same MFMA fragments as the
// main path; rare at PillarNeXt-B resolution, common only for coarse voxels.
template <int F, int KS>
__device__ __forceinline__ void big_pillar(const uint32_t* __restrict__ rec, const PnxGeomDev& g, uint32_t st, uint32_t c, int r,
                                           const float* w0f, const float* w1a, const float* w1b, const float4* __restrict__ s1lane,
                                           const PfnOut& out, int l) {
  constexpr int C0 = F + 5;
  const int col = l & 31, h = l >> 5;
  const float NI = -__builtin_inff();
  double sx = 0, sy = 0, sz = 0;
  for (uint32_t t = col; t < c; t += 32) {
    const uint4 a = *reinterpret_cast<const uint4*>(rec + (int64_t)(st + t) * 8);
    sx += (double)__uint_as_float(a.x);
    sy += (double)__uint_as_float(a.y);
    sz += (double)__uint_as_float(a.z);
  }
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) {  // both halves hold the same points: reduce inside each half
    sx += __shfl_xor(sx, d);
    sy += __shfl_xor(sy, d);
    sz += __shfl_xor(sz, d);
  }
  const float fc = (float)c;
  const float mx = __fdiv_rn((float)sx, fc), my = __fdiv_rn((float)sy, fc), mz = __fdiv_rn((float)sz, fc);
  float g0[16];
#pragma unroll
  for (int i = 0; i < 16; i++) g0[i] = NI;
  for (uint32_t t0 = 0; t0 < c; t0 += 32) {
    const bool act = t0 + col < c;
    const Rec cur = load_rec(rec, st + min(t0 + (uint32_t)col, c - 1));
    float f[C0 + 2], ff[KS];
    decorate_rec<F>(cur, mx, my, mz, g, f);
#pragma unroll
    for (int kk = 0; kk < KS; kk++) ff[kk] = act ? (h ? f[2 * kk + 1] : f[2 * kk]) : 0.f;
    v16f d0;
#pragma unroll
    for (int i = 0; i < 16; i++) d0[i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS; kk++) d0 = PNX_MFMA(w0f[kk], ff[kk], d0);
#pragma unroll
    for (int i = 0; i < 16; i++) g0[i] = fmaxf(g0[i], act ? d0[i] : NI);
  }
#pragma unroll
  for (int i = 0; i < 16; i++) {
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) g0[i] = fmaxf(g0[i], __shfl_xor(g0[i], d));
    g0[i] = fmaxf(g0[i], 0.f);
  }
  float pa[16], pb[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    pa[i] = NI;
    pb[i] = NI;
  }
  for (uint32_t t0 = 0; t0 < c; t0 += 32) {
    const bool act = t0 + col < c;
    const Rec cur = load_rec(rec, st + min(t0 + (uint32_t)col, c - 1));
    float f[C0 + 2], ff[KS];
    decorate_rec<F>(cur, mx, my, mz, g, f);
#pragma unroll
    for (int kk = 0; kk < KS; kk++) ff[kk] = act ? (h ? f[2 * kk + 1] : f[2 * kk]) : 0.f;
    v16f d0, da, db;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      d0[i] = 0.f;
      da[i] = 0.f;
      db[i] = 0.f;
    }
#pragma unroll
    for (int kk = 0; kk < KS; kk++) d0 = PNX_MFMA(w0f[kk], ff[kk], d0);
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const float u = fmaxf(d0[i], 0.f);
      da = PNX_MFMA(w1a[i], u, da);
      db = PNX_MFMA(w1b[i], u, db);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
      da = PNX_MFMA(w1a[16 + i], g0[i], da);
      db = PNX_MFMA(w1b[16 + i], g0[i], db);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
      pa[i] = fmaxf(pa[i], act ? da[i] : NI);
      pb[i] = fmaxf(pb[i], act ? db[i] : NI);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; i++) {
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) {
      pa[i] = fmaxf(pa[i], __shfl_xor(pa[i], d));
      pb[i] = fmaxf(pb[i], __shfl_xor(pb[i], d));
    }
  }
  if (col == 0) {
    const int64_t cell = out.canvas ? (int64_t)out.cell[r] : 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float4 sa = s1lane[j], sb = s1lane[4 + j];
      store_piece(out, r, cell, 8 * j + 4 * h, make_float4(fmaxf(pa[4 * j] + sa.x, 0.f), fmaxf(pa[4 * j + 1] + sa.y, 0.f),
                                                          fmaxf(pa[4 * j + 2] + sa.z, 0.f), fmaxf(pa[4 * j + 3] + sa.w, 0.f)));
      store_piece(out, r, cell, 32 + 8 * j + 4 * h, make_float4(fmaxf(pb[4 * j] + sb.x, 0.f), fmaxf(pb[4 * j + 1] + sb.y, 0.f),
                                                               fmaxf(pb[4 * j + 2] + sb.z, 0.f), fmaxf(pb[4 * j + 3] + sb.w, 0.f)));
    }
  }
}

// counters[3] = number of big pillars appended to biglist
template <int F, int R>
__global__ __launch_bounds__(256) void k_pfn_mfma(const uint32_t* __restrict__ rec, PnxGeomDev g, const uint32_t* __restrict__ count,
                                                 const uint32_t* __restrict__ cpre, const uint32_t* __restrict__ cblk,
                                                 int32_t* __restrict__ counters, int32_t* __restrict__ biglist, int bigcap,
                                                 const float* __restrict__ P, PfnOut out, unsigned long long* __restrict__ dbg) {
  constexpr int C0 = F + 5, KS = (C0 + 2) / 2;     // K = C0 features + one constant-1 column that carries the folded BN shift
  constexpr int FR = 32 * C0 + 32 + 64 * 64 + 64;  // start of the fragment-ordered block (k_fold_bn)
  const int l = threadIdx.x & 63, col = l & 31, h = l >> 5;
  const int n_kept = counters[1];
#ifdef PNX_PFN_TIMERS  // section timers (build with -DPNX_PFN_TIMERS; costs ~18 VGPRs)
  unsigned long long T[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tk = __builtin_amdgcn_s_memtime();
#define TOCK(k)                                                 \
  {                                                             \
    const unsigned long long _n = __builtin_amdgcn_s_memtime(); \
    T[k] += _n - tk;                                            \
    tk = _n;                                                    \
  }
#else
#define TOCK(k)
#endif

  // ---- weight fragments: coalesced loads, once per (persistent) wave
  const float* __restrict__ FP = P + FR + l;
  float w0f[KS];
#pragma unroll
  for (int kk = 0; kk < KS; kk++) w0f[kk] = FP[kk * 64];
  float w1a[32], w1b[32];
#pragma unroll
  for (int i = 0; i < 32; i++) {
    w1a[i] = FP[(23 + i) * 64];
    w1b[i] = FP[(55 + i) * 64];
  }
  const float4* __restrict__ s1lane = reinterpret_cast<const float4*>(P + FR + 64 * 89 + l * 32);  // s1 in this lane's channel order
  float s1a[16], s1b[16];  // folded-BN shift of layer 1 for this lane's 2 x 16 channels
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float4 sa = s1lane[j], sb = s1lane[4 + j];
    s1a[4 * j + 0] = sa.x, s1a[4 * j + 1] = sa.y, s1a[4 * j + 2] = sa.z, s1a[4 * j + 3] = sa.w;
    s1b[4 * j + 0] = sb.x, s1b[4 * j + 1] = sb.y, s1b[4 * j + 2] = sb.z, s1b[4 * j + 3] = sb.w;
  }

  TOCK(0);
  // results of the previous tile, stored one tile late so that the record prefetch never waits behind fresh stores
  float pa[16], pb[16];
  bool p_store = false;
  int p_rank = 0;
  auto flush = [&]() {
    if (p_store) {
      const int64_t cell = out.canvas ? (int64_t)out.cell[p_rank] : 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float4 oa = make_float4(pa[4 * j + 0], pa[4 * j + 1], pa[4 * j + 2], pa[4 * j + 3]);
        const float4 ob = make_float4(pb[4 * j + 0], pb[4 * j + 1], pb[4 * j + 2], pb[4 * j + 3]);
        store_piece(out, p_rank, cell, 8 * j + 4 * h, oa);
        store_piece(out, p_rank, cell, 32 + 8 * j + 4 * h, ob);
      }
    }
    p_store = false;
  };

  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int64_t pass = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);; pass += n_waves) {
    const int64_t slot0 = pass * R;
    if (slot0 >= n_kept) break;
    const int64_t slot1 = (slot0 + R < n_kept) ? slot0 + R : n_kept;
    // pillars owned by this pass = those whose first slot lies in [slot0, slot1)
    bool head0;
    const uint32_t e0 = pillar_end_at(rec, slot0, count, cpre, cblk, &head0);
    const uint32_t base = head0 ? (uint32_t)slot0 : e0;
    uint32_t end = (uint32_t)n_kept;
    if (slot1 < n_kept) {
      bool head1;
      const uint32_t e1 = pillar_end_at(rec, slot1, count, cpre, cblk, &head1);
      end = head1 ? (uint32_t)slot1 : e1;
    }
    TOCK(1);
    if (base >= end) continue;

    uint32_t ts = base;
    Rec nxt = load_rec(rec, min(ts + (uint32_t)col, end - 1));
    while (ts < end) {
      const Rec cur = nxt;
      const bool in_range = ts + (uint32_t)col < end;
      const int idx = (int)(cur.b.z & 0xFFFFu), rem = (int)(cur.b.z >> 16);
      const int r = (int)cur.b.w;
      const bool complete = in_range && (col + rem <= 31);
      const uint32_t V = (uint32_t)__ballot(complete && h == 0);
      const int nv = __builtin_popcount(V);
      if (nv == 0) {
        // the pillar at ts has more than 32 points: hand it to k_pfn_big (keeps this kernel's register budget) and step over it
        const int q = __builtin_amdgcn_readfirstlane(r);
        const uint32_t c = count[q] + 1u;
        if (l == 0) {
          const int at = atomicAdd(&counters[3], 1);
          if (at < bigcap) biglist[at] = q;
        }
        ts += c;
        if (ts < end) nxt = load_rec(rec, min(ts + (uint32_t)col, end - 1));
        continue;
      }
      const uint32_t ts_next = ts + (uint32_t)nv;
      nxt = load_rec(rec, min(ts_next + (uint32_t)col, end - 1));  // prefetch the next tile while this one computes
      flush();                                                      // previous tile's rows (issued behind the prefetch)
      const bool act = col < nv;
      const int cnt = idx + rem + 1;
      const int tail_lane = act ? l + rem : l;  // same half
      ScanPlan pl;
      pl.s1 = __ballot(act && idx >= 1) != 0;
      pl.s2 = __ballot(act && idx >= 2) != 0;
      pl.s4 = __ballot(act && idx >= 4) != 0;
      pl.s8 = __ballot(act && idx >= 8) != 0;
      TOCK(2);

      // ---- per-pillar mean of xyz (scatter_mean, pe:113-114): exact fp64 sum, fp32 divide
      float mx, my, mz;
      {
        double sx = act ? (double)__uint_as_float(cur.a.x) : 0.0;
        double sy = act ? (double)__uint_as_float(cur.a.y) : 0.0;
        double sz = act ? (double)__uint_as_float(cur.a.z) : 0.0;
        if (pl.s1) {
          sx = from_lane_d(seg_sum(sx, idx, col, pl), tail_lane);
          sy = from_lane_d(seg_sum(sy, idx, col, pl), tail_lane);
          sz = from_lane_d(seg_sum(sz, idx, col, pl), tail_lane);
        }
        const float fc = (float)cnt;
        mx = __fdiv_rn((float)sx, fc);
        my = __fdiv_rn((float)sy, fc);
        mz = __fdiv_rn((float)sz, fc);
      }
      TOCK(3);
      // ---- layer 0 (lane = point, registers = channels)
      float ff[KS];
      {
        float f[C0 + 2];
        decorate_rec<F>(cur, mx, my, mz, g, f);
#pragma unroll
        for (int kk = 0; kk < KS; kk++) ff[kk] = act ? (h ? f[2 * kk + 1] : f[2 * kk]) : 0.f;
      }
      v16f d0;
#pragma unroll
      for (int i = 0; i < 16; i++) d0[i] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KS; kk++) d0 = PNX_MFMA(w0f[kk], ff[kk], d0);
      // ---- "max" half of the concat: per-pillar max of layer 0, delivered to every point of the pillar
      float g0[16];
#pragma unroll
      for (int i = 0; i < 16; i++) g0[i] = fmaxf(d0[i], 0.f);  // ReLU first: max(relu(x)) == relu(max(x)), and the scan may use 0 as identity
      if (pl.s1) {
        seg_max_nn16(g0, idx, col, pl);
#pragma unroll
        for (int i = 0; i < 16; i++) g0[i] = from_lane(g0[i], tail_lane);
      }
      TOCK(4);
      // ---- layer 1: 64 output channels as two 32-row tiles, K in accumulator-register order
      v16f da, db;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        da[i] = 0.f;
        db[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const float u = fmaxf(d0[i], 0.f);
        da = PNX_MFMA(w1a[i], u, da);
        db = PNX_MFMA(w1b[i], u, db);
      }
#pragma unroll
      for (int i = 0; i < 16; i++) {
        da = PNX_MFMA(w1a[16 + i], g0[i], da);
        db = PNX_MFMA(w1b[16 + i], g0[i], db);
      }
      TOCK(5);
      // ---- per-pillar max of layer 1.  Shift + ReLU first (max(relu(x + s)) == relu(max(x) + s), x -> relu(x + s) is monotone), so
      // the scan runs on non-negative values; the stores happen in flush(), one tile later
#pragma unroll
      for (int i = 0; i < 16; i++) {
        pa[i] = fmaxf(da[i] + s1a[i], 0.f);
        pb[i] = fmaxf(db[i] + s1b[i], 0.f);
      }
      seg_max_nn16(pa, idx, col, pl);
      seg_max_nn16(pb, idx, col, pl);
      p_store = act && rem == 0;
      p_rank = r;
      ts = ts_next;
      TOCK(6);
    }
  }
  flush();
#ifdef PNX_PFN_TIMERS
  if (dbg && l == 0) {
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
#pragma unroll
    for (int k = 0; k < 8; k++) dbg[gw * 8 + k] = T[k];
  }
#endif
}

// Pillars with more than 32 points: one wave per pillar (see big_pillar).
template <int F>
__global__ __launch_bounds__(256) void k_pfn_big(const uint32_t* __restrict__ rec, PnxGeomDev g, const uint32_t* __restrict__ count,
                                                const uint32_t* __restrict__ cpre, const uint32_t* __restrict__ cblk,
                                                const int32_t* __restrict__ counters, const int32_t* __restrict__ biglist, int bigcap,
                                                const float* __restrict__ P, PfnOut out) {
  constexpr int C0 = F + 5, KS = (C0 + 2) / 2;
  constexpr int FR = 32 * C0 + 32 + 64 * 64 + 64;
  const int l = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  int nbig = counters[3];
  if (nbig > bigcap) nbig = bigcap;
  if (wave >= nbig) return;
  const float* __restrict__ FP = P + FR + l;
  float w0f[KS], w1a[32], w1b[32];
#pragma unroll
  for (int kk = 0; kk < KS; kk++) w0f[kk] = FP[kk * 64];
#pragma unroll
  for (int i = 0; i < 32; i++) {
    w1a[i] = FP[(23 + i) * 64];
    w1b[i] = FP[(55 + i) * 64];
  }
  const float4* __restrict__ s1lane = reinterpret_cast<const float4*>(P + FR + 64 * 89 + l * 32);
  for (int b = wave; b < nbig; b += nwaves) {
    const int q = biglist[b];
    big_pillar<F, KS>(rec, g, pstart(q, cpre, cblk), count[q] + 1u, q, w0f, w1a, w1b, s1lane, out, l);
  }
}

template <int F>
int launch_f(int R, const uint32_t* rec, const PnxGeomDev& g, const uint32_t* count, const uint32_t* cpre, const uint32_t* cblk,
             int32_t* counters, int32_t* biglist, int64_t bigcap, const float* folded, const PfnOut& out, int64_t n, int max_blocks,
             hipStream_t st) {
  static unsigned long long* dbg = nullptr;
  static int dbg_calls = 0;
  const bool want_dbg = getenv("PNX_PFN_TIMING") != nullptr;
  if (want_dbg && !dbg) (void)hipMalloc(&dbg, 8192 * 4 * 8 * sizeof(unsigned long long));
  int64_t nb = ((n + R - 1) / R + 3) / 4;  // 4 waves per block; windows are handed out dynamically
  if (nb > max_blocks) nb = max_blocks;
  if (nb < 1) nb = 1;
  if (getenv("PNX_DEBUG")) {
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_pfn_mfma<F, 64>, 256, 0);
    fprintf(stderr, "[pnx] k_pfn_mfma<%d,%d>: occupancy API %d blocks (x4 waves)/CU, %lld blocks\n", F, R, occ, (long long)nb);
  }
  const int bc = (int)(bigcap > 0x7fffffff ? 0x7fffffff : bigcap);
  if (R == 128) k_pfn_mfma<F, 128><<<(int)nb, 256, 0, st>>>(rec, g, count, cpre, cblk, counters, biglist, bc, folded, out, want_dbg ? dbg : nullptr);
  else if (R == 32) k_pfn_mfma<F, 32><<<(int)nb, 256, 0, st>>>(rec, g, count, cpre, cblk, counters, biglist, bc, folded, out, want_dbg ? dbg : nullptr);
  else k_pfn_mfma<F, 64><<<(int)nb, 256, 0, st>>>(rec, g, count, cpre, cblk, counters, biglist, bc, folded, out, want_dbg ? dbg : nullptr);
  if (want_dbg && ++dbg_calls == 20) {
    std::vector<unsigned long long> hbuf((size_t)nb * 4 * 8);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(hbuf.data(), dbg, hbuf.size() * 8, hipMemcpyDeviceToHost);
    double acc[8] = {0};
    for (size_t w = 0; w < (size_t)nb * 4; w++)
      for (int k = 0; k < 8; k++) acc[k] += (double)hbuf[w * 8 + k];
    const char* names[8] = {"weight frags", "dequeue+ownership", "rec wait+plan", "mean", "layer0+g0 scan", "layer1 MFMA", "scan+store", ""};
    for (int k = 0; k < 7; k++) fprintf(stderr, "[pnx-timing] %-20s %10.0f ticks/wave\n", names[k], acc[k] / (nb * 4));
  }
  k_pfn_big<F><<<64, 256, 0, st>>>(rec, g, count, cpre, cblk, counters, biglist, bc, folded, out);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // namespace

int pnx_launch_pfn_mfma(int F, const uint32_t* rec, const PnxGeomDev& geom, const uint32_t* count, const uint32_t* cpre,
                        const uint32_t* cblk, int32_t* counters, int32_t* biglist, int64_t bigcap, const float* folded, float* g1,
                        int64_t g1_rows, void* canvas, const int32_t* cell_of_pillar, int canvas_dt, int64_t n_points, hipStream_t st) {
  PfnOut out;
  out.g1 = g1;
  out.g1_rows = g1_rows;
  out.canvas = canvas;
  out.cell = cell_of_pillar;
  out.dt = canvas_dt;
  const char* r_env = getenv("PNX_PFN_R");  // slots per window: 32 | 64 | 128
  const int R = r_env ? atoi(r_env) : 64;
  const char* b_env = getenv("PNX_PFN_BLOCKS");
  const int max_blocks = b_env ? atoi(b_env) : 512;  // 256 CUs x 2 blocks x 4 waves = 2 waves per SIMD
  switch (F) {
    case 3: return launch_f<3>(R, rec, geom, count, cpre, cblk, counters, biglist, bigcap, folded, out, n_points, max_blocks, st);
    case 4: return launch_f<4>(R, rec, geom, count, cpre, cblk, counters, biglist, bigcap, folded, out, n_points, max_blocks, st);
    case 5: return launch_f<5>(R, rec, geom, count, cpre, cblk, counters, biglist, bigcap, folded, out, n_points, max_blocks, st);
    case 6: return launch_f<6>(R, rec, geom, count, cpre, cblk, counters, biglist, bigcap, folded, out, n_points, max_blocks, st);
  }
  pnx_set_error("num_point_features %d not in 3..6", F);
  return PNX_ERR_UNSUPPORTED;
}
