// pfn_mfma.hip -- the PFN (pillar_encoder.py:35-50 x2, :174-182) as a wave-tiled fp32-MFMA kernel.
//
// Input: the per-slot records built by reader.hip's k_fill (points of one pillar are contiguous "slots", pillars in
// torch.unique order; 32 bytes per slot = point row + pillar rank, so every load here is coalesced).  One wave owns the pillars whose first slot lies in its window of R slots
// and walks their points in tiles of 32 (the M/N size of v_mfma_f32_32x32x2_f32, which is an exact fp32
// fmaf chain at the fp32 vector rate -- MI355X_MICROARCH.md).  Two lanes share a point: lane = (point, h)
// with h = lane>>5 picking the even/odd K element, which is exactly the A/B fragment layout
//   A[i = lane&31][k = lane>>5],  B[k = lane>>5][j = lane&31],  D: col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5).
//
// Phase 1 (per tile): H0^T-orientation  D = F(points x K) * W0'(K x 32ch): a lane holds ONE channel for 16
//   points; one cross-half exchange gives it all 32 points, and the per-pillar max is a sequential in-register
//   scan driven by two wave-uniform bit masks (ballots: "first point of a pillar", "valid point").  Pillars may
//   straddle tiles: the running max simply carries over.  Maxima go to LDS G0[pillar][32].
// Phase 2 (per tile): H0-orientation  D = W0'(32ch x K) * F^T(K x points): now a lane holds 16 CHANNELS of one
//   point -- precisely the A fragment of the next GEMM if K is visited in the order the accumulator registers
//   are laid out (channel (i&3)+8*(i>>2)+4h at step i).  So layer 0 feeds layer 1 with no transpose; the
//   "max" half of the concat comes from G0 with four ds_read_b128.  64 MFMAs give H1 for 64 channels; after
//   one exchange lane l holds channel l for all 32 points, and the same sequential scan emits one coalesced
//   256-byte feat_max row per pillar.
// No atomics, no cross-wave traffic, deterministic.  BN is pre-folded (reader.hip k_fold_bn).
#include <vector>

#include "pnx_common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t pstart(int32_t r, const uint32_t* __restrict__ cpre, const uint32_t* __restrict__ cblk) {
  return cblk[r >> PNX_SCAN_SHIFT] + cpre[r];
}

template <int F>
__device__ __forceinline__ void decorate_pt(const float* __restrict__ p, float mx, float my, float mz, const PnxGeomDev& g, float* f) {
#pragma unroll
  for (int k = 0; k < F; k++) f[k] = p[1 + k];
  const float x = p[1], y = p[2], z = p[3];
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
}

// LDS traffic of one wave is executed in order; this only stops the compiler from moving accesses across the point.
#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)

#define PNX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// One 32-byte record per CSR slot: words 0..F-1 = x, y, z, f.. ; word 7 = pillar rank (written by k_fill).
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

template <int F>
__device__ __forceinline__ void decorate_rec(const Rec& q, float mx, float my, float mz, const PnxGeomDev& g, float* f) {
  const float w[7] = {__uint_as_float(q.a.x), __uint_as_float(q.a.y), __uint_as_float(q.a.z), __uint_as_float(q.a.w),
                      __uint_as_float(q.b.x), __uint_as_float(q.b.y), __uint_as_float(q.b.z)};
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
}

template <int F, int R>
__global__ __launch_bounds__(256) void k_pfn_mfma(const uint32_t* __restrict__ rec, PnxGeomDev g, const uint32_t* __restrict__ count,
                                                 const uint32_t* __restrict__ cpre, const uint32_t* __restrict__ cblk,
                                                 const int32_t* __restrict__ counters, const float* __restrict__ P,
                                                 float* __restrict__ g1, int64_t g1_rows, unsigned long long* __restrict__ dbg) {
  constexpr int C0 = F + 5, KS = (C0 + 2) / 2;  // K = C0 features + one constant-1 column that carries the folded BN shift
  constexpr int FR = 32 * C0 + 32 + 64 * 64 + 64;  // start of the fragment-ordered block (k_fold_bn)
  constexpr int GST = 36;                          // G0 row stride in floats: 16-byte aligned rows, banks spread
  // 4 independent waves per workgroup (one per SIMD by construction); each wave owns a private LDS slice and never
  // synchronises with the others (they run different trip counts), so only wave-level ordering is used.
  __shared__ __attribute__((aligned(16))) float sG0_all[4][R * GST];
  __shared__ float sMean_all[4][R * 3];
  const int wv = threadIdx.x >> 6;
  float* sG0 = sG0_all[wv];
  float* sMean = sMean_all[wv];

  const int l = threadIdx.x & 63, col = l & 31, h = l >> 5;
  unsigned long long T[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tk = 0;
#define TICK() (dbg ? __builtin_amdgcn_s_memtime() : 0ULL)
#define TOCK(k)                          \
  if (dbg) {                             \
    const unsigned long long _n = __builtin_amdgcn_s_memtime(); \
    T[k] += _n - tk;                     \
    tk = _n;                             \
  }
  const int n_kept = counters[1], Ptot = counters[0];

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
  const float s1a = FP[87 * 64], s1b = FP[88 * 64];

  for (int64_t slot0 = (int64_t)(blockIdx.x * 4 + wv) * R; slot0 < n_kept; slot0 += (int64_t)gridDim.x * 4 * R) {
    const int64_t slot1 = (slot0 + R < n_kept) ? slot0 + R : n_kept;
    tk = TICK();
    // pillars owned by this pass = those whose first slot is in [slot0, slot1)
    const int q0 = (int)rec[slot0 * 8 + 7];
    const uint32_t st0 = pstart(q0, cpre, cblk);
    const int p_lo = (st0 == (uint32_t)slot0) ? q0 : q0 + 1;
    const uint32_t base = (st0 == (uint32_t)slot0) ? st0 : st0 + count[q0] + 1u;
    int p_hi;
    uint32_t end;
    if (slot1 >= n_kept) {
      p_hi = Ptot;
      end = (uint32_t)n_kept;
    } else {
      const int q1 = (int)rec[slot1 * 8 + 7];
      const uint32_t st1 = pstart(q1, cpre, cblk);
      if (st1 == (uint32_t)slot1) {
        p_hi = q1;
        end = st1;
      } else {
        p_hi = q1 + 1;
        end = st1 + count[q1] + 1u;
      }
    }
    if (p_lo >= p_hi) continue;
    const int nown = p_hi - p_lo;  // <= R
    TOCK(0);
    WAVE_SYNC();                   // previous pass done with sG0 / sMean

    // ---- phase 0: per-pillar mean of xyz (scatter_mean, pe:113-114): fp64 sum, fp32 divide
    for (int s = l; s < nown; s += 64) {
      const int q = p_lo + s;
      const uint32_t st = pstart(q, cpre, cblk), c = count[q] + 1u;
      double sx = 0, sy = 0, sz = 0;
      for (uint32_t k = 0; k < c; k++) {
        const uint4 a = *reinterpret_cast<const uint4*>(rec + (int64_t)(st + k) * 8);
        sx += (double)__uint_as_float(a.x);
        sy += (double)__uint_as_float(a.y);
        sz += (double)__uint_as_float(a.z);
      }
      const float fc = (float)c;
      sMean[s * 3 + 0] = __fdiv_rn((float)sx, fc);
      sMean[s * 3 + 1] = __fdiv_rn((float)sy, fc);
      sMean[s * 3 + 2] = __fdiv_rn((float)sz, fc);
    }
    WAVE_SYNC();

    const int ntiles = (int)((end - base + 31) >> 5);
    TOCK(1);

    // ---- phase 1: layer 0, per-pillar max -> G0
    {
      float m = 0.f;
      int seg = 0;
      int prev_rank = p_lo - 1;  // rank of the slot before the tile (wave-uniform)
      Rec cur = load_rec(rec, min(base + (uint32_t)col, end - 1));
      for (int t = 0; t < ntiles; t++) {
        const uint32_t slot0_of_tile = base + 32u * t;
        const uint32_t slot = slot0_of_tile + col;
        const bool act = slot < end;
        const Rec nxt = load_rec(rec, min(slot + 32u, end - 1));  // prefetch the next tile's record
        const int r = (int)cur.b.w;
        int rp = __shfl_up(r, 1);
        if (col == 0) rp = prev_rank;
        const bool is_head = act && (r != rp);
        float ff[KS];
        {
          const int sl = act ? r - p_lo : 0;
          float f[C0 + 2];
          decorate_rec<F>(cur, sMean[sl * 3], sMean[sl * 3 + 1], sMean[sl * 3 + 2], g, f);
          f[C0] = 1.f;
          f[C0 + 1] = 0.f;
#pragma unroll
          for (int kk = 0; kk < KS; kk++) ff[kk] = act ? (h ? f[2 * kk + 1] : f[2 * kk]) : 0.f;
        }
        v16f acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; kk++) acc = PNX_MFMA(ff[kk], w0f[kk], acc);
        float p0[16], p1[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const float v = fmaxf(acc[i], 0.f);
          const float y = __shfl_xor(v, 32);
          p0[i] = h ? y : v;
          p1[i] = h ? v : y;
        }
        const uint32_t heads = (uint32_t)__ballot(is_head && h == 0);
        const uint32_t valid = (uint32_t)__ballot(act && h == 0);
        // point p ends a pillar if p+1 is invalid or a head; for p = 31 look at the prefetched first record of the next tile
        const int r_last = __shfl(r, 31), r_next = __shfl((int)nxt.b.w, 0);
        const uint32_t t31 = (slot0_of_tile + 32u >= end || r_next != r_last) ? 0x80000000u : 0u;
        const uint32_t tails = valid & ((((heads >> 1) | ~(valid >> 1)) & 0x7fffffffu) | t31);
#pragma unroll
        for (int p = 0; p < 32; p++) {
          const int i = (p & 3) + 4 * (p >> 3);
          const float v = ((p >> 2) & 1) ? p1[i] : p0[i];
          m = fmaxf(((heads >> p) & 1u) ? 0.f : m, ((valid >> p) & 1u) ? v : 0.f);
          if ((tails >> p) & 1u) {
            if (l < 32) sG0[(seg + __builtin_popcount(heads & ((2u << p) - 1u)) - 1) * GST + l] = m;
          }
        }
        seg += __builtin_popcount(heads);
        prev_rank = r_last;
        cur = nxt;
      }
    }
    WAVE_SYNC();

    TOCK(2);
    // ---- phase 2: layer 0 again (other orientation) chained into layer 1, per-pillar max -> feat_max rows
    {
      float m = 0.f;
      int seg = 0;
      int prev_rank = p_lo - 1;
      Rec cur = load_rec(rec, min(base + (uint32_t)col, end - 1));
      for (int t = 0; t < ntiles; t++) {
        const uint32_t slot0_of_tile = base + 32u * t;
        const uint32_t slot = slot0_of_tile + col;
        const bool act = slot < end;
        const Rec nxt = load_rec(rec, min(slot + 32u, end - 1));
        const int r = (int)cur.b.w;
        int rp = __shfl_up(r, 1);
        if (col == 0) rp = prev_rank;
        const bool is_head = act && (r != rp);
        const int sl = act ? r - p_lo : 0;
        float ff[KS];
        {
          float f[C0 + 2];
          decorate_rec<F>(cur, sMean[sl * 3], sMean[sl * 3 + 1], sMean[sl * 3 + 2], g, f);
          f[C0] = 1.f;
          f[C0 + 1] = 0.f;
#pragma unroll
          for (int kk = 0; kk < KS; kk++) ff[kk] = act ? (h ? f[2 * kk + 1] : f[2 * kk]) : 0.f;
        }
        v16f d0;
#pragma unroll
        for (int i = 0; i < 16; i++) d0[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; kk++) d0 = PNX_MFMA(w0f[kk], ff[kk], d0);
        TOCK(3);
        // "max" half of the concat: G0[pillar][8j + 4h .. +3], j = 0..3  == channel order of d0's registers
        float4 gq[4];
#pragma unroll
        for (int j = 0; j < 4; j++) gq[j] = *reinterpret_cast<const float4*>(&sG0[sl * GST + 8 * j + 4 * h]);
        v16f da, db;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          da[i] = s1a;
          db[i] = s1b;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const float a = fmaxf(d0[i], 0.f);
          da = PNX_MFMA(a, w1a[i], da);
          db = PNX_MFMA(a, w1b[i], db);
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const float4 q = gq[i >> 2];
          const float a = (i & 3) == 0 ? q.x : (i & 3) == 1 ? q.y : (i & 3) == 2 ? q.z : q.w;
          da = PNX_MFMA(a, w1a[16 + i], da);
          db = PNX_MFMA(a, w1b[16 + i], db);
        }
        float p0[16], p1[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const float va = fmaxf(da[i], 0.f), vb = fmaxf(db[i], 0.f);
          const float y = __shfl_xor(h ? va : vb, 32);
          p0[i] = h ? y : va;
          p1[i] = h ? vb : y;
        }
        if (dbg) { asm volatile("" :: "v"(p0[0]), "v"(p1[15])); }
        TOCK(4);
        const uint32_t heads = (uint32_t)__ballot(is_head && h == 0);
        const uint32_t valid = (uint32_t)__ballot(act && h == 0);
        // point p ends a pillar if p+1 is invalid or a head; for p = 31 look at the prefetched first record of the next tile
        const int r_last = __shfl(r, 31), r_next = __shfl((int)nxt.b.w, 0);
        const uint32_t t31 = (slot0_of_tile + 32u >= end || r_next != r_last) ? 0x80000000u : 0u;
        const uint32_t tails = valid & ((((heads >> 1) | ~(valid >> 1)) & 0x7fffffffu) | t31);
#pragma unroll
        for (int p = 0; p < 32; p++) {
          const int i = (p & 3) + 4 * (p >> 3);
          const float v = ((p >> 2) & 1) ? p1[i] : p0[i];
          m = fmaxf(((heads >> p) & 1u) ? 0.f : m, ((valid >> p) & 1u) ? v : 0.f);
          if ((tails >> p) & 1u) {
            const int64_t row = (int64_t)(p_lo + seg + __builtin_popcount(heads & ((2u << p) - 1u)) - 1);
            if (row < g1_rows) g1[row * 64 + l] = m;
          }
        }
        seg += __builtin_popcount(heads);
        prev_rank = r_last;
        cur = nxt;
        TOCK(5);
      }
    }
  }
  if (dbg && l == 0) {
    const int gw = blockIdx.x * 4 + wv;
#pragma unroll
    for (int k = 0; k < 8; k++) dbg[gw * 8 + k] = T[k];
  }
}

template <int F>
int launch_f(int R, const uint32_t* rec, const PnxGeomDev& g, const uint32_t* count, const uint32_t* cpre, const uint32_t* cblk,
             const int32_t* counters, const float* folded, float* g1, int64_t g1_rows, int64_t n, int max_blocks, hipStream_t st) {
  static unsigned long long* dbg = nullptr;
  static int dbg_calls = 0;
  const bool want_dbg = getenv("PNX_PFN_TIMING") != nullptr;
  if (want_dbg && !dbg) (void)hipMalloc(&dbg, 8192 * 4 * 8 * sizeof(unsigned long long));
  int64_t nb = ((n + R - 1) / R + 3) / 4;  // 4 waves per block, one slot window per wave and pass
  if (nb > max_blocks) nb = max_blocks;    // persistent: each wave strides over the slot windows
  if (nb < 1) nb = 1;
  if (getenv("PNX_DEBUG")) {
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_pfn_mfma<F, 64>, 256, 0);
    fprintf(stderr, "[pnx] k_pfn_mfma<%d,64>: occupancy API says %d blocks (x4 waves)/CU, launching %lld blocks, R=%d\n", F, occ, (long long)nb, R);
  }
  if (R == 128) k_pfn_mfma<F, 128><<<(int)nb, 256, 0, st>>>(rec, g, count, cpre, cblk, counters, folded, g1, g1_rows, want_dbg ? dbg : nullptr);
  else if (R == 32) k_pfn_mfma<F, 32><<<(int)nb, 256, 0, st>>>(rec, g, count, cpre, cblk, counters, folded, g1, g1_rows, want_dbg ? dbg : nullptr);
  else k_pfn_mfma<F, 64><<<(int)nb, 256, 0, st>>>(rec, g, count, cpre, cblk, counters, folded, g1, g1_rows, want_dbg ? dbg : nullptr);
  PNX_LAUNCH_CHECK();
  if (want_dbg && ++dbg_calls == 20) {
    std::vector<unsigned long long> hbuf((size_t)nb * 4 * 8);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(hbuf.data(), dbg, hbuf.size() * 8, hipMemcpyDeviceToHost);
    double acc[8] = {0};
    for (size_t w = 0; w < (size_t)nb * 4; w++)
      for (int k = 0; k < 8; k++) acc[k] += (double)hbuf[w * 8 + k];
    const char* names[8] = {"ownership loads", "phase0 mean", "phase1 total", "p2: load+decorate+d0", "p2: 64 MFMA + exchange", "p2: scan+stores", "", ""};
    for (int k = 0; k < 6; k++) fprintf(stderr, "[pnx-timing] %-24s %10.0f ticks/wave\n", names[k], acc[k] / (nb * 4));
  }
  return PNX_OK;
}

}  // namespace

int pnx_launch_pfn_mfma(int F, const uint32_t* rec, const PnxGeomDev& geom, const uint32_t* count, const uint32_t* cpre,
                        const uint32_t* cblk, const int32_t* counters, const float* folded, float* g1, int64_t g1_rows, int64_t n_points,
                        hipStream_t st) {
  const char* r_env = getenv("PNX_PFN_R");  // slots per pass: 32 | 64 | 128
  const int R = r_env ? atoi(r_env) : 64;
  const char* b_env = getenv("PNX_PFN_BLOCKS");
  const int max_blocks = b_env ? atoi(b_env) : 512;  // 256 CUs x 2 blocks x 4 waves = 2 waves per SIMD at ~230 VGPRs
  switch (F) {
    case 3: return launch_f<3>(R, rec, geom, count, cpre, cblk, counters, folded, g1, g1_rows, n_points, max_blocks, st);
    case 4: return launch_f<4>(R, rec, geom, count, cpre, cblk, counters, folded, g1, g1_rows, n_points, max_blocks, st);
    case 5: return launch_f<5>(R, rec, geom, count, cpre, cblk, counters, folded, g1, g1_rows, n_points, max_blocks, st);
    case 6: return launch_f<6>(R, rec, geom, count, cpre, cblk, counters, folded, g1, g1_rows, n_points, max_blocks, st);
  }
  pnx_set_error("num_point_features %d not in 3..6", F);
  return PNX_ERR_UNSUPPORTED;
}
