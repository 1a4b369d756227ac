// pfn_v3.hip -- the PFN (pillar_encoder.py:35-50 x2, :174-182) over the pillar-sorted, pre-decorated 64-byte records that
// reader_bins.h::k_bin_sort writes, fused with the zero-fill of the pillar-free canvas cells (pnx_fill.h).
//
// MFMA mapping: one wave = one tile of <= 32 points cut at pillar boundaries, lane = (point, h),
// h = lane>>5 selecting the even/odd K element of v_mfma_f32_32x32x2_f32 (an exact fp32 fmaf chain at the fp32 vector rate):
//   layer 0   D0 = W0'(32ch x K) * F^T: the lane's six operand words ARE the six record words it loaded (no select, no decoration,
//             no per-pillar mean here -- k_bin_sort did that once per point); D0 leaves 16 channels of the point in the lane,
//             which is the B fragment of layer 1 when K is visited in accumulator-register order (no transpose)
//   max       per-pillar max = segmented scan ACROSS LANES with DPP row shifts on the post-ReLU values (pnx_dppscan.h: two
//             instructions per register and step, steps no pillar of the tile needs skipped by a ballot); the pillar's layer-0
//             max reaches every point of the pillar with one ds_bpermute per register (the "max" half of the concat, pe:44,49).
//             Measured alternatives on C2/8 frames: ds_max_u32 into per-pillar LDS rows 399 us (LDS atomics serialise), a
//             lanes = channels running-max walk over LDS rows 374 us (a scalar-controlled 32-step loop per pass).
//   layer 1   fp16x3 (default): layer 0 leaves h0 pre-scaled by 2^SU, every value is split into fp16 hi (round-toward-zero) + lo
//             (the exact remainder, rounded toward zero): 22 significant bits; W1' * 2^SW is split the same way at fold time, and
//             hi*hi + hi*lo + lo*hi runs as 24 v_mfma_f32_32x32x16_f16 (fp32 accumulation) instead of 64 fp32 MFMAs of twice the
//             duration: 768 instead of 4 096 matrix-pipe cycles per tile, error ~1e-6 relative (on the order of the fp32 chain's own
//             rounding; tools/study_f16x3.py).  K is a contraction index, so the only layout requirement is that A and B put a
//             channel into the same K slot: slot (step s, lane half kg, element e) = the channel lane-half kg already holds as its
//             value 8s + e -- no transpose.  A tile whose pillar maximum would overflow fp16 (h0 >= 937) hands its pillars to
//             k_pfn3_tail (fp32 MFMA).  PNX_PFN_F16X3=0 selects the plain fp32 form: 64 MFMAs; shift + ReLU before the max (x -> relu(x + s) is monotone); the TAIL lane of every pillar writes the
//             finished row, in NATURAL channel order, into the wave's private LDS rows
//   store     8 lanes per pillar read 16-byte pieces of the finished rows and store them: one store instruction writes 8 complete
//             128-byte lines (bf16) of the NHWC canvas (round 1: 16 instructions of scattered 8-byte pieces per tile)
// Fused fill: blocks [0, n_fill) of the SAME launch run pnx_fill_tile over the 32x32-cell tiles (HBM-write bound, almost no
// ALU), the other blocks run the PFN (MFMA bound, little HBM) -- the two roles overlap on every CU and each canvas byte is still
// written exactly once.  Pillars with more than 32 points go to the big-pillar role blocks (pfn3_big_walk).
#include <vector>

#include "pnx_common.h"
#include "pnx_dppscan.h"
#include "pnx_fill.h"
#include "pfn_common.h"

namespace {

// One word hands out ~88 tickets/us (MI355X guide, "dequeue"): 16 ticket words in separate 128-byte lines, word s serving the
// windows s, s+16, ...; a wave starts on word (block & 15) and moves on when its word runs dry.  Returns the window or -1.
constexpr int kTickShards = 16, kTickStride = 32;
__device__ __forceinline__ int64_t next_window(int32_t* tick, int& shard, int& tried, int tk, int64_t nwin, int lane) {
  for (;;) {
    const int64_t w = (int64_t)tk * kTickShards + shard;
    if (w < nwin) return w;
    if (++tried >= kTickShards) return -1;
    shard = (shard + 1) & (kTickShards - 1);
    tk = ticket_wait(ticket_issue(tick + shard * kTickStride, lane));
  }
}

// Pillars with more than 32 points: one wave per pillar, two sweeps over its tiles (layer-0 max; layer 1 + max) with per-lane
// running maxima, then one all-lane reduction per register.  Rare at PillarNeXt-B resolution, common only for coarse voxels.
// k_bin_sort lists them (counters[3], biglist[0, bigcap)); a few role blocks of k_pfn3 take them by ticket (counters[5]) while the
// other blocks run the tiled PFN, and k_pfn3_tail takes what is left plus the pillars of tiles that left the fp16x3 range
// (counters[4], biglist[bigcap, 2 bigcap)).  fp32 MFMA with the unscaled weights in both cases.
template <int F>
struct BigWeights {
  static constexpr int C0 = F + 5, KS = (C0 + 2) / 2;
  float w0f[KS], w1a[32], w1b[32];
  const float4* s1lane;
  __device__ __forceinline__ void load(const float* __restrict__ P, int l) {
    constexpr int FR = 32 * C0 + 32 + 64 * 64 + 64;
    const float* __restrict__ FP = P + FR + l;
#pragma unroll
    for (int kk = 0; kk < KS; kk++) w0f[kk] = FP[kk * 64];
#pragma unroll
    for (int i = 0; i < 32; i++) {
      w1a[i] = FP[(23 + i) * 64];
      w1b[i] = FP[(55 + i) * 64];
    }
    s1lane = reinterpret_cast<const float4*>(P + FR + 64 * 89 + l * 32);
  }
};

template <int F>
__device__ __forceinline__ void pfn3_big_pillar(const int r, const BigWeights<F>& Wt, const uint4* __restrict__ rec, const uint32_t* __restrict__ pfirst,
                                                const uint32_t* __restrict__ pcnt, const int32_t* __restrict__ cell_of_pillar, const Pfn3Out& out,
                                                int col, int h) {
  constexpr int KS = BigWeights<F>::KS;
  const float* w0f = Wt.w0f;
  const float* w1a = Wt.w1a;
  const float* w1b = Wt.w1b;
  const float4* __restrict__ s1lane = Wt.s1lane;
  const uint32_t st = pfirst[r], c = pcnt[r];
  float g0[16];
#pragma unroll
  for (int i = 0; i < 16; i++) g0[i] = 0.f;  // post-ReLU maxima
  for (uint32_t t0 = 0; t0 < c; t0 += 32) {
    const bool act = t0 + col < c;
    const Half cur = load_half(rec, st + min(t0 + (uint32_t)col, c - 1), h);
    const float ff[6] = {__uint_as_float(cur.a.x), __uint_as_float(cur.a.y), __uint_as_float(cur.a.z),
                         __uint_as_float(cur.a.w), __uint_as_float(cur.b.x), __uint_as_float(cur.b.y)};
    v16f d0;
#pragma unroll
    for (int i = 0; i < 16; i++) d0[i] = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS; kk++) d0 = PNX_MFMA(w0f[kk], act ? ff[kk] : 0.f, d0);
#pragma unroll
    for (int i = 0; i < 16; i++) g0[i] = fmaxf(g0[i], act ? fmaxf(d0[i], 0.f) : 0.f);
  }
#pragma unroll
  for (int i = 0; i < 16; i++) {
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) g0[i] = fmaxf(g0[i], __shfl_xor(g0[i], d));  // inside each half
  }
  const float NI = -__builtin_inff();
  float pa[16], pb[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    pa[i] = NI;
    pb[i] = NI;
  }
  for (uint32_t t0 = 0; t0 < c; t0 += 32) {
    const bool act = t0 + col < c;
    const Half cur = load_half(rec, st + min(t0 + (uint32_t)col, c - 1), h);
    const float ff[6] = {__uint_as_float(cur.a.x), __uint_as_float(cur.a.y), __uint_as_float(cur.a.z),
                         __uint_as_float(cur.a.w), __uint_as_float(cur.b.x), __uint_as_float(cur.b.y)};
    v16f d0, da, db;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      d0[i] = 0.f;
      da[i] = 0.f;
      db[i] = 0.f;
    }
#pragma unroll
    for (int kk = 0; kk < KS; kk++) d0 = PNX_MFMA(w0f[kk], act ? ff[kk] : 0.f, d0);
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
  if (col == 0) {  // one lane per half: its 2 x 16 channels as 4-channel pieces
    const int64_t cell = (int64_t)cell_of_pillar[r];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float4 sa = s1lane[j], sb = s1lane[4 + j];
      const float va[4] = {fmaxf(pa[4 * j] + sa.x, 0.f), fmaxf(pa[4 * j + 1] + sa.y, 0.f), fmaxf(pa[4 * j + 2] + sa.z, 0.f),
                           fmaxf(pa[4 * j + 3] + sa.w, 0.f)};
      const float vb[4] = {fmaxf(pb[4 * j] + sb.x, 0.f), fmaxf(pb[4 * j + 1] + sb.y, 0.f), fmaxf(pb[4 * j + 2] + sb.z, 0.f),
                           fmaxf(pb[4 * j + 3] + sb.w, 0.f)};
#pragma unroll
      for (int half2 = 0; half2 < 2; half2++) {
        const float* v = half2 ? vb : va;
        const int chan0 = 32 * half2 + 8 * j + 4 * h;
        const int64_t grow = out.row_of != nullptr ? (int64_t)out.row_of[r] : (int64_t)r;
        if (out.g1 != nullptr && grow < out.g1_rows)
          *reinterpret_cast<float4*>(out.g1 + grow * 64 + chan0) = make_float4(v[0], v[1], v[2], v[3]);
        if (out.canvas != nullptr) {
          if (out.dt == PNX_F32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(out.canvas) + cell * 64 + chan0) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 p;
            if (out.dt == PNX_BF16) {
              p.x = bf16_rne(v[0]) | (bf16_rne(v[1]) << 16);
              p.y = bf16_rne(v[2]) | (bf16_rne(v[3]) << 16);
            } else {
              p.x = f16_rne(v[0]) | (f16_rne(v[1]) << 16);
              p.y = f16_rne(v[2]) | (f16_rne(v[3]) << 16);
            }
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out.canvas) + cell * 64 + chan0) = p;
          }
        }
      }
    }
  }
}

// ticketed walk over biglist[0, nbig): at most `cap` pillars per wave (cap < 0: no limit)
template <int F>
__device__ __forceinline__ void pfn3_big_walk(const BigWeights<F>& Wt, int32_t* counters, const int32_t* __restrict__ biglist, int nbig, int cap,
                                              const uint4* __restrict__ rec, const uint32_t* __restrict__ pfirst, const uint32_t* __restrict__ pcnt,
                                              const int32_t* __restrict__ cell_of_pillar, const Pfn3Out& out, int l) {
  const int col = l & 31, h = l >> 5;
  for (int done = 0; cap < 0 || done < cap; done++) {
    int bi = 0;
    if (l == 0) bi = atomicAdd(&counters[5], 1);
    bi = __builtin_amdgcn_readfirstlane(bi);
    if (bi >= nbig) break;
    pfn3_big_pillar<F>(biglist[bi], Wt, rec, pfirst, pcnt, cell_of_pillar, out, col, h);
  }
}

// counters: [0] = P, [1] = N' (kept points = sorted records), [3] = pillars of > 32 points (listed by k_bin_sort), [4] = pillars of
// tiles outside the fp16x3 range, [5] = big-pillar tickets; tick = window tickets
constexpr int kBigBlocks = 8;    // role blocks of k_pfn3 for the big pillars (32 waves)
constexpr int kBigPerWave = 64;  // ... each wave takes at most this many; k_pfn3_tail takes the rest with a full grid
template <int F, int R, int DT, bool PACK, bool H16>
__global__ __launch_bounds__(256) void k_pfn3(const uint4* __restrict__ rec, const uint32_t* __restrict__ pfirst, const uint32_t* __restrict__ pcnt,
                                             const int32_t* __restrict__ cell_of_pillar, int32_t* counters, int32_t* __restrict__ tick,
                                             int32_t* __restrict__ biglist, int bigcap, const float* __restrict__ P, Pfn3Out out, int n_fill, int n_bigb,
                                             PnxGeomDev g, PnxFillJob fj) {
  constexpr int C0 = F + 5, KS = (C0 + 2) / 2;     // K = C0 features + the constant-1 column that carries the folded BN shift
  constexpr int FR = 32 * C0 + 32 + 64 * 64 + 64;  // start of the fragment-ordered block (k_fold_bn)
  __shared__ __align__(16) uint32_t s_lds[4 * kWaveLds];
  if ((int)blockIdx.x < n_fill) {  // ---- fill role (block-uniform): this launch's share of the zero-fill tiles
    pnx_fill_share_dt<DT>(fj, g, s_lds, threadIdx.x, 256);
    return;
  }
  if ((int)blockIdx.x < n_fill + n_bigb) {  // ---- big-pillar role: the pillars k_bin_sort listed, one wave each, by ticket
    int nbig = counters[3];
    if (nbig > bigcap) nbig = bigcap;
    if (nbig > 0) {
      BigWeights<F> Wt;
      Wt.load(P, threadIdx.x & 63);
      pfn3_big_walk<F>(Wt, counters, biglist, nbig, kBigPerWave, rec, pfirst, pcnt, cell_of_pillar, out, threadIdx.x & 63);
    }
    return;
  }
  // ---- PFN role
  const int l = threadIdx.x & 63, col = l & 31, h = l >> 5, wv = threadIdx.x >> 6;
  uint32_t* s_out = s_lds + wv * kWaveLds;  // 32 pillar rows
  uint32_t* s_rank = s_out + 32 * kZS;
  uint32_t* s_cell = s_rank + 32;
  const int n_kept = counters[1];

  // weight fragments: coalesced loads, once per (persistent) wave
  const float* __restrict__ FP = P + FR + l;
  const float* __restrict__ FP2 = FP + 64 * 121;  // fp16x3 block (k_fold_bn)
  float w0f[KS];
#pragma unroll
  for (int kk = 0; kk < KS; kk++) w0f[kk] = H16 ? FP2[kk * 64] : FP[kk * 64];
  // fp32 form: w1a/w1b = rows 0..31 / 32..63 of W1' for the lane's K elements; fp16x3 form: the same 64 registers hold the hi
  // (wq[0..31]) and lo (wq[32..63]) fragments, index ((mt*4 + s)*4 + tq)
  float w1a[32], w1b[32];
  uint32_t wq[64];
  if (H16) {
#pragma unroll
    for (int i = 0; i < 64; i++) wq[i] = __float_as_uint(FP2[(7 + i) * 64]);
  } else {
#pragma unroll
    for (int i = 0; i < 32; i++) {
      w1a[i] = FP[(23 + i) * 64];
      w1b[i] = FP[(55 + i) * 64];
    }
  }
  const float4* __restrict__ s1lane = reinterpret_cast<const float4*>(P + FR + 64 * 89 + l * 32);  // s1 in this lane's channel order
  float s1a[16], s1b[16];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float4 sa = s1lane[j], sb = s1lane[4 + j];
    s1a[4 * j + 0] = sa.x, s1a[4 * j + 1] = sa.y, s1a[4 * j + 2] = sa.z, s1a[4 * j + 3] = sa.w;
    s1b[4 * j + 0] = sb.x, s1b[4 * j + 1] = sb.y, s1b[4 * j + 2] = sb.z, s1b[4 * j + 3] = sb.w;
  }

  // Windows of R sorted slots are handed out by ticket counters: while the fill blocks occupy their share of every CU only part of
  // the PFN blocks is resident, and a static deal would leave the late blocks' share for after the fill.
  const int64_t nwin = ((int64_t)n_kept + R - 1) / R;
  int shard = (int)(blockIdx.x & (kTickShards - 1)), tried = 0;
  int64_t pass = next_window(tick, shard, tried, ticket_wait(ticket_issue(tick + shard * kTickStride, l)), nwin, l);
  while (pass >= 0) {
    const int64_t slot0 = pass * R;
    const int64_t slot1 = (slot0 + R < n_kept) ? slot0 + R : n_kept;
    // pillars owned by this pass = those whose first slot lies in [slot0, slot1)
    bool head0;
    const uint32_t e0 = pillar_end_at(rec, slot0, pfirst, pcnt, &head0);
    const uint32_t base = head0 ? (uint32_t)slot0 : e0;
    uint32_t end = (uint32_t)n_kept;
    if (slot1 < n_kept) {
      bool head1;
      const uint32_t e1 = pillar_end_at(rec, slot1, pfirst, pcnt, &head1);
      end = head1 ? (uint32_t)slot1 : e1;
    }
    uint32_t ts = base;
    Half nxt;
    if (base < end) nxt = load_half(rec, min(ts + (uint32_t)col, end - 1), h);
    const int tk_next = ticket_issue(tick + shard * kTickStride, l);  // next window's ticket: issued behind this window's first record load
    while (ts < end) {
      const Half cur = nxt;
      const bool in_range = ts + (uint32_t)col < end;
      const int idx = (int)(cur.b.z & 0xFFFFu), rem = (int)(cur.b.z >> 16);
      const bool complete = in_range && (col + rem <= 31);
      const uint32_t V = (uint32_t)__ballot(complete && h == 0);
      const int nv = __builtin_popcount(V);
      if (nv == 0) {
        // the pillar at ts has more than 32 points: k_bin_sort listed it for the big-pillar role; step over it
        const int q = __builtin_amdgcn_readfirstlane((int)cur.b.w);  // lane 0 is in half 0: word = rank
        const uint32_t c = pcnt[q];
        ts += c;
        if (ts < end) nxt = load_half(rec, min(ts + (uint32_t)col, end - 1), h);
        continue;
      }
      const uint32_t ts_next = ts + (uint32_t)nv;
      nxt = load_half(rec, min(ts_next + (uint32_t)col, end - 1), h);  // prefetch the next tile while this one computes
      const bool act = col < nv;
      const uint32_t heads = (uint32_t)__ballot(act && idx == 0 && h == 0);
      const int npil = __builtin_popcount(heads);
      const int pid = __builtin_popcount(heads & (0xFFFFFFFFu >> (31 - col))) - 1;  // pillar of this lane inside the tile (act lanes)
      const int tail_lane = act ? l + rem : l;                                     // same half
      ScanPlan pl;
      pl.s1 = __ballot(act && idx >= 1) != 0;
      pl.s2 = __ballot(act && idx >= 2) != 0;
      pl.s4 = __ballot(act && idx >= 4) != 0;
      pl.s8 = __ballot(act && idx >= 8) != 0;
      uint32_t sm[5];
      scan_masks(sm, act ? idx : 0, col);

      // ---- layer 0 (lane = point, registers = channels)
      float ff[6] = {__uint_as_float(cur.a.x), __uint_as_float(cur.a.y), __uint_as_float(cur.a.z),
                     __uint_as_float(cur.a.w), __uint_as_float(cur.b.x), __uint_as_float(cur.b.y)};
      v16f d0;
#pragma unroll
      for (int i = 0; i < 16; i++) d0[i] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KS; kk++) d0 = PNX_MFMA(w0f[kk], act ? ff[kk] : 0.f, d0);
      // ---- "max" half of the concat (pe:43-44,49): per-pillar max of relu(layer 0), delivered to every point of the pillar.
      // ReLU first: max(relu(x)) == relu(max(x)), and 0 is then the identity of the masked scan.  The scan's 5 x 16 register-steps
      // are dealt, five at a time, into the gaps of the 32 layer-1 MFMAs that only need the point's own h0 (each fp32 32x32x2 MFMA
      // keeps the matrix pipe busy for 64 cycles): one in-order wave then overlaps its own VALU with its own MFMAs -- two waves of a
      // SIMD were measured NOT to overlap each other's phases (512 vs 256 blocks: 347 vs 398 us).
      float u[16], g0[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        u[i] = fmaxf(d0[i], 0.f);
        g0[i] = u[i];
      }
      scan_fence16(g0);
      v16f da, db;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        da[i] = 0.f;
        db[i] = 0.f;
      }
#define PNX_G0_PAIR(Pq)                                                             \
  {                                                                                 \
    constexpr int st_ = (Pq) / 16, rg_ = (Pq) % 16;                                 \
    scan_pair_f32<st_>(g0[rg_], sm);                                                \
  }
      bool ovf = false;
      if (H16) {
        // B operands of K steps 0, 1: the point's own (pre-scaled) h0
        uint32_t bh[16], bl[16];
#pragma unroll
        for (int tq = 0; tq < 8; tq++) split2_f16(u[2 * tq], u[2 * tq + 1], bh[tq], bl[tq]);
#define PNX_L1H(S, PROD, I0, N)                                                                                         \
  {                                                                                                     \
    da = PNX_MFMA16(as_v8h(&wq[((PROD) == 2 ? 32 : 0) + (0 * 4 + (S)) * 4]), as_v8h((PROD) == 1 ? &bl[4 * ((S) & 1)] : &bh[4 * ((S) & 1)]), da); \
    db = PNX_MFMA16(as_v8h(&wq[((PROD) == 2 ? 32 : 0) + (1 * 4 + (S)) * 4]), as_v8h((PROD) == 1 ? &bl[4 * ((S) & 1)] : &bh[4 * ((S) & 1)]), db); \
  }                                                                                                                     \
  __builtin_amdgcn_sched_barrier(0);                                                                                    \
  if ((N) > 0) {                                                                                          \
    _Pragma("unroll") for (int pq_ = 0; pq_ < (N); pq_++) {                                                             \
      switch (((I0) + pq_) / 16) {                                                                                      \
        case 0: scan_pair_f32<0>(g0[((I0) + pq_) % 16], sm); break;                                                     \
        case 1: scan_pair_f32<1>(g0[((I0) + pq_) % 16], sm); break;                                                     \
        case 2: scan_pair_f32<2>(g0[((I0) + pq_) % 16], sm); break;                                                     \
        case 3: scan_pair_f32<3>(g0[((I0) + pq_) % 16], sm); break;                                                     \
        default: scan_pair_f32<4>(g0[((I0) + pq_) % 16], sm); break;                                                    \
      }                                                                                                                 \
    }                                                                                                                   \
  }                                                                                                                     \
  __builtin_amdgcn_sched_barrier(0);
        // 12 MFMAs (K steps 0, 1 x {hi*hi, hi*lo, lo*hi} x 2 row tiles) with the 80 register-steps of the g0 scan in between
        PNX_L1H(0, 0, 0, 14) PNX_L1H(0, 1, 14, 14) PNX_L1H(0, 2, 28, 14) PNX_L1H(1, 0, 42, 14) PNX_L1H(1, 1, 56, 14) PNX_L1H(1, 2, 70, 10)
        if (pl.s1) {
#pragma unroll
          for (int i = 0; i < 16; i++)
            g0[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(tail_lane << 2, __builtin_bit_cast(int, g0[i])));
        }
        // the pillar maximum bounds every value of the pillar: one range test covers both operand halves
        float gm = fmaxf(fmaxf(fmaxf(g0[0], g0[1]), fmaxf(g0[2], g0[3])), fmaxf(fmaxf(g0[4], g0[5]), fmaxf(g0[6], g0[7])));
        gm = fmaxf(gm, fmaxf(fmaxf(fmaxf(g0[8], g0[9]), fmaxf(g0[10], g0[11])), fmaxf(fmaxf(g0[12], g0[13]), fmaxf(g0[14], g0[15]))));
        ovf = __ballot(act && !(gm < 60000.f)) != 0;
#pragma unroll
        for (int tq = 0; tq < 8; tq++) split2_f16(g0[2 * tq], g0[2 * tq + 1], bh[tq], bl[tq]);
        PNX_L1H(2, 0, 0, 0) PNX_L1H(2, 1, 0, 0) PNX_L1H(2, 2, 0, 0) PNX_L1H(3, 0, 0, 0) PNX_L1H(3, 1, 0, 0) PNX_L1H(3, 2, 0, 0)
#undef PNX_L1H
      } else {
#define PNX_L1A(I)                                                                  \
  {                                                                 \
    da = PNX_MFMA(w1a[I], u[I], da);                                                \
    db = PNX_MFMA(w1b[I], u[I], db);                                                \
  }                                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                \
  {                                                                 \
    PNX_G0_PAIR(5 * (I) + 0) PNX_G0_PAIR(5 * (I) + 1) PNX_G0_PAIR(5 * (I) + 2) PNX_G0_PAIR(5 * (I) + 3) PNX_G0_PAIR(5 * (I) + 4) \
  }                                                                                 \
  __builtin_amdgcn_sched_barrier(0);
        PNX_L1A(0) PNX_L1A(1) PNX_L1A(2) PNX_L1A(3) PNX_L1A(4) PNX_L1A(5) PNX_L1A(6) PNX_L1A(7)
        PNX_L1A(8) PNX_L1A(9) PNX_L1A(10) PNX_L1A(11) PNX_L1A(12) PNX_L1A(13) PNX_L1A(14) PNX_L1A(15)
#undef PNX_L1A
        if (pl.s1) {
#pragma unroll
          for (int i = 0; i < 16; i++)
            g0[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(tail_lane << 2, __builtin_bit_cast(int, g0[i])));
        }
        {
#pragma unroll
          for (int i = 0; i < 16; i++) {
            da = PNX_MFMA(w1a[16 + i], g0[i], da);
            db = PNX_MFMA(w1b[16 + i], g0[i], db);
          }
        }
      }
#undef PNX_G0_PAIR
      if (ovf) {
        // outside the fp16 range: every pillar of the tile goes to k_pfn3_tail (fp32 MFMA, unscaled weights)
        if (act && idx == 0 && h == 0) {
          const int at = atomicAdd(&counters[4], 1);
          if (at < bigcap) biglist[bigcap + at] = (int)cur.b.w;
        }
        ts = ts_next;
        continue;
      }
      // fp16x3: the accumulators carry the scale 2^(SU+SW); an exact power of two, folded into the shift's fma
      constexpr float kDs = H16 ? 1.0f / (float)(1 << (PNX_PFN_SU + PNX_PFN_SW)) : 1.0f;
      // ---- per-pillar max of relu(layer 1 + shift): scan on non-negative values, the result sits in the pillar's tail lane.
      // The tail lanes then write the finished rows in NATURAL channel order: accumulator registers 4j..4j+3 of half h are channels
      // 8j + 4h .. +3 (da) / 32 + those (db).
      if (PACK) {
        // 16-bit canvas and no fp32 feat_max output: round FIRST (round-to-nearest-even is monotone, so the max of the rounded values
        // is the rounded max, bit for bit) and scan two channels per register with v_pk_max_u16
        uint32_t q[16];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float a0 = fmaxf(__builtin_fmaf(da[4 * j], kDs, s1a[4 * j]), 0.f), a1 = fmaxf(__builtin_fmaf(da[4 * j + 1], kDs, s1a[4 * j + 1]), 0.f);
          const float a2 = fmaxf(__builtin_fmaf(da[4 * j + 2], kDs, s1a[4 * j + 2]), 0.f), a3 = fmaxf(__builtin_fmaf(da[4 * j + 3], kDs, s1a[4 * j + 3]), 0.f);
          const float b0 = fmaxf(__builtin_fmaf(db[4 * j], kDs, s1b[4 * j]), 0.f), b1 = fmaxf(__builtin_fmaf(db[4 * j + 1], kDs, s1b[4 * j + 1]), 0.f);
          const float b2 = fmaxf(__builtin_fmaf(db[4 * j + 2], kDs, s1b[4 * j + 2]), 0.f), b3 = fmaxf(__builtin_fmaf(db[4 * j + 3], kDs, s1b[4 * j + 3]), 0.f);
          q[2 * j] = cvt_pk16<DT>(a0, a1), q[2 * j + 1] = cvt_pk16<DT>(a2, a3);
          q[8 + 2 * j] = cvt_pk16<DT>(b0, b1), q[8 + 2 * j + 1] = cvt_pk16<DT>(b2, b3);
        }
        seg_max_pk16(q, sm, pl);
        // The record prefetch (issued a whole tile ago) is waited for HERE, before this tile's stores go out: otherwise the next
        // tile's first use of it waits behind those stores -- vmcnt is in-order and hipcc cannot count a data-dependent number of stores.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if (act && rem == 0) {
          uint32_t* dst = s_out + pid * kZSP + 2 * h;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            *reinterpret_cast<uint2*>(dst + 4 * j) = make_uint2(q[2 * j], q[2 * j + 1]);
            *reinterpret_cast<uint2*>(dst + 16 + 4 * j) = make_uint2(q[8 + 2 * j], q[8 + 2 * j + 1]);
          }
          if (h == 0) s_rank[pid] = cur.b.w;  // where the row goes
          else s_cell[pid] = cur.b.w;
        }
        wave_lds_sync();
        // ---- stores: lane -> (pillar l>>3 + 8*it, 16 bytes = channels 8*(l&7) .. +7): one instruction writes 8 complete 128-byte lines
        const int qq = l & 7;
        for (int p = l >> 3; p < npil; p += 8) {
          const uint4 x = *reinterpret_cast<const uint4*>(s_out + p * kZSP + 4 * qq);
          *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(out.canvas) + (int64_t)(int32_t)s_cell[p] * 64 + 8 * qq) = x;
        }
      } else {
        float pa[16], pb[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
          pa[i] = fmaxf(__builtin_fmaf(da[i], kDs, s1a[i]), 0.f);
          pb[i] = fmaxf(__builtin_fmaf(db[i], kDs, s1b[i]), 0.f);
        }
        {
          seg_max_nn16(pa, idx, col, pl);
          seg_max_nn16(pb, idx, col, pl);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see above
        if (act && rem == 0) {
          uint32_t* dst = s_out + pid * kZS + 4 * h;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            *reinterpret_cast<uint4*>(dst + 8 * j) =
                make_uint4(__float_as_uint(pa[4 * j]), __float_as_uint(pa[4 * j + 1]), __float_as_uint(pa[4 * j + 2]), __float_as_uint(pa[4 * j + 3]));
            *reinterpret_cast<uint4*>(dst + 32 + 8 * j) =
                make_uint4(__float_as_uint(pb[4 * j]), __float_as_uint(pb[4 * j + 1]), __float_as_uint(pb[4 * j + 2]), __float_as_uint(pb[4 * j + 3]));
          }
          if (h == 0) s_rank[pid] = cur.b.w;  // where the row goes
          else s_cell[pid] = cur.b.w;
        }
        wave_lds_sync();
        // ---- stores: lane -> (pillar l>>3 + 8*it, channels 8*(l&7) .. +7)
        const int qq = l & 7;
        for (int p = l >> 3; p < npil; p += 8) {
          const uint4* src = reinterpret_cast<const uint4*>(s_out + p * kZS + 8 * qq);
          const uint4 x0 = src[0], x1 = src[1];
          const float v[8] = {__uint_as_float(x0.x), __uint_as_float(x0.y), __uint_as_float(x0.z), __uint_as_float(x0.w),
                              __uint_as_float(x1.x), __uint_as_float(x1.y), __uint_as_float(x1.z), __uint_as_float(x1.w)};
          store_chunk<DT>(out, (int)s_rank[p], (int64_t)(int32_t)s_cell[p], qq, v);
        }
      }
      wave_lds_sync();  // the next tile rewrites the rows
      ts = ts_next;
    }
    pass = next_window(tick, shard, tried, ticket_wait(tk_next), nwin, l);
  }
}

template <int F>
__global__ __launch_bounds__(256) void k_pfn3_tail(const uint4* __restrict__ rec, const uint32_t* __restrict__ pfirst,
                                                  const uint32_t* __restrict__ pcnt, const int32_t* __restrict__ cell_of_pillar, int32_t* counters,
                                                  const int32_t* __restrict__ biglist, int bigcap, const float* __restrict__ P, Pfn3Out out,
                                                  int stat = 0) {
  const int l = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  int nbig = counters[3], novf = counters[4];
  if (nbig > bigcap) nbig = bigcap;
  if (novf > bigcap) novf = bigcap;
  // the role blocks of k_pfn3 normally drained the list (tickets handed out >= nbig) and no tile overflowed: nothing to do
  if (!stat && __hip_atomic_load(&counters[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nbig && wave >= novf) return;
  if (stat && wave >= nbig && wave >= novf) return;
  BigWeights<F> Wt;
  Wt.load(P, l);
  if (stat) {  // nobody else draws tickets (pfn_spans.hip): a static deal, no atomic round trip per pillar
    for (int bi = wave; bi < nbig; bi += nwaves) pfn3_big_pillar<F>(biglist[bi], Wt, rec, pfirst, pcnt, cell_of_pillar, out, l & 31, l >> 5);
  } else {
    pfn3_big_walk<F>(Wt, counters, biglist, nbig, -1, rec, pfirst, pcnt, cell_of_pillar, out, l);
  }
  for (int bi = wave; bi < novf; bi += nwaves) pfn3_big_pillar<F>(biglist[bigcap + bi], Wt, rec, pfirst, pcnt, cell_of_pillar, out, l & 31, l >> 5);
}

template <int F>
int launch3(const uint4* rec, const uint32_t* pfirst, const uint32_t* pcnt, const int32_t* cell_of_pillar, int32_t* counters, int32_t* tick, int32_t* biglist,
            int64_t bigcap, const float* folded, const Pfn3Out& out, int64_t n, int n_fill, const PnxGeomDev& g, const PnxFillJob& fj, hipStream_t st) {
  constexpr int R = 256;
  const char* b_env = getenv("PNX_PFN_BLOCKS");
  const int max_blocks = b_env ? atoi(b_env) : 512;  // 256 CUs x 2 blocks x 4 waves = 2 waves per SIMD
  int64_t nb = ((n + R - 1) / R + 3) / 4;
  if (nb > max_blocks) nb = max_blocks;
  if (n <= 0) nb = 0;
  const int bc = (int)(bigcap > 0x7fffffff ? 0x7fffffff : bigcap);
  const int n_bigb = nb > 0 ? kBigBlocks : 0;
  if (nb + n_fill > 0) {
    const int grid = (int)(nb + n_fill + n_bigb);
    const bool pack = out.g1 == nullptr && out.canvas != nullptr && out.dt != PNX_F32;
    const char* h_env = getenv("PNX_PFN_F16X3");  // 0: plain fp32 MFMA layer 1
    const bool h16 = !(h_env && h_env[0] == '0');
#define PNX_GO(DT_, PACK_)                                                                                                                        \
  {                                                                                                                                               \
    if (h16) k_pfn3<F, R, DT_, PACK_, true><<<grid, 256, 0, st>>>(rec, pfirst, pcnt, cell_of_pillar, counters, tick, biglist, bc, folded, out, n_fill, n_bigb, g, fj); \
    else k_pfn3<F, R, DT_, PACK_, false><<<grid, 256, 0, st>>>(rec, pfirst, pcnt, cell_of_pillar, counters, tick, biglist, bc, folded, out, n_fill, n_bigb, g, fj);    \
  }
    if (out.dt == PNX_F32) {
      PNX_GO(PNX_F32, false)
    } else if (out.dt == PNX_BF16) {
      if (pack) PNX_GO(PNX_BF16, true) else PNX_GO(PNX_BF16, false)
    } else {
      if (pack) PNX_GO(PNX_F16, true) else PNX_GO(PNX_F16, false)
    }
#undef PNX_GO
    PNX_LAUNCH_CHECK();
  }
  if (n > 0) {
    k_pfn3_tail<F><<<64, 256, 0, st>>>(rec, pfirst, pcnt, cell_of_pillar, counters, biglist, bc, folded, out);
    PNX_LAUNCH_CHECK();
  }
  return PNX_OK;
}

}  // namespace

// n_fill > 0: blocks [0, n_fill) of the launch take the zero-fill tiles of `fj` (pnx_fill.h) concurrently with the PFN.
int pnx_launch_pfn_v3(int F, const uint32_t* rec64, const uint32_t* pfirst, const uint32_t* pcnt, const int32_t* cell_of_pillar,
                      int32_t* counters, int32_t* tick, int32_t* biglist, int64_t bigcap, const float* folded, float* g1, int64_t g1_rows,
                      void* canvas, int canvas_dt, int64_t n_points, int n_fill, const PnxGeomDev& geom, const PnxFillJob& fj, hipStream_t st) {
  Pfn3Out out;
  out.g1 = g1;
  out.g1_rows = g1_rows;
  out.canvas = canvas;
  out.dt = canvas_dt;
  const uint4* rec = reinterpret_cast<const uint4*>(rec64);
  switch (F) {
    case 3: return launch3<3>(rec, pfirst, pcnt, cell_of_pillar, counters, tick, biglist, bigcap, folded, out, n_points, n_fill, geom, fj, st);
    case 4: return launch3<4>(rec, pfirst, pcnt, cell_of_pillar, counters, tick, biglist, bigcap, folded, out, n_points, n_fill, geom, fj, st);
    case 5: return launch3<5>(rec, pfirst, pcnt, cell_of_pillar, counters, tick, biglist, bigcap, folded, out, n_points, n_fill, geom, fj, st);
    case 6: return launch3<6>(rec, pfirst, pcnt, cell_of_pillar, counters, tick, biglist, bigcap, folded, out, n_points, n_fill, geom, fj, st);
  }
  pnx_set_error("num_point_features %d not in 3..6", F);
  return PNX_ERR_UNSUPPORTED;
}

// k_pfn3_tail alone, for the LDS-sorted path (pfn_spans.hip): the pillars it spilled to the 64-byte record stream -- more than 32
// points (biglist[0, bigcap), counters[3]) or a tile outside the fp16x3 range (biglist[bigcap, 2 bigcap), counters[4]).
int pnx_launch_pfn3_tail(int F, const uint32_t* rec64, const uint32_t* pfirst, const uint32_t* pcnt, const int32_t* cell_of_pillar, int32_t* counters,
                         const int32_t* biglist, int64_t bigcap, const float* folded, float* g1, int64_t g1_rows, void* canvas, int canvas_dt, int blocks,
                         hipStream_t st, const int32_t* row_of) {
  Pfn3Out out;
  out.g1 = g1;
  out.g1_rows = g1_rows;
  out.canvas = canvas;
  out.dt = canvas_dt;
  out.row_of = row_of;
  const uint4* rec = reinterpret_cast<const uint4*>(rec64);
  const int bc = (int)(bigcap > 0x7fffffff ? 0x7fffffff : bigcap);
  switch (F) {
    case 3: k_pfn3_tail<3><<<blocks, 256, 0, st>>>(rec, pfirst, pcnt, cell_of_pillar, counters, biglist, bc, folded, out, 1); break;
    case 4: k_pfn3_tail<4><<<blocks, 256, 0, st>>>(rec, pfirst, pcnt, cell_of_pillar, counters, biglist, bc, folded, out, 1); break;
    case 5: k_pfn3_tail<5><<<blocks, 256, 0, st>>>(rec, pfirst, pcnt, cell_of_pillar, counters, biglist, bc, folded, out, 1); break;
    case 6: k_pfn3_tail<6><<<blocks, 256, 0, st>>>(rec, pfirst, pcnt, cell_of_pillar, counters, biglist, bc, folded, out, 1); break;
    default: pnx_set_error("num_point_features %d not in 3..6", F); return PNX_ERR_UNSUPPORTED;
  }
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}
