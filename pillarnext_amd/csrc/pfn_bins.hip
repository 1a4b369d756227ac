// pfn_bins.hip -- pillar sort INSIDE LDS + PFN + canvas stores in ONE launch (gfx950): the pillar-sorted record stream never exists
// in HBM.  Reference semantics: pillar_encoder.py:106-123 (torch.unique inverse, scatter_mean, feature decoration), :35-50 x2
// (PFNLayer), :174-182 (PillarFeatureNet.forward) and the dense canvas of sparse_resnet.py:63-68.
//
// Round 2 (reader_bins.h::k_bin_sort + pfn_v3.hip::k_pfn3) wrote every kept point as a 64-byte decorated record to HBM and read it
// back: 308 MB of the reader's 3.0 GB per 8 nuScenes frames; its PFN walked tiles cut at pillar boundaries, a chain of dependent
// loads, and spent half of its ~950 instructions per tile on MASKED segmented max scans across lanes (two instructions per register
// and step, five steps, plus the wait states of DPP).  Here one workgroup owns one bin (2^sh consecutive pillars of torch.unique
// order, all of its raw 32-byte records contiguous in the bin buffer that k_bin_scatter wrote) and does, per bin:
//   pass 1   points per pillar + exact fp64 coordinate sums with LDS atomics; the raw records of the first rounds stay in registers
//   scan     exclusive scans over the pillar counts (slots of the virtual global sorted order; padded LDS slots); per pillar: mean
//            (fp32 divide of the fp64 sum, pe:113-114), pillar centre (pe:119-120), canvas cell
//   layout   the order of the records in LDS is OURS to choose, so pillars are grouped by SIZE CLASS: a pillar of n <= 32 points
//            gets an aligned group of G = 1, 2, 4, 8, 16 or 32 slots (the next power of two; the spare slots repeat its last point,
//            which no maximum notices), and a tile = 32 consecutive slots of ONE class = 32/G whole pillars
//   pass 2   every point's RAW 32-byte record -> its slot; the tile decorates it (pe:116-123) from the pillar's constants into the operand
//            order of v_mfma_f32_32x32x2_f32 (lane (point, h) feeds K elements 2kk+h).  (48-byte decorated LDS records were tried: fewer
//            record slots fit, bins became multi-segment, slower.)
//   PFN      per tile: layer 0 (fp32 MFMA), per-pillar max = an UNMASKED xor butterfly inside aligned lane groups (one
//            v_max_f32_dpp per register and step, log2 G steps, the result lands in every lane of the group: no masks, no
//            ballots, no bpermute, nothing for single-point pillars), layer 1 (fp16x3 MFMA, pfn_v3.hip), the same butterfly on the
//            outputs, finished pillar lines stored straight into the NHWC canvas / feat_max rows.  Tile addresses are plain
//            arithmetic (no dependent loads): the next tile's records are fetched from LDS under this tile's MFMAs.
// A bin whose padded points exceed the LDS record slots is processed in pillar-aligned segments (layout + pass 2 + PFN per segment,
// the raw records re-read from L2).  Pillars of > 32 points and pillars whose layer-0 maximum leaves the fp16x3 range are written to
// the 64-byte record stream in the round-2 format and listed for k_pfn3_tail (fp32 MFMA, one wave per pillar) -- rare at PillarNeXt
// resolutions.  Bins are handed out by a ticket counter; blocks [0, n_fill) of the launch carry zero-fill tiles (pnx_fill.h).
// Nothing computed depends on where a pillar sits in a tile or on the order of its records (every MFMA column is independent, max
// is order-free, the mean is an exact fp64 sum): the reader stays deterministic and permutation-invariant bit for bit.
#include <type_traits>

#include "pnx_common.h"
#include "pnx_fill.h"
#include "pfn_common.h"

namespace {

constexpr int kRecW = 8;      // words per LDS record: the raw 32-byte bin record [x y z f3 | f4 f5 key pillar]
constexpr int kKeepR = 4;     // rounds of raw records kept in registers between pass 1 and pass 2
constexpr int kBinBlock = 256;
constexpr int kClasses = 6;   // group sizes 1, 2, 4, 8, 16, 32
constexpr int kClassSlack = kClasses * 32;  // every class region is rounded up to whole tiles

// Section timers (build with PNX_BINS_TIMERS=1): lane 0 of every wave adds the s_memtime delta since its previous mark to an LDS
// slot; the sums go to A.timers at the end.  Phases: 0 ticket + top barrier, 1 bin range + clears, 2 pass 1 (loads, tallies), 3 scans +
// pillar constants, 4 layout + pass 2, 5 own tiles, 6 end-of-bin barrier (imbalance), 7 weight loads; 8 = tiles, 9 = bins (counts).
#ifdef PNX_BINS_TIMERS
#define PNX_TMARK(k)                                                    \
  do {                                                                  \
    if (l == 0) {                                                       \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
      s_tim[wv * 16 + (k)] += now_ - tlast;                             \
      tlast = now_;                                                     \
    }                                                                   \
  } while (0)
#define PNX_TCOUNT(k, n)                      \
  do {                                        \
    if (l == 0) s_tim[wv * 16 + (k)] += (n);  \
  } while (0)
#else
#define PNX_TMARK(k) \
  do {               \
  } while (0)
#define PNX_TCOUNT(k, n) \
  do {                   \
  } while (0)
#endif

struct BinPfnArgs {
  const uint32_t* binbuf;        // raw 32-byte records [x y z f3 f4 f5 | cell key | pillar inside the bin], bins contiguous
  const uint32_t *hpre, *hblk;   // prefix of the (bin x workgroup) histogram matrix (pnx_scan.h)
  int32_t* counters;             // [0] P  [1] N'  [3] big pillars listed  [4] overflow pillars listed  [5] tail tickets
  int32_t* tick;                 // bin tickets
  uint32_t* rec64;               // 64-byte record stream: only spilled pillars are written
  uint32_t *pfirst, *pcnt;
  int32_t* cell_of_pillar;
  int32_t* coords;
  int64_t pillar_capacity;
  int32_t* biglist;
  int bigcap;
  const float* P;                // folded parameters (k_fold_bn)
  int sh, nwg, K1, cap, n_fill, write_pillars;
  int64_t matlen;
  unsigned long long* timers;  // PNX_BINS_TIMERS builds: 16 x u64 of wave-cycle sums per phase
};

__device__ __forceinline__ uint32_t bin_prefix(int64_t v, const uint32_t* __restrict__ hpre, const uint32_t* __restrict__ hblk) {
  return hblk[v >> PNX_SCAN_SHIFT] + hpre[v];
}

// size class of a pillar of 1..32 points: log2 of the next power of two
__device__ __forceinline__ uint32_t size_class(uint32_t cnt) { return cnt <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(cnt - 1u); }

// LDS words of one workgroup for bins of 2^sh pillars and `cap` record slots
template <bool PACK>
constexpr int wave_out_words() {
  return (PACK ? 32 * kZSP : 32 * kZS) + 64;
}
static inline size_t bin_pfn_lds_bytes(int sh, int cap, bool pack) {
  const size_t S = (size_t)1 << sh;
  const size_t head = (3 * (S + 4) + 3 * S) * 4 + 3 * S * 8 + (48 + 4 * 24 + 64) * 4;
  return head + 4 * (size_t)(pack ? wave_out_words<true>() : wave_out_words<false>()) * 4 + (size_t)cap * kRecW * 4;
}

// ---- all-reduce max inside aligned groups of 2^LG lanes (of each 32-lane half): xor butterfly, one DPP max per step
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int LG, int N>
__device__ __forceinline__ void group_max(float* v) {
  if (LG >= 1) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = fmaxf(v[i], dpp_perm<0xB1>(v[i]));  // quad_perm [1,0,3,2]
  }
  if (LG >= 2) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = fmaxf(v[i], dpp_perm<0x4E>(v[i]));  // quad_perm [2,3,0,1]
  }
  if (LG >= 3) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = fmaxf(v[i], dpp_perm<0x141>(v[i]));  // row_half_mirror: the other quad of the 8
  }
  if (LG >= 4) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = fmaxf(v[i], dpp_perm<0x140>(v[i]));  // row_mirror: the other 8 of the row
  }
  if (LG >= 5) {
    // the other 16-lane row of the half: v_permlane16_swap trades the odd rows of its first operand with the even rows of the second
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t x = __float_as_uint(v[i]);
      const u32x2 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
      v[i] = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
  }
}

template <int F, int DT, bool PACK>
__global__ __launch_bounds__(kBinBlock, 2) void k_bin_pfn(BinPfnArgs A, Pfn3Out out, PnxGeomDev g, PnxFillJob fj) {
  constexpr int C0 = F + 5, KS = (C0 + 2) / 2;     // K = C0 features + the constant-1 column that carries the folded BN shift
  constexpr int FR = 32 * C0 + 32 + 64 * 64 + 64;  // start of the fragment-ordered block (k_fold_bn)
  constexpr int WL = wave_out_words<PACK>();
  extern __shared__ __align__(16) uint32_t s_raw[];
  const int t = threadIdx.x;
  if ((int)blockIdx.x < A.n_fill) {  // ---- fill role (block-uniform): this launch's share of the zero-fill tiles
    pnx_fill_share_dt<DT>(fj, g, s_raw, t, kBinBlock);
    return;
  }
  const int sh = A.sh, S = 1 << sh;
  uint32_t* s_cnt = s_raw;               // S + 1: points per pillar
  uint32_t* s_gst = s_cnt + (S + 4);     // S + 1: exclusive starts, all pillars (slots of the global sorted order, relative to the bin)
  uint32_t* s_pst = s_gst + (S + 4);     // S + 1: exclusive starts of the PADDED sizes of the pillars of <= 32 points
  uint32_t* s_slot = s_pst + (S + 4);    // S: first LDS slot of the pillar in the current segment
  uint32_t* s_cur = s_slot + S;          // S cursors
  uint32_t* s_key = s_cur + S;           // S cell keys
  double* s_sum = reinterpret_cast<double*>(s_key + S);  // 3 doubles per pillar; later {mean x y z, centre x y, cell} as 6 words
  uint32_t* s_misc = reinterpret_cast<uint32_t*>(s_sum + 3 * S);  // [0..7] wave sums [8] ticket [9] segment end [10..15] pillars per class [16..39] per wave
  uint32_t* s_wt = s_misc + 48;  // per wave: [0..5] first slot of the wave's pillars per class, [8..13] first slot of the class, [16..21] pillars of the class
  float* s_s1 = reinterpret_cast<float*>(s_wt + 4 * 24);  // 2 x 32 pre-scaled layer-1 shifts
  uint32_t* s_outb = s_misc + 48 + 4 * 24 + 64;
  uint32_t* s_rec = s_outb + 4 * WL;

  const int l = t & 63, col = l & 31, h = l >> 5, wv = __builtin_amdgcn_readfirstlane(t >> 6);  // wave id in an SGPR: tile arithmetic stays scalar
  uint32_t* s_out = s_outb + wv * WL;  // 32 finished pillar rows of this wave
  uint32_t* s_rank = s_out + (PACK ? 32 * kZSP : 32 * kZS);
  uint32_t* s_cellrow = s_rank + 32;
#ifdef PNX_BINS_TIMERS
  __shared__ unsigned long long s_tim[4 * 16];
  if (t < 64) s_tim[t] = 0ull;
  __syncthreads();
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif

  // weight fragments: coalesced loads, once per (persistent) wave -- fp16x3 block of k_fold_bn
  const float* __restrict__ FP2 = A.P + FR + 64 * 121 + l;
  float w0f[KS];
#pragma unroll
  for (int kk = 0; kk < KS; kk++) w0f[kk] = FP2[kk * 64];
  uint32_t wq[64];  // hi (wq[0..31]) and lo (wq[32..63]) fragments of W1' * 2^SW, index ((mt*4 + s)*4 + tq)
#pragma unroll
  for (int i = 0; i < 64; i++) wq[i] = __float_as_uint(FP2[(7 + i) * 64]);
  // the folded BN shift of layer 1 in this lane's channel order (two variants, by lane half), pre-scaled by 2^(SU+SW): the layer-1
  // accumulators START from it (ds_read broadcast per tile instead of 32 resident registers and 32 fma in the epilogue)
  {
    const float* __restrict__ s1lane = A.P + FR + 64 * 89 + l * 32;
    if (wv == 0 && col == 0)
      for (int i = 0; i < 32; i++) s_s1[h * 32 + i] = __fmul_rn(s1lane[i], (float)(1 << (PNX_PFN_SU + PNX_PFN_SW)));
  }
  __syncthreads();

  const int64_t Ptot = A.counters[0];
  const uint32_t n_kept = (uint32_t)A.counters[1];
  const uint32_t cap = (uint32_t)A.cap - (uint32_t)kClassSlack;  // padded points of a segment; the class regions add at most the slack
  const uint32_t* __restrict__ binbuf = A.binbuf;
  int tk = ticket_issue(A.tick, t);  // thread 0 only; the value is read at the top of the loop
  PNX_TMARK(7);
  for (;;) {
    // the grouping phases are short chains of LDS latencies and barriers: they get the issue slots before the other workgroup's tiles
    __builtin_amdgcn_s_setprio(2);
    if (wv == 0) {
      const int bq = ticket_wait(tk);
      if (t == 0) s_misc[8] = (uint32_t)bq;
    }
    __syncthreads();
    const int b = __builtin_amdgcn_readfirstlane((int)s_misc[8]);
    const int64_t r0 = (int64_t)b << sh;
    PNX_TMARK(0);
    if (b >= A.K1 || r0 >= Ptot) break;  // tickets come in bin order: every later bin is empty as well
    PNX_TCOUNT(9, 1);
    const int64_t v0 = (int64_t)b * A.nwg, v1 = v0 + A.nwg;
    const uint32_t bs = bin_prefix(v0, A.hpre, A.hblk);
    const uint32_t be = v1 >= A.matlen ? n_kept : bin_prefix(v1, A.hpre, A.hblk);

    for (int p = t; p < S; p += kBinBlock) {
      s_cnt[p] = 0u;
      s_sum[3 * p + 0] = 0.0;
      s_sum[3 * p + 1] = 0.0;
      s_sum[3 * p + 2] = 0.0;
    }
    __syncthreads();
    PNX_TMARK(1);
    // ---- pass 1: points per pillar, exact coordinate sums (scatter_mean numerator, pe:113), the pillar's cell key
    uint4 ka[kKeepR], kc[kKeepR];
    auto tally = [&](const uint4& a, const uint4& c) {
      const uint32_t rl = c.w;
      atomicAdd(&s_cnt[rl], 1u);
      s_key[rl] = c.z;  // every point of the pillar stores the same key
      atomicAdd(&s_sum[3 * rl + 0], (double)__uint_as_float(a.x));
      atomicAdd(&s_sum[3 * rl + 1], (double)__uint_as_float(a.y));
      atomicAdd(&s_sum[3 * rl + 2], (double)__uint_as_float(a.z));
    };
#pragma unroll
    for (int it = 0; it < kKeepR; it++) {
      const uint32_t j = bs + it * kBinBlock + t;
      ka[it] = make_uint4(0u, 0u, 0u, 0u);
      kc[it] = make_uint4(0u, 0u, 0u, 0u);
      if (j < be) {
        const uint4* q = reinterpret_cast<const uint4*>(binbuf + (int64_t)j * 8);
        ka[it] = q[0];
        kc[it] = q[1];
      }
    }
#pragma unroll
    for (int it = 0; it < kKeepR; it++)
      if (bs + it * kBinBlock + t < be) tally(ka[it], kc[it]);
    for (uint32_t j = bs + kKeepR * kBinBlock + t; j < be; j += kBinBlock) {
      const uint4* q = reinterpret_cast<const uint4*>(binbuf + (int64_t)j * 8);
      tally(q[0], q[1]);
    }
    __syncthreads();
    PNX_TMARK(2);
    // ---- two exclusive scans of the S counts (thread t owns the E = S/256 entries t*E ..), pillar constants
    uint32_t cbase[kClasses], ncls[kClasses];  // first slot / pillars of every class in the current segment (wave-uniform); a class = whole tiles
    {
      const int E = S >> 8;
      uint32_t cg[8];
      uint32_t sg = 0, sp = 0;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        cg[e] = e < E ? s_cnt[t * E + e] : 0u;
        sg += cg[e];
        sp += (cg[e] >= 1u && cg[e] <= 32u) ? (1u << size_class(cg[e])) : 0u;
      }
      uint32_t ig = sg, ip = sp;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t yg = __shfl_up(ig, d), yp = __shfl_up(ip, d);
        if (l >= d) ig += yg, ip += yp;
      }
      if (l == 63) s_misc[wv] = ig, s_misc[4 + wv] = ip;
      // class layout of the single-segment case, without another barrier: ordinal of every pillar inside its size class = pillars
      // of that class in earlier waves + in earlier entries / lower lanes of this wave (ballots)
      uint32_t ordv[8], wcls[kClasses];
#pragma unroll
      for (int c = 0; c < kClasses; c++) wcls[c] = 0u;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        ordv[e] = 0u;
        if (e < E) {
          const uint32_t cnt = cg[e];
          const bool small = cnt >= 1u && cnt <= 32u;
          const uint32_t c = small ? size_class(cnt) : 0xFFu;
#pragma unroll
          for (int q = 0; q < kClasses; q++) {
            const uint64_t m = __ballot(c == (uint32_t)q);
            if (c == (uint32_t)q) ordv[e] = wcls[q] + (uint32_t)__builtin_popcountll(m & ((1ull << l) - 1ull));
            wcls[q] += (uint32_t)__builtin_popcountll(m);
          }
        }
      }
      if (l == 0) {
#pragma unroll
        for (int q = 0; q < kClasses; q++) s_misc[16 + wv * kClasses + q] = wcls[q];
      }
      __syncthreads();
      uint32_t og = 0, op = 0;
      for (int w = 0; w < wv; w++) og += s_misc[w], op += s_misc[4 + w];
      uint32_t eg = og + ig - sg, ep = op + ip - sp;
      // class table of this wave (lane c < 6 owns class c; kept in LDS so that nothing is indexed in registers): pillars of the class,
      // first slot of the class (a class = whole tiles), first slot of THIS wave's pillars of the class
      {
        uint32_t n = 0, before = 0;
        if (l < kClasses) {
          for (int w = 0; w < 4; w++) {
            const uint32_t x = s_misc[16 + w * kClasses + l];
            n += x;
            before += w < wv ? x : 0u;
          }
        }
        const uint32_t sz = l < kClasses ? ((((n << l) + 31u) >> 5) << 5) : 0u;
        uint32_t inc = sz;
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
          const uint32_t y = __shfl_up(inc, d);
          if (l >= d) inc += y;
        }
        if (l < kClasses) {
          const uint32_t cb = inc - sz;
          s_wt[wv * 24 + l] = cb + (before << l);
          s_wt[wv * 24 + 8 + l] = cb;
          s_wt[wv * 24 + 16 + l] = n;
        }
      }
      wave_lds_sync();
#pragma unroll
      for (int c = 0; c < kClasses; c++) {
        cbase[c] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_wt[wv * 24 + 8 + c]);
        ncls[c] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_wt[wv * 24 + 16 + c]);
      }
#pragma unroll
      for (int e = 0; e < 8; e++) {
        if (e < E) {
          const int p = t * E + e;
          const uint32_t cnt = cg[e];
          s_gst[p] = eg;
          s_pst[p] = ep;
          s_cur[p] = 0u;
          if (cnt >= 1u && cnt <= 32u) {
            const uint32_t c = size_class(cnt);
            s_slot[p] = s_wt[wv * 24 + c] + (ordv[e] << c);
          }
          if (cnt > 0u) {
            const double sx = s_sum[3 * p + 0], sy = s_sum[3 * p + 1], sz = s_sum[3 * p + 2];
            const float fc = (float)cnt;
            const int32_t k = (int32_t)s_key[p];
            const int yi = k % g.gyp;
            const int tq = k / g.gyp;
            const int xi = tq % g.gx, bi = tq / g.gx;
            // rows mode (pnx_reader_forward_rows, write_pillars & 2): the 'canvas' is a (P, 64) row buffer, a pillar's line goes to its rank
            const int32_t cell = (A.write_pillars & 2) ? (int32_t)(r0 + p) : (bi * g.gy + yi) * g.gx + xi;
            float* info = reinterpret_cast<float*>(&s_sum[3 * p]);  // overlays this thread's own three sums
            // mean: fp32 divide of the fp64 sum (pe:113-114); centre: idx*vs + vs/2 + min, each step rounded (pe:119-120)
            const float mx = __fdiv_rn((float)sx, fc), my = __fdiv_rn((float)sy, fc), mz = __fdiv_rn((float)sz, fc);
            const float ctrx = __fadd_rn(__fadd_rn(__fmul_rn((float)xi, g.vx), __fdiv_rn(g.vx, 2.0f)), g.minx);
            const float ctry = __fadd_rn(__fadd_rn(__fmul_rn((float)yi, g.vy), __fdiv_rn(g.vy, 2.0f)), g.miny);
            info[0] = mx, info[1] = my, info[2] = mz, info[3] = ctrx, info[4] = ctry;
            info[5] = __int_as_float(cell);
            const int64_t gr = r0 + p;
            if ((A.write_pillars & 1) || cnt > 32u) {
              A.pfirst[gr] = bs + eg;
              A.pcnt[gr] = cnt;
              A.cell_of_pillar[gr] = cell;
            }
            if (cnt > 32u) {  // more points than one MFMA tile holds: one wave per pillar in k_pfn3_tail
              const int at = atomicAdd(&A.counters[3], 1);
              if (at < A.bigcap) A.biglist[at] = (int)gr;
            }
            if (A.coords != nullptr && gr < A.pillar_capacity) {
              A.coords[gr * 3 + 0] = bi;  // [b, yi, xi]  (pe:125 swaps x/y)
              A.coords[gr * 3 + 1] = yi;
              A.coords[gr * 3 + 2] = xi;
            }
          }
          eg += cnt;
          ep += (cnt >= 1u && cnt <= 32u) ? (1u << size_class(cnt)) : 0u;
        }
      }
      if (t == kBinBlock - 1) s_gst[S] = eg, s_pst[S] = ep;
    }
    __syncthreads();
    PNX_TMARK(3);
    const uint32_t npad = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pst[S]);
    // ---- segments of at most `cap` padded slots, cut at pillar boundaries (normally one)
    uint32_t p0 = 0, base = 0;
    // pillars [p0, p1) = the longest run from p0 whose padded points fit into `cap` slots; then the class layout of that run
    auto open_segment = [&]() -> uint32_t {
      uint32_t p1 = (uint32_t)S;
      if (t < kClasses) s_misc[10 + t] = 0u;
      if (npad - base > cap) {
        const uint32_t lim = base + cap;
        for (uint32_t p = t; p < (uint32_t)S; p += kBinBlock)
          if (p >= p0 && s_pst[p] <= lim && s_pst[p + 1] > lim) s_misc[9] = p;  // exactly one p; > p0 because a padded pillar is <= 32 <= cap
        __syncthreads();
        p1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_misc[9]);
      } else {
        __syncthreads();
      }
      // ordinal of every pillar inside its class (any bijection will do: nothing depends on where a pillar sits)
      const int E = S >> 8;
      uint32_t ord[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        ord[e] = 0u;
        if (e < E) {
          const uint32_t p = (uint32_t)(t * E + e);
          const uint32_t cnt = s_cnt[p];
          if (p >= p0 && p < p1 && cnt >= 1u && cnt <= 32u) ord[e] = atomicAdd(&s_misc[10 + size_class(cnt)], 1u);
        }
      }
      __syncthreads();
      uint32_t cb = 0;
#pragma unroll
      for (int c = 0; c < kClasses; c++) {
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_misc[10 + c]);
        cbase[c] = cb, ncls[c] = n;
        cb += (((n << c) + 31u) >> 5) << 5;
      }
      if (l == 0) {
#pragma unroll
        for (int c = 0; c < kClasses; c++) s_wt[wv * 24 + 8 + c] = cbase[c];
      }
      wave_lds_sync();
#pragma unroll
      for (int e = 0; e < 8; e++) {
        if (e < E) {
          const uint32_t p = (uint32_t)(t * E + e);
          const uint32_t cnt = s_cnt[p];
          s_cur[p] = 0u;
          if (p >= p0 && p < p1 && cnt >= 1u && cnt <= 32u) {
            const uint32_t c = size_class(cnt);
            s_slot[p] = s_wt[wv * 24 + 8 + c] + (ord[e] << c);
          }
        }
      }
      __syncthreads();
      return p1;
    };
    // (one segment: the layout came out of the scans; otherwise it is rebuilt per segment with LDS atomics)
    uint32_t p1 = npad > cap ? open_segment() : (uint32_t)S;
    // ---- pass 2: every point of the segment's pillars to its LDS slot, decorated (pe:116-123); the points of big pillars to the
    // 64-byte record stream (with the first segment)
    auto place = [&](const uint4& a, const uint4& c, const bool with_big) {
      const uint32_t rl = c.w;
      const uint32_t cnt = s_cnt[rl];
      const bool big = cnt > 32u;
      if (big ? !with_big : (rl < p0 || rl >= p1)) return;
      const uint32_t idx = atomicAdd(&s_cur[rl], 1u);
      if (!big) {
        // raw record to its slot (the tile decorates: 32 instead of 48 bytes of LDS per point = one segment for most bins)
        uint4* d = reinterpret_cast<uint4*>(s_rec + (s_slot[rl] + idx) * kRecW);
        d[0] = a, d[1] = c;
        if (idx == cnt - 1u) {  // the point that completes the pillar also fills the group's spare slots
          const uint32_t G = 1u << size_class(cnt);
          for (uint32_t k = cnt; k < G; k++) {
            d += 2;
            d[0] = a, d[1] = c;
          }
        }
      } else {
        const float* info = reinterpret_cast<const float*>(&s_sum[3 * rl]);
        const float raw[6] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z),
                              __uint_as_float(a.w), __uint_as_float(c.x), __uint_as_float(c.y)};
        float f[12];
#pragma unroll
        for (int k = 0; k < 12; k++) f[k] = 0.f;
#pragma unroll
        for (int k = 0; k < F; k++) f[k] = raw[k];
        f[F + 0] = __fsub_rn(raw[0], info[0]);
        f[F + 1] = __fsub_rn(raw[1], info[1]);
        f[F + 2] = __fsub_rn(raw[2], info[2]);
        f[F + 3] = __fsub_rn(raw[0], info[3]);
        f[F + 4] = __fsub_rn(raw[1], info[4]);
        const uint32_t rem = cnt - 1u - idx;
        const uint32_t aux = min(idx, 0xFFFFu) | (min(rem, 0xFFFFu) << 16);
        f[C0] = 1.f;  // multiplies the folded-BN shift column of W0' (k_fold_bn)
        uint4* d = reinterpret_cast<uint4*>(A.rec64 + (int64_t)(bs + s_gst[rl] + idx) * 16);
        d[0] = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[2]), __float_as_uint(f[4]), __float_as_uint(f[6]));
        d[1] = make_uint4(__float_as_uint(f[8]), __float_as_uint(f[10]), aux, (uint32_t)(r0 + rl));
        d[2] = make_uint4(__float_as_uint(f[1]), __float_as_uint(f[3]), __float_as_uint(f[5]), __float_as_uint(f[7]));
        d[3] = make_uint4(__float_as_uint(f[9]), __float_as_uint(f[11]), aux, __float_as_uint(info[5]));
      }
    };
    // the rounds that are still in registers belong to the first segment's pass: the registers die here, in front of the tile loop
#pragma unroll
    for (int it = 0; it < kKeepR; it++)
      if (bs + it * kBinBlock + t < be) place(ka[it], kc[it], true);
    uint32_t jfirst = bs + kKeepR * kBinBlock;
    PNX_TMARK(10);
    for (;;) {
      const bool first = p0 == 0u;
      for (uint32_t j = jfirst + t; j < be; j += kBinBlock) {
        const uint4* q = reinterpret_cast<const uint4*>(binbuf + (int64_t)j * 8);
        place(q[0], q[1], first);
      }
      PNX_TMARK(11);
      __syncthreads();
      __builtin_amdgcn_s_setprio(0);
      if (first) tk = ticket_issue(A.tick, t);  // the next bin's ticket resolves while this bin's tiles are computed
      PNX_TMARK(4);

      // ---- PFN over the segment's tiles: wave wv takes tiles wv, wv + 4, ...
      {
        // a lane's view of one record: the raw point
        struct Rec {
          uint4 a, c;
        };
        auto load_rec = [&](uint32_t slot) -> Rec {
          const uint4* r = reinterpret_cast<const uint4*>(s_rec + slot * kRecW);
          Rec v;
          v.a = r[0];
          v.c = r[1];
          return v;
        };
        // one size class at a time (the class is a compile-time constant inside): global tile g = (tiles in front) + j belongs to wave g & 3
        auto run_class = [&](auto lgc) {
          constexpr int cls = decltype(lgc)::value;
          const uint32_t nslot = ncls[cls] << cls, cb = cbase[cls], nt = (nslot + 31u) >> 5;
          uint32_t j = ((uint32_t)wv - (cb >> 5)) & 3u;  // cb >> 5 = tiles of the classes in front
          Rec nxt;
          if (j < nt) nxt = load_rec(cb + (j << 5) + (uint32_t)col);
          while (j < nt) {
          const Rec cur = nxt;
          const uint32_t left = nslot - (j << 5);
          const uint32_t used = left < 32u ? left : 32u;
          j += 4;
          if (j < nt) nxt = load_rec(cb + (j << 5) + (uint32_t)col);  // the next tile's records under this tile's MFMAs
          const bool act = (uint32_t)col < used;
          const uint32_t rl = act ? cur.c.w : 0u;

          // ---- decoration (pe:116-123): [raw F | xyz - pillar mean | xy - pillar centre]; then the lane's K elements 2kk + h of layer 0
          // (lane = point, registers = channels); K elements beyond the features: the constant 1, then zeros
          float ff[6];
          {
            const float2* ip = reinterpret_cast<const float2*>(&s_sum[3 * rl]);  // {mean x y, mean z centre x, centre y cell}
            const float2 i0 = ip[0], i1 = ip[1], i2 = ip[2];
            const float raw[6] = {__uint_as_float(cur.a.x), __uint_as_float(cur.a.y), __uint_as_float(cur.a.z),
                                  __uint_as_float(cur.a.w), __uint_as_float(cur.c.x), __uint_as_float(cur.c.y)};
            float f[12];
#pragma unroll
            for (int k = 0; k < 12; k++) f[k] = 0.f;
#pragma unroll
            for (int k = 0; k < F; k++) f[k] = raw[k];
            f[F + 0] = __fsub_rn(raw[0], i0.x);
            f[F + 1] = __fsub_rn(raw[1], i0.y);
            f[F + 2] = __fsub_rn(raw[2], i1.x);
            f[F + 3] = __fsub_rn(raw[0], i1.y);
            f[F + 4] = __fsub_rn(raw[1], i2.x);
            f[C0] = 1.f;  // multiplies the folded-BN shift column of W0' (k_fold_bn)
#pragma unroll
            for (int kk = 0; kk < 6; kk++) ff[kk] = h ? f[2 * kk + 1] : f[2 * kk];
          }
          v16f d0;
#pragma unroll
          for (int i = 0; i < 16; i++) d0[i] = 0.f;
#pragma unroll
          for (int kk = 0; kk < KS; kk++) d0 = PNX_MFMA(w0f[kk], act ? ff[kk] : 0.f, d0);
          float u[16], g0[16];
#pragma unroll
          for (int i = 0; i < 16; i++) {
            u[i] = fmaxf(d0[i], 0.f);  // ReLU first: max(relu(x)) == relu(max(x))
            g0[i] = u[i];
          }
          // ---- "max" half of the concat (pe:43-44,49): per-pillar max of relu(layer 0) in every lane of the pillar's group
          group_max<cls, 16>(g0);
          // ---- layer 1, fp16x3 (pfn_v3.hip): hi*hi + hi*lo + lo*hi as 24 v_mfma_f32_32x32x16_f16; K steps 0, 1 = the point's own h0
          // (pre-scaled by 2^SU), K steps 2, 3 = the pillar maximum
          v16f da, db;
          {
            const float4* sp = reinterpret_cast<const float4*>(s_s1 + h * 32);
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float4 sa = sp[j], sb = sp[4 + j];
              da[4 * j] = sa.x, da[4 * j + 1] = sa.y, da[4 * j + 2] = sa.z, da[4 * j + 3] = sa.w;
              db[4 * j] = sb.x, db[4 * j + 1] = sb.y, db[4 * j + 2] = sb.z, db[4 * j + 3] = sb.w;
            }
          }
          uint32_t bh[16], bl[16];
#pragma unroll
          for (int tq = 0; tq < 8; tq++) split2_f16(u[2 * tq], u[2 * tq + 1], bh[tq], bl[tq]);
#pragma unroll
          for (int tq = 0; tq < 8; tq++) split2_f16(g0[2 * tq], g0[2 * tq + 1], bh[8 + tq], bl[8 + tq]);
#define PNX_L1H(S_, PROD)                                                                                                   \
  da = PNX_MFMA16(as_v8h(&wq[((PROD) == 2 ? 32 : 0) + (0 * 4 + (S_)) * 4]), as_v8h((PROD) == 1 ? &bl[4 * (S_)] : &bh[4 * (S_)]), da); \
  db = PNX_MFMA16(as_v8h(&wq[((PROD) == 2 ? 32 : 0) + (1 * 4 + (S_)) * 4]), as_v8h((PROD) == 1 ? &bl[4 * (S_)] : &bh[4 * (S_)]), db);
          PNX_L1H(0, 0) PNX_L1H(0, 1) PNX_L1H(0, 2) PNX_L1H(1, 0) PNX_L1H(1, 1) PNX_L1H(1, 2)
          PNX_L1H(2, 0) PNX_L1H(2, 1) PNX_L1H(2, 2) PNX_L1H(3, 0) PNX_L1H(3, 1) PNX_L1H(3, 2)
#undef PNX_L1H
          // the pillar maximum bounds every value of the pillar: one range test covers both operand halves
          float gm = fmaxf(fmaxf(fmaxf(g0[0], g0[1]), fmaxf(g0[2], g0[3])), fmaxf(fmaxf(g0[4], g0[5]), fmaxf(g0[6], g0[7])));
          gm = fmaxf(gm, fmaxf(fmaxf(fmaxf(g0[8], g0[9]), fmaxf(g0[10], g0[11])), fmaxf(fmaxf(g0[12], g0[13]), fmaxf(g0[14], g0[15]))));
          // A PILLAR whose layer-0 maximum leaves the fp16 range goes to k_pfn3_tail (fp32 MFMA, unscaled weights) through the 64-byte
          // record stream, in the format of reader_bins.h; its lanes compute garbage in their own MFMA columns only, and its row is
          // not stored here.  (Per pillar, not per tile: which pillars share a tile is nobody's business.)
          bool pov = false;
          const uint64_t ovm = __ballot(act && !(gm < 60000.f));
          if (ovm != 0ull) {
            pov = (((uint32_t)ovm | (uint32_t)(ovm >> 32)) >> col) & 1u;  // either half of the pillar's channels
            const uint32_t cnt = (act && pov) ? s_cnt[rl] : 0u;
            const uint32_t idx = (uint32_t)col & ((1u << cls) - 1u);
            if (idx < cnt) {  // the spare slots of a group are not points
              const float* info = reinterpret_cast<const float*>(&s_sum[3 * rl]);
              const uint32_t gslot = bs + s_gst[rl] + idx;
              const uint32_t aux = idx | ((cnt - 1u - idx) << 16);
              uint4* d = reinterpret_cast<uint4*>(A.rec64 + (int64_t)gslot * 16 + 8 * h);
              d[0] = make_uint4(__float_as_uint(ff[0]), __float_as_uint(ff[1]), __float_as_uint(ff[2]), __float_as_uint(ff[3]));
              d[1] = make_uint4(__float_as_uint(ff[4]), __float_as_uint(ff[5]), aux, h ? __float_as_uint(info[5]) : (uint32_t)(r0 + rl));
              if (idx == 0u && h == 0) {
                const int64_t gr = r0 + rl;
                A.pfirst[gr] = gslot;
                A.pcnt[gr] = cnt;
                A.cell_of_pillar[gr] = __float_as_int(info[5]);
                const int at = atomicAdd(&A.counters[4], 1);
                if (at < A.bigcap) A.biglist[A.bigcap + at] = (int)gr;
              }
            }
          }
          // the accumulators carry the scale 2^(SU+SW) (shift included): an exact power of two
          constexpr float kDs = 1.0f / (float)(1 << (PNX_PFN_SU + PNX_PFN_SW));
          // ---- per-pillar max of relu(layer 1 + shift): the same butterfly; then the first lane of every group (of each half) writes
          // the finished row in NATURAL channel order: accumulator registers 4j..4j+3 of half h are channels 8j + 4h .. +3 (da) /
          // 32 + those (db)
          float pa[16], pb[16];
#pragma unroll
          for (int i = 0; i < 16; i++) {
            pa[i] = fmaxf(__fmul_rn(da[i], kDs), 0.f);
            pb[i] = fmaxf(__fmul_rn(db[i], kDs), 0.f);
          }
          group_max<cls, 16>(pa);
          group_max<cls, 16>(pb);
          const bool lead = act && ((uint32_t)col & ((1u << cls) - 1u)) == 0u;
          const int pid = col >> cls;           // pillar of this lane inside the tile
          const int npil = (int)(used >> cls);  // whole pillars in the tile
          if (PACK) {
            // 16-bit canvas and no fp32 feat_max output: round-to-nearest-even of the maximum == maximum of the rounded values
            if (lead) {
              uint32_t* dst = s_out + pid * kZSP + 2 * h;
#pragma unroll
              for (int j = 0; j < 4; j++) {
                *reinterpret_cast<uint2*>(dst + 4 * j) = make_uint2(cvt_pk16<DT>(pa[4 * j], pa[4 * j + 1]), cvt_pk16<DT>(pa[4 * j + 2], pa[4 * j + 3]));
                *reinterpret_cast<uint2*>(dst + 16 + 4 * j) = make_uint2(cvt_pk16<DT>(pb[4 * j], pb[4 * j + 1]), cvt_pk16<DT>(pb[4 * j + 2], pb[4 * j + 3]));
              }
              if (h == 1) s_cellrow[pid] = pov ? 0xFFFFFFFFu : __float_as_uint(reinterpret_cast<const float*>(&s_sum[3 * rl])[5]);  // where the row goes
            }
            wave_lds_sync();
            // ---- stores: lane -> (pillar l>>3 + 8*it, 16 bytes = channels 8*(l&7) .. +7): one instruction writes 8 complete 128-byte lines
            const int qq = l & 7;
            for (int p = l >> 3; p < npil; p += 8) {
              const uint4 x = *reinterpret_cast<const uint4*>(s_out + p * kZSP + 4 * qq);
              const int32_t cl = (int32_t)s_cellrow[p];
              if (cl >= 0) *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(out.canvas) + (int64_t)cl * 64 + 8 * qq) = x;
            }
          } else {
            if (lead) {
              uint32_t* dst = s_out + pid * kZS + 4 * h;
#pragma unroll
              for (int j = 0; j < 4; j++) {
                *reinterpret_cast<uint4*>(dst + 8 * j) =
                    make_uint4(__float_as_uint(pa[4 * j]), __float_as_uint(pa[4 * j + 1]), __float_as_uint(pa[4 * j + 2]), __float_as_uint(pa[4 * j + 3]));
                *reinterpret_cast<uint4*>(dst + 32 + 8 * j) =
                    make_uint4(__float_as_uint(pb[4 * j]), __float_as_uint(pb[4 * j + 1]), __float_as_uint(pb[4 * j + 2]), __float_as_uint(pb[4 * j + 3]));
              }
              if (h == 0) s_rank[pid] = (uint32_t)(r0 + rl);  // where the row goes
              else s_cellrow[pid] = pov ? 0xFFFFFFFFu : __float_as_uint(reinterpret_cast<const float*>(&s_sum[3 * rl])[5]);
            }
            wave_lds_sync();
            // ---- stores: lane -> (pillar l>>3 + 8*it, channels 8*(l&7) .. +7)
            const int qq = l & 7;
            for (int p = l >> 3; p < npil; p += 8) {
              const uint4* src = reinterpret_cast<const uint4*>(s_out + p * kZS + 8 * qq);
              const uint4 x0 = src[0], x1 = src[1];
              const float v[8] = {__uint_as_float(x0.x), __uint_as_float(x0.y), __uint_as_float(x0.z), __uint_as_float(x0.w),
                                  __uint_as_float(x1.x), __uint_as_float(x1.y), __uint_as_float(x1.z), __uint_as_float(x1.w)};
              const int32_t cl = (int32_t)s_cellrow[p];
              if (cl >= 0) store_chunk<DT>(out, (int)s_rank[p], (int64_t)cl, qq, v);
            }
          }
          wave_lds_sync();  // the next tile rewrites the rows
          PNX_TCOUNT(8, 1);
          }
        };
        run_class(std::integral_constant<int, 0>{});
        run_class(std::integral_constant<int, 1>{});
        run_class(std::integral_constant<int, 2>{});
        run_class(std::integral_constant<int, 3>{});
        run_class(std::integral_constant<int, 4>{});
        run_class(std::integral_constant<int, 5>{});
      }
      PNX_TMARK(5);
      __syncthreads();  // the records, cursors and pillar constants are rewritten by the next segment / bin
      PNX_TMARK(6);
      if (p1 >= (uint32_t)S) break;
      __builtin_amdgcn_s_setprio(2);
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pst[p1]);
      p0 = p1;
      jfirst = bs;
      p1 = open_segment();
    }
  }
#ifdef PNX_BINS_TIMERS
  if (l == 0 && A.timers != nullptr)
    for (int k = 0; k < 16; k++) atomicAdd(&A.timers[k], s_tim[wv * 16 + k]);
#endif
}

template <int F>
int launch_bins(const BinPfnArgs& A0, const Pfn3Out& out, const PnxGeomDev& g, const PnxFillJob& fj, int64_t n, hipStream_t st) {
  BinPfnArgs A = A0;
  const bool pack = out.g1 == nullptr && out.canvas != nullptr && out.dt != PNX_F32;
  // LDS: two workgroups per CU (the kernel's ~216 VGPRs allow two waves per SIMD): 160 KiB / 2 minus a margin for a co-resident
  // zero-fill kernel's few bytes
  const char* l_env = getenv("PNX_BINS_LDS");
  const size_t budget = l_env ? (size_t)atoi(l_env) : 80 * 1024 - 512;
  const size_t fixed = bin_pfn_lds_bytes(A.sh, 0, pack);
  PNX_REQUIRE(fixed + (size_t)(kClassSlack + 64) * kRecW * 4 <= budget, PNX_ERR_UNSUPPORTED, "bins of 2^%d pillars do not fit the LDS budget", A.sh);
  int cap = (int)((budget - fixed) / (kRecW * 4));
  const char* c_env = getenv("PNX_BINS_CAP");  // experiments / tests: force multi-segment bins
  if (c_env && atoi(c_env) >= 32 && atoi(c_env) + kClassSlack < cap) cap = atoi(c_env) + kClassSlack;
  A.cap = cap;
  const size_t lds = bin_pfn_lds_bytes(A.sh, cap, pack);
  const char* b_env = getenv("PNX_PFN_BLOCKS");
  const int max_blocks = b_env ? atoi(b_env) : 512;  // 256 CUs x 2 workgroups
  int nb = A.K1 < max_blocks ? A.K1 : max_blocks;
  if (n <= 0) nb = 0;
  const int grid = nb + A.n_fill;
  if (grid <= 0) return PNX_OK;
#define PNX_GO(DT_, PACK_)                                                                                                              \
  {                                                                                                                                     \
    static size_t lds_set = 0;                                                                                                          \
    if (lds > lds_set) {                                                                                                                \
      PNX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_pfn<F, DT_, PACK_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      lds_set = lds;                                                                                                                    \
    }                                                                                                                                   \
    k_bin_pfn<F, DT_, PACK_><<<grid, kBinBlock, lds, st>>>(A, out, g, fj);                                                              \
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
  return PNX_OK;
}

}  // namespace

// The fused bin sort + PFN launch.  tick[0] must be zero (the reader's memset), counters as left by the grouping kernels.
// n_fill > 0: blocks [0, n_fill) of the launch take the zero-fill tiles of `fj` (pnx_fill.h) concurrently.
int pnx_launch_bin_pfn(int F, const uint32_t* binbuf, const uint32_t* hpre, const uint32_t* hblk, int64_t matlen, int sh, int nwg, int K1,
                       int32_t* counters, int32_t* tick, uint32_t* rec64, uint32_t* pfirst, uint32_t* pcnt, int32_t* cell_of_pillar, int32_t* coords,
                       int64_t pillar_capacity, int write_pillars, int32_t* biglist, int64_t bigcap, const float* folded, float* g1, int64_t g1_rows,
                       void* canvas, int canvas_dt, int64_t n_points, int n_fill, const PnxGeomDev& geom, const PnxFillJob& fj, hipStream_t st) {
  BinPfnArgs A;
  A.binbuf = binbuf, A.hpre = hpre, A.hblk = hblk, A.counters = counters, A.tick = tick, A.rec64 = rec64, A.pfirst = pfirst, A.pcnt = pcnt;
  A.cell_of_pillar = cell_of_pillar, A.coords = coords, A.pillar_capacity = pillar_capacity, A.biglist = biglist;
  A.bigcap = (int)(bigcap > 0x7fffffff ? 0x7fffffff : bigcap);
  A.timers = nullptr;
#ifdef PNX_BINS_TIMERS
  static unsigned long long* d_tim = nullptr;
  if (d_tim == nullptr) PNX_CHECK_HIP(hipMalloc(&d_tim, 16 * sizeof(unsigned long long)));
  PNX_CHECK_HIP(hipMemsetAsync(d_tim, 0, 16 * sizeof(unsigned long long), st));
  A.timers = d_tim;
#endif
  A.P = folded, A.sh = sh, A.nwg = nwg, A.K1 = K1, A.cap = 0, A.n_fill = n_fill, A.write_pillars = write_pillars, A.matlen = matlen;
  Pfn3Out out;
  out.g1 = g1, out.g1_rows = g1_rows, out.canvas = canvas, out.dt = canvas_dt;
  int rc;
  switch (F) {
    case 3: rc = launch_bins<3>(A, out, geom, fj, n_points, st); break;
    case 4: rc = launch_bins<4>(A, out, geom, fj, n_points, st); break;
    case 5: rc = launch_bins<5>(A, out, geom, fj, n_points, st); break;
    default: pnx_set_error("LDS-sorted PFN is built for 3..5 point features, got %d", F); return PNX_ERR_UNSUPPORTED;
  }
#ifdef PNX_BINS_TIMERS
  if (rc == PNX_OK && getenv("PNX_BINS_TIMERS_PRINT")) {
    unsigned long long h_tim[16];
    PNX_CHECK_HIP(hipMemcpyAsync(h_tim, d_tim, sizeof(h_tim), hipMemcpyDeviceToHost, st));
    PNX_CHECK_HIP(hipStreamSynchronize(st));
    static const char* nm[12] = {"ticket+top barrier", "range+clears", "pass1", "scans", "pass2 barrier", "tiles", "end barrier", "weights", "#tiles", "#bins x4", "layout+place regs", "place loop"};
    unsigned long long tot = h_tim[10] + h_tim[11];
    for (int k = 0; k < 8; k++) tot += h_tim[k];
    fprintf(stderr, "[pnx bins timers] wave-cycles:");
    for (int k = 0; k < 12; k++) fprintf(stderr, " %s=%llu(%.1f%%)", nm[k], h_tim[k], (k < 8 || k > 9) ? 100.0 * h_tim[k] / (tot ? tot : 1) : 0.0);
    fprintf(stderr, "  cycles/tile=%.0f\n", h_tim[8] ? (double)h_tim[5] / h_tim[8] : 0.0);
  }
#endif
  return rc;
}
