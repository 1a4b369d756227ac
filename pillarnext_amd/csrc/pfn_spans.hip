// pfn_spans.hip -- pillar grouping INSIDE LDS + PFN + canvas stores for one span of the canvas per workgroup (gfx950); consumer of the
// chunk-sorted records of chunk_sort.hip (vocabulary in spans.h).  Reference semantics: pillar_encoder.py:106-123 (torch.unique
// inverse, scatter_mean, feature decoration), :35-50 x2 (PFNLayer), :174-182 (PillarFeatureNet.forward) and the dense canvas of
// sparse_resnet.py:63-68.
//
// One workgroup owns one span (<= 16 consecutive slabs of 512 canvas cells of one frame, <= 512 pillars) and does, per span:
//   gather   the span's records are ~one short run per chunk of the frame: thread i reads the two table entries of row i (run start and
//            end) and walks its run -- the first records stay in registers for the later passes, the rest is re-read from L2
//   rank     every record ORs its cell into the span's occupancy bitmap in LDS; a popcount prefix over the 256 words numbers the span's
//            pillars in cell order (torch.unique order inside the span) -- the per-point rank lookup of rounds 2-3 is gone, and with it
//            the global bitmap scan on the canvas path
//   pass 1   points per pillar + exact fp64 coordinate sums with LDS atomics
//   scan     exclusive scans over the pillar counts; per pillar: mean (fp32 divide of the fp64 sum, pe:113-114), pillar centre
//            (pe:119-120), canvas cell; with rank outputs (feat_max / coords): the pillar's global rank from the reader's bitmap prefix
//   layout / pass 2 / PFN   exactly the size-class tiles of round 3: a pillar of n <= 32 points gets an aligned group of 1..32 LDS
//            slots, a tile = 32 slots of one class, layer 0 fp32 MFMA, per-pillar max by unmasked DPP butterflies, layer 1 fp16x3 MFMA,
//            finished 128-byte lines stored straight into the NHWC canvas (or feat_max rows)
// A span whose padded points exceed the LDS record slots is processed in pillar-aligned segments.  Pillars of > 32 points and pillars
// whose layer-0 maximum leaves the fp16x3 range go to the 64-byte record stream (slots handed out by a counter) and are listed for
// k_pfn3_tail under a spill id.  Spans are handed out in canvas order by a ticket counter: the pillar lines of neighbouring spans fall
// into neighbouring DRAM pages.  Nothing computed depends on where a pillar sits in a tile or on the order of its records.
#include <type_traits>

#include "pnx_common.h"
#include "pfn_common.h"
#include "spans.h"

namespace {

constexpr int kRecW = 8;      // words per LDS record: the raw 32-byte record [x y z f3 | f4 f5 point pillar]
constexpr int kKeepR = 4;     // records of a thread's first run kept in registers between the passes
constexpr int kSpBlock = 256;
constexpr int kClasses = 6;   // group sizes 1, 2, 4, 8, 16, 32
constexpr int kClassSlack = kClasses * 32;  // every class region is rounded up to whole tiles
constexpr int kS = kSpanPillars;
constexpr int kBitWords = kSpanMaxSlabs * kSlabCells / 32;  // 256
static_assert(kBitWords == kSpBlock && (kS % kSpBlock) == 0, "one bitmap word and kS / 256 pillars per thread");

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

struct SpanPfnArgs {
  SpanTables T;
  SpanGeom sg;
  int32_t* counters;             // spans.h: kCnt*
  int32_t* tick;                 // span tickets
  uint32_t* rec64;               // 64-byte record stream: only spilled pillars are written
  uint32_t *pfirst, *pcnt;       // per spill id
  int32_t* cell_of_pillar;
  int32_t* row_of;               // feat_max row (global pillar rank) per spill id; rank outputs only
  int32_t* biglist;
  int bigcap, idcap;
  const uint2* wcomb;            // {bitmap word, popcount prefix} of the reader's key-order bitmap: rank outputs only (else null)
  const uint32_t* wblk;
  int32_t* coords;
  int64_t pillar_capacity;
  const float* P;                // folded parameters (k_fold_bn)
  int cap;
  unsigned long long* timers;
};

// size class of a pillar of 1..32 points: log2 of the next power of two
__device__ __forceinline__ uint32_t size_class(uint32_t cnt) { return cnt <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(cnt - 1u); }

template <bool PACK>
constexpr int wave_out_words() {
  return (PACK ? 32 * kZSP : 32 * kZS) + 64;
}
static inline size_t span_pfn_lds_bytes(int cap, bool pack, int B) {
  const size_t head = (2 * (kS + 4) + 3 * kS) * 4 + 3 * kS * 8 + 2 * kBitWords * 4 + (48 + 4 * 24 + 64) * 4 + 3 * (size_t)((B + 1 + 3) & ~3) * 4;
  return head + 4 * (size_t)(pack ? wave_out_words<true>() : wave_out_words<false>()) * 4 + (size_t)cap * kRecW * 4;
}

template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// all-reduce max inside aligned groups of 2^LG lanes (of each 32-lane half): xor butterfly, one DPP max per step
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
    for (int i = 0; i < N; i++) v[i] = fmaxf(v[i], dpp_perm<0x141>(v[i]));  // row_half_mirror
  }
  if (LG >= 4) {
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = fmaxf(v[i], dpp_perm<0x140>(v[i]));  // row_mirror
  }
  if (LG >= 5) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < N; i++) {
      const uint32_t x = __float_as_uint(v[i]);
      const u32x2 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
      v[i] = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
  }
}

// rank of a cell in the reader's key order (pillar_encoder.py:109-110: unique rows of [b, xi, yi] sorted): key = (b*gx + xi)*gyp + yi
__device__ __forceinline__ int32_t key_rank(int32_t key, const uint2* __restrict__ wcomb, const uint32_t* __restrict__ wblk) {
  const int32_t w = key >> 5;
  const uint2 c = wcomb[w];
  return (int32_t)(wblk[w >> PNX_SCAN_SHIFT] + c.y + __popc(c.x & ((1u << (key & 31)) - 1u)));
}

// Registers: the kernel is capped at 240 per lane (amdgpu_num_vgpr counts the arch half of gfx950's unified file: 120), 16-24 bytes of
// scratch per lane instead of 256 registers and none -- so that the two workgroups of a CU leave 32 registers per SIMD lane, which is what
// a wave of the zero-fill kernel (reader.hip: k_canvas_fill_bytes, 25 registers, no LDS to speak of) needs to be co-resident: the fill
// runs on a second stream BESIDE this kernel, one workgroup per CU, instead of taking half of this kernel's workgroup slots
// (PNX_SPAN_NUM_VGPR at build time for experiments; 128 = no cap).
#ifndef PNX_SPAN_NUM_VGPR
#define PNX_SPAN_NUM_VGPR 120
#endif
template <int F, int DT, bool PACK>
__global__ __launch_bounds__(kSpBlock, 2) __attribute__((amdgpu_num_vgpr(PNX_SPAN_NUM_VGPR))) void k_span_pfn(SpanPfnArgs A, Pfn3Out out, PnxGeomDev g) {
  constexpr int C0 = F + 5, KS = (C0 + 2) / 2;     // K = C0 features + the constant-1 column that carries the folded BN shift
  constexpr int FR = 32 * C0 + 32 + 64 * 64 + 64;  // start of the fragment-ordered block (k_fold_bn)
  constexpr int WL = wave_out_words<PACK>();
  constexpr int S = kS, E = kS >> 8;
  extern __shared__ __align__(16) uint32_t s_raw[];
  const int t = threadIdx.x;
  uint32_t* s_cnt = s_raw;               // S + 1: points per pillar
  uint32_t* s_pst = s_cnt + (S + 4);     // S + 1: exclusive starts of the PADDED sizes of the pillars of <= 32 points
  uint32_t* s_slot = s_pst + (S + 4);    // S: first LDS slot of the pillar in the current segment (pillars of > 32 points: first slot of the spill stream)
  uint32_t* s_cur = s_slot + S;          // S cursors
  uint32_t* s_key = s_cur + S;           // S: canvas cell of the pillar; after the scans its feat_max row (rank outputs)
  double* s_sum = reinterpret_cast<double*>(s_key + S);  // 3 doubles per pillar; later {mean x y z, centre x y, cell} as 6 words
  uint32_t* s_bits = reinterpret_cast<uint32_t*>(s_sum + 3 * S);  // occupancy of the span's cells
  uint32_t* s_bpre = s_bits + kBitWords;                           // popcount prefix inside each 64-word quarter
  uint32_t* s_misc = s_bpre + kBitWords;  // [0..7] wave sums [8] ticket [9] segment end [10..15] pillars per class [16..39] per wave [44..45] tickets [46] run cursor
  uint32_t* s_wt = s_misc + 48;  // per wave: [0..5] first slot of the wave's pillars per class, [8..13] first slot of the class, [16..21] pillars of the class
  float* s_s1 = reinterpret_cast<float*>(s_wt + 4 * 24);  // 2 x 32 pre-scaled layer-1 shifts
  uint32_t* s_fb = s_misc + 48 + 4 * 24 + 64;              // B + 1: spans in front of every frame
  uint32_t* s_flo = s_fb + ((A.sg.B + 1 + 3) & ~3);         // B: first table row that can hold records of the frame
  uint32_t* s_fnp = s_flo + ((A.sg.B + 1 + 3) & ~3);        // B: number of such rows among the chunks' own rows
  uint32_t* s_outb = s_fnp + ((A.sg.B + 1 + 3) & ~3);
  uint32_t* s_rec = s_outb + 4 * WL;

  const int l = t & 63, col = l & 31, h = l >> 5, wv = __builtin_amdgcn_readfirstlane(t >> 6);
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
  {
    const float* __restrict__ s1lane = A.P + FR + 64 * 89 + l * 32;
    if (wv == 0 && col == 0)
      for (int i = 0; i < 32; i++) s_s1[h * 32 + i] = __fmul_rn(s1lane[i], (float)(1 << (PNX_PFN_SU + PNX_PFN_SW)));
  }
  // spans in front of every frame (one wave, B <= 1024)
  const int B = A.sg.B, nf = A.sg.nf;
  if (wv == 0) {
    uint32_t carry = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
      const uint32_t v = b0 + l < B ? (uint32_t)A.T.nspan[b0 + l] : 0u;
      uint32_t inc = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(inc, d);
        if (l >= d) inc += y;
      }
      if (b0 + l < B) s_fb[b0 + l] = carry + inc - v;
      carry += __shfl(inc, 63);
    }
    if (l == 0) s_fb[B] = carry;
  }
  for (int bb = t; bb < B; bb += kSpBlock) {
    const int hi_e = A.T.frame_hi[bb], rlo = A.sg.nchunks - A.T.frame_lo[bb];
    s_flo[bb] = (uint32_t)rlo;
    s_fnp[bb] = hi_e > 0 ? (uint32_t)(hi_e - rlo) : 0u;
  }
  __syncthreads();
  const int nspans = (int)s_fb[B];
  const bool ranked = A.wcomb != nullptr;
  const uint32_t cap = (uint32_t)A.cap - (uint32_t)kClassSlack;  // padded points of a segment; the class regions add at most the slack
  const uint4* __restrict__ recs = A.T.recs;
  int novf = A.counters[kCntRows];
  novf = novf < A.sg.ovf_cap ? novf : A.sg.ovf_cap;
  int tk = ticket_issue(A.tick, t);  // thread 0 only; the value is read by publish_ticket
  // The next span's ticket is published to LDS in front of the LAST barrier of the current span (two slots in turn: a wave may still be
  // reading this span's slot while wave 0 writes the next one): a span has no barrier of its own at the top.
  auto publish_ticket = [&](int iter_next) {
    if (wv == 0) {
      const int bq = ticket_wait(tk);
      if (t == 0) s_misc[44 + (iter_next & 1)] = (uint32_t)bq;
    }
  };
  publish_ticket(0);
  if (t == 0) s_misc[46] = 0u;  // run cursor of the index list (zero at the top of every span)
  __syncthreads();
  PNX_TMARK(7);
  for (int iter = 0;; iter++) {
    __builtin_amdgcn_s_setprio(2);
    const int k = __builtin_amdgcn_readfirstlane((int)s_misc[44 + (iter & 1)]);
    PNX_TMARK(0);
    if (k >= nspans) break;
    // frame of the span
    int b = 0;
    {
      int lo = 0, hi = B;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)s_fb[mid] <= k) lo = mid;
        else hi = mid;
      }
      b = __builtin_amdgcn_readfirstlane(lo);
    }
    const uint2* __restrict__ sd = A.T.span_desc + (int64_t)b * (nf + 1) + (k - (int)s_fb[b]);
    const uint2 d0 = sd[0], d1 = sd[1];  // one round trip: {first slab, points in front} of this span and of the next
    const int fa = __builtin_amdgcn_readfirstlane((int)d0.x), fz = __builtin_amdgcn_readfirstlane((int)d1.x);
    const uint32_t nrec = (uint32_t)__builtin_amdgcn_readfirstlane((int)(d1.y - d0.y));  // points of the span
    if (nrec == 0u) {  // an empty span costs one barrier
      tk = ticket_issue(A.tick, t);
      publish_ticket(iter + 1);
      __syncthreads();
      continue;
    }
    PNX_TCOUNT(9, 1);
    const int32_t fcell = b * A.sg.cpf;                 // first cell of the frame
    const int32_t cell0 = fcell + (fa << kSlabShift);    // first cell of the span
    // rows that can hold records of this frame: the chunks' own rows [rlo, rlo + npr) and the overflow pool
    const int rlo = __builtin_amdgcn_readfirstlane((int)s_flo[b]);
    const int npr = __builtin_amdgcn_readfirstlane((int)s_fnp[b]);
    const int nrows = npr + novf;
    auto run_of = [&](int ri, uint32_t& src, uint32_t& len) {
      const int r = ri < npr ? rlo + ri : A.sg.nchunks + (ri - npr);
      const uint16_t* __restrict__ tr = A.T.tab + (int64_t)r * A.sg.tabw;
      const int rf = A.T.rowframe[r];  // the four loads are independent: one round trip
      const uint32_t s = tr[fa], e = tr[fz], rb = A.T.rowbase[r];
      const bool mine = rf == b;
      src = mine ? rb + s : 0u;
      len = mine ? e - s : 0u;
    };
    uint32_t src0 = 0u, len0 = 0u;
    if (t < nrows) run_of(t, src0, len0);
    // The runs are short and uneven (thread i walking run i would pay one L2 latency per record): with one run per thread and few enough
    // records, an INDEX LIST in LDS (the rows of the tile phase are free until then) deals the records round-robin to the threads --
    // record j of the span = record s_idx[j] of the chunk pieces -- and all loads of a pass are in flight together.
    uint32_t* s_idx = s_outb;
    const bool indexed = nrows <= kSpBlock && nrec <= (uint32_t)(4 * WL);  // block-uniform
    // where the run's indices go: any order will do, so an LDS cursor hands out the ranges (no scan, no barrier)
    uint32_t off0 = 0u;
    if (indexed && len0 > 0u) off0 = atomicAdd(&s_misc[46], len0);
    auto write_index = [&]() {
      if (indexed)
        for (uint32_t q = 0; q < len0; q++) s_idx[off0 + q] = src0 + q;
    };

    // A span normally holds <= S pillars and is done in one pass; one with more (nearly every point a pillar of its own) takes a pass
    // per slice of S pillar ranks, each pass from the top (the bitmap and its prefix come out the same every time).
    uint32_t nslices = 1u;
    for (uint32_t slice = 0u; slice < nslices; slice++) {
    const uint32_t rbase = slice * (uint32_t)S;
    for (int p = t; p < S; p += kSpBlock) {
      s_cnt[p] = 0u;
      s_sum[3 * p + 0] = 0.0;
      s_sum[3 * p + 1] = 0.0;
      s_sum[3 * p + 2] = 0.0;
    }
    s_bits[t] = 0u;  // kBitWords == kSpBlock
    write_index();
    __syncthreads();
    PNX_TMARK(1);
    // ---- pass 0: the first records into registers (indexed: records t, t + 256, ...; else the head of the thread's run); every
    // record's cell into the occupancy bitmap
    uint4 ka[kKeepR], kc[kKeepR];
    auto kept = [&](int it) -> bool { return indexed ? (uint32_t)(it * kSpBlock + t) < nrec : (uint32_t)it < len0; };
    auto mark = [&](uint32_t cellw) {
      const uint32_t lc = cellw - (uint32_t)cell0;
      atomicOr(&s_bits[lc >> 5], 1u << (lc & 31u));
    };
#pragma unroll
    for (int it = 0; it < kKeepR; it++) {
      ka[it] = make_uint4(0u, 0u, 0u, 0u);
      kc[it] = make_uint4(0u, 0u, 0u, 0u);
      if (kept(it)) {
        const uint32_t at = indexed ? s_idx[it * kSpBlock + t] : src0 + it;
        const uint4* q = recs + (int64_t)at * 2;
        ka[it] = q[0];
        kc[it] = q[1];
      }
    }
#pragma unroll
    for (int it = 0; it < kKeepR; it++)
      if (kept(it)) mark(kc[it].w);
    if (indexed) {
      for (uint32_t j = kKeepR * kSpBlock + t; j < nrec; j += kSpBlock) mark(recs[(int64_t)s_idx[j] * 2 + 1].w);
    } else {
      for (uint32_t q = kKeepR; q < len0; q++) mark(recs[(int64_t)(src0 + q) * 2 + 1].w);
      for (int ri = t + kSpBlock; ri < nrows; ri += kSpBlock) {
        uint32_t src, len;
        run_of(ri, src, len);
        for (uint32_t q = 0; q < len; q++) mark(recs[(int64_t)(src + q) * 2 + 1].w);
      }
    }
    __syncthreads();
    // popcount prefix over the 256 words: EVERY wave computes all of it (lane i: words 4i .. 4i + 3) and stores the same values --
    // no barrier between the scan and its first use
    uint32_t npil;
    {
      const uint4 bw = reinterpret_cast<const uint4*>(s_bits)[l];
      const uint32_t c0 = (uint32_t)__popc(bw.x), c1 = (uint32_t)__popc(bw.y), c2 = (uint32_t)__popc(bw.z), c3 = (uint32_t)__popc(bw.w);
      const uint32_t pc = c0 + c1 + c2 + c3;
      uint32_t inc = pc;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(inc, d);
        if (l >= d) inc += y;
      }
      const uint32_t ex = inc - pc;
      reinterpret_cast<uint4*>(s_bpre)[l] = make_uint4(ex, ex + c0, ex + c0 + c1, ex + c0 + c1 + c2);
      npil = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);  // pillars of the span
      wave_lds_sync();
    }
    nslices = (npil + (uint32_t)S - 1u) / (uint32_t)S;
    // pillar of a cell inside the slice = set bits below it in the span - first rank of the slice (>= S: not in this slice)
    auto pillar_of = [&](uint32_t cellw) -> uint32_t {
      const uint32_t lc = cellw - (uint32_t)cell0, w = lc >> 5;
      return s_bpre[w] + (uint32_t)__popc(s_bits[w] & ((1u << (lc & 31u)) - 1u)) - rbase;
    };
    // the records beyond the register-kept ones (with_head: those too), from L2: fn(a, c) with c.w = the pillar inside the span
    auto walk_l2 = [&](const bool with_head, auto&& fn) {
      if (indexed) {
        for (uint32_t j = (with_head ? 0u : (uint32_t)(kKeepR * kSpBlock)) + t; j < nrec; j += kSpBlock) {
          const uint4* p = recs + (int64_t)s_idx[j] * 2;
          uint4 a = p[0], c = p[1];
          c.w = pillar_of(c.w);
          if (c.w < (uint32_t)S) fn(a, c);
        }
        return;
      }
      for (uint32_t q = with_head ? 0u : (uint32_t)kKeepR; q < len0; q++) {
        const uint4* p = recs + (int64_t)(src0 + q) * 2;
        uint4 a = p[0], c = p[1];
        c.w = pillar_of(c.w);
        if (c.w < (uint32_t)S) fn(a, c);
      }
      for (int ri = t + kSpBlock; ri < nrows; ri += kSpBlock) {
        uint32_t src, len;
        run_of(ri, src, len);
        for (uint32_t q = 0; q < len; q++) {
          const uint4* p = recs + (int64_t)(src + q) * 2;
          uint4 a = p[0], c = p[1];
          c.w = pillar_of(c.w);
          if (c.w < (uint32_t)S) fn(a, c);
        }
      }
    };
    // ---- pass 1: points per pillar, exact coordinate sums (scatter_mean numerator, pe:113), the pillar's cell
#pragma unroll
    for (int it = 0; it < kKeepR; it++)
      if (kept(it)) kc[it].w = pillar_of(kc[it].w);
    auto tally = [&](const uint4& a, const uint4& c) {
      const uint32_t rl = c.w;
      atomicAdd(&s_cnt[rl], 1u);
      atomicAdd(&s_sum[3 * rl + 0], (double)__uint_as_float(a.x));
      atomicAdd(&s_sum[3 * rl + 1], (double)__uint_as_float(a.y));
      atomicAdd(&s_sum[3 * rl + 2], (double)__uint_as_float(a.z));
    };
#pragma unroll
    for (int it = 0; it < kKeepR; it++)
      if (kept(it) && kc[it].w < (uint32_t)S) tally(ka[it], kc[it]);
    walk_l2(false, tally);
    // the cell of every pillar: the set bits of the bitmap in order (thread = word)
    {
      uint32_t bits = s_bits[t];
      uint32_t rk = s_bpre[t];
      while (bits != 0u) {
        const uint32_t bpos = (uint32_t)__builtin_ctz(bits);
        bits &= bits - 1u;
        const uint32_t r2 = rk++ - rbase;
        if (r2 < (uint32_t)S) s_key[r2] = (uint32_t)cell0 + ((uint32_t)t << 5) + bpos;
      }
    }
    __syncthreads();
    PNX_TMARK(2);
    // ---- exclusive scan of the padded sizes (thread t owns the E entries t*E ..), class layout, pillar constants
    uint32_t cbase[kClasses], ncls[kClasses];  // first slot / pillars of every class in the current segment (wave-uniform); a class = whole tiles
    {
      uint32_t cg[E];
      uint32_t sp = 0;
#pragma unroll
      for (int e = 0; e < E; e++) {
        cg[e] = s_cnt[t * E + e];
        sp += (cg[e] >= 1u && cg[e] <= 32u) ? (1u << size_class(cg[e])) : 0u;
      }
      uint32_t ip = sp;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t yp = __shfl_up(ip, d);
        if (l >= d) ip += yp;
      }
      if (l == 63) s_misc[4 + wv] = ip;
      // class layout of the single-segment case, without another barrier: ordinal of every pillar inside its size class = pillars
      // of that class in earlier waves + in earlier entries / lower lanes of this wave (ballots)
      uint32_t ordv[E], wcls[kClasses];
#pragma unroll
      for (int c = 0; c < kClasses; c++) wcls[c] = 0u;
#pragma unroll
      for (int e = 0; e < E; e++) {
        ordv[e] = 0u;
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
      if (l == 0) {
#pragma unroll
        for (int q = 0; q < kClasses; q++) s_misc[16 + wv * kClasses + q] = wcls[q];
      }
      __syncthreads();
      uint32_t op = 0;
      for (int w = 0; w < wv; w++) op += s_misc[4 + w];
      uint32_t ep = op + ip - sp;
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
      for (int e = 0; e < E; e++) {
        const int p = t * E + e;
        const uint32_t cnt = cg[e];
        s_pst[p] = ep;
        s_cur[p] = 0u;
        if (cnt >= 1u && cnt <= 32u) {
          const uint32_t c = size_class(cnt);
          s_slot[p] = s_wt[wv * 24 + c] + (ordv[e] << c);
        }
        if (cnt > 0u) {
          const double sx = s_sum[3 * p + 0], sy = s_sum[3 * p + 1], sz = s_sum[3 * p + 2];
          const float fc = (float)cnt;
          const int32_t cell = (int32_t)s_key[p];
          const int cif = cell - fcell;
          const int yi = cif / g.gx;
          const int xi = cif - yi * g.gx;
          float* info = reinterpret_cast<float*>(&s_sum[3 * p]);  // overlays this thread's own three sums
          // mean: fp32 divide of the fp64 sum (pe:113-114); centre: idx*vs + vs/2 + min, each step rounded (pe:119-120)
          const float mx = __fdiv_rn((float)sx, fc), my = __fdiv_rn((float)sy, fc), mz = __fdiv_rn((float)sz, fc);
          const float ctrx = __fadd_rn(__fadd_rn(__fmul_rn((float)xi, g.vx), __fdiv_rn(g.vx, 2.0f)), g.minx);
          const float ctry = __fadd_rn(__fadd_rn(__fmul_rn((float)yi, g.vy), __fdiv_rn(g.vy, 2.0f)), g.miny);
          info[0] = mx, info[1] = my, info[2] = mz, info[3] = ctrx, info[4] = ctry;
          info[5] = __int_as_float(cell);
          int32_t gr = 0;
          if (ranked) {
            gr = key_rank((b * g.gx + xi) * g.gyp + yi, A.wcomb, A.wblk);
            s_key[p] = (uint32_t)gr;  // from here on: the pillar's feat_max row
            if (A.coords != nullptr && gr < A.pillar_capacity) {
              A.coords[(int64_t)gr * 3 + 0] = b;  // [b, yi, xi]  (pe:125 swaps x/y)
              A.coords[(int64_t)gr * 3 + 1] = yi;
              A.coords[(int64_t)gr * 3 + 2] = xi;
            }
          }
          if (cnt > 32u) {  // more points than one MFMA tile holds: one wave per pillar in k_pfn3_tail, under a spill id
            const int at = atomicAdd(&A.counters[kCntBig], 1);
            const uint32_t base = (uint32_t)atomicAdd(&A.counters[kCntSpill], (int)cnt);
            s_slot[p] = base;
            if (at < A.bigcap) {
              A.biglist[at] = at;
              A.pfirst[at] = base;
              A.pcnt[at] = cnt;
              A.cell_of_pillar[at] = cell;
              if (ranked) A.row_of[at] = gr;
            }
          }
        }
        ep += (cnt >= 1u && cnt <= 32u) ? (1u << size_class(cnt)) : 0u;
      }
      if (t == kSpBlock - 1) s_pst[S] = ep;
      if (t == 0 && !ranked && slice == 0u) atomicAdd(&A.counters[kCntP], (int)npil);
    }
    __syncthreads();
    PNX_TMARK(3);
    const uint32_t npad = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pst[S]);
    // ---- segments of at most `cap` padded slots, cut at pillar boundaries (normally one)
    uint32_t p0 = 0, base = 0;
    auto open_segment = [&]() -> uint32_t {
      uint32_t p1 = (uint32_t)S;
      if (t < kClasses) s_misc[10 + t] = 0u;
      if (npad - base > cap) {
        const uint32_t lim = base + cap;
        for (uint32_t p = t; p < (uint32_t)S; p += kSpBlock)
          if (p >= p0 && s_pst[p] <= lim && s_pst[p + 1] > lim) s_misc[9] = p;  // exactly one p; > p0 because a padded pillar is <= 32 <= cap
        __syncthreads();
        p1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_misc[9]);
      } else {
        __syncthreads();
      }
      uint32_t ord[E];
#pragma unroll
      for (int e = 0; e < E; e++) {
        ord[e] = 0u;
        const uint32_t p = (uint32_t)(t * E + e);
        const uint32_t cnt = s_cnt[p];
        if (p >= p0 && p < p1 && cnt >= 1u && cnt <= 32u) ord[e] = atomicAdd(&s_misc[10 + size_class(cnt)], 1u);
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
      for (int e = 0; e < E; e++) {
        const uint32_t p = (uint32_t)(t * E + e);
        const uint32_t cnt = s_cnt[p];
        s_cur[p] = 0u;
        if (p >= p0 && p < p1 && cnt >= 1u && cnt <= 32u) {
          const uint32_t c = size_class(cnt);
          s_slot[p] = s_wt[wv * 24 + 8 + c] + (ord[e] << c);
        }
      }
      __syncthreads();
      return p1;
    };
    uint32_t p1 = npad > cap ? open_segment() : (uint32_t)S;
    // ---- pass 2: every point of the segment's pillars to its LDS slot; the points of big pillars to the 64-byte record stream,
    // decorated (pe:116-123) (with the first segment)
    auto place = [&](const uint4& a, const uint4& c, const bool with_big) {
      const uint32_t rl = c.w;
      const uint32_t cnt = s_cnt[rl];
      const bool big = cnt > 32u;
      if (big ? !with_big : (rl < p0 || rl >= p1)) return;
      const uint32_t idx = atomicAdd(&s_cur[rl], 1u);
      if (!big) {
        uint4* d = reinterpret_cast<uint4*>(s_rec + (s_slot[rl] + idx) * kRecW);
        d[0] = a, d[1] = c;
        if (idx == cnt - 1u) {  // the point that completes the pillar also fills the group's spare slots
          const uint32_t G = 1u << size_class(cnt);
          for (uint32_t q = cnt; q < G; q++) {
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
        for (int q = 0; q < 12; q++) f[q] = 0.f;
#pragma unroll
        for (int q = 0; q < F; q++) f[q] = raw[q];
        f[F + 0] = __fsub_rn(raw[0], info[0]);
        f[F + 1] = __fsub_rn(raw[1], info[1]);
        f[F + 2] = __fsub_rn(raw[2], info[2]);
        f[F + 3] = __fsub_rn(raw[0], info[3]);
        f[F + 4] = __fsub_rn(raw[1], info[4]);
        const uint32_t rem = cnt - 1u - idx;
        const uint32_t aux = min(idx, 0xFFFFu) | (min(rem, 0xFFFFu) << 16);
        f[C0] = 1.f;  // multiplies the folded-BN shift column of W0' (k_fold_bn)
        uint4* d = reinterpret_cast<uint4*>(A.rec64 + (int64_t)(s_slot[rl] + idx) * 16);
        d[0] = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[2]), __float_as_uint(f[4]), __float_as_uint(f[6]));
        d[1] = make_uint4(__float_as_uint(f[8]), __float_as_uint(f[10]), aux, 0u);
        d[2] = make_uint4(__float_as_uint(f[1]), __float_as_uint(f[3]), __float_as_uint(f[5]), __float_as_uint(f[7]));
        d[3] = make_uint4(__float_as_uint(f[9]), __float_as_uint(f[11]), aux, __float_as_uint(info[5]));
      }
    };
    // the records that are still in registers belong to the first segment's pass: the registers die here, in front of the tile loop
#pragma unroll
    for (int it = 0; it < kKeepR; it++)
      if (kept(it) && kc[it].w < (uint32_t)S) place(ka[it], kc[it], true);
    PNX_TMARK(10);
    for (;;) {
      const bool first = p0 == 0u;
      walk_l2(!first, [&](const uint4& a, const uint4& c) { place(a, c, first); });
      PNX_TMARK(11);
      __syncthreads();
      __builtin_amdgcn_s_setprio(0);
      if (first && slice == 0u) tk = ticket_issue(A.tick, t);  // the next span's ticket resolves while this span's tiles are computed
      PNX_TMARK(4);

      // ---- PFN over the segment's tiles: wave wv takes tiles wv, wv + 4, ...
      {
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
          float ff[6];
          {
            const float2* ip = reinterpret_cast<const float2*>(&s_sum[3 * rl]);  // {mean x y, mean z centre x, centre y cell}
            const float2 i0 = ip[0], i1 = ip[1], i2 = ip[2];
            const float raw[6] = {__uint_as_float(cur.a.x), __uint_as_float(cur.a.y), __uint_as_float(cur.a.z),
                                  __uint_as_float(cur.a.w), __uint_as_float(cur.c.x), __uint_as_float(cur.c.y)};
            float f[12];
#pragma unroll
            for (int q = 0; q < 12; q++) f[q] = 0.f;
#pragma unroll
            for (int q = 0; q < F; q++) f[q] = raw[q];
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
          // ---- layer 1, fp16x3 (pfn_v3.hip): hi*hi + hi*lo + lo*hi as 24 v_mfma_f32_32x32x16_f16
          v16f da, db;
          {
            const float4* sp = reinterpret_cast<const float4*>(s_s1 + h * 32);
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const float4 sa = sp[q], sb = sp[4 + q];
              da[4 * q] = sa.x, da[4 * q + 1] = sa.y, da[4 * q + 2] = sa.z, da[4 * q + 3] = sa.w;
              db[4 * q] = sb.x, db[4 * q + 1] = sb.y, db[4 * q + 2] = sb.z, db[4 * q + 3] = sb.w;
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
          // record stream; its lanes compute garbage in their own MFMA columns only, and its row is not stored here.
          bool pov = false;
          const uint64_t ovm = __ballot(act && !(gm < 60000.f));
          if (ovm != 0ull) {
            pov = (((uint32_t)ovm | (uint32_t)(ovm >> 32)) >> col) & 1u;  // either half of the pillar's channels
            const uint32_t cnt = (act && pov) ? s_cnt[rl] : 0u;
            const uint32_t idx = (uint32_t)col & ((1u << cls) - 1u);
            // the group's first lane of half 0 draws the spill id and the slots; the group reads them from that lane
            int at = 0;
            uint32_t sbase = 0u;
            if (cnt > 0u && idx == 0u && h == 0) {
              at = atomicAdd(&A.counters[kCntOvf16], 1);
              sbase = (uint32_t)atomicAdd(&A.counters[kCntSpill], (int)cnt);
            }
            const int lead_lane = col & ~((1 << cls) - 1);
            at = __shfl(at, lead_lane);
            sbase = (uint32_t)__shfl((int)sbase, lead_lane);
            const int id = A.idcap - 1 - at;  // spill ids of this kind count down from the top
            if (idx < cnt && at < A.bigcap) {  // the spare slots of a group are not points
              const float* info = reinterpret_cast<const float*>(&s_sum[3 * rl]);
              const uint32_t gslot = sbase + idx;
              const uint32_t aux = idx | ((cnt - 1u - idx) << 16);
              uint4* d = reinterpret_cast<uint4*>(A.rec64 + (int64_t)gslot * 16 + 8 * h);
              d[0] = make_uint4(__float_as_uint(ff[0]), __float_as_uint(ff[1]), __float_as_uint(ff[2]), __float_as_uint(ff[3]));
              d[1] = make_uint4(__float_as_uint(ff[4]), __float_as_uint(ff[5]), aux, h ? __float_as_uint(info[5]) : 0u);
              if (idx == 0u && h == 0) {
                A.pfirst[id] = gslot;
                A.pcnt[id] = cnt;
                A.cell_of_pillar[id] = __float_as_int(info[5]);
                if (ranked) A.row_of[id] = (int32_t)s_key[rl];
                A.biglist[A.bigcap + at] = id;
              }
            }
          }
          // the accumulators carry the scale 2^(SU+SW) (shift included): an exact power of two
          constexpr float kDs = 1.0f / (float)(1 << (PNX_PFN_SU + PNX_PFN_SW));
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
          const int npl = (int)(used >> cls);   // whole pillars in the tile
          if (PACK) {
            // 16-bit canvas and no fp32 feat_max output: round-to-nearest-even of the maximum == maximum of the rounded values
            if (lead) {
              uint32_t* dst = s_out + pid * kZSP + 2 * h;
#pragma unroll
              for (int q = 0; q < 4; q++) {
                *reinterpret_cast<uint2*>(dst + 4 * q) = make_uint2(cvt_pk16<DT>(pa[4 * q], pa[4 * q + 1]), cvt_pk16<DT>(pa[4 * q + 2], pa[4 * q + 3]));
                *reinterpret_cast<uint2*>(dst + 16 + 4 * q) = make_uint2(cvt_pk16<DT>(pb[4 * q], pb[4 * q + 1]), cvt_pk16<DT>(pb[4 * q + 2], pb[4 * q + 3]));
              }
              if (h == 1) s_cellrow[pid] = pov ? 0xFFFFFFFFu : __float_as_uint(reinterpret_cast<const float*>(&s_sum[3 * rl])[5]);  // where the row goes
            }
            wave_lds_sync();
            // ---- stores: lane -> (pillar l>>3 + 8*it, 16 bytes = channels 8*(l&7) .. +7): one instruction writes 8 complete 128-byte lines
            const int qq = l & 7;
            for (int p = l >> 3; p < npl; p += 8) {
              const uint4 x = *reinterpret_cast<const uint4*>(s_out + p * kZSP + 4 * qq);
              const int32_t cl = (int32_t)s_cellrow[p];
              if (cl >= 0) canvas_store16(reinterpret_cast<uint16_t*>(out.canvas) + (int64_t)cl * 64 + 8 * qq, x, out.nt);
            }
          } else {
            if (lead) {
              uint32_t* dst = s_out + pid * kZS + 4 * h;
#pragma unroll
              for (int q = 0; q < 4; q++) {
                *reinterpret_cast<uint4*>(dst + 8 * q) =
                    make_uint4(__float_as_uint(pa[4 * q]), __float_as_uint(pa[4 * q + 1]), __float_as_uint(pa[4 * q + 2]), __float_as_uint(pa[4 * q + 3]));
                *reinterpret_cast<uint4*>(dst + 32 + 8 * q) =
                    make_uint4(__float_as_uint(pb[4 * q]), __float_as_uint(pb[4 * q + 1]), __float_as_uint(pb[4 * q + 2]), __float_as_uint(pb[4 * q + 3]));
              }
              if (h == 0) s_rank[pid] = s_key[rl];  // the feat_max row (rank outputs only)
              else s_cellrow[pid] = pov ? 0xFFFFFFFFu : __float_as_uint(reinterpret_cast<const float*>(&s_sum[3 * rl])[5]);
            }
            wave_lds_sync();
            const int qq = l & 7;
            for (int p = l >> 3; p < npl; p += 8) {
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
      if (p1 >= (uint32_t)S && slice + 1u >= nslices) {  // the span's last barrier: the next span's ticket goes with it
        publish_ticket(iter + 1);
        if (t == 0) s_misc[46] = 0u;
      }
      __syncthreads();  // the records, cursors and pillar constants are rewritten by the next segment / span
      PNX_TMARK(6);
      if (p1 >= (uint32_t)S) break;
      __builtin_amdgcn_s_setprio(2);
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pst[p1]);
      p0 = p1;
      write_index();  // the tile rows overwrote the index list; open_segment's first barrier orders the writes
      p1 = open_segment();
    }
    }  // slices
  }
#ifdef PNX_BINS_TIMERS
  if (l == 0 && A.timers != nullptr)
    for (int q = 0; q < 16; q++) atomicAdd(&A.timers[q], s_tim[wv * 16 + q]);
#endif
}

template <int F>
int launch_spans(const SpanPfnArgs& A0, const Pfn3Out& out, const PnxGeomDev& g, int64_t n, hipStream_t st) {
  SpanPfnArgs A = A0;
  const bool pack = out.g1 == nullptr && out.canvas != nullptr && out.dt != PNX_F32;
  // two workgroups per CU: 160 KiB / 2 minus room for the zero-fill workgroup that shares the CU -- LDS is handed out in 1280-byte granules:
  // 2 x 63 granules (80 000 B) + 1 for the fill = 127 of 128; at 81 000 B (2 x 64) the fill no longer fits and the reader takes 840 instead of 620 us
  const char* l_env = getenv("PNX_BINS_LDS");
  const size_t budget = l_env ? (size_t)atoi(l_env) : 80000;
  const size_t fixed = span_pfn_lds_bytes(0, pack, A.sg.B);
  PNX_REQUIRE(fixed + (size_t)(kClassSlack + 64) * kRecW * 4 <= budget, PNX_ERR_UNSUPPORTED, "the span tables of %d frames do not fit the LDS budget", A.sg.B);
  int cap = (int)((budget - fixed) / (kRecW * 4));
  const char* c_env = getenv("PNX_BINS_CAP");  // experiments / tests: force multi-segment spans
  if (c_env && atoi(c_env) >= 32 && atoi(c_env) + kClassSlack < cap) cap = atoi(c_env) + kClassSlack;
  A.cap = cap;
  const size_t lds = span_pfn_lds_bytes(cap, pack, A.sg.B);
  const char* b_env = getenv("PNX_PFN_BLOCKS");
  const int nb = n > 0 ? (b_env ? atoi(b_env) : 512) : 0;  // 256 CUs x 2 workgroups, persistent
  const int grid = nb;
  if (grid <= 0) return PNX_OK;
#define PNX_GO(DT_, PACK_)                                                                                                               \
  {                                                                                                                                     \
    static size_t lds_set = 0;                                                                                                          \
    if (lds > lds_set) {                                                                                                                \
      PNX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_span_pfn<F, DT_, PACK_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      lds_set = lds;                                                                                                                    \
    }                                                                                                                                   \
    k_span_pfn<F, DT_, PACK_><<<grid, kSpBlock, lds, st>>>(A, out, g);                                                                      \
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

// The span grouping + PFN launch.  tick[0] must be zero (the reader's memset), tables as left by pnx_launch_chunk_sort.
// wcomb / wblk (the reader's key-order bitmap prefix) select the rank outputs: g1 rows by global pillar rank, coords, row_of.
int pnx_launch_span_pfn(int F, const SpanTables& T, const SpanGeom& sg, int32_t* counters, int32_t* tick, uint32_t* rec64,
                        uint32_t* pfirst, uint32_t* pcnt, int32_t* cell_of_pillar, int32_t* row_of, int32_t* biglist, int64_t bigcap, int64_t idcap,
                        const uint2* wcomb, const uint32_t* wblk, int32_t* coords, int64_t pillar_capacity, const float* folded, float* g1,
                        int64_t g1_rows, void* canvas, int canvas_dt, int canvas_nt, int64_t n_points, const PnxGeomDev& geom, hipStream_t st) {
  SpanPfnArgs A;
  A.T = T, A.sg = sg, A.counters = counters, A.tick = tick, A.rec64 = rec64, A.pfirst = pfirst, A.pcnt = pcnt;
  A.cell_of_pillar = cell_of_pillar, A.row_of = row_of, A.biglist = biglist;
  A.bigcap = (int)(bigcap > 0x7fffffff ? 0x7fffffff : bigcap);
  A.idcap = (int)(idcap > 0x7fffffff ? 0x7fffffff : idcap);
  A.wcomb = wcomb, A.wblk = wblk, A.coords = coords, A.pillar_capacity = pillar_capacity, A.P = folded, A.cap = 0;
  A.timers = nullptr;
#ifdef PNX_BINS_TIMERS
  static unsigned long long* d_tim = nullptr;
  if (d_tim == nullptr) PNX_CHECK_HIP(hipMalloc(&d_tim, 16 * sizeof(unsigned long long)));
  PNX_CHECK_HIP(hipMemsetAsync(d_tim, 0, 16 * sizeof(unsigned long long), st));
  A.timers = d_tim;
#endif
  Pfn3Out out;
  out.g1 = g1, out.g1_rows = g1_rows, out.canvas = canvas, out.dt = canvas_dt;
  out.row_of = nullptr, out.nt = canvas_nt;
  int rc;
  switch (F) {
    case 3: rc = launch_spans<3>(A, out, geom, n_points, st); break;
    case 4: rc = launch_spans<4>(A, out, geom, n_points, st); break;
    case 5: rc = launch_spans<5>(A, out, geom, n_points, st); break;
    default: pnx_set_error("the span PFN is built for 3..5 point features, got %d", F); return PNX_ERR_UNSUPPORTED;
  }
#ifdef PNX_BINS_TIMERS
  if (rc == PNX_OK && getenv("PNX_BINS_TIMERS_PRINT")) {
    unsigned long long h_tim[16];
    PNX_CHECK_HIP(hipMemcpyAsync(h_tim, d_tim, sizeof(h_tim), hipMemcpyDeviceToHost, st));
    PNX_CHECK_HIP(hipStreamSynchronize(st));
    static const char* nm[12] = {"ticket+top barrier", "span+rows+clears", "gather+rank+pass1", "scans", "pass2 barrier", "tiles", "end barrier", "weights", "#tiles", "#spans x4", "layout regs", "place"};
    unsigned long long tot = h_tim[10] + h_tim[11];
    for (int q = 0; q < 8; q++) tot += h_tim[q];
    fprintf(stderr, "[pnx span timers] wave-cycles:");
    for (int q = 0; q < 12; q++) fprintf(stderr, " %s=%llu(%.1f%%)", nm[q], h_tim[q], (q < 8 || q > 9) ? 100.0 * h_tim[q] / (tot ? tot : 1) : 0.0);
    fprintf(stderr, "  cycles/tile=%.0f\n", h_tim[8] ? (double)h_tim[5] / h_tim[8] : 0.0);
  }
#endif
  return rc;
}
