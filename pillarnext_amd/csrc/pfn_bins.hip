// pfn_bins.hip -- pillar sort INSIDE LDS + PFN + canvas stores in ONE launch (gfx950): the pillar-sorted record stream never exists
// in HBM.  Reference semantics: pillar_encoder.py:106-123 (torch.unique inverse, scatter_mean, feature decoration), :35-50 x2
// (PFNLayer), :174-182 (PillarFeatureNet.forward) and the dense canvas of sparse_resnet.py:63-68.
//
// Round 2 (reader_bins.h::k_bin_sort + pfn_v3.hip::k_pfn3) wrote every kept point as a 64-byte decorated record to HBM and read it
// back: 308 MB of the reader's 3.0 GB per 8 nuScenes frames, and the PFN's tile walk was a chain of dependent HBM loads (the next
// tile's address comes out of this tile's records): 118 us with nothing but layer 0 in the loop.  Here one workgroup owns one bin
// (2^sh consecutive pillars of torch.unique order, all of its raw 32-byte records contiguous in the bin buffer that k_bin_scatter
// wrote) and does, per bin:
//   pass 1   points per pillar + exact fp64 coordinate sums with LDS atomics; the raw records of the first rounds stay in registers
//   scan     two exclusive scans over the pillar counts: slots in the (virtual) global sorted order, and slots in LDS, where only
//            pillars of <= 32 points live; per pillar: mean (fp32 divide of the fp64 sum, pe:113-114), pillar centre (pe:119-120),
//            canvas cell
//   pass 2   every point of a small pillar -> a 48-byte decorated record [f0 f2 f4 f6 | f1 f3 f5 f7 | f8 aux f9 pillar] at its
//            sorted LDS slot (the operand order of v_mfma_f32_32x32x2_f32: lane (point, h) feeds K elements 2kk+h)
//   PFN      the four waves walk the sorted slots in tiles of <= 32 points cut at pillar boundaries -- ds_read instead of HBM
//            latency -- with the tile body of pfn_v3.hip (fp32 MFMA layer 0, DPP segmented max, fp16x3 MFMA layer 1, packed
//            output scan) and store finished pillar lines straight into the NHWC canvas / feat_max rows
// A bin with more small-pillar points than the LDS holds is processed in pillar-aligned segments (pass 2 + PFN per segment, the raw
// records re-read from L2).  Pillars of > 32 points and the pillars of a tile that leaves the fp16x3 range are written to the
// 64-byte record stream in the round-2 format and listed for k_pfn3_tail (fp32 MFMA, one wave per pillar) -- rare at PillarNeXt
// resolutions.  Bins are handed out by a ticket counter; blocks [0, n_fill) of the launch carry zero-fill tiles (pnx_fill.h).
// Results do not depend on the order of the records inside a pillar (max is order-free, the mean is an exact fp64 sum).
#include "pnx_common.h"
#include "pnx_dppscan.h"
#include "pnx_fill.h"
#include "pfn_common.h"

namespace {

constexpr int kRecW = 12;     // words per LDS record
constexpr int kKeepR = 4;     // rounds of raw records kept in registers between pass 1 and pass 2
constexpr int kBinBlock = 256;

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
};

__device__ __forceinline__ uint32_t bin_prefix(int64_t v, const uint32_t* __restrict__ hpre, const uint32_t* __restrict__ hblk) {
  return hblk[v >> PNX_SCAN_SHIFT] + hpre[v];
}

// LDS words of one workgroup for bins of 2^sh pillars and `cap` record slots
template <bool PACK>
constexpr int wave_out_words() {
  return (PACK ? 32 * kZSP : 32 * kZS) + 64;
}
static inline size_t bin_pfn_lds_bytes(int sh, int cap, bool pack) {
  const size_t S = (size_t)1 << sh;
  const size_t head = (3 * (S + 4) + 2 * S) * 4 + 3 * S * 8 + 16 * 4;
  return head + 4 * (size_t)(pack ? wave_out_words<true>() : wave_out_words<false>()) * 4 + (size_t)cap * kRecW * 4;
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
  uint32_t* s_lst = s_gst + (S + 4);     // S + 1: exclusive starts, pillars of <= 32 points only (LDS slots)
  uint32_t* s_cur = s_lst + (S + 4);     // S cursors
  uint32_t* s_key = s_cur + S;           // S cell keys
  double* s_sum = reinterpret_cast<double*>(s_key + S);  // 3 doubles per pillar; later {mean x y z, centre x y, cell} as 6 words
  uint32_t* s_misc = reinterpret_cast<uint32_t*>(s_sum + 3 * S);
  uint32_t* s_outb = s_misc + 16;
  uint32_t* s_rec = s_outb + 4 * WL;

  const int l = t & 63, col = l & 31, h = l >> 5, wv = t >> 6;
  uint32_t* s_out = s_outb + wv * WL;  // 32 finished pillar rows of this wave
  uint32_t* s_rank = s_out + (PACK ? 32 * kZSP : 32 * kZS);
  uint32_t* s_cellrow = s_rank + 32;

  // weight fragments: coalesced loads, once per (persistent) wave -- fp16x3 block of k_fold_bn
  const float* __restrict__ FP2 = A.P + FR + 64 * 121 + l;
  float w0f[KS];
#pragma unroll
  for (int kk = 0; kk < KS; kk++) w0f[kk] = FP2[kk * 64];
  uint32_t wq[64];  // hi (wq[0..31]) and lo (wq[32..63]) fragments of W1' * 2^SW, index ((mt*4 + s)*4 + tq)
#pragma unroll
  for (int i = 0; i < 64; i++) wq[i] = __float_as_uint(FP2[(7 + i) * 64]);
  const float4* __restrict__ s1lane = reinterpret_cast<const float4*>(A.P + FR + 64 * 89 + l * 32);  // s1 in this lane's channel order
  float s1a[16], s1b[16];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float4 sa = s1lane[j], sb = s1lane[4 + j];
    s1a[4 * j + 0] = sa.x, s1a[4 * j + 1] = sa.y, s1a[4 * j + 2] = sa.z, s1a[4 * j + 3] = sa.w;
    s1b[4 * j + 0] = sb.x, s1b[4 * j + 1] = sb.y, s1b[4 * j + 2] = sb.z, s1b[4 * j + 3] = sb.w;
  }

  const int64_t Ptot = A.counters[0];
  const uint32_t n_kept = (uint32_t)A.counters[1];
  const uint32_t cap = (uint32_t)A.cap;
  const uint32_t* __restrict__ binbuf = A.binbuf;
  int tk = ticket_issue(A.tick, t);  // thread 0 only; the value is read at the top of the loop
  for (;;) {
    if (wv == 0) {
      const int bq = ticket_wait(tk);
      if (t == 0) s_misc[8] = (uint32_t)bq;
    }
    __syncthreads();
    const int b = __builtin_amdgcn_readfirstlane((int)s_misc[8]);
    const int64_t r0 = (int64_t)b << sh;
    if (b >= A.K1 || r0 >= Ptot) break;  // tickets come in bin order: every later bin is empty as well
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
    // ---- two exclusive scans of the S counts (thread t owns the E = S/256 entries t*E ..), pillar constants
    {
      const int E = S >> 8;
      uint32_t cg[8];
      uint32_t sg = 0, sl = 0;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        cg[e] = e < E ? s_cnt[t * E + e] : 0u;
        sg += cg[e];
        sl += cg[e] <= 32u ? cg[e] : 0u;
      }
      uint32_t ig = sg, il = sl;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t yg = __shfl_up(ig, d), yl = __shfl_up(il, d);
        if (l >= d) ig += yg, il += yl;
      }
      if (l == 63) s_misc[wv] = ig, s_misc[4 + wv] = il;
      __syncthreads();
      uint32_t og = 0, ol = 0;
      for (int w = 0; w < wv; w++) og += s_misc[w], ol += s_misc[4 + w];
      uint32_t eg = og + ig - sg, el = ol + il - sl;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        if (e < E) {
          const int p = t * E + e;
          const uint32_t cnt = cg[e];
          s_gst[p] = eg;
          s_lst[p] = el;
          s_cur[p] = el;
          if (cnt > 0u) {
            const double sx = s_sum[3 * p + 0], sy = s_sum[3 * p + 1], sz = s_sum[3 * p + 2];
            const float fc = (float)cnt;
            const int32_t k = (int32_t)s_key[p];
            const int yi = k % g.gyp;
            const int tq = k / g.gyp;
            const int xi = tq % g.gx, bi = tq / g.gx;
            const int32_t cell = (bi * g.gy + yi) * g.gx + xi;
            float* info = reinterpret_cast<float*>(&s_sum[3 * p]);  // overlays this thread's own three sums
            // mean: fp32 divide of the fp64 sum (pe:113-114); centre: idx*vs + vs/2 + min, each step rounded (pe:119-120)
            const float mx = __fdiv_rn((float)sx, fc), my = __fdiv_rn((float)sy, fc), mz = __fdiv_rn((float)sz, fc);
            const float ctrx = __fadd_rn(__fadd_rn(__fmul_rn((float)xi, g.vx), __fdiv_rn(g.vx, 2.0f)), g.minx);
            const float ctry = __fadd_rn(__fadd_rn(__fmul_rn((float)yi, g.vy), __fdiv_rn(g.vy, 2.0f)), g.miny);
            info[0] = mx, info[1] = my, info[2] = mz, info[3] = ctrx, info[4] = ctry;
            info[5] = __int_as_float(cell);
            const int64_t gr = r0 + p;
            if (A.write_pillars || cnt > 32u) {
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
          el += cnt <= 32u ? cnt : 0u;
        }
      }
      if (t == kBinBlock - 1) s_gst[S] = eg, s_lst[S] = el;
    }
    __syncthreads();
    const uint32_t nl = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_lst[S]);
    // ---- segments of at most `cap` LDS slots, cut at pillar boundaries (normally one)
    uint32_t p0 = 0, base = 0;
    // pillars [p0, p1) = the longest run from p0 whose small-pillar points fit into `cap` slots
    auto segment_end = [&]() -> uint32_t {
      if (nl - base <= cap) return (uint32_t)S;
      const uint32_t lim = base + cap;
      for (uint32_t p = t; p < (uint32_t)S; p += kBinBlock)
        if (p >= p0 && s_lst[p] <= lim && s_lst[p + 1] > lim) s_misc[9] = p;  // exactly one p; > p0 because a small pillar is <= 32 <= cap
      __syncthreads();
      return (uint32_t)__builtin_amdgcn_readfirstlane((int)s_misc[9]);
    };
    uint32_t p1 = segment_end();
    // ---- pass 2: every point of the segment's small pillars to its sorted LDS slot, decorated (pe:116-123); the points of big
    // pillars to the 64-byte record stream (with the first segment)
    auto place = [&](const uint4& a, const uint4& c, const bool with_big) {
      const uint32_t rl = c.w;
      const uint32_t cnt = s_cnt[rl];
      const bool big = cnt > 32u;
      if (big ? !with_big : (rl < p0 || rl >= p1)) return;
      const uint32_t pos = atomicAdd(&s_cur[rl], 1u);
      const uint32_t idx = pos - s_lst[rl], rem = cnt - 1u - idx;
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
      const uint32_t aux = min(idx, 0xFFFFu) | (min(rem, 0xFFFFu) << 16);
      if (!big) {
        uint4* d = reinterpret_cast<uint4*>(s_rec + (pos - base) * kRecW);
        d[0] = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[2]), __float_as_uint(f[4]), __float_as_uint(f[6]));
        d[1] = make_uint4(__float_as_uint(f[1]), __float_as_uint(f[3]), __float_as_uint(f[5]), __float_as_uint(f[7]));
        d[2] = make_uint4(__float_as_uint(f[8]), aux, __float_as_uint(f[9]), rl);
      } else {
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
    for (;;) {
      const uint32_t nseg = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_lst[p1]) - base;
      const bool first = p0 == 0u;
      for (uint32_t j = jfirst + t; j < be; j += kBinBlock) {
        const uint4* q = reinterpret_cast<const uint4*>(binbuf + (int64_t)j * 8);
        place(q[0], q[1], first);
      }
      if (first) tk = ticket_issue(A.tick, t);  // the next bin's ticket resolves while this bin's tiles are computed
      __syncthreads();

      // ---- PFN over the LDS slots [0, nseg): wave wv takes the pillars whose first slot lies in its quarter
      {
        auto head_from = [&](uint32_t s) -> uint32_t {  // first pillar head at or behind slot s
          if (s >= nseg) return nseg;
          const uint32_t aux = s_rec[s * kRecW + 9];
          const uint32_t idx = aux & 0xFFFFu, rem = aux >> 16;
          return idx == 0u ? s : s + rem + 1u;
        };
        uint32_t ts = (uint32_t)__builtin_amdgcn_readfirstlane((int)head_from((nseg * (uint32_t)wv) >> 2));
        const uint32_t end = (uint32_t)__builtin_amdgcn_readfirstlane((int)head_from((nseg * (uint32_t)(wv + 1)) >> 2));
        // a lane's view of one record: its operand quad, {f8|f9, aux|pillar} and the other half's control word
        struct Rec {
          uint4 q;
          uint2 p;
          uint32_t x;
        };
        auto load_rec = [&](uint32_t slot) -> Rec {
          const uint32_t* r = s_rec + slot * kRecW;
          Rec v;
          v.q = *reinterpret_cast<const uint4*>(r + 4 * h);
          v.p = *reinterpret_cast<const uint2*>(r + 8 + 2 * h);
          v.x = r[h ? 9 : 11];
          return v;
        };
        Rec nxt;
        if (ts < end) nxt = load_rec(min(ts + (uint32_t)col, end - 1u));
        while (ts < end) {
          const Rec cur = nxt;
          const uint32_t aux = h ? cur.x : cur.p.y;
          const uint32_t rl = h ? cur.p.y : cur.x;
          const bool in_range = ts + (uint32_t)col < end;
          const int idx = (int)(aux & 0xFFFFu), rem = (int)(aux >> 16);
          const bool complete = in_range && (col + rem <= 31);
          const uint32_t V = (uint32_t)__ballot(complete && h == 0);
          const int nv = __builtin_popcount(V);  // >= 1: every pillar in LDS has <= 32 points
          const uint32_t ts_next = ts + (uint32_t)nv;
          if (ts_next < end) nxt = load_rec(min(ts_next + (uint32_t)col, end - 1u));  // the next tile's records under this tile's MFMAs
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

          // ---- layer 0 (lane = point, registers = channels); K elements beyond the features: the constant 1, then zeros
          float ff[6];
          ff[0] = __uint_as_float(cur.q.x), ff[1] = __uint_as_float(cur.q.y), ff[2] = __uint_as_float(cur.q.z), ff[3] = __uint_as_float(cur.q.w);
          {
            const float one0 = (8 == C0) ? 1.f : 0.f, one1 = (9 == C0) ? 1.f : 0.f;  // K elements 8 / 9 when they are not features
            const float w8 = (8 < C0) ? __uint_as_float(cur.p.x) : one0, w9 = (9 < C0) ? __uint_as_float(cur.p.x) : one1;
            ff[4] = h ? w9 : w8;
            ff[5] = h ? ((11 == C0) ? 1.f : 0.f) : ((10 == C0) ? 1.f : 0.f);
          }
          v16f d0;
#pragma unroll
          for (int i = 0; i < 16; i++) d0[i] = 0.f;
#pragma unroll
          for (int kk = 0; kk < KS; kk++) d0 = PNX_MFMA(w0f[kk], act ? ff[kk] : 0.f, d0);
          // ---- "max" half of the concat (pe:43-44,49): per-pillar max of relu(layer 0), delivered to every point of the pillar;
          // the scan's 80 register-steps sit in the gaps of the layer-1 MFMAs that only need the point's own h0 (pfn_v3.hip)
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
          uint32_t bh[16], bl[16];
#pragma unroll
          for (int tq = 0; tq < 8; tq++) split2_f16(u[2 * tq], u[2 * tq + 1], bh[tq], bl[tq]);
#define PNX_L1H(S_, PROD, I0, N)                                                                                         \
  da = PNX_MFMA16(as_v8h(&wq[((PROD) == 2 ? 32 : 0) + (0 * 4 + (S_)) * 4]), as_v8h((PROD) == 1 ? &bl[4 * ((S_) & 1)] : &bh[4 * ((S_) & 1)]), da); \
  db = PNX_MFMA16(as_v8h(&wq[((PROD) == 2 ? 32 : 0) + (1 * 4 + (S_)) * 4]), as_v8h((PROD) == 1 ? &bl[4 * ((S_) & 1)] : &bh[4 * ((S_) & 1)]), db); \
  __builtin_amdgcn_sched_barrier(0);                                                                                     \
  if ((N) > 0) {                                                                                                         \
    _Pragma("unroll") for (int pq_ = 0; pq_ < (N); pq_++) {                                                              \
      switch (((I0) + pq_) / 16) {                                                                                       \
        case 0: scan_pair_f32<0>(g0[((I0) + pq_) % 16], sm); break;                                                      \
        case 1: scan_pair_f32<1>(g0[((I0) + pq_) % 16], sm); break;                                                      \
        case 2: scan_pair_f32<2>(g0[((I0) + pq_) % 16], sm); break;                                                      \
        case 3: scan_pair_f32<3>(g0[((I0) + pq_) % 16], sm); break;                                                      \
        default: scan_pair_f32<4>(g0[((I0) + pq_) % 16], sm); break;                                                     \
      }                                                                                                                  \
    }                                                                                                                    \
  }                                                                                                                      \
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
          const bool ovf = __ballot(act && !(gm < 60000.f)) != 0;
#pragma unroll
          for (int tq = 0; tq < 8; tq++) split2_f16(g0[2 * tq], g0[2 * tq + 1], bh[tq], bl[tq]);
          PNX_L1H(2, 0, 0, 0) PNX_L1H(2, 1, 0, 0) PNX_L1H(2, 2, 0, 0) PNX_L1H(3, 0, 0, 0) PNX_L1H(3, 1, 0, 0) PNX_L1H(3, 2, 0, 0)
#undef PNX_L1H
          if (ovf) {
            // outside the fp16 range: every pillar of the tile goes to k_pfn3_tail (fp32 MFMA, unscaled weights) through the
            // 64-byte record stream, in the format of reader_bins.h
            if (act) {
              const float* info = reinterpret_cast<const float*>(&s_sum[3 * rl]);
              const uint32_t gslot = bs + s_gst[rl] + (uint32_t)idx;
              uint4* d = reinterpret_cast<uint4*>(A.rec64 + (int64_t)gslot * 16 + 8 * h);
              d[0] = cur.q;
              d[1] = make_uint4(__float_as_uint(ff[4]), __float_as_uint(ff[5]), aux, h ? __float_as_uint(info[5]) : (uint32_t)(r0 + rl));
              if (idx == 0 && h == 0) {
                const int64_t gr = r0 + rl;
                A.pfirst[gr] = gslot;
                A.pcnt[gr] = (uint32_t)(rem + 1);
                A.cell_of_pillar[gr] = __float_as_int(info[5]);
                const int at = atomicAdd(&A.counters[4], 1);
                if (at < A.bigcap) A.biglist[A.bigcap + at] = (int)gr;
              }
            }
            ts = ts_next;
            continue;
          }
          // the accumulators carry the scale 2^(SU+SW); an exact power of two, folded into the shift's fma
          constexpr float kDs = 1.0f / (float)(1 << (PNX_PFN_SU + PNX_PFN_SW));
          // ---- per-pillar max of relu(layer 1 + shift): scan on non-negative values, the result sits in the pillar's tail lane.
          // The tail lanes then write the finished rows in NATURAL channel order: accumulator registers 4j..4j+3 of half h are
          // channels 8j + 4h .. +3 (da) / 32 + those (db).
          if (PACK) {
            // 16-bit canvas and no fp32 feat_max output: round FIRST (round-to-nearest-even is monotone, so the max of the rounded
            // values is the rounded max, bit for bit) and scan two channels per register with v_pk_max_u16
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
            if (act && rem == 0) {
              uint32_t* dst = s_out + pid * kZSP + 2 * h;
#pragma unroll
              for (int j = 0; j < 4; j++) {
                *reinterpret_cast<uint2*>(dst + 4 * j) = make_uint2(q[2 * j], q[2 * j + 1]);
                *reinterpret_cast<uint2*>(dst + 16 + 4 * j) = make_uint2(q[8 + 2 * j], q[8 + 2 * j + 1]);
              }
              if (h == 1) s_cellrow[pid] = __float_as_uint(reinterpret_cast<const float*>(&s_sum[3 * rl])[5]);  // where the row goes
            }
            wave_lds_sync();
            // ---- stores: lane -> (pillar l>>3 + 8*it, 16 bytes = channels 8*(l&7) .. +7): one instruction writes 8 complete 128-byte lines
            const int qq = l & 7;
            for (int p = l >> 3; p < npil; p += 8) {
              const uint4 x = *reinterpret_cast<const uint4*>(s_out + p * kZSP + 4 * qq);
              *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(out.canvas) + (int64_t)(int32_t)s_cellrow[p] * 64 + 8 * qq) = x;
            }
          } else {
            float pa[16], pb[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
              pa[i] = fmaxf(__builtin_fmaf(da[i], kDs, s1a[i]), 0.f);
              pb[i] = fmaxf(__builtin_fmaf(db[i], kDs, s1b[i]), 0.f);
            }
            seg_max_nn16(pa, idx, col, pl);
            seg_max_nn16(pb, idx, col, pl);
            if (act && rem == 0) {
              uint32_t* dst = s_out + pid * kZS + 4 * h;
#pragma unroll
              for (int j = 0; j < 4; j++) {
                *reinterpret_cast<uint4*>(dst + 8 * j) =
                    make_uint4(__float_as_uint(pa[4 * j]), __float_as_uint(pa[4 * j + 1]), __float_as_uint(pa[4 * j + 2]), __float_as_uint(pa[4 * j + 3]));
                *reinterpret_cast<uint4*>(dst + 32 + 8 * j) =
                    make_uint4(__float_as_uint(pb[4 * j]), __float_as_uint(pb[4 * j + 1]), __float_as_uint(pb[4 * j + 2]), __float_as_uint(pb[4 * j + 3]));
              }
              if (h == 0) s_rank[pid] = (uint32_t)(r0 + rl);  // where the row goes
              else s_cellrow[pid] = __float_as_uint(reinterpret_cast<const float*>(&s_sum[3 * rl])[5]);
            }
            wave_lds_sync();
            // ---- stores: lane -> (pillar l>>3 + 8*it, channels 8*(l&7) .. +7)
            const int qq = l & 7;
            for (int p = l >> 3; p < npil; p += 8) {
              const uint4* src = reinterpret_cast<const uint4*>(s_out + p * kZS + 8 * qq);
              const uint4 x0 = src[0], x1 = src[1];
              const float v[8] = {__uint_as_float(x0.x), __uint_as_float(x0.y), __uint_as_float(x0.z), __uint_as_float(x0.w),
                                  __uint_as_float(x1.x), __uint_as_float(x1.y), __uint_as_float(x1.z), __uint_as_float(x1.w)};
              store_chunk<DT>(out, (int)s_rank[p], (int64_t)(int32_t)s_cellrow[p], qq, v);
            }
          }
          wave_lds_sync();  // the next tile rewrites the rows
          ts = ts_next;
        }
      }
      __syncthreads();  // the records, cursors and pillar constants are rewritten by the next segment / bin
      if (p1 >= (uint32_t)S) break;
      p0 = p1;
      base += nseg;
      jfirst = bs;
      p1 = segment_end();
    }
  }
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
  PNX_REQUIRE(fixed + 64 * kRecW * 4 <= budget, PNX_ERR_UNSUPPORTED, "bins of 2^%d pillars do not fit the LDS budget", A.sh);
  int cap = (int)((budget - fixed) / (kRecW * 4));
  const char* c_env = getenv("PNX_BINS_CAP");  // experiments / tests: force multi-segment bins
  if (c_env && atoi(c_env) >= 32 && atoi(c_env) < cap) cap = atoi(c_env);
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
  A.P = folded, A.sh = sh, A.nwg = nwg, A.K1 = K1, A.cap = 0, A.n_fill = n_fill, A.write_pillars = write_pillars, A.matlen = matlen;
  Pfn3Out out;
  out.g1 = g1, out.g1_rows = g1_rows, out.canvas = canvas, out.dt = canvas_dt;
  switch (F) {
    case 3: return launch_bins<3>(A, out, geom, fj, n_points, st);
    case 4: return launch_bins<4>(A, out, geom, fj, n_points, st);
    case 5: return launch_bins<5>(A, out, geom, fj, n_points, st);
  }
  pnx_set_error("LDS-sorted PFN is built for 3..5 point features, got %d", F);
  return PNX_ERR_UNSUPPORTED;
}
