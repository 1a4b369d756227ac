// chunk_sort.hip -- the reader's grouping front end (gfx950): every point is read ONCE, sorted by canvas slab inside LDS and written
// back as part of a contiguous 64 KiB piece; no per-point rank lookup, no scatter of 32-byte records through HBM, no device-scope
// atomic per point.  Vocabulary in spans.h.  Reference semantics: pillar_encoder.py:95-109 (voxel index + range mask of every
// point); the grouping itself replaces torch.unique(dim=0) + its inverse (pe:110-111), which pfn_spans.hip completes per span.
//
//   k_chunk_sort   one workgroup per chunk of 2048 points: cell of every point (fp32 IEEE divide, pe:95-101), an occupancy byte
//                  per kept point (plain store, every writer stores the same value), then per frame present in the chunk (one for a
//                  collated batch): LDS histogram over the frame's slabs -- the atomic's return value is the point's position inside
//                  its (chunk, slab) run --, exclusive scan = the row of the run table, records placed in LDS in slab order, the
//                  piece written out with full-line stores
//   k_slab_totals  points per slab = column sums of the table rows of a frame (coalesced 2-byte reads, 4 row groups per block)
//   k_span_carve   one workgroup per frame: prefix over the slab totals, span boundaries by a LOCAL rule (parallel): a span ends where
//                  the running total crosses a multiple of the quota, every kSpanMaxSlabs slabs, and (optionally) in front of and
//                  behind a slab of more than `solo` points; one descriptor per span: {first slab, points in front of it}
// (Round 3's chain -- k_keys, k_pack_scan, 2 x k_scan_blocks, k_bin_count, k_scan_local, k_bin_scatter -- read the points three
// times, looked a rank up with a random 8-byte read per point and scattered 32-byte records: 244 us per 12 C2 frames.)
#include "pnx_common.h"
#include "spans.h"

namespace {

constexpr int kCsBlock = 256;
constexpr int kPP = kChunk / kCsBlock;  // points per thread

// A barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global store of the wave (hipcc emits
// s_waitcnt vmcnt(0) in front of s_barrier): the table rows and occupancy bytes this kernel stores are read by nobody in the launch,
// and their acknowledgements (microseconds under load) would be exposed four times per chunk.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Section timers (PNX_BINS_TIMERS builds): wave-cycle sums of lane 0 of every wave: 0 loads + cells, 1 frame + histogram, 2 scan,
// 3 table row + placing, 4 piece + bytes
#ifdef PNX_BINS_TIMERS
__device__ unsigned long long g_cs_tim[8];
#define PNX_CS_MARK(k)                                                   \
  do {                                                                   \
    if (lane == 0) {                                                     \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
      atomicAdd(&g_cs_tim[k], now_ - tlast);                             \
      tlast = now_;                                                      \
    }                                                                    \
  } while (0)
#else
#define PNX_CS_MARK(k) \
  do {                 \
  } while (0)
#endif

template <int STRIDE, bool V2>
__global__ __launch_bounds__(kCsBlock, 2) void k_chunk_sort(const float* __restrict__ pts, int64_t n, PnxGeomDev g, SpanGeom sg, uint4* __restrict__ recs,
                                                          uint32_t* __restrict__ tab, int32_t* __restrict__ rowframe, uint32_t* __restrict__ rowbase,
                                                          int32_t* counters, int32_t* frame_lo, int32_t* frame_hi, uint8_t* __restrict__ bytemap) {
  extern __shared__ __align__(16) uint32_t s_mem[];
  uint4* s_rec = reinterpret_cast<uint4*>(s_mem);  // kChunk records of 32 bytes
  uint32_t* s_hist = s_mem + kChunk * 8;           // tabw / 2 words: two 16-bit counters each (a chunk holds <= 2048 points)
  const int hw = sg.tabw >> 1;
  uint32_t* s_misc = s_hist + hw;                  // [0] lowest pending frame  [1] table row of the round  [4..7] wave sums
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c = blockIdx.x;
  const int64_t i0 = (int64_t)c * kChunk;
#ifdef PNX_BINS_TIMERS
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif

  // ---- the chunk's points: cell and frame of every kept point (pe:95-109), occupancy byte
  float pv[kPP][6];
  int32_t cell[kPP], fr[kPP];
#pragma unroll
  for (int j = 0; j < kPP; j++) {
    const int64_t i = i0 + j * kCsBlock + t;
    cell[j] = -1;
    fr[j] = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 6; k++) pv[j][k] = 0.f;
    if (i < n) {
      const float* p = pts + i * STRIDE;
      float row[STRIDE];
      if (V2) {
#pragma unroll
        for (int k = 0; k < STRIDE / 2; k++) {
          const float2 v = reinterpret_cast<const float2*>(p)[k];
          row[2 * k] = v.x, row[2 * k + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int k = 0; k < STRIDE; k++) row[k] = p[k];
      }
#pragma unroll
      for (int k = 0; k < STRIDE - 1; k++) pv[j][k] = row[1 + k];
      const float bf = row[0];
      // fp32 subtract, then IEEE divide (never a reciprocal multiply: SURVEY H1); compares on the float coordinate
      const float cx = __fdiv_rn(__fsub_rn(row[1], g.minx), g.vx);
      const float cy = __fdiv_rn(__fsub_rn(row[2], g.miny), g.vy);
      bool keep = (cx >= 0.f) && (cx < (float)g.gx) && (cy >= 0.f) && (cy < (float)g.gy);
      keep = keep && (bf > -1.0f) && (bf < (float)g.B);  // (long)b in [0, B): truncation maps (-1, 0) to 0
      if (keep) {
        const int xi = (int)cx, yi = (int)cy, bi = (int)bf;
        cell[j] = (bi * g.gy + yi) * g.gx + xi;
        fr[j] = bi;
      }
    }
  }
  if (t == 0) rowframe[c] = -1;
  if (cell[0] == 0x12345678) s_misc[15] = 1u;  // (keeps the loads in front of the first mark)
  PNX_CS_MARK(0);

  // ---- one round per frame present in the chunk
  uint32_t done = 0;
  for (int round = 0;; round++) {
    int fmin = fr[0];
#pragma unroll
    for (int j = 1; j < kPP; j++) fmin = min(fmin, fr[j]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) fmin = min(fmin, __shfl_xor(fmin, d));
    if (t == 0) s_misc[0] = 0x7fffffffu;
    for (int w = t; w < hw; w += kCsBlock) s_hist[w] = 0u;
    lds_barrier();
    if (lane == 0 && fmin != 0x7fffffff) atomicMin(reinterpret_cast<int*>(&s_misc[0]), fmin);
    lds_barrier();
    const int cur = (int)s_misc[0];
    if (cur == 0x7fffffff) break;  // block-uniform
    if (t == 0) s_misc[1] = round == 0 ? (uint32_t)c : (uint32_t)sg.nchunks + (uint32_t)atomicAdd(&counters[kCntRows], 1);
    const int32_t cbase = cur * sg.cpf;
    uint32_t pos[kPP];
#pragma unroll
    for (int j = 0; j < kPP; j++) {
      pos[j] = 0u;
      if (fr[j] == cur) {
        const int sl = (cell[j] - cbase) >> kSlabShift;
        const int sh = (sl & 1) << 4;
        const uint32_t old = atomicAdd(&s_hist[sl >> 1], 1u << sh);  // the old count = this point's position inside its (chunk, slab) run
        pos[j] = (old >> sh) & 0xFFFFu;
      }
    }
    lds_barrier();
    PNX_CS_MARK(1);
    // exclusive scan over the packed counters: thread t owns the words [t * wpt, (t + 1) * wpt)
    const int wpt = (hw + kCsBlock - 1) / kCsBlock;
    const int w0 = min(t * wpt, hw), w1 = min(w0 + wpt, hw);
    uint32_t sum = 0;
    for (int w = w0; w < w1; w++) {
      const uint32_t v = s_hist[w];
      sum += (v & 0xFFFFu) + (v >> 16);
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(inc, d);
      if (lane >= d) inc += y;
    }
    if (lane == 63) s_misc[4 + wave] = inc;
    lds_barrier();
    uint32_t run = inc - sum;
    for (int q = 0; q < wave; q++) run += s_misc[4 + q];
    const uint32_t nk = s_misc[4] + s_misc[5] + s_misc[6] + s_misc[7];  // points of this frame in the chunk
    for (int w = w0; w < w1; w++) {
      const uint32_t v = s_hist[w];
      const uint32_t a = v & 0xFFFFu, b = v >> 16;
      s_hist[w] = run | ((run + a) << 16);
      run += a + b;
    }
    lds_barrier();
    PNX_CS_MARK(2);
    uint32_t row = s_misc[1];
    if (row >= (uint32_t)(sg.nchunks + sg.ovf_cap)) row = (uint32_t)c;  // cannot happen (the pool holds nchunks * (B - 1) rows); never write outside
    uint32_t* trow = tab + (int64_t)row * hw;
    for (int w = t; w < hw; w += kCsBlock) trow[w] = s_hist[w];
    if (t == 0) {
      rowframe[row] = cur;
      rowbase[row] = (uint32_t)(i0 + done);
      if (round == 0) {
        atomicMax(&frame_lo[cur], sg.nchunks - c);
        atomicMax(&frame_hi[cur], c + 1);
      }
    }
#pragma unroll
    for (int j = 0; j < kPP; j++) {
      if (fr[j] == cur) {
        const int sl = (cell[j] - cbase) >> kSlabShift;
        const uint32_t st = (s_hist[sl >> 1] >> ((sl & 1) << 4)) & 0xFFFFu;
        const uint32_t dst = done + st + pos[j];
        s_rec[2 * dst] = make_uint4(__float_as_uint(pv[j][0]), __float_as_uint(pv[j][1]), __float_as_uint(pv[j][2]), __float_as_uint(pv[j][3]));
        s_rec[2 * dst + 1] = make_uint4(__float_as_uint(pv[j][4]), __float_as_uint(pv[j][5]), (uint32_t)(i0 + j * kCsBlock + t), (uint32_t)cell[j]);
        fr[j] = 0x7fffffff;
      }
    }
    done += nk;
    lds_barrier();  // the counters are cleared at the top of the next round
    PNX_CS_MARK(3);
  }
  // ---- the piece: `done` records, contiguous
  uint4* dst = recs + i0 * 2;
  // plain stores: the span kernel re-reads the records (nontemporal: 632 -> 692 us).  The occupancy bytes go out with them, in SORTED order: the
  // 32 records a wave's odd lanes hold lie in one slab or two, so a store instruction touches the 512 occupancy bytes of a slab (4 lines) instead
  // of 64 lines all over the frame as in point order; every writer of a cell stores the same value and nothing waits for the stores.
  for (uint32_t q = t; q < done * 2u; q += kCsBlock) {
    const uint4 v = s_rec[q];
    dst[q] = v;
    if (bytemap != nullptr && (q & 1u)) bytemap[(int32_t)v.w] = 1;
  }
  if (t == 0 && done > 0u) atomicAdd(&counters[kCntKept], (int)done);
  PNX_CS_MARK(4);
}

// points per slab: T[b * nf + sl] = sum over the rows of frame b of row[sl + 1] - row[sl].  64 slabs x 8 row groups per block; the rows
// of a group are read in batches of 8 with all loads in flight (a row-by-row loop ran at one memory latency per row: 25 us).
constexpr int kTotGroups = 8, kTotBatch = 8;
__global__ __launch_bounds__(64 * kTotGroups) void k_slab_totals(const uint16_t* __restrict__ tab, const int32_t* __restrict__ rowframe,
                                                                const int32_t* __restrict__ frame_lo, const int32_t* __restrict__ frame_hi,
                                                                const int32_t* __restrict__ counters, SpanGeom sg, uint32_t* __restrict__ slab_tot) {
  __shared__ uint32_t s_part[kTotGroups][64];
  __builtin_amdgcn_s_setprio(3);  // runs beside the canvas zero-fill (reader.hip): short and latency-bound, so first in line
  const int t = threadIdx.x, ls = t & 63, grp = t >> 6;
  const int b = blockIdx.y, sl0 = blockIdx.x * 64 + ls;
  const bool in = sl0 < sg.nf;
  const int sl = in ? sl0 : 0;
  uint32_t sum = 0;
  auto rows = [&](int first, int last) {  // rows first, first + kTotGroups, ... <= last
    for (int r0 = first; r0 <= last; r0 += kTotGroups * kTotBatch) {
      int rf[kTotBatch];
      uint32_t a[kTotBatch], c[kTotBatch];
#pragma unroll
      for (int u = 0; u < kTotBatch; u++) {
        const int r = r0 + u * kTotGroups;
        const int rr = r <= last ? r : first;
        const uint16_t* tr = tab + (int64_t)rr * sg.tabw + sl;
        rf[u] = r <= last ? rowframe[rr] : -1;
        a[u] = tr[0];
        c[u] = tr[1];
      }
#pragma unroll
      for (int u = 0; u < kTotBatch; u++)
        if (rf[u] == b) sum += c[u] - a[u];
    }
  };
  const int hi_e = frame_hi[b];
  if (hi_e > 0) rows(sg.nchunks - frame_lo[b] + grp, hi_e - 1);
  int novf = counters[kCntRows];
  novf = novf < sg.ovf_cap ? novf : sg.ovf_cap;
  if (novf > 0) rows(sg.nchunks + grp, sg.nchunks + novf - 1);
  s_part[grp][ls] = sum;
  __syncthreads();
  if (grp == 0 && in) {
    uint32_t tot = 0;
#pragma unroll
    for (int q = 0; q < kTotGroups; q++) tot += s_part[q][ls];
    slab_tot[(int64_t)b * sg.nf + sl0] = tot;
  }
}

// span boundaries of one frame
__global__ __launch_bounds__(1024) void k_span_carve(const uint32_t* __restrict__ slab_tot, SpanGeom sg, uint2* __restrict__ span_desc,
                                                     int32_t* __restrict__ nspan) {
  __shared__ uint32_t s_wave[2][16];
  __builtin_amdgcn_s_setprio(3);  // as k_slab_totals
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = blockIdx.x, nf = sg.nf;
  const uint32_t* T = slab_tot + (int64_t)b * nf;
  const int per = (nf + 1023) / 1024;
  const int s0 = min(t * per, nf), s1 = min(s0 + per, nf);
  uint32_t sum = 0;
  for (int s = s0; s < s1; s++) sum += T[s];
  uint32_t inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(inc, d);
    if (lane >= d) inc += y;
  }
  if (lane == 63) s_wave[0][wave] = inc;
  __syncthreads();
  uint32_t pre = inc - sum;
  for (int q = 0; q < wave; q++) pre += s_wave[0][q];
  // boundary flags (each depends on the slab, its predecessor and the exclusive prefix only)
  uint32_t flags = 0;  // per <= 32 slabs per thread (nf <= 32768)
  const uint32_t quota = (uint32_t)sg.quota, solo = sg.solo > 0 ? (uint32_t)sg.solo : 0xFFFFFFFFu;
  {
    uint32_t p = pre;
    for (int s = s0; s < s1; s++) {
      const uint32_t ts = T[s], tp = s > 0 ? T[s - 1] : 0u;
      const bool cut = s == 0 || (s % kSpanMaxSlabs) == 0 || (p / quota) != ((p - tp) / quota) || ts > solo || tp > solo;
      flags |= (cut ? 1u : 0u) << (s - s0);
      p += ts;
    }
  }
  const uint32_t nc = (uint32_t)__popc(flags);
  uint32_t inc2 = nc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(inc2, d);
    if (lane >= d) inc2 += y;
  }
  if (lane == 63) s_wave[1][wave] = inc2;
  __syncthreads();
  uint32_t at = inc2 - nc;
  for (int q = 0; q < wave; q++) at += s_wave[1][q];
  uint2* out = span_desc + (int64_t)b * (nf + 1);
  {
    uint32_t p = pre;
    for (int s = s0; s < s1; s++) {
      if ((flags >> (s - s0)) & 1u) out[at++] = make_uint2((uint32_t)s, p);
      p += T[s];
    }
    if (t == 1023) {
      out[at] = make_uint2((uint32_t)nf, p);
      nspan[b] = (int32_t)at;
    }
  }
}

}  // namespace

size_t pnx_chunk_sort_lds(const SpanGeom& sg) { return (size_t)kChunk * 32 + (size_t)(sg.tabw / 2) * 4 + 64; }

// keys + chunk sort + slab totals + span carve.  counters / frame_lo / frame_hi must be zero; bytemap (optional, B * gy * gx bytes)
// must be zero and receives a 1 for every occupied cell.
int pnx_launch_chunk_sort(const float* points, int64_t n, int stride, const PnxGeomDev& g, const SpanGeom& sg, uint4* recs, uint16_t* tab,
                          int32_t* rowframe, uint32_t* rowbase, int32_t* counters, int32_t* frame_lo, int32_t* frame_hi, uint8_t* bytemap,
                          uint32_t* slab_tot, uint2* span_desc, int32_t* nspan, hipStream_t st, hipEvent_t sorted) {
  PNX_REQUIRE(sg.nf <= 32768 && sg.B <= 1024, PNX_ERR_UNSUPPORTED, "%d slabs per frame / %d frames exceed the span tables", sg.nf, sg.B);
  if (sg.nchunks > 0) {
    const size_t lds = pnx_chunk_sort_lds(sg);
    const bool v2 = (reinterpret_cast<uintptr_t>(points) & 7) == 0;
    uint32_t* tab32 = reinterpret_cast<uint32_t*>(tab);
#define PNX_CS(S_, V_)                                                                                                                          \
  {                                                                                                                                            \
    static size_t lds_set = 0;                                                                                                                  \
    if (lds > lds_set) {                                                                                                                        \
      PNX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chunk_sort<S_, V_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      lds_set = lds;                                                                                                                            \
    }                                                                                                                                           \
    k_chunk_sort<S_, V_><<<sg.nchunks, kCsBlock, lds, st>>>(points, n, g, sg, recs, tab32, rowframe, rowbase, counters, frame_lo, frame_hi, bytemap); \
  }
    switch (stride) {
      case 4: if (v2) PNX_CS(4, true) else PNX_CS(4, false) break;
      case 5: PNX_CS(5, false) break;
      case 6: if (v2) PNX_CS(6, true) else PNX_CS(6, false) break;
      case 7: PNX_CS(7, false) break;
      default: pnx_set_error("row_stride %d", stride); return PNX_ERR_UNSUPPORTED;
    }
#undef PNX_CS
    PNX_LAUNCH_CHECK();
  }
#ifdef PNX_BINS_TIMERS
  if (getenv("PNX_BINS_TIMERS_PRINT")) {
    unsigned long long h[8];
    PNX_CHECK_HIP(hipStreamSynchronize(st));
    PNX_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cs_tim), sizeof(h)));
    unsigned long long tot = h[0] + h[1] + h[2] + h[3] + h[4];
    fprintf(stderr, "[pnx chunk sort timers] loads+cells=%.1f%% frame+hist=%.1f%% scan=%.1f%% table+place=%.1f%% piece+bytes=%.1f%%  (cumulative)\n", 100.0 * h[0] / tot,
            100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * h[3] / tot, 100.0 * h[4] / tot);
  }
#endif
  if (sorted != nullptr) PNX_CHECK_HIP(hipEventRecord(sorted, st));  // the occupancy bytes are complete: the zero-fill may start
  k_slab_totals<<<dim3((unsigned)((sg.nf + 63) / 64), (unsigned)sg.B), 64 * kTotGroups, 0, st>>>(tab, rowframe, frame_lo, frame_hi, counters, sg, slab_tot);
  k_span_carve<<<sg.B, 1024, 0, st>>>(slab_tot, sg, span_desc, nspan);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}
