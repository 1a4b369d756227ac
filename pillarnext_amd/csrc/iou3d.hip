// iou3d.hip -- rotated-box BEV overlap / IoU and bitmask NMS with an on-device greedy scan, gfx950.
//
// Arithmetic follows det3d/core/iou3d_nms/src/iou3d_nms_kernel.cu ("cu:") operation for operation in fp32
// (no contraction: this file is built with -ffp-contract=off), with cos/sin/atan2 taken from pnx_detmath.h
// so that results are bit-identical to oracle/pnx_oracle.c's *_det functions on any host.
// What is different from the reference is everything around the arithmetic:
//   * per-box work (sin/cos of +-heading, rotated corners, half extents + margin) is hoisted out of the
//     pair loop and staged in LDS once per 64-box column tile (the reference recomputes 4 sincos + 8 more
//     inside check_in_box2d for every pair, cu:51-61,137-149);
//   * polygon vertices and their atan2 keys live in lane-major LDS arrays (no scratch), the key is computed
//     once per vertex instead of twice per comparison (cu:100-102);
//   * lower-triangle mask tiles, which nms_gpu never reads (iou3d_nms.cpp:150), are not computed (cu:288);
//   * the greedy suppression runs on the device, 64 boxes per step, for all (sample, task, class) segments
//     of a frame in one launch -- no cudaMalloc / blocking D2H / host loop / H2D per class
//     (iou3d_nms.cpp:126-155, box_torch_ops.py:19-26).
#pragma clang fp contract(off)
#include "pnx_common.h"

#define PNX_HD __device__ __forceinline__
#define PNX_GEOM __device__ __forceinline__
#include "iou3d_geom.h"

namespace {

// ---- N x M pairs (boxes_overlap_kernel cu:236-249, boxes_iou_bev_kernel cu:264-278) and aligned pairs
enum { MODE_OVERLAP = 0, MODE_IOU = 1, MODE_IOU3D = 2 };

template <int MODE>
__global__ __launch_bounds__(256) void k_pairs(const float* __restrict__ a, int64_t n, const float* __restrict__ b, int64_t m,
                                               float* __restrict__ out, int aligned) {
  __shared__ float spx[kMaxPts * 256], spy[kMaxPts * 256], sang[kMaxPts * 256];
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = aligned ? n : n * m;
  if (idx >= total) return;
  const int64_t i = aligned ? idx : idx / m, j = aligned ? idx : idx % m;
  const float* pa = a + i * 7;
  const float* pb = b + j * 7;
  const BoxPre A = make_box(pa), B = make_box(pb);
  float r;
  if (MODE == MODE_OVERLAP) {
    r = box_overlap<256>(A, B, spx, spy, sang, threadIdx.x);
  } else if (MODE == MODE_IOU) {
    r = iou_bev<256>(A, B, spx, spy, sang, threadIdx.x);
  } else {  // boxes_aligned_iou3d_gpu (iou3d_nms_utils.py:66-89)
    const float bev = box_overlap<256>(A, B, spx, spy, sang, threadIdx.x);
    const float a_max = pa[2] + pa[5] / 2, a_min = pa[2] - pa[5] / 2;
    const float b_max = pb[2] + pb[5] / 2, b_min = pb[2] - pb[5] / 2;
    const float max_of_min = a_min > b_min ? a_min : b_min;
    const float min_of_max = a_max < b_max ? a_max : b_max;
    float hh = min_of_max - max_of_min;
    if (hh < 0.f) hh = 0.f;
    const float ov = bev * hh;
    const float va = pa[3] * pa[4] * pa[5], vb = pb[3] * pb[4] * pb[5];
    float den = va + vb - ov;
    if (den < 1e-6f) den = 1e-6f;
    r = ov / den;
  }
  out[idx] = r;
}

// ---- NMS bitmask (nms_kernel cu:280-324 / nms_normal_kernel cu:341-385), batched over segments.
// grid = (col tiles, row tiles, segments), one wave per tile; only tiles on or above the diagonal.
// mask word (row i, col tile c) of a segment lives at mask[(off + i) * cbmax + c].
__device__ __forceinline__ int seg_size(const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_len, int seg) {
  int n = seg_off[seg + 1] - seg_off[seg];
  if (seg_len) n = min(n, max(seg_len[seg], 0));
  return n;
}

// Rotated variant.  The reference lets every lane walk its 64 columns serially through the full polygon clipping (cu:311-321), although almost
// all pairs are far apart.  Round 5 splits the work by what it costs:
//   k_box_pre    once per BOX: sin / cos of +-heading, rotated corners, extents, bounding radius (64 bytes into the workspace) -- a 1000-box list has
//                136 tiles, every box used to be set up 17 times -- and the box's row of mask words zeroed
//   k_nms_cand   one wave per 64 x 64 tile on or above the diagonal: every lane (= row) builds the 64-bit set of columns whose bounding circles touch
//                its own (a dozen flops per pair, the column data through one broadcast 16-byte LDS read each) and the tile APPENDS its surviving
//                (row, col) pairs to one global list (one atomic per tile); a tile that no longer fits the list goes to an overflow list
//   k_nms_pairs  the expensive clipping, one pair per thread over the global list: every lane busy whatever the tiles looked like (a sparse scene
//                has 2-3 candidate pairs per tile: the per-tile form ran the clipping with 2-3 of 64 lanes); hits are OR-ed into the mask words
//   k_nms_tiles  the overflow tiles, processed tile by tile as in rounds 1-4 (pairs packed in LDS, dealt to the 64 lanes)
// Pairs rejected by the bounding circles have overlap exactly 0 in the reference too (cnt = 0), and OR is order-free: the mask is bit-identical.
// pre[seg * cbmax * 64 + i] = box i of segment seg.
__global__ __launch_bounds__(256) void k_box_pre(const float* __restrict__ boxes, const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_len,
                                                 int cbmax, BoxPre* __restrict__ pre, uint64_t* __restrict__ mask, unsigned int* __restrict__ counts) {
  const int seg = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (seg == 0 && i < 2) counts[i] = 0;  // the two list counters of this call (k_nms_cand starts behind this kernel)
  if (i >= seg_size(seg_off, seg_len, seg)) return;
  const int64_t gi = seg_off[seg] + i;
  pre[(int64_t)seg * cbmax * 64 + i] = make_box(boxes + gi * 7);
  for (int c = i >> 6; c < cbmax; c++) mask[gi * cbmax + c] = 0;  // the words on or above the diagonal: the only ones anybody reads (iou3d_nms.cpp:150)
}

struct NmsLists {
  unsigned long long* pairs;  // seg << 40 | row << 20 | col (row, col: box index inside the segment)
  unsigned long long* tiles;  // overflow: seg << 40 | rb << 20 | cb
  unsigned int* counts;       // {pairs appended, overflow tiles}
  unsigned int pair_cap, tile_cap;
};

__device__ __forceinline__ uint64_t tile_candidates(const BoxPre* __restrict__ prow, const float4* s_ccr, int t, int row_size, int col_size, bool diag, float thr) {
  uint64_t cand = 0;
  if (t < row_size) {
    const int start = diag ? t + 1 : 0;
    if (thr < 0.f) {  // iou 0 > thr only for a negative threshold: then every pair is a candidate
      for (int k = start; k < col_size; k++) cand |= 1ULL << k;
    } else {
      const float mx = prow[t].cx, my = prow[t].cy, mr = prow[t].rad;
#pragma unroll 8
      for (int k = 0; k < 64; k++) {
        const float4 c = s_ccr[k < col_size ? k : 0];
        const float ddx = mx - c.x, ddy = my - c.y, rr = mr + c.z;
        const bool far = ddx * ddx + ddy * ddy > rr * rr;  // far_apart's expression; false for NaN -> full path
        if (k >= start && k < col_size && !far) cand |= 1ULL << k;
      }
    }
  }
  return cand;
}

constexpr int kStrip = 16;  // at most this many column tiles per workgroup of k_nms_cand (`strip` = 4, 8 or 16, chosen by the host from the number of
                            // tiles): ONE atomic on the list counter per strip -- same-address atomics run at ~90 per us, one per tile was 69 us for
                            // the 6 240 tiles of three 4 096-box lists; few tiles (ten 1000-box lists: 1 360) want small strips and many workgroups
__global__ __launch_bounds__(256) void k_nms_cand(const BoxPre* __restrict__ pre, const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_len,
                                                  const float* __restrict__ thresh, int cbmax, int strip, NmsLists L) {
  const int seg = blockIdx.z, rb = blockIdx.y, cb0 = blockIdx.x * strip;
  if (cb0 + strip <= rb) return;  // the whole strip lies below the diagonal
  const int n = seg_size(seg_off, seg_len, seg);
  if (rb * 64 >= n || cb0 * 64 >= n) return;
  const int row_size = min(n - rb * 64, 64);
  const int t = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __shared__ float4 s_ccr[4][64];
  __shared__ unsigned long long s_cand[kStrip][64];
  __shared__ unsigned int s_tot[kStrip], s_pre[kStrip];
  __shared__ unsigned int s_base, s_fill;
  const BoxPre* pseg = pre + (int64_t)seg * cbmax * 64;
  const BoxPre* prow = pseg + rb * 64;
  const float thr = thresh[seg];
  for (int q = 0; q < strip / 4; q++) {  // wave wv takes tiles wv, wv + 4, ... of the strip
    const int ti = q * 4 + wv, cb = cb0 + ti;
    uint64_t cand = 0;
    const bool live = cb >= rb && cb * 64 < n;  // wave-uniform
    if (live) {
      const int col_size = min(n - cb * 64, 64);
      if (t < col_size) s_ccr[wv][t] = make_float4(pseg[cb * 64 + t].cx, pseg[cb * 64 + t].cy, pseg[cb * 64 + t].rad, 0.f);
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes are visible to its reads (one wave per s_ccr slice)
      cand = tile_candidates(prow, s_ccr[wv], t, row_size, col_size, rb == cb, thr);
    }
    s_cand[ti][t] = cand;
    int tot = __popcll(cand);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
    if (t == 0) s_tot[ti] = (unsigned int)tot;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int run = 0;
    for (int i = 0; i < strip; i++) {
      s_pre[i] = run;
      run += s_tot[i];
    }
    unsigned int base = run ? atomicAdd(&L.counts[0], run) : 0u;
    if (run && (unsigned long long)base + run > L.pair_cap) {  // the list is full: the strip's tiles are done the old way (k_nms_tiles)
      s_fill = base;  // the one strip that straddles the end of the list marks its slots empty (below, all threads)
      unsigned int live_tiles = 0;
      for (int i = 0; i < strip; i++) live_tiles += s_tot[i] ? 1u : 0u;
      unsigned int k = atomicAdd(&L.counts[1], live_tiles);
      for (int i = 0; i < strip; i++)
        if (s_tot[i] && k < L.tile_cap) L.tiles[k++] = ((unsigned long long)seg << 40) | ((unsigned long long)rb << 20) | (unsigned long long)(cb0 + i);
      base = 0xffffffffu;
    }
    s_base = base;
  }
  __syncthreads();
  const unsigned int base = s_base;
  if (base == 0xffffffffu) {
    for (unsigned long long q = (unsigned long long)s_fill + threadIdx.x; q < L.pair_cap; q += 256) L.pairs[q] = ~0ull;
    return;
  }
  for (int q = 0; q < strip / 4; q++) {
    const int ti = q * 4 + wv;
    if (s_tot[ti] == 0) continue;
    uint64_t c = s_cand[ti][t];
    const int cnt = __popcll(c);
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(inc, d);
      if (t >= d) inc += y;
    }
    unsigned int o = base + s_pre[ti] + (unsigned int)(inc - cnt);
    const unsigned long long hi = ((unsigned long long)seg << 40) | ((unsigned long long)(rb * 64 + t) << 20) | (unsigned long long)((cb0 + ti) * 64);
    while (c) {
      const int k = __ffsll((long long)c) - 1;
      c &= c - 1;
      L.pairs[o++] = hi | (unsigned long long)k;
    }
  }
}

__global__ __launch_bounds__(256) void k_nms_pairs(const BoxPre* __restrict__ pre, const int32_t* __restrict__ seg_off, const float* __restrict__ thresh,
                                                   uint64_t* __restrict__ mask, int cbmax, NmsLists L) {
  __shared__ float spx[kMaxPts * 256], spy[kMaxPts * 256], sang[kMaxPts * 256];
  const unsigned int n = min(L.counts[0], L.pair_cap);  // entries behind the capacity belong to tiles that went to the overflow list
  for (unsigned int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const unsigned long long p = L.pairs[e];
    if (p == ~0ull) continue;  // slot of the tile that straddled the end of the list
    const int seg = (int)(p >> 40), i = (int)((p >> 20) & 0xfffff), k = (int)(p & 0xfffff);
    const BoxPre* ps = pre + (int64_t)seg * cbmax * 64;
    const BoxPre A = ps[i], B = ps[k];
    const float v = iou_bev<256>(A, B, spx, spy, sang, threadIdx.x);  // row box first, as cu:317
    if (v > thresh[seg]) {
      unsigned int* w = reinterpret_cast<unsigned int*>(mask + (int64_t)(seg_off[seg] + i) * cbmax + (k >> 6));
      atomicOr(w + ((k >> 5) & 1), 1u << (k & 31));
    }
  }
}

// The per-tile form of rounds 1-4, for the tiles the pair list had no room for (dense clusters of thousands of boxes): candidates packed in LDS and
// dealt out to the 64 lanes; the tile owns its mask words.
__global__ __launch_bounds__(64) void k_nms_tiles(const BoxPre* __restrict__ pre, const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_len,
                                                  const float* __restrict__ thresh, uint64_t* __restrict__ mask, int cbmax, NmsLists L) {
  __shared__ float4 s_ccr[64];
  __shared__ unsigned short s_pairs[4096];
  __shared__ unsigned int s_bits[64 * 2];
  __shared__ float spx[kMaxPts * 64], spy[kMaxPts * 64], sang[kMaxPts * 64];
  const int t = threadIdx.x;
  const unsigned int n_tiles = min(L.counts[1], L.tile_cap);
  for (unsigned int ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
    const unsigned long long code = L.tiles[ti];
    const int seg = (int)(code >> 40), rb = (int)((code >> 20) & 0xfffff), cb = (int)(code & 0xfffff);
    const int off = seg_off[seg], n = seg_size(seg_off, seg_len, seg);
    const float thr = thresh[seg];
    const int row_size = min(n - rb * 64, 64), col_size = min(n - cb * 64, 64);
    const BoxPre* prow = pre + (int64_t)seg * cbmax * 64 + rb * 64;
    const BoxPre* pcol = pre + (int64_t)seg * cbmax * 64 + cb * 64;
    __syncthreads();  // the previous tile's readers are done
    if (t < col_size) s_ccr[t] = make_float4(pcol[t].cx, pcol[t].cy, pcol[t].rad, 0.f);
    s_bits[t] = 0;
    s_bits[64 + t] = 0;
    __syncthreads();
    const uint64_t cand = tile_candidates(prow, s_ccr, t, row_size, col_size, rb == cb, thr);
    const int cnt = __popcll(cand);
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int y = __shfl_up(inc, d);
      if (t >= d) inc += y;
    }
    const int total = __shfl(inc, 63);
    int o = inc - cnt;
    uint64_t c = cand;
    while (c) {
      const int k = __ffsll((long long)c) - 1;
      c &= c - 1;
      s_pairs[o++] = (unsigned short)((t << 6) | k);
    }
    __syncthreads();
    for (int e = t; e < total; e += 64) {
      const int pr = s_pairs[e];
      const int i = pr >> 6, k = pr & 63;
      const BoxPre A = prow[i], B = pcol[k];
      const float v = iou_bev<64>(A, B, spx, spy, sang, t);
      if (v > thr) atomicOr(&s_bits[2 * i + (k >> 5)], 1u << (k & 31));
    }
    __syncthreads();
    if (t < row_size) mask[(int64_t)(off + rb * 64 + t) * cbmax + cb] = (uint64_t)s_bits[2 * t] | ((uint64_t)s_bits[2 * t + 1] << 32);
  }
}

template <bool ROTATED>
__global__ __launch_bounds__(64) void k_nms_mask(const float* __restrict__ boxes, const int32_t* __restrict__ seg_off,
                                                 const int32_t* __restrict__ seg_len, const float* __restrict__ thresh,
                                                 uint64_t* __restrict__ mask, int cbmax) {
  const int seg = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  const int off = seg_off[seg], n = seg_size(seg_off, seg_len, seg);
  if (rb * 64 >= n || cb * 64 >= n) return;
  const float thr = thresh[seg];
  const int row_size = min(n - rb * 64, 64), col_size = min(n - cb * 64, 64);
  const int t = threadIdx.x;
  __shared__ float s_raw[64 * 7];
  if (t < col_size) {
    const float* p = boxes + (int64_t)(off + cb * 64 + t) * 7;
#pragma unroll
    for (int k = 0; k < 7; k++) s_raw[t * 7 + k] = p[k];
  }
  __syncthreads();
  if (t < row_size) {
    const int i = rb * 64 + t;
    const float* pr = boxes + (int64_t)(off + i) * 7;
    float ra[7];
#pragma unroll
    for (int k = 0; k < 7; k++) ra[k] = pr[k];
    uint64_t bits = 0;
    const int start = (rb == cb) ? t + 1 : 0;
    for (int k = start; k < col_size; k++)
      if (iou_normal(ra, s_raw + k * 7) > thr) bits |= 1ULL << k;
    mask[(int64_t)(off + i) * cbmax + cb] = bits;
  }
}

// ---- greedy scan (nms_gpu host loop, iou3d_nms.cpp:139-155), one wave per segment, 64 boxes per step.
__global__ __launch_bounds__(64) void k_nms_greedy(const uint64_t* __restrict__ mask, const int32_t* __restrict__ seg_off,
                                                   const int32_t* __restrict__ seg_len, int cbmax, int post_max,
                                                   int32_t* __restrict__ keep, int32_t* __restrict__ keep_count) {
  extern __shared__ uint64_t s_remv[];  // cbmax words
  const int seg = blockIdx.x, t = threadIdx.x;
  const int off = seg_off[seg], n = seg_size(seg_off, seg_len, seg);
  const int cb = (n + 63) >> 6;
  for (int j = t; j < cb; j += 64) s_remv[j] = 0;
  __syncthreads();
  int nkept = 0;
  // diagonal words: lane t holds the in-block suppression set of box nb*64+t; the NEXT block's words are requested before this block is resolved
  // (one dependent L2 round trip per block otherwise: 64 of them for a 4096-box list)
  uint64_t diag_next = (t < min(64, n)) ? mask[(int64_t)(off + t) * cbmax] : 0ULL;
  for (int nb = 0; nb < cb; nb++) {
    const int bsz = min(64, n - nb * 64);
    const uint64_t diag = diag_next;
    if (nb + 1 < cb) diag_next = (t < min(64, n - (nb + 1) * 64)) ? mask[(int64_t)(off + (nb + 1) * 64 + t) * cbmax + nb + 1] : 0ULL;
    uint64_t cur = s_remv[nb];
    uint64_t kept = 0;
    for (int k = 0; k < bsz; k++) {  // wave-uniform serial resolve
      const uint64_t dk = __shfl(diag, k);
      if (!((cur >> k) & 1ULL)) {
        kept |= 1ULL << k;
        cur |= dk;
      }
    }
    // emit kept indices in ascending order
    if ((kept >> t) & 1ULL) {
      const int pos = nkept + __popcll(kept & ((1ULL << t) - 1ULL));
      keep[off + pos] = nb * 64 + t;
    }
    nkept += __popcll(kept);
    if (post_max > 0 && nkept >= post_max) break;
    // fold the kept rows into the words still to come: sixteen rows' words in flight per step (one load per step waited for its own L2 round trip: 173 us for three 4 096-box lists)
    for (int j = nb + 1 + t; j < cb; j += 64) {
      uint64_t acc = s_remv[j];
      uint64_t kk = kept;
      const uint64_t* col = mask + (int64_t)(off + nb * 64) * cbmax + j;
      while (kk) {
        int r[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
          r[q] = kk ? __ffsll((long long)kk) - 1 : r[0];  // fewer than sixteen left: repeat the first (OR is idempotent)
          kk &= kk - 1;
        }
        uint64_t w[16];
#pragma unroll
        for (int q = 0; q < 16; q++) w[q] = col[(int64_t)r[q] * cbmax];
#pragma unroll
        for (int q = 0; q < 16; q++) acc |= w[q];
      }
      s_remv[j] = acc;
    }
    __syncthreads();
  }
  if (t == 0) keep_count[seg] = (post_max > 0 && nkept > post_max) ? post_max : nkept;
}

int launch_pairs(int mode, const float* a, int64_t n, const float* b, int64_t m, float* out, int aligned, hipStream_t st) {
  const int64_t total = aligned ? n : n * m;
  if (total <= 0) return PNX_OK;
  PNX_REQUIRE(a && b && out, PNX_ERR_INVALID, "null pointer");
  PNX_REQUIRE(total < ((int64_t)1 << 31) * 256, PNX_ERR_UNSUPPORTED, "too many pairs");
  const unsigned nb = (unsigned)((total + 255) / 256);
  if (mode == MODE_OVERLAP) k_pairs<MODE_OVERLAP><<<nb, 256, 0, st>>>(a, n, b, m, out, aligned);
  else if (mode == MODE_IOU) k_pairs<MODE_IOU><<<nb, 256, 0, st>>>(a, n, b, m, out, aligned);
  else k_pairs<MODE_IOU3D><<<nb, 256, 0, st>>>(a, n, b, m, out, aligned);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

// Host-side layout of the rotated NMS workspace in front of the mask words (all sizes follow from num_segments and max_seg_len).
struct NmsLayout {
  size_t pairs_off, tiles_off, counts_off, mask_off;
  unsigned int pair_cap, tile_cap;
};
int g_nms_pair_cap_dbg = 0;  // 0: the default capacity
NmsLayout nms_layout(int num_segments, int max_seg_len) {
  NmsLayout l;
  const size_t cb = (size_t)((max_seg_len + 63) / 64), slots = (size_t)num_segments * cb * 64;
  size_t pc = slots * 32;  // room for 32 candidate pairs per box on average; what does not fit is done tile by tile
  if (pc > ((size_t)1 << 24)) pc = (size_t)1 << 24;
  if (pc < 4096) pc = 4096;
  // tests: a tiny list sends (almost) every tile down the overflow path (pnx_debug_nms_pair_cap; the environment is NOT read at call time any more -- a
  // workspace sized at one value could be laid out with another, ADVICE r5 -- and the launch checks the workspace against the full layout)
  if (g_nms_pair_cap_dbg > 0) pc = (size_t)(g_nms_pair_cap_dbg > 64 ? g_nms_pair_cap_dbg : 64);
  size_t tc = (size_t)num_segments * cb * (cb + 1) / 2;  // every tile on or above the diagonal
  if (tc > ((size_t)1 << 24)) tc = (size_t)1 << 24;
  l.pair_cap = (unsigned int)pc, l.tile_cap = (unsigned int)(tc < 1 ? 1 : tc);
  l.pairs_off = pnx_align_up(slots * sizeof(BoxPre), 256);
  l.tiles_off = pnx_align_up(l.pairs_off + pc * sizeof(unsigned long long), 256);
  l.counts_off = pnx_align_up(l.tiles_off + (size_t)l.tile_cap * sizeof(unsigned long long), 256);
  l.mask_off = l.counts_off + 256;
  return l;
}

template <bool ROTATED>
int nms_batched(const float* boxes, const int32_t* seg_offsets, const int32_t* seg_len, int32_t num_segments, int32_t max_seg_len, const float* thresh,
                int32_t post_max, int32_t* keep, int32_t* keep_count, void* workspace, size_t workspace_bytes, hipStream_t st) {
  PNX_REQUIRE(num_segments >= 0 && max_seg_len >= 0, PNX_ERR_INVALID, "negative sizes");
  if (num_segments == 0) return PNX_OK;
  PNX_REQUIRE(seg_offsets && thresh && keep_count, PNX_ERR_INVALID, "null pointer");
  if (max_seg_len == 0) {
    PNX_CHECK_HIP(hipMemsetAsync(keep_count, 0, sizeof(int32_t) * num_segments, st));
    return PNX_OK;
  }
  PNX_REQUIRE(boxes && keep && workspace, PNX_ERR_INVALID, "null pointer");
  PNX_REQUIRE(num_segments <= 65535, PNX_ERR_UNSUPPORTED, "more than 65535 segments");
  const int cbmax = (max_seg_len + 63) / 64;
  PNX_REQUIRE(cbmax * 8 <= 64 * 1024, PNX_ERR_UNSUPPORTED, "segment longer than 524288 boxes");
  // workspace_bytes is validated by the caller against pnx_nms_workspace_bytes(total, ...): we cannot know
  // `total` here without reading seg_offsets back, so only the alignment is checked.
  PNX_REQUIRE(((uintptr_t)workspace & 7) == 0 && workspace_bytes >= 8, PNX_ERR_WORKSPACE, "bad workspace");
  uint64_t* mask = (uint64_t*)workspace;
  dim3 grid(cbmax, cbmax, num_segments);
  if (ROTATED) {
    // workspace = [segment slots of BoxPre | pair list | overflow tile list | counters | mask words]; every size is known on the host
    const NmsLayout lay = nms_layout(num_segments, max_seg_len);
    PNX_REQUIRE(workspace_bytes > lay.mask_off + 8, PNX_ERR_WORKSPACE, "workspace smaller than pnx_nms_workspace_bytes");
    PNX_REQUIRE(max_seg_len < (1 << 20) && (size_t)num_segments * cbmax * (cbmax + 1) / 2 <= ((size_t)1 << 24), PNX_ERR_UNSUPPORTED,
                "rotated NMS: segments of at most 2^20 boxes, at most 2^24 mask tiles in all");
    char* wsb = reinterpret_cast<char*>(workspace);
    BoxPre* pre = reinterpret_cast<BoxPre*>(wsb);
    NmsLists L;
    L.pairs = reinterpret_cast<unsigned long long*>(wsb + lay.pairs_off);
    L.tiles = reinterpret_cast<unsigned long long*>(wsb + lay.tiles_off);
    L.counts = reinterpret_cast<unsigned int*>(wsb + lay.counts_off);
    L.pair_cap = lay.pair_cap, L.tile_cap = lay.tile_cap;
    mask = reinterpret_cast<uint64_t*>(wsb + lay.mask_off);
    const size_t n_tiles = (size_t)num_segments * cbmax * (cbmax + 1) / 2;
    const int strip = n_tiles <= 2048 ? 4 : (n_tiles <= 8192 ? 8 : kStrip);
    k_box_pre<<<dim3((unsigned)((max_seg_len + 255) / 256), (unsigned)num_segments), 256, 0, st>>>(boxes, seg_offsets, seg_len, cbmax, pre, mask, L.counts);
    k_nms_cand<<<dim3((unsigned)((cbmax + strip - 1) / strip), (unsigned)cbmax, (unsigned)num_segments), 256, 0, st>>>(pre, seg_offsets, seg_len, thresh, cbmax,
                                                                                                                      strip, L);
    k_nms_pairs<<<1024, 256, 0, st>>>(pre, seg_offsets, thresh, mask, cbmax, L);
    k_nms_tiles<<<2048, 64, 0, st>>>(pre, seg_offsets, seg_len, thresh, mask, cbmax, L);
  } else {
    k_nms_mask<false><<<grid, 64, 0, st>>>(boxes, seg_offsets, seg_len, thresh, mask, cbmax);
  }
  k_nms_greedy<<<num_segments, 64, cbmax * sizeof(uint64_t), st>>>(mask, seg_offsets, seg_len, cbmax, post_max, keep, keep_count);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // namespace

extern "C" {

int pnx_boxes_overlap_bev(const float* a, int64_t n, const float* b, int64_t m, float* out, pnx_stream_t s) {
  return launch_pairs(MODE_OVERLAP, a, n, b, m, out, 0, (hipStream_t)s);
}
int pnx_boxes_iou_bev(const float* a, int64_t n, const float* b, int64_t m, float* out, pnx_stream_t s) {
  return launch_pairs(MODE_IOU, a, n, b, m, out, 0, (hipStream_t)s);
}
int pnx_boxes_aligned_overlap_bev(const float* a, const float* b, int64_t n, float* out, pnx_stream_t s) {
  return launch_pairs(MODE_OVERLAP, a, n, b, n, out, 1, (hipStream_t)s);
}
int pnx_boxes_aligned_iou3d(const float* a, const float* b, int64_t n, float* out, pnx_stream_t s) {
  return launch_pairs(MODE_IOU3D, a, n, b, n, out, 1, (hipStream_t)s);
}

int32_t pnx_debug_nms_pair_cap(int32_t cap) {  // test hook: capacity of the rotated NMS's candidate-pair list (0 = default); returns the previous value
  const int prev = g_nms_pair_cap_dbg;
  g_nms_pair_cap_dbg = cap > 0 ? cap : 0;
  return prev;
}

size_t pnx_nms_workspace_bytes(int64_t total_boxes, int32_t num_segments, int32_t max_seg_len) {
  if (total_boxes <= 0 || max_seg_len <= 0) return 8;
  // (rotated NMS) per-box precompute, candidate-pair list, overflow tile list, counters -- then the mask words
  return nms_layout(num_segments > 0 ? num_segments : 0, max_seg_len).mask_off + (size_t)total_boxes * (size_t)((max_seg_len + 63) / 64) * sizeof(uint64_t) + 8;
}

int pnx_nms_rotated_batched(const float* boxes, const int32_t* seg_offsets, const int32_t* seg_len, int32_t num_segments, int32_t max_seg_len,
                            const float* thresh, int32_t post_max, int32_t* keep, int32_t* keep_count, void* workspace,
                            size_t workspace_bytes, pnx_stream_t stream) {
  return nms_batched<true>(boxes, seg_offsets, seg_len, num_segments, max_seg_len, thresh, post_max, keep, keep_count, workspace, workspace_bytes,
                           (hipStream_t)stream);
}
int pnx_nms_normal_batched(const float* boxes, const int32_t* seg_offsets, const int32_t* seg_len, int32_t num_segments, int32_t max_seg_len,
                           const float* thresh, int32_t post_max, int32_t* keep, int32_t* keep_count, void* workspace,
                           size_t workspace_bytes, pnx_stream_t stream) {
  return nms_batched<false>(boxes, seg_offsets, seg_len, num_segments, max_seg_len, thresh, post_max, keep, keep_count, workspace, workspace_bytes,
                            (hipStream_t)stream);
}

}  // extern "C"
