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

// Rotated variant.  The reference lets every lane walk its 64 columns serially through the full polygon clipping
// (cu:311-321), although almost all pairs are far apart.  Here a tile works in two phases:
//   1. every lane (= row) builds the 64-bit set of columns whose bounding circles touch its own (a dozen flops per pair);
//   2. the surviving (row, col) pairs of the whole tile are packed into one LDS list and dealt out evenly to the 64 lanes,
//      so the expensive clipping runs with all lanes busy; results are OR-ed into the row masks with LDS atomics.
// Pairs rejected in phase 1 have overlap exactly 0 in the reference too (cnt = 0), so the mask is bit-identical.
// Per-box work once per BOX, not once per tile a box takes part in (a 1000-box list has 136 tiles: every box was set up 17 times, two fp64 sincos
// each): one thread per box writes its BoxPre (64 bytes) into the workspace, the mask kernel reads them back with coalesced 16-byte loads.
// pre[seg * cbmax * 64 + i] = box i of segment seg (the slots a tile of the mask kernel reads are exactly the ones written here).
__global__ __launch_bounds__(256) void k_box_pre(const float* __restrict__ boxes, const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_len,
                                                 int cbmax, BoxPre* __restrict__ pre) {
  const int seg = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i < seg_size(seg_off, seg_len, seg)) pre[(int64_t)seg * cbmax * 64 + i] = make_box(boxes + (int64_t)(seg_off[seg] + i) * 7);
}

__global__ __launch_bounds__(64) void k_nms_mask_rot(const BoxPre* __restrict__ pre, const int32_t* __restrict__ seg_off,
                                                     const int32_t* __restrict__ seg_len, const float* __restrict__ thresh,
                                                     uint64_t* __restrict__ mask, int cbmax) {
  const int seg = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  const int off = seg_off[seg], n = seg_size(seg_off, seg_len, seg);
  if (rb * 64 >= n || cb * 64 >= n) return;
  const float thr = thresh[seg];
  const int row_size = min(n - rb * 64, 64), col_size = min(n - cb * 64, 64);
  const int t = threadIdx.x;
  __shared__ float4 s_ccr[64];  // centre and bounding radius of the column boxes: all phase 1 needs (one broadcast 16-byte read per column)
  __shared__ unsigned short s_pairs[4096];
  __shared__ unsigned int s_bits[64 * 2];
  __shared__ float spx[kMaxPts * 64], spy[kMaxPts * 64], sang[kMaxPts * 64];
  const BoxPre* prow = pre + (int64_t)seg * cbmax * 64 + rb * 64;
  const BoxPre* pcol = pre + (int64_t)seg * cbmax * 64 + cb * 64;
  float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < col_size) s_ccr[t] = make_float4(pcol[t].cx, pcol[t].cy, pcol[t].rad, 0.f);
  if (t < row_size) me = make_float4(prow[t].cx, prow[t].cy, prow[t].rad, 0.f);
  s_bits[t] = 0;
  s_bits[64 + t] = 0;
  __syncthreads();
  // phase 1: candidate columns of my row (far_apart on the three numbers; same expression, same result)
  uint64_t cand = 0;
  if (t < row_size) {
    const int start = (rb == cb) ? t + 1 : 0;
    if (thr < 0.f) {  // iou 0 > thr only for a negative threshold: then every pair is a candidate
      for (int k = start; k < col_size; k++) cand |= 1ULL << k;
    } else {
#pragma unroll 8
      for (int k = 0; k < 64; k++) {
        const float4 c = s_ccr[k < col_size ? k : 0];
        const float ddx = me.x - c.x, ddy = me.y - c.y, rr = me.z + c.z;
        const bool far = ddx * ddx + ddy * ddy > rr * rr;  // false for NaN -> full path
        if (k >= start && k < col_size && !far) cand |= 1ULL << k;
      }
    }
  }
  // pack the pairs: exclusive prefix of the per-row counts across the wave
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
  // phase 2: the real work, evenly spread; the two boxes of a pair come from the precomputed array (L1 / L2: 128 boxes per tile)
  for (int e = t; e < total; e += 64) {
    const int pr = s_pairs[e];
    const int i = pr >> 6, k = pr & 63;
    const BoxPre A = prow[i], B = pcol[k];
    const float v = iou_bev<64>(A, B, spx, spy, sang, t);  // row box first, as cu:317
    if (v > thr) atomicOr(&s_bits[2 * i + (k >> 5)], 1u << (k & 31));
  }
  __syncthreads();
  if (t < row_size) mask[(int64_t)(off + rb * 64 + t) * cbmax + cb] = (uint64_t)s_bits[2 * t] | ((uint64_t)s_bits[2 * t + 1] << 32);
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
  for (int nb = 0; nb < cb; nb++) {
    const int bsz = min(64, n - nb * 64);
    // diagonal words of this block: lane t holds the in-block suppression set of box nb*64+t
    const uint64_t diag = (t < bsz) ? mask[(int64_t)(off + nb * 64 + t) * cbmax + nb] : 0ULL;
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
    // fold the kept rows into the words still to come
    for (int j = nb + 1 + t; j < cb; j += 64) {
      uint64_t acc = s_remv[j];
      uint64_t kk = kept;
      while (kk) {
        const int k = __ffsll((long long)kk) - 1;
        kk &= kk - 1;
        acc |= mask[(int64_t)(off + nb * 64 + k) * cbmax + j];
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

size_t nms_pre_bytes(int num_segments, int max_seg_len) { return (size_t)num_segments * (size_t)((max_seg_len + 63) / 64) * 64 * sizeof(BoxPre); }

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
    // workspace = [num_segments * cbmax * 64 BoxPre | mask words]: the per-box precompute in front (its size is known on the host), the mask behind
    const size_t pre_bytes = nms_pre_bytes(num_segments, max_seg_len);
    PNX_REQUIRE(workspace_bytes > pre_bytes + 8, PNX_ERR_WORKSPACE, "workspace smaller than pnx_nms_workspace_bytes");
    BoxPre* pre = reinterpret_cast<BoxPre*>(workspace);
    mask = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(workspace) + pre_bytes);
    k_box_pre<<<dim3((unsigned)((max_seg_len + 255) / 256), (unsigned)num_segments), 256, 0, st>>>(boxes, seg_offsets, seg_len, cbmax, pre);
    k_nms_mask_rot<<<grid, 64, 0, st>>>(pre, seg_offsets, seg_len, thresh, mask, cbmax);
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

size_t pnx_nms_workspace_bytes(int64_t total_boxes, int32_t num_segments, int32_t max_seg_len) {
  if (total_boxes <= 0 || max_seg_len <= 0) return 8;
  // (rotated NMS) the per-box precompute of every segment slot, then the mask words
  return nms_pre_bytes(num_segments > 0 ? num_segments : 0, max_seg_len) + (size_t)total_boxes * (size_t)((max_seg_len + 63) / 64) * sizeof(uint64_t) + 8;
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
