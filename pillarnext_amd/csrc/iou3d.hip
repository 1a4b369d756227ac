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
#include "pnx_detmath.h"

namespace {

constexpr float kEps = 1e-8f;
constexpr int kMaxPts = 16;  // 2 convex quadrilaterals: <= 8 edge crossings + <= 8 contained corners

struct Pt {
  float x, y;
};

// Everything box_overlap needs from one box.
struct BoxPre {
  float cx, cy;    // centre
  float hxm, hym;  // dx/2 + MARGIN, dy/2 + MARGIN            (cu:52,60)
  float cn, sn;    // cos(-heading), sin(-heading)            (cu:56)
  Pt c[4];         // rotated corners                         (cu:124-149)
  float area;      // dx*dy                                   (cu:230-231)
  float rad;       // conservative bounding radius (half diagonal + margin + slack) for the exact-zero early out
};

__device__ __forceinline__ BoxPre make_box(const float* __restrict__ b) {
  BoxPre o;
  const float MARGIN = 1e-2f;
  const float x = b[0], y = b[1], dx = b[3], dy = b[4], ang = b[6];
  o.cx = x;
  o.cy = y;
  o.hxm = dx / 2 + MARGIN;
  o.hym = dy / 2 + MARGIN;
  pnx_sincosf(-ang, &o.sn, &o.cn);
  float s, c;
  pnx_sincosf(ang, &s, &c);
  const float dxh = dx / 2, dyh = dy / 2;
  const float x1 = x - dxh, y1 = y - dyh, x2 = x + dxh, y2 = y + dyh;
  const float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
#pragma unroll
  for (int k = 0; k < 4; k++) {  // rotate_around_center (cu:94-98)
    o.c[k].x = (px[k] - x) * c + (py[k] - y) * (-s) + x;
    o.c[k].y = (px[k] - x) * s + (py[k] - y) * c + y;
  }
  o.area = dx * dy;
  o.rad = sqrtf(dxh * dxh + dyh * dyh) * 1.001f + 0.05f;
  return o;
}

__device__ __forceinline__ float cross3(Pt p1, Pt p2, Pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

__device__ __forceinline__ bool rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {  // cu:43-49
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) && fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) &&
         fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

__device__ __forceinline__ bool seg_isect(Pt p1, Pt p0, Pt q1, Pt q0, Pt* ans) {  // cu:63-92
  if (!rect_cross(p0, p1, q0, q1)) return false;
  const float s1 = cross3(q0, p1, p0);
  const float s2 = cross3(p1, q1, p0);
  const float s3 = cross3(p0, q1, q0);
  const float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > kEps) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

__device__ __forceinline__ bool in_box(const BoxPre& b, Pt p) {  // cu:51-61
  const float rot_x = (p.x - b.cx) * b.cn + (p.y - b.cy) * (-b.sn);
  const float rot_y = (p.x - b.cx) * b.sn + (p.y - b.cy) * b.cn;
  return fabsf(rot_x) < b.hxm && fabsf(rot_y) < b.hym;
}

// Overlap area of two rotated rectangles (cu:104-225).  spx/spy/sang: lane-major LDS scratch,
// element k of this thread at [k * STRIDE + tid].
// Boxes whose bounding circles (inflated well beyond the 1e-2 in-box margin and any rounding) do not touch
// have no edge crossing and no contained corner: the reference computes cnt = 0 -> area exactly 0.
__device__ __forceinline__ bool far_apart(const BoxPre& A, const BoxPre& B) {
  const float ddx = A.cx - B.cx, ddy = A.cy - B.cy, rr = A.rad + B.rad;
  return ddx * ddx + ddy * ddy > rr * rr;  // false for NaN -> full path
}

template <int STRIDE>
__device__ __forceinline__ float box_overlap(const BoxPre& A, const BoxPre& B, float* spx, float* spy, float* sang, int tid) {
  if (far_apart(A, B)) return 0.f;
  int cnt = 0;
  float sumx = 0.f, sumy = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      Pt ans;
      if (seg_isect(A.c[(i + 1) & 3], A.c[i], B.c[(j + 1) & 3], B.c[j], &ans)) {
        sumx = sumx + ans.x;
        sumy = sumy + ans.y;
        if (cnt < kMaxPts) {
          spx[cnt * STRIDE + tid] = ans.x;
          spy[cnt * STRIDE + tid] = ans.y;
        }
        cnt++;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {  // B[k] in A, then A[k] in B (cu:177-194)
    if (in_box(A, B.c[k])) {
      sumx = sumx + B.c[k].x;
      sumy = sumy + B.c[k].y;
      if (cnt < kMaxPts) {
        spx[cnt * STRIDE + tid] = B.c[k].x;
        spy[cnt * STRIDE + tid] = B.c[k].y;
      }
      cnt++;
    }
    if (in_box(B, A.c[k])) {
      sumx = sumx + A.c[k].x;
      sumy = sumy + A.c[k].y;
      if (cnt < kMaxPts) {
        spx[cnt * STRIDE + tid] = A.c[k].x;
        spy[cnt * STRIDE + tid] = A.c[k].y;
      }
      cnt++;
    }
  }
  if (cnt > kMaxPts) cnt = kMaxPts;
  if (cnt < 3) return 0.f;  // fewer than 3 vertices: the fan below sums to exactly 0 (cnt==0: loops do not run)
  const float ccx = sumx / cnt, ccy = sumy / cnt;
  for (int k = 0; k < cnt; k++) sang[k * STRIDE + tid] = pnx_atan2f(spy[k * STRIDE + tid] - ccy, spx[k * STRIDE + tid] - ccx);
  // bubble sort ascending, swap iff a > b (cu:200-209)
  for (int j = 0; j < cnt - 1; j++) {
    for (int i = 0; i < cnt - j - 1; i++) {
      const float a0 = sang[i * STRIDE + tid], a1 = sang[(i + 1) * STRIDE + tid];
      if (a0 > a1) {
        sang[i * STRIDE + tid] = a1;
        sang[(i + 1) * STRIDE + tid] = a0;
        const float tx = spx[i * STRIDE + tid], ty = spy[i * STRIDE + tid];
        spx[i * STRIDE + tid] = spx[(i + 1) * STRIDE + tid];
        spy[i * STRIDE + tid] = spy[(i + 1) * STRIDE + tid];
        spx[(i + 1) * STRIDE + tid] = tx;
        spy[(i + 1) * STRIDE + tid] = ty;
      }
    }
  }
  const float x0 = spx[tid], y0 = spy[tid];
  float area = 0.f;
  for (int k = 0; k < cnt - 1; k++) {  // fan shoelace about vertex 0 (cu:219-224)
    const float ux = spx[k * STRIDE + tid] - x0, uy = spy[k * STRIDE + tid] - y0;
    const float vx = spx[(k + 1) * STRIDE + tid] - x0, vy = spy[(k + 1) * STRIDE + tid] - y0;
    area += ux * vy - uy * vx;
  }
  return fabsf(area) / 2.0f;
}

template <int STRIDE>
__device__ __forceinline__ float iou_bev(const BoxPre& A, const BoxPre& B, float* spx, float* spy, float* sang, int tid) {  // cu:227-234
  const float s = box_overlap<STRIDE>(A, B, spx, spy, sang, tid);
  return s / fmaxf(A.area + B.area - s, kEps);
}

__device__ __forceinline__ float iou_normal(const float* a, const float* b) {  // cu:327-338
  const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  const float interS = width * height;
  const float Sa = a[3] * a[4];
  const float Sb = b[3] * b[4];
  return interS / fmaxf(Sa + Sb - interS, kEps);
}

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
__global__ __launch_bounds__(64) void k_nms_mask_rot(const float* __restrict__ boxes, const int32_t* __restrict__ seg_off,
                                                     const int32_t* __restrict__ seg_len, const float* __restrict__ thresh,
                                                     uint64_t* __restrict__ mask, int cbmax) {
  const int seg = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  const int off = seg_off[seg], n = seg_size(seg_off, seg_len, seg);
  if (rb * 64 >= n || cb * 64 >= n) return;
  const float thr = thresh[seg];
  const int row_size = min(n - rb * 64, 64), col_size = min(n - cb * 64, 64);
  const int t = threadIdx.x;
  __shared__ BoxPre s_col[64], s_row[64];
  __shared__ unsigned short s_pairs[4096];
  __shared__ unsigned int s_bits[64 * 2];
  __shared__ float spx[kMaxPts * 64], spy[kMaxPts * 64], sang[kMaxPts * 64];
  if (t < col_size) s_col[t] = make_box(boxes + (int64_t)(off + cb * 64 + t) * 7);
  if (t < row_size) s_row[t] = make_box(boxes + (int64_t)(off + rb * 64 + t) * 7);
  s_bits[t] = 0;
  s_bits[64 + t] = 0;
  __syncthreads();
  // phase 1: candidate columns of my row
  uint64_t cand = 0;
  if (t < row_size) {
    const BoxPre A = s_row[t];
    const int start = (rb == cb) ? t + 1 : 0;
    for (int k = start; k < col_size; k++)
      if (thr < 0.f || !far_apart(A, s_col[k])) cand |= 1ULL << k;  // iou 0 > thr only for a negative threshold
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
  // phase 2: the real work, evenly spread
  for (int e = t; e < total; e += 64) {
    const int pr = s_pairs[e];
    const int i = pr >> 6, k = pr & 63;
    const float v = iou_bev<64>(s_row[i], s_col[k], spx, spy, sang, t);  // row box first, as cu:317
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
  if (ROTATED) k_nms_mask_rot<<<grid, 64, 0, st>>>(boxes, seg_offsets, seg_len, thresh, mask, cbmax);
  else k_nms_mask<false><<<grid, 64, 0, st>>>(boxes, seg_offsets, seg_len, thresh, mask, cbmax);
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
  (void)num_segments;
  if (total_boxes <= 0 || max_seg_len <= 0) return 8;
  return (size_t)total_boxes * (size_t)((max_seg_len + 63) / 64) * sizeof(uint64_t) + 8;
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
