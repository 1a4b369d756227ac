/* orc_iou_impl.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of the reference's rotated-box BEV overlap / IoU / NMS arithmetic.
 * Source of truth (read, not copied): det3d/core/iou3d_nms/src/iou3d_nms_kernel.cu and its
 * host twin det3d/core/iou3d_nms/src/iou3d_cpu.cpp (same arithmetic, "cu:" / "cpu:" below).
 *
 * This file is included twice by pnx_oracle.c:
 *   ORC_SUFFIX = _libm : cosf/sinf/atan2f from the host libm  -> bit-identical to the compiled
 *                        reference (oracle/_ref, glibc) ; this is the variant that PINS the oracle.
 *   ORC_SUFFIX = _det  : pnx_detmath.h routines                -> bit-identical to the HIP kernels.
 * All arithmetic is fp32, evaluated in the reference's order; build with -ffp-contract=off.
 */

#define ORC_CAT2(a, b) a##b
#define ORC_CAT(a, b) ORC_CAT2(a, b)
#define ORC_FN(name) ORC_CAT(name, ORC_SUFFIX)

/* cu:35-41 / cpu:59-65 -- 2-D cross products */
static inline float ORC_FN(cross2)(orc_pt a, orc_pt b) { return a.x * b.y - a.y * b.x; }
static inline float ORC_FN(cross3)(orc_pt p1, orc_pt p2, orc_pt p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

/* cu:43-49 / cpu:67-73 -- inclusive bounding-box rejection test */
static inline int ORC_FN(rect_cross)(orc_pt p1, orc_pt p2, orc_pt q1, orc_pt q2) {
  return orc_minf(p1.x, p2.x) <= orc_maxf(q1.x, q2.x) && orc_minf(q1.x, q2.x) <= orc_maxf(p1.x, p2.x) &&
         orc_minf(p1.y, p2.y) <= orc_maxf(q1.y, q2.y) && orc_minf(q1.y, q2.y) <= orc_maxf(p1.y, p2.y);
}

/* cu:51-61 / cpu:75-85 -- point-in-rotated-box with a 1e-2 margin; rotates by -heading */
static inline int ORC_FN(in_box2d)(const float* box, orc_pt p) {
  const float MARGIN = 1e-2f;
  float cx = box[0], cy = box[1];
  float angle_cos, angle_sin;
  ORC_SINCOS(-box[6], &angle_sin, &angle_cos);
  float rot_x = (p.x - cx) * angle_cos + (p.y - cy) * (-angle_sin);
  float rot_y = (p.x - cx) * angle_sin + (p.y - cy) * angle_cos;
  return (fabsf(rot_x) < box[3] / 2 + MARGIN && fabsf(rot_y) < box[4] / 2 + MARGIN);
}

/* cu:63-92 / cpu:87-118 -- proper segment intersection (strict straddle test) */
static inline int ORC_FN(seg_isect)(orc_pt p1, orc_pt p0, orc_pt q1, orc_pt q0, orc_pt* ans) {
  if (!ORC_FN(rect_cross)(p0, p1, q0, q1)) return 0;
  float s1 = ORC_FN(cross3)(q0, p1, p0);
  float s2 = ORC_FN(cross3)(p1, q1, p0);
  float s3 = ORC_FN(cross3)(p0, q1, q0);
  float s4 = ORC_FN(cross3)(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = ORC_FN(cross3)(q1, p1, p0);
  if (fabsf(s5 - s1) > ORC_EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

/* cu:94-98 / cpu:120-124 */
static inline orc_pt ORC_FN(rot_about)(orc_pt c, float ac, float as, orc_pt p) {
  orc_pt r;
  r.x = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
  r.y = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
  return r;
}

/* cu:104-225 / cpu:128-220 -- overlap area of two rotated rectangles */
static float ORC_FN(orc_box_overlap)(const float* box_a, const float* box_b) {
  float a_angle = box_a[6], b_angle = box_b[6];
  float a_dx_half = box_a[3] / 2, b_dx_half = box_b[3] / 2;
  float a_dy_half = box_a[4] / 2, b_dy_half = box_b[4] / 2;
  float a_x1 = box_a[0] - a_dx_half, a_y1 = box_a[1] - a_dy_half;
  float a_x2 = box_a[0] + a_dx_half, a_y2 = box_a[1] + a_dy_half;
  float b_x1 = box_b[0] - b_dx_half, b_y1 = box_b[1] - b_dy_half;
  float b_x2 = box_b[0] + b_dx_half, b_y2 = box_b[1] + b_dy_half;
  orc_pt ca = {box_a[0], box_a[1]}, cb = {box_b[0], box_b[1]};

  orc_pt A[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
  orc_pt B[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};
  float a_cos, a_sin, b_cos, b_sin;
  ORC_SINCOS(a_angle, &a_sin, &a_cos);
  ORC_SINCOS(b_angle, &b_sin, &b_cos);
  for (int k = 0; k < 4; k++) {
    A[k] = ORC_FN(rot_about)(ca, a_cos, a_sin, A[k]);
    B[k] = ORC_FN(rot_about)(cb, b_cos, b_sin, B[k]);
  }
  A[4] = A[0];
  B[4] = B[0];

  /* the reference declares cross_points[16]; geometry yields <= 16 in practice, we keep room
     for the theoretical 24 so an adversarial input cannot smash the oracle's stack */
  orc_pt cp[24];
  orc_pt centre = {0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < 4; j++) {
      if (ORC_FN(seg_isect)(A[i + 1], A[i], B[j + 1], B[j], &cp[cnt])) {
        centre.x = centre.x + cp[cnt].x;
        centre.y = centre.y + cp[cnt].y;
        cnt++;
      }
    }
  }
  for (int k = 0; k < 4; k++) { /* interleaved order: B[k] in A, then A[k] in B  (cu:177-194) */
    if (ORC_FN(in_box2d)(box_a, B[k])) {
      centre.x = centre.x + B[k].x;
      centre.y = centre.y + B[k].y;
      cp[cnt++] = B[k];
    }
    if (ORC_FN(in_box2d)(box_b, A[k])) {
      centre.x = centre.x + A[k].x;
      centre.y = centre.y + A[k].y;
      cp[cnt++] = A[k];
    }
  }
  centre.x /= cnt; /* cnt==0 -> NaN centre; loops below do not run (cu:196-197) */
  centre.y /= cnt;

  /* bubble sort ascending by atan2 about the centroid, comparator a > b  (cu:100-102,200-209) */
  for (int j = 0; j < cnt - 1; j++) {
    for (int i = 0; i < cnt - j - 1; i++) {
      float ang_i = ORC_ATAN2(cp[i].y - centre.y, cp[i].x - centre.x);
      float ang_n = ORC_ATAN2(cp[i + 1].y - centre.y, cp[i + 1].x - centre.x);
      if (ang_i > ang_n) {
        orc_pt t = cp[i];
        cp[i] = cp[i + 1];
        cp[i + 1] = t;
      }
    }
  }
  /* fan shoelace about cp[0]  (cu:219-224) */
  float area = 0.f;
  for (int k = 0; k < cnt - 1; k++) {
    orc_pt u = {cp[k].x - cp[0].x, cp[k].y - cp[0].y};
    orc_pt v = {cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y};
    area += ORC_FN(cross2)(u, v);
  }
  return fabsf(area) / 2.0f;
}

/* cu:227-234 / cpu:222-229 */
static float ORC_FN(orc_iou_bev1)(const float* a, const float* b) {
  float sa = a[3] * a[4];
  float sb = b[3] * b[4];
  float s = ORC_FN(orc_box_overlap)(a, b);
  return s / fmaxf(sa + sb - s, ORC_EPS);
}

/* N x M overlap areas  (boxes_overlap_kernel cu:236-249) */
void ORC_FN(orc_boxes_overlap_bev)(const float* a, int64_t n, const float* b, int64_t m, float* out) {
  for (int64_t i = 0; i < n; i++)
    for (int64_t j = 0; j < m; j++) out[i * m + j] = ORC_FN(orc_box_overlap)(a + i * 7, b + j * 7);
}
/* N x M IoU-BEV  (boxes_iou_bev_kernel cu:264-278 ; boxes_iou_bev_cpu cpu:232-252) */
void ORC_FN(orc_boxes_iou_bev)(const float* a, int64_t n, const float* b, int64_t m, float* out) {
  for (int64_t i = 0; i < n; i++)
    for (int64_t j = 0; j < m; j++) out[i * m + j] = ORC_FN(orc_iou_bev1)(a + i * 7, b + j * 7);
}
/* pairwise overlap area  (boxes_aligned_overlap_kernel cu:251-262) */
void ORC_FN(orc_boxes_aligned_overlap_bev)(const float* a, const float* b, int64_t n, float* out) {
  for (int64_t i = 0; i < n; i++) out[i] = ORC_FN(orc_box_overlap)(a + i * 7, b + i * 7);
}
/* pairwise IoU-BEV  (boxes_aligned_iou_bev_cpu cpu:254-273) */
void ORC_FN(orc_boxes_aligned_iou_bev)(const float* a, const float* b, int64_t n, float* out) {
  for (int64_t i = 0; i < n; i++) out[i] = ORC_FN(orc_iou_bev1)(a + i * 7, b + i * 7);
}

/* Rotated NMS = bitmask (nms_kernel cu:280-324: bit t of word c in row i set iff
 * iou_bev(box_i, box_{64c+t}) > thresh, diagonal tile only for t > i-64c) followed by the greedy
 * host scan of nms_gpu (iou3d_nms.cpp:139-155).  Boxes must already be score-sorted.
 * mask_out (optional) receives the n x ceil(n/64) words; lower-triangle tiles are computed like the
 * reference does (cu:288 keeps the early-return commented out).  Returns the number kept. */
int64_t ORC_FN(orc_nms_rotated)(const float* boxes, int64_t n, float thresh, int64_t* keep, uint64_t* mask_out) {
  int64_t cb = (n + 63) / 64;
  uint64_t* mask = mask_out ? mask_out : (uint64_t*)calloc((size_t)(n * cb + 1), sizeof(uint64_t));
  for (int64_t i = 0; i < n; i++) {
    for (int64_t c = 0; c < cb; c++) {
      int64_t col_size = n - c * 64 < 64 ? n - c * 64 : 64;
      int64_t start = (i / 64 == c) ? (i % 64) + 1 : 0;
      uint64_t t = 0;
      for (int64_t k = start; k < col_size; k++)
        if (ORC_FN(orc_iou_bev1)(boxes + i * 7, boxes + (c * 64 + k) * 7) > thresh) t |= 1ULL << k;
      mask[i * cb + c] = t;
    }
  }
  int64_t nk = orc_greedy_scan(mask, n, keep);
  if (!mask_out) free(mask);
  return nk;
}

#undef ORC_FN
#undef ORC_CAT
#undef ORC_CAT2
