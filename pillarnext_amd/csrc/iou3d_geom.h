// iou3d_geom.h -- the rotated-rectangle arithmetic of det3d/core/iou3d_nms/src/iou3d_nms_kernel.cu:35-234 ("cu:") shared by the HIP kernels
// (iou3d.hip) and the host entry points (iou3d_host.cpp: the _cpu twins of det3d/core/iou3d_nms/src/iou3d_cpu.cpp:232-273).  One source, fp32
// operation for operation, no contraction (both includers are built with -ffp-contract=off), cos / sin / atan2 from pnx_detmath.h: device and
// host results are bit-identical.  The includer defines PNX_GEOM (function qualifiers) and PNX_HD (pnx_detmath.h) before including.
#pragma once
#include <math.h>
#include <stdint.h>

#include "pnx_detmath.h"

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

PNX_GEOM BoxPre make_box(const float* __restrict__ b) {
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

PNX_GEOM float cross3(Pt p1, Pt p2, Pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

PNX_GEOM bool rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {  // cu:43-49
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) && fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) &&
         fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

PNX_GEOM bool seg_isect(Pt p1, Pt p0, Pt q1, Pt q0, Pt* ans) {  // cu:63-92
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

PNX_GEOM bool in_box(const BoxPre& b, Pt p) {  // cu:51-61
  const float rot_x = (p.x - b.cx) * b.cn + (p.y - b.cy) * (-b.sn);
  const float rot_y = (p.x - b.cx) * b.sn + (p.y - b.cy) * b.cn;
  return fabsf(rot_x) < b.hxm && fabsf(rot_y) < b.hym;
}

// Overlap area of two rotated rectangles (cu:104-225).  spx/spy/sang: lane-major LDS scratch,
// element k of this thread at [k * STRIDE + tid].
// Boxes whose bounding circles (inflated well beyond the 1e-2 in-box margin and any rounding) do not touch
// have no edge crossing and no contained corner: the reference computes cnt = 0 -> area exactly 0.
PNX_GEOM bool far_apart(const BoxPre& A, const BoxPre& B) {
  const float ddx = A.cx - B.cx, ddy = A.cy - B.cy, rr = A.rad + B.rad;
  return ddx * ddx + ddy * ddy > rr * rr;  // false for NaN -> full path
}

template <int STRIDE>
PNX_GEOM float box_overlap(const BoxPre& A, const BoxPre& B, float* spx, float* spy, float* sang, int tid) {
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
PNX_GEOM float iou_bev(const BoxPre& A, const BoxPre& B, float* spx, float* spy, float* sang, int tid) {  // cu:227-234
  const float s = box_overlap<STRIDE>(A, B, spx, spy, sang, tid);
  return s / fmaxf(A.area + B.area - s, kEps);
}

PNX_GEOM float iou_normal(const float* a, const float* b) {  // cu:327-338
  const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  const float interS = width * height;
  const float Sa = a[3] * a[4];
  const float Sb = b[3] * b[4];
  return interS / fmaxf(Sa + Sb - interS, kEps);
}
