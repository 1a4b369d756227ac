// iou3d_host.cpp -- host twins of the rotated-IoU entry points: det3d/core/iou3d_nms/src/iou3d_cpu.cpp:232-273 (boxes_iou_bev_cpu,
// boxes_aligned_iou_bev_cpu) work on CPU tensors without a GPU.  Same source as the device kernels (iou3d_geom.h) compiled for the
// host with the same flags (no contraction, pnx_detmath.h transcendentals): pnx_boxes_iou_bev_cpu(a, b) == pnx_boxes_iou_bev(a, b) bit for bit.
// Plain loops over the pairs, rows dealt to std::thread workers for large inputs: this is the reference's double loop (:243-249), not a hot path.
#pragma clang fp contract(off)
#include <thread>
#include <vector>

#include "pnx_common.h"

#define PNX_HD static inline
#define PNX_GEOM static inline
#include "iou3d_geom.h"

namespace {

void rows_iou(const float* a, int64_t r0, int64_t r1, const float* b, int64_t m, float* out) {
  float spx[kMaxPts], spy[kMaxPts], sang[kMaxPts];
  std::vector<BoxPre> B((size_t)m);
  for (int64_t j = 0; j < m; j++) B[(size_t)j] = make_box(b + j * 7);
  for (int64_t i = r0; i < r1; i++) {
    const BoxPre A = make_box(a + i * 7);
    for (int64_t j = 0; j < m; j++) out[i * m + j] = iou_bev<1>(A, B[(size_t)j], spx, spy, sang, 0);
  }
}

}  // namespace

extern "C" int pnx_boxes_iou_bev_cpu(const float* boxes_a_host, int64_t n, const float* boxes_b_host, int64_t m, float* out_host) {
  PNX_REQUIRE(n >= 0 && m >= 0, PNX_ERR_INVALID, "negative sizes");
  if (n == 0 || m == 0) return PNX_OK;
  PNX_REQUIRE(boxes_a_host && boxes_b_host && out_host, PNX_ERR_INVALID, "null pointer");
  unsigned nt = std::thread::hardware_concurrency();
  if (nt > 32) nt = 32;
  if (n * m < 65536 || nt < 2) {
    rows_iou(boxes_a_host, 0, n, boxes_b_host, m, out_host);
    return PNX_OK;
  }
  std::vector<std::thread> th;
  const int64_t per = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; t++) {
    const int64_t r0 = (int64_t)t * per, r1 = r0 + per < n ? r0 + per : n;
    if (r0 < r1) th.emplace_back(rows_iou, boxes_a_host, r0, r1, boxes_b_host, m, out_host);
  }
  for (auto& x : th) x.join();
  return PNX_OK;
}

extern "C" int pnx_boxes_aligned_iou_bev_cpu(const float* boxes_a_host, const float* boxes_b_host, int64_t n, float* out_host) {
  PNX_REQUIRE(n >= 0, PNX_ERR_INVALID, "negative size");
  if (n == 0) return PNX_OK;
  PNX_REQUIRE(boxes_a_host && boxes_b_host && out_host, PNX_ERR_INVALID, "null pointer");
  float spx[kMaxPts], spy[kMaxPts], sang[kMaxPts];
  for (int64_t i = 0; i < n; i++) out_host[i] = iou_bev<1>(make_box(boxes_a_host + i * 7), make_box(boxes_b_host + i * 7), spx, spy, sang, 0);
  return PNX_OK;
}
