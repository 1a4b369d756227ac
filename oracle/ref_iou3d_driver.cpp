// ref_iou3d_driver.cpp -- TEST INFRASTRUCTURE.  A C-ABI doorway into the reference's OWN CPU
// rotated-IoU code (det3d/core/iou3d_nms/src/iou3d_cpu.cpp:232-273), compiled in place from
// /root/reference by oracle/Makefile into oracle/_ref/libref_iou3d.so.  Nothing from the reference
// is copied: this file only declares the two functions that iou3d_cpu.h:9-10 exports and wraps raw
// float buffers as at::Tensor views for them.
#include <torch/extension.h>

int boxes_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);
int boxes_aligned_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);

static at::Tensor view(const float* p, int64_t r, int64_t c) {
  return torch::from_blob(const_cast<float*>(p), {r, c}, torch::TensorOptions().dtype(torch::kFloat32));
}

extern "C" int ref_boxes_iou_bev_cpu(const float* a, int64_t n, const float* b, int64_t m, float* out) {
  return boxes_iou_bev_cpu(view(a, n, 7), view(b, m, 7), view(out, n, m));
}
extern "C" int ref_boxes_aligned_iou_bev_cpu(const float* a, const float* b, int64_t n, float* out) {
  return boxes_aligned_iou_bev_cpu(view(a, n, 7), view(b, n, 7), view(out, n, 1));
}
