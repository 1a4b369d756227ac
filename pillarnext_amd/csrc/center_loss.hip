// center_loss.hip -- the CenterHead training losses over the (B, M = 500) label lists, forward and backward, for gfx950.
//
// Reference: det3d/models/loss/centerloss.py -- FastFocalLoss :8-37, RegLoss :40-60, IouLoss :63-87, IouRegLoss :90-110 with the
// axis-aligned DIoU of :139-176 -- as called by CenterHead.loss, det3d/models/heads/centerhead.py:142-229; label format of
// det3d/datasets/pipelines/assign.py:23-116 (per task: hm (B,C,H,W), ind / mask / cat (B,M), anno_box (B,M,10), gt_boxes (B,M,7)).
// The reference runs ~40 autograd nodes per task (transpose + gather of every map, boolean-mask compaction with a host sync, the
// dense box decode of EVERY cell for the two IoU losses).  Here, per task:
//   k_focal_neg       streaming reduction of the dense negative focal term over the heat map (HBM-bound: two reads per cell),
//                     deterministic two-level sum; its backward k_focal_neg_bwd writes the dense hm gradient in one pass
//   k_loss_gather     one thread per (b, k): the 10 regression channels, the class logit and the iou logit AT `ind` (NCHW gathers),
//                     the decoded box (centerhead.py:178-195: exp of the clamped size, atan2, cell + offset -> metres) -- only the M
//                     listed cells are decoded, not H x W
//   (pnx_boxes_aligned_iou3d of iou3d.hip turns the decoded boxes into the IoU-loss targets: same arithmetic as the reference's native op)
//   k_loss_reduce     ONE workgroup: positive focal term, per-channel L1 with the NaN-target rule, IoU L1, DIoU -> the four losses,
//                     normalised on the device (no .item(), no boolean indexing); fixed summation order
//   k_loss_scatter    backward of all of the above for the listed cells: atomic adds into the (zeroed) gradients of the maps --
//                     several objects may share a cell
// fp32 NCHW maps (the training graph's own layout).  All sums run in fp64 inside k_loss_reduce / the second level of k_focal_neg.
#include "pnx_common.h"

namespace {

constexpr int kLB = 256;

struct LossMaps {  // the head outputs of one task, (B, C, H, W) fp32 each
  const float *hm, *reg, *height, *dim, *rot, *vel, *iou;  // iou may be null
};
struct LossGrads {
  float *hm, *reg, *height, *dim, *rot, *vel, *iou;
};
struct LossGeom {
  int B, ncls, H, W, M;
  float kx, ky, minx, miny;  // metres per head cell (out_size_factor * voxel_size) and the range origin
};

__device__ __forceinline__ float sigmoid_clamped(float x) {
  const float p = 1.0f / (1.0f + expf(-x));
  return fminf(fmaxf(p, 1e-4f), 1.0f - 1e-4f);
}

// ---- dense negative focal term: sum p^2 (1 - gt)^4 log(1 - p)
__global__ __launch_bounds__(kLB) void k_focal_neg(const float* __restrict__ hm, const float* __restrict__ gt, int64_t n, double* __restrict__ part) {
  __shared__ double s_w[kLB / 64];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kLB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLB) {
    const float p = sigmoid_clamped(hm[i]);
    const float g = 1.0f - gt[i];
    const float g2 = g * g;
    acc += (double)(p * p * (g2 * g2) * logf(1.0f - p));
  }
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kLB / 64; w++) t += s_w[w];
    part[blockIdx.x] = t;
  }
}

// gradient of  coef * sum(...)  with respect to the logits; coef is a device scalar (sign and 1/num_pos folded in by the caller)
__global__ __launch_bounds__(kLB) void k_focal_neg_bwd(const float* __restrict__ hm, const float* __restrict__ gt, int64_t n, const float* __restrict__ coef,
                                                       float* __restrict__ ghm) {
  const float c = coef[0];
  for (int64_t i = (int64_t)blockIdx.x * kLB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLB) {
    const float x = hm[i];
    const float ps = 1.0f / (1.0f + expf(-x));
    float gr = 0.f;
    if (ps > 1e-4f && ps < 1.0f - 1e-4f) {  // torch.clamp passes the gradient inside the range only
      const float g = 1.0f - gt[i];
      const float g4 = (g * g) * (g * g);
      const float q = 1.0f - ps;
      // d/dp [p^2 g4 log(1-p)] * dp/dx,  dp/dx = p (1 - p)
      gr = g4 * (2.0f * ps * logf(q) - ps * ps / q) * (ps * q);
    }
    ghm[i] = c * gr;
  }
}

// gathered record of one (b, k): [0..9] regression prediction in anno_box order (reg 2, height 1, dim 3, vel 2, rot 2),
// [10] class logit at (b, cat, ind), [11] iou logit, [12..18] decoded box x y z dx dy dz r
constexpr int kGW = 20;

__global__ __launch_bounds__(kLB) void k_loss_gather(LossMaps m, LossGeom g, const int64_t* __restrict__ ind, const int64_t* __restrict__ cat,
                                                     float* __restrict__ gathered, float* __restrict__ boxes7) {
  const int i = blockIdx.x * kLB + threadIdx.x;
  if (i >= g.B * g.M) return;
  const int b = i / g.M;
  const int64_t HW = (int64_t)g.H * g.W;
  int64_t cell = ind[i];
  cell = cell < 0 ? 0 : (cell >= HW ? HW - 1 : cell);
  int c = (int)cat[i];
  c = c < 0 ? 0 : (c >= g.ncls ? g.ncls - 1 : c);
  float* o = gathered + (int64_t)i * kGW;
  const float reg0 = m.reg[((int64_t)b * 2 + 0) * HW + cell], reg1 = m.reg[((int64_t)b * 2 + 1) * HW + cell];
  const float hgt = m.height[(int64_t)b * HW + cell];
  const float d0 = m.dim[((int64_t)b * 3 + 0) * HW + cell], d1 = m.dim[((int64_t)b * 3 + 1) * HW + cell], d2 = m.dim[((int64_t)b * 3 + 2) * HW + cell];
  const float v0 = m.vel[((int64_t)b * 2 + 0) * HW + cell], v1 = m.vel[((int64_t)b * 2 + 1) * HW + cell];
  const float r0 = m.rot[((int64_t)b * 2 + 0) * HW + cell], r1 = m.rot[((int64_t)b * 2 + 1) * HW + cell];
  o[0] = reg0, o[1] = reg1, o[2] = hgt, o[3] = d0, o[4] = d1, o[5] = d2, o[6] = v0, o[7] = v1, o[8] = r0, o[9] = r1;
  o[10] = m.hm[((int64_t)b * g.ncls + c) * HW + cell];
  o[11] = m.iou != nullptr ? m.iou[(int64_t)b * HW + cell] : 0.f;
  const int ys = (int)(cell / g.W), xs = (int)(cell - (int64_t)ys * g.W);
  float* bx = boxes7 + (int64_t)i * 7;
  // centerhead.py:178-195 -- (cell + offset) * out_size_factor * voxel_size + range origin; exp of the clamped size; atan2(sin, cos)
  bx[0] = ((float)xs + reg0) * g.kx + g.minx;
  bx[1] = ((float)ys + reg1) * g.ky + g.miny;
  bx[2] = hgt;
  bx[3] = expf(fminf(fmaxf(d0, -5.f), 5.f));
  bx[4] = expf(fminf(fmaxf(d1, -5.f), 5.f));
  bx[5] = expf(fminf(fmaxf(d2, -5.f), 5.f));
  bx[6] = atan2f(r0, r1);
#pragma unroll
  for (int k = 0; k < 7; k++) o[12 + k] = bx[k];
}

// axis-aligned DIoU of centerloss.py:139-176 and, optionally, its gradient with respect to the predicted box (x y z dx dy dz)
__device__ __forceinline__ float diou_aa(const float* p, const float* q, float* grad /* 6 or null */) {
  float in_[3], out_[3];
  float pmin[3], pmax[3], qmin[3], qmax[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    pmin[a] = p[a] - 0.5f * p[3 + a], pmax[a] = p[a] + 0.5f * p[3 + a];
    qmin[a] = q[a] - 0.5f * q[3 + a], qmax[a] = q[a] + 0.5f * q[3 + a];
    in_[a] = fmaxf(fminf(pmax[a], qmax[a]) - fmaxf(pmin[a], qmin[a]), 0.f);
    out_[a] = fmaxf(fmaxf(pmax[a], qmax[a]) - fminf(pmin[a], qmin[a]), 0.f);
  }
  const float vi = in_[0] * in_[1] * in_[2];
  const float pv = p[3] * p[4] * p[5], qv = q[3] * q[4] * q[5];
  const float vu = qv + pv - vi;
  float din = 0.f;
#pragma unroll
  for (int a = 0; a < 3; a++) din += (q[a] - p[a]) * (q[a] - p[a]);
  const float dout = out_[0] * out_[0] + out_[1] * out_[1] + out_[2] * out_[2];
  const float raw = vi / vu - din / dout;
  const float v = fminf(fmaxf(raw, -1.f), 1.f);
  if (grad != nullptr) {
#pragma unroll
    for (int a = 0; a < 6; a++) grad[a] = 0.f;
    if (raw > -1.f && raw < 1.f) {
      const float dvi = (vu + vi) / (vu * vu), dpv = -vi / (vu * vu);
      const float ddin = -1.f / dout, ddout = din / (dout * dout);
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const float others = (a == 0 ? in_[1] * in_[2] : (a == 1 ? in_[0] * in_[2] : in_[0] * in_[1]));
        float gmax = 0.f, gmin = 0.f;  // gradients with respect to pmax[a], pmin[a]
        if (in_[a] > 0.f) {
          const float t = dvi * others;
          if (pmax[a] < qmax[a]) gmax += t;
          if (pmin[a] > qmin[a]) gmin -= t;
        }
        if (out_[a] > 0.f) {
          const float t = ddout * 2.f * out_[a];
          if (pmax[a] > qmax[a]) gmax += t;
          if (pmin[a] < qmin[a]) gmin -= t;
        }
        grad[a] += gmax + gmin + ddin * (-2.f) * (q[a] - p[a]);
        grad[3 + a] += 0.5f * gmax - 0.5f * gmin;
      }
      grad[3] += dpv * p[4] * p[5];
      grad[4] += dpv * p[3] * p[5];
      grad[5] += dpv * p[3] * p[4];
    }
  }
  return v;
}

// losses[0] hm_loss, [1..10] per-element regression loss (before code weights), [11] iou_loss, [12] iou_reg_loss, [13] num_pos,
// [14] the factor the hm gradient is scaled with (-1/num_pos or -1)
// 256 threads: with 1024 (four waves per SIMD, 128 VGPRs each) the 14 fp64 accumulators beside the inlined DIoU spilled 1.3 KB per lane
constexpr int kLossReduceThreads = 256;
__global__ __launch_bounds__(kLossReduceThreads) void k_loss_reduce(const float* __restrict__ gathered, const uint8_t* __restrict__ mask, const float* __restrict__ anno,
                                                      const float* __restrict__ gt_boxes, const float* __restrict__ iou3d, const double* __restrict__ neg_part,
                                                      int n_neg_part, int n, int has_iou, int has_diou, float* __restrict__ losses) {
  __shared__ double s_acc[kLossReduceThreads / 64][16];
  double acc[14];
#pragma unroll
  for (int k = 0; k < 14; k++) acc[k] = 0.0;
  for (int i = threadIdx.x; i < n; i += kLossReduceThreads) {
    if (!mask[i]) continue;  // every term carries the mask (a NaN target under a zero mask contributes nothing either, :55-58)
    const float* o = gathered + (int64_t)i * kGW;
    const float p = sigmoid_clamped(o[10]);
    acc[0] += (double)(logf(p) * (1.f - p) * (1.f - p));
    acc[1] += 1.0;
#pragma unroll
    for (int c = 0; c < 10; c++) {
      const float t = anno[(int64_t)i * 10 + c];
      if (!(t != t)) acc[2 + c] += (double)fabsf(o[c] - t);  // NaN target: replaced by the prediction = no loss, no gradient
    }
    if (has_iou) acc[12] += (double)fabsf(o[11] - (2.f * iou3d[i] - 1.f));
    if (has_diou) acc[13] += (double)(1.f - diou_aa(o + 12, gt_boxes + (int64_t)i * 7, nullptr));
  }
  double neg = 0.0;
  for (int i = threadIdx.x; i < n_neg_part; i += kLossReduceThreads) neg += neg_part[i];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 14; k++) {
    double v = acc[k];
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (lane == 0) s_acc[wave][k] = v;
  }
  for (int d = 32; d >= 1; d >>= 1) neg += __shfl_xor(neg, d);
  if (lane == 0) s_acc[wave][14] = neg;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[15];
    for (int k = 0; k < 15; k++) {
      t[k] = 0.0;
      for (int w = 0; w < kLossReduceThreads / 64; w++) t[k] += s_acc[w][k];
    }
    const double npos = t[1];
    losses[0] = (float)(npos > 0.0 ? -(t[0] + t[14]) / npos : -t[14]);
    for (int c = 0; c < 10; c++) losses[1 + c] = (float)(t[2 + c] / (npos + 1e-4));
    losses[11] = (float)(npos > 0.0 ? t[12] / (npos + 1e-4) : 0.0);
    losses[12] = (float)(npos > 0.0 ? t[13] / (npos + 1e-4) : 0.0);
    losses[13] = (float)npos;
    losses[14] = (float)(npos > 0.0 ? -1.0 / npos : -1.0);
  }
}

// upstream: [0] d/d hm_loss, [1..10] d/d box_loss_elem, [11] d/d iou_loss, [12] d/d iou_reg_loss; losses: k_loss_reduce's output
__global__ __launch_bounds__(kLB) void k_loss_scatter(LossGrads gm, LossGeom g, const float* __restrict__ gathered, const int64_t* __restrict__ ind,
                                                      const int64_t* __restrict__ cat, const uint8_t* __restrict__ mask, const float* __restrict__ anno,
                                                      const float* __restrict__ gt_boxes, const float* __restrict__ iou3d, const float* __restrict__ losses,
                                                      const float* __restrict__ upstream, int has_iou, int has_diou) {
  const int i = blockIdx.x * kLB + threadIdx.x;
  if (i >= g.B * g.M || !mask[i]) return;
  const int b = i / g.M;
  const int64_t HW = (int64_t)g.H * g.W;
  int64_t cell = ind[i];
  cell = cell < 0 ? 0 : (cell >= HW ? HW - 1 : cell);
  int c = (int)cat[i];
  c = c < 0 ? 0 : (c >= g.ncls ? g.ncls - 1 : c);
  const float* o = gathered + (int64_t)i * kGW;
  const float npos = losses[13];
  const float inv = 1.0f / (npos + 1e-4f);
  // positive focal term: d/dx [log p (1-p)^2], p = clamp(sigmoid(x))
  {
    const float ps = 1.0f / (1.0f + expf(-o[10]));
    if (ps > 1e-4f && ps < 1.0f - 1e-4f) {
      const float q = 1.f - ps;
      const float dp = q * q / ps - 2.f * q * logf(ps);
      atomicAdd(&gm.hm[((int64_t)b * g.ncls + c) * HW + cell], upstream[0] * losses[14] * dp * (ps * q));
    }
  }
  float gch[10];
#pragma unroll
  for (int k = 0; k < 10; k++) {
    const float t = anno[(int64_t)i * 10 + k];
    const float d = o[k] - t;
    gch[k] = (t != t) ? 0.f : upstream[1 + k] * inv * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  }
  if (has_diou && npos > 0.f) {  // d(1 - diou) through the decoded box: x <- reg0, y <- reg1, z <- height, sizes <- exp(clamp(dim))
    float gb[6];
    diou_aa(o + 12, gt_boxes + (int64_t)i * 7, gb);
    const float s = -upstream[12] * inv;
    gch[0] += s * gb[0] * g.kx;
    gch[1] += s * gb[1] * g.ky;
    gch[2] += s * gb[2];
#pragma unroll
    for (int a = 0; a < 3; a++)
      if (o[3 + a] > -5.f && o[3 + a] < 5.f) gch[3 + a] += s * gb[3 + a] * o[15 + a];
  }
  atomicAdd(&gm.reg[((int64_t)b * 2 + 0) * HW + cell], gch[0]);
  atomicAdd(&gm.reg[((int64_t)b * 2 + 1) * HW + cell], gch[1]);
  atomicAdd(&gm.height[(int64_t)b * HW + cell], gch[2]);
  atomicAdd(&gm.dim[((int64_t)b * 3 + 0) * HW + cell], gch[3]);
  atomicAdd(&gm.dim[((int64_t)b * 3 + 1) * HW + cell], gch[4]);
  atomicAdd(&gm.dim[((int64_t)b * 3 + 2) * HW + cell], gch[5]);
  atomicAdd(&gm.vel[((int64_t)b * 2 + 0) * HW + cell], gch[6]);
  atomicAdd(&gm.vel[((int64_t)b * 2 + 1) * HW + cell], gch[7]);
  atomicAdd(&gm.rot[((int64_t)b * 2 + 0) * HW + cell], gch[8]);
  atomicAdd(&gm.rot[((int64_t)b * 2 + 1) * HW + cell], gch[9]);
  if (has_iou && gm.iou != nullptr && npos > 0.f) {
    const float d = o[11] - (2.f * iou3d[i] - 1.f);
    atomicAdd(&gm.iou[(int64_t)b * HW + cell], upstream[11] * inv * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
  }
}

constexpr int kNegBlocks = 1024;

}  // namespace

extern "C" {

size_t pnx_center_loss_workspace_bytes(int32_t batch, int32_t max_objs) {
  if (batch <= 0 || max_objs <= 0) return 0;
  const size_t n = (size_t)batch * max_objs;
  return pnx_align_up(n * kGW * 4, 256) + pnx_align_up(n * 7 * 4, 256) + pnx_align_up(n * 4, 256) + pnx_align_up(kNegBlocks * 8, 256);
}

// maps7 / grads7: HOST arrays of 7 device pointers in the order hm, reg, height, dim, rot, vel, iou (iou may be NULL)
int pnx_center_loss_forward(const void* const* maps7, const float* hm_target, const int64_t* ind, const uint8_t* mask, const int64_t* cat,
                            const float* anno_box, const float* gt_boxes, int32_t batch, int32_t n_classes, int32_t h, int32_t w, int32_t max_objs,
                            const float* geom4_host, int32_t with_reg_iou, float* losses15, void* workspace, size_t workspace_bytes,
                            pnx_stream_t stream) {
  PNX_REQUIRE(maps7 && hm_target && ind && mask && cat && anno_box && gt_boxes && geom4_host && losses15 && workspace, PNX_ERR_INVALID, "null pointer");
  for (int k = 0; k < 6; k++) PNX_REQUIRE(maps7[k] != nullptr, PNX_ERR_INVALID, "head map %d is NULL", k);
  PNX_REQUIRE(batch > 0 && n_classes > 0 && h > 0 && w > 0 && max_objs > 0, PNX_ERR_INVALID, "bad sizes");
  PNX_REQUIRE(workspace_bytes >= pnx_center_loss_workspace_bytes(batch, max_objs), PNX_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  PnxCarver c(workspace);
  const int n = batch * max_objs;
  float* gathered = c.take<float>((size_t)n * kGW);
  float* boxes7 = c.take<float>((size_t)n * 7);
  float* iou3d = c.take<float>((size_t)n);
  double* part = c.take<double>(kNegBlocks);
  LossMaps m = {(const float*)maps7[0], (const float*)maps7[1], (const float*)maps7[2], (const float*)maps7[3], (const float*)maps7[4],
                (const float*)maps7[5], (const float*)maps7[6]};
  LossGeom g = {batch, n_classes, h, w, max_objs, geom4_host[0], geom4_host[1], geom4_host[2], geom4_host[3]};
  const int64_t ncell = (int64_t)batch * n_classes * h * w;
  k_focal_neg<<<kNegBlocks, kLB, 0, st>>>(m.hm, hm_target, ncell, part);
  k_loss_gather<<<(n + kLB - 1) / kLB, kLB, 0, st>>>(m, g, ind, cat, gathered, boxes7);
  PNX_LAUNCH_CHECK();
  const int has_iou = m.iou != nullptr;
  if (has_iou) {
    const int rc = pnx_boxes_aligned_iou3d(boxes7, gt_boxes, n, iou3d, stream);
    if (rc != PNX_OK) return rc;
  }
  k_loss_reduce<<<1, kLossReduceThreads, 0, st>>>(gathered, mask, anno_box, gt_boxes, iou3d, part, kNegBlocks, n, has_iou, with_reg_iou != 0, losses15);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

// grads7: gradient buffers of the seven maps; hm is written densely, the other six must be ZERO on entry (the listed cells are
// accumulated with atomics); upstream13 = device array of the gradients of the 13 loss outputs; workspace as left by the forward call
int pnx_center_loss_backward(const void* const* maps7, void* const* grads7, const float* hm_target, const int64_t* ind, const uint8_t* mask,
                             const int64_t* cat, const float* anno_box, const float* gt_boxes, int32_t batch, int32_t n_classes, int32_t h, int32_t w,
                             int32_t max_objs, const float* geom4_host, int32_t with_reg_iou, const float* losses15, const float* upstream13,
                             float* coef_scratch, void* workspace, size_t workspace_bytes, pnx_stream_t stream);

}  // extern "C"

namespace {
__global__ void k_hm_coef(const float* __restrict__ losses, const float* __restrict__ upstream, float* __restrict__ coef) {
  coef[0] = upstream[0] * losses[14];  // d hm_loss / d neg_sum = -1/num_pos (or -1)
}
}  // namespace

extern "C" int pnx_center_loss_backward(const void* const* maps7, void* const* grads7, const float* hm_target, const int64_t* ind, const uint8_t* mask,
                                        const int64_t* cat, const float* anno_box, const float* gt_boxes, int32_t batch, int32_t n_classes, int32_t h,
                                        int32_t w, int32_t max_objs, const float* geom4_host, int32_t with_reg_iou, const float* losses15,
                                        const float* upstream13, float* coef_scratch, void* workspace, size_t workspace_bytes, pnx_stream_t stream) {
  PNX_REQUIRE(maps7 && grads7 && hm_target && ind && mask && cat && anno_box && gt_boxes && geom4_host && losses15 && upstream13 && coef_scratch && workspace,
              PNX_ERR_INVALID, "null pointer");
  for (int k = 0; k < 6; k++) PNX_REQUIRE(maps7[k] != nullptr && grads7[k] != nullptr, PNX_ERR_INVALID, "head map / gradient %d is NULL", k);
  PNX_REQUIRE(workspace_bytes >= pnx_center_loss_workspace_bytes(batch, max_objs), PNX_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  PnxCarver c(workspace);
  const int n = batch * max_objs;
  const float* gathered = c.take<float>((size_t)n * kGW);
  (void)c.take<float>((size_t)n * 7);
  const float* iou3d = c.take<float>((size_t)n);
  LossGrads gm = {(float*)grads7[0], (float*)grads7[1], (float*)grads7[2], (float*)grads7[3], (float*)grads7[4], (float*)grads7[5], (float*)grads7[6]};
  LossGeom g = {batch, n_classes, h, w, max_objs, geom4_host[0], geom4_host[1], geom4_host[2], geom4_host[3]};
  const int64_t ncell = (int64_t)batch * n_classes * h * w;
  k_hm_coef<<<1, 1, 0, st>>>(losses15, upstream13, coef_scratch);
  k_focal_neg_bwd<<<kNegBlocks, kLB, 0, st>>>((const float*)maps7[0], hm_target, ncell, coef_scratch, gm.hm);
  k_loss_scatter<<<(n + kLB - 1) / kLB, kLB, 0, st>>>(gm, g, gathered, ind, cat, mask, anno_box, gt_boxes, iou3d, losses15, upstream13,
                                                      maps7[6] != nullptr, with_reg_iou != 0);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}
