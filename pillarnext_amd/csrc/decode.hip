// decode.hip -- CenterHead.predict / post_processing up to the NMS input, fused (det3d/models/heads/centerhead.py:231-363).
//
// The reference runs ~25 elementwise kernels per task, then a host-driven loop over samples and classes with
// boolean-mask compactions (one device->host sync each).  Here:
//   k_decode_keys   one pass over the packed NHWC head output of a task: sigmoid / class max (:259,:341), centre decode
//                   (:285-297), score > thr and centre-in-range mask (:342-346), IoU-rectified score (:352-354)
//                   -> ONE 64-bit sort key per cell: (segment << 32) | ~score_bits, segment = sample * n_classes + class;
//                   masked-out cells get the all-ones key.  A single stable device sort of the keys then orders every
//                   (sample, class) candidate list by descending score at once (box_torch_ops.py:13).
//   k_decode_boxes  decodes full 9-d boxes (:259-303) only for the first pre_max candidates of each segment.
//   k_gather_kept   after the batched NMS: the first post_max kept boxes of every segment into a dense block.
// Channel order of the packed tensor = SepHead's dict order: reg(2) height(1) dim(3) rot(2) vel(2) [iou(1)] hm(ncls).
#include <string.h>

#include "pnx_common.h"

namespace {

struct DecodeTask {
  int C;         // channel stride of the packed NHWC tensor (padded)
  int has_iou;   // 1 if the iou branch exists
  int ncls;      // classes of this task (<= 4)
  int cls_off;   // global class offset of the task
  int H, W;
  float osf, vx, vy, pcx, pcy;  // applied in the reference's order: ((xs * osf) * vs) + pc
  float score_thr;
  float lim[6];  // post_center_limit_range
  int use_lim;
  float rect[4];  // rectifier per class
};

template <typename T>
__device__ __forceinline__ float ldf(const T* p, int i);
template <>
__device__ __forceinline__ float ldf<float>(const float* p, int i) {
  return p[i];
}
template <>
__device__ __forceinline__ float ldf<uint16_t>(const uint16_t* p, int i) {
  return __uint_as_float((uint32_t)p[i] << 16);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <typename T>
__global__ __launch_bounds__(256) void k_decode_keys(const T* __restrict__ x, DecodeTask tk, int B, int n_classes_total,
                                                     unsigned long long* __restrict__ keys) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int HW = tk.H * tk.W;
  if (idx >= (int64_t)B * HW) return;
  const int cell = (int)(idx % HW), b = (int)(idx / HW);
  const T* p = x + idx * tk.C;
  const int o_hm = 10 + tk.has_iou;
  float best = -1.f;
  int lab = 0;
  for (int c = 0; c < tk.ncls; c++) {
    const float s = sigmoidf_(ldf<T>(p, o_hm + c));
    if (s > best) {  // torch.max returns the first maximal index
      best = s;
      lab = c;
    }
  }
  bool ok = best > tk.score_thr;
  const float xs = ((float)(cell % tk.W) + ldf<T>(p, 0)) * tk.osf * tk.vx + tk.pcx;
  const float ys = ((float)(cell / tk.W) + ldf<T>(p, 1)) * tk.osf * tk.vy + tk.pcy;
  const float z = ldf<T>(p, 2);
  if (tk.use_lim)
    ok = ok && xs >= tk.lim[0] && ys >= tk.lim[1] && z >= tk.lim[2] && xs <= tk.lim[3] && ys <= tk.lim[4] && z <= tk.lim[5];
  unsigned long long key = ~0ULL;
  if (ok) {
    float iou = 1.f;
    if (tk.has_iou) iou = fminf(fmaxf((ldf<T>(p, 10) + 1.f) * 0.5f, 0.f), 1.f);
    const float a = tk.rect[lab];
    const float sc = powf(best, 1.f - a) * powf(iou, a);
    const unsigned seg = (unsigned)(b * n_classes_total + tk.cls_off + lab);
    key = ((unsigned long long)seg << 32) | (unsigned long long)(0xFFFFFFFFu - __float_as_uint(sc));
  }
  keys[idx] = key;
}

// One thread per (segment, rank < pre_max): decode the candidate found at sorted position seg_start[s] + j.
template <typename T>
__global__ __launch_bounds__(256) void k_decode_boxes(const T* const* __restrict__ task_x, const DecodeTask* __restrict__ tasks,
                                                      const int64_t* __restrict__ task_key_off, int n_tasks, int B,
                                                      const unsigned long long* __restrict__ sorted_keys,
                                                      const int64_t* __restrict__ order, const int64_t* __restrict__ seg_start,
                                                      const int32_t* __restrict__ seg_len, int S, int pre_max,
                                                      float* __restrict__ boxes9, float* __restrict__ boxes7, float* __restrict__ scores) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)S * pre_max) return;
  const int s = (int)(idx / pre_max), j = (int)(idx % pre_max);
  if (j >= seg_len[s]) return;
  const int64_t pos = seg_start[s] + j;
  const int64_t src = order[pos];
  int t = 0;
  while (t + 1 < n_tasks && src >= task_key_off[t + 1]) t++;
  const DecodeTask tk = tasks[t];
  const int64_t local = src - task_key_off[t];  // = b * HW + cell
  const int HW = tk.H * tk.W;
  const int cell = (int)(local % HW);
  const T* p = task_x[t] + local * tk.C;
  float o[9];
  o[0] = ((float)(cell % tk.W) + ldf<T>(p, 0)) * tk.osf * tk.vx + tk.pcx;
  o[1] = ((float)(cell / tk.W) + ldf<T>(p, 1)) * tk.osf * tk.vy + tk.pcy;
  o[2] = ldf<T>(p, 2);
  o[3] = expf(ldf<T>(p, 3));
  o[4] = expf(ldf<T>(p, 4));
  o[5] = expf(ldf<T>(p, 5));
  o[6] = ldf<T>(p, 8);
  o[7] = ldf<T>(p, 9);
  o[8] = atan2f(ldf<T>(p, 6), ldf<T>(p, 7));
  float* b9 = boxes9 + idx * 9;
#pragma unroll
  for (int k = 0; k < 9; k++) b9[k] = o[k];
  float* b7 = boxes7 + idx * 7;
#pragma unroll
  for (int k = 0; k < 6; k++) b7[k] = o[k];
  b7[6] = o[8];
  scores[idx] = __uint_as_float(0xFFFFFFFFu - (unsigned)(sorted_keys[pos] & 0xFFFFFFFFull));
}

__global__ __launch_bounds__(256) void k_gather_kept(const float* __restrict__ boxes9, const float* __restrict__ scores,
                                                     const int32_t* __restrict__ keep, const int32_t* __restrict__ keep_count, int S,
                                                     int pre_max, int post_max, float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= S * post_max) return;
  const int s = idx / post_max, j = idx % post_max;
  float* o = out + (int64_t)idx * 10;
  if (j >= keep_count[s]) {
#pragma unroll
    for (int k = 0; k < 10; k++) o[k] = 0.f;
    return;
  }
  const int64_t src = (int64_t)s * pre_max + keep[(int64_t)s * pre_max + j];
#pragma unroll
  for (int k = 0; k < 9; k++) o[k] = boxes9[src * 9 + k];
  o[9] = scores[src];
}

}  // namespace

extern "C" {

// task_desc_host: 32 floats/ints per task, see pillarnext_amd/decode.py::pack_task (copied into a DecodeTask)
int pnx_decode_keys(const void* packed, int32_t dtype, int32_t batch, int32_t n_classes_total, const void* task_desc_host,
                    uint64_t* keys, pnx_stream_t stream) {
  PNX_REQUIRE(packed && task_desc_host && keys && batch > 0, PNX_ERR_INVALID, "bad arguments");
  DecodeTask tk;
  memcpy(&tk, task_desc_host, sizeof(DecodeTask));
  PNX_REQUIRE(tk.ncls >= 1 && tk.ncls <= 4 && tk.C >= 10 + tk.has_iou + tk.ncls, PNX_ERR_INVALID, "bad task descriptor");
  const int64_t n = (int64_t)batch * tk.H * tk.W;
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PNX_F32) k_decode_keys<float><<<nb, 256, 0, st>>>((const float*)packed, tk, batch, n_classes_total, (unsigned long long*)keys);
  else if (dtype == PNX_BF16) k_decode_keys<uint16_t><<<nb, 256, 0, st>>>((const uint16_t*)packed, tk, batch, n_classes_total, (unsigned long long*)keys);
  else PNX_REQUIRE(false, PNX_ERR_UNSUPPORTED, "decode is built for fp32 and bf16 head outputs");
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

size_t pnx_decode_task_desc_bytes(void) { return sizeof(DecodeTask); }

int pnx_decode_boxes(const void* const* task_ptrs_dev, const void* task_descs_dev, const int64_t* task_key_off_dev, int32_t n_tasks,
                     int32_t dtype, int32_t batch, const uint64_t* sorted_keys, const int64_t* order, const int64_t* seg_start,
                     const int32_t* seg_len, int32_t num_segments, int32_t pre_max, float* boxes9, float* boxes7, float* scores,
                     pnx_stream_t stream) {
  PNX_REQUIRE(task_ptrs_dev && task_descs_dev && task_key_off_dev && sorted_keys && order && seg_start && seg_len && boxes9 && boxes7 && scores,
              PNX_ERR_INVALID, "null pointer");
  PNX_REQUIRE(n_tasks > 0 && num_segments > 0 && pre_max > 0, PNX_ERR_INVALID, "bad sizes");
  const int64_t n = (int64_t)num_segments * pre_max;
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PNX_F32)
    k_decode_boxes<float><<<nb, 256, 0, st>>>((const float* const*)task_ptrs_dev, (const DecodeTask*)task_descs_dev, task_key_off_dev, n_tasks, batch,
                                              (const unsigned long long*)sorted_keys, order, seg_start, seg_len, num_segments, pre_max, boxes9, boxes7, scores);
  else if (dtype == PNX_BF16)
    k_decode_boxes<uint16_t><<<nb, 256, 0, st>>>((const uint16_t* const*)task_ptrs_dev, (const DecodeTask*)task_descs_dev, task_key_off_dev, n_tasks,
                                                 batch, (const unsigned long long*)sorted_keys, order, seg_start, seg_len, num_segments, pre_max, boxes9,
                                                 boxes7, scores);
  else PNX_REQUIRE(false, PNX_ERR_UNSUPPORTED, "decode is built for fp32 and bf16 head outputs");
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

int pnx_gather_kept(const float* boxes9, const float* scores, const int32_t* keep, const int32_t* keep_count, int32_t num_segments,
                    int32_t pre_max, int32_t post_max, float* out, pnx_stream_t stream) {
  PNX_REQUIRE(boxes9 && scores && keep && keep_count && out && num_segments > 0 && pre_max > 0 && post_max > 0, PNX_ERR_INVALID, "bad arguments");
  const int n = num_segments * post_max;
  k_gather_kept<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(boxes9, scores, keep, keep_count, num_segments, pre_max, post_max, out);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // extern "C"
