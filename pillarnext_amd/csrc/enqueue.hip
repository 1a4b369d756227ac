// enqueue.hip -- one-call enqueue of prebuilt launch tables (include/pnx.h: pnx_enqueue, pnx_decode_lazy_enqueue).
//
// The reference's step is a Python call tree (det3d/models/detectors/single_stage.py:22-33: reader -> backbone -> neck -> head, then
// CenterHead.predict, centerhead.py:231-384) that issues a few hundred launches one by one.  Here every launch of the backbone, of the
// head and of the decoder is a C-ABI call of this library with arguments that do not change from frame batch to frame batch (persistent
// workspaces, weights, shapes), so the host side freezes them ONCE into a table and hands the table over per step: the launch thread
// spends its time in hipLaunchKernel, not in an interpreter, and a step costs the host ~15 calls instead of ~150.
// Nothing here computes: every entry forwards to the entry point it names (same checks, same error channel).
#include "pnx_common.h"

// conv3x3.hip: pnx_conv_tile_list without the memset of *tile_count (pnx_enqueue clears the counters of a whole table in one launch)
extern "C" int pnx_conv_tile_list_prezeroed(const uint8_t* mask, const uint8_t* const* row_dirty, int32_t n_dirty, int32_t batch, int32_t h, int32_t w,
                                            int32_t tile_rows, int32_t* tile_list, int32_t* tile_count, pnx_stream_t stream);

namespace {

__global__ __launch_bounds__(256) void k_lazy_cells(const int64_t* __restrict__ order, const int32_t* __restrict__ seg_len, const int64_t* __restrict__ list_key_off,
                                                    int32_t pre_max, int64_t n, int64_t* __restrict__ local) {
  // candidate slot (s, j) -> cell index inside the list's task map (b*H*W + cell): position of the key in the concatenated key array minus
  // the key offset of the list's task; slots behind seg_len[s] hold 0 (pnx_sephead_lazy_bf16 skips them)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t s = i / pre_max;
  const int32_t j = (int32_t)(i - s * pre_max);
  local[i] = j < seg_len[s] ? order[i] - list_key_off[s] : 0;
}

// up to 8 ranges of 32-bit words zeroed by one launch: the decoder's four memsets (order, boxes7, flag, keep_count) and the backbone plan's tile counters
// used to be one __amd_rocclr_fillBufferAligned dispatch each -- nine per step (profiles/r05_bench_steady_trace.md)
struct ZeroRanges {
  uint32_t* p[8];
  int64_t end[8];  // running sum of the ranges' word counts
  int n;
};
__global__ __launch_bounds__(256) void k_zero_ranges(ZeroRanges z) {
  const int64_t total = z.end[z.n - 1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int r = 0;
    while (i >= z.end[r]) r++;
    z.p[r][i - (r ? z.end[r - 1] : 0)] = 0u;
  }
}
int launch_zero_ranges(ZeroRanges& z, hipStream_t st) {
  if (z.n == 0) return PNX_OK;
  const int64_t total = z.end[z.n - 1];
  int nb = (int)((total + 256 * 8 - 1) / (256 * 8));
  nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
  k_zero_ranges<<<nb, 256, 0, st>>>(z);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}
void add_range(ZeroRanges& z, void* p, int64_t words) {
  z.p[z.n] = (uint32_t*)p;
  z.end[z.n] = (z.n ? z.end[z.n - 1] : 0) + words;
  z.n++;
}

int run_op(const pnx_op& o, hipStream_t st) {
  const int32_t* i = o.i;
  const void* const* p = o.p;
  switch (o.kind) {
    case PNX_OP_MASK_POOL3:
      return pnx_mask_pool3((const uint8_t*)p[0], i[0], i[1], i[2], i[3], (uint8_t*)p[1], st);
    case PNX_OP_TILE_LIST:
      PNX_REQUIRE(i[0] >= 0 && i[0] <= 8, PNX_ERR_INVALID, "tile list over %d row_dirty arrays (at most 8)", i[0]);
      return pnx_conv_tile_list_prezeroed((const uint8_t*)p[0], (const uint8_t* const*)&p[3], i[0], i[1], i[2], i[3], i[4], (int32_t*)p[1], (int32_t*)p[2], st);
    case PNX_OP_CONV3X3:
      return (i[7] == PNX_F16 ? pnx_conv3x3_f16 : pnx_conv3x3_bf16)(p[0], p[1], (const float*)p[2], p[3], (const uint8_t*)p[4], (void*)p[5], i[0], i[1], i[2],
                                                                    i[3], i[4], i[5], i[6], (uint8_t*)p[6], (const int32_t*)p[7], (const int32_t*)p[8], st);
    case PNX_OP_DECONV2X2:
      return (i[6] == PNX_F16 ? pnx_deconv2x2_f16 : pnx_deconv2x2_bf16)(p[0], p[1], (const float*)p[2], (void*)p[3], i[0], i[1], i[2], i[3], i[4], i[5], st);
    case PNX_OP_SEPHEAD_OUT:
      return (i[4] == PNX_F16 ? pnx_sephead_out_f16 : pnx_sephead_out_bf16)(p[0], p[1], (const float*)p[2], (void*)p[3], i[0], i[1], i[2], i[3], st);
    default:
      pnx_set_error("unknown op kind %d", o.kind);
      return PNX_ERR_INVALID;
  }
}

}  // namespace

extern "C" int pnx_enqueue(const pnx_op* ops, int32_t n_ops, pnx_stream_t stream) {
  PNX_REQUIRE(ops != nullptr && n_ops >= 0, PNX_ERR_INVALID, "pnx_enqueue: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  {  // the tile counters of the table's tile-list entries: one launch in front of the table instead of one memset per entry
    ZeroRanges z;
    z.n = 0;
    for (int32_t k = 0; k < n_ops; k++) {
      if (ops[k].kind != PNX_OP_TILE_LIST) continue;
      if (z.n == 8) {
        const int rc = launch_zero_ranges(z, st);
        if (rc != PNX_OK) return rc;
        z.n = 0;
      }
      add_range(z, const_cast<void*>(ops[k].p[2]), 1);
    }
    const int rc = launch_zero_ranges(z, st);
    if (rc != PNX_OK) return rc;
  }
  for (int32_t k = 0; k < n_ops; k++) {
    const int rc = run_op(ops[k], st);
    if (rc != PNX_OK) {
      char msg[400];
      snprintf(msg, sizeof msg, "%s", pnx_last_error());
      pnx_set_error("pnx_enqueue: entry %d (kind %d): %s", k, ops[k].kind, msg);
      return rc;
    }
  }
  return PNX_OK;
}

extern "C" int pnx_decode_lazy_enqueue(const pnx_lazy_decode* d, pnx_stream_t stream) {
  PNX_REQUIRE(d != nullptr, PNX_ERR_INVALID, "pnx_decode_lazy_enqueue: null descriptor");
  PNX_REQUIRE(d->n_tasks > 0 && d->n_tasks <= 8 && d->n_classes_total > 0 && d->batch > 0 && d->pre_max > 0 && d->post_max > 0, PNX_ERR_INVALID,
              "pnx_decode_lazy_enqueue: bad sizes");
  PNX_REQUIRE(d->dense_host && d->task_descs_host && d->task_descs_dev && d->task_key_off_host && d->task_key_off_dev && d->list_key_off_dev &&
                  d->lazy_tasks_host && d->class_task_host && d->seg_off_dev && d->nms_thresh_dev,
              PNX_ERR_INVALID, "pnx_decode_lazy_enqueue: null table");
  PNX_REQUIRE(d->keys && d->sorted_keys && d->order && d->seg_start && d->seg_len && d->seg_total && d->local && d->cand && d->boxes9 && d->boxes7 &&
                  d->scores && d->flag && d->keep && d->keep_count && d->topk_ws && d->nms_ws && d->out,
              PNX_ERR_INVALID, "pnx_decode_lazy_enqueue: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const int32_t S = d->batch * d->n_classes_total;
  const int64_t n_rows = (int64_t)S * d->pre_max;
  const int64_t n_keys = d->task_key_off_host[d->n_tasks];
  const size_t desc_bytes = pnx_decode_task_desc_bytes();
  int rc;
  // keys of every task's dense [iou] hm map (centerhead.py:283-300: sigmoid, per-class scores, score threshold)
  for (int32_t t = 0; t < d->n_tasks; t++) {
    rc = pnx_decode_keys(d->dense_host[t], d->dtype, d->batch, d->n_classes_total, (const char*)d->task_descs_host + (size_t)t * desc_bytes,
                         d->keys + d->task_key_off_host[t], st);
    if (rc != PNX_OK) return rc;
  }
  {  // what the chain below expects zeroed, in one launch: candidate order, NMS boxes, fallback flag, keep counts
    ZeroRanges z;
    z.n = 0;
    add_range(z, d->order, n_rows * 2);
    add_range(z, d->boxes7, n_rows * 7);
    add_range(z, d->flag, 1);
    add_range(z, d->keep_count, S);
    rc = launch_zero_ranges(z, st);
    if (rc != PNX_OK) return rc;
  }
  // the first pre_max of every (sample, class) list, in score order (centerhead.py:341-363)
  rc = pnx_decode_topk(d->keys, n_keys, S, d->pre_max, d->sorted_keys, d->order, d->seg_start, d->seg_len, d->seg_total, d->topk_ws, d->topk_ws_bytes, st);
  if (rc != PNX_OK) return rc;
  k_lazy_cells<<<(unsigned)((n_rows + 255) / 256), 256, 0, st>>>(d->order, d->seg_len, d->list_key_off_dev, d->pre_max, n_rows, d->local);
  PNX_LAUNCH_CHECK();
  // the regression branches at those cells only
  rc = (d->dtype == PNX_F16 ? pnx_sephead_lazy_f16 : pnx_sephead_lazy_bf16)(d->lazy_tasks_host, d->n_tasks, d->class_task_host, d->n_classes_total, d->batch, d->local, d->seg_len, d->pre_max, d->cand, st);
  if (rc != PNX_OK) return rc;
  rc = pnx_decode_boxes_lazy(d->task_descs_dev, d->task_key_off_dev, d->n_tasks, d->n_classes_total, d->sorted_keys, d->order, d->seg_start, d->seg_len,
                             d->seg_total, S, d->pre_max, d->cand, d->boxes9, d->boxes7, d->scores, d->flag, st);
  if (rc != PNX_OK) return rc;
  rc = pnx_nms_rotated_batched(d->boxes7, d->seg_off_dev, d->seg_len, S, d->pre_max, d->nms_thresh_dev, d->post_max, d->keep, d->keep_count, d->nms_ws,
                               d->nms_ws_bytes, st);
  if (rc != PNX_OK) return rc;
  rc = pnx_gather_kept(d->boxes9, d->scores, d->keep, d->keep_count, S, d->pre_max, d->post_max, d->out, st);
  if (rc != PNX_OK) return rc;
  // the one device -> host hand-off of the frame batch (pinned memory, asynchronous: the caller records an event behind this call)
  if (d->out_host) PNX_CHECK_HIP(hipMemcpyAsync(d->out_host, d->out, (size_t)S * d->post_max * 10 * sizeof(float), hipMemcpyDeviceToHost, st));
  if (d->keep_count_host) PNX_CHECK_HIP(hipMemcpyAsync(d->keep_count_host, d->keep_count, (size_t)S * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  if (d->flag_host) PNX_CHECK_HIP(hipMemcpyAsync(d->flag_host, d->flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  return PNX_OK;
}

// layout checks for host bindings (ctypes mirrors of the two structures)
extern "C" size_t pnx_op_bytes(void) { return sizeof(pnx_op); }
extern "C" size_t pnx_lazy_decode_bytes(void) { return sizeof(pnx_lazy_decode); }
