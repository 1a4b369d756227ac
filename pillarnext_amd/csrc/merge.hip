// merge.hip -- multi-sweep merge on the device: what the reference's dataset classes do on the CPU for every
// sample before collation --
//   det3d/datasets/nuscenes/nusc.py:76-121  read_sweep / remove_close / load_pointcloud: every past sweep is moved into the key
//     frame with a 4x4 transform (fp64 matmul, result stored back as fp32), points with |x| < r AND |y| < r are removed (r = 1.0,
//     past sweeps only), a time-lag column is appended, all sweeps are concatenated;
//   det3d/datasets/waymo/waymo.py:49-67     the same with inv(pose) @ sweep_pose and a timestamp column, no close-point filter;
//   det3d/datasets/loader/collate.py:15-22  the sample index is prepended as a float column --
// as one stable compaction over the raw sweeps already resident in HBM, writing the collated (N, 2+C) buffer the reader consumes.
// Segment s = rows [seg_offsets[s], seg_offsets[s+1]) of `raw`; per segment: optional 3x4 transform (fp64, row-major), close-point
// radius (0 = keep all), time value, batch index.  Row order is preserved (sweep order, then the file's order), like np.concatenate.
// Rows [n_out, n_raw) of `out` are filled with batch index -1 (dropped by the reader): `out` is usable as a whole without reading n_out.
#include "pnx_common.h"
#include "pnx_scan.h"

namespace {

struct SegDesc {
  double t[12];   // rows of the 3x4 transform [R | t]
  int64_t begin, end;
  float radius, time;
  int32_t batch, has_transform;
};

__device__ __forceinline__ int find_seg(const SegDesc* __restrict__ segs, int nseg, int64_t i) {
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (i >= segs[mid].end) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// x' = t0*x + t1*y + t2*z + t3 in fp64, rounded to fp32 (numpy: float64 dot, assigned into a float32 array)
__device__ __forceinline__ void xform(const SegDesc& s, const float* __restrict__ p, float* o) {
  if (!s.has_transform) {
    o[0] = p[0], o[1] = p[1], o[2] = p[2];
    return;
  }
  const double x = p[0], y = p[1], z = p[2];
#pragma unroll
  for (int r = 0; r < 3; r++) o[r] = (float)(((s.t[4 * r] * x + s.t[4 * r + 1] * y) + s.t[4 * r + 2] * z) + s.t[4 * r + 3]);
}

__global__ __launch_bounds__(kBlock) void k_merge_flags(const float* __restrict__ raw, int stride, int64_t n, const SegDesc* __restrict__ segs, int nseg,
                                                        uint32_t* __restrict__ keep) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const SegDesc& s = segs[find_seg(segs, nseg, i)];
  float o[3];
  xform(s, raw + i * stride, o);
  const bool close = s.radius > 0.f && fabsf(o[0]) < s.radius && fabsf(o[1]) < s.radius;  // nusc.py:91-99
  keep[i] = close ? 0u : 1u;
}

__global__ __launch_bounds__(kBlock) void k_merge_write(const float* __restrict__ raw, int stride, int64_t n, const SegDesc* __restrict__ segs, int nseg,
                                                        const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pre, const uint32_t* __restrict__ blk,
                                                        int ncopy, float* __restrict__ out, int out_stride, const int32_t* __restrict__ n_out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t before = (int64_t)blk[i >> PNX_SCAN_SHIFT] + pre[i];  // kept rows in front of row i
  if (!keep[i]) {
    // a dropped row becomes one of the rows [n_out, n) of `out`: batch index -1, which the reader masks out (pillar_encoder.py:98-104
    // keeps only rows inside the grid; reader.hip drops b outside [0,B)) -- the consumer can take N = n from the shape, no host sync
    float* o = out + ((int64_t)n_out[0] + (i - before)) * out_stride;
    o[0] = -1.0f;
    for (int k = 1; k < out_stride; k++) o[k] = 0.f;
    return;
  }
  const SegDesc& s = segs[find_seg(segs, nseg, i)];
  const float* p = raw + i * stride;
  float* o = out + before * out_stride;
  float x[3];
  xform(s, p, x);
  o[0] = (float)s.batch;
  o[1] = x[0], o[2] = x[1], o[3] = x[2];
  for (int k = 3; k < ncopy; k++) o[1 + k] = p[k];  // intensity (and whatever other per-point columns are kept)
  o[1 + ncopy] = s.time;
}

}  // namespace

extern "C" {

size_t pnx_merge_sweeps_desc_bytes(void) { return sizeof(SegDesc); }

size_t pnx_merge_sweeps_workspace_bytes(int64_t n_raw) {
  if (n_raw < 0) return 0;
  const int64_t nblk = (n_raw + PNX_SCAN_ITEMS - 1) / PNX_SCAN_ITEMS + 8;
  return pnx_align_up((size_t)(n_raw + 8) * 4, 256) * 2 + pnx_align_up((size_t)nblk * 4, 256);
}

int pnx_merge_sweeps(const float* raw, int64_t n_raw, int32_t raw_stride, int32_t n_copy, const void* seg_descs_dev, int32_t n_segments, float* out,
                     int32_t* n_out_dev, void* workspace, size_t workspace_bytes, pnx_stream_t stream) {
  PNX_REQUIRE(n_raw >= 0 && n_segments >= 1 && raw_stride >= 3 && n_copy >= 3 && n_copy <= raw_stride, PNX_ERR_INVALID, "bad sizes");
  PNX_REQUIRE(n_raw == 0 || (raw && seg_descs_dev && out), PNX_ERR_INVALID, "null pointer");
  PNX_REQUIRE(n_out_dev != nullptr, PNX_ERR_INVALID, "n_out is NULL");
  PNX_REQUIRE(n_raw < ((int64_t)1 << 31) - 64, PNX_ERR_UNSUPPORTED, "more than 2^31 rows");
  PNX_REQUIRE(workspace && workspace_bytes >= pnx_merge_sweeps_workspace_bytes(n_raw), PNX_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  PnxCarver c(workspace);
  uint32_t* keep = c.take<uint32_t>(n_raw + 8);
  uint32_t* pre = c.take<uint32_t>(n_raw + 8);
  const int nblk = (int)((n_raw + PNX_SCAN_ITEMS - 1) / PNX_SCAN_ITEMS);
  uint32_t* blk = c.take<uint32_t>(nblk + 8);
  if (n_raw == 0) {
    PNX_CHECK_HIP(hipMemsetAsync(n_out_dev, 0, sizeof(int32_t), st));
    return PNX_OK;
  }
  const SegDesc* segs = reinterpret_cast<const SegDesc*>(seg_descs_dev);
  const int nb = (int)((n_raw + kBlock - 1) / kBlock);
  k_merge_flags<<<nb, kBlock, 0, st>>>(raw, raw_stride, n_raw, segs, n_segments, keep);
  k_scan_local<SCAN_IDENT><<<nblk, kBlock, 0, st>>>(keep, n_raw, pre, blk);
  k_scan_blocks<<<1, kBlock, 0, st>>>(blk, nblk, n_out_dev);
  k_merge_write<<<nb, kBlock, 0, st>>>(raw, raw_stride, n_raw, segs, n_segments, keep, pre, blk, n_copy, out, n_copy + 2, n_out_dev);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // extern "C"
