// group.hip -- the voxel and multi-view readers on the device, gfx950.
//
//   pnx_group_points     point -> cell grouping of det3d/models/readers/voxel_encoder.py:25-72 (VoxelNet: 3-D cells, rows outside the range
//                        dropped) and det3d/models/readers/mvf_encoder.py:39-86 / :88-141 (PillarVoxelNet / CylinderNet: 2-D cells, cell index
//                        CLAMPED to the grid, cylinder coordinates phi [deg], z, rho), with the per-cell mean (torch_scatter.scatter_mean,
//                        voxel_encoder.py:20, mvf_encoder.py:71,125) and the point decoration (mvf_encoder.py:73-83, 127-138)
//   pnx_pfn_layer_eval   PFNLayer in eval mode with arbitrary widths (pillar_encoder.py:35-50 as SingleView uses it, mvf_encoder.py:150-163,
//                        187-188): Linear + folded BatchNorm + ReLU + per-cell max, the concat [x, max[inv]] read in place by the next layer
//   pnx_bilinear_gather  SingleView.bilinear_interpolate (mvf_encoder.py:208-246) on a channels-last map
//
// Grouping without sorting a point, like the pillar reader's rank outputs: torch.unique(dim=0) over [b, c0, c1(, c2)] rows orders the cells
// lexicographically = by the linear key ((b*g0 + c0)*g1 + c1)(*g2 + c2); every occupied key sets one bit of a key-order bitmap, a popcount
// prefix over the bitmap words is the cell's rank (= its row in `unq`), a point's unq_inv is one lookup.  Sums for the means are fp64
// atomics: exact for these magnitudes, so the result does not depend on the order in which the points arrive (deterministic), and the
// mean is fp32(sum) / fp32(count) as in the pillar reader.
// HBM-bound glue around two hash-free passes over the points; nothing here is shaped into a GEMM.
#include "pnx_common.h"
#define PNX_HD __device__ __forceinline__
#include "pnx_detmath.h"
#include "pnx_scan.h"

namespace {

struct GroupGeomDev {
  float mn[3], vs[3];
  int g[3];
  int mode, B, prefilter;
  float kmin[3], kmax[3];
};

// the three grouped coordinates of a row: (x, y, z), or (phi [deg], z, rho) for the cylinder view (mvf_encoder.py:99-105).
// phi = atan2(y, x) / pi * 180 in fp32 -- atan2 from pnx_detmath.h (the fp64 value rounded once: torch's CPU and CUDA atan2f differ from it
// and from each other by an ulp on ~3-15 % of the points; tests/test_gpu_readers_voxel_mvf.py states the tolerance), rho = the IEEE sqrt.
__device__ __forceinline__ void group_coords(const float* __restrict__ row, int mode, float (&u)[3]) {
  const float x = row[1], y = row[2], z = row[3];
  if (mode == PNX_GROUP_CYLINDER_CLAMP) {
    u[0] = __fmul_rn(__fdiv_rn(pnx_atan2f(y, x), 3.14159274101257324f), 180.0f);
    u[1] = z;
    u[2] = __fsqrt_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)));
  } else {
    u[0] = x, u[1] = y, u[2] = z;
  }
}

// key of a row, or -1 if the row is dropped
__device__ __forceinline__ int64_t group_key(const float* __restrict__ row, const GroupGeomDev& G, const float (&u)[3]) {
  const int b = (int)row[0];  // points[:, 0:1].long(): truncation
  bool keep = b >= 0 && b < G.B;
  if (G.prefilter) {  // MVFFeatureNet.forward's range mask on the raw x, y, z (mvf_encoder.py:290-297); NaN fails every comparison
#pragma unroll
    for (int k = 0; k < 3; k++) keep = keep && row[1 + k] >= G.kmin[k] && row[1 + k] < G.kmax[k];
  }
  int idx[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float pc = __fdiv_rn(__fsub_rn(u[k], G.mn[k]), G.vs[k]);  // fp32 subtract and IEEE divide, as torch evaluates them
    if (G.mode == PNX_GROUP_VOXEL) {
      keep = keep && pc >= 0.f && pc < (float)G.g[k];         // float comparisons against the grid size (voxel_encoder.py:53-58)
    } else {
      pc = fminf(fmaxf(pc, 0.f), (float)(G.g[k] - 1));        // torch.clamp on the float coordinate (mvf_encoder.py:57-62); NaN -> 0
    }
    idx[k] = (int)pc;                                          // .long(): truncation
  }
  if (!keep) return -1;
  int64_t key = ((int64_t)b * G.g[0] + idx[0]) * G.g[1] + idx[1];
  if (G.mode == PNX_GROUP_VOXEL) key = key * G.g[2] + idx[2];
  return key;
}

__global__ __launch_bounds__(kBlock) void k_group_keys(const float* __restrict__ points, int64_t n, int stride, GroupGeomDev G, int64_t* __restrict__ key,
                                                       int32_t* __restrict__ kept, uint32_t* __restrict__ bitmap) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float* row = points + i * stride;
  float u[3];
  group_coords(row, G.mode, u);
  const int64_t k = group_key(row, G, u);
  key[i] = k;
  kept[i] = k >= 0 ? 0 : -1;  // SCAN_KEPT counts the non-negative entries
  if (k >= 0) atomicOr(&bitmap[k >> 5], 1u << (k & 31));
}

// rank (torch.unique order), unq_inv in kept-point order, points per cell, fp64 coordinate / feature sums, the cell's coords row
__global__ __launch_bounds__(kBlock) void k_group_rank(const float* __restrict__ points, int64_t n, int stride, GroupGeomDev G, const int64_t* __restrict__ key,
                                                       const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ wpre,
                                                       const uint32_t* __restrict__ wblk, const uint32_t* __restrict__ kpre,
                                                       const uint32_t* __restrict__ kblk, int32_t* __restrict__ rank_of, int64_t* __restrict__ unq_inv,
                                                       uint32_t* __restrict__ count, double* __restrict__ sum, int cm, int32_t* __restrict__ coords,
                                                       int64_t cap) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t k = key[i];
  if (k < 0) {
    rank_of[i] = -1;
    return;
  }
  const int64_t w = k >> 5;
  const uint32_t r = wblk[w >> PNX_SCAN_SHIFT] + wpre[w] + (uint32_t)__popc(bitmap[w] & ((1u << (k & 31)) - 1u));
  rank_of[i] = (int32_t)r;
  if (unq_inv) unq_inv[kblk[i >> PNX_SCAN_SHIFT] + kpre[i]] = (int64_t)r;
  const uint32_t old = atomicAdd(&count[r], 1u);
  const float* row = points + i * stride;
  if (G.mode == PNX_GROUP_VOXEL) {
    for (int c = 0; c < cm; c++) atomicAdd(&sum[(int64_t)r * cm + c], (double)row[1 + c]);
  } else {
    float u[3];
    group_coords(row, G.mode, u);
#pragma unroll
    for (int c = 0; c < 3; c++) atomicAdd(&sum[(int64_t)r * 3 + c], (double)u[c]);
  }
  if (old == 0 && coords && (int64_t)r < cap) {  // first arrival writes the cell's row of `unq`, permuted as the reference returns it
    int64_t t = k;
    if (G.mode == PNX_GROUP_VOXEL) {
      const int c2 = (int)(t % G.g[2]);
      t /= G.g[2];
      const int c1 = (int)(t % G.g[1]);
      t /= G.g[1];
      const int c0 = (int)(t % G.g[0]);
      const int b = (int)(t / G.g[0]);
      int4 q = {b, c2, c1, c0};  // unq[:, [0, 3, 2, 1]] (voxel_encoder.py:70)
      *reinterpret_cast<int4*>(coords + (int64_t)r * 4) = q;
    } else {
      const int c1 = (int)(t % G.g[1]);
      t /= G.g[1];
      const int c0 = (int)(t % G.g[0]);
      const int b = (int)(t / G.g[0]);
      coords[(int64_t)r * 3 + 0] = b, coords[(int64_t)r * 3 + 1] = c1, coords[(int64_t)r * 3 + 2] = c0;  // unq[:, [0, 2, 1]] (mvf_encoder.py:86,141)
    }
  }
}

// per kept point: the output feature row.  VOXEL: the row itself (points[mask][:, 1:], voxel_encoder.py:68); CLAMP modes: the decorated row
// [u0 u1 u2 f.. | u - mean of the cell | u[:2] - centre of the cell] (mvf_encoder.py:73-83), each term evaluated as torch does (fp32, left to right).
__global__ __launch_bounds__(kBlock) void k_group_features(const float* __restrict__ points, int64_t n, int stride, GroupGeomDev G,
                                                           const int64_t* __restrict__ key, const int32_t* __restrict__ rank_of,
                                                           const uint32_t* __restrict__ kpre, const uint32_t* __restrict__ kblk,
                                                           const uint32_t* __restrict__ count, const double* __restrict__ sum, float* __restrict__ out,
                                                           int ld) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t r = rank_of[i];
  if (r < 0) return;
  const float* row = points + i * stride;
  float* o = out + (int64_t)(kblk[i >> PNX_SCAN_SHIFT] + kpre[i]) * ld;
  if (G.mode == PNX_GROUP_VOXEL) {
    for (int c = 0; c < stride - 1; c++) o[c] = row[1 + c];
    return;
  }
  float u[3];
  group_coords(row, G.mode, u);
  const int nf = stride - 1;
#pragma unroll
  for (int c = 0; c < 3; c++) o[c] = u[c];
  for (int c = 3; c < nf; c++) o[c] = row[1 + c];
  const float cnt = (float)count[r];
#pragma unroll
  for (int c = 0; c < 3; c++) o[nf + c] = __fsub_rn(u[c], __fdiv_rn((float)sum[(int64_t)r * 3 + c], cnt));
  int64_t t = key[i];
  const int c1 = (int)(t % G.g[1]);
  t /= G.g[1];
  const int c0 = (int)(t % G.g[0]);
  const int ci[2] = {c0, c1};
#pragma unroll
  for (int c = 0; c < 2; c++)  // idx * vs + vs / 2 + min, left to right (mvf_encoder.py:76-79)
    o[nf + 3 + c] = __fsub_rn(u[c], __fadd_rn(__fadd_rn(__fmul_rn((float)ci[c], G.vs[c]), __fdiv_rn(G.vs[c], 2.0f)), G.mn[c]));
}

__global__ __launch_bounds__(kBlock) void k_group_mean(const uint32_t* __restrict__ count, const double* __restrict__ sum, const int32_t* __restrict__ counters,
                                                       int cm, int64_t cap, float* __restrict__ mean) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t r = e / cm;
  if (r >= (int64_t)counters[0] || r >= cap) return;
  mean[e] = __fdiv_rn((float)sum[e], (float)count[r]);
}

struct GroupWs {
  int32_t* counters;  // {G, N'}
  uint32_t *bitmap, *count;
  double* sum;
  size_t zero_bytes;  // counters | bitmap | count | sum are contiguous: one memset
  uint32_t *wpre, *wblk, *kpre, *kblk;
  int64_t* key;
  int32_t *kept, *rank_of;
  int64_t nwords;
  int nblk_w, nblk_k;
  size_t bytes;
};

int64_t group_cells(const pnx_group_geom* g, int batch) {
  int64_t c = (int64_t)batch * g->grid[0] * g->grid[1];
  if (g->mode == PNX_GROUP_VOXEL) c *= g->grid[2];
  return c;
}
int group_cm(const pnx_group_geom* g, int stride) { return g->mode == PNX_GROUP_VOXEL ? stride - 1 : 3; }

GroupWs group_carve(void* ws, int64_t n, int stride, int batch, const pnx_group_geom* g) {
  GroupWs w;
  PnxCarver c(ws);
  const int64_t cells = group_cells(g, batch);
  w.nwords = (cells + 31) / 32;
  w.nblk_w = (int)((w.nwords + PNX_SCAN_ITEMS - 1) / PNX_SCAN_ITEMS);
  w.nblk_k = (int)((n + PNX_SCAN_ITEMS - 1) / PNX_SCAN_ITEMS);
  if (w.nblk_k < 1) w.nblk_k = 1;
  const int64_t gmax = n < cells ? n : cells;  // at most one cell per point
  w.counters = c.take<int32_t>(64);
  w.bitmap = c.take<uint32_t>(w.nwords + 8);
  w.count = c.take<uint32_t>(gmax + 8);
  w.sum = c.take<double>((gmax + 8) * group_cm(g, stride));
  w.zero_bytes = c.used();
  w.wpre = c.take<uint32_t>(w.nwords + 8);
  w.wblk = c.take<uint32_t>(w.nblk_w + 8);
  w.kpre = c.take<uint32_t>(n + 8);
  w.kblk = c.take<uint32_t>(w.nblk_k + 8);
  w.key = c.take<int64_t>(n + 8);
  w.kept = c.take<int32_t>(n + 8);
  w.rank_of = c.take<int32_t>(n + 8);
  w.bytes = c.used();
  return w;
}

int group_check_geom(const pnx_group_geom* g, int batch, int stride) {
  PNX_REQUIRE(g != nullptr, PNX_ERR_INVALID, "geom is NULL");
  PNX_REQUIRE(g->mode >= PNX_GROUP_VOXEL && g->mode <= PNX_GROUP_CYLINDER_CLAMP, PNX_ERR_INVALID, "mode %d", g->mode);
  PNX_REQUIRE(batch >= 1 && stride >= 4 && stride <= 64, PNX_ERR_INVALID, "batch %d, row_stride %d", batch, stride);
  for (int k = 0; k < 3; k++) PNX_REQUIRE(g->grid[k] >= 1 && g->voxel[k] > 0.f, PNX_ERR_INVALID, "grid / voxel size of axis %d", k);
  PNX_REQUIRE(group_cells(g, batch) < ((int64_t)1 << 36), PNX_ERR_UNSUPPORTED, "more than 2^36 cells");
  return PNX_OK;
}

// ---- PFNLayer, eval mode, any widths.  Lanes = output channels, a wave walks points: the point's input row sits one element per lane
// and is broadcast with v_readlane, the (transposed) folded weight is read from LDS with consecutive lanes on consecutive banks; the
// per-cell maximum is an integer atomicMax on the bit pattern (the values are >= 0 behind the ReLU and every cell starts at +0).
__global__ __launch_bounds__(kBlock) void k_pfn_layer(const float* __restrict__ xa, int lda, int ca, const float* __restrict__ gb, int cb,
                                                      const int64_t* __restrict__ inv, const float* __restrict__ wt, const float* __restrict__ shift, int cout,
                                                      int64_t n, float* __restrict__ y, int ldy, uint32_t* __restrict__ gmax) {
  extern __shared__ float s_wt[];  // [cin][cout]
  const int cin = ca + cb;
  for (int e = threadIdx.x; e < cin * cout; e += kBlock) s_wt[e] = wt[e];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t p = wave; p < n; p += nwaves) {
    const int64_t g = inv ? inv[p] : 0;
    float f[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int k = lane + 64 * h;
      float v = 0.f;
      if (k < ca) v = xa[p * lda + k];
      else if (k < cin) v = gb[g * cb + (k - ca)];
      f[h] = v;
    }
    for (int c0 = 0; c0 < cout; c0 += 64) {
      const int c = c0 + lane;
      const bool on = c < cout;
      float acc = on ? shift[c] : 0.f;
      const float* wc = s_wt + (on ? c : 0);
      const int k1 = cin < 64 ? cin : 64;
      for (int k = 0; k < k1; k++) acc = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f[0]), k)), wc[k * cout], acc);
      for (int k = 64; k < cin; k++)
        acc = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, f[1]), k - 64)), wc[k * cout], acc);
      acc = acc > 0.f ? acc : 0.f;  // never -0.0: the unsigned atomicMax below orders bit patterns, and 0x80000000 would beat every positive value
      if (on) {
        if (y) y[p * ldy + c] = acc;
        if (gmax) atomicMax(&gmax[g * cout + c], __float_as_uint(acc));
      }
    }
  }
}

// ---- bilinear sampling of a channels-last map at the points (mvf_encoder.py:208-246): one wave per point, lanes = channels
template <typename T>
__device__ __forceinline__ float ld_f(const T* p);
template <>
__device__ __forceinline__ float ld_f<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld_f<uint16_t>(const uint16_t* p) { return __uint_as_float((uint32_t)*p << 16); }  // bf16
template <>
__device__ __forceinline__ float ld_f<_Float16>(const _Float16* p) { return (float)*p; }
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bilinear(const T* __restrict__ img, int B, int H, int W, int C, const float* __restrict__ pos, int ldp,
                                                     float mn0, float mn1, float vs0, float vs1, const int32_t* __restrict__ cell_coords,
                                                     const int64_t* __restrict__ inv, float inv_ds, int64_t n, float* __restrict__ out, int ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t p = wave; p < n; p += nwaves) {
    const int b = cell_coords[inv[p] * 3];                 // unq[unq_inv][:, 0] (mvf_encoder.py:203)
    // (feature_pos - bias) / voxel_size (mvf_encoder.py:184), then / ds_rate (:204): ds_rate is a power of two, so * (1 / ds) is exact
    const float x = __fmul_rn(__fdiv_rn(__fsub_rn(pos[p * ldp + 0], mn0), vs0), inv_ds);
    const float y = __fmul_rn(__fdiv_rn(__fsub_rn(pos[p * ldp + 1], mn1), vs1), inv_ds);
    int x0 = (int)floorf(x), y0 = (int)floorf(y);
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), W - 1), x1 = min(max(x1, 0), W - 1);
    y0 = min(max(y0, 0), H - 1), y1 = min(max(y1, 0), H - 1);
    // the weights use the CLAMPED corners, as the reference does (:233-240)
    const float wa = __fmul_rn(__fsub_rn((float)x1, x), __fsub_rn((float)y1, y));
    const float wb = __fmul_rn(__fsub_rn((float)x1, x), __fsub_rn(y, (float)y0));
    const float wc = __fmul_rn(__fsub_rn(x, (float)x0), __fsub_rn((float)y1, y));
    const float wd = __fmul_rn(__fsub_rn(x, (float)x0), __fsub_rn(y, (float)y0));
    const bool okb = b >= 0 && b < B;
    const T* ia = img + (((int64_t)(okb ? b : 0) * H + y0) * W + x0) * C;
    const T* ib = img + (((int64_t)(okb ? b : 0) * H + y1) * W + x0) * C;
    const T* ic = img + (((int64_t)(okb ? b : 0) * H + y0) * W + x1) * C;
    const T* id = img + (((int64_t)(okb ? b : 0) * H + y1) * W + x1) * C;
    for (int c = lane; c < C; c += 64) {
      float v = __fmul_rn(ld_f(ia + c), wa);
      v = __fadd_rn(v, __fmul_rn(ld_f(ib + c), wb));
      v = __fadd_rn(v, __fmul_rn(ld_f(ic + c), wc));
      v = __fadd_rn(v, __fmul_rn(ld_f(id + c), wd));
      out[p * ldo + c] = okb ? v : 0.f;
    }
  }
}

}  // namespace

extern "C" {

size_t pnx_group_workspace_bytes(int64_t n_points, int32_t row_stride, int32_t batch, const pnx_group_geom* geom_host) {
  if (n_points < 0 || group_check_geom(geom_host, batch, row_stride) != PNX_OK) return 0;
  return group_carve(nullptr, n_points, row_stride, batch, geom_host).bytes;
}

int pnx_group_points(const float* points, int64_t n, int32_t stride, int32_t batch, const pnx_group_geom* g, float* point_features, int32_t feature_ld,
                     int32_t* coords, int64_t group_capacity, int64_t* unq_inv, float* group_mean, int32_t* counts, void* workspace, size_t workspace_bytes,
                     pnx_stream_t stream) {
  int rc = group_check_geom(g, batch, stride);
  if (rc != PNX_OK) return rc;
  PNX_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) - 4096, PNX_ERR_INVALID, "n_points %lld", (long long)n);
  PNX_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 255) == 0, PNX_ERR_INVALID, "workspace must be 256-byte aligned");
  PNX_REQUIRE(n == 0 || points != nullptr, PNX_ERR_INVALID, "points is NULL");
  const int nf_out = g->mode == PNX_GROUP_VOXEL ? stride - 1 : stride - 1 + 5;
  PNX_REQUIRE(point_features == nullptr || feature_ld >= nf_out, PNX_ERR_INVALID, "feature_ld %d < %d columns", feature_ld, nf_out);
  PNX_REQUIRE((coords == nullptr && group_mean == nullptr) || group_capacity > 0, PNX_ERR_INVALID, "group_capacity must be > 0");
  PNX_REQUIRE(coords == nullptr || ((uintptr_t)coords & 15) == 0, PNX_ERR_INVALID, "coords must be 16-byte aligned");
  const GroupWs w = group_carve(workspace, n, stride, batch, g);
  PNX_REQUIRE(workspace_bytes >= w.bytes, PNX_ERR_WORKSPACE, "workspace %zu bytes < %zu needed", workspace_bytes, w.bytes);
  hipStream_t st = (hipStream_t)stream;
  GroupGeomDev G;
  for (int k = 0; k < 3; k++) G.mn[k] = g->min[k], G.vs[k] = g->voxel[k], G.g[k] = g->grid[k], G.kmin[k] = g->keep_min[k], G.kmax[k] = g->keep_max[k];
  G.mode = g->mode, G.B = batch, G.prefilter = g->prefilter;
  const int cm = group_cm(g, stride);
  PNX_CHECK_HIP(hipMemsetAsync(w.counters, 0, w.zero_bytes, st));
  const unsigned nbn = (unsigned)((n + kBlock - 1) / kBlock);
  if (n > 0) {
    k_group_keys<<<nbn, kBlock, 0, st>>>(points, n, stride, G, w.key, w.kept, w.bitmap);
    PNX_LAUNCH_CHECK();
  }
  k_scan_local<SCAN_POPC><<<w.nblk_w, kBlock, 0, st>>>(w.bitmap, w.nwords, w.wpre, w.wblk);
  k_scan_blocks<<<1, kBlock, 0, st>>>(w.wblk, w.nblk_w, w.counters + 0);
  k_scan_local<SCAN_KEPT><<<w.nblk_k, kBlock, 0, st>>>(reinterpret_cast<const uint32_t*>(w.kept), n, w.kpre, w.kblk);
  k_scan_blocks<<<1, kBlock, 0, st>>>(w.kblk, w.nblk_k, w.counters + 1);
  PNX_LAUNCH_CHECK();
  if (n > 0) {
    k_group_rank<<<nbn, kBlock, 0, st>>>(points, n, stride, G, w.key, w.bitmap, w.wpre, w.wblk, w.kpre, w.kblk, w.rank_of, unq_inv, w.count, w.sum, cm, coords,
                                         group_capacity);
    if (point_features)
      k_group_features<<<nbn, kBlock, 0, st>>>(points, n, stride, G, w.key, w.rank_of, w.kpre, w.kblk, w.count, w.sum, point_features, feature_ld);
    if (group_mean) {
      const int64_t gmax = (n < group_capacity ? n : group_capacity) * cm;
      k_group_mean<<<(unsigned)((gmax + kBlock - 1) / kBlock), kBlock, 0, st>>>(w.count, w.sum, w.counters, cm, group_capacity, group_mean);
    }
    PNX_LAUNCH_CHECK();
  }
  if (counts) PNX_CHECK_HIP(hipMemcpyAsync(counts, w.counters, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return PNX_OK;
}

int pnx_pfn_layer_eval(const float* xa, int32_t lda, int32_t ca, const float* gb, int32_t cb, const int64_t* inv, const float* wt, const float* shift,
                       int32_t cout, int64_t n, int64_t num_groups, float* y, int32_t ldy, float* gmax, pnx_stream_t stream) {
  PNX_REQUIRE(n >= 0 && ca >= 0 && cb >= 0 && ca + cb >= 1 && ca + cb <= 128 && cout >= 1 && cout <= 256, PNX_ERR_UNSUPPORTED,
              "pnx_pfn_layer_eval: %d + %d inputs (at most 128), %d outputs (at most 256)", ca, cb, cout);
  PNX_REQUIRE(wt && shift && (ca == 0 || (xa && lda >= ca)) && (cb == 0 || (gb && inv)) && (gmax == nullptr || inv) && (y == nullptr || ldy >= cout),
              PNX_ERR_INVALID, "pnx_pfn_layer_eval: null pointer / leading dimension");
  PNX_REQUIRE((size_t)(ca + cb) * cout * sizeof(float) <= 64 * 1024, PNX_ERR_UNSUPPORTED, "pnx_pfn_layer_eval: weight larger than 64 KiB");
  hipStream_t st = (hipStream_t)stream;
  if (gmax && num_groups > 0) PNX_CHECK_HIP(hipMemsetAsync(gmax, 0, (size_t)num_groups * cout * sizeof(float), st));  // +0: below every ReLU output
  if (n == 0) return PNX_OK;
  int64_t nb = (n + 3) / 4;
  if (nb > 4096) nb = 4096;
  k_pfn_layer<<<(unsigned)nb, kBlock, (size_t)(ca + cb) * cout * sizeof(float), st>>>(xa, lda, ca, gb, cb, inv, wt, shift, cout, n, y, ldy,
                                                                                      reinterpret_cast<uint32_t*>(gmax));
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

int pnx_bilinear_gather(const void* image, int32_t dtype, int32_t batch, int32_t h, int32_t w, int32_t channels, const float* pos, int32_t pos_ld,
                        const float* pos_min2_host, const float* pos_voxel2_host, const int32_t* cell_coords, const int64_t* unq_inv, int32_t ds_rate,
                        int64_t n, float* out, int32_t out_ld, pnx_stream_t stream) {
  PNX_REQUIRE(image && pos && pos_min2_host && pos_voxel2_host && cell_coords && unq_inv && out && batch > 0 && h > 0 && w > 0 && channels > 0 && pos_ld >= 2 && out_ld >= channels && n >= 0,
              PNX_ERR_INVALID, "pnx_bilinear_gather: bad arguments");
  PNX_REQUIRE(dtype == PNX_F32 || dtype == PNX_BF16 || dtype == PNX_F16, PNX_ERR_UNSUPPORTED, "pnx_bilinear_gather: fp32, bf16 or fp16 maps");
  PNX_REQUIRE(ds_rate >= 1 && (ds_rate & (ds_rate - 1)) == 0, PNX_ERR_UNSUPPORTED, "pnx_bilinear_gather: ds_rate %d is not a power of two", ds_rate);
  if (n == 0) return PNX_OK;
  int64_t nb = (n + 3) / 4;
  if (nb > 8192) nb = 8192;
  hipStream_t st = (hipStream_t)stream;
  const float inv_ds = 1.0f / (float)ds_rate;
  if (dtype == PNX_F32)
    k_bilinear<float><<<(unsigned)nb, kBlock, 0, st>>>((const float*)image, batch, h, w, channels, pos, pos_ld, pos_min2_host[0], pos_min2_host[1], pos_voxel2_host[0],
                                                       pos_voxel2_host[1], cell_coords, unq_inv, inv_ds, n, out, out_ld);
  else if (dtype == PNX_BF16)
    k_bilinear<uint16_t><<<(unsigned)nb, kBlock, 0, st>>>((const uint16_t*)image, batch, h, w, channels, pos, pos_ld, pos_min2_host[0], pos_min2_host[1],
                                                          pos_voxel2_host[0], pos_voxel2_host[1], cell_coords, unq_inv, inv_ds, n, out, out_ld);
  else
    k_bilinear<_Float16><<<(unsigned)nb, kBlock, 0, st>>>((const _Float16*)image, batch, h, w, channels, pos, pos_ld, pos_min2_host[0], pos_min2_host[1],
                                                          pos_voxel2_host[0], pos_voxel2_host[1], cell_coords, unq_inv, inv_ds, n, out, out_ld);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // extern "C"
