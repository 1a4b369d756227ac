/* pnx.h -- C ABI of libpnx_hip.so: the MI355X (gfx950) PillarNeXt hot path.
 *
 * Drop-in boundary for two reference surfaces (paths relative to the reference tree):
 *   B1  det3d/models/readers/pillar_encoder.py  PillarFeatureNet.forward  (:174-182)
 *       = PillarNet.forward (:78-125) + 2 x PFNLayer.forward (:35-50) + scatter_max (:180),
 *       plus the dense canvas of SparseConvTensor(...).dense() (det3d/models/backbones/sparse_resnet.py:63-68)
 *   B2  det3d/core/iou3d_nms/src/iou3d_nms_api.cpp:11-19 (the 7 pybind exports of iou3d_nms_cuda)
 *       and det3d/core/bbox/box_torch_ops.py:5-31 (rotate_nms_pcdet)
 *
 * Conventions
 *   - plain C, no torch types; every pointer is a DEVICE pointer unless its name ends in _host
 *   - the caller owns every output and the workspace; nothing is allocated on the hot path
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *   - return value: PNX_OK (0) or a negative pnx_status; pnx_last_error() gives a message.
 *     Unlike the reference (iou3d_nms.cpp:14-38) nothing ever calls exit().
 *   - not re-entrant on the same workspace; one workspace per stream.
 */
#ifndef PNX_H
#define PNX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pnx_stream_t; /* hipStream_t */

typedef enum pnx_status {
  PNX_OK = 0,
  PNX_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, misaligned buffer) */
  PNX_ERR_UNSUPPORTED = -2, /* configuration outside what the kernels are built for */
  PNX_ERR_WORKSPACE = -3,   /* workspace too small */
  PNX_ERR_HIP = -4          /* a HIP runtime call failed */
} pnx_status;

typedef enum pnx_dtype { PNX_F32 = 0, PNX_BF16 = 1, PNX_F16 = 2 } pnx_dtype;
typedef enum pnx_layout {
  PNX_NHWC = 0, /* [B][gy][gx][C]  (torch channels_last memory of a (B,C,gy,gx) tensor) */
  PNX_NCHW = 1  /* [B][C][gy][gx]  (what .dense() returns, sparse_resnet.py:68) */
} pnx_layout;

/* Voxel grid.  pc_min/voxel are the fp32 casts the reference applies (pillar_encoder.py:91-93);
 * gx,gy = np.round((max-min)/voxel) evaluated in fp64, half-to-even (:87-89). */
typedef struct pnx_geom {
  float pc_min[3];
  float voxel[3];
  int32_t gx, gy;
} pnx_geom;

const char* pnx_last_error(void);
const char* pnx_version(void);

/* pillar_encoder.py:87-93 -- host helper, fills *geom from the YAML's fp64 lists. */
int pnx_geom_init(const double* pc_range6_host, const double* voxel_size3_host, pnx_geom* geom_host);

/* ------------------------------------------------------------------------------------------------
 * Reader, inference (BatchNorm in eval mode).
 *
 * PFN parameters are folded once per weight update into one device buffer of PNX_PFN_FOLDED_FLOATS(F)
 * floats: [W0' (32 x (F+5)) | s0 (32) | W1' (64 x 64) | s1 (64)] with a = gamma / sqrt(running_var + eps),
 * W'[c,:] = a[c] * W[c,:] and s = beta - running_mean * a   (BatchNorm1d eval folded into the Linear, :32-33,37-38;
 * SURVEY.md H8 measured this fold at 1.4e-6 abs from the reference), followed by the same numbers re-ordered as
 * MFMA fragments (64 lanes x 121 registers, then 64 x 71 for the fp16x3 form of layer 1) so that the PFN kernels load them with
 * coalesced reads.
 * Only num_filters = [64, 64] (every PillarNeXt config) and 3 <= F <= 6 are built.
 */
#define PNX_PFN_FOLDED_FLOATS(F) (32 * ((F) + 5) + 32 + 64 * 64 + 64 + 64 * 121 + 64 * 71)

int pnx_pfn_fold_bn(int32_t num_point_features, /* F */
                    const float* w0, const float* gamma0, const float* beta0, const float* mean0, const float* var0,
                    const float* w1, const float* gamma1, const float* beta1, const float* mean1, const float* var1,
                    float eps, float* folded_out, pnx_stream_t stream);

/* Bytes of workspace pnx_reader_forward needs for at most n_points rows and `batch` samples. */
size_t pnx_reader_workspace_bytes(int64_t n_points, int32_t batch, const pnx_geom* geom_host);

/* PillarFeatureNet.forward (+ dense canvas).
 *   points      (n_points, row_stride) fp32 rows [b, x, y, z, f4..]; row_stride = 1 + F
 *   batch       number of samples B; rows with b outside [0,B) are dropped
 * Outputs (each may be NULL):
 *   canvas      B x 64 x gy x gx in canvas_dtype / canvas_layout; every element is written exactly once
 *               (zeros where no pillar)  -- the dense input of the backbone
 *   occupancy   B x gy x gx uint8, 1 where a pillar exists (the active-site set of the reference's
 *               SparseConvTensor; needed because a pillar's features may all be zero); requires canvas
 *   feat_max    (pillar_capacity, 64) fp32, row r = pillar of rank r            (:180-182)
 *   coords      (pillar_capacity, 3) int32 [b, yi, xi], torch.unique order      (:110-111,125)
 *   unq_inv     (n_points) int64: pillar rank of the j-th KEPT point, in input order (:110)
 *   pillar_of_point (n_points) int32: pillar rank of input row i, -1 if the row was dropped
 *   counts      int32[2] = {P (number of pillars), N' (number of kept points)}
 * If P would exceed pillar_capacity the rows beyond it are not written (P is still reported).
 * Streams: with an NHWC canvas the zero-fill of the pillar-free cells runs on an internal stream of the library (one per host thread and
 * device) beside the grouping / PFN kernels; it is forked from and joined back into `stream` with events inside this call, so for the
 * caller everything is ordered on `stream` as for any other call.
 */
int pnx_reader_forward(const float* points, int64_t n_points, int32_t row_stride, int32_t batch, const pnx_geom* geom_host,
                       const float* pfn_folded, void* canvas, int32_t canvas_dtype, int32_t canvas_layout, uint8_t* occupancy,
                       float* feat_max, int32_t* coords, int64_t pillar_capacity, int64_t* unq_inv, int32_t* pillar_of_point,
                       int32_t* counts, void* workspace, size_t workspace_bytes, pnx_stream_t stream);

/* Percent of the canvas zero-fill tiles that the fused reader hands to extra blocks of its three grouping kernels (the PFN launch
 * takes the rest); PNX_FILL_SPLIT="a,b,c" overrides the built-in split.  For reports only. */
void pnx_reader_fill_split(int32_t* percent3_host);

/* Measurement hooks for bench.py (no effect on results): between begin and end every pnx_reader_forward records
 * HIP events on ITS stream around (a) the whole reader, (b) the canvas writer (HBM-bound) and (c) the PFN kernel (MFMA-bound).
 * pnx_profile_end synchronises those events and returns average microseconds per call. */
int pnx_profile_begin(int32_t max_samples);
int pnx_profile_end(float* reader_us_avg_host, float* canvas_us_avg_host, int32_t* samples_host);
float pnx_profile_last_pfn_us(void); /* average PFN-kernel microseconds of the interval closed by the last pnx_profile_end */
float pnx_profile_last_voxelize_us(void); /* same interval: reader start -> pillar-sorted records ready (keys, scans, binning, bin sort) */

/* ------------------------------------------------------------------------------------------------
 * Reader, TRAINING mode (BatchNorm1d with batch statistics, pillar_encoder.py:33,38) and its backward, fused: no (N',32/64)
 * activation or gradient tensor is ever materialised, every pass recomputes the per-point chain from the pillar-sorted records that
 * pass 0 leaves in `workspace` (csrc/pfn_train.hip).  The passes return small per-block / per-wave partial sums; between two
 * passes the caller reduces them (fp64) and -- under SyncBatchNorm, tools/train.py:56 -- all-reduces the statistics (the 65- and
 * 129-float vectors torch's SyncBatchNorm exchanges forward, 2x64 and 2x32 backward) before it builds the next parameter block.
 * The host side of this protocol, including the BatchNorm algebra for dW0/dW1, is pillarnext_amd/pfn_train.py.
 *   params    pnx_pfn_train_param_floats(F) floats: W1 (64x64) | mu1 invstd1 gamma1 beta1 m1 m2 (6 x 64) | mu0 invstd0 gamma0 beta0
 *             (4 x 32) | W0 (32 x (F+5));  m1 = sum(dz1)/N, m2 = sum(dz1*xhat1)/N (backward pass 1 only)
 *   partials  pnx_pfn_train_partial_floats(F, which) floats, which = 0,1 (forward pass 0,1), 3,4 (backward pass 0,1)
 * pnx_pfn_forward_train: pass 0 groups the points (coords, pillar_of_point, counts = {P, N'} as in pnx_reader_forward) and returns the
 *   Gram sums of the decorated features; pass 1 (params with mu0/invstd0) the Gram sums of u = [h0, max h0]; pass 2 (mu1/invstd1 too)
 *   writes feat_max (P,64) fp32.
 * pnx_pfn_backward: pass 0: grad_feat_max routed through the pillar maxima and the ReLU -> sum dz1, sum dz1*xhat1, sum dz1^T u;
 *   pass 1 (params with m1, m2): -> sum dz0, sum dz0*xhat0, sum dz0^T f.  No gradient flows to the points (:91-123 is index math). */
size_t pnx_pfn_train_param_floats(int32_t num_point_features);
size_t pnx_pfn_train_partial_floats(int32_t num_point_features, int32_t which);
int pnx_pfn_forward_train(int32_t pass, const float* points, int64_t n_points, int32_t row_stride, int32_t batch, const pnx_geom* geom_host,
                          const float* params, float* partials, float* feat_max, int64_t pillar_capacity, int32_t* coords,
                          int32_t* pillar_of_point, int32_t* counts, void* workspace, size_t workspace_bytes, pnx_stream_t stream);
int pnx_pfn_backward(int32_t pass, int64_t n_points, int32_t row_stride, int32_t batch, const pnx_geom* geom_host, const float* params,
                     const float* grad_feat_max, const float* feat_max, float* partials, void* workspace, size_t workspace_bytes,
                     pnx_stream_t stream);

/* Voxelizer alone (PillarNet.forward :78-125): indices plus the decorated (N', F+5) features
 * (rows in kept-point order; may be NULL).  The reference-API surface of PillarNet and the UNFUSED training fallback
 * (PNX_TRAIN_FUSED=0, PFN shapes other than [64,64]: torch Linear/BatchNorm on these features + pnx_scatter_max); the default
 * training path is pnx_pfn_forward_train / pnx_pfn_backward above, which never materialises the features. */
int pnx_voxelize(const float* points, int64_t n_points, int32_t row_stride, int32_t batch, const pnx_geom* geom_host,
                 float* features, int32_t* coords, int64_t pillar_capacity, int64_t* unq_inv, int32_t* pillar_of_point,
                 int32_t* counts, void* workspace, size_t workspace_bytes, pnx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Voxel and multi-view readers: det3d/models/readers/voxel_encoder.py:25-87, det3d/models/readers/mvf_encoder.py:39-246.
 *
 * pnx_group_points = the point -> cell grouping of VoxelNet.forward (mode PNX_GROUP_VOXEL: 3-D cells, rows outside the range dropped by the
 * float comparisons of voxel_encoder.py:53-58), PillarVoxelNet.forward (PNX_GROUP_PILLAR_CLAMP: 2-D cells over x, y; the cell index is
 * CLAMPED, nothing is dropped, mvf_encoder.py:57-62) and CylinderNet.forward (PNX_GROUP_CYLINDER_CLAMP: the same over phi = atan2(y, x) / pi *
 * 180 [deg], z; rho = sqrt(x^2 + y^2) is the third coordinate, :99-105), with torch.unique(dim=0)'s order and torch_scatter.scatter_mean:
 *   points         (n_points, row_stride) fp32 rows [b, x, y, z, f..]
 *   geom           min / voxel / grid of the three grouped axes IN THE ORDER OF THE YAML LISTS ((x, y, z) or (phi, z, rho)), mode, and an optional
 *                  prefilter: rows whose raw x, y, z are outside [keep_min, keep_max) are dropped first (MVFFeatureNet.forward, :290-297);
 *                  rows with b outside [0, batch) are always dropped
 *   point_features (N', feature_ld) fp32, row j = the j-th KEPT point in input order:
 *                    VOXEL            [x y z f..]                      (row_stride - 1 columns: points[mask][:, 1:], voxel_encoder.py:68)
 *                    *_CLAMP          [u0 u1 u2 f.. | u - mean(cell) (3) | u[:2] - centre(cell) (2)]   (row_stride + 4 columns, :73-83 / :127-138)
 *   coords         (group_capacity, 4) int32 [b, c2, c1, c0] for VOXEL (= unq[:, [0,3,2,1]], i.e. [b, z, y, x]); (group_capacity, 3) [b, c1, c0] for
 *                  the 2-D modes (= unq[:, [0,2,1]]); rows in torch.unique order
 *   unq_inv        (N') int64: cell rank of the j-th kept point
 *   group_mean     (group_capacity, M) fp32: per-cell mean of the row's row_stride - 1 columns (VOXEL: DynamicVoxelEncoder.forward,
 *                  voxel_encoder.py:19-22) or of the three grouped coordinates (2-D modes)
 *   counts         int32[2] = {G cells, N' kept points}
 * Each output may be NULL.  Deterministic: sums are fp64 atomics (exact here), mean = fp32(sum) / fp32(count).
 * Indices are bit-exact with torch on the CPU for the Cartesian modes; in the cylinder mode atan2 is pnx_detmath.h's (the fp64 value
 * rounded once), which differs from torch's CPU / CUDA atan2f by one ulp on a few percent of the points: phi within 2 ulp(180) = 3.1e-5 deg,
 * rho within 1 ulp (torch's CPU sqrt is not correctly rounded), a point within that distance of a bin edge may fall into the neighbouring cell. */
typedef enum pnx_group_mode { PNX_GROUP_VOXEL = 0, PNX_GROUP_PILLAR_CLAMP = 1, PNX_GROUP_CYLINDER_CLAMP = 2 } pnx_group_mode;
typedef struct pnx_group_geom {
  float min[3], voxel[3]; /* fp32 casts of pc_range[:3] / voxel_size, as the reference applies them */
  int32_t grid[3];        /* np.round((max - min) / voxel) in fp64 */
  int32_t mode;           /* pnx_group_mode */
  int32_t prefilter;      /* != 0: drop rows with raw x, y, z outside [keep_min, keep_max) */
  float keep_min[3], keep_max[3];
} pnx_group_geom;
size_t pnx_group_workspace_bytes(int64_t n_points, int32_t row_stride, int32_t batch, const pnx_group_geom* geom_host);
int pnx_group_points(const float* points, int64_t n_points, int32_t row_stride, int32_t batch, const pnx_group_geom* geom_host, float* point_features,
                     int32_t feature_ld, int32_t* coords, int64_t group_capacity, int64_t* unq_inv, float* group_mean, int32_t* counts, void* workspace,
                     size_t workspace_bytes, pnx_stream_t stream);
/* PFNLayer.forward in eval mode for any layer widths (pillar_encoder.py:35-50 as SingleView stacks it, mvf_encoder.py:150-163,187-188):
 *   x = relu(W' [xa | gb[inv]] + shift)   with BatchNorm1d(eval) folded into W' / shift (pnx_pfn_fold_bn's algebra);
 *   xa (n, lda) fp32: the first ca input columns; gb (num_groups, cb) fp32: the per-cell maximum of the PREVIOUS layer, read through inv -- the
 *   torch.cat([x, x_max[unq_inv]]) of a non-last layer is never materialised; wt = W' transposed, (ca + cb, cout) fp32; shift (cout);
 *   y (n, ldy) = x per point (may be NULL: the last layer), gmax (num_groups, cout) = scatter_max(x, inv) (may be NULL), fully written
 *   (cells without a point hold 0).  ca + cb <= 128, cout <= 256.  The maximum is order-free: deterministic. */
int pnx_pfn_layer_eval(const float* xa, int32_t lda, int32_t ca, const float* gb, int32_t cb, const int64_t* inv, const float* wt, const float* shift,
                       int32_t cout, int64_t n, int64_t num_groups, float* y, int32_t ldy, float* gmax, pnx_stream_t stream);
/* SingleView.bilinear_interpolate (mvf_encoder.py:208-246) on a channels-last map: image (batch, h, w, channels) fp32 / bf16 / fp16;
 *   sample position of point p = ((pos[p, 0:2] - pos_min) / pos_voxel) / ds_rate in (w, h) order (:184,:204), sample image = cell_coords[unq_inv[p]][0]
 *   (:203); corners clamped to the map, weights from the CLAMPED corners (:227-240); out (n, out_ld) fp32.  ds_rate must be a power of two. */
int pnx_bilinear_gather(const void* image, int32_t dtype, int32_t batch, int32_t h, int32_t w, int32_t channels, const float* pos, int32_t pos_ld,
                        const float* pos_min2_host, const float* pos_voxel2_host, const int32_t* cell_coords, const int64_t* unq_inv, int32_t ds_rate,
                        int64_t n, float* out, int32_t out_ld, pnx_stream_t stream);

/* torch_scatter.scatter_max(x, unq_inv, dim=0) over pillars (call sites :43,:180), fp32, values of
 * any sign.  x (n, channels); index (n) int64 in [0,P); out (P, channels); argmax (P, channels) int64 =
 * row of the maximum (lowest row index wins ties) or n for an empty pillar.  Deterministic. */
size_t pnx_scatter_max_workspace_bytes(int64_t n, int64_t num_pillars);
int pnx_scatter_max(const float* x, const int64_t* index, int64_t n, int32_t channels, int64_t num_pillars, float* out,
                    int64_t* argmax, void* workspace, size_t workspace_bytes, pnx_stream_t stream);
/* its backward: grad_x[argmax[p,c], c] = grad_out[p,c], zero elsewhere (grad_x is fully written). */
int pnx_scatter_max_backward(const float* grad_out, const int64_t* argmax, int64_t n, int32_t channels, int64_t num_pillars,
                             float* grad_x, pnx_stream_t stream);

/* Scatter an existing (P,64) fp32 pillar list to the dense canvas (sparse_resnet.py:63-68). */
int pnx_scatter_canvas(const float* feat_max, const int32_t* coords, const int32_t* num_pillars_dev, int64_t pillar_capacity,
                       int32_t batch, int32_t gy, int32_t gx, void* canvas, int32_t canvas_dtype, int32_t canvas_layout,
                       pnx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Input contract: multi-sweep merge + collation on the device.
 * det3d/datasets/nuscenes/nusc.py:76-121 (read_sweep, remove_close, load_pointcloud), det3d/datasets/waymo/waymo.py:49-67 and
 * det3d/datasets/loader/collate.py:15-22 as one stable compaction over raw sweeps resident in HBM.
 *   raw        (n_raw, raw_stride) fp32 rows [x, y, z, c3, ...]; segment s = rows [begin, end) = one sweep of one sample
 *   seg_descs  n_segments descriptors of pnx_merge_sweeps_desc_bytes() bytes each, in row order (device memory; layout:
 *              pillarnext_amd/io.py::pack_segments): 3x4 fp64 transform (applied in fp64, stored as fp32 -- numpy's
 *              `T.dot(vstack(p, 1))[:3]` assigned into a float32 array), has_transform, close-point radius (0: keep every point; the
 *              reference removes |x| < 1 AND |y| < 1 from PAST sweeps only), time value (time_lag / timestamp), batch index
 *   out        (n_raw, n_copy + 2) fp32 rows [b, x', y', z', c3 .. c(n_copy-1), time]; the first *n_out rows are valid, in input order
 */
size_t pnx_merge_sweeps_desc_bytes(void);
size_t pnx_merge_sweeps_workspace_bytes(int64_t n_raw);
int pnx_merge_sweeps(const float* raw, int64_t n_raw, int32_t raw_stride, int32_t n_copy, const void* seg_descs_dev, int32_t n_segments, float* out,
                     int32_t* n_out_dev, void* workspace, size_t workspace_bytes, pnx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Epilogue of the masked-dense backbone (the dense stand-in for det3d/models/utils/sparse_conv.py:16-63 with BatchNorm
 * folded into the conv):  out = [relu]( x + bias[c] [+ residual] ) * mask[site]   in one pass, bf16 NHWC.
 *   relu: 0 none, 1 as written, 2 = ( relu(x + bias[c]) + residual ) * mask  (det3d/models/utils/conv.py BasicBlock of the neck:
 *   the identity joins after block2's own ReLU)
 *   x, residual, out  (sites, channels) bf16 (residual may be NULL; out may alias x)
 *   bias              fp32[channels];  mask  uint8[sites] or NULL (= all active)
 */
int pnx_bias_act_mask(const void* x, const void* residual, const float* bias, const uint8_t* mask, void* out, int64_t sites,
                      int32_t channels, int32_t dtype, int32_t relu, pnx_stream_t stream);
/* ASPP neck with the 1x1 post_conv folded into the branches (det3d/models/necks/aspp.py:19-32: post_conv(cat(x, conv1x1(x),
 * conv_d(x, W) for d in 1,6,12,18)) == sum of six convolutions of x with post-multiplied weights): the partial results are summed
 * in fp32 in one pass,  out = [relu]( sum_k src_k + bias[c] ), NHWC in `dtype` (PNX_BF16 / PNX_F16).  srcs = HOST array of n_src (1..8)
 * device pointers. */
int pnx_sum_bias_act(const void* const* srcs, int32_t n_src, const float* bias, void* out, int64_t sites, int32_t channels, int32_t dtype,
                     int32_t relu, pnx_stream_t stream);
/* ConvTranspose2d(cin, cout, kernel 2, stride 2, no bias) + folded BatchNorm + [ReLU] in one kernel, bf16 NHWC, fp32 accumulation:
 * the deblock of a SepHead (det3d/models/heads/centerhead.py:17-21 via det3d/models/utils/conv.py ConvBlock with
 * conv_layer=ConvTranspose2d).  x (B,h,w,cin) -> y (B,2h,2w,cout); wfrag = weights in MFMA-fragment order
 * [ky*2+kx][cin/16][cout/32][lane][8] (pillarnext_amd/ops.py::deconv2x2_pack_weights); bias fp32[cout].  Kernels: 64 -> 64. */
int pnx_deconv2x2_bf16(const void* x, const void* wfrag, const float* bias, void* y, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                       int32_t relu, pnx_stream_t stream);
int pnx_deconv2x2_f16(const void* x, const void* wfrag, const float* bias, void* y, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                       int32_t relu, pnx_stream_t stream); /* IEEE-half twin */
/* Masked 3x3 convolution (pad 1, stride 1 or 2) with the same epilogue fused, bf16 NHWC, fp32 accumulation on MFMA:
 *   y = mask_out * [relu]( conv3x3(x, W) + bias [+ residual] ),  rows/tiles of the output without an active site are skipped.
 *   x (B,h,w,cin), y/residual (B,ho,wo,cout), mask uint8 (B,ho,wo) or NULL; wfrag = weights in MFMA-fragment order
 *   (pillarnext_amd/ops.py::conv3x3_pack_weights).  Built for (cin,cout) in {(64,64), (64,128), (128,128)} at stride 1|2 and {(64,320), (64,384), (64,448)} at stride 1.
 *   row_dirty (optional, with a mask): uint8 (B, ho, ceil(wo/32)), one flag per 32-pixel row segment of y -- "may hold non-zero
 *   data".  With it the kernel keeps y sparse-in-dense: segments without an active site are zero-filled only when their flag is
 *   set (then cleared), segments with active sites are written and flagged; an all-empty segment of a PERSISTENT output buffer
 *   costs no HBM traffic at all.  Contract: (y, row_dirty) start zeroed and y is only ever written through this function. */
int pnx_conv3x3_bf16(const void* x, const void* wfrag, const float* bias, const void* residual, const uint8_t* mask, void* y, int32_t batch,
                     int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride, int32_t relu, uint8_t* row_dirty, const int32_t* tile_list,
                     const int32_t* tile_count, pnx_stream_t stream);
/* The same kernels on IEEE half (fp16 activations and wfrag; BASELINE configs[4], waymo_det_pp18_aspp_iou_car_sp.yaml in fp16): identical
 * arguments and fp32 accumulation; only the MFMA opcode and the rounding of the stored result differ. */
int pnx_conv3x3_f16(const void* x, const void* wfrag, const float* bias, const void* residual, const uint8_t* mask, void* y, int32_t batch,
                    int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride, int32_t relu, uint8_t* row_dirty, const int32_t* tile_list,
                    const int32_t* tile_count, pnx_stream_t stream);
/* Weight gradient of the masked stride-1 3x3 convolution (training; det3d/models/utils/sparse_conv.py:16-63 under autograd: spconv accumulates
 * over the active output sites):  dw[co][ci][ky][kx] = sum over the sites p with mask[p] != 0 of dy[p][co] * x[stride*p + (ky-1, kx-1)][ci]
 *   stride 1 or 2 (pad 1); x (B,h,w,cin), dy (B,ho,wo,cout) bf16 NHWC with ho = (h-1)/stride + 1 (x zero at inactive sites, as every map of the masked-dense stand-in is), mask uint8 (B,ho,wo) of the
 *   OUTPUT sites, dw fp32 (cout, cin, 3, 3); cin, cout multiples of 64 up to 512.  Deterministic (static tile deal + a fixed-order reduction of
 *   per-workgroup partials in `workspace`, pnx_conv3x3_wgrad_workspace_bytes). */
size_t pnx_conv3x3_wgrad_workspace_bytes(int32_t cin, int32_t cout);
int pnx_conv3x3_wgrad_bf16(const void* x, const void* dy, const uint8_t* mask, float* dw, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout,
                           int32_t stride, void* workspace, size_t workspace_bytes, pnx_stream_t stream);
/* The fp32 graph's weight gradient from the bf16 halves of both operands (pnx_split_f32): x_hi dY_hi + x_lo dY_hi + x_hi dY_lo in one pass, four operand
 * tiles staged for the three products, fp32 accumulation, deterministic; shapes, mask and workspace as pnx_conv3x3_wgrad_bf16. */
int pnx_conv3x3_wgrad_x3(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, const uint8_t* mask, float* dw, int32_t batch, int32_t h,
                         int32_t w, int32_t cin, int32_t cout, int32_t stride, void* workspace, size_t workspace_bytes, pnx_stream_t stream);
/* (cout, cin, 3, 3) fp32 / bf16 weights -> wfrag of pnx_conv3x3_bf16 (9*cout*cin bf16) in one launch; transposed != 0: the weights of the data
 * gradient of a stride-1 layer, wt[ci][co][ky][kx] = w[co][ci][2-ky][2-kx], i.e. the wfrag of a cout -> cin convolution. */
int pnx_conv3x3_pack_weights(const void* w, int32_t dtype, int32_t cout, int32_t cin, int32_t transposed, void* wfrag, pnx_stream_t stream);
/* Optional tile list for the stride-1 kernels: the submanifold blocks of a backbone stage share one active-site mask, so the
 * tiles that need any work (an active site, or a stale row in one of the persistent output buffers) are listed once per stage
 * and every convolution of the stage walks the list instead of all tiles.
 *   pnx_conv3x3_tile_rows   rows of a tile of the kernel serving (cin, cout, stride); 0 = that kernel takes no list
 *   pnx_conv_tile_list      mask uint8 (batch,h,w); row_dirty = HOST array of n_dirty (0..4) device pointers (the row_dirty arrays of
 *                           the buffers the stage writes); tile_list int32[batch * ceil(h/tile_rows) * ceil(w/32)], tile_count int32[1]
 *                           (both device; the count is reset here).  Pass both to pnx_conv3x3_bf16 (or NULL, NULL). */
int pnx_conv3x3_tile_rows(int32_t cin, int32_t cout, int32_t stride);
int pnx_conv_tile_list(const uint8_t* mask, const uint8_t* const* row_dirty, int32_t n_dirty, int32_t batch, int32_t h, int32_t w, int32_t tile_rows,
                       int32_t* tile_list, int32_t* tile_count, pnx_stream_t stream);
/* Final convolution of the merged SepHead branches of one task (det3d/models/heads/centerhead.py:12-59, the last
 * Conv2d(64, k_j, 3, padding=1, bias=True) of every branch j):  y[b,oy,ox,o] = bias[o] + sum_{j,ky,kx,c} x[b,oy+ky-1,ox+kx-1,64j+c] * W[o][64j+c][ky][kx]
 * with W block diagonal (output o belongs to exactly one branch).  x (B,h,w,64*n_branch) bf16, y (B,h,w,16) bf16 (sum k_j <= 16,
 * unused outputs have zero weights), bias fp32[16], wfrag from pillarnext_amd/ops.py::sephead_pack_weights; n_branch in {1, 2, 5, 6, 7}
 * (1/2: the dense [iou] hm branches of the lazy head). */
int pnx_sephead_out_bf16(const void* x, const void* wfrag, const float* bias, void* y, int32_t batch, int32_t h, int32_t w, int32_t n_branch,
                         pnx_stream_t stream);
int pnx_sephead_out_f16(const void* x, const void* wfrag, const float* bias, void* y, int32_t batch, int32_t h, int32_t w, int32_t n_branch,
                         pnx_stream_t stream); /* IEEE-half twin */
/* Lazy SepHead: the five regression branches of every task (reg 2, height 1, dim 3, rot 2, vel 2: centerhead.py:12-59, conv3x3 64->64 + BN + ReLU,
 * then conv3x3 64->k) evaluated only at the candidate cells CenterHead.post_processing keeps (centerhead.py:341-363) instead of over the map, all
 * tasks in one launch.  Candidate lists: batch*nc_total lists (list s = sample s / nc_total, class s % nc_total, task class_task[class]) of pre_max
 * slots; local int64[n_lists*pre_max] = b*h*w + y*w + x inside the list's task map; slots j >= seg_len[s] are not evaluated (their rows are zeroed).
 * Per task: up (B,h,w,64) bf16 = the head's shared-conv output; wfrag1 = ops.conv3x3_pack_weights of the five first convolutions stacked
 * (320,64,3,3) with BN folded, bias1 fp32[320]; w2c fp32[10][9][32][3] = ops.sephead_lazy_pack_w2 (per 32-channel tile and 3x3 position: the <=3
 * output weights of the tile's branch, bf16-rounded); bias2 fp32[10].  out fp32 (n_lists*pre_max, 10), rounded to bf16 as the dense output is.
 * class_task is a HOST array of nc_total entries. */
typedef struct {
  const void* up;
  const void* wfrag1;
  const float* bias1;
  const float* w2c;
  const float* bias2;
  int32_t h, w;
} PnxLazyTask;
int pnx_sephead_lazy_bf16(const PnxLazyTask* tasks, int32_t n_tasks, const int32_t* class_task, int32_t nc_total, int32_t batch, const int64_t* local,
                          const int32_t* seg_len, int32_t pre_max, float* out, pnx_stream_t stream);
int pnx_sephead_lazy_f16(const PnxLazyTask* tasks, int32_t n_tasks, const int32_t* class_task, int32_t nc_total, int32_t batch, const int64_t* local,
                          const int32_t* seg_len, int32_t pre_max, float* out, pnx_stream_t stream); /* IEEE-half twin */
/* The fp32 training graph's SparseConv2d (reference: det3d/models/backbones/scn2d.py:12-47 under torch autograd in fp32, the reference's default
 * training precision, tools/train.py) on the bf16 matrix cores: every fp32 operand is split into two bf16 halves (pnx_split_f32: hi = RNE(x),
 * lo = RNE(x - hi), 16 mantissa bits together) and pnx_conv3x3_x3 accumulates x_hi W_hi + x_hi W_lo + x_lo W_hi in fp32 inside one launch.  x_hi, x_lo:
 * NHWC bf16; wfrag_hi, wfrag_lo: pnx_conv3x3_pack_weights of the two weight halves; y: fp32 NHWC (batch, ho, wo, cout), zeros at inactive sites, every
 * site written; bias: cout fp32 values added at the active sites, or NULL; no activation (the training graph's BatchNorm follows).  mask NULL: every site
 * is active (the dense head / neck layers: det3d/models/heads/centerhead.py:24-41, det3d/models/utils/conv.py).  Shapes: stride 1 64->64, 128->128, 256->256; stride 2 64->128,
 * 128->256, 256->256.  n of pnx_split_f32: a multiple of 8; mask (may be NULL): x is an NHWC map of n / channels sites that is zero at the sites where
 * mask is 0 -- those sites are not read, zeros are written (the maps of the sparse backbone: x at its active set, the gradient at the output's). */
int pnx_split_f32(const float* x, void* hi, void* lo, int64_t n, const uint8_t* mask, int32_t channels, pnx_stream_t stream);
int pnx_conv3x3_x3(const void* x_hi, const void* x_lo, const void* wfrag_hi, const void* wfrag_lo, const float* bias, const uint8_t* mask, float* y,
                   int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t stride, pnx_stream_t stream);
/* Data gradient of the backbone's stride-2 SparseConv2d layers in training (csrc/conv_dgrad_s2.h; reference: det3d/models/utils/sparse_conv.py:16-39 under
 * autograd): the four parity planes of dx as four small convolutions of the upstream gradient (1, 2, 2, 4 taps).  g: NHWC (batch, ho, wo, cout) bf16 with
 * ho = (h - 1) / 2 + 1; wfrag_t: pnx_conv3x3_pack_weights(transposed = 1) of the (cout, cin, 3, 3) weights; mask_in: uint8 (batch, h, w), the layer's INPUT active
 * set; dx: NHWC (batch, h, w, cin), zeros at inactive sites, every site written.  (cin, cout): (64, 128), (128, 256), (256, 256).  _x3: the fp32 graph's form
 * (bf16 halves of g and of the weights, three products, fp32 dx). */
int pnx_conv3x3_dgrad_s2_bf16(const void* g, const void* wfrag_t, const uint8_t* mask_in, void* dx, int32_t batch, int32_t h, int32_t w, int32_t cin,
                              int32_t cout, pnx_stream_t stream);
int pnx_conv3x3_dgrad_s2_x3(const void* g_hi, const void* g_lo, const void* wfrag_t_hi, const void* wfrag_t_lo, const uint8_t* mask_in, float* dx, int32_t batch,
                            int32_t h, int32_t w, int32_t cin, int32_t cout, pnx_stream_t stream);
/* The SepHead's output convolutions in training (csrc/head_train.hip; reference: det3d/models/heads/centerhead.py:31-41, nn.Conv2d(64, k, 3, padding 1,
 * bias=True) with k = classes / 2 / 1 / 3 / 2 / 2, under autograd): bandwidth-bound layers MIOpen runs at 1/7 of the HBM rate.
 *   pnx_conv3x3_smallk        y = conv3x3(x, weight) + bias: x NHWC (batch, h, w, 64), y NHWC (batch, h, w, k), both `dtype` (PNX_F32 or PNX_BF16);
 *                             weight fp32 (k, 64, 3, 3) as the module holds it, bias fp32 (k) or NULL; fp32 accumulation
 *   pnx_conv3x3_smallk_wgrad  dw (k, 64, 3, 3) and dbias (k; may be NULL) fp32 from x and the upstream gradient dy (NHWC, k channels), deterministic;
 *                             workspace: pnx_conv3x3_smallk_wgrad_workspace_bytes(k)
 * cin must be 64, 1 <= k <= 4. */
int pnx_conv3x3_smallk(const void* x, const float* weight, const float* bias, void* y, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t k,
                       int32_t dtype, pnx_stream_t stream);
size_t pnx_conv3x3_smallk_wgrad_workspace_bytes(int32_t k);
int pnx_conv3x3_smallk_wgrad(const void* x, const void* dy, float* dw, float* dbias, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t k,
                             int32_t dtype, void* workspace, size_t workspace_bytes, pnx_stream_t stream);
/* Active-site rule of SparseConv2d(k=3, stride, pad=1): mask_out = maxpool3x3(mask_in, stride, 1); uint8 (B,h,w) -> (B,ho,wo). */
int pnx_mask_pool3(const uint8_t* mask_in, int32_t batch, int32_t h, int32_t w, int32_t stride, uint8_t* mask_out, pnx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Rotated IoU / NMS.  Boxes are (n,7) fp32 [x, y, z, dx, dy, dz, heading].
 * Arithmetic = iou3d_nms_kernel.cu:35-234 in fp32, with cos/sin/atan2 from pnx_detmath.h.
 */
/* boxes_overlap_bev_gpu (iou3d_nms.cpp:65-85): out (n,m) overlap areas */
int pnx_boxes_overlap_bev(const float* boxes_a, int64_t n, const float* boxes_b, int64_t m, float* out, pnx_stream_t stream);
/* boxes_iou_bev_gpu (:87-106): out (n,m) IoU */
int pnx_boxes_iou_bev(const float* boxes_a, int64_t n, const float* boxes_b, int64_t m, float* out, pnx_stream_t stream);
/* boxes_aligned_overlap_bev_gpu (:40-63): out (n) overlap area of pair i */
int pnx_boxes_aligned_overlap_bev(const float* boxes_a, const float* boxes_b, int64_t n, float* out, pnx_stream_t stream);
/* boxes_aligned_iou3d_gpu (iou3d_nms_utils.py:49-89): out (n) 3-D IoU of pair i, fused */
int pnx_boxes_aligned_iou3d(const float* boxes_a, const float* boxes_b, int64_t n, float* out, pnx_stream_t stream);

/* Host twins of pnx_boxes_iou_bev / the aligned pair form: det3d/core/iou3d_nms/src/iou3d_cpu.cpp:232-273 (boxes_iou_bev_cpu (N,M),
 * boxes_aligned_iou_bev_cpu (N,1)) -- HOST pointers, no GPU needed, synchronous.  Compiled from the same source as the device kernels
 * (csrc/iou3d_geom.h) with the same flags: results equal pnx_boxes_iou_bev's bit for bit (and the reference's libm-based values to 1e-5). */
int pnx_boxes_iou_bev_cpu(const float* boxes_a_host, int64_t n, const float* boxes_b_host, int64_t m, float* out_host);
int pnx_boxes_aligned_iou_bev_cpu(const float* boxes_a_host, const float* boxes_b_host, int64_t n, float* out_host);
/* nms_gpu / nms_normal_gpu (iou3d_nms.cpp:113-159 / :162-211) for `num_segments` independent,
 * already score-sorted box lists laid end to end:  segment s = boxes[seg_offsets[s] .. seg_offsets[s+1]).
 * The bitmask AND the greedy scan run on the device (the reference copies the mask to the host).
 *   seg_offsets   int32[num_segments+1] (device)
 *   seg_len       int32[num_segments] (device) or NULL: only the first seg_len[s] boxes of segment s take part
 *                 (the `order[:pre_maxsize]` cut of box_torch_ops.py:14-15 without compacting the box list)
 *   thresh        fp32[num_segments] (device) IoU threshold per segment
 *   keep          int32, same length as boxes: keep[seg_offsets[s] + k] = index (within the segment) of the
 *                 k-th kept box, ascending -- exactly what nms_gpu writes into `keep`
 *   keep_count    int32[num_segments]: number kept (already min'ed with post_max if post_max > 0)
 *   max_seg_len   an upper bound on any segment's length (host value; sizes the launch)
 */
size_t pnx_nms_workspace_bytes(int64_t total_boxes, int32_t num_segments, int32_t max_seg_len);
/* Test hook: capacity (entries) of the rotated NMS's candidate-pair list, 0 = the default (32 per box); returns the previous value.  A small list sends
 * the tiles that do not fit down the tile-by-tile path, whose keep indices must be identical.  Changes the layout pnx_nms_workspace_bytes describes:
 * query the workspace size again after calling it.  Not thread-safe. */
int32_t pnx_debug_nms_pair_cap(int32_t cap);
int pnx_nms_rotated_batched(const float* boxes, const int32_t* seg_offsets, const int32_t* seg_len, int32_t num_segments, int32_t max_seg_len,
                            const float* thresh, int32_t post_max, int32_t* keep, int32_t* keep_count, void* workspace,
                            size_t workspace_bytes, pnx_stream_t stream);
int pnx_nms_normal_batched(const float* boxes, const int32_t* seg_offsets, const int32_t* seg_len, int32_t num_segments, int32_t max_seg_len,
                           const float* thresh, int32_t post_max, int32_t* keep, int32_t* keep_count, void* workspace,
                           size_t workspace_bytes, pnx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused CenterHead decode (det3d/models/heads/centerhead.py:231-363), feeding pnx_nms_rotated_batched.
 * A task's head output is ONE packed NHWC tensor (B, H, W, C) fp32 or bf16 with channels
 * [reg 2 | height 1 | dim 3 | rot 2 | vel 2 | (iou 1) | hm ncls | padding]; a task is described by a host-side
 * descriptor of pnx_decode_task_desc_bytes() bytes (layout: pillarnext_amd/decode.py::pack_task).
 *   pnx_decode_keys   keys[b*H*W + cell] = (segment << 32) | ~bits(rectified score), all-ones if masked out
 *                     (score <= thr or centre outside post_center_limit_range); segment = b*n_classes_total + class
 *   pnx_decode_boxes  after a stable ascending sort of all tasks' keys (`order` = source index of every sorted key,
 *                     `seg_start`/`seg_len` per segment): 9-d boxes [x,y,z,dx,dy,dz,vx,vy,rot], NMS boxes
 *                     [x,y,z,dx,dy,dz,rot] and scores for the first pre_max candidates of each segment, row = s*pre_max + j
 *   pnx_gather_kept   out[s][j] = [box9 | score] of the j-th kept candidate (j < min(keep_count[s], post_max)), else zeros
 */
size_t pnx_decode_task_desc_bytes(void);
/* Segmented top-k over the keys of all tasks (box_torch_ops.py:13-15 for every (sample, class) list at once): the first pre_max
 * candidates of each segment in descending-score order, ties in ascending key index (= what a stable sort of `keys` would give), without
 * sorting the key array: an exact radix select on the composite (score key, key index), most significant digit first -- per pass one
 * histogram launch (LDS histograms per 16 384-key chunk, no global atomics) and one per-segment digit selection; segments finish early when
 * they hold fewer than pre_max valid keys or a digit bucket is wanted completely; then one collect pass and a per-segment sort of the
 * <= pre_max survivors.  pre_max <= 4096.  Outputs are laid out as pnx_decode_boxes expects them: row s*pre_max + j = j-th candidate of
 * segment s, seg_start[s] = s*pre_max, seg_len[s] <= pre_max; seg_total (optional) = number of valid keys of the segment. */
size_t pnx_decode_topk_workspace_bytes(int64_t n_keys, int32_t num_segments);
int pnx_decode_topk(const uint64_t* keys, int64_t n_keys, int32_t num_segments, int32_t pre_max, uint64_t* sorted_keys, int64_t* order,
                    int64_t* seg_start, int32_t* seg_len, int32_t* seg_total, void* workspace, size_t workspace_bytes, pnx_stream_t stream);
int pnx_decode_keys(const void* packed, int32_t dtype, int32_t batch, int32_t n_classes_total, const void* task_desc_host, uint64_t* keys,
                    pnx_stream_t stream);
/* Stable sort of the candidate keys of pnx_decode_keys (all tasks, all samples) with their positions as payload -- what
 * centerhead.py:296-300 does per class with torch.sort/topk, here once for every (sample, class) list: keys are
 * (segment << 32 | ~score bits), invalid = all ones, so only 32 + bit_length(num_segments) bits are sorted.
 *   sorted_keys uint64[n_keys], order int64[n_keys] (position of each sorted key in `keys`); workspace from ..._workspace_bytes. */
size_t pnx_sort_keys_workspace_bytes(int64_t n_keys);
int pnx_sort_keys(const uint64_t* keys, int64_t n_keys, int32_t num_segments, uint64_t* sorted_keys, int64_t* order, void* workspace,
                  size_t workspace_bytes, pnx_stream_t stream);
int pnx_decode_boxes(const void* const* task_ptrs_dev, const void* task_descs_dev, const int64_t* task_key_off_dev, int32_t n_tasks,
                     int32_t dtype, int32_t batch, const uint64_t* sorted_keys, const int64_t* order, const int64_t* seg_start,
                     const int32_t* seg_len, int32_t num_segments, int32_t pre_max, float* boxes9, float* boxes7, float* scores,
                     pnx_stream_t stream);
/* Lazy head (pillarnext_amd/models.py FusedPillarNeXt, PNX_HEAD_LAZY): only the class (and iou) branches of a SepHead are computed for
 * every cell; the regression branches (reg, height, dim, rot, vel) are evaluated at the selected candidates only.  pnx_decode_keys then
 * runs on the [iou] hm packing (task descriptor: o_hm, o_iou, lazy = 1: no centre / range test), the sort selects the candidates, and
 * pnx_decode_boxes_lazy decodes them from cand = (num_segments * pre_max, 10) fp32 [reg 2, height 1, dim 3, rot 2, vel 2], applies the
 * range test of centerhead.py:343-346 and compacts the survivors (seg_len updated in place).  seg_total = candidates per segment before
 * the pre_max cut; *flag_dev |= 1 when a cut segment lost a candidate to the range test (the reference would have admitted the next
 * one: the caller re-runs that batch through the dense path). */
int pnx_decode_boxes_lazy(const void* task_descs_dev, const int64_t* task_key_off_dev, int32_t n_tasks, int32_t n_classes_total,
                          const uint64_t* sorted_keys, const int64_t* order, const int64_t* seg_start, int32_t* seg_len, const int32_t* seg_total,
                          int32_t num_segments, int32_t pre_max, const float* cand, float* boxes9, float* boxes7, float* scores, int32_t* flag_dev,
                          pnx_stream_t stream);
int pnx_gather_kept(const float* boxes9, const float* scores, const int32_t* keep, const int32_t* keep_count, int32_t num_segments,
                    int32_t pre_max, int32_t post_max, float* out, pnx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Train-mode BatchNorm over the ACTIVE sites of a dense NHWC map (the reference: BatchNorm1d on the features of a SparseConvTensor,
 * det3d/models/utils/sparse_conv.py:31-37,57-60) fused with the residual add, ReLU and the active-site mask, forward and backward.
 * x, residual, y, gy, dx, dresidual: (n_sites, channels) bf16 or fp32 (channels_last maps); mask fp32[n_sites], 0 = inactive (never read,
 * written as zeros), or NULL = every site active (a dense BatchNorm2d: no mask word in front of a site's loads); channels in {8,16,32,64,128,256}.
 *   stats      partials fp32 [pnx_masked_bn_blocks()][2*channels + 1]: per workgroup sum d | sum d^2 | active sites with d = x - center[c]
 *              (center: fp32[channels] or NULL = 0; the running mean keeps the one-pass variance sum d^2 / n - (sum d / n)^2 from
 *              cancelling when |mean| >> std; every rank of a SyncBatchNorm group must pass the same values) -- the caller adds the rows
 *              (fp64), all-reduces them under SyncBatchNorm (tools/train.py:56), forms mean = center + sum d / n, invstd and
 *              scale = gamma * invstd, shift = beta - mean * scale
 *   apply      y = [relu](x * scale + shift [+ residual]) at the active sites
 *   bwd_stats  partials fp32 [blocks][2*channels]: sum g | sum g * xhat with g = gy * [pre-activation > 0], xhat = (x - mean) * invstd
 *   bwd_apply  dx = scale * (g - mean_g - xhat * mean_gx), dresidual (optional) = g; mean_g = sum g / count, mean_gx = sum g*xhat / count
 * The per-channel arithmetic between those passes, as three small launches instead of ~30 host tensor statements per layer:
 *   reduce        sums fp64[cols] = the rows of a partials array added in a fixed order (all-reduce `sums` under SyncBatchNorm before the next call)
 *   finalize      from sums = [sum d | sum d^2 | count]: mean, invstd, scale, shift, count (fp32 vectors for apply / the backward) and torch's running-statistics
 *                 update in place (running_mean / running_var fp32, both or neither; unbiased variance; num_batches_tracked += 1 when not NULL); center may
 *                 alias running_mean
 *   bwd_finalize  dgamma / dbeta from the LOCAL sums [sum g | sum g xhat], mean_g / mean_gx from the global ones (NULL: the local ones) */
int32_t pnx_masked_bn_blocks(void);
int pnx_masked_bn_reduce(const float* partials, int32_t rows, int32_t cols, double* sums, pnx_stream_t stream);
int pnx_masked_bn_finalize(const double* sums, int32_t channels, const float* center, const float* weight, const float* bias, double eps, double momentum,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                           float* count, pnx_stream_t stream);
int pnx_masked_bn_bwd_finalize(const double* sums_local, const double* sums_global, int32_t channels, const float* count, float* dgamma, float* dbeta,
                               float* mean_g, float* mean_gx, pnx_stream_t stream);
int pnx_masked_bn_stats(const void* x, int32_t dtype, const float* mask, int64_t n_sites, int32_t channels, const float* center, float* partials,
                        pnx_stream_t stream);
int pnx_masked_bn_apply(const void* x, const void* residual, int32_t dtype, const float* mask, int64_t n_sites, int32_t channels, const float* scale,
                        const float* shift, int32_t relu, void* y, pnx_stream_t stream);
int pnx_masked_bn_bwd_stats(const void* gy, const void* x, const void* residual, int32_t dtype, const float* mask, int64_t n_sites, int32_t channels,
                            const float* scale, const float* shift, const float* mean, const float* invstd, int32_t relu, float* partials,
                            pnx_stream_t stream);
int pnx_masked_bn_bwd_apply(const void* gy, const void* x, const void* residual, int32_t dtype, const float* mask, int64_t n_sites, int32_t channels,
                            const float* scale, const float* shift, const float* mean, const float* invstd, int32_t relu, const float* mean_g,
                            const float* mean_gx, void* dx, void* dresidual, pnx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * CenterHead training losses over the (B, M) label lists of ONE task, forward and backward (csrc/center_loss.hip):
 * det3d/models/loss/centerloss.py FastFocalLoss :8-37, RegLoss :40-60 (NaN-target rule :55-56), IouLoss :63-87 (target 2*IoU3D-1 from the
 * aligned rotated IoU), IouRegLoss :90-110 + DIoU :139-176, as combined by det3d/models/heads/centerhead.py:142-229; labels in the format
 * of det3d/datasets/pipelines/assign.py:113-114.
 *   maps7      HOST array of 7 device pointers, fp32 (B,C,H,W): hm (C = n_classes, LOGITS), reg (2), height (1), dim (3), rot (2), vel (2),
 *              iou (1) or NULL (no IoU head)
 *   hm_target  (B,n_classes,H,W) fp32;  ind, cat (B,M) int64;  mask (B,M) uint8;  anno_box (B,M,10) fp32 in the order reg height dim vel rot
 *              (NaN = "no target");  gt_boxes (B,M,7)
 *   geom4_host {out_size_factor*voxel_x, out_size_factor*voxel_y, pc_min_x, pc_min_y}
 *   losses15   device out: [0] hm_loss, [1..10] regression loss per box-code element (before code weights), [11] iou_loss,
 *              [12] iou_reg_loss (0 unless with_reg_iou), [13] number of positives, [14] internal
 * backward: grads7 = gradient buffers of the seven maps (hm is written densely; the other six must be zero on entry, the listed cells
 * are accumulated); upstream13 = device array, gradients of losses15[0..12]; workspace = the forward call's, untouched in between. */
size_t pnx_center_loss_workspace_bytes(int32_t batch, int32_t max_objs);
int pnx_center_loss_forward(const void* const* maps7, const float* hm_target, const int64_t* ind, const uint8_t* mask, const int64_t* cat,
                            const float* anno_box, const float* gt_boxes, int32_t batch, int32_t n_classes, int32_t h, int32_t w, int32_t max_objs,
                            const float* geom4_host, int32_t with_reg_iou, float* losses15, void* workspace, size_t workspace_bytes,
                            pnx_stream_t stream);
int pnx_center_loss_backward(const void* const* maps7, void* const* grads7, const float* hm_target, const int64_t* ind, const uint8_t* mask,
                             const int64_t* cat, const float* anno_box, const float* gt_boxes, int32_t batch, int32_t n_classes, int32_t h, int32_t w,
                             int32_t max_objs, const float* geom4_host, int32_t with_reg_iou, const float* losses15, const float* upstream13,
                             float* coef_scratch, void* workspace, size_t workspace_bytes, pnx_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * One-call enqueue of prebuilt launch tables (csrc/enqueue.hip).  The reference's step is a Python call tree
 * (det3d/models/detectors/single_stage.py:22-33, then CenterHead.predict centerhead.py:231-384); the arguments of this library's calls for
 * the backbone, the head and the decoder do not change between frame batches (persistent workspaces, weights, shapes), so a host
 * integration freezes them once and replays the table per step -- the launch thread then paces the GPU from C, not from an interpreter.
 * An entry forwards to the call it names: i[] = that call's integer arguments, p[] = its pointer arguments, both in signature order:
 *   PNX_OP_MASK_POOL3   pnx_mask_pool3        p: mask_in, mask_out                     i: batch, h, w, stride
 *   PNX_OP_TILE_LIST    pnx_conv_tile_list    p: mask, tile_list, tile_count, row_dirty[0..n_dirty)   i: n_dirty (<= 8), batch, h, w, tile_rows
 *   PNX_OP_CONV3X3      pnx_conv3x3_bf16      p: x, wfrag, bias, residual, mask, y, row_dirty, tile_list, tile_count
 *                                             i: batch, h, w, cin, cout, stride, relu, dtype
 *   PNX_OP_DECONV2X2    pnx_deconv2x2_bf16    p: x, wfrag, bias, y                     i: batch, h, w, cin, cout, relu, dtype
 *   PNX_OP_SEPHEAD_OUT  pnx_sephead_out_bf16  p: x, wfrag, bias, y                     i: batch, h, w, n_branch, dtype
 * dtype (the last integer of the three convolution entries): PNX_F16 selects the pnx_*_f16 twin, anything else (0 in tables built before the
 * twins existed) the bf16 entry point.
 * The table is HOST memory and is read during the call only.  On failure the status of the failing entry is returned and pnx_last_error()
 * names its index. */
enum { PNX_OP_MASK_POOL3 = 1, PNX_OP_TILE_LIST = 2, PNX_OP_CONV3X3 = 3, PNX_OP_DECONV2X2 = 4, PNX_OP_SEPHEAD_OUT = 5 };
typedef struct pnx_op {
  int32_t kind;
  int32_t i[9];
  const void* p[11];
} pnx_op; /* 128 bytes */
int pnx_enqueue(const pnx_op* ops_host, int32_t n_ops, pnx_stream_t stream);
size_t pnx_op_bytes(void); /* sizeof(pnx_op), for bindings to check their mirror of the structure */

/* The lazy-head decoder of a frame batch as ONE call: pnx_decode_keys per task -> pnx_decode_topk -> candidate cells -> pnx_sephead_lazy_bf16 ->
 * pnx_decode_boxes_lazy -> pnx_nms_rotated_batched -> pnx_gather_kept [-> asynchronous copies into pinned host memory]; the clears that the
 * individual calls expect of their caller (order, boxes7, flag, keep_count) are part of it.  S = batch * n_classes_total lists.
 * Buffers (device, caller-owned): keys/sorted_keys uint64[n_keys] and [S*pre_max], order/local int64[S*pre_max], seg_start int64[S],
 * seg_len/seg_total/keep_count int32[S], cand float[S*pre_max*10], boxes9/boxes7/scores float[S*pre_max*{9,7,1}], flag int32[1],
 * keep int32[S*pre_max], out float[S*post_max*10]; workspaces sized by pnx_decode_topk_workspace_bytes / pnx_nms_workspace_bytes.
 * flag != 0 after the call: the centre range test rejected a candidate, the selection has to be redone on the dense path (decode.py). */
typedef struct pnx_lazy_decode {
  int32_t n_tasks, n_classes_total, batch, pre_max, post_max, dtype;
  const void* const* dense_host;     /* [n_tasks] device pointers: (batch, h_t, w_t, 16) maps holding [iou] hm */
  const void* task_descs_host;       /* n_tasks * pnx_decode_task_desc_bytes() */
  const void* task_descs_dev;
  const int64_t* task_key_off_host;  /* [n_tasks + 1]: first key of every task, then n_keys */
  const int64_t* task_key_off_dev;
  const int64_t* list_key_off_dev;   /* [S]: task_key_off of the list's task */
  const PnxLazyTask* lazy_tasks_host;
  const int32_t* class_task_host;    /* [n_classes_total] */
  const int32_t* seg_off_dev;        /* [S + 1] = s * pre_max */
  const float* nms_thresh_dev;       /* [S] */
  uint64_t *keys, *sorted_keys;
  int64_t *order, *seg_start, *local;
  int32_t *seg_len, *seg_total;
  float *cand, *boxes9, *boxes7, *scores;
  int32_t *flag, *keep, *keep_count;
  void* topk_ws;
  size_t topk_ws_bytes;
  void* nms_ws;
  size_t nms_ws_bytes;
  float* out;
  float* out_host;                   /* optional pinned copies */
  int32_t* keep_count_host;
  int32_t* flag_host;
} pnx_lazy_decode;
int pnx_decode_lazy_enqueue(const pnx_lazy_decode* desc_host, pnx_stream_t stream);
size_t pnx_lazy_decode_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* PNX_H */
