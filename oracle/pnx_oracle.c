/* pnx_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU oracle for the PillarNeXt hot path.
 *
 * A plain-C restatement of what the reference computes on the path named by BASELINE.json's
 * north_star: dynamic pillarisation + PFN + scatter (reader) and rotated
 * IoU / NMS.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (pillarnext_amd/) never does.
 *
 * Pinning (see DESIGN.md "Oracle"):
 *   - the *_libm IoU/NMS functions are checked bit-for-bit against the reference's own
 *     iou3d_cpu.cpp compiled in place (oracle/_ref/libref_iou3d.so, built by oracle/Makefile);
 *   - the reader functions are checked against golden vectors produced by importing the
 *     reference's det3d/models/readers/pillar_encoder.py (tests/golden/, oracle/gen_golden.py).
 *
 * Line references are to files under /root/reference (never copied, never needed at run time):
 *   pe:  det3d/models/readers/pillar_encoder.py
 *   cu:  det3d/core/iou3d_nms/src/iou3d_nms_kernel.cu     cpu: .../src/iou3d_cpu.cpp
 *   nms: det3d/core/iou3d_nms/src/iou3d_nms.cpp
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PNX_HD static inline
#include "../pillarnext_amd/csrc/pnx_detmath.h"

typedef struct {
  float x, y;
} orc_pt;

#define ORC_EPS 1e-8f
/* cpu:29-35 define min/max as plain conditionals */
static inline float orc_minf(float a, float b) { return a > b ? b : a; }
static inline float orc_maxf(float a, float b) { return a > b ? a : b; }

/* Greedy suppression over the bitmask  (nms:139-155): i ascending; keep i iff its bit is not yet
 * set in remv; then OR row i into remv for words j >= i/64. */
static int64_t orc_greedy_scan(const uint64_t* mask, int64_t n, int64_t* keep) {
  int64_t cb = (n + 63) / 64;
  uint64_t* remv = (uint64_t*)calloc((size_t)cb + 1, sizeof(uint64_t));
  int64_t nk = 0;
  for (int64_t i = 0; i < n; i++) {
    int64_t nb = i / 64, ib = i % 64;
    if (!(remv[nb] & (1ULL << ib))) {
      keep[nk++] = i;
      const uint64_t* p = mask + i * cb;
      for (int64_t j = nb; j < cb; j++) remv[j] |= p[j];
    }
  }
  free(remv);
  return nk;
}

/* ---- variant 1: host libm (bit-identical to the compiled reference) ---- */
static inline void orc_sincos_libm(float a, float* s, float* c) {
  *c = cosf(a);
  *s = sinf(a);
}
#define ORC_SUFFIX _libm
#define ORC_SINCOS(a, s, c) orc_sincos_libm((a), (s), (c))
#define ORC_ATAN2(y, x) atan2f((y), (x))
#include "orc_iou_impl.h"
#undef ORC_SUFFIX
#undef ORC_SINCOS
#undef ORC_ATAN2

/* ---- variant 2: deterministic math shared with the HIP kernels ---- */
#define ORC_SUFFIX _det
#define ORC_SINCOS(a, s, c) pnx_sincosf((a), (s), (c))
#define ORC_ATAN2(y, x) pnx_atan2f((y), (x))
#include "orc_iou_impl.h"
#undef ORC_SUFFIX
#undef ORC_SINCOS
#undef ORC_ATAN2

/* Axis-aligned IoU (iou_normal cu:327-338) and its NMS (nms_normal_kernel cu:341-385). */
static float orc_iou_normal1(const float* a, const float* b) {
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  float interS = width * height;
  float Sa = a[3] * a[4];
  float Sb = b[3] * b[4];
  return interS / fmaxf(Sa + Sb - interS, ORC_EPS);
}
int64_t orc_nms_normal(const float* boxes, int64_t n, float thresh, int64_t* keep) {
  int64_t cb = (n + 63) / 64;
  uint64_t* mask = (uint64_t*)calloc((size_t)(n * cb + 1), sizeof(uint64_t));
  for (int64_t i = 0; i < n; i++)
    for (int64_t c = 0; c < cb; c++) {
      int64_t col_size = n - c * 64 < 64 ? n - c * 64 : 64;
      int64_t start = (i / 64 == c) ? (i % 64) + 1 : 0;
      uint64_t t = 0;
      for (int64_t k = start; k < col_size; k++)
        if (orc_iou_normal1(boxes + i * 7, boxes + (c * 64 + k) * 7) > thresh) t |= 1ULL << k;
      mask[i * cb + c] = t;
    }
  int64_t nk = orc_greedy_scan(mask, n, keep);
  free(mask);
  return nk;
}

/* 3-D IoU of aligned pairs: BEV overlap x height overlap / union volume, clamp 1e-6
 * (boxes_aligned_iou3d_gpu, det3d/core/iou3d_nms/iou3d_nms_utils.py:49-89).  det != 0 selects the
 * deterministic-math BEV overlap. */
void orc_boxes_aligned_iou3d(const float* a, const float* b, int64_t n, int det, float* out) {
  for (int64_t i = 0; i < n; i++) {
    const float *p = a + i * 7, *q = b + i * 7;
    float a_max = p[2] + p[5] / 2, a_min = p[2] - p[5] / 2;
    float b_max = q[2] + q[5] / 2, b_min = q[2] - q[5] / 2;
    float bev = det ? orc_box_overlap_det(p, q) : orc_box_overlap_libm(p, q);
    float max_of_min = a_min > b_min ? a_min : b_min;
    float min_of_max = a_max < b_max ? a_max : b_max;
    float h = min_of_max - max_of_min;
    if (h < 0.f) h = 0.f;
    float ov = bev * h;
    float va = p[3] * p[4] * p[5], vb = q[3] * q[4] * q[5];
    float den = va + vb - ov;
    if (den < 1e-6f) den = 1e-6f;
    out[i] = ov / den;
  }
}

/* ======================================================================================
 * Reader: PillarNet.forward (pe:78-125), PFNLayer.forward (pe:35-50),
 * PillarFeatureNet.forward (pe:174-182).
 * ====================================================================================== */

/* grid = np.round((pc_range[3:] - pc_range[:3]) / voxel_size)  in fp64, half-to-even  (pe:87-89) */
void orc_grid_size(const double* pc_range, const double* voxel, int64_t* grid3) {
  for (int i = 0; i < 3; i++) grid3[i] = (int64_t)nearbyint((pc_range[3 + i] - pc_range[i]) / voxel[i]);
}

typedef struct {
  int64_t key;
  int64_t idx;
} orc_ki;
static int orc_ki_cmp(const void* a, const void* b) {
  const orc_ki *p = (const orc_ki*)a, *q = (const orc_ki*)b;
  if (p->key != q->key) return p->key < q->key ? -1 : 1;
  return p->idx < q->idx ? -1 : (p->idx > q->idx);
}

/* Dynamic pillarisation.  pts: n rows of `stride` floats [b, x, y, z, f...].  pc_min/vs are the
 * fp32 casts of the fp64 config (pe:91-93).  Outputs (caller allocates for n rows):
 *   kept[n']   original row index of every kept point, original order      (mask compaction pe:103)
 *   inv[n']    pillar rank of each kept point = torch.unique(dim=0) inverse (pe:110)
 *   coords[P][3] int32 [b, yi, xi]                                          (pe:111,125)
 * returns n', writes P.  Pillar order = lexicographic [b, xi, yi] = ascending key. */
int64_t orc_voxelize(const float* pts, int64_t n, int stride, const float* pc_min, const float* vs, int64_t gx,
                     int64_t gy, int64_t* kept, int64_t* inv, int32_t* coords, int64_t* P_out) {
  orc_ki* ki = (orc_ki*)malloc(sizeof(orc_ki) * (size_t)(n + 1));
  int64_t m = 0;
  for (int64_t i = 0; i < n; i++) {
    const float* p = pts + i * stride;
    float cx = (p[1] - pc_min[0]) / vs[0]; /* fp32 subtract then true divide  (pe:95-96) */
    float cy = (p[2] - pc_min[1]) / vs[1];
    /* float compares against the integer grid size; NaN drops, -0.0 stays  (pe:98-101) */
    if (!(cx >= 0.f && cx < (float)gx && cy >= 0.f && cy < (float)gy)) continue;
    int64_t xi = (int64_t)cx, yi = (int64_t)cy, bi = (int64_t)p[0]; /* .long() truncation (pe:106-107) */
    ki[m].key = (bi * gx + xi) * gy + yi;
    ki[m].idx = m; /* position in the compacted list */
    kept[m] = i;
    m++;
  }
  qsort(ki, (size_t)m, sizeof(orc_ki), orc_ki_cmp);
  int64_t P = 0;
  for (int64_t j = 0; j < m; j++) {
    if (j == 0 || ki[j].key != ki[j - 1].key) {
      int64_t key = ki[j].key;
      int64_t yi = key % gy, t = key / gy, xi = t % gx, bi = t / gx;
      coords[P * 3 + 0] = (int32_t)bi;
      coords[P * 3 + 1] = (int32_t)yi;
      coords[P * 3 + 2] = (int32_t)xi;
      P++;
    }
    inv[ki[j].idx] = P - 1;
  }
  free(ki);
  *P_out = P;
  return m;
}

/* Point decoration (pe:113-123): per-pillar mean of xyz (sum in original point order, then true
 * divide by the count -- torch_scatter.scatter_mean semantics, third-party, restated), cluster
 * offset, pillar-centre offset, concat -> feat[n'][F+5] with F = stride-1. */
void orc_decorate(const float* pts, int stride, const int64_t* kept, const int64_t* inv, int64_t m, int64_t P,
                  const float* pc_min, const float* vs, float* feat) {
  int F = stride - 1, C = F + 5;
  float* sum = (float*)calloc((size_t)P * 3 + 3, sizeof(float));
  float* cnt = (float*)calloc((size_t)P + 1, sizeof(float));
  for (int64_t j = 0; j < m; j++) {
    const float* p = pts + kept[j] * stride;
    int64_t r = inv[j];
    sum[r * 3 + 0] += p[1];
    sum[r * 3 + 1] += p[2];
    sum[r * 3 + 2] += p[3];
    cnt[r] += 1.f;
  }
  for (int64_t r = 0; r < P; r++)
    for (int k = 0; k < 3; k++) sum[r * 3 + k] = sum[r * 3 + k] / cnt[r];
  float half_x = vs[0] / 2, half_y = vs[1] / 2;
  for (int64_t j = 0; j < m; j++) {
    const float* p = pts + kept[j] * stride;
    float* f = feat + j * C;
    int64_t r = inv[j];
    for (int k = 0; k < F; k++) f[k] = p[1 + k];
    f[F + 0] = p[1] - sum[r * 3 + 0];
    f[F + 1] = p[2] - sum[r * 3 + 1];
    f[F + 2] = p[3] - sum[r * 3 + 2];
    float cx = (p[1] - pc_min[0]) / vs[0];
    float cy = (p[2] - pc_min[1]) / vs[1];
    float xi = (float)(int64_t)cx, yi = (float)(int64_t)cy;
    /* idx*vs + vs/2 + min, each step rounded to fp32, no FMA  (pe:119-120) */
    float ctr_x = xi * vs[0];
    ctr_x = ctr_x + half_x;
    ctr_x = ctr_x + pc_min[0];
    float ctr_y = yi * vs[1];
    ctr_y = ctr_y + half_y;
    ctr_y = ctr_y + pc_min[1];
    f[F + 3] = p[1] - ctr_x;
    f[F + 4] = p[2] - ctr_y;
  }
  free(sum);
  free(cnt);
}

/* One PFN layer in eval mode (pe:35-50): Linear(no bias) -> BatchNorm1d(running stats, eps) ->
 * ReLU -> per-pillar max -> (last ? nothing : concat[x, max[inv]]).
 * params = [W (units x cin) | gamma | beta | running_mean | running_var] (each `units` long).
 * x_out: m x units activations; gmax: P x units pillar maxima. */
static void orc_pfn_layer_eval(const float* in, int64_t m, int cin, int units, const float* params, float eps,
                               const int64_t* inv, int64_t P, float* x_out, float* gmax) {
  const float* W = params;
  const float* gamma = W + (size_t)units * cin;
  const float* beta = gamma + units;
  const float* mean = beta + units;
  const float* var = mean + units;
  float* alpha = (float*)malloc(sizeof(float) * 2 * (size_t)units);
  float* shift = alpha + units;
  for (int c = 0; c < units; c++) { /* eval BN as scale/shift: y = x*alpha + (beta - mean*alpha) */
    float invstd = 1.f / sqrtf(var[c] + eps);
    alpha[c] = invstd * gamma[c];
    shift[c] = beta[c] - mean[c] * alpha[c];
  }
  for (int64_t r = 0; r < P * units; r++) gmax[r] = -INFINITY;
  for (int64_t j = 0; j < m; j++) {
    const float* f = in + j * cin;
    float* xo = x_out + j * units;
    float* g = gmax + inv[j] * units;
    for (int c = 0; c < units; c++) {
      const float* w = W + (size_t)c * cin;
      float acc = 0.f;
      for (int k = 0; k < cin; k++) acc += f[k] * w[k];
      float y = acc * alpha[c] + shift[c];
      y = y > 0.f ? y : 0.f;
      xo[c] = y;
      if (y > g[c]) g[c] = y;
    }
  }
  free(alpha);
}

/* PillarFeatureNet.forward in eval mode (pe:174-182) on already-decorated features.
 * num_filters[0] = F+5, num_filters[1..L] as in the YAML.  params = concatenation of the per-layer
 * blocks described above.  feat_max: P x num_filters[L].  The trailing scatter_max at pe:180 is
 * idempotent (it re-maxes values already gathered from the max) and is therefore folded away. */
void orc_pfn_eval(const float* feat, int64_t m, const int64_t* inv, int64_t P, int n_layers, const int* num_filters,
                  const float* params, float eps, float* feat_max) {
  const float* cur = feat;
  float* owned = NULL;
  int cin = num_filters[0];
  for (int l = 0; l < n_layers; l++) {
    int last = (l == n_layers - 1);
    int units = last ? num_filters[l + 1] : num_filters[l + 1] / 2;
    float* x = (float*)malloc(sizeof(float) * (size_t)(m + 1) * units);
    float* g = last ? feat_max : (float*)malloc(sizeof(float) * (size_t)(P + 1) * units);
    orc_pfn_layer_eval(cur, m, cin, units, params, eps, inv, P, x, g);
    params += (size_t)units * cin + 4 * (size_t)units;
    if (!last) {
      int cout = 2 * units;
      float* nxt = (float*)malloc(sizeof(float) * (size_t)(m + 1) * cout);
      for (int64_t j = 0; j < m; j++) {
        memcpy(nxt + j * cout, x + j * units, sizeof(float) * units);
        memcpy(nxt + j * cout + units, g + inv[j] * units, sizeof(float) * units);
      }
      free(g);
      free(owned);
      owned = nxt;
      cur = nxt;
      cin = cout;
    }
    free(x);
  }
  free(owned);
}

/* Dense BEV canvas: canvas[b, :, yi, xi] = feat_max[rank], zero elsewhere -- the dense equivalent of
 * SparseConvTensor(feats, coors, (ny,nx), B).dense()  (det3d/models/backbones/sparse_resnet.py:63-68).
 * Layout here: NCHW fp32 [B][C][gy][gx]. */
void orc_scatter_canvas(const float* feat_max, const int32_t* coords, int64_t P, int C, int64_t B, int64_t gy,
                        int64_t gx, float* canvas) {
  memset(canvas, 0, sizeof(float) * (size_t)(B * C * gy * gx));
  for (int64_t r = 0; r < P; r++) {
    int64_t b = coords[r * 3], yi = coords[r * 3 + 1], xi = coords[r * 3 + 2];
    for (int c = 0; c < C; c++) canvas[((b * C + c) * gy + yi) * gx + xi] = feat_max[r * C + c];
  }
}

/* Whole reader in one call (used as bench.py's cpu_baseline "port"): returns P, or -1.
 * Scratch is allocated inside.  coords/feat_max sized for n rows by the caller; canvas may be NULL. */
int64_t orc_reader_forward(const float* pts, int64_t n, int stride, const float* pc_min, const float* vs, int64_t gx,
                           int64_t gy, int n_layers, const int* num_filters, const float* params, float eps,
                           int32_t* coords, float* feat_max, int64_t B, float* canvas) {
  int64_t* kept = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
  int64_t* inv = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n + 1));
  int64_t P = 0;
  int64_t m = orc_voxelize(pts, n, stride, pc_min, vs, gx, gy, kept, inv, coords, &P);
  float* feat = (float*)malloc(sizeof(float) * (size_t)(m + 1) * (stride + 4));
  orc_decorate(pts, stride, kept, inv, m, P, pc_min, vs, feat);
  orc_pfn_eval(feat, m, inv, P, n_layers, num_filters, params, eps, feat_max);
  if (canvas) orc_scatter_canvas(feat_max, coords, P, num_filters[n_layers], B, gy, gx, canvas);
  free(feat);
  free(kept);
  free(inv);
  return P;
}
