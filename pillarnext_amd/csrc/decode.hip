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
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
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
  int o_hm, o_iou;  // channel of the first class logit / of the iou logit in the packed tensor (10 + has_iou / 10 for the full packing)
  int lazy;         // 1: the packed tensor holds [iou] hm only; centre / range test happen after the regression branches were
                    //    evaluated at the selected candidates (k_decode_boxes_lazy)
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
struct f16bits {  // an fp16 head output: the same 16-bit storage, IEEE half instead of bfloat16
  unsigned short v;
};
template <>
__device__ __forceinline__ float ldf<f16bits>(const f16bits* p, int i) {
  return (float)__builtin_bit_cast(_Float16, p[i].v);
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
  const int o_hm = tk.o_hm;
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
  if (tk.use_lim && !tk.lazy) {
    const float xs = ((float)(cell % tk.W) + ldf<T>(p, 0)) * tk.osf * tk.vx + tk.pcx;
    const float ys = ((float)(cell / tk.W) + ldf<T>(p, 1)) * tk.osf * tk.vy + tk.pcy;
    const float z = ldf<T>(p, 2);
    ok = ok && xs >= tk.lim[0] && ys >= tk.lim[1] && z >= tk.lim[2] && xs <= tk.lim[3] && ys <= tk.lim[4] && z <= tk.lim[5];
  }
  unsigned long long key = ~0ULL;
  if (ok) {
    float iou = 1.f;
    if (tk.has_iou) iou = fminf(fmaxf((ldf<T>(p, tk.o_iou) + 1.f) * 0.5f, 0.f), 1.f);
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

// Lazy head (models.FusedPillarNeXt): only the class / iou maps are dense; the regression branches were evaluated at the first
// seg_len[s] candidates of every segment (cand, 10 values per slot: reg 2, height 1, dim 3, rot 2, vel 2, in slot order s * pre_max + j).
// One workgroup per segment decodes its candidates, applies the centre range test of centerhead.py:343-346 -- which the reference
// applies BEFORE the top-pre_max cut -- and compacts the survivors in order.  If a segment that was cut at pre_max loses a candidate
// here, the reference would have admitted a lower-ranked one: *flag is raised and the host re-runs that batch through the dense path.
__global__ __launch_bounds__(256) void k_decode_boxes_lazy(const DecodeTask* __restrict__ tasks, const int64_t* __restrict__ task_key_off, int n_tasks,
                                                           int n_classes_total, const unsigned long long* __restrict__ sorted_keys,
                                                           const int64_t* __restrict__ order, const int64_t* __restrict__ seg_start,
                                                           int32_t* __restrict__ seg_len, const int32_t* __restrict__ seg_total, int pre_max,
                                                           const float* __restrict__ cand, float* __restrict__ boxes9, float* __restrict__ boxes7,
                                                           float* __restrict__ scores, int32_t* __restrict__ flag) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int s = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int n = seg_len[s];
  const int cls = s % n_classes_total;
  int ti = 0;
  while (ti + 1 < n_tasks && cls >= tasks[ti + 1].cls_off) ti++;
  const DecodeTask tk = tasks[ti];
  const int HW = tk.H * tk.W;
  if (t == 0) s_base = 0;
  __syncthreads();
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int j = j0 + t;
    bool ok = false;
    float o[9], sc = 0.f;
    if (j < n) {
      const int64_t pos = seg_start[s] + j;
      const int64_t local = order[pos] - task_key_off[ti];  // = b * HW + cell
      const int cell = (int)(local % HW);
      const float* v = cand + ((int64_t)s * pre_max + j) * 10;
      o[0] = ((float)(cell % tk.W) + v[0]) * tk.osf * tk.vx + tk.pcx;
      o[1] = ((float)(cell / tk.W) + v[1]) * tk.osf * tk.vy + tk.pcy;
      o[2] = v[2];
      o[3] = expf(v[3]);
      o[4] = expf(v[4]);
      o[5] = expf(v[5]);
      o[6] = v[8];
      o[7] = v[9];
      o[8] = atan2f(v[6], v[7]);
      sc = __uint_as_float(0xFFFFFFFFu - (unsigned)(sorted_keys[pos] & 0xFFFFFFFFull));
      ok = !tk.use_lim || (o[0] >= tk.lim[0] && o[1] >= tk.lim[1] && o[2] >= tk.lim[2] && o[0] <= tk.lim[3] && o[1] <= tk.lim[4] && o[2] <= tk.lim[5]);
    }
    const unsigned long long m = __ballot(ok);
    if (lane == 0) s_wave[wv] = __popcll(m);
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wv; w++) before += s_wave[w];
    if (ok) {
      const int64_t dst = (int64_t)s * pre_max + before + __popcll(m & ((1ull << lane) - 1ull));
#pragma unroll
      for (int k = 0; k < 9; k++) boxes9[dst * 9 + k] = o[k];
#pragma unroll
      for (int k = 0; k < 6; k++) boxes7[dst * 7 + k] = o[k];
      boxes7[dst * 7 + 6] = o[8];
      scores[dst] = sc;
    }
    __syncthreads();
    if (t == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  if (t == 0) {
    if (s_base < n && seg_total[s] > pre_max) atomicOr(flag, 1);
    seg_len[s] = s_base;
  }
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Segmented top-k: the first pre_max candidates of every (sample, class) segment in descending-score order (box_torch_ops.py:13-15:
// scores.sort(descending) + [:pre_maxsize]), ties in ascending cell order (what a stable sort of the key array gives) -- WITHOUT
// sorting the 6 M keys of a frame batch, of which a few percent are valid: a 4096-bin score histogram per segment finds the score
// bin that the pre_max-th candidate falls into, the candidates at or above that bin (pre_max + at most one bin's population) are
// collected per segment and sorted in LDS.
constexpr int kBins = 4096, kSortCap = 8192;

__device__ __forceinline__ int score_bin(uint32_t low32) {
  const uint32_t bits = 0xFFFFFFFFu - low32;  // fp32 bits of a score in (0, 1]
  const uint32_t base = 0x3D800000u;          // 2^-4; everything below shares bin 0 (monotone clamp)
  const uint32_t d = bits > base ? (bits - base) >> 13 : 0u;
  return (int)(d < (uint32_t)kBins ? d : (uint32_t)kBins - 1u);
}

// Wave-aggregated counter update: lanes that hit the same counter are served by ONE device atomic (same-address atomics serialise at
// ~13 ns each: a freshly initialised head puts half of all cells into the same score bin -- 3 M atomics on one word = 39 ms).
// Returns the lane's position (old value + its rank among the lanes of its group); inactive lanes get 0.
__device__ __forceinline__ uint32_t wave_agg_add(uint32_t* __restrict__ counters, uint32_t index, bool active) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(active);
  uint32_t pos = 0;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t v = (uint32_t)__shfl((int)index, leader);
    const unsigned long long grp = __ballot(active && index == v) & todo;
    uint32_t old = 0;
    if (lane == leader) old = atomicAdd(&counters[v], (uint32_t)__popcll(grp));
    old = (uint32_t)__shfl((int)old, leader);
    if ((grp >> lane) & 1ull) pos = old + (uint32_t)__popcll(grp & ((1ull << lane) - 1ull));
    todo &= ~grp;
  }
  return pos;
}

__global__ __launch_bounds__(256) void k_topk_hist(const unsigned long long* __restrict__ keys, int64_t n, uint32_t* __restrict__ hist) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long k = i < n ? keys[i] : ~0ULL;
  const bool ok = k != ~0ULL;
  wave_agg_add(hist, ok ? (uint32_t)(k >> 32) * kBins + (uint32_t)score_bin((uint32_t)k) : 0u, ok);
}

// one block per segment: threshold bin tb[s] = the highest bin b with count(bins >= b) >= pre_max (0 if the segment has fewer
// candidates), need[s] = count(bins >= tb[s])
__global__ __launch_bounds__(256) void k_topk_select(const uint32_t* __restrict__ hist, int pre_max, int32_t* __restrict__ tb, uint32_t* __restrict__ need) {
  __shared__ uint32_t s_sum[256];
  __shared__ int s_tb;
  const int s = blockIdx.x, t = threadIdx.x;
  const uint32_t* h = hist + (size_t)s * kBins;
  uint32_t loc[16], tot = 0;  // thread t owns bins [16t, 16t+16)
#pragma unroll
  for (int k = 0; k < 16; k++) {
    loc[k] = h[16 * t + k];
    tot += loc[k];
  }
  s_sum[t] = tot;
  if (t == 0) s_tb = 0;
  __syncthreads();
  uint32_t above = 0;  // candidates in the bins of higher threads
  for (int q = t + 1; q < 256; q++) above += s_sum[q];
  __syncthreads();
  // walking down from the top, the first bin where the running count reaches pre_max lies in exactly one thread's range
  if (above < (uint32_t)pre_max && above + tot >= (uint32_t)pre_max) {
    uint32_t run = above;
    for (int k = 15; k >= 0; k--) {
      run += loc[k];
      if (run >= (uint32_t)pre_max) {
        s_tb = 16 * t + k;
        break;
      }
    }
  }
  __syncthreads();
  const int b = s_tb;
  uint32_t part = 0;
#pragma unroll
  for (int k = 0; k < 16; k++)
    if (16 * t + k >= b) part += loc[k];
  __syncthreads();
  s_sum[t] = part;
  __syncthreads();
  if (t == 0) {
    uint32_t n = 0;
    for (int q = 0; q < 256; q++) n += s_sum[q];
    tb[s] = b;
    need[s] = n;
  }
}

__global__ __launch_bounds__(256) void k_topk_offsets(const uint32_t* __restrict__ need, int S, uint32_t* __restrict__ base, uint32_t* __restrict__ cursor) {
  if (threadIdx.x == 0) {  // S is a few hundred at most
    uint32_t run = 0;
    for (int s = 0; s < S; s++) {
      base[s] = run;
      cursor[s] = 0;
      run += need[s];
    }
    base[S] = run;
  }
}

__global__ __launch_bounds__(256) void k_topk_collect(const unsigned long long* __restrict__ keys, int64_t n, const int32_t* __restrict__ tb,
                                                      const uint32_t* __restrict__ base, uint32_t* __restrict__ cursor,
                                                      unsigned long long* __restrict__ list) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long k = i < n ? keys[i] : ~0ULL;
  const uint32_t seg = (uint32_t)(k >> 32), low = (uint32_t)k;
  const bool ok = k != ~0ULL && score_bin(low) >= tb[seg];
  const uint32_t pos = wave_agg_add(cursor, ok ? seg : 0u, ok);
  if (ok) list[base[seg] + pos] = ((unsigned long long)low << 32) | (uint32_t)i;  // ascending = score descending, then key index ascending
}

__device__ __forceinline__ void bitonic_lds(unsigned long long* a, int n2, int t) {  // n2: power of two, 256 threads
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < n2; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long x = a[i], y = a[l];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            a[i] = y;
            a[l] = x;
          }
        }
      }
      __syncthreads();
    }
}

// one block per segment: sort the collected candidates and emit the first pre_max
__global__ __launch_bounds__(256) void k_topk_sort(const unsigned long long* __restrict__ list, const uint32_t* __restrict__ base,
                                                   const uint32_t* __restrict__ need, int pre_max, unsigned long long* __restrict__ sorted_keys,
                                                   int64_t* __restrict__ order, int64_t* __restrict__ seg_start, int32_t* __restrict__ seg_len) {
  __shared__ unsigned long long s_a[kSortCap];
  const int s = blockIdx.x, t = threadIdx.x;
  const uint32_t n = need[s];
  const unsigned long long* src = list + base[s];
  const int keepn = (int)min(n, (uint32_t)pre_max);
  // chunks of at most kSortCap - (what is kept so far): sort, keep the first pre_max, merge the next chunk
  int have = 0;
  uint32_t done = 0;
  while (done < n || have == 0) {
    const int room = kSortCap - have;
    const int take = (int)min((uint32_t)room, n - done);
    for (int i = t; i < take; i += 256) s_a[have + i] = src[done + i];
    const int cnt = have + take;
    int n2 = 1;
    while (n2 < cnt) n2 <<= 1;
    for (int i = cnt + t; i < n2; i += 256) s_a[i] = ~0ULL;
    __syncthreads();
    bitonic_lds(s_a, n2, t);
    done += (uint32_t)take;
    have = min(cnt, pre_max);
    if (n == 0) break;
  }
  const unsigned long long segbits = (unsigned long long)(uint32_t)s << 32;
  for (int j = t; j < keepn; j += 256) {
    const unsigned long long v = s_a[j];
    sorted_keys[(int64_t)s * pre_max + j] = segbits | (v >> 32);
    order[(int64_t)s * pre_max + j] = (int64_t)(uint32_t)v;
  }
  if (t == 0) {
    seg_start[s] = (int64_t)s * pre_max;
    seg_len[s] = keepn;
  }
}

// ---------------------------------------------------------------------------------------------------------------- exact radix select
// Segmented top-k WITHOUT sorting the key array and without a weak spot: the histogram form above collects the whole threshold bin and
// sorts it, which is fine for a trained head (a few thousand candidates) and a disaster for a freshly initialised one, where half of all
// cells pass the score threshold and tens of thousands share one bf16-quantised score (6.4 ms against 0.87 ms for the sort).  Here the
// k-th smallest COMPOSITE  C = (low 32 key bits = ~score bits) << 32 | key index  of every segment is found exactly, most significant
// digit first (11 + 11 + 10 bits of the score, then -- only while a segment still has more ties than it needs -- 11 + 11 + 10 bits of the
// index): composites are unique, so "C <= threshold" selects exactly min(k, valid) keys, ties resolved by ascending index like a stable sort.
// A pass = one histogram launch (a workgroup owns 16 384 consecutive keys; they belong to at most a handful of segments, so it
// histograms into 4 LDS slots of 2 048 bins -- no global atomics -- and writes its rows; keys of a fifth segment fall back to global
// atomics) + one select launch (per segment: sum the rows of its slots, find the digit, narrow the prefix).  Segments finish early: fewer
// than k valid keys (every pass after the first is then a no-op for them) or a digit bucket that is needed completely.
constexpr int kRsBins = 2048, kRsSlots = 4, kRsChunk = 16384, kRsThreads = 512, kRsMaxK = 4096;

__device__ __forceinline__ int rs_shift(int p) { return p == 0 ? 53 : p == 1 ? 42 : p == 2 ? 32 : p == 3 ? 21 : p == 4 ? 10 : 0; }
__device__ __forceinline__ int rs_width(int p) { return (p == 2 || p == 5) ? 10 : 11; }

struct RsState {
  unsigned long long* prefix;  // [S] digits selected so far
  unsigned long long* thr;     // [S] final threshold composite (inclusive), valid once flag != 0
  uint32_t* rem;               // [S] rank (1-based) of the wanted composite inside the current prefix bucket
  uint32_t* flag;              // [S] 1 = finished
  uint32_t* total;             // [S] valid keys of the segment
  uint32_t* cursor;            // [S] collect cursor
  int32_t* n_open;             // segments not finished
};

// per-list state of the select + the overflow histograms (S x kRsBins words, zero) in ONE launch (round 6: the memset of `ovf` was one of nine
// fillBufferAligned dispatches per step)
__global__ __launch_bounds__(256) void k_rs_init(RsState st, int S, int k, uint32_t* __restrict__ ovf) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < S) st.prefix[s] = 0ull, st.thr[s] = 0ull, st.rem[s] = (uint32_t)k, st.flag[s] = 0u, st.total[s] = 0u, st.cursor[s] = 0u;
  if (s == 0) *st.n_open = S;
  for (int64_t e = s; e < (int64_t)S * kRsBins; e += (int64_t)gridDim.x * 256) ovf[e] = 0u;
}

__global__ __launch_bounds__(kRsThreads) void k_rs_hist(const unsigned long long* __restrict__ keys, int64_t n, int p, RsState st, int32_t* __restrict__ slotseg,
                                                        uint32_t* __restrict__ rows, uint32_t* __restrict__ ovf) {
  __shared__ uint32_t s_hist[kRsSlots * kRsBins];
  __shared__ int s_seg[kRsSlots];
  __shared__ unsigned long long s_pref[kRsSlots];
  __shared__ int s_fin[kRsSlots];
  if (*st.n_open <= 0) return;  // uniform: every segment is finished
  const int t = threadIdx.x, lane = t & 63;
  const int chunk = blockIdx.x;
  const int64_t i0 = (int64_t)chunk * kRsChunk;
  for (int e = t; e < kRsSlots * kRsBins; e += kRsThreads) s_hist[e] = 0u;
  if (t < kRsSlots) {
    int sg = -1;
    if (p > 0) sg = slotseg[chunk * kRsSlots + t];
    s_seg[t] = sg;
    s_pref[t] = sg >= 0 ? st.prefix[sg] : 0ull;
    s_fin[t] = sg >= 0 ? (int)st.flag[sg] : 1;
  }
  __syncthreads();
  const int shift = rs_shift(p), width = rs_width(p);
  const uint32_t dmask = (1u << width) - 1u;
  for (int j = t; j < kRsChunk; j += kRsThreads) {
    const int64_t i = i0 + j;
    const unsigned long long key = i < n ? keys[i] : ~0ULL;
    const bool valid = key != ~0ULL;
    const int seg = (int)(key >> 32);
    const unsigned long long C = (key << 32) | (unsigned long long)(uint32_t)i;
    int slot = -1;
#pragma unroll
    for (int q = 0; q < kRsSlots; q++)
      if (s_seg[q] == seg) slot = q;
    if (p == 0 && valid && slot < 0) {  // first pass: the chunk's segments claim the slots in the order they turn up
      for (int q = 0; q < kRsSlots && slot < 0; q++) {
        const int old = atomicCAS(&s_seg[q], -1, seg);
        if (old == -1 || old == seg) slot = q;
      }
    }
    bool part = valid;
    if (p > 0 && valid) {
      if (slot >= 0) part = !s_fin[slot] && (C >> (shift + width)) == s_pref[slot];
      else part = st.flag[seg] == 0u && (C >> (shift + width)) == st.prefix[seg];
    }
    const uint32_t bin = (uint32_t)(C >> shift) & dmask;
    if (part && slot < 0) {
      atomicAdd(&ovf[(size_t)seg * kRsBins + bin], 1u);
      part = false;
    }
    // same-address LDS atomics serialise: when many lanes of the wave hit one bin (a fresh head: every key in one or two score digits;
    // the high index digits of a chunk are equal by construction) the group is served by one add
    const int mine = part ? (int)(slot * kRsBins + bin) : -1;
    const unsigned long long act = __ballot(part);
    if (act) {
      const int lead = __shfl(mine, __ffsll((long long)act) - 1);
      const unsigned long long grp = __ballot(mine == lead);
      if (__popcll(grp) >= 8) {
        if (lane == __ffsll((long long)grp) - 1) atomicAdd(&s_hist[lead], (uint32_t)__popcll(grp));
        if (mine == lead) part = false;
      }
    }
    if (part) atomicAdd(&s_hist[mine], 1u);
  }
  __syncthreads();
#pragma unroll 1
  for (int q = 0; q < kRsSlots; q++) {
    if (s_seg[q] < 0) continue;  // uniform
    uint32_t* dst = rows + ((size_t)chunk * kRsSlots + q) * kRsBins;
    for (int b = t; b < kRsBins; b += kRsThreads) dst[b] = s_hist[q * kRsBins + b];
  }
  if (p == 0 && t < kRsSlots) slotseg[chunk * kRsSlots + t] = s_seg[t];
}

__global__ __launch_bounds__(256) void k_rs_select(int p, RsState st, const int32_t* __restrict__ slotseg, int n_slots, const uint32_t* __restrict__ rows,
                                                   uint32_t* __restrict__ ovf) {
  __shared__ int s_list[256];
  __shared__ int s_nlist;
  __shared__ uint32_t s_sum[256];
  __shared__ uint32_t s_res[4];
  const int seg = blockIdx.x, t = threadIdx.x;
  if (st.flag[seg] != 0u) return;  // uniform
  if (t == 0) s_nlist = 0;
  __syncthreads();
  for (int e = t; e < n_slots; e += 256)
    if (slotseg[e] == seg) {
      const int pos = atomicAdd(&s_nlist, 1);
      if (pos < 256) s_list[pos] = e;
    }
  __syncthreads();
  const int nl = s_nlist;
  uint32_t c[8];
  {
    uint32_t* o = ovf + (size_t)seg * kRsBins + 8 * t;
#pragma unroll
    for (int b = 0; b < 8; b++) c[b] = o[b], o[b] = 0u;  // the overflow row is re-armed for the next pass
  }
  if (nl <= 256) {
    for (int l = 0; l < nl; l++) {
      const uint4* r = reinterpret_cast<const uint4*>(rows + (size_t)s_list[l] * kRsBins + 8 * t);
      const uint4 a = r[0], b = r[1];
      c[0] += a.x, c[1] += a.y, c[2] += a.z, c[3] += a.w, c[4] += b.x, c[5] += b.y, c[6] += b.z, c[7] += b.w;
    }
  } else {  // a segment spread over more than 256 chunk slots: walk the table
    for (int e = 0; e < n_slots; e++)
      if (slotseg[e] == seg) {
        const uint32_t* r = rows + (size_t)e * kRsBins + 8 * t;
#pragma unroll
        for (int b = 0; b < 8; b++) c[b] += r[b];
      }
  }
  uint32_t tot = 0;
#pragma unroll
  for (int b = 0; b < 8; b++) tot += c[b];
  s_sum[t] = tot;
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (int q = 0; q < 256; q++) {
    const uint32_t v = s_sum[q];
    if (q < t) before += v;
    total += v;
  }
  const uint32_t rem = st.rem[seg];
  if (p == 0 && t == 0) st.total[seg] = total;
  if (total < rem) {  // fewer valid keys than k (first pass only): everything is taken
    if (t == 0) {
      st.thr[seg] = ~0ULL;
      st.flag[seg] = 1u;
      atomicSub(st.n_open, 1);
    }
    return;
  }
  if (before < rem && before + tot >= rem) {  // exactly one thread
    uint32_t run = before;
    int d = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      if (run < rem && run + c[b] >= rem) {
        d = b;
        s_res[0] = (uint32_t)(8 * t + b), s_res[1] = run, s_res[2] = c[b];
      }
      run += c[b];
    }
    (void)d;
  }
  __syncthreads();
  if (t == 0) {
    const int shift = rs_shift(p), width = rs_width(p);
    const unsigned long long np = (st.prefix[seg] << width) | (unsigned long long)s_res[0];
    const uint32_t nrem = rem - s_res[1], bucket = s_res[2];
    if (bucket == nrem || p == 5) {  // the whole bucket is wanted (or the composite is pinned down to its last bit)
      st.thr[seg] = ((np + 1ull) << shift) - 1ull;
      st.flag[seg] = 1u;
      atomicSub(st.n_open, 1);
    } else {
      st.prefix[seg] = np;
      st.rem[seg] = nrem;
    }
  }
}

__global__ __launch_bounds__(256) void k_rs_collect(const unsigned long long* __restrict__ keys, int64_t n, RsState st, int k, unsigned long long* __restrict__ list) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long key = i < n ? keys[i] : ~0ULL;
  const uint32_t seg = (uint32_t)(key >> 32);
  const unsigned long long C = (key << 32) | (unsigned long long)(uint32_t)i;
  const bool ok = key != ~0ULL && C <= st.thr[seg];
  const uint32_t pos = wave_agg_add(st.cursor, ok ? seg : 0u, ok);
  if (ok && pos < (uint32_t)k) list[(size_t)seg * k + pos] = C;
}

// one block per segment: sort the <= k collected composites, emit them in pnx_decode_boxes' layout
__global__ __launch_bounds__(256) void k_rs_sort(const unsigned long long* __restrict__ list, RsState st, int k, unsigned long long* __restrict__ sorted_keys,
                                                 int64_t* __restrict__ order, int64_t* __restrict__ seg_start, int32_t* __restrict__ seg_len,
                                                 int32_t* __restrict__ seg_total) {
  __shared__ unsigned long long s_a[kRsMaxK];
  const int s = blockIdx.x, t = threadIdx.x;
  const int n = (int)min(st.cursor[s], (uint32_t)k);
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  for (int i = t; i < n2; i += 256) s_a[i] = i < n ? list[(size_t)s * k + i] : ~0ULL;
  __syncthreads();
  if (n2 > 1) bitonic_lds(s_a, n2, t);
  const unsigned long long segbits = (unsigned long long)(uint32_t)s << 32;
  for (int j = t; j < n; j += 256) {
    const unsigned long long v = s_a[j];
    sorted_keys[(int64_t)s * k + j] = segbits | (v >> 32);
    order[(int64_t)s * k + j] = (int64_t)(uint32_t)v;
  }
  if (t == 0) {
    seg_start[s] = (int64_t)s * k;
    seg_len[s] = n;
    if (seg_total != nullptr) seg_total[s] = (int32_t)st.total[s];
  }
}

}  // namespace

extern "C" {

size_t pnx_decode_topk_workspace_bytes(int64_t n_keys, int32_t num_segments) {
  if (n_keys < 0 || num_segments < 1) return 0;
  const size_t nchunks = (size_t)((n_keys + kRsChunk - 1) / kRsChunk) + 1;
  const size_t S = (size_t)num_segments;
  return pnx_align_up(nchunks * kRsSlots * kRsBins * 4, 256) + pnx_align_up(nchunks * kRsSlots * 4, 256) + pnx_align_up(S * kRsBins * 4, 256) +
         2 * pnx_align_up((S + 8) * 8, 256) + 5 * pnx_align_up((S + 8) * 4, 256) + pnx_align_up(S * kRsMaxK * 8, 256) + 256;
}

// keys = the concatenated outputs of pnx_decode_keys for all tasks; outputs in the layout pnx_decode_boxes consumes
int pnx_decode_topk(const uint64_t* keys, int64_t n_keys, int32_t num_segments, int32_t pre_max, uint64_t* sorted_keys, int64_t* order, int64_t* seg_start,
                    int32_t* seg_len, int32_t* seg_total, void* workspace, size_t workspace_bytes, pnx_stream_t stream) {
  PNX_REQUIRE(keys && sorted_keys && order && seg_start && seg_len && workspace, PNX_ERR_INVALID, "null pointer");
  PNX_REQUIRE(n_keys >= 0 && n_keys < ((int64_t)1 << 32) && num_segments >= 1 && pre_max >= 1 && pre_max <= kRsMaxK, PNX_ERR_INVALID,
              "bad sizes (pre_max <= %d)", kRsMaxK);
  PNX_REQUIRE(workspace_bytes >= pnx_decode_topk_workspace_bytes(n_keys, num_segments), PNX_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int S = num_segments;
  const int nchunks = (int)((n_keys + kRsChunk - 1) / kRsChunk);
  PnxCarver c(workspace);
  uint32_t* rows = c.take<uint32_t>((size_t)(nchunks + 1) * kRsSlots * kRsBins);
  int32_t* slotseg = c.take<int32_t>((size_t)(nchunks + 1) * kRsSlots);
  uint32_t* ovf = c.take<uint32_t>((size_t)S * kRsBins);
  RsState rs;
  rs.prefix = c.take<unsigned long long>(S + 8);
  rs.thr = c.take<unsigned long long>(S + 8);
  rs.rem = c.take<uint32_t>(S + 8);
  rs.flag = c.take<uint32_t>(S + 8);
  rs.total = c.take<uint32_t>(S + 8);
  rs.cursor = c.take<uint32_t>(S + 8);
  rs.n_open = c.take<int32_t>(S + 8);
  unsigned long long* list = c.take<unsigned long long>((size_t)S * kRsMaxK);
  {
    int nb = (int)(((int64_t)S * kRsBins + 256 * 8 - 1) / (256 * 8));
    nb = nb < (S + 255) / 256 ? (S + 255) / 256 : (nb > 1024 ? 1024 : nb);
    k_rs_init<<<nb, 256, 0, st>>>(rs, S, pre_max, ovf);
  }
  for (int p = 0; p < 6; p++) {
    if (nchunks > 0) k_rs_hist<<<nchunks, kRsThreads, 0, st>>>((const unsigned long long*)keys, n_keys, p, rs, slotseg, rows, ovf);
    k_rs_select<<<S, 256, 0, st>>>(p, rs, slotseg, nchunks * kRsSlots, rows, ovf);
  }
  const unsigned nb = (unsigned)((n_keys + 255) / 256);
  if (nb > 0) k_rs_collect<<<nb, 256, 0, st>>>((const unsigned long long*)keys, n_keys, rs, pre_max, list);
  k_rs_sort<<<S, 256, 0, st>>>(list, rs, pre_max, (unsigned long long*)sorted_keys, order, seg_start, seg_len, seg_total);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

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
  else if (dtype == PNX_F16) k_decode_keys<f16bits><<<nb, 256, 0, st>>>((const f16bits*)packed, tk, batch, n_classes_total, (unsigned long long*)keys);
  else PNX_REQUIRE(false, PNX_ERR_UNSUPPORTED, "decode is built for fp32, bf16 and fp16 head outputs");
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

size_t pnx_decode_task_desc_bytes(void) { return sizeof(DecodeTask); }

// One stable radix sort of all candidate keys with their positions as payload, over the bits that can differ only: a key is
// (segment << 32 | ~score bits) or all ones, so 32 + bit_length(num_segments) bits order everything (invalid keys stay last) --
// 5 onesweep passes instead of the 8 a generic 64-bit sort runs.  rocPRIM is the library primitive here (as rocBLAS would be
// for a plain GEMM); torch.sort is the same primitive without the bit range.
size_t pnx_sort_keys_workspace_bytes(int64_t n_keys) {
  size_t bytes = 0;
  if (n_keys <= 0) return 256;
  rocprim::counting_iterator<int64_t> iota(0);
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, iota, (int64_t*)nullptr,
                                  (size_t)n_keys, 0u, 64u, (hipStream_t) nullptr);
  return bytes + 256;
}

int pnx_sort_keys(const uint64_t* keys, int64_t n_keys, int32_t num_segments, uint64_t* sorted_keys, int64_t* order, void* workspace,
                  size_t workspace_bytes, pnx_stream_t stream) {
  PNX_REQUIRE(keys && sorted_keys && order && workspace && n_keys > 0 && num_segments > 0, PNX_ERR_INVALID, "bad arguments");
  PNX_REQUIRE(workspace_bytes >= pnx_sort_keys_workspace_bytes(n_keys), PNX_ERR_INVALID, "workspace too small");
  unsigned bits = 0;
  while ((1 << bits) <= num_segments) bits++;  // 2^bits > num_segments: the all-ones pattern of an invalid key exceeds every valid segment
  size_t bytes = workspace_bytes;
  rocprim::counting_iterator<int64_t> iota(0);
  PNX_CHECK_HIP(rocprim::radix_sort_pairs(workspace, bytes, (const unsigned long long*)keys, (unsigned long long*)sorted_keys, iota, order,
                                          (size_t)n_keys, 0u, 32u + bits, (hipStream_t)stream));
  return PNX_OK;
}

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
  else if (dtype == PNX_F16)
    k_decode_boxes<f16bits><<<nb, 256, 0, st>>>((const f16bits* const*)task_ptrs_dev, (const DecodeTask*)task_descs_dev, task_key_off_dev, n_tasks,
                                                batch, (const unsigned long long*)sorted_keys, order, seg_start, seg_len, num_segments, pre_max, boxes9,
                                                boxes7, scores);
  else PNX_REQUIRE(false, PNX_ERR_UNSUPPORTED, "decode is built for fp32, bf16 and fp16 head outputs");
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

// Lazy-head variant of pnx_decode_boxes (see k_decode_boxes_lazy): cand = (num_segments * pre_max, 10) fp32 regression values per
// candidate slot; seg_len is updated in place to the survivors of the centre range test; seg_total = candidates per segment before
// the pre_max cut; *flag_dev |= 1 if a cut segment lost a candidate (the caller then falls back to the dense path).
int pnx_decode_boxes_lazy(const void* task_descs_dev, const int64_t* task_key_off_dev, int32_t n_tasks, int32_t n_classes_total,
                          const uint64_t* sorted_keys, const int64_t* order, const int64_t* seg_start, int32_t* seg_len, const int32_t* seg_total,
                          int32_t num_segments, int32_t pre_max, const float* cand, float* boxes9, float* boxes7, float* scores, int32_t* flag_dev,
                          pnx_stream_t stream) {
  PNX_REQUIRE(task_descs_dev && task_key_off_dev && sorted_keys && order && seg_start && seg_len && seg_total && cand && boxes9 && boxes7 && scores && flag_dev,
              PNX_ERR_INVALID, "null pointer");
  PNX_REQUIRE(n_tasks > 0 && num_segments > 0 && pre_max > 0 && n_classes_total > 0, PNX_ERR_INVALID, "bad sizes");
  k_decode_boxes_lazy<<<num_segments, 256, 0, (hipStream_t)stream>>>((const DecodeTask*)task_descs_dev, task_key_off_dev, n_tasks, n_classes_total,
                                                                     (const unsigned long long*)sorted_keys, order, seg_start, seg_len, seg_total, pre_max,
                                                                     cand, boxes9, boxes7, scores, flag_dev);
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
