// reader_bins.h -- grouping the kept points by pillar WITHOUT a per-point device-scope atomic and without a random scatter
// (device code, included by reader.hip after cell_rank()).  Reference semantics: pillar_encoder.py:106-123 (the
// torch.unique(dim=0) inverse + scatter_mean + feature decoration); nothing like its sort.
//
// Round 1 drew every point's slot inside its pillar from a global atomicAdd (k_rank) and scattered a 32-byte record per point
// (k_fill): device-scope atomics run at ~20/ns on the fabric and isolated 32-byte writes at ~150 lines/ns, which made those two
// kernels 170 us of a 1.09 ms reader.  Here:
//   k_bin_count    per point: pillar rank (bitmap popcount prefix), bin = rank >> sh (a bin = 2^sh consecutive pillars); a
//                  workgroup histograms its chunk of points over the bins in LDS and writes ONE row of a (bin x workgroup) matrix
//   scan           exclusive prefix over the matrix in bin-major order (pnx_scan.h): every (bin, workgroup) pair gets a private,
//                  deterministic range of the bin buffer -- no atomics, no contention
//   k_bin_scatter  the same chunks again: LDS cursors hand out positions, raw 32-byte records [x y z f.. | key | rank in bin] go to
//                  the bin buffer in runs (points of one chunk that share a bin are adjacent)
//   k_bin_sort     one workgroup per bin: counting sort by pillar INSIDE LDS (LDS atomics), exact fp64 per-pillar sums -> mean, the
//                  pillar centre and canvas cell from the key, then every point is written ONCE, fully decorated and in the PFN
//                  kernel's operand order, to its final slot of the pillar-sorted 64-byte record stream.
// A sorted record (16 words, one 64-byte line): words of the decorated feature vector f[0..11] (pe:123, then the constant 1 of the
// folded-BN shift column, then zeros) split by parity, because lane (point, h) of v_mfma_f32_32x32x2_f32 feeds K elements 2kk+h:
//   half 0: f0 f2 f4 f6 | f8 f10 aux rank        half 1: f1 f3 f5 f7 | f9 f11 aux cell
// aux = idx | rem << 16 (position inside the pillar / points still to come, each clamped to 16 bits), rank = pillar rank
// (torch.unique order), cell = (b*gy + yi)*gx + xi.  Record order inside a pillar depends on LDS atomic timing; nothing computed
// from the records does (max is order-free, the mean is an exact fp64 sum, every point's MFMA column is independent).
#pragma once

// ---- k_bin_count: rank of every point + one histogram row per workgroup
__global__ __launch_bounds__(1024) void k_bin_count(const int32_t* __restrict__ key, int64_t n, int chunk, int sh, int K1, int nwg,
                                                      const uint2* __restrict__ wcomb, const uint32_t* __restrict__ wblk,
                                                      int32_t* __restrict__ rank_out,
                                                      int32_t* __restrict__ pillar_of_point, uint32_t* __restrict__ histmat, PnxGeomDev g,
                                                      PnxFillJob fj) {
  extern __shared__ uint32_t s_hist[];
  const int t = threadIdx.x, nt = blockDim.x;  // 256..1024 threads: few, large chunks keep the (bin x workgroup) matrix small
  if (fj.quota > 0 && (int)blockIdx.x >= fj.n_main) {  // fill share (pnx_fill.h)
    pnx_fill_share(fj, g, s_hist, t, nt);
    return;
  }
  for (int b = t; b < K1; b += nt) s_hist[b] = 0u;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * chunk;
  const int64_t i1 = i0 + chunk < n ? i0 + chunk : n;
  for (int64_t i = i0 + t; i < i1; i += nt) {
    const int32_t k = key[i];
    int32_t r = -1;
    if (k >= 0) {
      r = cell_rank2(k, wcomb, wblk);
      atomicAdd(&s_hist[r >> sh], 1u);  // LDS
    }
    rank_out[i] = r;
    if (pillar_of_point) pillar_of_point[i] = r;
  }
  __syncthreads();
  for (int b = t; b < K1; b += nt) histmat[(int64_t)b * nwg + blockIdx.x] = s_hist[b];
}

__device__ __forceinline__ uint32_t mat_prefix(int64_t v, const uint32_t* __restrict__ hpre, const uint32_t* __restrict__ hblk) {
  return hblk[v >> PNX_SCAN_SHIFT] + hpre[v];
}

// ---- k_bin_scatter: raw records into the bins
__global__ __launch_bounds__(1024) void k_bin_scatter(const float* __restrict__ pts, int stride, const int32_t* __restrict__ key,
                                                        const int32_t* __restrict__ rank, int64_t n, int chunk, int sh, int K1, int nwg,
                                                        const uint32_t* __restrict__ hpre, const uint32_t* __restrict__ hblk,
                                                        uint32_t* __restrict__ binbuf, PnxGeomDev g, PnxFillJob fj) {
  extern __shared__ uint32_t s_cur[];
  const int t = threadIdx.x, nt = blockDim.x;
  if (fj.quota > 0 && (int)blockIdx.x >= fj.n_main) {  // fill share (pnx_fill.h)
    pnx_fill_share(fj, g, s_cur, t, nt);
    return;
  }
  for (int b = t; b < K1; b += nt) s_cur[b] = mat_prefix((int64_t)b * nwg + blockIdx.x, hpre, hblk);
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * chunk;
  const int64_t i1 = i0 + chunk < n ? i0 + chunk : n;
  const uint32_t smask = (1u << sh) - 1u;
  for (int64_t i = i0 + t; i < i1; i += nt) {
    const int32_t r = rank[i];
    if (r < 0) continue;
    const uint32_t pos = atomicAdd(&s_cur[r >> sh], 1u);  // LDS; the range [prefix, prefix + hist) is private to this workgroup
    const float* p = pts + i * stride;
    uint32_t v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = (k < stride - 1) ? __float_as_uint(p[1 + k]) : 0u;
    // (two 16-byte stores per lane; letting lane pairs write each other's halves so that one instruction covers whole records
    // measured SLOWER: 58 vs 36 us at C2 / 8 frames)
    uint4* o = reinterpret_cast<uint4*>(binbuf + (int64_t)pos * 8);
    o[0] = make_uint4(v[0], v[1], v[2], v[3]);
    o[1] = make_uint4(v[4], v[5], (uint32_t)key[i], (uint32_t)r & smask);
  }
}

// 4x4 transpose of 16-byte quads across the four lanes of a DPP quad: in  W[4j + w] = word w of quad j of THIS lane's record,
// out W[4i + w] = word w of quad (lane & 3) of the record of lane (quad base + i).  Two butterfly steps (lane^1, lane^2).
__device__ __forceinline__ void quad_transpose16(uint32_t* W, int lq) {
  const bool o1 = (lq & 1) != 0, o2 = (lq & 2) != 0;
#pragma unroll
  for (int k = 0; k < 2; k++)
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t send = o1 ? W[4 * (2 * k) + w] : W[4 * (2 * k + 1) + w];
      const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
      W[4 * (2 * k) + w] = o1 ? recv : W[4 * (2 * k) + w];
      W[4 * (2 * k + 1) + w] = o1 ? W[4 * (2 * k + 1) + w] : recv;
    }
#pragma unroll
  for (int k = 0; k < 2; k++)
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t send = o2 ? W[4 * k + w] : W[4 * (k + 2) + w];
      const uint32_t recv = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
      W[4 * k + w] = o2 ? recv : W[4 * k + w];
      W[4 * (k + 2) + w] = o2 ? W[4 * (k + 2) + w] : recv;
    }
}

// ---- k_bin_sort: one workgroup per bin
constexpr int kSortBlock = 1024;

template <int F>
__global__ __launch_bounds__(kSortBlock) void k_bin_sort(const uint32_t* __restrict__ binbuf, PnxGeomDev g, int sh, int nwg, int64_t matlen,
                                                         const uint32_t* __restrict__ hpre, const uint32_t* __restrict__ hblk,
                                                         int32_t* counters, uint32_t* __restrict__ rec64,
                                                         uint32_t* __restrict__ pillar_first, uint32_t* __restrict__ pillar_cnt,
                                                         int32_t* __restrict__ cell_of_pillar, int32_t* __restrict__ coords,
                                                         int64_t pillar_capacity, int32_t* __restrict__ biglist, int bigcap,
                                                         PnxFillJob fj) {
  constexpr int C0 = F + 5;
  extern __shared__ __align__(16) unsigned char s_raw[];
  const int S = 1 << sh;
  uint32_t* s_start = reinterpret_cast<uint32_t*>(s_raw);  // S+1: counts, then exclusive starts (s_start[S] = records of the bin)
  uint32_t* s_cur = s_start + (S + 4);                     // S cursors
  uint32_t* s_key = s_cur + S;                             // S cell keys
  double* s_sum = reinterpret_cast<double*>(s_key + S);    // 3 doubles per pillar; later {mean x y z, centre x y, cell} as 6 words
  __shared__ uint32_t s_wave[kSortBlock / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = blockIdx.x;
  if (fj.quota > 0 && b >= fj.n_main) {  // fill share (pnx_fill.h)
    pnx_fill_share(fj, g, reinterpret_cast<uint32_t*>(s_raw), t, kSortBlock);
    return;
  }
  const int64_t P = counters[0];
  const int64_t r0 = (int64_t)b << sh;
  if (r0 >= P) return;  // block-uniform
  const int64_t v0 = (int64_t)b * nwg, v1 = v0 + nwg;
  const uint32_t bs = mat_prefix(v0, hpre, hblk);
  const uint32_t be = v1 >= matlen ? (uint32_t)counters[1] : mat_prefix(v1, hpre, hblk);
  const uint32_t nb = be - bs;

  for (int p = t; p < S; p += kSortBlock) {
    s_start[p] = 0u;
    s_sum[3 * p + 0] = 0.0;
    s_sum[3 * p + 1] = 0.0;
    s_sum[3 * p + 2] = 0.0;
  }
  __syncthreads();
  // pass 1: points per pillar, exact coordinate sums (scatter_mean numerator, pe:113), the pillar's cell key.  The raw records of the
  // first kKeep rounds stay in registers for pass 2 (a bin holds ~3 rounds of points on the nuScenes grid): one read of binbuf, not two.
  constexpr int kKeep = 4;
  uint4 ka[kKeep], kc[kKeep];
  auto tally = [&](const uint4& a, const uint4& c) {
    const uint32_t rl = c.w;
    atomicAdd(&s_start[rl], 1u);
    s_key[rl] = c.z;  // every point of the pillar stores the same key
    atomicAdd(&s_sum[3 * rl + 0], (double)__uint_as_float(a.x));
    atomicAdd(&s_sum[3 * rl + 1], (double)__uint_as_float(a.y));
    atomicAdd(&s_sum[3 * rl + 2], (double)__uint_as_float(a.z));
  };
#pragma unroll
  for (int it = 0; it < kKeep; it++) {
    const uint32_t j = bs + it * kSortBlock + t;
    ka[it] = make_uint4(0u, 0u, 0u, 0u);
    kc[it] = make_uint4(0u, 0u, 0u, 0u);
    if (j < be) {
      const uint4* q = reinterpret_cast<const uint4*>(binbuf + (int64_t)j * 8);
      ka[it] = q[0];
      kc[it] = q[1];
      tally(ka[it], kc[it]);
    }
  }
  for (uint32_t j = bs + kKeep * kSortBlock + t; j < be; j += kSortBlock) {
    const uint4* q = reinterpret_cast<const uint4*>(binbuf + (int64_t)j * 8);
    tally(q[0], q[1]);
  }
  __syncthreads();
  // exclusive scan of the S counts (S <= 2 * kSortBlock): thread t owns entries 2t, 2t+1
  uint32_t c0 = 0, c1 = 0;
  if (2 * t < S) c0 = s_start[2 * t];
  if (2 * t + 1 < S) c1 = s_start[2 * t + 1];
  const uint32_t mine = c0 + c1;
  uint32_t inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(inc, d);
    if (lane >= d) inc += y;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wave; w++) woff += s_wave[w];
  const uint32_t e0 = woff + inc - mine, e1 = e0 + c0;
  if (2 * t < S) {
    s_start[2 * t] = e0;
    s_cur[2 * t] = e0;
  }
  if (2 * t + 1 < S) {
    s_start[2 * t + 1] = e1;
    s_cur[2 * t + 1] = e1;
  }
  if (t == 0) s_start[S] = nb;
  // per pillar: mean (fp32 divide of the fp64 sum, pe:113-114), pillar centre (pe:119-120, each step rounded), canvas cell, coords
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int p = 2 * t + u;
    const uint32_t cnt = u ? c1 : c0;
    if (p < S && cnt > 0u) {
      const double sx = s_sum[3 * p + 0], sy = s_sum[3 * p + 1], sz = s_sum[3 * p + 2];
      const float fc = (float)cnt;
      const int32_t k = (int32_t)s_key[p];
      const int yi = k % g.gyp;
      const int tq = k / g.gyp;
      const int xi = tq % g.gx, bi = tq / g.gx;
      const int32_t cell = (bi * g.gy + yi) * g.gx + xi;
      float* info = reinterpret_cast<float*>(&s_sum[3 * p]);  // overlays this thread's own three sums
      const float mx = __fdiv_rn((float)sx, fc), my = __fdiv_rn((float)sy, fc), mz = __fdiv_rn((float)sz, fc);
      const float ctrx = __fadd_rn(__fadd_rn(__fmul_rn((float)xi, g.vx), __fdiv_rn(g.vx, 2.0f)), g.minx);
      const float ctry = __fadd_rn(__fadd_rn(__fmul_rn((float)yi, g.vy), __fdiv_rn(g.vy, 2.0f)), g.miny);
      info[0] = mx;
      info[1] = my;
      info[2] = mz;
      info[3] = ctrx;
      info[4] = ctry;
      info[5] = __int_as_float(cell);
      const int64_t gr = r0 + p;
      pillar_first[gr] = bs + (u ? e1 : e0);
      pillar_cnt[gr] = cnt;
      cell_of_pillar[gr] = cell;
      if (cnt > 32u && biglist != nullptr) {  // more points than one MFMA tile holds: the wave-per-pillar role of k_pfn3 (pfn_v3.hip)
        const int at = atomicAdd(&counters[3], 1);
        if (at < bigcap) biglist[at] = (int)gr;
      }
      if (coords != nullptr && gr < pillar_capacity) {
        coords[gr * 3 + 0] = bi;  // [b, yi, xi]  (pe:125 swaps x/y)
        coords[gr * 3 + 1] = yi;
        coords[gr * 3 + 2] = xi;
      }
    }
  }
  __syncthreads();
  // pass 2: every point to its final slot, decorated (pe:116-123) and in MFMA operand order.  A lane that stored its own
  // 64-byte record would issue four (after hipcc's regrouping: five) requests to one line; instead the four lanes of a quad
  // store ONE record per instruction, 16 bytes each, so that an instruction writes 16 complete lines: a 4x4 transpose of the
  // 16-byte quads across the quad's lanes (two DPP butterfly steps) leaves lane i holding quad i of all four records.
  const int lq = t & 3;
  auto emit = [&](const bool live, const uint4& a, const uint4& c) {  // every lane takes part (DPP reads the quad's lanes)
    uint32_t W[16];
#pragma unroll
    for (int k = 0; k < 16; k++) W[k] = 0u;
    uint32_t dst = 0xFFFFFFFFu;
    if (live) {
      const uint32_t rl = c.w;
      const uint32_t pos = atomicAdd(&s_cur[rl], 1u);
      const uint32_t st = s_start[rl], cnt = s_start[rl + 1] - st;
      const uint32_t idx = pos - st, rem = cnt - 1u - idx;
      const float* info = reinterpret_cast<const float*>(&s_sum[3 * rl]);
      const float raw[6] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z),
                            __uint_as_float(a.w), __uint_as_float(c.x), __uint_as_float(c.y)};
      float f[12];
#pragma unroll
      for (int k = 0; k < 12; k++) f[k] = 0.f;
#pragma unroll
      for (int k = 0; k < F; k++) f[k] = raw[k];
      f[F + 0] = __fsub_rn(raw[0], info[0]);
      f[F + 1] = __fsub_rn(raw[1], info[1]);
      f[F + 2] = __fsub_rn(raw[2], info[2]);
      f[F + 3] = __fsub_rn(raw[0], info[3]);
      f[F + 4] = __fsub_rn(raw[1], info[4]);
      f[C0] = 1.f;  // multiplies the folded-BN shift column of W0' (k_fold_bn)
      const uint32_t aux = min(idx, 0xFFFFu) | (min(rem, 0xFFFFu) << 16);
      W[0] = __float_as_uint(f[0]), W[1] = __float_as_uint(f[2]), W[2] = __float_as_uint(f[4]), W[3] = __float_as_uint(f[6]);
      W[4] = __float_as_uint(f[8]), W[5] = __float_as_uint(f[10]), W[6] = aux, W[7] = (uint32_t)(r0 + rl);
      W[8] = __float_as_uint(f[1]), W[9] = __float_as_uint(f[3]), W[10] = __float_as_uint(f[5]), W[11] = __float_as_uint(f[7]);
      W[12] = __float_as_uint(f[9]), W[13] = __float_as_uint(f[11]), W[14] = aux, W[15] = __float_as_uint(info[5]);
      dst = bs + pos;
    }
    quad_transpose16(W, lq);  // W[4i..4i+3] = quad lq of the record of lane (quad base + i)
#define PNX_QUAD_ROUND(Q)                                                                                                    \
  {                                                                                                                         \
    constexpr int CTRL = (Q) | ((Q) << 2) | ((Q) << 4) | ((Q) << 6); /* quad_perm [Q,Q,Q,Q]: the owner's slot */             \
    const uint32_t d = (uint32_t)__builtin_amdgcn_mov_dpp((int)dst, CTRL, 0xF, 0xF, true);                                  \
    if (d != 0xFFFFFFFFu)                                                                                                   \
      *reinterpret_cast<uint4*>(rec64 + (int64_t)d * 16 + 4 * lq) = make_uint4(W[4 * (Q)], W[4 * (Q) + 1], W[4 * (Q) + 2], W[4 * (Q) + 3]); \
  }
    PNX_QUAD_ROUND(0)
    PNX_QUAD_ROUND(1)
    PNX_QUAD_ROUND(2)
    PNX_QUAD_ROUND(3)
#undef PNX_QUAD_ROUND
  };
#pragma unroll
  for (int it = 0; it < kKeep; it++) {
    const uint32_t jb = bs + it * kSortBlock;
    if (jb < be) emit(jb + t < be, ka[it], kc[it]);  // block-uniform condition
  }
  for (uint32_t jb = bs + kKeep * kSortBlock; jb < be; jb += kSortBlock) {
    const uint32_t j = jb + t;
    uint4 a = make_uint4(0u, 0u, 0u, 0u), c = a;
    if (j < be) {
      const uint4* q = reinterpret_cast<const uint4*>(binbuf + (int64_t)j * 8);
      a = q[0];
      c = q[1];
    }
    emit(j < be, a, c);
  }
}


// LDS bytes of k_bin_sort for bins of 2^sh pillars
static inline size_t bin_sort_lds(int sh) {
  const size_t S = (size_t)1 << sh;
  return (S + 4) * 4 + S * 4 + S * 4 + S * 24;
}
