// conv_pc.h -- producer / consumer form of the stride-1 masked 3x3 convolution (included by conv3x3.hip, which supplies the fragment
// helpers, the tap loop, the tile scheduler and the element type).  SubMConv2d + folded BN + [residual] + ReLU + active-site mask of
// det3d/models/utils/sparse_conv.py:16-63 / det3d/models/backbones/sparse_resnet.py:50-68, same arithmetic as k_conv3x3_lds / k_conv3x3_ldsx.
//
// Why a second form.  The row-split kernels run a workgroup through mask -> barrier -> stage -> barrier -> taps -> epilogue, and two such
// workgroups per CU are all the latency hiding there is: the SQ counters show their waves parked 42 % of the time on the LiDAR masks and
// the MFMA pipe 28 % busy (profiles/r02_conv_sq_counters.md).  A wave cannot prefetch the next tile behind its own weight stream because
// LDS-DMA and the weight-fragment loads share one in-order vmcnt.  Here the roles are separate waves of ONE 12-wave workgroup per CU:
//   waves 8..11  producers: tile schedule (tickets three tiles ahead), mask bytes -> row masks, the deal of the tile's active rows to the row groups
//                (producer 0), zero-fill of rows that went inactive, and the HBM -> LDS staging of the NEXT tile (global_load_lds) into a ring of
//                halo-tile slots, one fill ahead of the consumers; their vmcnt holds nothing else;
//   waves 0..7   consumers: NRG row groups x NCG groups of 32 output channels; a wave = up to 4 row segments x 32 channels (64 accumulator
//                registers, 140-167 in all and no scratch, so three waves fit a SIMD: two consumers + one producer), weights through a buffer
//                resource four k-steps ahead (conv_taps), B fragments from the ring slot, epilogue = complete 64-byte half lines.
// Measured (profiles/r06_conv_pc_ab.txt): the staging was not what bounded the row-split kernels -- with their rolled tap loop this form was 3 % slower;
// with the unrolled loop it is 5 % faster at 64 channels (the default there: PNX_CONV_PC bit 0) and equal at 128 / 256 (bit 1), where every B fragment is
// read by four consumer waves instead of two.
// One s_barrier per fill hands a slot over in both directions (the consumers have finished reading slot g - NSLOT + 1 .. and fill g + 1 has
// landed); the consumers' barrier does not wait for their stores (s_waitcnt lgkmcnt(0) only).
//   CIN = 64         : 16 x 32 tiles, 4 row groups x 2 channel groups, COUT / 64 passes over one fill, ring of 2 slots (2 x 76.5 KiB)
//   CIN = 128 / 256  : 8 x 32 tiles, 2 row groups x 4 channel groups, fills = (pass of 128 output channels, 64-channel input slab), ring of 3
// (A stride-2 form -- 4 x 32 output tiles, four consumer + four producer waves, two 73 KiB slots -- was built and measured in round 6: correct, and 5-18 %
// slower than k_conv3x3_s2 on all three entry layers; removed.  profiles/r06_conv_pc_ab.txt (10).)
#pragma once

__device__ __forceinline__ void pc_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void pc_barrier_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef PNX_CONV_TIMERS
#define PC_T_DECL unsigned long long pc_T[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pc_tk = __builtin_amdgcn_s_memtime();
#define PC_TOCK(k)                                               \
  {                                                              \
    const unsigned long long _n = __builtin_amdgcn_s_memtime();  \
    pc_T[k] += _n - pc_tk;                                       \
    pc_tk = _n;                                                  \
  }
#define PC_T_FLUSH                                                   \
  if ((threadIdx.x & 63) == 0) {                                     \
    for (int k = 0; k < 8; k++) atomicAdd(&g_conv_T[k], pc_T[k]);    \
  }
#else
#define PC_T_DECL
#define PC_TOCK(k)
#define PC_T_FLUSH
#endif

// Everything a consumer wave does for one tile: NPASS passes (CIN = 64: over the one fill; otherwise over NSLAB fills each), every fill closed
// by pc_barrier_lds().  NR = 0: the wave has no row in this tile and only keeps the barrier count.
// X3: the fp32 product from bf16 pairs (pnx_conv3x3_x3): the fills alternate between the tensor of high halves (taps with W_hi, then with W_lo) and the
// tensor of low halves (taps with W_hi); no bias, no ReLU, and the accumulators go out as fp32.
template <int NR, int CIN, int COUT, bool HAS_RES, int NRG, int NSLOT, int SLOT_N, bool X3 = false>
__device__ __forceinline__ void pc_rows(const uint4* __restrict__ s_ring, int& g, const uint4* __restrict__ wfrag, const uint4* __restrict__ wfrag2,
                                        const float* __restrict__ bias,
                                        const uint16_t* __restrict__ res_t, uint16_t* __restrict__ y_t, const int (&rrow)[4], const uint32_t (&rmask)[4],
                                        int W, int n_valid, int cg, int relu, int px, int kb, int lane
#ifdef PNX_CONV_TIMERS
                                        , unsigned long long (&pc_T)[8], unsigned long long& pc_tk
#endif
) {
  constexpr int NCG = 8 / NRG, PASS_C = NCG * 32, NPASS = COUT / PASS_C, NSLAB = CIN / 64, NRA = NR > 0 ? NR : 1;
  constexpr int MTALL = COUT / 32, CB = CIN / 16;
  int rbase[4];
#pragma unroll
  for (int j = 0; j < 4; j++) rbase[j] = rrow[j] * LDS_HW * 8;
  const uint32_t res_off = (uint32_t)(px * COUT + 8 * kb) * 2u;
#pragma unroll 1
  for (int pass = 0; pass < NPASS; pass++) {
    const int mt = pass * NCG + cg;  // 32-channel output tile of this wave
    v16f acc[NRA][1];
    if (NR > 0) {
      v16f bq;
      if (X3 && bias == nullptr) {
#pragma unroll
        for (int i = 0; i < 16; i++) bq[i] = 0.f;
      } else {
        bq = bias_tile(bias, mt * 32, kb);
      }
#pragma unroll
      for (int j = 0; j < NR; j++) acc[j][0] = bq;
    }
    uint4 rq[2];
#pragma unroll 1
    for (int sq = 0; sq < NSLAB * (X3 ? 2 : 1); sq++) {
      const int sl = X3 ? sq % NSLAB : sq;
      if (NR > 0) {
        if (HAS_RES && sl == NSLAB - 1) {  // residual lines of row 0: requested before the last slab's taps, arrive under them
          const bool a0 = (rmask[0] >> px) & 1u;
          const char* rp = reinterpret_cast<const char*>(res_t + ((int64_t)rrow[0] * W) * COUT + mt * 32) + res_off;
#pragma unroll
          for (int t = 0; t < 2; t++) {
            rq[t] = make_uint4(0, 0, 0, 0);
            if (a0) rq[t] = *reinterpret_cast<const uint4*>(rp + 32 * t);
          }
        }
        conv_taps<NRA, MTALL, CB, 1, 1>(acc, s_ring + (g % NSLOT) * SLOT_N, wfrag, rbase, mt, px, kb, lane, 4 * sl);
        if constexpr (X3) {
          if (sq < NSLAB) conv_taps<NRA, MTALL, CB, 1, 1>(acc, s_ring + (g % NSLOT) * SLOT_N, wfrag2, rbase, mt, px, kb, lane, 4 * sl);
        }
      }
      PC_TOCK(1)
      if (X3 || NSLAB > 1 || pass == NPASS - 1) {
        pc_barrier_lds();
        g++;
      }
      PC_TOCK(2)
    }
    if (NR > 0) {
#pragma unroll
      for (int j = 0; j < NR; j++) {
        const bool act = (rmask[j] >> px) & 1u;
        if constexpr (X3) {
          store_tile_f32<COUT>(acc[j][0], act, reinterpret_cast<float*>(y_t) + ((int64_t)rrow[j] * W) * COUT + mt * 32, n_valid, px, kb);
          continue;
        }
        uint4 rc[2];
        if (HAS_RES) {
          rc[0] = rq[0], rc[1] = rq[1];
          if (j + 1 < NR) {  // the next row's lines before this row is packed and stored: never a load behind a store it must wait for
            const bool a1 = (rmask[j + 1] >> px) & 1u;
            const char* rp = reinterpret_cast<const char*>(res_t + ((int64_t)rrow[j + 1] * W) * COUT + mt * 32) + res_off;
#pragma unroll
            for (int t = 0; t < 2; t++) {
              rq[t] = make_uint4(0, 0, 0, 0);
              if (a1) rq[t] = *reinterpret_cast<const uint4*>(rp + 32 * t);
            }
          }
        }
        uint4 pk[2];
        pack_tile(acc[j][0], act, relu, pk, HAS_RES ? rc : nullptr);
        // (measured: storing each lane's own two 16-byte chunks instead -- 32 lines of 32 bytes per instruction, no transposition -- is 1 % slower end to end
        // and the epilogue's share stays at 20 %: it is the stores' issue, not the transposition's VALU: profiles/r06_conv_pc_ab.txt)
        transpose_row32(pk, lane);
        store_row32<COUT>(pk, y_t + ((int64_t)rrow[j] * W) * COUT + mt * 32, n_valid, lane);
      }
    }
    PC_TOCK(3)
  }
}

template <int CIN, int COUT, bool HAS_RES, bool X3 = false>
__global__ __launch_bounds__(768, 3) void k_conv3x3_pc(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2, const uint4* __restrict__ wfrag,
                                                       const uint4* __restrict__ wfrag2, const float* __restrict__ bias,
                                                       const uint16_t* __restrict__ res, const uint8_t* __restrict__ mask, uint16_t* __restrict__ y,
                                                       int B, int H, int W, int relu, uint8_t* __restrict__ row_dirty, int slot,
                                                       const int32_t* __restrict__ tlist, const int32_t* __restrict__ tcount) {
  constexpr int TH = CIN == 64 ? 16 : 8;
  constexpr int NRG = TH / 4, NCG = 8 / NRG, PASS_C = NCG * 32, NPASS = COUT / PASS_C, NSLAB = CIN / 64;
  constexpr int NSLOT = CIN == 64 ? 2 : 3, DEPTH = NSLOT - 1;
  constexpr int NFILL = X3 ? NPASS * NSLAB * 2 : CIN == 64 ? 1 : NPASS * NSLAB;  // fills per tile with an active site
  constexpr int YS = X3 ? 2 : 1;                                                 // output element in units of uint16_t
  static_assert(!X3 || (!HAS_RES && (CIN > 64 || NPASS == 1)), "fp32 product: no residual; one pass when the tile is filled once per source");
  constexpr int SLOT_N = (TH + 2) * LDS_HW * 8;
  static_assert(COUT % PASS_C == 0 && CIN % 64 == 0, "passes of NCG x 32 output channels over 64-channel input slabs");
  __shared__ uint4 s_ring[NSLOT * SLOT_N];
  __shared__ int4 s_tile[4];               // this workgroup's n-th tile (n & 3): {image b (-1 = no more tiles), y0, x0, rows with an active site}
  __shared__ uint32_t s_rowmask[4][TH];    // its active-site masks, one word per row
  // the deal of the tile's active rows to the row groups, done ONCE by producer 0 (round 6: every consumer wave redoing it from the row masks was 11 % of the
  // consumers' time): group rg takes the active rows number rg, rg + NRG, ...: row index (-1: none) and column mask of its j-th row
  __shared__ int4 s_drow[4][NRG];
  __shared__ uint4 s_dmask[4][NRG];
  __shared__ unsigned int s_tick[4];       // ticket that names tile n (n & 3), drawn three tiles earlier
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tiles_x = (W + 31) >> 5, tiles_y = (H + TH - 1) / TH;
  const int64_t n_tiles = tlist != nullptr ? (int64_t)__builtin_amdgcn_readfirstlane(*tcount) : (int64_t)B * tiles_y * tiles_x;
  PC_T_DECL

  if (wv >= 8) {
    // ------------------------------------------------------------------------------------------------ producers
    const int pw = wv - 8;
    // Staging pattern: halo row r = 272 consecutive 16-byte slots (34 columns x 8), DMA instruction i covers slots 64 i .. 64 i + 63 (i = 4: the 16
    // slots of columns 32, 33).  Slot k of column c receives chunk k ^ swz(c) of the pixel's line: the swizzle is applied on the source side.
    uint32_t voff[5];
    int colv[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const int col = 8 * i + (lane >> 3), k = lane & 7;
      colv[i] = col;
      voff[i] = (uint32_t)(col * CIN + ((k ^ lds_swz(col)) * 8)) * 2u;
    }
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    auto tile_index = [&](int n) -> int64_t {  // id of this workgroup's n-th tile
      const int64_t idx = (n < 3 || slot < 0) ? (int64_t)blockIdx.x + (int64_t)n * gridDim.x : (int64_t)s_tick[n & 3] + 3 * (int64_t)gridDim.x;
      return tile_at(tlist, idx, n_tiles);
    };
    uint32_t mb[TH / 2];  // mask bytes of the tile whose meta data is in flight: rows 2k + (lane >> 5), pixel lane & 31
    uint32_t dirty = 0;   // row_dirty flag of row `lane`
    auto load_meta = [&](int64_t t) {
#pragma unroll
      for (int k = 0; k < TH / 2; k++) mb[k] = 0;
      dirty = 0;
      if (t < 0) return;
      const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y), b = (int)(t / ((int64_t)tiles_x * tiles_y));
      const int ox = tx * 32 + (lane & 31);
#pragma unroll
      for (int k = 0; k < TH / 2; k++) {
        const int oy = ty * TH + 2 * k + (lane >> 5);
        if (ox < W && oy < H) mb[k] = mask != nullptr ? (uint32_t)mask[((int64_t)b * H + oy) * W + ox] : 1u;
      }
      const int oy2 = ty * TH + lane;
      if (lane < TH && oy2 < H) dirty = row_dirty != nullptr ? (uint32_t)row_dirty[((int64_t)b * H + oy2) * tiles_x + tx] : 1u;
    };

    int n = 0, f = 0, nf = 0, g = 0;
    int64_t t_next = tile_index(0);
    load_meta(t_next);
    // current tile
    int cb = 0, cy0 = 0, cx0 = 0;
    uint32_t am = 0;

    auto advance = [&]() -> bool {
      const int64_t t = t_next;
      if (t < 0) {
        if (pw == 0 && lane == 0) s_tile[n & 3] = make_int4(-1, 0, 0, 0);
        return false;
      }
      const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y);
      cb = (int)(t / ((int64_t)tiles_x * tiles_y));
      cx0 = tx * 32, cy0 = ty * TH;
      // row masks from the bytes requested one tile ago
      uint32_t my_rm = 0;  // lane r < TH: mask word of row r
      am = 0;
#pragma unroll
      for (int k = 0; k < TH / 2; k++) {
        const unsigned long long bal = __ballot(mb[k] != 0);
        const uint32_t lo = (uint32_t)bal, hi = (uint32_t)(bal >> 32);
        if (lane == 2 * k) my_rm = lo;
        if (lane == 2 * k + 1) my_rm = hi;
        am |= (lo != 0 ? 1u : 0u) << (2 * k) | (hi != 0 ? 1u : 0u) << (2 * k + 1);
      }
      const uint32_t was = (uint32_t)__ballot(dirty != 0) & ((1u << TH) - 1u);
      if (pw == 0) {
        if (lane < TH) s_rowmask[n & 3][lane] = my_rm;
        if (lane == 0) s_tile[n & 3] = make_int4(cb, cy0, cx0, (int)am);
        if (lane < NRG * 4) {  // lane = position in the list of active rows = rg + j * NRG
          uint32_t rest = am;
          for (int k = 0; k < lane && rest; k++) rest &= rest - 1;
          const int row = rest != 0 ? __builtin_ctz(rest) : -1;
          const uint32_t m = row >= 0 ? s_rowmask[n & 3][row] : 0u;  // written above by this wave: LDS operations of a wave execute in order
          reinterpret_cast<int*>(&s_drow[n & 3][lane % NRG])[lane / NRG] = row;
          reinterpret_cast<uint32_t*>(&s_dmask[n & 3][lane % NRG])[lane / NRG] = m;
        }
      }
      // The ticket that will name tile n + 3, published at once (the producers have the slack for the atomic's round trip).  Slot (n + 3) & 3 was last read at
      // advance(n - 2) and is next read at advance(n + 2): at most two advances run without a barrier between them (the ring's prologue fill and the first
      // fill of the loop), so a barrier separates the write from both.  (A deferred publish in the main loop lost the first ticket when tile 0 of a
      // three-slot ring had no active row: two advances, one publish.)
      if (slot >= 0 && pw == 0 && lane == 0) s_tick[(n + 3) & 3] = atomicAdd(&g_tile_ctr[slot][0], 1u);
      // next tile's meta data: requested now, consumed at the next advance
      t_next = tile_index(n + 1);
      load_meta(t_next);
      // rows of this producer (r = pw, pw + 4, ...): zero-fill what went inactive, keep row_dirty current
#pragma unroll 1
      for (int r = pw; r < TH; r += 4) {
        const int oy = cy0 + r;
        if (oy >= H) break;
        const bool active = (am >> r) & 1u, w = (was >> r) & 1u;
        if (!active && w) {
          char* dst = reinterpret_cast<char*>(y + (((int64_t)cb * H + oy) * W + cx0) * COUT * YS);
          const int nbytes = (W - cx0 < 32 ? W - cx0 : 32) * COUT * 2 * YS;
#pragma unroll 1
          for (int o = lane * 16; o < nbytes; o += 1024) *reinterpret_cast<uint4*>(dst + o) = make_uint4(0, 0, 0, 0);
        }
        if (row_dirty != nullptr && lane == 0 && w != active) row_dirty[((int64_t)cb * H + oy) * tiles_x + tx] = active ? 1 : 0;
      }
      nf = am != 0 ? NFILL : 1;
      f = 0;
      n++;
      return true;
    };

    auto next_fill = [&]() -> bool {
      if (f == nf && !advance()) return false;
      if (am != 0) {
        const int sl = f % NSLAB;
        uint4* dstslot = s_ring + (g % NSLOT) * SLOT_N;
        const uint16_t* xs = X3 && (f / NSLAB) % 2 == 1 ? x2 : x;  // fp32 product: per pass the NSLAB slabs of the high halves, then those of the low halves
        const uint16_t* xt = xs + (((int64_t)cb * H + (cy0 - 1)) * W + (cx0 - 1)) * CIN + 64 * sl;  // element (0, 0) of the halo tile
        bool col_ok[5];
#pragma unroll
        for (int i = 0; i < 5; i++) col_ok[i] = (unsigned)(cx0 - 1 + colv[i]) < (unsigned)W && (i < 4 || lane < 16);
#pragma unroll 1
        for (int r = pw; r < TH + 2; r += 4) {
          const int iy = cy0 - 1 + r;
          const bool row_ok = iy >= 0 && iy < H;  // wave-uniform
          const char* rowp = reinterpret_cast<const char*>(xt + (int64_t)r * W * CIN);
#pragma unroll
          for (int i = 0; i < 5; i++) {
            if (row_ok && col_ok[i]) {
              __builtin_amdgcn_global_load_lds((gptr_t)(rowp + voff[i]), (lptr_t)(dstslot + r * LDS_HW * 8 + 64 * i), 16, 0, 0);
            } else if (i < 4 || lane < 16) {
              dstslot[r * LDS_HW * 8 + 64 * i + lane] = make_uint4(0, 0, 0, 0);
            }
          }
        }
      }
      f++;
      g++;
      return true;
    };

    bool alive = true;
    int nbar = 0, nbar_total = 0x7fffffff;
    PC_TOCK(7)
    for (int d = 0; d < DEPTH - 1 && alive; d++) {
      alive = next_fill();
      if (!alive) nbar_total = g + 1;
    }
    for (;;) {
      if (alive) {
        alive = next_fill();
        if (!alive) nbar_total = g + 1;
      }
      PC_TOCK(5)
      pc_barrier_all();
      PC_TOCK(6)
      if (++nbar == nbar_total) break;
    }
    PC_T_FLUSH
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumers
  const int rg = wv % NRG, cg = wv / NRG;
  const int px = lane & 31, kb = lane >> 5;
  pc_barrier_lds();  // fill 0 has landed, tile 0's masks are published
  int g = 0;
  PC_TOCK(7)
  for (int n = 0;; n++) {
    const int4 ti = s_tile[n & 3];
    const int b = __builtin_amdgcn_readfirstlane(ti.x);
    if (b < 0) break;
    const int y0 = __builtin_amdgcn_readfirstlane(ti.y), x0 = __builtin_amdgcn_readfirstlane(ti.z);
    if (__builtin_amdgcn_readfirstlane(ti.w) == 0) {  // a tile of the list that only had stale rows: the producers zero-filled them
      pc_barrier_lds();
      g++;
      PC_TOCK(2)
      continue;
    }
    const int4 dr = s_drow[n & 3][rg];
    const uint4 dm = s_dmask[n & 3][rg];
    int rrow[4] = {__builtin_amdgcn_readfirstlane(dr.x), __builtin_amdgcn_readfirstlane(dr.y), __builtin_amdgcn_readfirstlane(dr.z),
                   __builtin_amdgcn_readfirstlane(dr.w)};
    const uint32_t rmask[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)dm.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)dm.y),
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)dm.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)dm.w)};
    int nr = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      nr += rrow[j] >= 0 ? 1 : 0;
      rrow[j] = rrow[j] >= 0 ? rrow[j] : 0;  // dummy rows point at row 0
    }
    const uint16_t* res_t = HAS_RES ? res + (((int64_t)b * H + y0) * W + x0) * COUT : nullptr;
    uint16_t* y_t = y + (((int64_t)b * H + y0) * W + x0) * COUT * YS;
    const int n_valid = W - x0;
    PC_TOCK(0)
#ifdef PNX_CONV_TIMERS
#define PC_ROWS(N_) pc_rows<N_, CIN, COUT, HAS_RES, NRG, NSLOT, SLOT_N, X3>(s_ring, g, wfrag, wfrag2, bias, res_t, y_t, rrow, rmask, W, n_valid, cg, relu, px, kb, lane, pc_T, pc_tk)
#else
#define PC_ROWS(N_) pc_rows<N_, CIN, COUT, HAS_RES, NRG, NSLOT, SLOT_N, X3>(s_ring, g, wfrag, wfrag2, bias, res_t, y_t, rrow, rmask, W, n_valid, cg, relu, px, kb, lane)
#endif
    switch (nr) {  // wave-uniform; every case runs the same barriers
      case 0: PC_ROWS(0); break;
      case 1: PC_ROWS(1); break;
      case 2: PC_ROWS(2); break;
      case 3: PC_ROWS(3); break;
      default: PC_ROWS(4); break;
    }
#undef PC_ROWS
  }
  if (threadIdx.x == 0) sched_done(slot);
  PC_T_FLUSH
}

template <int CIN, int COUT>
int launch_pc(const void* x, const void* wfrag, const float* bias, const void* res, const uint8_t* mask, void* y, int B, int H, int W, int relu,
              uint8_t* row_dirty, const int32_t* tlist, const int32_t* tcount, hipStream_t st) {
  constexpr int TH = CIN == 64 ? 16 : 8;
  int64_t nb = (int64_t)B * ((H + TH - 1) / TH) * ((W + 31) / 32);
  if (nb > 256) nb = 256;  // one 12-wave workgroup per CU (LDS: the ring takes 128-153 KiB)
  const int slot = mask != nullptr ? next_sched_slot() : -1;
  if (res != nullptr)
    k_conv3x3_pc<CIN, COUT, true><<<(unsigned)nb, 768, 0, st>>>((const uint16_t*)x, nullptr, (const uint4*)wfrag, nullptr, bias, (const uint16_t*)res, mask, (uint16_t*)y, B, H,
                                                              W, relu, row_dirty, slot, tlist, tcount);
  else
    k_conv3x3_pc<CIN, COUT, false><<<(unsigned)nb, 768, 0, st>>>((const uint16_t*)x, nullptr, (const uint4*)wfrag, nullptr, bias, nullptr, mask, (uint16_t*)y, B, H, W, relu,
                                                               row_dirty, slot, tlist, tcount);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

// fp32 product of bf16 pairs (pnx_conv3x3_x3): x = x_hi + x_lo, W = W_hi + W_lo, y = x_hi W_hi + x_hi W_lo + x_lo W_hi in one launch, fp32 out.
template <int CIN, int COUT>
int launch_pc_x3(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, const uint8_t* mask, float* y, int B, int H, int W,
                 hipStream_t st) {
  constexpr int TH = CIN == 64 ? 16 : 8;
  int64_t nb = (int64_t)B * ((H + TH - 1) / TH) * ((W + 31) / 32);
  if (nb > 256) nb = 256;
  const int slot = mask != nullptr ? next_sched_slot() : -1;
  k_conv3x3_pc<CIN, COUT, false, true><<<(unsigned)nb, 768, 0, st>>>((const uint16_t*)x_hi, (const uint16_t*)x_lo, (const uint4*)w_hi, (const uint4*)w_lo,
                                                                     bias, nullptr, mask, (uint16_t*)y, B, H, W, 0, nullptr, slot, nullptr, nullptr);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}
