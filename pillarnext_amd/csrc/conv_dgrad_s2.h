// conv_dgrad_s2.h -- data gradient of the sparse backbone's stride-2 entry convolutions in training (gfx950; included by conv3x3.hip, bf16 build only).
//
// Reference: det3d/models/utils/sparse_conv.py:16-39 under autograd -- SparseConv2d(k = 3, stride 2, pad 1): y[oy][ox] = sum W[ky][kx] x[2 oy + ky - 1][2 ox + kx - 1],
// so the gradient at input site (iy, ix) collects the output sites with 2 oy + ky - 1 = iy.  By the parity of (iy, ix) = (2a + py, 2b + qx):
//     py = 0:  ky = 1 (oy = a)                     py = 1:  ky = 0 (oy = a + 1),  ky = 2 (oy = a)          -- the same in x
// i.e. FOUR small convolutions of the upstream gradient g (1, 2, 2 and 4 taps: nine (tap, parity) products in all, exactly the forward's work), each writing one
// parity plane of dx.  MIOpen / CK run this as a dense transposed convolution (3.0 ms for 64 -> 128 at 1440^2 x 4 frames in fp32, 1.3 ms in bf16); here:
//   tile      TG rows x 32 columns of g cells = 2 TG x 64 input sites; g halo tile (TG + 1) x 33 cells x CO channels staged once in LDS (64-channel slabs,
//             16-byte chunks swizzled as in the forward kernels); tiles without an active INPUT site are zero-filled and skipped (the gradient is only needed there)
//   wave      two g rows x four parities x one 32-channel tile of dx (8 accumulators, 128 registers): per 16-channel k-step 6 B fragments from LDS
//             (3 rows x 2 column offsets), 9 weight fragments (one per tap, from the TRANSPOSED pack pnx_conv3x3_pack_weights(transposed=1): M = ci, K = co,
//             tap index (2 - ky) * 3 + (2 - kx)) and 18 MFMAs; 8 waves = (TG / 2) row pairs x (8 / (TG / 2)) channel tiles = all of CI
//   epilogue  pack_tile (8 consecutive channels per lane), zeros at inactive input sites, 16-byte stores (fp32 out for the three-product form)
// X3: the fp32 graph's form (pnx_conv3x3_x3's companion): g = g_hi + g_lo, W = W_hi + W_lo; the tile of g_hi runs with both weight halves, then the tile of
// g_lo with W_hi, the accumulators running through; fp32 output.

template <int CO, int CI, bool X3>
struct Dg2Geo {
  static constexpr int MT = CI / 32;             // 32-channel tiles of dx
  static constexpr int NRP = 8 / MT;             // row pairs per tile (8 waves)
  static constexpr int TG = 2 * NRP;             // g rows per tile
  static constexpr int NSLAB = CO / 64;
  static constexpr int ROW_N = LDS_HW * 8;       // uint4 per staged row of one slab
  static constexpr int SLAB_N = (TG + 1) * ROW_N;
  static constexpr int LDS_BYTES = NSLAB * SLAB_N * 16;
  static_assert(MT == 2 || MT == 4 || MT == 8, "CI = 64, 128 or 256");
};

template <int CO, int CI, bool X3>
__global__ __launch_bounds__(512, 2) void k_dgrad_s2(const uint16_t* __restrict__ g, const uint16_t* __restrict__ g2, const uint4* __restrict__ wt,
                                                     const uint4* __restrict__ wt2, const uint8_t* __restrict__ mask_in, void* __restrict__ dx, int B, int H, int W,
                                                     int Ho, int Wo, int slot) {
  using G = Dg2Geo<CO, CI, X3>;
  constexpr int MT = G::MT, NRP = G::NRP, TG = G::TG, NSLAB = G::NSLAB, CB = CO / 16;
  extern __shared__ uint4 s_g[];  // [slab][row 0..TG][col 0..33][8 chunks]
  __shared__ unsigned int s_next[2];
  __shared__ unsigned int s_any;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int px = lane & 31, kb = lane >> 5;
  const int rp = wv % NRP, mt = wv / NRP;  // this wave's row pair and channel tile
  const int tiles_x = (Wo + 31) >> 5, tiles_y = (Ho + TG - 1) / TG;
  const int64_t n_tiles = (int64_t)B * tiles_y * tiles_x;
  int64_t next = 0;
  int it = 0;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile = next, it++) {
    sched_draw(s_next, it, slot);
    if (t == 0) s_any = 0u;
    const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y), b = (int)(tile / ((int64_t)tiles_x * tiles_y));
    const int a0 = ty * TG, b0 = tx * 32;  // first g cell of the tile; input sites (2 a0 .., 2 b0 ..)
    __syncthreads();                       // s_any cleared; everybody is done with the previous tile's LDS image
    // ---- active input sites of the tile: 2 TG rows x 64 columns of mask bytes, one 16-byte piece per thread (2 TG * 4 <= 64 pieces)
    uint32_t any = 0;
    if (t < 2 * TG * 4) {
      const int r = t >> 2, q = t & 3;
      const int iy = 2 * a0 + r, ix = 2 * b0 + 16 * q;
      if (iy < H && ix < W) {
        const uint8_t* mp = mask_in + ((int64_t)b * H + iy) * W + ix;
        if (ix + 16 <= W && (((uintptr_t)mp) & 15) == 0) {
          const uint4 m = *reinterpret_cast<const uint4*>(mp);
          any = m.x | m.y | m.z | m.w;
        } else {
          for (int k = 0; k < 16 && ix + k < W; k++) any |= mp[k];
        }
      }
    }
    if (any != 0) s_any = 1u;  // benign race: every writer stores the same value
    __syncthreads();
    next = sched_next(s_next, it, slot, tile);
    const bool live = s_any != 0u;
    if (!live) {  // no active input site: the gradient is not needed here -- zeros (every site of dx is written)
      constexpr int ESZ = X3 ? 4 : 2;
      const int rows = min(2 * TG, H - 2 * a0), cols = min(64, W - 2 * b0);
      const int row_bytes = cols * CI * ESZ;
      for (int r = wv; r < rows; r += 8) {
        char* dst = reinterpret_cast<char*>(dx) + (((int64_t)b * H + 2 * a0 + r) * W + 2 * b0) * CI * ESZ;
        for (int o = lane * 16; o < row_bytes; o += 1024) *reinterpret_cast<uint4*>(dst + o) = make_uint4(0, 0, 0, 0);
      }
      continue;
    }
    v16f acc[2][4];  // [row of the pair][parity 2 py + qx]
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[j][p][i] = 0.f;

#pragma unroll 1
    for (int src = 0; src < (X3 ? 2 : 1); src++) {
      const uint16_t* gs = src == 0 ? g : g2;
      if (src) __syncthreads();  // the first image has been consumed
      // ---- stage the g halo tile: rows a0 .. a0 + TG, columns b0 .. b0 + 32, all CO channels; zeros outside the map
      constexpr int NCH = (TG + 1) * 33 * (CO / 8);
      for (int e = t; e < NCH; e += 512) {
        const int ch = e % (CO / 8), cell = e / (CO / 8), col = cell % 33, row = cell / 33;
        const int oy = a0 + row, ox = b0 + col;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (oy < Ho && ox < Wo) q = *reinterpret_cast<const uint4*>(gs + (((int64_t)b * Ho + oy) * Wo + ox) * CO + ch * 8);
        s_g[(ch >> 3) * G::SLAB_N + row * G::ROW_N + col * 8 + ((ch & 7) ^ lds_swz(col))] = q;
      }
      __syncthreads();
      // ---- k loop: slabs of 64 channels x 4 k-steps of 16
#pragma unroll 1
      for (int ks = 0; ks < CB; ks++) {
        const int slab = ks >> 2, cbl = ks & 3;
        const uint4* sb = s_g + slab * G::SLAB_N + (2 * rp) * G::ROW_N;
        el8 q[3][2];  // [g row a, a + 1, a + 2 of the pair's window][column offset 0, 1]
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int d = 0; d < 2; d++) {
            const int c = px + d;
            q[r][d] = __builtin_bit_cast(el8, sb[r * G::ROW_N + c * 8 + ((2 * cbl + kb) ^ lds_swz(c))]);
          }
#pragma unroll
        for (int half = 0; half < (X3 ? 2 : 1); half++) {
          if (X3 && half == 1 && src == 1) break;  // g_lo runs with W_hi only
          const uint4* wp = (half == 0 ? wt : wt2) + ((int64_t)ks * MT + mt) * 64 + lane;
          el8 w[9];  // by (ky, kx): fragment of tap (2 - ky) * 3 + (2 - kx) of the transposed pack
#pragma unroll
          for (int k = 0; k < 9; k++) w[k] = __builtin_bit_cast(el8, wp[(int64_t)(8 - k) * CB * MT * 64]);
#pragma unroll
          for (int j = 0; j < 2; j++) {
            // parity (0,0): (ky,kx) = (1,1) at (a, b)
            acc[j][0] = PNX_MFMA32(w[4], q[j][0], acc[j][0]);
            // parity (0,1): kx = 0 at b + 1, kx = 2 at b
            acc[j][1] = PNX_MFMA32(w[3], q[j][1], acc[j][1]);
            acc[j][1] = PNX_MFMA32(w[5], q[j][0], acc[j][1]);
            // parity (1,0): ky = 0 at a + 1, ky = 2 at a
            acc[j][2] = PNX_MFMA32(w[1], q[j + 1][0], acc[j][2]);
            acc[j][2] = PNX_MFMA32(w[7], q[j][0], acc[j][2]);
            // parity (1,1)
            acc[j][3] = PNX_MFMA32(w[0], q[j + 1][1], acc[j][3]);
            acc[j][3] = PNX_MFMA32(w[2], q[j + 1][0], acc[j][3]);
            acc[j][3] = PNX_MFMA32(w[6], q[j][1], acc[j][3]);
            acc[j][3] = PNX_MFMA32(w[8], q[j][0], acc[j][3]);
          }
        }
      }
    }
    // ---- epilogue: parity planes of the pair's two g rows
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int a = a0 + 2 * rp + j;
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const int iy = 2 * a + (p >> 1), ix = 2 * (b0 + px) + (p & 1);
        const bool inb = iy < H && ix < W;
        const bool act = inb && mask_in[((int64_t)b * H + iy) * W + ix] != 0;
        if constexpr (X3) {
          float* row = reinterpret_cast<float*>(dx) + (((int64_t)b * H + iy) * W + ix) * CI + mt * 32 + 4 * kb;
          if (inb) {
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
              float4 v = make_float4(acc[j][p][4 * gq], acc[j][p][4 * gq + 1], acc[j][p][4 * gq + 2], acc[j][p][4 * gq + 3]);
              if (!act) v = make_float4(0.f, 0.f, 0.f, 0.f);
              *reinterpret_cast<float4*>(row + 8 * gq) = v;
            }
          }
        } else {
          uint4 pk[2];
          pack_tile(acc[j][p], act, 0, pk);  // all 64 lanes (the half-wave swap); pk[t] = channels 16 t + 8 kb + 0..7
          if (inb) {
            uint16_t* row = reinterpret_cast<uint16_t*>(dx) + (((int64_t)b * H + iy) * W + ix) * CI + mt * 32 + 8 * kb;
            *reinterpret_cast<uint4*>(row) = pk[0];
            *reinterpret_cast<uint4*>(row + 16) = pk[1];
          }
        }
      }
    }
  }
  sched_done(slot);
}

template <int CO, int CI, bool X3>
int launch_dgrad_s2(const void* g, const void* g2, const void* wt, const void* wt2, const uint8_t* mask_in, void* dx, int B, int H, int W, hipStream_t st) {
  using G = Dg2Geo<CO, CI, X3>;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  int64_t nb = (int64_t)B * ((Ho + G::TG - 1) / G::TG) * ((Wo + 31) / 32);
  const int per_cu = G::LDS_BYTES <= 75 * 1024 ? 2 : 1;
  if (nb > 256 * per_cu) nb = 256 * per_cu;
  auto kern = k_dgrad_s2<CO, CI, X3>;
  static bool attr_done = false;
  if (!attr_done) {
    PNX_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr_done = true;
  }
  const int slot = next_sched_slot();
  kern<<<(unsigned)nb, 512, G::LDS_BYTES, st>>>((const uint16_t*)g, (const uint16_t*)g2, (const uint4*)wt, (const uint4*)wt2, mask_in, dx, B, H, W, Ho, Wo, slot);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}
