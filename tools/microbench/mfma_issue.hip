// mfma_issue.hip -- what limits the tap loop of conv3x3.hip?  The loop body of conv_taps<NR=4> (8 x v_mfma_f32_32x32x16_bf16 per
// k-step, 4 ds_read_b128 of B fragments, 2 global_load_dwordx4 of weight fragments) rebuilt piece by piece on synthetic data:
//   mode bit 0: B fragments from LDS every k-step      bit 1: weight fragments from global memory (L1/L2) every k-step
//   bit 2: per-step address VALU like the real loop      bit 3: the real epilogue (pack, lane transpose, 128-byte-line stores) after
//   every 9 taps (= one 64-channel pass), accumulators reset      bit 4: the same without the stores
// Reported: MFMA pipe utilisation = 32 cycles x MFMAs / (SIMD cycles), for 1 and 2 workgroups (= waves per SIMD) per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/mfma_issue tools/microbench/mfma_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int kTile = 18 * 34 * 8;  // uint4: the 76.5 KiB halo tile of k_conv3x3_lds
__device__ __forceinline__ int swz(int c) { return (c & 7) ^ ((c >> 3) & 1); }
// ---- the epilogue helpers of conv3x3.hip, verbatim
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void pack_tile(const v16f& a, bool act, int relu, uint4 (&out)[2]) {
#pragma unroll
  for (int t = 0; t < 2; t++) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[8 * t + i]), __float_as_uint(a[8 * t + 4 + i]), false, false);
      v[i] = __uint_as_float(r.x);
      v[4 + i] = __uint_as_float(r.y);
    }
    if (relu) {
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = fmaxf(v[i], 0.f);
    }
    uint4 p;  // v_cvt_pk_bf16_f32: round-to-nearest-even, one instruction per channel pair
    p.x = pack_bf16(v[0], v[1]);
    p.y = pack_bf16(v[2], v[3]);
    p.z = pack_bf16(v[4], v[5]);
    p.w = pack_bf16(v[6], v[7]);
    if (!act) p = make_uint4(0, 0, 0, 0);
    out[t] = p;
  }
}

__device__ __forceinline__ void cswap(bool c, uint4& x, uint4& y) {
  const uint4 a = x, b = y;
  x.x = c ? b.x : a.x, x.y = c ? b.y : a.y, x.z = c ? b.z : a.z, x.w = c ? b.w : a.w;
  y.x = c ? a.x : b.x, y.y = c ? a.y : b.y, y.z = c ? a.z : b.z, y.w = c ? a.w : b.w;
}

// R[s], s = 2*(tile within the 64-channel group) + t: chunk 2s + kb of pixel px = lane & 31.  On return R[d] of lane L is chunk
// L & 7 of pixel 8d + (L >> 3).
__device__ __forceinline__ void transpose_row64(uint4 (&R)[4], int lane) {
  const int a_src = (lane & 31) >> 3;       // pixel octet of this lane as a source
  const int s_dst = (lane & 7) >> 1;        // slot this lane stores as a destination
  // rotate: U[k] = R[k ^ a_src]
  cswap(a_src & 1, R[0], R[1]);
  cswap(a_src & 1, R[2], R[3]);
  cswap(a_src & 2, R[0], R[2]);
  cswap(a_src & 2, R[1], R[3]);
  // permute: round k fetches slot register k of source lane (octet k ^ s_dst, pixel-in-octet L>>3, half L&1)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int src = (8 * (k ^ s_dst) + (lane >> 3)) + 32 * (lane & 1);
    R[k].x = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)R[k].x);
    R[k].y = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)R[k].y);
    R[k].z = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)R[k].z);
    R[k].w = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)R[k].w);
  }
  // rotate back: D[d] = V[d ^ s_dst]
  cswap(s_dst & 1, R[0], R[1]);
  cswap(s_dst & 1, R[2], R[3]);
  cswap(s_dst & 2, R[0], R[2]);
  cswap(s_dst & 2, R[1], R[3]);
}

// `row` (wave-uniform) points at channel 0 of the 64-channel group for pixel 0 of the 32-pixel row segment; pixels >= n_valid
// are not stored; CSTRIDE = channels per pixel.  Uniform base + 32-bit lane offset: no per-store 64-bit address arithmetic.
template <int CSTRIDE>
__device__ __forceinline__ void store_row64(const uint4 (&D)[4], uint16_t* __restrict__ row, int n_valid, int lane) {
  const uint32_t voff = (uint32_t)((lane >> 3) * CSTRIDE + (lane & 7) * 8) * 2u;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    const int P = 8 * d + (lane >> 3);
    if (P < n_valid) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(row) + voff + (uint32_t)(d * 8 * CSTRIDE * 2)) = D[d];
  }
}



template <int MODE>
__global__ __launch_bounds__(256, 2) void k_loop(const uint4* __restrict__ wfrag, float* __restrict__ out, int taps, int wperiod, uint16_t* __restrict__ ybuf) {
  __shared__ uint4 s_in[kTile];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, px = lane & 31, kb = lane >> 5;
  for (int i = threadIdx.x; i < kTile; i += 256) s_in[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  __syncthreads();
  v16f acc[4][2];
  for (int j = 0; j < 4; j++)
    for (int m = 0; m < 2; m++)
      for (int i = 0; i < 16; i++) acc[j][m][i] = 0.f;
  uint4 w[4][2], q[4];
  for (int c = 0; c < 4; c++)
    for (int m = 0; m < 2; m++) w[c][m] = wfrag[(c * 2 + m) * 64 + lane];
  for (int j = 0; j < 4; j++) q[j] = s_in[((wv * 4 + j) * 34 + px) * 8 + (kb ^ swz(px))];
  int rb[4];
  for (int j = 0; j < 4; j++) rb[j] = __builtin_amdgcn_readfirstlane((wv * 4 + j) * 34 * 8);
  for (int tap = 0; tap < taps; tap++) {
    const int t9 = tap % 9, dy = t9 / 3, dx = t9 - 3 * dy;
    int c = px + dx, cbase = (dy * 34 + c) * 8, sw = swz(c);
    if (!(MODE & 4)) c = px, cbase = px * 8, sw = swz(px);
#pragma unroll
    for (int cbl = 0; cbl < 4; cbl++) {
      if (MODE & 1) {
        const int chunk = ((cbl * 2) + kb) ^ sw;
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = s_in[rb[j] + cbase + chunk];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bf16x8 bfr = __builtin_bit_cast(bf16x8, q[j]);
#pragma unroll
        for (int m = 0; m < 2; m++) acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[cbl][m]), bfr, acc[j][m], 0, 0, 0);
      }
      if (MODE & 2) {
#pragma unroll
        for (int m = 0; m < 2; m++) w[cbl][m] = wfrag[(((tap % wperiod) * 4 + cbl) * 2 + m) * 64 + lane];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((MODE & 24) && t9 == 8) {  // end of a pass
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint4 D[4];
#pragma unroll
        for (int m = 0; m < 2; m++) {
          uint4 pk[2];
          pack_tile(acc[j][m], px != 7, 1, pk);
          D[2 * m] = pk[0], D[2 * m + 1] = pk[1];
        }
        transpose_row64(D, lane);
        if (MODE & 8) store_row64<64>(D, ybuf + ((size_t)(blockIdx.x * 16 + wv * 4 + j) * 2048 + (size_t)((tap / 9) & 63) * 1048576 * 16) % ((size_t)1 << 28), 32, lane);
        else if (D[0].x == 0x12345u) out[lane] = 1.f;
#pragma unroll
        for (int m = 0; m < 2; m++)
          for (int i = 0; i < 16; i++) acc[j][m][i] = (float)j;
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; j++)
    for (int m = 0; m < 2; m++)
      for (int i = 0; i < 16; i++) s += acc[j][m][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const uint4* w, float* out, int blocks, int taps, int wperiod, uint16_t* ybuf) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_loop<MODE><<<blocks, 256>>>(w, out, taps, wperiod, ybuf);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_loop<MODE><<<blocks, 256>>>(w, out, taps, wperiod, ybuf);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * 4 * taps * 4 * 8;
  const double tflops = mfma * 32768.0 / (ms * 1e-3) / 1e12;
  printf("mode %2d (%s%s%s%s%s) blocks %4d wperiod %3d: %8.1f us  %7.1f TFLOP/s = %4.1f %% of 2500\n", MODE, (MODE & 1) ? "lds " : "", (MODE & 2) ? "wgt " : "",
         (MODE & 4) ? "addr " : "", (MODE & 8) ? "epilogue " : "", (MODE & 16) ? "epilogue-without-stores " : "", blocks, wperiod, ms * 1e3, tflops, tflops / 25.0);
}

int main() {
  uint4* w;
  float* out;
  const int wper_max = 54;  // 54 taps x 4 x 2 KiB = 432 KiB: the weights of a 64 -> 384 convolution
  hipMalloc(&w, (size_t)wper_max * 4 * 2 * 64 * 16);
  hipMemset(w, 0x3c, (size_t)wper_max * 4 * 2 * 64 * 16);
  hipMalloc(&out, 1024 * 256 * 4);
  uint16_t* ybuf;
  hipMalloc(&ybuf, (size_t)1 << 30);
  const int taps = 9 * 64;
  for (int blocks : {256, 512}) {
    run<0>(w, out, blocks, taps, 9, ybuf);
    run<1>(w, out, blocks, taps, 9, ybuf);
    run<2>(w, out, blocks, taps, 54, ybuf);
    run<7>(w, out, blocks, taps, 54, ybuf);
    run<23>(w, out, blocks, taps, 54, ybuf);
    run<15>(w, out, blocks, taps, 54, ybuf);
  }
  return 0;
}
