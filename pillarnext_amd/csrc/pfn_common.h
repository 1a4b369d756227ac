// pfn_common.h -- helpers shared by the PFN kernels over sorted records (pfn_v3.hip: records in HBM; pfn_spans.hip: records sorted
// in LDS by the same workgroup).  Reference arithmetic: pillar_encoder.py:35-50, :174-182.
#pragma once
#include "pnx_common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
#define PNX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
#define PNX_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ v8h as_v8h(const uint32_t* w) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 t = {w[0], w[1], w[2], w[3]};
  return __builtin_bit_cast(v8h, t);
}
// two non-negative fp32 values -> packed fp16 hi (round toward zero) and lo = fp16(x - hi) (x - hi is exact in fp32)
__device__ __forceinline__ void split2_f16(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  const auto hv = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  const float r0 = __fsub_rn(x0, (float)hv[0]), r1 = __fsub_rn(x1, (float)hv[1]);
  const auto lv = __builtin_amdgcn_cvt_pkrtz(r0, r1);
  hi = __builtin_bit_cast(uint32_t, hv);
  lo = __builtin_bit_cast(uint32_t, lv);
}

constexpr int kZS = 68;                 // words per LDS row (64 channels + 4: rows stay 16-byte aligned, 8 consecutive rows cover all banks)
constexpr int kZSP = 36;                // the same for rows of 64 16-bit values (128 bytes + 16)
constexpr int kWaveLds = 32 * kZS + 64;  // per wave: 32 pillar rows + rank[32] + cell[32], in words

struct Pfn3Out {
  float* g1;  // (rows, 64) fp32 or null
  int64_t g1_rows;
  void* canvas;  // NHWC canvas or null
  int dt;        // PNX_F32 / PNX_BF16 / PNX_F16
  const int32_t* row_of = nullptr;  // k_pfn3_tail behind pfn_spans.hip: feat_max row of a spill id (null: the pillar id is the row)
  int nt = 0;  // canvas rows as nontemporal stores (large canvases, reader.hip): nobody re-reads them before the caches have turned over
};

__device__ __forceinline__ uint32_t bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);  // inputs are finite post-ReLU values
  return u >> 16;
}
__device__ __forceinline__ uint32_t f16_rne(float f) {
  const _Float16 hv = (_Float16)f;
  return (uint32_t)__builtin_bit_cast(unsigned short, hv);
}

// the two 16-byte quads of one half of a sorted record (reader_bins.h)
struct Half {
  uint4 a, b;  // a = f[h], f[2+h], f[4+h], f[6+h]   b = f[8+h], f[10+h], aux, (h ? cell : rank)
};
__device__ __forceinline__ Half load_half(const uint4* __restrict__ rec, uint32_t slot, int h) {
  const uint4* p = rec + (int64_t)slot * 4 + 2 * h;
  Half r;
  r.a = p[0];
  r.b = p[1];
  return r;
}

// First slot of the pillar after the one that contains `slot`.
__device__ __forceinline__ uint32_t pillar_end_at(const uint4* __restrict__ rec, int64_t slot, const uint32_t* __restrict__ pfirst,
                                                  const uint32_t* __restrict__ pcnt, bool* is_head) {
  const uint4 w = rec[slot * 4 + 1];
  const uint32_t idx = w.z & 0xFFFFu, rem = w.z >> 16;
  *is_head = idx == 0;
  if (idx < 0xFFFFu && rem < 0xFFFFu) return (uint32_t)slot + rem + 1u;
  return pfirst[w.w] + pcnt[w.w];  // 16-bit fields saturated: a pillar with >= 65535 points
}

// wave-synchronous LDS exchange: the LDS unit executes one wave's DS instructions in order; the fences only pin the compiler
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// two fp32 -> two 16-bit floats of the canvas dtype, round-to-nearest-even, low half = first argument
template <int DT>
__device__ __forceinline__ uint32_t cvt_pk16(float a, float b) {
  uint32_t r;
  if (DT == PNX_BF16) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// 16 bytes of a canvas line.  Nontemporal on large canvases: the pillar lines are scattered single lines that nothing re-reads soon,
// and as plain stores they sit in L2 until evicted, between the zero-fill's streaming stores -- the reader's two writers then slow
// each other down (C2 x 12 frames: 725 -> 627 us with the nontemporal form, profiles/r04_reader_ab.txt).
__device__ __forceinline__ void canvas_store16(void* dst, const uint4& x, int nt) {
  if (nt) {  // asm: as two builtin stores to one address the branches are merged into a plain store (the nontemporal flag is dropped)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {x.x, x.y, x.z, x.w};
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
  } else {
    *reinterpret_cast<uint4*>(dst) = x;
  }
}

// 8 consecutive channels (chan0 = 8q) of one pillar: canvas cell and/or feat_max row
template <int DT>
__device__ __forceinline__ void store_chunk(const Pfn3Out& o, int rank, int64_t cell, int q, const float* v) {
  if (o.g1 != nullptr && (int64_t)rank < o.g1_rows) {
    float4* d = reinterpret_cast<float4*>(o.g1 + (int64_t)rank * 64 + 8 * q);
    d[0] = make_float4(v[0], v[1], v[2], v[3]);
    d[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
  if (o.canvas != nullptr) {
    if (DT == PNX_F32) {
      float* d = reinterpret_cast<float*>(o.canvas) + cell * 64 + 8 * q;
      canvas_store16(d, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])), o.nt);
      canvas_store16(d + 4, make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])), o.nt);
    } else {
      uint4 p;
      if (DT == PNX_BF16) {
        p.x = bf16_rne(v[0]) | (bf16_rne(v[1]) << 16);
        p.y = bf16_rne(v[2]) | (bf16_rne(v[3]) << 16);
        p.z = bf16_rne(v[4]) | (bf16_rne(v[5]) << 16);
        p.w = bf16_rne(v[6]) | (bf16_rne(v[7]) << 16);
      } else {
        p.x = f16_rne(v[0]) | (f16_rne(v[1]) << 16);
        p.y = f16_rne(v[2]) | (f16_rne(v[3]) << 16);
        p.z = f16_rne(v[4]) | (f16_rne(v[5]) << 16);
        p.w = f16_rne(v[6]) | (f16_rne(v[7]) << 16);
      }
      canvas_store16(reinterpret_cast<uint16_t*>(o.canvas) + cell * 64 + 8 * q, p, o.nt);
    }
  }
}

// Window tickets without a stall: hipcc's atomic optimizer expands atomicAdd into ballot + one atomic + an immediate
// s_waitcnt vmcnt(0) + readfirstlane, i.e. a fabric round trip (and a drain of the record prefetch) in front of every window.
// The asm form returns into a VGPR that nobody reads until ticket_wait(); the hardware vmcnt only ever makes hipcc's own waits
// more conservative (MI355X guide 5.7: an uncounted asm memory op).
__device__ __forceinline__ int ticket_issue(int32_t* counter, int lane) {
  int ret = 0;
  const int one = 1;
  if (lane == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ret) : "v"(counter), "v"(one) : "memory");
  return ret;
}
// The wait and the read are ONE asm statement with the ticket as a plain input: written as "+v"(tk) followed by a readfirstlane the
// compiler was seen to copy the register in front of the s_waitcnt (a stale read if the atomic were still in flight).
__device__ __forceinline__ int ticket_wait(int tk) {
  int r;
  asm volatile("s_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 %0, %1" : "=s"(r) : "v"(tk) : "memory");
  return r;
}

}  // namespace
