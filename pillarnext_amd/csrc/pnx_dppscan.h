// pnx_dppscan.h -- segmented inclusive max across the 32 lanes of each wave half with DPP row shifts, for the PFN kernels
// (pfn_v3.hip): lane = point, the points of a pillar are adjacent lanes, idx = position inside the pillar.
#pragma once
#include "pnx_common.h"

namespace {

constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118, DPP_ROW_BCAST15 = 0x142;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {  // lanes without a source get 0.0
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// which scan steps the tile needs: s<d> set if some lane has idx >= d (wave-uniform)
struct ScanPlan {
  bool s1, s2, s4, s8;
};

// One Hillis-Steele step of a segmented inclusive max along the 32 lanes of each half, for N registers at once.
// Lanes whose DPP source is outside the 16-lane row receive their own value (old = v), which max() ignores.
template <int CTRL, int ROW_MASK, int N>
__device__ __forceinline__ void max_step(float* v, bool take) {
#pragma unroll
  for (int i = 0; i < N; i++) {
    const float m = fmaxf(dpp_f<CTRL, ROW_MASK>(v[i], v[i]), v[i]);  // one v_max_f32_dpp
    v[i] = take ? m : v[i];
  }
}
// idx = position of the lane's point inside its pillar: lane l-d belongs to the same pillar iff idx >= d.
template <int N>
__device__ __forceinline__ void seg_max_n(float* v, int idx, int col, const ScanPlan& pl) {
  if (pl.s1) max_step<DPP_ROW_SHR1, 0xF, N>(v, idx >= 1);
  if (pl.s2) max_step<DPP_ROW_SHR2, 0xF, N>(v, idx >= 2);
  if (pl.s4) max_step<DPP_ROW_SHR4, 0xF, N>(v, idx >= 4);
  if (pl.s8) max_step<DPP_ROW_SHR8, 0xF, N>(v, idx >= 8);
  if (pl.s1) max_step<DPP_ROW_BCAST15, 0xA, N>(v, idx > (col & 15));  // pillar straddling the two 16-lane rows of a half
}
// The same scan for NON-NEGATIVE values (post-ReLU): 0 is then the identity of max, so a lane that must not take its neighbour
// ANDs the shifted value with a per-step lane mask (one VGPR per step, shared by all registers) and lanes without a DPP source
// read 0 (bound_ctrl) -- two instructions per register and step, v_and_b32_dpp + v_max_f32, where the general form above
// compiles to five (v_mov_dpp, s_nop, v_max, v_cndmask, v_mov: 1 200 of the ~2 000 instructions of a tile).  hipcc does not form
// the DPP operand by itself here (it turns the AND back into a select), hence inline asm, eight registers per statement: every
// DPP read then sits >= 8 instructions behind the write of its register (the 2 wait states a VALU-write -> DPP-read needs), the
// leading s_nop covers the first one.  row_bcast:15 runs with all rows enabled: rows 0 and 2 receive 0 / the other half's
// lane 31, both discarded by the lane mask (idx <= column inside a tile).
#define PNX_AND_DPP(i, CTRL) "v_and_b32_dpp %[t" #i "], %[v" #i "], %[m] " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define PNX_MAXF(i) "v_max_f32 %[v" #i "], %[v" #i "], %[t" #i "]\n"
#define PNX_SCAN8(CTRL, v, m)                                                                                                       \
  {                                                                                                                                 \
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7;                                                                                        \
    asm volatile("s_nop 1\n" PNX_AND_DPP(0, CTRL) PNX_AND_DPP(1, CTRL) PNX_AND_DPP(2, CTRL) PNX_AND_DPP(3, CTRL) PNX_AND_DPP(4, CTRL)  \
                     PNX_AND_DPP(5, CTRL) PNX_AND_DPP(6, CTRL) PNX_AND_DPP(7, CTRL) PNX_MAXF(0) PNX_MAXF(1) PNX_MAXF(2) PNX_MAXF(3)   \
                         PNX_MAXF(4) PNX_MAXF(5) PNX_MAXF(6) PNX_MAXF(7)                                                             \
                 : [v0] "+v"((v)[0]), [v1] "+v"((v)[1]), [v2] "+v"((v)[2]), [v3] "+v"((v)[3]), [v4] "+v"((v)[4]), [v5] "+v"((v)[5]),  \
                   [v6] "+v"((v)[6]), [v7] "+v"((v)[7]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),               \
                   [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7)                                                     \
                 : [m] "v"(m));                                                                                                     \
  }
// 16 registers, one step
#define PNX_SCAN16(CTRL, v, m) \
  PNX_SCAN8(CTRL, v, m)        \
  PNX_SCAN8(CTRL, (v) + 8, m)
__device__ __forceinline__ void seg_max_nn16(float* v, int idx, int col, const ScanPlan& pl) {
  if (pl.s1) {
    const uint32_t m = idx >= 1 ? 0xFFFFFFFFu : 0u;
    PNX_SCAN16("row_shr:1", v, m)
  }
  if (pl.s2) {
    const uint32_t m = idx >= 2 ? 0xFFFFFFFFu : 0u;
    PNX_SCAN16("row_shr:2", v, m)
  }
  if (pl.s4) {
    const uint32_t m = idx >= 4 ? 0xFFFFFFFFu : 0u;
    PNX_SCAN16("row_shr:4", v, m)
  }
  if (pl.s8) {
    const uint32_t m = idx >= 8 ? 0xFFFFFFFFu : 0u;
    PNX_SCAN16("row_shr:8", v, m)
  }
  if (pl.s1) {  // pillar straddling the two 16-lane rows of a half
    const uint32_t m = idx > (col & 15) ? 0xFFFFFFFFu : 0u;
    PNX_SCAN16("row_bcast:15", v, m)
  }
}

// ---- single-register forms, for streams that interleave the scan with MFMAs (pfn_v3.hip).  STEP 0..4 = row_shr:1,2,4,8, row_bcast:15;
// m[STEP] = the step's lane mask.  No s_nop inside: the caller guarantees >= 2 issued instructions between a VALU write of `v`
// and the pair (the 2 wait states a VALU-write -> DPP-read needs), e.g. with scan_fence16 below.
#define PNX_PAIR_ASM(OP, CTRL, v, mm)                                                                              \
  {                                                                                                                \
    uint32_t t_;                                                                                                   \
    asm volatile("v_and_b32_dpp %1, %0, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" OP " %0, %0, %1"    \
                 : "+v"(v), "=&v"(t_)                                                                              \
                 : "v"(mm));                                                                                       \
  }
template <int STEP>
__device__ __forceinline__ void scan_pair_f32(float& v, const uint32_t* m) {
  if (STEP == 0) PNX_PAIR_ASM("v_max_f32", "row_shr:1", v, m[0])
  if (STEP == 1) PNX_PAIR_ASM("v_max_f32", "row_shr:2", v, m[1])
  if (STEP == 2) PNX_PAIR_ASM("v_max_f32", "row_shr:4", v, m[2])
  if (STEP == 3) PNX_PAIR_ASM("v_max_f32", "row_shr:8", v, m[3])
  if (STEP == 4) PNX_PAIR_ASM("v_max_f32", "row_bcast:15", v, m[4])
}
// two non-negative 16-bit floats (bf16 or fp16) per register: they order like unsigned integers
template <int STEP>
__device__ __forceinline__ void scan_pair_pk16(uint32_t& v, const uint32_t* m) {
  if (STEP == 0) PNX_PAIR_ASM("v_pk_max_u16", "row_shr:1", v, m[0])
  if (STEP == 1) PNX_PAIR_ASM("v_pk_max_u16", "row_shr:2", v, m[1])
  if (STEP == 2) PNX_PAIR_ASM("v_pk_max_u16", "row_shr:4", v, m[2])
  if (STEP == 3) PNX_PAIR_ASM("v_pk_max_u16", "row_shr:8", v, m[3])
  if (STEP == 4) PNX_PAIR_ASM("v_pk_max_u16", "row_bcast:15", v, m[4])
}
// all 16 registers are final before this point and nothing that writes them may sink below it; covers the DPP wait states
template <typename T>
__device__ __forceinline__ void scan_fence16(T* v) {
  asm volatile("s_nop 1"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                 "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}
__device__ __forceinline__ void scan_masks(uint32_t* m, int idx, int col) {
  m[0] = idx >= 1 ? 0xFFFFFFFFu : 0u;
  m[1] = idx >= 2 ? 0xFFFFFFFFu : 0u;
  m[2] = idx >= 4 ? 0xFFFFFFFFu : 0u;
  m[3] = idx >= 8 ? 0xFFFFFFFFu : 0u;
  m[4] = idx > (col & 15) ? 0xFFFFFFFFu : 0u;  // pillar straddling the two 16-lane rows of a half
}
// 16 packed registers, the steps the tile needs
template <int STEP>
__device__ __forceinline__ void scan_step16_pk16(uint32_t* v, const uint32_t* m) {
#pragma unroll
  for (int i = 0; i < 16; i++) scan_pair_pk16<STEP>(v[i], m);
}
__device__ __forceinline__ void seg_max_pk16(uint32_t* v, const uint32_t* m, const ScanPlan& pl) {
  scan_fence16(v);
  if (pl.s1) scan_step16_pk16<0>(v, m);
  if (pl.s2) scan_step16_pk16<1>(v, m);
  if (pl.s4) scan_step16_pk16<2>(v, m);
  if (pl.s8) scan_step16_pk16<3>(v, m);
  if (pl.s1) scan_step16_pk16<4>(v, m);
}

}  // namespace
