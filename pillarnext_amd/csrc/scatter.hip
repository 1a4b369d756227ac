// scatter.hip -- deterministic per-pillar max (torch_scatter.scatter_max stand-in for the training path;
// call sites det3d/models/readers/pillar_encoder.py:43,180) and its backward, gfx950.
//
// torch_scatter resolves the max with float atomics and a second pass for the argmax.  Here the rows are
// first grouped per pillar by a counting sort (integer atomics only), then one wave per pillar streams the
// pillar's rows with lane = channel: every load is a coalesced row segment, the running (max, argmax) lives
// in registers, and ties pick the lowest row index, so the result does not depend on thread timing.
#include "pnx_common.h"
#include "pnx_scan.h"

namespace {

__global__ __launch_bounds__(kBlock) void k_sm_count(const int64_t* __restrict__ index, int64_t n, int64_t P, uint32_t* __restrict__ count,
                                                     int32_t* __restrict__ slot) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t p = index[i];
  slot[i] = (p >= 0 && p < P) ? (int32_t)atomicAdd(&count[p], 1u) : -1;
}

__global__ __launch_bounds__(kBlock) void k_sm_fill(const int64_t* __restrict__ index, const int32_t* __restrict__ slot, int64_t n,
                                                    const uint32_t* __restrict__ cpre, const uint32_t* __restrict__ cblk,
                                                    int32_t* __restrict__ plist) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t s = slot[i];
  if (s < 0) return;
  const int64_t p = index[i];
  plist[cblk[p >> PNX_SCAN_SHIFT] + cpre[p] + (uint32_t)s] = (int32_t)i;
}

// one wave per pillar (grid-strided); lane covers channels lane, lane+64, ...
__global__ __launch_bounds__(kBlock) void k_sm_max(const float* __restrict__ x, int64_t n, int C, int64_t P, const uint32_t* __restrict__ count,
                                                   const uint32_t* __restrict__ cpre, const uint32_t* __restrict__ cblk,
                                                   const int32_t* __restrict__ plist, float* __restrict__ out, int64_t* __restrict__ argmax) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
  for (int64_t p = wave; p < P; p += nwaves) {
    const uint32_t st = cblk[p >> PNX_SCAN_SHIFT] + cpre[p], c = count[p];
    for (int ch = lane; ch < C; ch += 64) {
      float best = 0.f;
      int64_t arg = n;
      for (uint32_t k = 0; k < c; k++) {
        const int32_t row = plist[st + k];
        const float v = x[(int64_t)row * C + ch];
        if (arg == n || v > best || (v == best && row < arg)) {
          best = v;
          arg = row;
        }
      }
      out[p * C + ch] = best;
      if (argmax) argmax[p * C + ch] = arg;
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_sm_bwd(const float* __restrict__ gout, const int64_t* __restrict__ argmax, int64_t n, int C, int64_t P,
                                                   float* __restrict__ gx) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= P * C) return;
  const int64_t row = argmax[idx];
  if (row >= 0 && row < n) gx[row * C + (idx % C)] = gout[idx];  // one (pillar, channel) per element: no collisions
}

struct SmWs {
  uint32_t *count, *cpre, *cblk;
  int32_t *slot, *plist;
  int nblk;
  size_t bytes;
};
SmWs sm_carve(void* ws, int64_t n, int64_t P) {
  SmWs w;
  PnxCarver c(ws);
  w.nblk = (int)((P + PNX_SCAN_ITEMS - 1) / PNX_SCAN_ITEMS);
  if (w.nblk < 1) w.nblk = 1;
  w.count = c.take<uint32_t>(P + 8);
  w.cpre = c.take<uint32_t>(P + 8);
  w.cblk = c.take<uint32_t>(w.nblk + 8);
  w.slot = c.take<int32_t>(n + 8);
  w.plist = c.take<int32_t>(n + 8);
  w.bytes = c.used();
  return w;
}

}  // namespace

extern "C" {

size_t pnx_scatter_max_workspace_bytes(int64_t n, int64_t num_pillars) {
  if (n < 0 || num_pillars < 0) return 0;
  return sm_carve(nullptr, n, num_pillars).bytes;
}

int pnx_scatter_max(const float* x, const int64_t* index, int64_t n, int32_t channels, int64_t P, float* out, int64_t* argmax,
                    void* workspace, size_t workspace_bytes, pnx_stream_t stream) {
  PNX_REQUIRE(n >= 0 && P >= 0 && channels > 0, PNX_ERR_INVALID, "bad sizes n=%lld P=%lld C=%d", (long long)n, (long long)P, channels);
  if (P == 0) return PNX_OK;
  PNX_REQUIRE(out && workspace && (n == 0 || (x && index)), PNX_ERR_INVALID, "null pointer");
  PNX_REQUIRE(n < ((int64_t)1 << 31) - 64 && P < ((int64_t)1 << 31) - 64, PNX_ERR_UNSUPPORTED, "more than 2^31 rows");
  PNX_REQUIRE(((uintptr_t)workspace & 255) == 0, PNX_ERR_INVALID, "workspace must be 256-byte aligned");
  const size_t need = pnx_scatter_max_workspace_bytes(n, P);
  PNX_REQUIRE(workspace_bytes >= need, PNX_ERR_WORKSPACE, "workspace %zu bytes < %zu needed", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const SmWs w = sm_carve(workspace, n, P);
  PNX_CHECK_HIP(hipMemsetAsync(w.count, 0, (size_t)P * 4, st));
  const int nbn = (int)((n + kBlock - 1) / kBlock);
  if (n > 0) k_sm_count<<<nbn, kBlock, 0, st>>>(index, n, P, w.count, w.slot);
  k_scan_local<SCAN_IDENT><<<w.nblk, kBlock, 0, st>>>(w.count, P, w.cpre, w.cblk);
  k_scan_blocks<<<1, kBlock, 0, st>>>(w.cblk, w.nblk, nullptr);
  if (n > 0) k_sm_fill<<<nbn, kBlock, 0, st>>>(index, w.slot, n, w.cpre, w.cblk, w.plist);
  int nb = (int)((P * 64 + kBlock - 1) / kBlock);
  if (nb > 8192) nb = 8192;
  k_sm_max<<<nb, kBlock, 0, st>>>(x, n, channels, P, w.count, w.cpre, w.cblk, w.plist, out, argmax);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

int pnx_scatter_max_backward(const float* grad_out, const int64_t* argmax, int64_t n, int32_t channels, int64_t P, float* grad_x,
                             pnx_stream_t stream) {
  PNX_REQUIRE(n >= 0 && P >= 0 && channels > 0, PNX_ERR_INVALID, "bad sizes");
  if (n == 0) return PNX_OK;
  PNX_REQUIRE(grad_x, PNX_ERR_INVALID, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  PNX_CHECK_HIP(hipMemsetAsync(grad_x, 0, (size_t)n * channels * sizeof(float), st));
  if (P == 0) return PNX_OK;
  PNX_REQUIRE(grad_out && argmax, PNX_ERR_INVALID, "null pointer");
  const int64_t total = P * channels;
  k_sm_bwd<<<(unsigned)((total + kBlock - 1) / kBlock), kBlock, 0, st>>>(grad_out, argmax, n, channels, P, grad_x);
  PNX_LAUNCH_CHECK();
  return PNX_OK;
}

}  // extern "C"
