// pnx_common.h -- shared declarations for the gfx950 kernels behind include/pnx.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pnx.h"

void pnx_set_error(const char* fmt, ...);

#define PNX_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      pnx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return PNX_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

#define PNX_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) {                   \
      pnx_set_error(__VA_ARGS__);    \
      return (code);                 \
    }                                \
  } while (0)

#define PNX_LAUNCH_CHECK()                                                         \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess) {                                                        \
      pnx_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return PNX_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

static inline size_t pnx_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct PnxCarver {
  char* base;
  size_t off;
  explicit PnxCarver(void* p) : base((char*)p), off(0) {}
  template <typename T>
  T* take(size_t count) {
    off = pnx_align_up(off, 256);
    T* r = (T*)(base ? base + off : nullptr);
    off += count * sizeof(T);
    return r;
  }
  size_t used() const { return pnx_align_up(off, 256); }
};

// Device-side copy of pnx_geom plus the padded row length of the occupancy bitmap.
struct PnxGeomDev {
  float minx, miny, minz, vx, vy, vz;
  int gx, gy, gyp;  // gyp = gy rounded up to a multiple of 32 (one bitmap word = 32 consecutive yi of one xi)
  int B;
};

// Two-level exclusive scan granularity: one 256-thread block scans 2048 items.
#define PNX_SCAN_ITEMS 2048
#define PNX_SCAN_SHIFT 11

// fp16x3 layer 1 of pfn_v3.hip: power-of-two pre-scales of the layer-0 output (2^SU) and of W1' (2^SW) that keep the fp16 hi/lo
// parts in the normal range; the product scale 2^-(SU+SW) is applied (exactly) in the epilogue.
#define PNX_PFN_SU 6
#define PNX_PFN_SW 8
