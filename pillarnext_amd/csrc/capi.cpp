// capi.cpp -- host-only pieces of the C ABI (include/pnx.h): error channel, version, geometry helper.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/pnx.h"

static thread_local char g_err[512] = "";

void pnx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* pnx_last_error(void) { return g_err; }
const char* pnx_version(void) { return "pillarnext_amd 0.1 (gfx950)"; }

// pillar_encoder.py:87-93: grid = np.round((max - min) / voxel) in fp64 (half-to-even); pc_min / voxel
// are cast to fp32 exactly as torch.from_numpy(...).type_as(points) does.
int pnx_geom_init(const double* pc_range, const double* voxel, pnx_geom* g) {
  if (!pc_range || !voxel || !g) {
    pnx_set_error("pnx_geom_init: null pointer");
    return PNX_ERR_INVALID;
  }
  for (int i = 0; i < 3; i++) {
    if (!(voxel[i] > 0.0) || !(pc_range[3 + i] > pc_range[i])) {
      pnx_set_error("pnx_geom_init: bad range/voxel on axis %d", i);
      return PNX_ERR_INVALID;
    }
    g->pc_min[i] = (float)pc_range[i];
    g->voxel[i] = (float)voxel[i];
  }
  const double gx = nearbyint((pc_range[3] - pc_range[0]) / voxel[0]);
  const double gy = nearbyint((pc_range[4] - pc_range[1]) / voxel[1]);
  if (gx < 1 || gy < 1 || gx > 65535 || gy > 65535) {
    pnx_set_error("pnx_geom_init: grid %g x %g out of range", gx, gy);
    return PNX_ERR_UNSUPPORTED;
  }
  g->gx = (int32_t)gx;
  g->gy = (int32_t)gy;
  return PNX_OK;
}

}  // extern "C"
