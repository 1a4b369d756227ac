// pnx_detmath.h -- deterministic sinf/cosf/atan2f for the rotated-IoU path.
//
// Why this exists: the reference's box_overlap
// (det3d/core/iou3d_nms/src/iou3d_nms_kernel.cu:104-225) calls cos/sin/atan2 and then
// bubble-sorts polygon vertices by atan2; a 1-ulp libm difference between hosts/devices can
// reorder vertices and move an IoU across the NMS threshold.  These routines use only IEEE-754
// fp64 + - * / and floor, in a fixed order, so the HIP device code and any host build produce
// bit-identical fp32 results (both sides must be compiled with -ffp-contract=off).
//
// Accuracy: results are the fp64 value (abs err < 1e-15) rounded once to fp32, i.e. correctly
// rounded except on ~1e-8 of inputs.  Arguments with |a| >= 1e9 (or non-finite) give NaN.
#ifndef PNX_DETMATH_H
#define PNX_DETMATH_H

#ifndef PNX_HD
#define PNX_HD static inline
#endif

PNX_HD double pnx_dm_floor(double x) { return __builtin_floor(x); }

// sin and cos of a (radians).  Cody-Waite 3-term reduction by pi/2 (33+33+53 bits), then the
// classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4].
PNX_HD void pnx_sincosf(float a, float* s_out, float* c_out) {
  double x = (double)a;
  double ax = x < 0.0 ? -x : x;
  if (!(ax < 1.0e9)) {  // also catches NaN
    float qnan = __builtin_nanf("");
    *s_out = qnan;
    *c_out = qnan;
    return;
  }
  double t = x * 0.63661977236758134308;  // 2/pi
  double k = pnx_dm_floor(t + 0.5);
  double r = x - k * 1.57079632673412561417e+00;
  r = r - k * 6.07710050630396597660e-11;
  r = r - k * 2.02226624879595063154e-21;
  long long q = (long long)k;
  double z = r * r;
  // sin kernel
  double ps = 8.33333333332248946124e-03 +
              z * (-1.98412698298579493134e-04 +
                   z * (2.75573137070700676789e-06 +
                        z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  double sn = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
  // cos kernel
  double pc = z * (4.16666666666666019037e-02 +
                   z * (-1.38888888888741095749e-03 +
                        z * (2.48015872894767294178e-05 +
                             z * (-2.75573143513906633035e-07 +
                                  z * (2.08757232129817482790e-09 +
                                       z * -1.13596475577881948265e-11)))));
  double cs = 1.0 - (0.5 * z - z * pc);
  double so, co;
  switch ((int)(q & 3)) {
    case 0: so = sn; co = cs; break;
    case 1: so = cs; co = -sn; break;
    case 2: so = -sn; co = -cs; break;
    default: so = -cs; co = sn; break;
  }
  *s_out = (float)so;
  *c_out = (float)co;
}

// atan2f(y, x): fp64 evaluation of atan on [0, tan(pi/8)] with the 11-coefficient odd kernel,
// argument halving through (a-1)/(a+1), then octant fix-up.
PNX_HD float pnx_atan2f(float yf, float xf) {
  double y = (double)yf, x = (double)xf;
  if (x != x || y != y) return __builtin_nanf("");
  double ax = x < 0.0 ? -x : x;
  double ay = y < 0.0 ? -y : y;
  double mx = ax > ay ? ax : ay;
  double mn = ax > ay ? ay : ax;
  double r;
  if (mx == 0.0) {
    r = 0.0;
  } else {
    double a = (mx == mn) ? 1.0 : mn / mx;  // inf/inf -> 1
    double base = 0.0, t = a;
    if (a > 0.41421356237309503) {
      t = (a - 1.0) / (a + 1.0);
      base = 0.78539816339744830962;
    }
    double z = t * t;
    double w = z * z;
    double s1 = z * (3.33333333333329318027e-01 +
                     w * (1.42857142725034663711e-01 +
                          w * (9.09088713343650656196e-02 +
                               w * (6.66107313738753120669e-02 +
                                    w * (4.97687799461593236017e-02 +
                                         w * 1.62858201153657823623e-02)))));
    double s2 = w * (-1.99999999998764832476e-01 +
                     w * (-1.11111104054623557880e-01 +
                          w * (-7.69187620504482999495e-02 +
                               w * (-5.83357013379057348645e-02 +
                                    w * -3.65315727442169155270e-02))));
    r = base + (t - t * (s1 + s2));
    if (ay > ax) r = 1.57079632679489661923 - r;
  }
  if (x < 0.0 || (x == 0.0 && __builtin_signbit(x))) r = 3.14159265358979323846 - r;
  if (y < 0.0 || (y == 0.0 && __builtin_signbit(y))) r = -r;
  return (float)r;
}

#endif  // PNX_DETMATH_H
