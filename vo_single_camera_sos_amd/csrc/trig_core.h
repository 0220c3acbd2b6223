// sin / cos / atan in double precision from + - * / and comparisons only (Cody-Waite reduction, minimax
// polynomials): the panorama geometry of the device code.  The CPU oracle evaluates the SAME text (oracle/trig_core.h, with
// its own prefix) so that both sides agree to the bit; tests/gen_device_headers.py keeps the two files identical
// and tests/test_abi.py checks it.  Edit both through that script.
#pragma once
#include <hip/hip_runtime.h>

/* x -> (*s, *c) = (sin x, cos x), |x| < ~1e5 */
__device__ static void sv_sincos(double x, double* s, double* c) {
  const double invpio2 = 6.36619772367581382433e-01;
  const double pio2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
  const double pio2_2 = 6.07710050630396597660e-11;  /* second 33 bits */
  const double pio2_2t = 2.02226624879595063154e-21; /* pi/2 - (pio2_1 + pio2_2) */
  const double t0 = x * invpio2;
  const long long n = (long long)(t0 + (t0 < 0.0 ? -0.5 : 0.5));
  const double fn = (double)n;
  const double r1 = x - (fn * pio2_1);
  const double w2 = fn * pio2_2;
  const double r2 = r1 - w2;
  const double w3 = (fn * pio2_2t) - ((r1 - r2) - w2);
  const double y0 = r2 - w3;           /* reduced argument, head */
  const double y1 = (r2 - y0) - w3;    /* and tail */
  const double z = y0 * y0;
  /* sine kernel */
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double v = z * y0;
  const double rs = S2 + (z * (S3 + (z * (S4 + (z * (S5 + (z * S6)))))));
  const double ks = y0 - (((z * ((0.5 * y1) - (v * rs))) - y1) - (v * S1));
  /* cosine kernel */
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double rc = z * (C1 + (z * (C2 + (z * (C3 + (z * (C4 + (z * (C5 + (z * C6))))))))));
  const double ay = y0 < 0.0 ? -y0 : y0;
  double kc;
  if (ay < 0.3) {
    kc = 1.0 - ((0.5 * z) - ((z * rc) - (y0 * y1)));
  } else {
    const double qx = ay > 0.78125 ? 0.28125 : (double)(float)(ay * 0.25); /* a short constant near |y|/4: 1 - qx is exact */
    const double hz = (0.5 * z) - qx;
    const double a = 1.0 - qx;
    kc = a - (hz - ((z * rc) - (y0 * y1)));
  }
  const long long q = n & 3LL;
  if (q == 0) {
    *s = ks;
    *c = kc;
  } else if (q == 1) {
    *s = kc;
    *c = -ks;
  } else if (q == 2) {
    *s = -ks;
    *c = -kc;
  } else {
    *s = -kc;
    *c = ks;
  }
  if (!(x == x)) { /* NaN in, NaN out (the integer conversion above is not meaningful then) */
    *s = x;
    *c = x;
  }
}

/* atan x for finite x (atan2(x, 1) of the panorama's elevation) */
__device__ static double sv_atan(double x) {
  if (!(x == x)) return x;
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
               aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
               aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
  const int neg = x < 0.0;
  double ax = neg ? -x : x;
  double hi = 0.0, lo = 0.0;
  int reduced = 1;
  if (ax >= 7.378697629483821e19) { /* 2^66 */
    const double big = 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
    return neg ? -big : big;
  }
  if (ax < 0.4375) {
    reduced = 0;
  } else if (ax < 0.6875) {
    hi = 4.63647609000806093515e-01;
    lo = 2.26987774529616870924e-17;
    ax = ((2.0 * ax) - 1.0) / (2.0 + ax);
  } else if (ax < 1.1875) {
    hi = 7.85398163397448278999e-01;
    lo = 3.06161699786838301793e-17;
    ax = (ax - 1.0) / (ax + 1.0);
  } else if (ax < 2.4375) {
    hi = 9.82793723247329054082e-01;
    lo = 1.39033110312309984516e-17;
    ax = (ax - 1.5) / (1.0 + (1.5 * ax));
  } else {
    hi = 1.57079632679489655800e+00;
    lo = 6.12323399573676603587e-17;
    ax = -1.0 / ax;
  }
  const double z = ax * ax, w = z * z;
  const double s1 = z * (aT0 + (w * (aT2 + (w * (aT4 + (w * (aT6 + (w * (aT8 + (w * aT10))))))))));
  const double s2 = w * (aT1 + (w * (aT3 + (w * (aT5 + (w * (aT7 + (w * aT9))))))));
  double r;
  if (!reduced)
    r = ax - (ax * (s1 + s2));
  else
    r = hi - (((ax * (s1 + s2)) - lo) - ax);
  return neg ? -r : r;
}
