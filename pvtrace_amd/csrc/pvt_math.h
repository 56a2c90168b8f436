// pvt_math.h — bit-reproducible double-precision elementary functions.
//
// The photon loop needs log, sin, cos, asin and acos.  Vendor libms (glibc on
// the host, OCML on gfx950) each round these differently in the last ulp, which
// would make a photon's history on the GPU drift away from the CPU referee
// after the first absorption.  These versions use only IEEE-754 +,-,*,/ and
// sqrt (all correctly rounded on x86-64 and on gfx950) and integer bit
// manipulation, evaluated in a fixed order with FMA contraction disabled
// (-ffp-contract=off on both compilers), so the SAME bits come out of gcc on
// the host and hipcc on the device.  That turns GPU-vs-oracle parity from a
// statistical statement into an exact one (whole event logs compare equal).
//
// Algorithms: the classic Sun/FreeBSD msun reductions and minimax polynomials
// (k_sin/k_cos/e_log/e_asin/e_acos; Cody-Waite 3-term pi/2 reduction).  Error
// < 1 ulp on the domains the tracer uses (|angle| <= 2*pi*2^20, log on (0,1]).
// tests/test_math.py measures them against libm.
#ifndef PVT_MATH_H
#define PVT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PVT_HD __host__ __device__ __forceinline__
#else
#define PVT_HD static inline
#endif

#if defined(__HIPCC__)
#define PVT_HD_STATIC static PVT_HD
#else
#define PVT_HD_STATIC PVT_HD
#endif

PVT_HD_STATIC uint64_t pvt_d2u(double x) {
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return u;
}
PVT_HD_STATIC double pvt_u2d(uint64_t u) {
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}
PVT_HD_STATIC uint32_t pvt_hi(double x) { return (uint32_t)(pvt_d2u(x) >> 32); }
PVT_HD_STATIC double pvt_clear_lo(double x) {
    return pvt_u2d(pvt_d2u(x) & 0xFFFFFFFF00000000ull);
}
PVT_HD_STATIC double pvt_sqrt(double x) { return __builtin_sqrt(x); }
PVT_HD_STATIC double pvt_fabs(double x) { return __builtin_fabs(x); }

// ---------------------------------------------------------------- log
PVT_HD_STATIC double pvt_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = pvt_d2u(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((u << 1) == 0) return -1.0 / (x * x);  // log(+-0) = -inf
        if (hx >> 31) return (x - x) / 0.0;        // log(-#) = NaN
        k -= 54;                                   // subnormal: scale up
        x *= 18014398509481984.0;                  // 2^54
        u = pvt_d2u(x);
        hx = (uint32_t)(u >> 32);
    } else if (hx >= 0x7ff00000u) {
        return x;
    } else if (hx == 0x3ff00000u && (u << 32) == 0) {
        return 0.0;
    }
    // reduce to [sqrt(2)/2, sqrt(2))
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    x = pvt_u2d(((uint64_t)hx << 32) | (u & 0xffffffffull));
    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

// ------------------------------------------------------------ sin / cos
// kernels on [-pi/4, pi/4]; (x, y) is the reduced argument head and tail
PVT_HD_STATIC double pvt_ksin(double x, double y, int iy) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double w = z * z;
    double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    double v = z * x;
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

PVT_HD_STATIC double pvt_kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double w = z * z;
    double r = z * (C1 + z * (C2 + z * C3)) + w * w * (C4 + z * (C5 + z * C6));
    double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

// x = n*pi/2 + (y0 + y1), |y0| <= pi/4 (+ a little).  Valid for |x| < ~1.6e6;
// the tracer only ever passes angles in [0, 2*pi].
PVT_HD_STATIC int pvt_rem_pio2(double x, double* y0, double* y1) {
    const double toint = 6755399441055744.0;  // 1.5 * 2^52
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const double pio2_3 = 2.02226624871116645580e-21, pio2_3t = 8.47842766036889956997e-32;
    uint32_t ix = pvt_hi(x) & 0x7fffffffu;
    double fn = x * invpio2 + toint - toint;
    int n = (int)fn;
    double r = x - fn * pio2_1;
    double w = fn * pio2_1t;
    double a = r - w;
    int ex = (int)(ix >> 20);
    int ey = (int)((pvt_hi(a) >> 20) & 0x7ff);
    if (ex - ey > 16) {
        double t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        a = r - w;
        ey = (int)((pvt_hi(a) >> 20) & 0x7ff);
        if (ex - ey > 49) {
            t = r;
            w = fn * pio2_3;
            r = t - w;
            w = fn * pio2_3t - ((t - r) - w);
            a = r - w;
        }
    }
    *y0 = a;
    *y1 = (r - a) - w;
    return n;
}

// One evaluation of each kernel whatever the argument: arguments inside [-pi/4, pi/4] are
// their own reduction (n = 0, tail 0), so SIMD lanes with small and large angles share the
// same instructions instead of serialising two copies of the polynomials.
PVT_HD_STATIC void pvt_sincos(double x, double* s, double* c) {
    uint32_t ix = pvt_hi(x) & 0x7fffffffu;
    if (ix >= 0x7ff00000u) {
        *s = *c = x - x;
        return;
    }
    double y0 = x, y1 = 0.0;
    int n = 0;
    if (ix > 0x3fe921fbu) n = pvt_rem_pio2(x, &y0, &y1);  // |x| > ~pi/4
    double ks = pvt_ksin(y0, y1, 1);
    double kc = pvt_kcos(y0, y1);
    if (ix < 0x3e500000u) ks = x;    // |x| < 2^-26: sin x == x
    if (ix < 0x3e46a09eu) kc = 1.0;  // |x| < 2^-27*sqrt2: cos x == 1
    switch (n & 3) {
        case 0: *s = ks; *c = kc; break;
        case 1: *s = kc; *c = -ks; break;
        case 2: *s = -ks; *c = -kc; break;
        default: *s = -kc; *c = ks; break;
    }
}

PVT_HD_STATIC double pvt_sin(double x) {
    double s, c;
    pvt_sincos(x, &s, &c);
    return s;
}

PVT_HD_STATIC double pvt_cos(double x) {
    double s, c;
    pvt_sincos(x, &s, &c);
    return c;
}

// ---------------------------------------------------------- asin / acos
PVT_HD_STATIC double pvt_asin_R(double z) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
    const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    return p / q;
}

// asin / acos evaluate the rational R(z) exactly once: the argument of R is selected
// first (z = x^2 for |x| < 0.5, (1-|x|)/2 otherwise), so no branch duplicates the
// polynomial and SIMD lanes on either side of 0.5 share one evaluation.
PVT_HD_STATIC double pvt_asin(double x) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    uint64_t u = pvt_d2u(x);
    uint32_t hx = (uint32_t)(u >> 32);
    uint32_t ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {  // |x| >= 1 or NaN
        if (((ix - 0x3ff00000u) | (uint32_t)u) == 0) return x * pio2_hi;
        return (x - x) / (x - x);
    }
    const int small = ix < 0x3fe00000u;  // |x| < 0.5
    if (small && ix < 0x3e500000u) return x;
    const double z = small ? x * x : (1.0 - pvt_fabs(x)) * 0.5;
    const double r = pvt_asin_R(z);
    if (small) return x + x * r;
    const double s = pvt_sqrt(z);
    double res;
    if (ix >= 0x3fef3333u) {  // |x| > 0.975
        res = pio2_hi - (2.0 * (s + s * r) - pio2_lo);
    } else {
        double f = pvt_clear_lo(s);
        double c = (z - f * f) / (s + f);
        res = 0.5 * pio2_hi - (2.0 * s * r - (pio2_lo - 2.0 * c) - (0.5 * pio2_hi - 2.0 * f));
    }
    return (hx >> 31) ? -res : res;
}

PVT_HD_STATIC double pvt_acos(double x) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    uint64_t u = pvt_d2u(x);
    uint32_t hx = (uint32_t)(u >> 32);
    uint32_t ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) {  // |x| >= 1 or NaN
        if (((ix - 0x3ff00000u) | (uint32_t)u) == 0) return (hx >> 31) ? 2.0 * pio2_hi : 0.0;
        return (x - x) / (x - x);
    }
    const int small = ix < 0x3fe00000u;  // |x| < 0.5
    if (small && ix <= 0x3c600000u) return pio2_hi;
    const double z = small ? x * x : (1.0 - pvt_fabs(x)) * 0.5;  // (1+x)/2 for x < -0.5: same bits
    const double r = pvt_asin_R(z);
    if (small) return pio2_hi - (x - (pio2_lo - x * r));
    const double s = pvt_sqrt(z);
    if (hx >> 31) {  // x < -0.5
        double w = r * s - pio2_lo;
        return 2.0 * (pio2_hi - (s + w));
    }
    double df = pvt_clear_lo(s);  // x > 0.5
    double c = (z - df * df) / (s + df);
    double w = r * s + c;
    return 2.0 * (df + w);
}

// ------------------------------------------------- composed functions
// The reference writes sin(acos(c)), cos(acos(c)) and cos(asin(s)) as two library calls each: a polar angle
// is sampled through its cosine (isotropic and Henyey-Greenstein phase functions, _kernel.pyx:455-476) or
// its sine (cone), the incidence angle is acos(n.d) and Fresnel's formulas take its sine and cosine
// (:406-419).  Composing two <= 1 ulp functions gives ~2.5 ulp; evaluating the composition directly is both
// more accurate and far cheaper:  cos(acos(c)) = c exactly, sin(acos(c)) = cos(asin(c)) = sqrt((1-c)(1+c))
// (one factor is exact for |c| >= 1/2, Sterbenz; < 1.3 ulp overall; NaN for |c| > 1 like acos/asin).
PVT_HD_STATIC double pvt_sqrt1m2(double c) { return pvt_sqrt((1.0 - c) * (1.0 + c)); }

// sin(2 pi g), cos(2 pi g) for a turn fraction g (an azimuth is sampled as phi = 2 pi u, _kernel.pyx:455-476).
// The quadrant reduction is EXACT in g (g - n/4 has no rounding for |g| <= 2^20), so no Cody-Waite passes
// are needed; the reduced angle 2 pi (g - n/4), |.| <= pi/4, is formed as head + tail with two explicitly
// written fused multiply-adds (v_fma_f64 on the device, the correctly rounded fma() of the C library on the
// host: the same bits; contraction of everything else stays off).  Error < 1 ulp of the exact sin / cos of
// 2 pi g -- the reference's phi = RN(2 pi u) alone is already half an ulp of phi away from it.
PVT_HD_STATIC void pvt_sincos2pi(double g, double* s, double* c) {
    const double two_pi_hi = 6.28318530717958623200e+00, two_pi_lo = 2.44929359829470641435e-16;
    const double fn = g * 4.0 + 6755399441055744.0 - 6755399441055744.0;   // nearest quarter turn (1.5 * 2^52 trick)
    const int n = (int)fn;
    const double r = g - fn * 0.25;                                          // exact
    const double t = r * two_pi_hi;
    const double tail = __builtin_fma(r, two_pi_lo, __builtin_fma(r, two_pi_hi, -t));
    const double ks = pvt_ksin(t, tail, 1);
    const double kc = pvt_kcos(t, tail);
    switch (n & 3) {
        case 0: *s = ks; *c = kc; break;
        case 1: *s = kc; *c = -ks; break;
        case 2: *s = -ks; *c = -kc; break;
        default: *s = -kc; *c = ks; break;
    }
}

#endif  // PVT_MATH_H
