// sin/cos for the analytic media (a1 with the shipped closures, NonObstacleDemo.jl:12,23; ObstacleDemo.jl:12,18) and
// the initial directions (NonObstacle.jl:62-65), spelled out in +, -, *, fma and one round-to-integer, every one a
// single IEEE rounding: the same bits on the GPU and on any conforming host, so that the analytic paths can be checked
// bit for bit against a CPU restatement (oracle flavour trig = 1) instead of "up to the two math libraries".
// The reference calls Julia's sin/cos (openlibm); results here stay within 1 ulp of the exact values
// (tests/test_trig.py measures it against long double), the same class of difference as between two libms.
//
// Algorithm: the classical one (Cody-Waite reduction by pi/2 in three pieces, minimax polynomials on [-pi/4, pi/4]
// with a two-double reduced argument, as in Sun's fdlibm, whose published coefficients these are), valid for
// |a| < 2^18; callers fall back to the platform's sincos beyond that (not reproducible across platforms, and far
// outside anything the unit disk produces).  `shift` advances a known (sin a0, cos a0) by a small angle, which is what
// the RK4 stages need: their arguments sit within w * dt * c of the step's first stage.
#pragma once

#if defined(__HIPCC__)
#define RAY_TRIG_FN __host__ __device__ __forceinline__
#else
#define RAY_TRIG_FN static inline
#endif

namespace ray {
namespace trig {

constexpr double kBound = 262144.0;        // 2^18: n * pio2_1 is exact for |n| < 2^20
constexpr double kShiftMax = 0.25;         // |d| up to which shift() is used

constexpr double kInvPio2 = 6.36619772367581382433e-01;    // 0x3FE45F306DC9C883
constexpr double kPio2_1 = 1.57079632673412561417e+00;     // 0x3FF921FB54400000  first 33 bits of pi/2
constexpr double kPio2_1t = 6.07710050650619224932e-11;    // 0x3DD0B4611A626331  pi/2 - kPio2_1

constexpr double kS1 = -1.66666666666666324348e-01, kS2 = 8.33333333332248946124e-03, kS3 = -1.98412698298579493134e-04,
                 kS4 = 2.75573137070700676789e-06, kS5 = -2.50507602534068634195e-08, kS6 = 1.58969099521155010221e-10;
constexpr double kC1 = 4.16666666666666019037e-02, kC2 = -1.38888888888741095749e-03, kC3 = 2.48015872894767294178e-05,
                 kC4 = -2.75573143513906633035e-07, kC5 = 2.08757232129817482790e-09, kC6 = -1.13596475577881948265e-11;

// fma(a, b, k) with a coefficient k.  On the GPU the coefficient is asked for in a scalar register and the result gets
// a destination of its own (v_fma_f64): left alone, the compiler parks the 26 coefficients in vector registers and
// pays a 64-bit copy per Horner step for its two-address v_fmac_f64.  Same operation, same bits.
RAY_TRIG_FN double fma_k(double a, double b, double k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
    return d;
#else
    return __builtin_fma(a, b, k);
#endif
}

RAY_TRIG_FN bool in_range(double a) { return __builtin_fabs(a) < kBound; }          // false for NaN too

// sin and cos of y0 + y1, |y0| <= pi/4 (+ rounding), |y1| << |y0|
RAY_TRIG_FN void kernel(double y0, double y1, double &s, double &c)
{
    const double z = y0 * y0;
    const double v = z * y0;
    const double rs = fma_k(z, fma_k(z, fma_k(z, fma_k(z, kS6, kS5), kS4), kS3), kS2);
    double u = __builtin_fma(-v, rs, 0.5 * y1);
    u = __builtin_fma(z, u, -y1);
    u = __builtin_fma(-v, kS1, u);
    s = y0 - u;
    const double rc = z * fma_k(z, fma_k(z, fma_k(z, fma_k(z, fma_k(z, kC6, kC5), kC4), kC3), kC2), kC1);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double e = (1.0 - w) - hz;
    const double t = __builtin_fma(z, rc, -(y0 * y1));
    c = w + (e + t);
}

// sin(a), cos(a) for |a| < kBound
RAY_TRIG_FN void sincos_bounded(double a, double &s, double &c)
{
    const double fn = __builtin_rint(a * kInvPio2);
    const int n = (int)fn;
    const double r = __builtin_fma(-fn, kPio2_1, a);          // exact
    const double w = fn * kPio2_1t;
    const double y0 = r - w;
    const double y1 = (r - y0) - w;
    double ks, kc;
    kernel(y0, y1, ks, kc);
    const double ss = (n & 1) ? kc : ks;
    const double cc = (n & 1) ? ks : kc;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}

// sin(a0 + d), cos(a0 + d) from s0 = sin a0, c0 = cos a0, |d| <= kShiftMax (Taylor sums of sin d and 1 - cos d: the
// truncation is below 3e-18 there)
RAY_TRIG_FN void shift(double s0, double c0, double d, double &s, double &c)
{
    const double z = d * d;
    const double ps = fma_k(z, fma_k(z, fma_k(z, fma_k(z, -2.50521083854417187751e-08, 2.75573192239858906526e-06),
                                              -1.98412698412698412698e-04),
                                    8.33333333333333333333e-03),
                            -1.66666666666666666667e-01);
    const double sd = d + (d * z) * ps;                       // sin d
    const double pc = fma_k(z, fma_k(z, fma_k(z, fma_k(z, fma_k(z, -2.08767569878680989792e-09, 2.75573192239858906526e-07),
                                                        -2.48015873015873015873e-05),
                                              1.38888888888888888889e-03),
                                    -4.16666666666666666667e-02),
                            0.5);
    const double v = z * pc;                                  // 1 - cos d
    s = s0 + __builtin_fma(c0, sd, -(s0 * v));
    c = c0 - __builtin_fma(s0, sd, c0 * v);
}

// The same advance with its multiply-adds contracted (the kFused flavour of the analytic right-hand sides): identical
// Taylor sums; sin d = fma(d z, ps, d), and the two angle-addition sums as nested fmas
//   s = fma(c0, sd, fma(-s0, v, s0)),  c = fma(-s0, sd, fma(-c0, v, c0)):  17 operations instead of 20.
RAY_TRIG_FN void shift_fused(double s0, double c0, double d, double &s, double &c)
{
    const double z = d * d;
    const double ps = fma_k(z, fma_k(z, fma_k(z, fma_k(z, -2.50521083854417187751e-08, 2.75573192239858906526e-06),
                                              -1.98412698412698412698e-04),
                                    8.33333333333333333333e-03),
                            -1.66666666666666666667e-01);
    const double sd = __builtin_fma(d * z, ps, d);            // sin d
    const double pc = fma_k(z, fma_k(z, fma_k(z, fma_k(z, fma_k(z, -2.08767569878680989792e-09, 2.75573192239858906526e-07),
                                                        -2.48015873015873015873e-05),
                                              1.38888888888888888889e-03),
                                    -4.16666666666666666667e-02),
                            0.5);
    const double v = z * pc;                                  // 1 - cos d
    s = __builtin_fma(c0, sd, __builtin_fma(-s0, v, s0));
    c = __builtin_fma(-s0, sd, __builtin_fma(-c0, v, c0));
}

// The advance by a TINY difference, |d| <= kTinyMax = 2^-10 (kFused analytic right-hand sides): RK4's third stage sits
// O(dt^2) from its second, the next step's first stage O(dt^3) from the fourth, so their sin/cos follow from the stage
// before by sin d = d - d^3/6 (next term 7.4e-18) and 1 - cos d = d^2/2 - d^4/24 (next term 1.2e-21): 9 operations
// instead of shift_fused's 17.
constexpr double kTinyMax = 0.0009765625;
RAY_TRIG_FN void shift_tiny(double s0, double c0, double d, double &s, double &c)
{
    const double z = d * d;
    const double sd = fma_k(d * z, -1.66666666666666666667e-01, d);
    const double v = z * fma_k(z, -4.16666666666666666667e-02, 0.5);
    s = __builtin_fma(c0, sd, __builtin_fma(-s0, v, s0));
    c = __builtin_fma(-s0, sd, __builtin_fma(-c0, v, c0));
}

// How many RK4 steps one full evaluation serves as the anchor of the advances (kFused analytic right-hand sides): the
// argument w x moves by at most |w| dt cmax per step (a ray's speed is c <= cmax), and every advance -- the last stage
// of the K-th step included -- must stay within kShiftMax of the anchor.  0.24: a margin below kShiftMax; a lane that
// still gets beyond it evaluates in full (and re-anchors), so K only decides cost, never accuracy.
RAY_TRIG_FN int anchor_period(double w, double dt, double cmax)
{
    const double k = __builtin_floor(0.24 / (__builtin_fabs(w) * dt * cmax));
    return k >= 64.0 ? 64 : (k >= 1.0 ? (int)k : 1);            // NaN, 0: every step
}

} // namespace trig
} // namespace ray
