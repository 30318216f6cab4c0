// ray_host_math.h -- host-side helpers of libray_amd.so.
//
// linspace(a, b, n): the reference builds its node, source-angle and direction grids with
// Julia's linspace (NonObstacle.jl:55-56,191; Obstacle.jl:12-13), whose elements are
// ref + (k-1)*step evaluated in twice-double precision and rounded once.  The equivalent here:
// the exact real value a + (k-1)(b-a)/(n-1) of the double inputs, carried as an unevaluated
// (hi, lo) pair through every operation and rounded to double at the end.
#pragma once
#include <cmath>

namespace ray {
namespace hostmath {

struct dd {
    double hi, lo;
};

inline dd quick_renorm(double hi, double lo)
{
    const double s = hi + lo;
    return dd{s, lo - (s - hi)};
}
inline dd add_exact(double x, double y)          // Knuth: x + y = hi + lo exactly
{
    const double s = x + y;
    const double v = s - x;
    return dd{s, (x - (s - v)) + (y - v)};
}
inline dd mul_exact(double x, double y)          // x * y = hi + lo exactly (fma)
{
    const double p = x * y;
    return dd{p, std::fma(x, y, -p)};
}
inline dd div_by_double(dd x, double d)          // (hi, lo) / d with a second-order correction
{
    const double q1 = x.hi / d;
    const dd back = mul_exact(q1, d);
    const double rem = ((x.hi - back.hi) - back.lo) + x.lo;
    const double q2 = rem / d;
    return add_exact(q1, q2);
}
inline dd mul_by_double(dd x, double f)
{
    dd p = mul_exact(x.hi, f);
    p.lo += x.lo * f;
    return p;
}

inline void linspace(double a, double b, int n, double *out)
{
    if (n <= 0) return;
    if (n == 1) { out[0] = a; return; }
    const dd span = add_exact(b, -a);
    const dd step = div_by_double(span, (double)(n - 1));
    for (int k = 0; k < n; ++k) {
        const dd shift = mul_by_double(step, (double)k);
        const dd base = add_exact(a, shift.hi);
        out[k] = base.hi + (base.lo + shift.lo);
    }
    out[n - 1] = b;
}

} // namespace hostmath
} // namespace ray
