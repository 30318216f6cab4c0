/*
 * ray_oracle.c -- scalar fp64 CPU restatement of lowrank/ray's forward ray-shooting path.
 *
 * TEST INFRASTRUCTURE ONLY (see ray_oracle.h).  "parity unpinned" by the reference: it has no
 * tests/golden vectors and cannot be executed here; see the header for how this file is pinned.
 *
 * Build:  gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC  (oracle/Makefile)
 * Every floating-point expression below is written in the reference's evaluation order;
 * -ffp-contract=off forbids the compiler from fusing a*b+c, so each operator is one IEEE
 * binary64 rounding exactly as in un-fused Julia code.  fma() appears only inside the
 * double-double helper used for linspace (Julia's TwicePrecision range indexing).
 *
 * Assumed Julia-0.6 semantics (not verifiable offline):
 *   norm(v) for short vectors = sqrt(v1^2 + v2^2 ...) left to right (generic_vecnorm2, no rescale)
 *   a'*b                      = left-to-right dot product
 *   linspace(a,b,n)[k]        = correctly rounded a + (k-1)(b-a)/(n-1)   (TwicePrecision)
 *   n-ary a+b+c, a*b*c        = left to right
 */
#include "ray_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * double-double helpers (only for linspace)
 * ---------------------------------------------------------------------------------------- */
static void two_sum(double a, double b, double *s, double *e)
{
    double ss = a + b;
    double bb = ss - a;
    *e = (a - (ss - bb)) + (b - bb);
    *s = ss;
}
static void two_prod(double a, double b, double *p, double *e)
{
    double pp = a * b;
    *e = fma(a, b, -pp);
    *p = pp;
}

/* NonObstacle.jl:55-56,191 / Obstacle.jl:12-13 : linspace(a,b,n); out[k-1] = p[k].
 * Julia 0.6 StepRangeLen{Float64,TwicePrecision}: ref + (k-1)*step evaluated in ~106 bits and
 * rounded once.  Restated as double-double arithmetic on the exact doubles a, b.             */
void orc_linspace(double a, double b, int n, double *out)
{
    if (n == 1) { out[0] = a; return; }
    double dh, dl;                      /* d = b - a, exact as a double-double */
    two_sum(b, -a, &dh, &dl);
    double m = (double)(n - 1);
    /* q = d / m in double-double */
    double qh = dh / m, ph, pl;
    two_prod(qh, m, &ph, &pl);
    double r = ((dh - ph) - pl) + dl;
    double ql = r / m;
    double t;
    two_sum(qh, ql, &qh, &t); ql = t;
    for (int k = 1; k <= n; ++k) {
        double u = (double)(k - 1);
        double th, tl;
        two_prod(qh, u, &th, &tl);
        tl += ql * u;
        double sh, sl;
        two_sum(a, th, &sh, &sl);
        sl += tl;
        out[k - 1] = sh + sl;
    }
    out[n - 1] = b;                     /* Julia ranges hit the endpoint exactly */
}

/* ------------------------------------------------------------------------------------------
 * a4: Q4  (NonObstacle.jl:103-105).  I,J are 1-based as in the reference.
 * ---------------------------------------------------------------------------------------- */
#define C_(I, J) c[((I) - 1) + (int64_t)((J) - 1) * N]
void orc_q4(const double *c, int N, int I, int J, double dx, double q[4])
{
    q[0] = C_(I, J);
    q[1] = (C_(I + 1, J) - C_(I, J)) / dx;
    q[2] = (C_(I, J + 1) - C_(I, J)) / dx;
    q[3] = (((C_(I + 1, J + 1) + C_(I, J)) - C_(I + 1, J)) - C_(I, J + 1)) / (dx * dx);
}

/* a5: table build (NonObstacle.jl:191-210; duplicated Obstacle.jl:224-241, 277-294).
 * eval (N-1,N-1,4), grad (N-2,N-2,8), column-major; grad entries with I==1 or J==1 are never
 * written by the reference (SharedArray is zero-filled) -> zero here.                        */
void orc_build_tables(const double *c, int N, double ext, double *eval, double *grad)
{
    const int64_t n1 = N - 1, n2 = N - 2;
    double dx = 2 * ext / (N - 1);
    memset(grad, 0, sizeof(double) * (size_t)(n2 * n2 * 8));
    for (int I = 1; I <= N - 1; ++I)
        for (int J = 1; J <= N - 1; ++J) {
            double q[4];
            orc_q4(c, N, I, J, dx, q);
            for (int k = 0; k < 4; ++k) eval[(I - 1) + (J - 1) * n1 + k * n1 * n1] = q[k];
        }
#define EV(I, J, k) eval[((I) - 1) + ((J) - 1) * n1 + (int64_t)(k) * n1 * n1]
    for (int I = 2; I <= N - 2; ++I)
        for (int J = 2; J <= N - 2; ++J)
            for (int k = 0; k < 4; ++k) {
                grad[(I - 1) + (J - 1) * n2 + (int64_t)k * n2 * n2] =
                    (EV(I + 1, J, k) - EV(I - 1, J, k)) / (2 * dx);
                grad[(I - 1) + (J - 1) * n2 + (int64_t)(k + 4) * n2 * n2] =
                    (EV(I, J + 1, k) - EV(I, J - 1, k)) / (2 * dx);
            }
#undef EV
}

/* cell index: I = Int64(floor((x - p[1]) / h)) + 1  (NonObstacle.jl:33-34).
 * The reference indexes out of bounds (UB under @inbounds) when I is outside [1, N-2];
 * here the index is clamped and *oob set.                                                   */
static int cell_index(double x, double p1, double h, int N, int *oob)
{
    double q = floor((x - p1) / h);
    int I;
    if (!(q >= 0.0)) { I = 1; *oob = 1; }                /* also catches NaN */
    else if (q > (double)(N - 3)) { I = N - 2; *oob = 1; }
    else I = (int)q + 1;
    return I;
}

/* bilinear table value  z' * tab[I,J,k0:k0+3]  left to right (NonObstacle.jl:37-41) */
static double zdot(double dx, double dy, double dxdy, const double *tab, int64_t base,
                   int64_t plane, int k0)
{
    double s = 1.0 * tab[base + (k0 + 0) * plane];
    s = s + dx * tab[base + (k0 + 1) * plane];
    s = s + dy * tab[base + (k0 + 2) * plane];
    s = s + dxdy * tab[base + (k0 + 3) * plane];
    return s;
}

/* a2: DiscreteHamilton without Hessian (NonObstacle.jl:30-52); a3's H output is identical
 * (NonObstacle.jl:6-18).  Returns 1 if an index had to be clamped.                          */
int orc_discrete_hamilton(const double X[4], const double *eval, const double *grad,
                          const double *p, int N, double H[4])
{
    int oob = 0;
    const int64_t n1 = N - 1, n2 = N - 2;
    double h = p[1] - p[0];
    int I = cell_index(X[0], p[0], h, N, &oob);
    int J = cell_index(X[1], p[0], h, N, &oob);
    double dx = X[0] - p[I - 1], dy = X[1] - p[J - 1];
    double dxdy = dx * dy;
    double tau = X[2] * X[2] + X[3] * X[3];
    double c = zdot(dx, dy, dxdy, eval, (I - 1) + (J - 1) * n1, n1 * n1, 0);
    double gcX = zdot(dx, dy, dxdy, grad, (I - 1) + (J - 1) * n2, n2 * n2, 0);
    double gcY = zdot(dx, dy, dxdy, grad, (I - 1) + (J - 1) * n2, n2 * n2, 4);
    double c2 = c * c;
    H[0] = c2 * X[2];
    H[1] = c2 * X[3];
    H[2] = ((-gcX) * c) * tau;
    H[3] = ((-gcY) * c) * tau;
    return oob;
}

/* ---- "fused" flavour --------------------------------------------------------------------------
 * The reference evaluates these functions under @fastmath (NonObstacle.jl:1,6,30,54), which lets
 * the compiler reassociate and contract.  This flavour spells out ONE such evaluation explicitly --
 * the one the HIP kernels use in RAY_ARITH_FUSED -- so that the kernel can be checked bit for bit;
 * against the as-written flavour above it differs by rounding only (tests bound it at 1e-10 with
 * identical events).
 *   cell coordinate  q = fma(x, rh, nb),  rh = RN(1/h), nb = RN(-p[1]*rh);  cell = trunc(q)  (== floor
 *               for q >= 0; the reference is undefined below the grid anyway: q in (-1,0) -> cell 1)
 *   in-cell offset   in CELL units, u = q - floor(q), kept below 1 (what v_fract_f64 returns), in place of
 *               dx = x - p[I]: algebraically u = dx/h
 *   dot         with the coefficients in cell units, t' = t .* [1, h, h, h*h] (one rounding each):
 *               fma(u, fma(v, t3', t1'), fma(v, t2', t0))
 *   tau         fma(py, py, px*px);  H[2:3] = (-g) * (c*tau)
 *   RK4         Y = fma(H, dt/2, X) / fma(H, dt, X);  acc = fma(2, H, acc);
 *               X += fma(acc + H4, dt*RN(1/6), X)                                              */
static int cell_index_fused(double x, double rh, double nb, int N, int *oob, double *u)
{
    double q = fma(x, rh, nb);
    double f = q - floor(q);
    if (f >= 1.0) f = 0x1.fffffffffffffp-1;               /* q = -tiny: the subtraction rounds to 1 */
    *u = f;
    q = trunc(q);
    int I;
    if (!(q > -1.0)) { I = 1; *oob = 1; }                 /* q <= -1 or NaN */
    else if (q > (double)(N - 3)) { I = N - 2; *oob = 1; }
    else I = (int)q + 1;
    return I;
}
static double zdot_fused(double u, double v, double h, const double *tab, int64_t base,
                         int64_t plane, int k0)
{
    const double h2 = h * h;
    const double t1 = tab[base + (k0 + 1) * plane] * h, t2 = tab[base + (k0 + 2) * plane] * h;
    const double t3 = tab[base + (k0 + 3) * plane] * h2;
    return fma(u, fma(v, t3, t1), fma(v, t2, tab[base + (k0 + 0) * plane]));
}
int orc_discrete_hamilton_fused(const double X[4], const double *eval, const double *grad,
                                const double *p, int N, double H[4])
{
    int oob = 0;
    const int64_t n1 = N - 1, n2 = N - 2;
    double h = p[1] - p[0];
    double rh = 1.0 / h, nb = -p[0] * rh;
    double u, v;
    int I = cell_index_fused(X[0], rh, nb, N, &oob, &u);
    int J = cell_index_fused(X[1], rh, nb, N, &oob, &v);
    double tau = fma(X[3], X[3], X[2] * X[2]);
    double c = zdot_fused(u, v, h, eval, (I - 1) + (J - 1) * n1, n1 * n1, 0);
    double gcX = zdot_fused(u, v, h, grad, (I - 1) + (J - 1) * n2, n2 * n2, 0);
    double gcY = zdot_fused(u, v, h, grad, (I - 1) + (J - 1) * n2, n2 * n2, 4);
    double c2 = c * c, w = c * tau;
    H[0] = c2 * X[2];
    H[1] = c2 * X[3];
    H[2] = (-gcX) * w;
    H[3] = (-gcY) * w;
    return oob;
}
static int rk4_step_fused(const double *eval, const double *grad, const double *p, int N,
                          double X[4], double dt)
{
    double H[4], Y[4], acc[4];
    const double hdt = 0.5 * dt, dt6 = dt * 0.16666666666666666;
    int oob = 0;
    oob |= orc_discrete_hamilton_fused(X, eval, grad, p, N, H);
    for (int k = 0; k < 4; ++k) { acc[k] = H[k]; Y[k] = fma(H[k], hdt, X[k]); }
    oob |= orc_discrete_hamilton_fused(Y, eval, grad, p, N, H);
    for (int k = 0; k < 4; ++k) { acc[k] = fma(2.0, H[k], acc[k]); Y[k] = fma(H[k], hdt, X[k]); }
    oob |= orc_discrete_hamilton_fused(Y, eval, grad, p, N, H);
    for (int k = 0; k < 4; ++k) { acc[k] = fma(2.0, H[k], acc[k]); Y[k] = fma(H[k], dt, X[k]); }
    oob |= orc_discrete_hamilton_fused(Y, eval, grad, p, N, H);
    for (int k = 0; k < 4; ++k) X[k] = fma(acc[k] + H[k], dt6, X[k]);
    return oob;
}

/* gridded c(x,y) alone: z' * eval[I,J,:] (used for c(sensor) when shooting on a grid) */
static double grid_speed(double x, double y, const double *eval, const double *p, int N, int *oob)
{
    const int64_t n1 = N - 1;
    double h = p[1] - p[0];
    int I = cell_index(x, p[0], h, N, oob);
    int J = cell_index(y, p[0], h, N, oob);
    double dx = x - p[I - 1], dy = y - p[J - 1];
    return zdot(dx, dy, dx * dy, eval, (I - 1) + (J - 1) * n1, n1 * n1, 0);
}

/* ---- "reproducible" sin/cos flavour (med->trig == 1) ---------------------------------------------
 * The reference calls Julia's sin and cos; any two math libraries differ in the last bit here and there,
 * and on the analytic media that is the only thing between the HIP kernels and this file.  This flavour
 * fixes one evaluation as a sequence of single IEEE roundings (+, -, *, fma, rint) -- the classical
 * one: Cody-Waite reduction by pi/2 (a 33-bit head, so that n * head is exact), then the minimax
 * polynomials for sin and cos on [-pi/4, pi/4] with a two-double argument (Sun fdlibm's published
 * coefficients).  Within 1 ulp of the exact value for |a| <= 100 (tests/test_trig.py), i.e. the same
 * class of difference as between two libms; the kernels evaluate the same sequence, so under this
 * flavour analytic media compare bit for bit.  |a| >= 2^18, inf, NaN: libm (not reproducible; nothing
 * in the unit disk gets there).
 * Inside an RK4 step stages 2-4 advance the first stage's (sin, cos) by the small difference d of the
 * arguments (angle addition with Taylor sums of sin d and 1 - cos d, |d| <= 1/4; beyond: the full
 * evaluation): absolute error 1.6e-16 measured, against the 1.1e-16 of a correctly rounded value.   */
static const double REP_BOUND = 262144.0, REP_SHIFT_MAX = 0.25, REP_TINY_MAX = 0.0009765625;
static void rep_kernel(double y, double t, double *sn, double *cs)
{
    static const double S[6] = {-1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
                                2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10};
    static const double Cc[6] = {4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
                                 -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11};
    double z = y * y, v = z * y;
    double r = S[5];
    for (int k = 4; k >= 1; --k) r = fma(z, r, S[k]);
    double u = fma(-v, r, 0.5 * t);
    u = fma(z, u, -t);
    u = fma(-v, S[0], u);
    *sn = y - u;
    double q = Cc[5];
    for (int k = 4; k >= 0; --k) q = fma(z, q, Cc[k]);
    q = z * q;
    double hz = 0.5 * z, w = 1.0 - hz;
    *cs = w + (((1.0 - w) - hz) + fma(z, q, -(y * t)));
}
void orc_rep_sincos(double a, double *sn, double *cs)
{
    if (!(fabs(a) < REP_BOUND)) { *sn = sin(a); *cs = cos(a); return; }
    double fn = rint(a * 6.36619772367581382433e-01);
    int n = (int)fn;
    double r = fma(-fn, 1.57079632673412561417e+00, a);         /* exact */
    double w = fn * 6.07710050650619224932e-11;
    double y = r - w, t = (r - y) - w;
    double ks, kc;
    rep_kernel(y, t, &ks, &kc);
    double ss = (n & 1) ? kc : ks, cc = (n & 1) ? ks : kc;
    *sn = (n & 2) ? -ss : ss;
    *cs = ((n + 1) & 2) ? -cc : cc;
}
void orc_rep_shift(double s0, double c0, double d, double *sn, double *cs)
{
    static const double Ts[5] = {-1.66666666666666666667e-01, 8.33333333333333333333e-03, -1.98412698412698412698e-04,
                                 2.75573192239858906526e-06, -2.50521083854417187751e-08};
    static const double Tc[6] = {0.5, -4.16666666666666666667e-02, 1.38888888888888888889e-03, -2.48015873015873015873e-05,
                                 2.75573192239858906526e-07, -2.08767569878680989792e-09};
    double z = d * d;
    double ps = Ts[4];
    for (int k = 3; k >= 0; --k) ps = fma(z, ps, Ts[k]);
    double sd = d + (d * z) * ps;                                 /* sin d */
    double pc = Tc[5];
    for (int k = 4; k >= 0; --k) pc = fma(z, pc, Tc[k]);
    double v = z * pc;                                            /* 1 - cos d */
    *sn = s0 + fma(c0, sd, -(s0 * v));
    *cs = c0 - fma(s0, sd, c0 * v);
}
/* the same advance with its multiply-adds contracted (the kernels' kFused analytic flavour, ray_trig.h shift_fused) */
void orc_rep_shift_fused(double s0, double c0, double d, double *sn, double *cs)
{
    static const double Ts[5] = {-1.66666666666666666667e-01, 8.33333333333333333333e-03, -1.98412698412698412698e-04,
                                 2.75573192239858906526e-06, -2.50521083854417187751e-08};
    static const double Tc[6] = {0.5, -4.16666666666666666667e-02, 1.38888888888888888889e-03, -2.48015873015873015873e-05,
                                 2.75573192239858906526e-07, -2.08767569878680989792e-09};
    double z = d * d;
    double ps = Ts[4];
    for (int k = 3; k >= 0; --k) ps = fma(z, ps, Ts[k]);
    double sd = fma(d * z, ps, d);                                /* sin d */
    double pc = Tc[5];
    for (int k = 4; k >= 0; --k) pc = fma(z, pc, Tc[k]);
    double v = z * pc;                                            /* 1 - cos d */
    *sn = fma(c0, sd, fma(-s0, v, s0));
    *cs = fma(-s0, sd, fma(-c0, v, c0));
}
/* the advance by a tiny difference, |d| <= REP_TINY_MAX (ray_trig.h shift_tiny): sin d = d - d^3/6, 1 - cos d = d^2/2 - d^4/24 */
void orc_rep_shift_tiny(double s0, double c0, double d, double *sn, double *cs)
{
    double z = d * d;
    double sd = fma(d * z, -1.66666666666666666667e-01, d);
    double v = z * fma(z, -4.16666666666666666667e-02, 0.5);
    *sn = fma(c0, sd, fma(-s0, v, s0));
    *cs = fma(-s0, sd, fma(-c0, v, c0));
}
/* steps one full evaluation anchors (ray_trig.h anchor_period): the argument w x moves by at most |w| dt cmax a step */
int orc_anchor_period(double w, double dt, double cmax)
{
    double k = floor(0.24 / (fabs(w) * dt * cmax));
    return k >= 64.0 ? 64 : (k >= 1.0 ? (int)k : 1);
}
static double med_sin(const orc_medium *med, double a)
{
    if (!med->trig) return sin(a);
    double s, c;
    orc_rep_sincos(a, &s, &c);
    return s;
}

/* a12 media (NonObstacleDemo.jl:12,23; ObstacleDemo.jl:12,18) */
static double sinprod_c(const orc_medium *med, double x, double y)
{
    const orc_sinprod *sp = &med->sp;
    return 1 + sp->A * med_sin(med, sp->wx * x) * med_sin(med, sp->wy * y);
}

/* a1: Hamilton (NonObstacle.jl:1-4):  [speed^2 * p ; -(p'p) * gradc * speed] */
static void hamilton_from_trig(const double X[4], const orc_sinprod *sp, double sx, double cx, double sy, double cy,
                               double H[4])
{
    double speed = 1 + sp->A * sx * sy;
    double g0 = sp->gx * (cx * sy);
    double g1 = sp->gy * (sx * cy);
    double tau = X[2] * X[2] + X[3] * X[3];
    double s2 = speed * speed;
    H[0] = s2 * X[2];
    H[1] = s2 * X[3];
    H[2] = ((-tau) * g0) * speed;
    H[3] = ((-tau) * g1) * speed;
}
void orc_hamilton_sinprod(const double X[4], const orc_sinprod *sp, double H[4])
{
    double sx = sin(sp->wx * X[0]), cx = cos(sp->wx * X[0]);
    double sy = sin(sp->wy * X[1]), cy = cos(sp->wy * X[1]);
    hamilton_from_trig(X, sp, sx, cx, sy, cy, H);
}
/* the reproducible flavour: stage 0 evaluates and remembers (argument, sin, cos) per coordinate in base[6],
 * stages 1..3 advance from it */
static void rep_stage(int stage, double a, double base[3], double *sn, double *cs)
{
    if (stage == 0) {
        orc_rep_sincos(a, sn, cs);
        base[0] = a; base[1] = *sn; base[2] = *cs;
        return;
    }
    double d = a - base[0];
    if (fabs(d) <= REP_SHIFT_MAX) orc_rep_shift(base[1], base[2], d, sn, cs);
    else orc_rep_sincos(a, sn, cs);
}
static void hamilton_sinprod_rep(const double X[4], const orc_sinprod *sp, int stage, double base[6], double H[4])
{
    double sx, cx, sy, cy;
    rep_stage(stage, sp->wx * X[0], base, &sx, &cx);
    rep_stage(stage, sp->wy * X[1], base + 3, &sy, &cy);
    hamilton_from_trig(X, sp, sx, cx, sy, cy, H);
}

/* a12 alternates (NonObstacleDemo.jl:7-11,15-22; ObstacleDemo.jl:9-11,15-17), left to right as written:
 *   1.0 + 0.2 * sin(pi * r) + 0.4 * sin(pi * v) + ...
 *   0.2 * pi * cos(pi * r)/r * [(x-0.5), (y-0.2)] + 0.4 * pi * cos(pi * v)/v * [...] + ...        */
static double radial_c(const orc_medium *med, double x, double y)
{
    const orc_radial *rd = &med->rad;
    double c = 1.0;
    for (int k = 0; k < rd->K; ++k) {
        double dx = x - rd->x[k], dy = y - rd->y[k];
        double r = sqrt(dx * dx + dy * dy);
        c = c + rd->a[k] * med_sin(med, M_PI * r);
    }
    return c;
}
/* reproducible flavour inside an RK4 step: sin and cos of pi r_k by stage (rep_stage), base + 3k per term */
static void hamilton_radial_rep(const double X[4], const orc_medium *med, int stage, double base[12], double H[4])
{
    const orc_radial *rd = &med->rad;
    double speed = 1.0, g[2] = {0, 0};
    for (int k = 0; k < rd->K; ++k) {
        double dx = X[0] - rd->x[k], dy = X[1] - rd->y[k];
        double r = sqrt(dx * dx + dy * dy);
        double sn, cs;
        rep_stage(stage, M_PI * r, base + 3 * k, &sn, &cs);
        speed = speed + rd->a[k] * sn;
        double f = ((rd->a[k] * M_PI) * cs) / r;
        if (k == 0) { g[0] = f * dx; g[1] = f * dy; }
        else { g[0] = g[0] + f * dx; g[1] = g[1] + f * dy; }
    }
    double tau = X[2] * X[2] + X[3] * X[3];
    double s2 = speed * speed;
    H[0] = s2 * X[2];
    H[1] = s2 * X[3];
    H[2] = ((-tau) * g[0]) * speed;
    H[3] = ((-tau) * g[1]) * speed;
}
static void radial_grad(const orc_medium *med, double x, double y, double g[2])
{
    const orc_radial *rd = &med->rad;
    g[0] = g[1] = 0;
    for (int k = 0; k < rd->K; ++k) {
        double dx = x - rd->x[k], dy = y - rd->y[k];
        double r = sqrt(dx * dx + dy * dy);
        double cs = cos(M_PI * r), sn;
        if (med->trig) orc_rep_sincos(M_PI * r, &sn, &cs);
        double f = ((rd->a[k] * M_PI) * cs) / r;
        if (k == 0) { g[0] = f * dx; g[1] = f * dy; }
        else { g[0] = g[0] + f * dx; g[1] = g[1] + f * dy; }
    }
}
/* a1: Hamilton (NonObstacle.jl:1-4) with those closures */
static void hamilton_radial(const double X[4], const orc_medium *med, double H[4])
{
    double speed = radial_c(med, X[0], X[1]);
    double g[2];
    radial_grad(med, X[0], X[1], g);
    double tau = X[2] * X[2] + X[3] * X[3];
    double s2 = speed * speed;
    H[0] = s2 * X[2];
    H[1] = s2 * X[3];
    H[2] = ((-tau) * g[0]) * speed;
    H[3] = ((-tau) * g[1]) * speed;
}

void orc_hamilton_radial(const double X[4], const orc_radial *rd, double H[4])
{
    orc_medium med;
    memset(&med, 0, sizeof med);
    med.kind = 2;
    med.rad = *rd;
    hamilton_radial(X, &med, H);
}

/* ---- "fused" flavour of the analytic right-hand sides (med->arith == 1, kinds 0 and 2) -------------------------
 * Hamilton is @fastmath in the reference (NonObstacle.jl:1), so its multiply-adds may be contracted.  This flavour
 * spells out the evaluation the HIP kernels use by default (ray_device.h rhs_sinprod / rhs_radial<kFused>):
 *   - sin/cos: the reproducible evaluation above; a full evaluation ANCHORS the advances of K consecutive steps
 *     (orc_anchor_period: the argument cannot leave the advance's range in K steps), every other look-up is ONE
 *     contracted advance from the anchor, argument difference d = fma(w, x, -anchor) -- except that stage 3 (O(dt^2)
 *     from stage 2) and a later step's stage 1 (O(dt^3) from the stage 4 before it) follow from the stage BEFORE them by
 *     orc_rep_shift_tiny where the two differences are within REP_TINY_MAX of each other; a look-up beyond the range
 *     evaluates in full and -- at a first stage -- re-anchors;
 *   - speed = fma(A sx, sy, 1), tau = fma(py, py, px px), H[2:3] = ((-g) (speed tau)) (cos sin);
 *   - the RK4 combination of the gridded fused flavour: Y = fma(H, dt/2, X), Xn = fma(sum, dt RN(1/6), X).        */
typedef struct { double base[12]; double prev[12]; int acnt[4]; int K[4]; } trig_state;
static void trig_state_init(trig_state *ts, const orc_medium *med, double dt)
{
    memset(ts, 0, sizeof *ts);
    for (int k = 0; k < 4; ++k) ts->K[k] = 1;
    if (med->kind == 0) {
        double cmax = 1 + fabs(med->sp.A);
        ts->K[0] = orc_anchor_period(med->sp.wx, dt, cmax);
        ts->K[1] = orc_anchor_period(med->sp.wy, dt, cmax);
    } else if (med->kind == 2) {
        double cmax = 1;
        for (int k = 0; k < med->rad.K; ++k) cmax += fabs(med->rad.a[k]);
        for (int k = 0; k < med->rad.K; ++k) ts->K[k] = orc_anchor_period(M_PI, dt, cmax);
    }
}
static void fused_stage(int stage, int anchor, double w, double x, double base[3], double prev[3], double *sn, double *cs)
{
    if (stage == 0 && anchor) {
        double a = w * x;
        orc_rep_sincos(a, sn, cs);
        base[0] = a; base[1] = *sn; base[2] = *cs;
        return;
    }
    double d = fma(w, x, -base[0]);
    double dl = d - prev[0];                       /* prev: the stage before (set by stages 2 and 4 only) */
    if ((stage == 0 || stage == 2) && fabs(dl) <= REP_TINY_MAX) orc_rep_shift_tiny(prev[1], prev[2], dl, sn, cs);
    else if (fabs(d) <= REP_SHIFT_MAX) orc_rep_shift_fused(base[1], base[2], d, sn, cs);
    else {
        double a = w * x;
        orc_rep_sincos(a, sn, cs);
        if (stage == 0) { base[0] = a; base[1] = *sn; base[2] = *cs; }
    }
    if (stage == 1 || stage == 3) { prev[0] = d; prev[1] = *sn; prev[2] = *cs; }
}
static void hamilton_sinprod_fused(const double X[4], const orc_sinprod *sp, int stage, unsigned anchor, double base[12], double prev[12], double H[4])
{
    double sx, cx, sy, cy;
    fused_stage(stage, anchor & 1u, sp->wx, X[0], base, prev, &sx, &cx);
    fused_stage(stage, anchor & 2u, sp->wy, X[1], base + 3, prev + 3, &sy, &cy);
    double speed = fma(sp->A * sx, sy, 1.0);
    double tau = fma(X[3], X[3], X[2] * X[2]);
    double s2 = speed * speed, w = speed * tau;
    H[0] = s2 * X[2];
    H[1] = s2 * X[3];
    H[2] = ((-sp->gx) * w) * (cx * sy);
    H[3] = ((-sp->gy) * w) * (sx * cy);
}
static void hamilton_radial_fused(const double X[4], const orc_medium *med, int stage, unsigned anchor, double base[12], double prev[12], double H[4])
{
    const orc_radial *rd = &med->rad;
    double speed = 1.0, g0 = 0, g1 = 0;
    for (int k = 0; k < rd->K; ++k) {
        double dx = X[0] - rd->x[k], dy = X[1] - rd->y[k];
        double r = sqrt(fma(dy, dy, dx * dx));
        double sn, cs;
        fused_stage(stage, anchor & (1u << k), M_PI, r, base + 3 * k, prev + 3 * k, &sn, &cs);
        double f = ((rd->a[k] * M_PI) * cs) / r;
        speed = fma(rd->a[k], sn, speed);
        g0 = fma(f, dx, g0);
        g1 = fma(f, dy, g1);
    }
    double s2 = speed * speed;
    double tau = fma(X[3], X[3], X[2] * X[2]);
    double w = speed * tau;
    H[0] = s2 * X[2];
    H[1] = s2 * X[3];
    H[2] = (-g0) * w;
    H[3] = (-g1) * w;
}
static void rk4_step_analytic_fused(const orc_medium *med, double X[4], double dt, trig_state *ts)
{
    double H[4], Y[4], acc[4];
    const double hdt = 0.5 * dt, dt6 = dt * 0.16666666666666666;
    unsigned anchor = 0;
    for (int k = 0; k < (med->kind == 0 ? 2 : 4); ++k) {
        if (ts->acnt[k] == 0) { anchor |= 1u << k; ts->acnt[k] = ts->K[k]; }
        --ts->acnt[k];
    }
#define RHS_F(P, STAGE)                                                                     \
    do {                                                                                    \
        if (med->kind == 0) hamilton_sinprod_fused(P, &med->sp, STAGE, anchor, ts->base, ts->prev, H); \
        else hamilton_radial_fused(P, med, STAGE, anchor, ts->base, ts->prev, H);           \
    } while (0)
    RHS_F(X, 0);
    for (int k = 0; k < 4; ++k) { acc[k] = H[k]; Y[k] = fma(H[k], hdt, X[k]); }
    RHS_F(Y, 1);
    for (int k = 0; k < 4; ++k) { acc[k] = fma(2.0, H[k], acc[k]); Y[k] = fma(H[k], hdt, X[k]); }
    RHS_F(Y, 2);
    for (int k = 0; k < 4; ++k) { acc[k] = fma(2.0, H[k], acc[k]); Y[k] = fma(H[k], dt, X[k]); }
    RHS_F(Y, 3);
    for (int k = 0; k < 4; ++k) X[k] = fma(acc[k] + H[k], dt6, X[k]);
#undef RHS_F
}

/* stage = 0..3 inside an RK4 step, base = the step's scratch (the reproducible sinprod flavour only) */
static int rhs(const orc_medium *med, const double X[4], double H[4], int stage, double base[12])
{
    if (med->kind == 0 && med->trig) { hamilton_sinprod_rep(X, &med->sp, stage, base, H); return 0; }
    if (med->kind == 2 && med->trig) { hamilton_radial_rep(X, med, stage, base, H); return 0; }
    if (med->kind == 0) { orc_hamilton_sinprod(X, &med->sp, H); return 0; }
    if (med->kind == 2) { hamilton_radial(X, med, H); return 0; }
    return orc_discrete_hamilton(X, med->eval, med->grad, med->p, med->N, H);
}

/* a6 init (NonObstacle.jl:55-65; Obstacle.jl:12-22) */
void orc_init_states(const orc_medium *med, int ns, int nd, double dir_lo, double dir_hi,
                     int src_begin, int src_end, double *m, int64_t ld)
{
    double *source = (double *)malloc(sizeof(double) * (size_t)(ns + 1));
    double *direct = (double *)malloc(sizeof(double) * (size_t)(nd + 2));
    orc_linspace(0.0, 2 * M_PI, ns + 1, source);
    orc_linspace(dir_lo, dir_hi, nd + 2, direct);
    for (int i = src_begin; i < src_end; ++i) {
        double sx = cos(source[i]), sy = sin(source[i]);
        if (med->trig) orc_rep_sincos(source[i], &sy, &sx);
        int oob = 0;
        double cs = med->kind == 0 ? sinprod_c(med, sx, sy)
                  : med->kind == 2 ? radial_c(med, sx, sy)
                                   : grid_speed(sx, sy, med->eval, med->p, med->N, &oob);
        for (int j = 0; j < nd; ++j) {
            double theta = source[i] + M_PI * 0.5 + direct[j + 1];
            int64_t r = (int64_t)(i - src_begin) * nd + j;
            m[r + 0 * ld] = sx;
            m[r + 1 * ld] = sy;
            double ct = cos(theta), st = sin(theta);
            if (med->trig) orc_rep_sincos(theta, &st, &ct);
            m[r + 2 * ld] = ct / cs;
            m[r + 3 * ld] = st / cs;
        }
    }
    free(source);
    free(direct);
}

static double norm2(double a, double b) { return sqrt(a * a + b * b); }

/* obstacle level set and normal (ObstacleDemo.jl:20-34): the shipped circle, and the commented
 * trefoil alternate  r - (0.4 - rho cos(3 theta))  (:24)  with
 * n = [x, y]/r - 3 rho sin(3 theta)/r * [-y, x]/r  (:31, a rho of its own), n/norm(n) (:33).
 * theta = atan2(y, x) enters only through cos(3 theta) and sin(3 theta).  Two flavours (ob->trig):
 *   0  as written: atan2, cos, sin of the host libm (results are defined up to that libm);
 *   1  "identity": cos(theta) = x/r, sin(theta) = y/r and the triple-angle identities
 *      cos 3t = c (4 c^2 - 3), sin 3t = s (3 - 4 s^2) -- only +,-,*,/,sqrt, every one a single IEEE
 *      rounding, so the result is the same bits on any conforming machine.  This is what the HIP
 *      kernels evaluate (ray_amd.hip ob_val / ob_normal); atan2(0, 0) = 0 -> cos(theta) = 1 at r = 0. */
static double ob_val(const orc_obstacle *ob, double x, double y)
{
    if (ob->kind == 2) {
        double r = norm2(x, y);
        if (ob->trig == 1) {
            double c = r > 0 ? x / r : 1.0;
            double cos3 = c * (4 * (c * c) - 3);
            return r - (ob->r0 - ob->rho_ob * cos3);
        }
        double theta = atan2(y, x);
        return r - (ob->r0 - ob->rho_ob * cos(3 * theta));
    }
    return norm2(x, y) - ob->r0;
}
static void ob_normal(const orc_obstacle *ob, double x, double y, double n[2])
{
    double r = norm2(x, y);
    double nx = x / r, ny = y / r;
    if (ob->kind == 2) {
        double sin3;
        if (ob->trig == 1) sin3 = ny * (3 - 4 * (ny * ny));
        else sin3 = sin(3 * atan2(y, x));
        double s = ((3 * ob->rho_n) * sin3) / r;                /* 3 * rho * sin(3 * theta)/r, left to right */
        nx = nx - (s * (-y)) / r;
        ny = ny - (s * x) / r;
    }
    double nn = norm2(nx, ny);
    n[0] = nx / nn;
    n[1] = ny / nn;
}
/* exported for the CPU test that bounds the identity flavour against the libm one */
double orc_ob_val(const orc_obstacle *ob, double x, double y) { return ob_val(ob, x, y); }
void orc_ob_normal(const orc_obstacle *ob, double x, double y, double n[2]) { ob_normal(ob, x, y, n); }

/* one classical RK4 step in the a6/a7 form: k = H*dt, X + k1/2, ... /6.0
 * (NonObstacle.jl:72-76; Obstacle.jl:34-38) */
static int rk4_step_kdt(const orc_medium *med, double X[4], double dt, trig_state *ts)
{
    double k1[4], k2[4], k3[4], k4[4], Y[4], H[4], base[12];
    int oob = 0;
    if (med->kind == 1 && med->arith == 1)
        return rk4_step_fused(med->eval, med->grad, med->p, med->N, X, dt);
    if (med->arith == 1) { rk4_step_analytic_fused(med, X, dt, ts); return 0; }
    oob |= rhs(med, X, H, 0, base);
    for (int k = 0; k < 4; ++k) { k1[k] = H[k] * dt; Y[k] = X[k] + k1[k] / 2; }
    oob |= rhs(med, Y, H, 1, base);
    for (int k = 0; k < 4; ++k) { k2[k] = H[k] * dt; Y[k] = X[k] + k2[k] / 2; }
    oob |= rhs(med, Y, H, 2, base);
    for (int k = 0; k < 4; ++k) { k3[k] = H[k] * dt; Y[k] = X[k] + k3[k]; }
    oob |= rhs(med, Y, H, 3, base);
    for (int k = 0; k < 4; ++k) {
        k4[k] = H[k] * dt;
        X[k] = X[k] + (((k1[k] + 2 * k2[k]) + 2 * k3[k]) + k4[k]) / 6.0;
    }
    return oob;
}

static void shoot_one(const orc_medium *med, const orc_obstacle *ob, double dt, int time_unit,
                      int max_steps, double *m, int64_t r, int64_t ld, int32_t *status,
                      int32_t *nsteps, int32_t *nreflect)
{
    double X[4], prev[4], E[4];
    for (int k = 0; k < 4; ++k) X[k] = m[r + k * ld];
    double t = 0;
    int rfl = 0, st = 0, steps = 0, nref = 0;
    const int has_ob = ob && ob->kind != 0;
    trig_state ts;
    trig_state_init(&ts, med, dt);
    for (;;) {
        if (steps >= max_steps) { st |= ORC_ST_STEP_CAP; break; }
        if (has_ob && ob_val(ob, X[0], X[1]) > 0) rfl = 0;          /* Obstacle.jl:28-30 */
        for (int k = 0; k < 4; ++k) prev[k] = X[k];
        if (rk4_step_kdt(med, X, dt, &ts)) st |= ORC_ST_OOB;
        ++steps;
        t = time_unit ? t + 1 : t + dt;            /* Obstacle.jl:39 / NonObstacle.jl:77 */
        if (norm2(X[0], X[1]) >= 1) {              /* NonObstacle.jl:78-93; Obstacle.jl:40-57 */
            double lo = 0, hi = 1, mid, e;
            int it = 0;
            mid = (lo + hi) / 2;
            for (int k = 0; k < 4; ++k) E[k] = (1 - mid) * prev[k] + mid * X[k];
            e = norm2(E[0], E[1]) - 1;
            while (fabs(e) > 1e-15) {
                if (++it > ORC_BISECT_CAP) { st |= ORC_ST_BISECT_CAP; break; }
                if (e > 0) hi = mid; else lo = mid;
                mid = (lo + hi) / 2;
                for (int k = 0; k < 4; ++k) E[k] = (1 - mid) * prev[k] + mid * X[k];
                e = norm2(E[0], E[1]) - 1;
            }
            t = time_unit ? t - (1 - mid) : t - (1 - mid) * dt;
            for (int k = 0; k < 4; ++k) X[k] = E[k];
            st |= ORC_ST_EXITED;
            break;
        }
        if (has_ob && ob_val(ob, X[0], X[1]) <= 0 && rfl == 0) {    /* Obstacle.jl:69-91 */
            double lo = 0, hi = 1, mid, e, n[2];
            int it = 0;
            rfl = 1;
            mid = (lo + hi) / 2;
            for (int k = 0; k < 4; ++k) E[k] = (1 - mid) * prev[k] + mid * X[k];
            e = ob_val(ob, E[0], E[1]);
            while (fabs(e) > 1e-15) {
                if (++it > ORC_BISECT_CAP) { st |= ORC_ST_BISECT_CAP; break; }
                if (e > 0) lo = mid; else hi = mid;
                mid = (lo + hi) / 2;
                for (int k = 0; k < 4; ++k) E[k] = (1 - mid) * prev[k] + mid * X[k];
                e = ob_val(ob, E[0], E[1]);
            }
            ob_normal(ob, E[0], E[1], n);                             /* Obstacle.jl:84 */
            /* X[3:4] = (eye(2) - 2*n*n') * R[3:4]                       Obstacle.jl:85 */
            double a0 = 2 * n[0], a1 = 2 * n[1];
            double m11 = 1 - a0 * n[0], m12 = 0 - a0 * n[1];
            double m21 = 0 - a1 * n[0], m22 = 1 - a1 * n[1];
            X[0] = E[0];
            X[1] = E[1];
            X[2] = m11 * E[2] + m12 * E[3];
            X[3] = m21 * E[2] + m22 * E[3];
            ++nref;
        }
    }
    for (int k = 0; k < 4; ++k) m[r + (4 + k) * ld] = X[k];
    m[r + 8 * ld] = t;
    if (status) status[r] = st;
    if (nsteps) nsteps[r] = steps;
    if (nreflect) nreflect[r] = nref;
}

void orc_shoot(const orc_medium *med, const orc_obstacle *ob, double dt, int time_unit,
               int max_steps, double *m, int64_t num, int64_t ld, int32_t *status,
               int32_t *nsteps, int32_t *nreflect, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t r = 0; r < num; ++r)
        shoot_one(med, ob, dt, time_unit, max_steps, m, r, ld, status, nsteps, nreflect);
}

/* RK4 step in the a8/a9 form: k = H, X + k/2*dt, X += (k1+2k2+2k3+k4)*dt/6.0
 * (NonObstacle.jl:161-164,171; Obstacle.jl:312-317) */
static int rk4_step_k(const double *eval, const double *grad, const double *p, int N,
                      double X[4], double dt)
{
    double k1[4], k2[4], k3[4], k4[4], Y[4];
    int oob = 0;
    oob |= orc_discrete_hamilton(X, eval, grad, p, N, k1);
    for (int k = 0; k < 4; ++k) Y[k] = X[k] + k1[k] / 2 * dt;
    oob |= orc_discrete_hamilton(Y, eval, grad, p, N, k2);
    for (int k = 0; k < 4; ++k) Y[k] = X[k] + k2[k] / 2 * dt;
    oob |= orc_discrete_hamilton(Y, eval, grad, p, N, k3);
    for (int k = 0; k < 4; ++k) Y[k] = X[k] + k3[k] * dt;
    oob |= orc_discrete_hamilton(Y, eval, grad, p, N, k4);
    for (int k = 0; k < 4; ++k)
        X[k] = X[k] + (((k1[k] + 2 * k2[k]) + 2 * k3[k]) + k4[k]) * dt / 6.0;
    return oob;
}

#define ORC_FIXED_STEP_CAP 10000000

void orc_trace_fixed_time(const double *eval, const double *grad, const double *p, int N,
                          double timeStep, int mode, int arith, const double *m_in, int64_t num,
                          int64_t ld, double *X_out, int64_t ldx, int32_t *status, int32_t *nsteps,
                          int nthreads)
{
    (void)nthreads;
    const double cease_r = 1 + 2 * (p[1] - p[0]);                  /* NonObstacle.jl:177 */
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < num; ++i) {
        double X[4];
        for (int k = 0; k < 4; ++k) X[k] = m_in[i + k * ld];
        const double Ti = m_in[i + 8 * ld];
        double t = 0, dt = timeStep;
        int st = 0, steps = 0;
        while (t < Ti) {
            if (steps >= ORC_FIXED_STEP_CAP) { st |= ORC_ST_STEP_CAP; break; }
            if (mode == 0 && (t + dt) > Ti) dt = Ti - t;          /* NonObstacle.jl:153-156 */
            t = t + dt;                                           /* :157 / Obstacle.jl:311 */
            if (arith ? rk4_step_fused(eval, grad, p, N, X, dt) : rk4_step_k(eval, grad, p, N, X, dt)) st |= ORC_ST_OOB;
            ++steps;
            if (mode == 0 && norm2(X[0], X[1]) > cease_r) {       /* NonObstacle.jl:177-181 */
                t = t + (Ti - t);
                st |= ORC_ST_CEASED;
            }
        }
        for (int k = 0; k < 4; ++k) X_out[i + k * ldx] = X[k];
        if (status) status[i] = st;
        if (nsteps) nsteps[i] = steps;
    }
}

/* a11: NonObstacle.jl:287,336,372; ObstacleDemo.jl:202,219-226,252 */
void orc_misfit(const double *m, const double *obs, int64_t num, int64_t ld, double *mismatch,
                double out[2], uint8_t *mask, double thresh)
{
    double sm = 0, st = 0;
    for (int64_t r = 0; r < num; ++r) {
        double d[4], a[4];
        for (int k = 0; k < 4; ++k) {
            a[k] = m[r + (4 + k) * ld];
            d[k] = a[k] - obs[r + (4 + k) * ld];
            if (mismatch) mismatch[4 * r + k] = d[k];
            sm = sm + d[k] * d[k];
            st = st + a[k] * a[k];
        }
        if (mask) {
            double nd_ = sqrt(((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3]);
            double na = sqrt(((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]) + a[3] * a[3]);
            mask[r] = (nd_ / na < thresh) ? 1 : 0;
        }
    }
    out[0] = sm;
    out[1] = st;
}


/* ==========================================================================================
 * NEXT-1: Jacobian rho and Frechet matrix M of ScatterForwardOperator.
 * NonObstacle.jl:6-28 (DiscreteHamilton with Hessian), :107-139 (dV), :141-187 (ChunkProcessing!),
 * :212-218 (hess table), :242 (droptol).
 * The reference goes through LAPACK `inv` and BLAS gemm here, so no restatement can be bit-exact
 * to it; operation order below is the natural one (k ascending) and parity is by tolerance.
 * ======================================================================================== */

/* hess (N-3,N-3,12) column-major; entries outside I,J in 3..N-3 are never written -> zero.
 * NonObstacle.jl:212-218 */
void orc_build_hess(const double *grad, int N, double ext, double *hess)
{
    const int64_t n2 = N - 2, n3 = N - 3;
    double dx = 2 * ext / (N - 1);
    memset(hess, 0, sizeof(double) * (size_t)(n3 * n3 * 12));
#define GR(I, J, k) grad[((I) - 1) + ((J) - 1) * n2 + (int64_t)(k) * n2 * n2]
    for (int I = 3; I <= N - 3; ++I)
        for (int J = 3; J <= N - 3; ++J)
            for (int k = 0; k < 4; ++k) {
                hess[(I - 1) + (J - 1) * n3 + (int64_t)k * n3 * n3] = (GR(I + 1, J, k) - GR(I - 1, J, k)) / (2 * dx);
                hess[(I - 1) + (J - 1) * n3 + (int64_t)(k + 4) * n3 * n3] = (GR(I, J + 1, k) - GR(I, J - 1, k)) / (2 * dx);
                hess[(I - 1) + (J - 1) * n3 + (int64_t)(k + 8) * n3 * n3] = (GR(I, J + 1, k + 4) - GR(I, J - 1, k + 4)) / (2 * dx);
            }
#undef GR
}

typedef struct {            /* everything one table look-up yields at a point */
    int I, J, oob;
    double dx, dy, c, gcX, gcY, tau;
} orc_point;

static void point_eval(const double X[4], const double *eval, const double *grad, const double *p, int N,
                       int arith, orc_point *q)
{
    const int64_t n1 = N - 1, n2 = N - 2;
    double h = p[1] - p[0];
    q->oob = 0;
    double u = 0, v = 0;
    if (arith) {
        double rh = 1.0 / h, nb = -p[0] * rh;
        q->I = cell_index_fused(X[0], rh, nb, N, &q->oob, &u);
        q->J = cell_index_fused(X[1], rh, nb, N, &q->oob, &v);
        q->dx = u * h;                                   /* physical offsets for the hess look-up and dV */
        q->dy = v * h;
    } else {
        q->I = cell_index(X[0], p[0], h, N, &q->oob);
        q->J = cell_index(X[1], p[0], h, N, &q->oob);
        q->dx = X[0] - p[q->I - 1];
        q->dy = X[1] - p[q->J - 1];
    }
    double dxdy = q->dx * q->dy;
    int64_t be = (q->I - 1) + (q->J - 1) * n1, bg = (q->I - 1) + (q->J - 1) * n2;
    if (arith) {
        q->tau = fma(X[3], X[3], X[2] * X[2]);
        q->c = zdot_fused(u, v, h, eval, be, n1 * n1, 0);
        q->gcX = zdot_fused(u, v, h, grad, bg, n2 * n2, 0);
        q->gcY = zdot_fused(u, v, h, grad, bg, n2 * n2, 4);
    } else {
        q->tau = X[2] * X[2] + X[3] * X[3];
        q->c = zdot(q->dx, q->dy, dxdy, eval, be, n1 * n1, 0);
        q->gcX = zdot(q->dx, q->dy, dxdy, grad, bg, n2 * n2, 0);
        q->gcY = zdot(q->dx, q->dy, dxdy, grad, bg, n2 * n2, 4);
    }
}

/* a3: DiscreteHamilton with Hessian: H and Jm = dH/dX (row-major 4x4).  NonObstacle.jl:6-28.
 * H is the value of the chosen arithmetic flavour (so the X trajectory equals orc_trace_fixed_time's).
 * hess look-ups outside I,J <= N-3 are undefined in the reference: clamped + flagged here.          */
static int hamilton_jac(const double X[4], const double *eval, const double *grad, const double *hess,
                        const double *p, int N, int arith, double H[4], double Jm[16])
{
    orc_point q;
    point_eval(X, eval, grad, p, N, arith, &q);
    int oob = q.oob;
    const int64_t n3 = N - 3;
    int Ih = q.I, Jh = q.J;
    if (Ih > N - 3) { Ih = N - 3; oob = 1; }
    if (Jh > N - 3) { Jh = N - 3; oob = 1; }
    double dxdy = q.dx * q.dy;
    int64_t bh = (Ih - 1) + (Jh - 1) * n3;
    double hXX = zdot(q.dx, q.dy, dxdy, hess, bh, n3 * n3, 0);
    double hXY = zdot(q.dx, q.dy, dxdy, hess, bh, n3 * n3, 4);
    double hYY = zdot(q.dx, q.dy, dxdy, hess, bh, n3 * n3, 8);
    double c = q.c, tau = q.tau, g[2] = {q.gcX, q.gcY}, hs[2][2] = {{hXX, hXY}, {hXY, hYY}};
    double c2 = c * c;
    if (arith) {
        double w = c * tau;
        H[0] = c2 * X[2]; H[1] = c2 * X[3]; H[2] = (-g[0]) * w; H[3] = (-g[1]) * w;
    } else {
        H[0] = c2 * X[2]; H[1] = c2 * X[3]; H[2] = ((-g[0]) * c) * tau; H[3] = ((-g[1]) * c) * tau;
    }
    /* M = [2*c*X[3:4]*g'  c^2*eye(2); -(c*h + g*g')*tau  -2*c*g*X[3:4]']    NonObstacle.jl:25 */
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            Jm[a * 4 + b] = ((2 * c) * X[2 + a]) * g[b];
            Jm[a * 4 + 2 + b] = (a == b) ? c2 : 0.0;
            Jm[(2 + a) * 4 + b] = (-(c * hs[a][b] + g[a] * g[b])) * tau;
            Jm[(2 + a) * 4 + 2 + b] = ((-2 * c) * g[a]) * X[2 + b];
        }
    return oob;
}

/* dV: 24 entries (row 0..3, 1-based linear node index, value).  NonObstacle.jl:107-139.
 * Node indices outside 1..N (I-1 < 1 or I+2 > N, same for J) would make the reference's sparse()
 * throw: such entries are dropped and flagged.                                                    */
static int dV(const double X[4], const double *eval, const double *grad, const double *p, int N, int arith,
              int rows[24], int64_t cols[24], double vals[24])
{
    orc_point q;
    point_eval(X, eval, grad, p, N, arith, &q);
    int oob = q.oob;
    double h = p[1] - p[0];
    double u = q.dx / h, v = q.dy / h;
    double phi[4] = {(1 - u) * (1 - v), u * (1 - v), (1 - u) * v, u * v};
    double psi[4] = {u * (1 - v), -(1 - u) * (1 - v), u * v, -(1 - u) * v};
    double gam[4] = {(1 - u) * v, u * v, -(1 - u) * (1 - v), -u * (1 - v)};
    double c = q.c, tau = q.tau;
    int I = q.I, J = q.J;
    int ci[24] = {I, I + 1, I, I + 1, I, I + 1, I, I + 1, I, I + 1, I, I + 1, I + 2, I - 1, I + 2, I - 1,
                  I, I + 1, I, I + 1, I, I + 1, I, I + 1};
    int cj[24] = {J, J, J + 1, J + 1, J, J, J + 1, J + 1, J, J, J + 1, J + 1, J, J, J + 1, J + 1,
                  J, J, J + 1, J + 1, J + 2, J + 2, J - 1, J - 1};
    for (int k = 0; k < 4; ++k) {
        rows[k] = 0;      vals[k] = 2 * c * X[2] * phi[k];
        rows[4 + k] = 1;  vals[4 + k] = 2 * c * X[3] * phi[k];
        rows[8 + k] = 2;  vals[8 + k] = -tau * q.gcX * phi[k] + c * tau * psi[k] / (2 * h);
        rows[12 + k] = 2; vals[12 + k] = -c * tau * psi[k] / (2 * h);
        rows[16 + k] = 3; vals[16 + k] = -tau * q.gcY * phi[k] + c * tau * gam[k] / (2 * h);
        rows[20 + k] = 3; vals[20 + k] = -c * tau * gam[k] / (2 * h);
    }
    for (int k = 0; k < 24; ++k) {
        if (ci[k] < 1 || ci[k] > N || cj[k] < 1 || cj[k] > N) { cols[k] = -1; oob = 1; }
        else cols[k] = (int64_t)(ci[k] - 1) + (int64_t)(cj[k] - 1) * N;        /* 0-based */
    }
    return oob;
}

static void mat4_mul(const double A[16], const double B[16], double C[16])
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s = s + A[i * 4 + k] * B[k * 4 + j];
            C[i * 4 + j] = s;
        }
}

/* inv(rho): Gauss-Jordan with partial pivoting (the reference calls LAPACK getrf/getri) */
static void mat4_inv(const double A[16], double Inv[16])
{
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = A[i * 4 + j]; a[i][4 + j] = (i == j); }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r) if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
        if (piv != col) for (int j = 0; j < 8; ++j) { double t = a[col][j]; a[col][j] = a[piv][j]; a[piv][j] = t; }
        double d = a[col][col];
        for (int j = 0; j < 8; ++j) a[col][j] = a[col][j] / d;
        for (int r = 0; r < 4; ++r) if (r != col) {
            double f = a[r][col];
            for (int j = 0; j < 8; ++j) a[r][j] = a[r][j] - f * a[col][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Inv[i * 4 + j] = a[i][4 + j];
}

/* Phi += inv(rho) * dV(X) * w   (w = dt/2 or T-t)                 NonObstacle.jl:158-159,174-175,178-179 */
static int accumulate(double *Phi /* 4 x N^2, column-major (4 rows fastest) */, const double rho[16],
                      const double X[4], const double *eval, const double *grad, const double *p, int N,
                      int arith, double w, int halve)
{
    int rows[24]; int64_t cols[24]; double vals[24], inv[16];
    int oob = dV(X, eval, grad, p, N, arith, rows, cols, vals);
    mat4_inv(rho, inv);
    for (int k = 0; k < 24; ++k) {
        if (cols[k] < 0) continue;
        for (int i = 0; i < 4; ++i) {
            double th = inv[i * 4 + rows[k]] * vals[k] * w;
            if (halve) th = th / 2;
            Phi[i + 4 * cols[k]] += th;
        }
    }
    return oob;
}

/* ScatterForwardOperator, complete (NonObstacle.jl:141-187, 189-243): for each ray the traced end
 * state X (num,4), the final Jacobian rho (num,16 row-major) and the 4 x N^2 block rho*Phi of M,
 * written DENSE into M_dense ((4*num) x N^2, column-major, leading dimension 4*num) like the
 * reference's SharedArray M before sparse()/droptol!.  Meant for small N / few rays (tests).   */
void orc_forward_operator(const double *eval, const double *grad, const double *hess, const double *p, int N,
                          double timeStep, int arith, const double *m_in, int64_t num, int64_t ld,
                          double *X_out, int64_t ldx, double *rho_out, double *M_dense,
                          int32_t *status, int32_t *nsteps, int nthreads)
{
    (void)nthreads;
    const double cease_r = 1 + 2 * (p[1] - p[0]);
    const int64_t NN = (int64_t)N * N;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
    for (int64_t i = 0; i < num; ++i) {
        double X[4], rho[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        double *Phi = (double *)calloc((size_t)(4 * NN), sizeof(double));
        for (int k = 0; k < 4; ++k) X[k] = m_in[i + k * ld];
        const double Ti = m_in[i + 8 * ld];
        double t = 0, dt = timeStep;
        int st = 0, steps = 0;
        while (t < Ti) {
            if (steps >= ORC_FIXED_STEP_CAP) { st |= ORC_ST_STEP_CAP; break; }
            if ((t + dt) > Ti) dt = Ti - t;
            t = t + dt;
            if (accumulate(Phi, rho, X, eval, grad, p, N, arith, dt, 1)) st |= ORC_ST_OOB;      /* :158-159 */
            double k1[4], k2[4], k3[4], k4[4], t1[16], t2[16], t3[16], t4[16], Y[4];
            double v1[16], v2[16], v3[16], v4[16], R[16];
            int o = 0;
            /* the X stages follow the chosen flavour exactly as orc_trace_fixed_time does */
            o |= hamilton_jac(X, eval, grad, hess, p, N, arith, k1, t1);
            for (int k = 0; k < 4; ++k) Y[k] = arith ? fma(k1[k], 0.5 * dt, X[k]) : X[k] + k1[k] / 2 * dt;
            o |= hamilton_jac(Y, eval, grad, hess, p, N, arith, k2, t2);
            for (int k = 0; k < 4; ++k) Y[k] = arith ? fma(k2[k], 0.5 * dt, X[k]) : X[k] + k2[k] / 2 * dt;
            o |= hamilton_jac(Y, eval, grad, hess, p, N, arith, k3, t3);
            for (int k = 0; k < 4; ++k) Y[k] = arith ? fma(k3[k], dt, X[k]) : X[k] + k3[k] * dt;
            o |= hamilton_jac(Y, eval, grad, hess, p, N, arith, k4, t4);
            if (o) st |= ORC_ST_OOB;
            mat4_mul(t1, rho, v1);                                                           /* :166 */
            for (int k = 0; k < 16; ++k) R[k] = rho[k] + v1[k] * dt / 2;
            mat4_mul(t2, R, v2);                                                             /* :167 */
            for (int k = 0; k < 16; ++k) R[k] = rho[k] + v2[k] * dt / 2;
            mat4_mul(t3, R, v3);                                                             /* :168 */
            for (int k = 0; k < 16; ++k) R[k] = rho[k] + v3[k] * dt;
            mat4_mul(t4, R, v4);                                                             /* :169 */
            if (arith) {
                const double dt6 = dt * 0.16666666666666666;
                for (int k = 0; k < 4; ++k) {
                    double acc = fma(2.0, k3[k], fma(2.0, k2[k], k1[k]));
                    X[k] = fma(acc + k4[k], dt6, X[k]);
                }
            } else {
                for (int k = 0; k < 4; ++k) X[k] = X[k] + (((k1[k] + 2 * k2[k]) + 2 * k3[k]) + k4[k]) * dt / 6.0;   /* :171 */
            }
            for (int k = 0; k < 16; ++k) rho[k] = rho[k] + (((v1[k] + 2 * v2[k]) + 2 * v3[k]) + v4[k]) * dt / 6.0;   /* :172 */
            ++steps;
            if (accumulate(Phi, rho, X, eval, grad, p, N, arith, dt, 1)) st |= ORC_ST_OOB;      /* :174-175 */
            if (norm2(X[0], X[1]) > cease_r) {                                                /* :177-181 */
                if (accumulate(Phi, rho, X, eval, grad, p, N, arith, Ti - t, 0)) st |= ORC_ST_OOB;
                t = t + (Ti - t);
                st |= ORC_ST_CEASED;
            }
        }
        for (int k = 0; k < 4; ++k) X_out[i + k * ldx] = X[k];
        if (rho_out) for (int k = 0; k < 16; ++k) rho_out[i * 16 + k] = rho[k];
        if (M_dense)                                                                          /* :184 */
            for (int64_t col = 0; col < NN; ++col)
                for (int r = 0; r < 4; ++r) {
                    double s = 0;
                    for (int k = 0; k < 4; ++k) s = s + rho[r * 4 + k] * Phi[k + 4 * col];
                    M_dense[(4 * i + r) + col * (4 * num)] = s;
                }
        free(Phi);
        if (status) status[i] = st;
        if (nsteps) nsteps[i] = steps;
    }
}
