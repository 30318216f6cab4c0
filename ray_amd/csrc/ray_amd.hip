// ray_amd.hip -- kernels + C ABI of libray_amd.so (see include/ray_amd.h).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared   (ray_amd/csrc/Makefile)
#include "../../include/ray_amd.h"
#include "ray_device.h"
#include "ray_frechet.h"
#include "ray_host_math.h"
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <thread>
#include <mutex>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

using namespace ray;

// =============================================================================================
// kernels
// =============================================================================================

// a4 + a5: Q4 and the eval/grad table build (NonObstacle.jl:103-105, 191-210), one thread per
// cell, written as one 96-byte record (ray_device.h).  True IEEE divisions, reference order.
__device__ __forceinline__ void q4(const double *c, int N, int i, int j, double dx, double q[4])
{
    const double c00 = c[(size_t)i + (size_t)j * N], c10 = c[(size_t)i + 1 + (size_t)j * N];
    const double c01 = c[(size_t)i + (size_t)(j + 1) * N], c11 = c[(size_t)i + 1 + (size_t)(j + 1) * N];
    q[0] = c00;
    q[1] = (c10 - c00) / dx;
    q[2] = (c01 - c00) / dx;
    q[3] = (((c11 + c00) - c10) - c01) / (dx * dx);
}

// recf: the same record with coefficient k of every quadruple scaled by [1, h, h, h*h] (h = p[2] - p[1], the
// cell size the kFused flavour's cell coordinate is built on; one rounding per coefficient).
__global__ __launch_bounds__(256) void k_build_tables(const double *__restrict__ c, int N, double dx, double h,
                                                       double *__restrict__ rec, double *__restrict__ recf)
{
    const int n1 = N - 1;
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t)n1 * n1) return;
    const int i = (int)(id % n1), j = (int)(id / n1);          // cell (I,J) = (i+1, j+1)
    double out[12];
    q4(c, N, i, j, dx, out);
#pragma unroll
    for (int k = 4; k < 12; ++k) out[k] = 0.0;                   // never written in the reference
    if (i >= 1 && i <= N - 3 && j >= 1 && j <= N - 3) {          // I,J in 2..N-2
        double a[4], b[4];
        const double two_dx = 2 * dx;
        q4(c, N, i + 1, j, dx, a);
        q4(c, N, i - 1, j, dx, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[4 + k] = (a[k] - b[k]) / two_dx;
        q4(c, N, i, j + 1, dx, a);
        q4(c, N, i, j - 1, dx, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[8 + k] = (a[k] - b[k]) / two_dx;
    }
    double2 *r = reinterpret_cast<double2 *>(rec) + (size_t)id * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) r[k] = make_double2(out[2 * k], out[2 * k + 1]);
    const double h2 = h * h;
    double2 *rf = reinterpret_cast<double2 *>(recf) + (size_t)id * 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rf[2 * k] = make_double2(out[4 * k], out[4 * k + 1] * h);
        rf[2 * k + 1] = make_double2(out[4 * k + 2] * h, out[4 * k + 3] * h2);
    }
}

#ifdef RAY_TIMELINE
// instrumentation build only (scripts/timeline.py): start / end clock and hardware id of every workgroup of the
// last k_shoot launch, to see how full the chip is over the launch
__device__ unsigned long long g_ray_timeline[3 * 65536];
#endif

#ifndef RAY_ARC_ROT
#define RAY_ARC_ROT(rank) (rank)
#endif

struct ShootArgs {
    MediumDev med;
    int ob_kind;
    double ob_r0, ob_rho, ob_rho_n;   // circle: r0; trefoil: R, rho of the level set, rho of the normal
    double ob_far2;                   // x^2 + y^2 above this: outside the obstacle for certain (ob > 0 without evaluating it)
    double ob_in2;                    // circle: the largest s = x^2 + y^2 with fl(sqrt(s)) <= r0, i.e. ob <= 0  <=>  s <= ob_in2
    double ob_lo2, ob_hi2;            // circle: |fl(sqrt(s)) - r0| <= 1e-15  <=>  ob_lo2 <= s <= ob_hi2 (the reflection bisection's stop)
    int anchor_k[kRadialMax];         // kFused analytic media: steps one full sin/cos evaluation anchors, per argument
    // init == 1: generate the boundary phase-space grid (NonObstacle.jl:55-65)
    int init;
    int nd;
    int nsrc;               // sources in this call (src_end - src_begin)
    int ngroup;             // source groups per direction chunk = ceil(nsrc / (waves per block * sources per wave))
    int dshift;             // log2 of the directions per wave: a wave is 2^(6-dshift) sources x 2^dshift directions
    const double *src;      // source[i]            (local index i - src_begin)
    const double *src_cos;  // cos(source[i])  (host, ray_trig.h)
    const double *src_sin;  // sin(source[i])
    const double *direct;   // direct[j], j = 0..nd-1
    double half_pi;         // Float64(pi) * 0.5
    double dt;
    int time_unit;
    int max_steps;
    double *m;
    int64_t num, ld;
    int32_t *status, *nsteps, *nreflect;
    unsigned long long *step_counter;
    // PATH kernels only: the polyline of every ray ([ray][point][x,y], at most max_pts points kept) as the
    // reference's *Plot tracers collect it (Obstacle.jl:114,142,175,211)
    double *path;
    int32_t *npts;
    int max_pts;
};

// obstacle level set and normal (ObstacleDemo.jl:20-34): the shipped circle  r - r0,  n = (x,y)/r, and the
// commented trefoil alternate  r - (R - rho cos(3 theta))  (:24) with
// n = [x,y]/r - 3 rho' sin(3 theta)/r * [-y,x]/r  (:31; rho' is a constant of its own there), n/norm(n) (:33).
// theta = atan2(y, x) enters only through cos 3theta and sin 3theta, taken here from cos theta = x/r and
// sin theta = y/r by the triple-angle identities (4c^3 - 3c, 3s - 4s^3): the same numbers to a few ulp as the
// three library calls -- which is also how far two libms differ -- at a tenth of their cost per step.
template <int OB>
__device__ __forceinline__ double ob_val(const ShootArgs &a, double x, double y)
{
    if (OB == kObTrefoil) {
        const double r = norm2(x, y);
        const double c = r > 0 ? x / r : 1.0;                  // atan2(0, 0) = 0
        const double cos3 = c * (4 * (c * c) - 3);
        return r - (a.ob_r0 - a.ob_rho * cos3);
    }
    return norm2(x, y) - a.ob_r0;
}
template <int OB>
__device__ __forceinline__ void ob_normal(const ShootArgs &a, double x, double y, double &nx, double &ny)
{
    const double r = norm2(x, y);
    nx = x / r; ny = y / r;
    if (OB == kObTrefoil) {
        const double sn = ny;                                  // sin(theta) = y/r
        const double sin3 = sn * (3 - 4 * (sn * sn));
        const double s = ((3 * a.ob_rho_n) * sin3) / r;
        nx = nx - (s * (-y)) / r;
        ny = ny - (s * x) / r;
    }
    const double nn = norm2(nx, ny);
    nx = nx / nn; ny = ny / nn;
}

__device__ __forceinline__ void lerp4(double E[4], const double P[4], const double X[4], double mid)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) E[k] = (1 - mid) * P[k] + mid * X[k];
}

// a6 / a7: ScatterRelation / ScatterRelationObstacle per-ray loop.
// NonObstacle.jl:67-97; Obstacle.jl:23-95.  Lanes = consecutive directions of one source, so a
// wavefront is a coherent fan: similar step counts (chord ~ 2 sin(phi)) and neighbouring cells.
//  - the RK4 step is executed by ALL 64 lanes while any lane of the wavefront is alive: the
//    gridded right-hand side fetches cell records wave-cooperatively (ray_device.h) and needs the
//    wave converged.  Only the lanes that were live at the start of a step take its result (four
//    exec-masked moves): a finished lane's state stays in place bit for bit, whatever its stages
//    evaluated to (garbage states that overflow included);
//  - all lanes of a wave start together, so elapsed time and step count are wave-uniform until a
//    lane leaves: they are kept once per wave and latched by each lane at its exit;
//  - the exit bisection (up to ~55 dependent iterations) is deferred until every lane of the
//    wavefront has left the disk and is then run by all lanes at once; the pre-exit state is
//    parked in the output row meanwhile.
#ifndef RAY_ANALYTIC_WAVES
#define RAY_ANALYTIC_WAVES 4            // sin-product medium: pure VALU, dependent fp64 chains: four waves per SIMD (<= 128 VGPRs)
#endif
#ifndef RAY_SHOOT_WAVES
#define RAY_SHOOT_WAVES 5
#endif
template <int MED, int OB, int AR, int PATH = 0>
__global__ __launch_bounds__(256, (MED == kGrid && (OB == kObNone || OB == kObCircle) && AR == kFused && !PATH) ? RAY_SHOOT_WAVES : (MED == kSinProd && !PATH) ? RAY_ANALYTIC_WAVES : 1) void k_shoot(const ShootArgs a)
{
    // Work mapping.  Explicit states: row = global thread id.  Generated boundary grid: a wave is
    // 64 consecutive directions ("chunk") of one source, a workgroup is the same chunk of
    // `waves per block` adjacent sources (equal ray lengths -> the waves retire together; adjacent
    // fans -> shared cache lines).  Chunks are dispatched longest-first -- chord length ~ 2 sin(phi)
    // peaks at the middle chunk -- which is the GPU form of the reference's sort-by-traveltime
    // balancer (NonObstacle.jl:223-233): the short grazing rays fill the tail of the launch.
    // Within a chunk, groups are dealt so that each XCD (blockIdx % 8) sweeps a contiguous arc of
    // sources and keeps that arc's cells in its own L2.
    int64_t r;
    bool valid;
    int64_t il = 0;
    int j = 0;
    if (a.init) {
        const int dmask = (1 << a.dshift) - 1;
        const int nchunk = (a.nd + dmask) >> a.dshift;
        const int rank = (int)(blockIdx.x / (unsigned)a.ngroup);
        int gi = (int)(blockIdx.x - (unsigned)rank * (unsigned)a.ngroup);
        // an XCD takes every 8th workgroup of the launch and nothing moves between XCDs afterwards: the arc it sweeps
        // rotates from chunk to chunk, so that every XCD sees every part of the medium (arcs differ in ray length
        // by 10-20 % in a non-symmetric medium; a fixed arc per XCD left half the chip idle for the last 15 %
        // of the launch -- profiles/r2/timeline.md)
        if ((a.ngroup & 7) == 0) gi = (((gi & 7) + RAY_ARC_ROT(rank)) & 7) * (a.ngroup >> 3) + (gi >> 3);
        const int mid = nchunk >> 1, hh = (rank + 1) >> 1;
        const int chunk = (rank & 1) ? mid - hh : mid + hh;
        const int lane = (int)(threadIdx.x & 63);
        il = (((int64_t)gi * (blockDim.x >> 6) + (threadIdx.x >> 6)) << (6 - a.dshift)) + (lane >> a.dshift);
        j = (chunk << a.dshift) + (lane & dmask);
        valid = il < a.nsrc && j < a.nd;
        r = il * a.nd + j;
    } else {
        r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        valid = r < a.num;
    }
#ifdef RAY_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < 65536) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_ray_timeline[3 * blockIdx.x] = wall_clock64();
        g_ray_timeline[3 * blockIdx.x + 2] = hw;
    }
#endif
    GridAcc acc{};
    if (MED == kGrid) acc = grid_acc_setup<AR == kExact>(a.med.g);
    double X[4] = {0, 0, 1, 0};                             // harmless state for padding lanes
    int st = 0, oob = 0;
    if (valid) {
        if (a.init) {
            const double sx = a.src_cos[il], sy = a.src_sin[il];
            const double theta = a.src[il] + a.half_pi + a.direct[j];
            double sn, cs;
            sincos_rep(theta, sn, cs);
            const double c0 = (MED == kGrid) ? speed_grid(a.med.g, AR == kExact ? lds_nodes() : a.med.g.p, sx, sy, oob)
                            : (MED == kRadial) ? speed_radial(a.med.rd, sx, sy)
                                             : speed_sinprod(a.med.sp, sx, sy);
            X[0] = sx; X[1] = sy; X[2] = cs / c0; X[3] = sn / c0;
#pragma unroll
            for (int k = 0; k < 4; ++k) a.m[r + k * a.ld] = X[k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) X[k] = a.m[r + k * a.ld];
        }
        if (!(X[0] == X[0]) || !(X[1] == X[1])) oob = 1;    // NaN position: see cell_of()
    }
    int np = 0;                                             // PATH: points recorded so far
    auto put = [&](double x, double y) {
        if (np < a.max_pts) {
            double2 *dst = reinterpret_cast<double2 *>(a.path) + (r * a.max_pts + np);
            *dst = make_double2(x, y);
        }
        ++np;
    };
    if (PATH && valid) put(X[0], X[1]);                     // res = X[1:2]           Obstacle.jl:114
    // The pre-step state (NonObstacle.jl:71 `prev`) of a lane at the step it leaves the disk is parked in
    // its output row (columns 5..8) until the deferred bisection picks it up: holding it in registers
    // would cost a wave per SIMD, and LDS is spent on record slots.
    const double dt = a.dt;
    double tw = 0, t = 0;          // tw: wave-uniform clock; t: this lane's, latched at its exit
    int it = 0, steps = 0;         // it: wave-uniform step count
    int nref = 0, rfl = 0;
    bool active = valid;
    // one back edge, one exit: wave-uniform condition; live lanes left at the cap get STEP_CAP below.
    // lm = the live lanes, balloted once per step.  Every lane computes the step (the fetch needs the wave
    // converged, and dt stays a scalar), but only the lanes that were live at its start TAKE it: a finished
    // or padding lane keeps its latched state untouched, whatever its right-hand side evaluates to.
    // The raw cell indices are max-ed into `o` over ALL steps and judged when the lane leaves (or at the
    // cap): what a finished lane adds to its own `o` afterwards is never looked at.  A NaN position never
    // leaves and is judged at the cap.
    unsigned long long lm = __builtin_amdgcn_ballot_w64(active);
    OobMax o = 0;
    RecState rs;
    TrigBase tb;                                            // analytic media: the sin/cos anchors (ray_device.h)
    int acnt[kRadialMax] = {0, 0, 0, 0};                    // wave-uniform countdowns to the next full evaluation
    // ob(X) at the top of a step (Obstacle.jl:28) is ob of the point the previous step tested (:69), or of
    // the reflection point R whose value ended the bisection (:79): evaluated once, carried
    double obv = (OB != kObNone) ? ob_val<OB>(a, X[0], X[1]) : 0.0;
    if (lm != 0 && a.max_steps > 0) do {
        const bool was = active;
        if (OB != kObNone) { if (active && obv > 0) rfl = 0; }                             // Obstacle.jl:28-30
        double Xn[4];
        unsigned anchor = 0;
        if (MED != kGrid && AR == kFused) {
#pragma unroll
            for (int k = 0; k < (MED == kSinProd ? 2 : kRadialMax); ++k) {
                if (acnt[k] == 0) { anchor |= 1u << k; acnt[k] = a.anchor_k[k]; }
                --acnt[k];
            }
        }
        rk4<MED, AR, 0>(a.med, acc, X, Xn, dt, lm, o, rs, tb, anchor);
        ++it;
        tw = a.time_unit ? tw + 1 : tw + dt;                // Obstacle.jl:39 / NonObstacle.jl:77
        // norm(X[1:2]) >= 1  <=>  x^2+y^2 >= 1 for a correctly rounded sqrt  (NonObstacle.jl:78)
        const double s = Xn[0] * Xn[0] + Xn[1] * Xn[1];
        if (active) {
            if (s >= 1.0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) a.m[r + (4 + k) * a.ld] = X[k];   // the pre-step state (`prev`)
                t = tw; steps = it;
                st |= kStExited; active = false;
                if (MED == kGrid) oob |= oob_hit(a.med.g, o);
            } else if (OB != kObNone) {
                // only the sign of ob is ever used here (:28, :69): beyond the trefoil's largest radius it is known
                // without the square root and the division (a wave of a coherent fan is mostly all inside or all
                // outside: -5.3 % kernel time; for the circle the test costs more than the square root it saves)
                // The circle's ob = fl(sqrt(s)) - r0 has the sign of s - ob_in2 exactly (the rounded root is monotone in s;
                // the host derives the threshold): no square root per step.
                if (OB == kObTrefoil && s > a.ob_far2) obv = 1.0;
                else if (OB == kObCircle) obv = s > a.ob_in2 ? 1.0 : 0.0;
                else obv = ob_val<OB>(a, Xn[0], Xn[1]);
                if (obv <= 0 && rfl == 0) {                                      // Obstacle.jl:69-91
                    double lo = 0, hi = 1, mid = 0.5, E[4], e;
                    int ib = 0;
                    rfl = 1;
                    if (OB == kObCircle) {
                        // e = fl(sqrt(sq)) - r0 (an exact difference next to r0): `abs(e) > 1e-15` and `e > 0` (Obstacle.jl:
                        // 76-79) are comparisons of sq = E1^2 + E2^2 with three doubles the host derived from r0 -- the rounded
                        // root is monotone in sq -- so the ~50 dependent iterations need no square root (as the exit bisection)
                        E[0] = (1 - mid) * X[0] + mid * Xn[0];
                        E[1] = (1 - mid) * X[1] + mid * Xn[1];
                        double sq = E[0] * E[0] + E[1] * E[1];
                        while (sq < a.ob_lo2 || sq > a.ob_hi2) {
                            if (++ib > kBisectCap) { st |= kStBisectCap; break; }
                            if (sq > a.ob_in2) lo = mid; else hi = mid;
                            mid = (lo + hi) * 0.5;
                            E[0] = (1 - mid) * X[0] + mid * Xn[0];
                            E[1] = (1 - mid) * X[1] + mid * Xn[1];
                            sq = E[0] * E[0] + E[1] * E[1];
                        }
                        E[2] = (1 - mid) * X[2] + mid * Xn[2];
                        E[3] = (1 - mid) * X[3] + mid * Xn[3];
                        e = sq > a.ob_in2 ? 1.0 : 0.0;           // only the sign of ob(R) is looked at again (the re-arming test)
                    } else {
                        lerp4(E, X, Xn, mid);
                        e = ob_val<OB>(a, E[0], E[1]);
                        while (__builtin_fabs(e) > 1e-15) {
                            if (++ib > kBisectCap) { st |= kStBisectCap; break; }
                            if (e > 0) lo = mid; else hi = mid;
                            mid = (lo + hi) * 0.5;
                            lerp4(E, X, Xn, mid);
                            e = ob_val<OB>(a, E[0], E[1]);
                        }
                    }
                    // n = gradObstacle(R); X[3:4] = (eye(2) - 2*n*n') * R[3:4]   Obstacle.jl:84-85
                    double nx, ny;
                    ob_normal<OB>(a, E[0], E[1], nx, ny);
                    const double a0 = 2 * nx, a1 = 2 * ny;
                    const double m11 = 1 - a0 * nx, m12 = 0 - a0 * ny;
                    const double m21 = 0 - a1 * nx, m22 = 1 - a1 * ny;
                    Xn[0] = E[0]; Xn[1] = E[1];
                    obv = e;                                 // = ob(R): the next step's re-arming test
                    Xn[2] = m11 * E[2] + m12 * E[3];
                    Xn[3] = m21 * E[2] + m22 * E[3];
                    ++nref;
                    if (PATH) put(E[0], E[1]);              // res = [res R[1:2]]      Obstacle.jl:175
                }
            }
            if (PATH && active) put(Xn[0], Xn[1]);          // res = [res X[1:2]]      Obstacle.jl:211
        }
        if (was) {                                          // exec-masked: frozen lanes keep their state bit for bit
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("v_mov_b64 %0, %1" : "+v"(X[k]) : "v"(Xn[k]));
        }
        lm = __builtin_amdgcn_ballot_w64(active);
    } while (lm != 0 && it < a.max_steps);
    if (active) {
        st |= kStStepCap; t = tw; steps = it;
        oob |= (MED == kGrid ? oob_hit(a.med.g, o) : 0) | !(X[0] == X[0]) | !(X[1] == X[1]);
    }
    if (st & kStExited) {                                       // NonObstacle.jl:82-91
        double lo = 0, hi = 1, mid = 0.5, E[4], P[4];
        int ib = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) P[k] = a.m[r + (4 + k) * a.ld];
        // e = norm(E[1:2]) - 1, `abs(e) > 1e-15`, `e > 0` (NonObstacle.jl:84-88) decided on
        // s = E1^2 + E2^2 without the square root: fl(sqrt(s)) is monotone in s, so each test is a
        // comparison of s with the double where the rounded root crosses the threshold
        // (tests/test_oracle.py::test_exit_bisection_thresholds derives and checks the three constants)
        constexpr double kSL = 0x1.fffffffffffeep-1;     // smallest s with sqrt(s) - 1 >= -1e-15
        constexpr double kSH = 0x1.0000000000009p+0;     // largest  s with sqrt(s) - 1 <=  1e-15
        constexpr double kS1 = 0x1.0000000000001p+0;     // largest  s with sqrt(s) <= 1
        double sq;
        E[0] = (1 - mid) * P[0] + mid * X[0];
        E[1] = (1 - mid) * P[1] + mid * X[1];
        sq = E[0] * E[0] + E[1] * E[1];
        while (sq < kSL || sq > kSH) {
            if (++ib > kBisectCap) { st |= kStBisectCap; break; }
            if (sq > kS1) hi = mid; else lo = mid;
            mid = (lo + hi) * 0.5;
            E[0] = (1 - mid) * P[0] + mid * X[0];
            E[1] = (1 - mid) * P[1] + mid * X[1];
            sq = E[0] * E[0] + E[1] * E[1];
        }
        E[2] = (1 - mid) * P[2] + mid * X[2];
        E[3] = (1 - mid) * P[3] + mid * X[3];
        t = a.time_unit ? t - (1 - mid) : t - (1 - mid) * dt;
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = E[k];
        if (PATH) put(E[0], E[1]);                          // res = [res E[1:2]]      Obstacle.jl:142
    }
    if (oob) st |= kStOob;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a.m[r + (4 + k) * a.ld] = X[k];
        a.m[r + 8 * a.ld] = t;
        if (a.status) a.status[r] = st;
        if (a.nsteps) a.nsteps[r] = steps;
        if (a.nreflect) a.nreflect[r] = nref;
        if (PATH && a.npts) a.npts[r] = np;
    }
    const unsigned wsum = wave_sum((unsigned)steps);
    if ((threadIdx.x & 63) == 0 && a.step_counter) atomicAdd(a.step_counter, (unsigned long long)wsum);
#ifdef RAY_TIMELINE
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 65536) g_ray_timeline[3 * blockIdx.x + 1] = wall_clock64();
#endif
}

struct TraceArgs {
    MediumDev med;
    const double *m;
    int64_t num, ld;
    double dt;
    int mode;
    double cease_r;       // 1 + 2*(p[2]-p[1])                      NonObstacle.jl:177
    double cease_s2;      // the largest s = x^2 + y^2 with fl(sqrt(s)) <= cease_r
    int max_steps;
    double *X;
    int64_t ldx;
    int32_t *status, *nsteps;
    unsigned long long *step_counter;
    double *path;         // PATH variant: [ray][point][x,y] polylines (NonReflectionPlot's `res`), else null
    int32_t *npts;
    int max_pts;
    // RAY_OPT_FAN_DIRECTIONS: the rows are nsrc fans of nd consecutive directions (row i*nd + j, as ScatterRelation
    // lays m out).  nd > 0 selects k_shoot's work mapping: a wave is a patch of 2^(6-dshift) adjacent sources x
    // 2^dshift directions, a workgroup the same direction chunk of adjacent source patches, chunks dispatched from the
    // middle of the fan outwards (longest chords first), the XCDs' source arcs rotating from chunk to chunk.
    int nd, nsrc, ngroup, dshift;
};

// a8 (mode OVERSHOOT, Obstacle.jl:306-326) and a9 X-trajectory (mode SHORTEN_CEASE,
// NonObstacle.jl:146-157,161-164,171,177-181,185).  Same wave-converged stepping as k_shoot;
// the clock is per lane here (every ray has its own T and its own shortened last step).
// PATH = 1 also records `res` of NonReflectionPlot (Obstacle.jl:256,266): the start and the position after
// every step.
#ifndef RAY_TRACE_WAVES
#define RAY_TRACE_WAVES 5               // waves per SIMD of the fused trace kernel (96 VGPRs, 5 spilled; as k_shoot: 31.7 KB of LDS per workgroup)
#endif
template <int AR, int PATH = 0>
__global__ __launch_bounds__(256, (AR == kFused && !PATH) ? RAY_TRACE_WAVES : 1) void k_trace(const TraceArgs a)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = r < a.num;
    if (a.nd > 0) {                                          // fan-aware mapping, as in k_shoot
        const int dmask = (1 << a.dshift) - 1;
        const int nchunk = (a.nd + dmask) >> a.dshift;
        const int rank = (int)(blockIdx.x / (unsigned)a.ngroup);
        int gi = (int)(blockIdx.x - (unsigned)rank * (unsigned)a.ngroup);
        if ((a.ngroup & 7) == 0) gi = (((gi & 7) + RAY_ARC_ROT(rank)) & 7) * (a.ngroup >> 3) + (gi >> 3);
        const int mid = nchunk >> 1, hh = (rank + 1) >> 1;
        const int chunk = (rank & 1) ? mid - hh : mid + hh;
        const int lane = (int)(threadIdx.x & 63);
        const int64_t il = (((int64_t)gi * (blockDim.x >> 6) + (threadIdx.x >> 6)) << (6 - a.dshift)) + (lane >> a.dshift);
        const int j = (chunk << a.dshift) + (lane & dmask);
        valid = il < a.nsrc && j < a.nd;
        r = valid ? il * a.nd + j : 0;
    }
    const GridAcc acc = grid_acc_setup<AR == kExact>(a.med.g);
    double X[4] = {0, 0, 1, 0};
    double T = 0;
    int steps = 0, st = 0, oob = 0, it = 0;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = a.m[r + k * a.ld];
        T = a.m[r + 8 * a.ld];
        if (!(X[0] == X[0]) || !(X[1] == X[1])) oob = 1;
    }
    double t = 0, dt = a.dt;
    bool active = valid && t < T;
    RecState rs;
    int np = 0;
    auto put = [&](double x, double y) {
        if (np < a.max_pts) reinterpret_cast<double2 *>(a.path)[r * a.max_pts + np] = make_double2(x, y);
        ++np;
    };
    if (PATH && valid) put(X[0], X[1]);                     // res = X[1:2]            Obstacle.jl:256
    // one back edge, one exit (wave-uniform); live lanes left at the cap get STEP_CAP below
    if (__builtin_amdgcn_ballot_w64(active) != 0 && a.max_steps > 0) do {
        double dtl = 0.0;
        if (active) {
            if (a.mode == RAY_TRACE_SHORTEN_CEASE && (t + dt) > T) dt = T - t;    // NonObstacle.jl:153-156
            t = t + dt;
            dtl = dt;
        }
        double Xn[4];
        OobMax o = 0;
        const bool was = active;
        TrigBase tb;                                         // (analytic media only: unused on the grid)
        rk4<kGrid, AR, 1>(a.med, acc, X, Xn, dtl, __builtin_amdgcn_ballot_w64(active), o, rs, tb, 0u);
        ++it;
        const double s = Xn[0] * Xn[0] + Xn[1] * Xn[1];
        if (active) {
            steps = it;
            oob |= oob_hit(a.med.g, o) | !(s == s);
            if (a.mode == RAY_TRACE_SHORTEN_CEASE) {
                // norm(X[1:2]) > 1 + 2h  <=>  s > cease_s2, the largest s whose rounded root is still <= 1 + 2h (the host
                // derives it: the rounded root is monotone in s) -- no square root per step.  (A "thin shell" form that
                // took the root only near the threshold was if-converted by the compiler into a root at every step.)
                if (s > a.cease_s2) { t = t + (T - t); st |= kStCeased; }       // NonObstacle.jl:177-181
            }
            if (PATH) put(Xn[0], Xn[1]);                    // res = [res X[1:2]]      Obstacle.jl:266
            active = t < T;
        }
        if (was) {                                          // exec-masked: a finished lane keeps its end state bit for bit
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("v_mov_b64 %0, %1" : "+v"(X[k]) : "v"(Xn[k]));
        }
    } while (__builtin_amdgcn_ballot_w64(active) != 0 && it < a.max_steps);
    if (PATH && valid && a.npts) a.npts[r] = np;
    if (active) st |= kStStepCap;
    if (oob) st |= kStOob;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a.X[r + k * a.ldx] = X[k];
        if (a.status) a.status[r] = st;
        if (a.nsteps) a.nsteps[r] = steps;
    }
    const unsigned wsum = wave_sum((unsigned)steps);
    if ((threadIdx.x & 63) == 0 && a.step_counter) atomicAdd(a.step_counter, (unsigned long long)wsum);
}

// ---- NEXT-3: ray classification (Utility.jl:1-39; ObstacleDemo.jl:48-67) ------------------------------
// similarity(X, Y) = cos(atan2(X[1], X[2]) - atan2(Y[1], Y[2]))                      Utility.jl:1-3
__device__ __forceinline__ double similarity(double x1, double x2, double y1, double y2)
{
    return cos(atan2(x1, x2) - atan2(y1, y2));
}

// orthoIndex filter (ObstacleDemo.jl:48-49) per ray, and arg = atan2(m[:,6], m[:,5]) (:59-60)
__global__ __launch_bounds__(256) void k_classify_rays(const double *__restrict__ m, int64_t num, int64_t ld,
                                                       uint8_t *__restrict__ ortho, double *__restrict__ arg)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= num) return;
    const double x0 = m[r], y0 = m[r + ld], p0 = m[r + 2 * ld], q0 = m[r + 3 * ld];
    const double x1 = m[r + 4 * ld], y1 = m[r + 5 * ld], p1 = m[r + 6 * ld], q1 = m[r + 7 * ld];
    if (ortho) ortho[r] = (similarity(x0, y0, x1, y1) > 0.995 && similarity(p0, q0, p1, q1) < -0.995) ? 1 : 0;
    arg[r] = atan2(y1, x1);
}

// alignment + derivativeCheck (Utility.jl:5-39) of one source's exit-angle signal, in place; one thread
// per source, sequential like the reference (every later value depends on the earlier corrections)
__global__ __launch_bounds__(64) void k_unbroken(double *__restrict__ arg, int64_t ns, int64_t nd,
                                                 int32_t *__restrict__ lo_out, int32_t *__restrict__ hi_out)
{
    const int64_t sidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= ns) return;
    double *sig = arg + sidx * nd;
    const int64_t n = nd;
    const double two_pi = 2 * 3.141592653589793, lim = 1.5 * 3.141592653589793;
    // signal[i:n] -= sign(jump) * 2pi at every jump: a running shift applied as the scan passes; the jump
    // test sees both neighbours shifted by the same corrections, exactly as in the in-place original
    double prev = n > 0 ? sig[0] : 0.0;
    for (int64_t i = 1; i < n; ++i) {
        // values right of i carry all corrections made so far: apply them one by one, in order
        double v = sig[i];
        sig[i] = v;
        const double d = v - prev;
        if (__builtin_fabs(d) > lim) {
            const double sh = (d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0)) * two_pi;
            for (int64_t k = i; k < n; ++k) sig[k] = sig[k] - sh;
            v = sig[i];
        }
        prev = v;
    }
    int64_t lo = 1, hi = n;
    for (int64_t i = 1; i <= n - 2; ++i) {                       // deriv[i] = signal[i+1] - signal[i], 1-based
        const double d0 = sig[i] - sig[i - 1], d1 = sig[i + 1] - sig[i];
        if (__builtin_fabs(d0 - d1) > 0.12) { lo = i; break; }
    }
    for (int64_t i = n - 1; i >= 2; --i) {
        const double d0 = sig[i] - sig[i - 1], dm = sig[i - 1] - sig[i - 2];
        if (__builtin_fabs(d0 - dm) > 0.12) { hi = i + 1; break; }
    }
    const int64_t gap = lo > hi ? lo - hi : hi - lo;
    if ((double)gap < (double)n / 8) { lo = 1; hi = n; }
    lo_out[sidx] = (int32_t)lo;
    hi_out[sidx] = (int32_t)hi;
}

// a11: misfit sums, per-ray selection mask, ray-major mismatch vector.
// Stage 1: fixed grid of kMisfitBlocks x 256 threads, grid-stride over rays, LDS tree per block.
// Stage 2: one block folds the kMisfitBlocks partials.  Order is fixed -> reproducible.
constexpr int kMisfitBlocks = 1024;

__device__ __forceinline__ void block_tree_sum2(double &u, double &v, double *lds)
{
    const int tid = threadIdx.x;
    lds[tid] = u;
    lds[256 + tid] = v;
    __syncthreads();
#pragma unroll
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { lds[tid] = lds[tid] + lds[tid + o]; lds[256 + tid] = lds[256 + tid] + lds[256 + tid + o]; }
        __syncthreads();
    }
    u = lds[0];
    v = lds[256];
}

__global__ __launch_bounds__(256) void k_misfit_partial(const double *__restrict__ m,
                                                         const double *__restrict__ obs, int64_t num,
                                                         int64_t ld, double thresh,
                                                         uint8_t *__restrict__ mask,
                                                         double *__restrict__ mismatch,
                                                         double *__restrict__ partial)
{
    __shared__ double lds[512];
    double sm = 0, stg = 0;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < num; r += (int64_t)gridDim.x * 256) {
        double d[4], a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = m[r + (4 + k) * ld];
            d[k] = a[k] - obs[r + (4 + k) * ld];
            sm = sm + d[k] * d[k];
            stg = stg + a[k] * a[k];
        }
        if (mismatch) {
            double2 *o = reinterpret_cast<double2 *>(mismatch + 4 * r);
            o[0] = make_double2(d[0], d[1]);
            o[1] = make_double2(d[2], d[3]);
        }
        if (mask) {
            const double nd_ = __builtin_sqrt(((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3]);
            const double na = __builtin_sqrt(((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]) + a[3] * a[3]);
            mask[r] = (nd_ / na < thresh) ? 1 : 0;
        }
    }
    block_tree_sum2(sm, stg, lds);
    if (threadIdx.x == 0) { partial[blockIdx.x] = sm; partial[kMisfitBlocks + blockIdx.x] = stg; }
}

__global__ __launch_bounds__(256) void k_misfit_final(const double *__restrict__ partial,
                                                       double *__restrict__ out)
{
    __shared__ double lds[512];
    double sm = 0, stg = 0;
    for (int k = threadIdx.x; k < kMisfitBlocks; k += 256) {
        sm = sm + partial[k];
        stg = stg + partial[kMisfitBlocks + k];
    }
    block_tree_sum2(sm, stg, lds);
    if (threadIdx.x == 0) { out[0] = sm; out[1] = stg; }
}

// Scratch of ray_misfit_static_dev (include/ray_amd.h): the 2 x kMisfitBlocks partial sums in a file-scope
// device array instead of a hipMalloc'ed buffer.  Every GPU holds its own copy at its own address; the
// host never uses the array's name as a pointer -- g_static_partial[dev] is resolved once per device with
// hipGetSymbolAddress (ray_ctx_create) and handed to the kernels as an argument.
static __device__ __align__(256) double g_misfit_scratch[2 * kMisfitBlocks];

// =============================================================================================
// host side
// =============================================================================================
static thread_local std::string g_err;
constexpr int kMaxDevices = 64;
static std::atomic<double *> g_static_partial[kMaxDevices];   // per device: address of g_misfit_scratch there (null until resolved)
static std::mutex g_static_mu;                                // serialises the once-per-device resolution in ray_ctx_create

static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (void)hipGetLastError();     /* reported here: do not leave it for the next launch's check */ \
            return fail(RAY_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
        }                                                                                      \
    } while (0)

// ---- host-pointer downloads: page-locked ring + copy workers ------------------------------------------------------
// What a blocking ccall hands over is pageable memory; a hipMemcpy into it goes through the runtime's own small
// staging buffers at a third of the PCIe rate, and hipHostRegister of the caller's array is refused for the sandboxed
// users of some installations (profiles/r2/host_pointer_path.jsonl).  The library therefore owns a ring of page-locked
// slots (hipHostMalloc): a block of result rows is DMA'd into a slot at the full rate on a copy stream of its own
// while the next block's kernel runs, and a few host threads copy the slot into the caller's array while the next
// block's DMA runs.  Threads are created on the first large download of a context and joined when it is destroyed.
struct CopyPiece { void *dst; const void *src; size_t bytes; };
class CopyPool {
public:
    explicit CopyPool(int n)
    {
        for (int k = 0; k < n; ++k) th_.emplace_back([this] { work(); });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> l(mu_); quit_ = true; }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    // copies every piece (cut into slices of at most 1 MiB, dealt to the workers and to the caller); returns when done
    void run(const std::vector<CopyPiece> &pieces)
    {
        std::vector<CopyPiece> cut;
        constexpr size_t kSlice = 1u << 20;
        for (const CopyPiece &p : pieces)
            for (size_t o = 0; o < p.bytes; o += kSlice)
                cut.push_back(CopyPiece{static_cast<char *>(p.dst) + o, static_cast<const char *>(p.src) + o, std::min(kSlice, p.bytes - o)});
        if (cut.empty()) return;
        {
            std::lock_guard<std::mutex> l(mu_);
            job_ = &cut; next_.store(0); left_ = (int)cut.size(); ++gen_;
        }
        cv_.notify_all();
        drain(cut);
        std::unique_lock<std::mutex> l(mu_);
        done_.wait(l, [this] { return left_ == 0; });
        job_ = nullptr;
    }
private:
    void drain(const std::vector<CopyPiece> &cut)
    {
        int mine = 0;
        for (;;) {
            const size_t k = next_.fetch_add(1);
            if (k >= cut.size()) break;
            std::memcpy(cut[k].dst, cut[k].src, cut[k].bytes);
            ++mine;
        }
        if (mine) {
            std::lock_guard<std::mutex> l(mu_);
            left_ -= mine;
            if (left_ == 0) done_.notify_all();
        }
    }
    void work()
    {
        unsigned long seen = 0;
        for (;;) {
            const std::vector<CopyPiece> *job;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [&] { return quit_ || (gen_ != seen && job_); });
                if (quit_) return;
                seen = gen_;
                job = job_;
            }
            drain(*job);
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::vector<CopyPiece> *job_ = nullptr;
    std::atomic<size_t> next_{0};
    int left_ = 0;
    unsigned long gen_ = 0;
    bool quit_ = false;
};

struct HostRing {
    static constexpr int kSlots = 2;
    void *slot[kSlots] = {};
    size_t cap = 0;                            // bytes per slot
    hipStream_t copy_stream = nullptr;
    hipEvent_t filled[kSlots] = {};            // the DMA into the slot has completed
    std::vector<hipEvent_t> blk_done;          // block k's kernel has completed (recorded on the compute stream)
    CopyPool *pool = nullptr;
    bool refused = false;                      // hipHostMalloc failed once on this context: do not ask again
    void release()
    {
        for (int k = 0; k < kSlots; ++k) {
            if (slot[k]) (void)hipHostFree(slot[k]);
            if (filled[k]) (void)hipEventDestroy(filled[k]);
            slot[k] = nullptr; filled[k] = nullptr;
        }
        for (hipEvent_t ev : blk_done) (void)hipEventDestroy(ev);
        blk_done.clear();
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        copy_stream = nullptr; cap = 0;
        delete pool;
        pool = nullptr;
    }
};

struct InitTables;
struct ray_ctx {
    int device = 0;
    InitTables *init = nullptr;                // cached boundary-grid tables
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    unsigned long long *d_steps = nullptr;     // device ray-step counter
    double *d_partial = nullptr;               // misfit partials, 2*kMisfitBlocks
    uint64_t rays = 0, launches = 0;
    int arith = RAY_ARITH_FUSED;
    bool timing = false;
    double kernel_ms = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    // grow-only device workspaces reused across calls (the Frechet stencil slab: a reconstruction loop
    // calls ScatterForwardOperator every iteration with the same sizes)
    static constexpr int kWs = 24;             // 0..11 the Frechet path, 12..23 staging of the host-pointer entry points
    void *ws[kWs] = {};
    size_t ws_cap[kWs] = {};
    HostRing ring;                             // page-locked staging of the host-pointer downloads
    bool skip_init_cols = false;               // RAY_OPT_SKIP_INIT_COLUMNS
    int src_stride = 1;                        // RAY_OPT_SOURCE_STRIDE
    int fan_nd = 0;                            // RAY_OPT_FAN_DIRECTIONS
    // tuning / test options (RAY_OPT_TUNE_*): explicit per-context settings, never read from the environment
    int tune_block = 256, tune_wave_dirs = 0, tune_sort_key = 0, tune_lds_rows = -1, tune_merge_grid = 0, tune_slab_mb = 0, tune_merge_entries = 0;
    // one retired set of matrix arrays (ray_frechet_destroy parks them here, the next ray_forward_operator
    // takes them back if they are large enough): the loop frees M and builds the next one every iteration
    static constexpr int kPool = 11;           // indptr, indices, data (CSR, on export) | uoff, ucols, uvals, rowcnt (block rows) | band offsets, M x (scatter passes)
    void *csr_pool[kPool] = {};
    size_t csr_cap[kPool] = {};
    // handles created on this context and not yet destroyed: ray_ctx_destroy releases their device memory
    // and detaches them, so that destroying a handle after its context is harmless (only the struct is left)
    std::vector<struct ray_medium *> media;
    std::vector<struct ray_frechet *> frechets;
};
static void detach_medium(struct ray_medium *m);
static void detach_frechet(struct ray_frechet *f);
template <class T> static void forget(std::vector<T *> &v, T *h)
{
    for (size_t k = 0; k < v.size(); ++k)
        if (v[k] == h) { v[k] = v.back(); v.pop_back(); return; }
}

static hipError_t ctx_workspace(ray_ctx *ctx, int k, size_t bytes, void **out)
{
    if (bytes > ctx->ws_cap[k]) {
        if (ctx->ws[k]) (void)hipFree(ctx->ws[k]);
        ctx->ws[k] = nullptr; ctx->ws_cap[k] = 0;
        hipError_t e = hipMalloc(&ctx->ws[k], bytes);
        if (e != hipSuccess) return e;
        ctx->ws_cap[k] = bytes;
    }
    *out = ctx->ws[k];
    return hipSuccess;
}

struct ray_medium {
    ray_ctx *ctx = nullptr;
    int kind = kSinProd;
    SinProdDev sp{};
    RadialDev rd{};
    int N = 0;
    double ext = 0;
    double *d_c = nullptr, *d_rec = nullptr, *d_recf = nullptr, *d_p = nullptr;
    double *d_hrec = nullptr;                  // hess records, built on first use by the Frechet path
    bool hess_valid = false;
    std::vector<double> p;                     // host copy of the nodes
};

struct DevBuf {                                // scoped device allocation for host-pointer calls
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

struct BorrowedBuf {                           // same accessors as DevBuf, memory owned by the context
    void *p;
    template <class T> T *as() { return static_cast<T *>(p); }
};
struct WsBuf {                                 // DevBuf's interface over workspace slot k of the context
    ray_ctx *ctx;
    int k;
    void *p = nullptr;
    WsBuf(ray_ctx *c, int slot) : ctx(c), k(slot) {}
    hipError_t alloc(size_t bytes) { return ctx_workspace(ctx, k, bytes ? bytes : 8, &p); }
    template <class T> T *as() { return static_cast<T *>(p); }
};
// CSR arrays of a ray_frechet: from the context's parked triple when it fits, else fresh
static hipError_t csr_take(ray_ctx *ctx, int which, size_t bytes, void **out)
{
    if (ctx->csr_pool[which] && ctx->csr_cap[which] >= bytes) {
        *out = ctx->csr_pool[which];
        ctx->csr_pool[which] = nullptr; ctx->csr_cap[which] = 0;
        return hipSuccess;
    }
    return hipMalloc(out, bytes ? bytes : 8);
}

static int use(ray_ctx *ctx)
{
    if (!ctx) return fail(RAY_ERR_INVALID, "null context");
    HIPCHK(hipSetDevice(ctx->device));
    return RAY_OK;
}

static void time_begin(ray_ctx *ctx, hipEvent_t &e0, hipEvent_t &e1)
{
    e0 = e1 = nullptr;
    if (!ctx->timing) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
    (void)hipEventRecord(e0, ctx->stream);
}
static void time_end(ray_ctx *ctx, hipEvent_t e0, hipEvent_t e1)
{
    if (!e0) return;
    (void)hipEventRecord(e1, ctx->stream);
    ctx->pending.emplace_back(e0, e1);
}
struct TimedSpan {                             // time_begin / time_end that survives early returns
    ray_ctx *ctx;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool done = false;
    explicit TimedSpan(ray_ctx *c) : ctx(c) { time_begin(ctx, e0, e1); }
    void end() { time_end(ctx, e0, e1); done = true; }
    ~TimedSpan()
    {
        if (!done && e0) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
    }
};
static void drain_timing(ray_ctx *ctx)
{
    for (auto &pr : ctx->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) ctx->kernel_ms += ms;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    ctx->pending.clear();
}

// the device-side staging of the boundary grid: source angles, their libm cos/sin, directions.
// Cached on the context and rebuilt only when the (ns, nd, dir, theta-block) key changes, so a
// reconstruction loop / sweep that re-shoots the same grid uploads nothing.
struct InitTables {
    DevBuf src, cs, sn, dir;
    int ns = -1, nd = -1, b = -1, e = -1, stride = 1;
    double lo = 0, hi = 0;
    bool matches(const ray_shoot_params *p, int st) const
    {
        return ns == p->ns && nd == p->nd && b == p->src_begin && e == p->src_end && stride == st && lo == p->dir_lo && hi == p->dir_hi;
    }
    int upload(ray_ctx *ctx, const ray_shoot_params *p)
    {
        const int st = ctx->src_stride, nloc = (p->src_end - p->src_begin + st - 1) / st;
        std::vector<double> source(p->ns + 1), direct(p->nd + 2);
        hostmath::linspace(0.0, 2 * M_PI, p->ns + 1, source.data());      // NonObstacle.jl:55
        hostmath::linspace(p->dir_lo, p->dir_hi, p->nd + 2, direct.data()); // NonObstacle.jl:56
        std::vector<double> s(nloc), c(nloc), sn_(nloc);
        for (int i = 0; i < nloc; ++i) {
            s[i] = source[p->src_begin + i * st];
            trig::sincos_bounded(s[i], sn_[i], c[i]);                      // NonObstacle.jl:57 (ray_trig.h: 0 <= s <= 2 pi)
        }
        HIPCHK(src.alloc(sizeof(double) * nloc));
        HIPCHK(cs.alloc(sizeof(double) * nloc));
        HIPCHK(sn.alloc(sizeof(double) * nloc));
        HIPCHK(dir.alloc(sizeof(double) * p->nd));
        HIPCHK(hipMemcpyAsync(src.p, s.data(), sizeof(double) * nloc, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(cs.p, c.data(), sizeof(double) * nloc, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(sn.p, sn_.data(), sizeof(double) * nloc, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(dir.p, direct.data() + 1, sizeof(double) * p->nd, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ns = p->ns; nd = p->nd; b = p->src_begin; e = p->src_end; stride = st; lo = p->dir_lo; hi = p->dir_hi;
        return RAY_OK;
    }
};

static int init_tables(ray_ctx *ctx, const ray_shoot_params *p, InitTables **out)
{
    if (ctx->init && ctx->init->matches(p, ctx->src_stride)) { *out = ctx->init; return RAY_OK; }
    HIPCHK(hipStreamSynchronize(ctx->stream));          // a launch may still read the old tables
    delete ctx->init;
    ctx->init = new InitTables;
    int rc = ctx->init->upload(ctx, p);
    if (rc) { delete ctx->init; ctx->init = nullptr; return rc; }
    *out = ctx->init;
    return RAY_OK;
}

extern "C" {

// 101: ray_ctx_set_stream took its `mode` argument (round 2 changed the signature in place under version 100: a caller
// built against the two-argument form must check ray_version() >= 101 before passing three)
int ray_version(void) { return 101; }

// Which HIP runtime and which RCCL this process bound the library's calls to (the dynamic loader resolves them once
// per process: a process that loaded another copy first -- torch bundles its own -- runs the library on THAT copy).
int ray_runtime_info(char *buf, size_t len)
{
    if (!buf || !len) return fail(RAY_ERR_INVALID, "ray_runtime_info: null buffer");
    int hip_rt = 0, nccl_v = 0;
    (void)hipRuntimeGetVersion(&hip_rt);
    (void)ncclGetVersion(&nccl_v);
    Dl_info a{}, b{};
    const char *hp = dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &a) && a.dli_fname ? a.dli_fname : "?";
    const char *np = dladdr(reinterpret_cast<void *>(&ncclAllReduce), &b) && b.dli_fname ? b.dli_fname : "?";
    std::snprintf(buf, len, "hip %d %s; rccl %d %s", hip_rt, hp, nccl_v, np);
    return RAY_OK;
}
const char *ray_last_error(void) { return g_err.c_str(); }

int ray_linspace(double a, double b, int n, double *out)
{
    if (!out || n < 1) return fail(RAY_ERR_INVALID, "linspace needs n >= 1 and an output buffer");
    hostmath::linspace(a, b, n, out);
    return RAY_OK;
}

int ray_device_count(int *n)
{
    if (!n) return fail(RAY_ERR_INVALID, "null n");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(RAY_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *n = c;
    return RAY_OK;
}

int ray_ctx_create(int device, ray_ctx **out)
{
    if (!out) return fail(RAY_ERR_INVALID, "null out");
    *out = nullptr;
    int n = 0;
    int rc = ray_device_count(&n);
    if (rc) return rc;
    if (device < 0 || device >= n) return fail(RAY_ERR_INVALID, "device index out of range (no GPU visible?)");
    HIPCHK(hipSetDevice(device));
    ray_ctx *c = new ray_ctx;
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_steps, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(c->d_steps, 0, sizeof(unsigned long long), c->own_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->own_stream);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_partial, sizeof(double) * 2 * kMisfitBlocks);
    // this device's copy of the static scratch (once per device, with the device current)
    if (e == hipSuccess && device < kMaxDevices) {
        std::lock_guard<std::mutex> lock(g_static_mu);       // two threads creating contexts on one device
        if (!g_static_partial[device].load(std::memory_order_acquire)) {
            double *ptr = nullptr;
            e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(g_misfit_scratch));
            if (e == hipSuccess) g_static_partial[device].store(ptr, std::memory_order_release);
        }
    }
    if (e != hipSuccess) {
        std::string msg = std::string("ray_ctx_create: ") + hipGetErrorString(e);
        ray_ctx_destroy(c);
        return fail(RAY_ERR_HIP, msg);
    }
    c->stream = c->own_stream;
    *out = c;
    return RAY_OK;
}

int ray_ctx_destroy(ray_ctx *ctx)
{
    if (!ctx) return RAY_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    drain_timing(ctx);
    for (ray_medium *m : ctx->media) detach_medium(m);
    for (ray_frechet *f : ctx->frechets) detach_frechet(f);
    delete ctx->init;
    if (ctx->d_steps) (void)hipFree(ctx->d_steps);
    if (ctx->d_partial) (void)hipFree(ctx->d_partial);
    for (int k = 0; k < ray_ctx::kWs; ++k) if (ctx->ws[k]) (void)hipFree(ctx->ws[k]);
    for (int k = 0; k < ray_ctx::kPool; ++k) if (ctx->csr_pool[k]) (void)hipFree(ctx->csr_pool[k]);
    ctx->ring.release();
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return RAY_OK;
}

int ray_ctx_set_stream(ray_ctx *ctx, int mode, void *hip_stream)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (mode != RAY_STREAM_OWN && mode != RAY_STREAM_BORROWED) return fail(RAY_ERR_INVALID, "bad stream mode");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    drain_timing(ctx);                              // its events were recorded on the stream being left
    // BORROWED: the caller's handle exactly as given -- NULL is the legacy default (null) stream
    ctx->stream = mode == RAY_STREAM_BORROWED ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return RAY_OK;
}

int ray_ctx_synchronize(ray_ctx *ctx)
{
    int rc = use(ctx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RAY_OK;
}

// gives back what the context keeps between calls (the Frechet workspaces, the parked CSR arrays)
int ray_ctx_release_workspaces(ray_ctx *ctx)
{
    int rc = use(ctx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < ray_ctx::kWs; ++k)
        if (ctx->ws[k]) { (void)hipFree(ctx->ws[k]); ctx->ws[k] = nullptr; ctx->ws_cap[k] = 0; }
    for (int k = 0; k < ray_ctx::kPool; ++k)
        if (ctx->csr_pool[k]) { (void)hipFree(ctx->csr_pool[k]); ctx->csr_pool[k] = nullptr; ctx->csr_cap[k] = 0; }
    const bool refused = ctx->ring.refused;
    ctx->ring.release();
    ctx->ring.refused = refused;
    return RAY_OK;
}

// ---- media ------------------------------------------------------------------------------------
static int build_tables(ray_medium *med)
{
    ray_ctx *ctx = med->ctx;
    const int n1 = med->N - 1;
    const int64_t cells = (int64_t)n1 * n1;
    const double dx = 2 * med->ext / (med->N - 1);                      // NonObstacle.jl:191
    hipLaunchKernelGGL(k_build_tables, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream,
                       med->d_c, med->N, dx, med->p[1] - med->p[0], med->d_rec, med->d_recf);
    HIPCHK(hipGetLastError());
    med->hess_valid = false;
    return RAY_OK;
}

static int ensure_hess(ray_medium *med)
{
    if (med->hess_valid) return RAY_OK;
    const int n1 = med->N - 1;
    const int64_t cells = (int64_t)n1 * n1;
    if (!med->d_hrec) HIPCHK(hipMalloc((void **)&med->d_hrec, (size_t)cells * 12 * sizeof(double)));
    const double dx = 2 * med->ext / (med->N - 1);                      // NonObstacle.jl:191
    hipLaunchKernelGGL(k_build_hess, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, med->ctx->stream,
                       med->d_rec, med->N, dx, med->d_hrec);
    HIPCHK(hipGetLastError());
    med->hess_valid = true;
    return RAY_OK;
}

int ray_medium_create_grid(ray_ctx *ctx, const double *c_host, int N, double ext, ray_medium **out)
{
    if (!out) return fail(RAY_ERR_INVALID, "null out");
    *out = nullptr;
    int rc = use(ctx);
    if (rc) return rc;
    if (!c_host || N < 4 || !(ext > 0)) return fail(RAY_ERR_INVALID, "grid medium needs c, N >= 4, ext > 0");
    if (N > 4096) return fail(RAY_ERR_UNSUPPORTED, "grid medium supports N <= 4096 (24-bit cell ids, LDS node table)");
    ray_medium *m = new ray_medium;
    m->ctx = ctx; m->kind = kGrid; m->N = N; m->ext = ext;
    m->p.resize(N);
    hostmath::linspace(-ext, ext, N, m->p.data());                     // NonObstacle.jl:191
    const size_t nc = (size_t)N * N, nr = (size_t)(N - 1) * (N - 1) * 12;
    hipError_t e = hipMalloc((void **)&m->d_c, nc * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_rec, nr * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_recf, nr * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_p, (size_t)N * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(m->d_p, m->p.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
        std::string msg = std::string("ray_medium_create_grid: ") + hipGetErrorString(e);
        ray_medium_destroy(m);
        return fail(e == hipErrorOutOfMemory ? RAY_ERR_NOMEM : RAY_ERR_HIP, msg);
    }
    rc = ray_medium_update_grid(m, c_host);
    if (rc) { ray_medium_destroy(m); return rc; }
    ctx->media.push_back(m);
    *out = m;
    return RAY_OK;
}

int ray_medium_update_grid(ray_medium *med, const double *c_host)
{
    if (!med || med->kind != kGrid || !c_host) return fail(RAY_ERR_INVALID, "update_grid: not a grid medium / null c");
    int rc = use(med->ctx);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(med->d_c, c_host, (size_t)med->N * med->N * sizeof(double), hipMemcpyHostToDevice, med->ctx->stream));
    rc = build_tables(med);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(med->ctx->stream));                     // c_host may be reused on return
    return RAY_OK;
}

int ray_medium_update_grid_dev(ray_medium *med, const double *c_dev)
{
    if (!med || med->kind != kGrid || !c_dev) return fail(RAY_ERR_INVALID, "update_grid_dev: not a grid medium / null c");
    int rc = use(med->ctx);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(med->d_c, c_dev, (size_t)med->N * med->N * sizeof(double), hipMemcpyDeviceToDevice, med->ctx->stream));
    return build_tables(med);
}

int ray_medium_export_tables(ray_medium *med, double *eval_host, double *grad_host, double *p_host)
{
    if (!med || med->kind != kGrid) return fail(RAY_ERR_INVALID, "export_tables: not a grid medium");
    int rc = use(med->ctx);
    if (rc) return rc;
    const int N = med->N;
    const int64_t n1 = N - 1, n2 = N - 2;
    std::vector<double> rec((size_t)n1 * n1 * 12);
    HIPCHK(hipMemcpyAsync(rec.data(), med->d_rec, rec.size() * sizeof(double), hipMemcpyDeviceToHost, med->ctx->stream));
    HIPCHK(hipStreamSynchronize(med->ctx->stream));
    if (eval_host)
        for (int64_t j = 0; j < n1; ++j)
            for (int64_t i = 0; i < n1; ++i)
                for (int k = 0; k < 4; ++k)
                    eval_host[i + j * n1 + k * n1 * n1] = rec[(size_t)(i + j * n1) * 12 + k];
    if (grad_host)
        for (int64_t j = 0; j < n2; ++j)
            for (int64_t i = 0; i < n2; ++i)
                for (int k = 0; k < 8; ++k)
                    grad_host[i + j * n2 + k * n2 * n2] = rec[(size_t)(i + j * n1) * 12 + 4 + k];
    if (p_host) std::memcpy(p_host, med->p.data(), (size_t)N * sizeof(double));
    return RAY_OK;
}

int ray_medium_create_sinprod(ray_ctx *ctx, double A, double wx, double wy, double gx, double gy, ray_medium **out)
{
    if (!out) return fail(RAY_ERR_INVALID, "null out");
    *out = nullptr;
    if (!ctx) return fail(RAY_ERR_INVALID, "null context");
    ray_medium *m = new ray_medium;
    m->ctx = ctx; m->kind = kSinProd;
    m->sp = SinProdDev{A, wx, wy, gx, gy};
    ctx->media.push_back(m);
    *out = m;
    return RAY_OK;
}

int ray_medium_create_radial(ray_ctx *ctx, int K, const double *a, const double *xk, const double *yk, ray_medium **out)
{
    if (!out) return fail(RAY_ERR_INVALID, "null out");
    *out = nullptr;
    if (!ctx) return fail(RAY_ERR_INVALID, "null context");
    if (K < 1 || K > kRadialMax || !a || !xk || !yk) return fail(RAY_ERR_INVALID, "radial medium: 1..4 terms (a, x, y)");
    ray_medium *m = new ray_medium;
    m->ctx = ctx; m->kind = kRadial;
    m->rd.K = K;
    for (int k = 0; k < K; ++k) { m->rd.a[k] = a[k]; m->rd.x[k] = xk[k]; m->rd.y[k] = yk[k]; }
    ctx->media.push_back(m);
    *out = m;
    return RAY_OK;
}

// device memory of a medium released, handle left without a context (ray_ctx_destroy, ray_medium_destroy)
static void detach_medium(ray_medium *med)
{
    if (med->d_c) (void)hipFree(med->d_c);
    if (med->d_rec) (void)hipFree(med->d_rec);
    if (med->d_recf) (void)hipFree(med->d_recf);
    if (med->d_p) (void)hipFree(med->d_p);
    if (med->d_hrec) (void)hipFree(med->d_hrec);
    med->d_c = med->d_rec = med->d_recf = med->d_p = med->d_hrec = nullptr;
    med->ctx = nullptr;
}

int ray_medium_destroy(ray_medium *med)
{
    if (!med) return RAY_OK;
    if (med->ctx) {                                   // else: its context went first and took the memory
        (void)hipSetDevice(med->ctx->device);
        (void)hipStreamSynchronize(med->ctx->stream);
        forget(med->ctx->media, med);
        detach_medium(med);
    }
    delete med;
    return RAY_OK;
}

static MediumDev medium_dev(const ray_medium *m)
{
    MediumDev d{};
    d.sp = m->sp;
    d.rd = m->rd;
    if (m->kind == kGrid) {
        d.g.rec = m->d_rec; d.g.recf = m->d_recf; d.g.hrec = m->d_hrec; d.g.p = m->d_p; d.g.N = m->N;
        d.g.p1 = m->p[0];
        d.g.h = m->p[1] - m->p[0];                                      // NonObstacle.jl:32
        d.g.rh = 1.0 / d.g.h;
        d.g.nb = -d.g.p1 * d.g.rh;
    }
    return d;
}

// workgroup size of the integration kernels (multiple of 64); RAY_OPT_TUNE_BLOCK_THREADS overrides for experiments
static int block_threads(const ray_ctx *ctx) { return ctx->tune_block; }
static size_t nodes_lds_bytes(const ray_medium *med, bool fused)
{
    return med->kind == kGrid ? grid_lds_bytes(med->N, block_threads(med->ctx), !fused) : 0;   // no node table in kFused
}

// Wave shape of the fan mapping: a patch of 2^(6-dshift) adjacent sources x 2^dshift directions.  Default 2 x 32
// (narrower fans touch fewer distinct cells per wave than 64 directions of one source: +3..7 % on every gridded
// workload, profiles/r1/ablations.md); fewer directions than that -> more sources per wave; fewer sources than a
// workgroup holds (a single fan, a group member with one or two) -> more directions per wave while there are
// directions to take, so that the waves are full instead of carrying lanes without a source.
static int pick_dshift(int nd, int nsrc, int wpb)
{
    int dshift = 5;
    while (dshift > 2 && (1 << (dshift - 1)) >= nd) --dshift;
    while (dshift < 6 && (wpb << (6 - dshift)) > nsrc && (1 << dshift) < nd) ++dshift;
    return dshift;
}

// ---- shooting -----------------------------------------------------------------------------------
static int launch_shoot(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob, ShootArgs &a)
{
    const int obk = ob ? ob->kind : RAY_OB_NONE;
    if (obk != RAY_OB_NONE && obk != RAY_OB_CIRCLE && obk != RAY_OB_TREFOIL) return fail(RAY_ERR_UNSUPPORTED, "unknown obstacle kind");
    if (obk == RAY_OB_CIRCLE && !(ob->r0 > 0)) return fail(RAY_ERR_INVALID, "circle obstacle needs r0 > 0");
    if (obk == RAY_OB_TREFOIL && !(ob->r0 > 0 && __builtin_fabs(ob->reserved[0]) < ob->r0))
        return fail(RAY_ERR_INVALID, "trefoil obstacle needs R > |rho| (reserved[0]) so that the level set encloses the origin");
    a.med = medium_dev(med);
    a.ob_kind = obk;
    a.ob_r0 = ob ? ob->r0 : 0.0;
    a.ob_rho = ob ? ob->reserved[0] : 0.0;
    a.ob_rho_n = ob ? ob->reserved[1] : 0.0;
    {   // the level set's largest radius (circle: r0; trefoil: R + |rho|) with a margin far above any rounding of ob
        const double far = (std::fabs(a.ob_r0) + std::fabs(a.ob_rho)) * (1 + 1e-9);
        a.ob_far2 = far * far * (1 + 1e-9);
    }
    a.ob_in2 = a.ob_lo2 = a.ob_hi2 = 0.0;
    if (obk == RAY_OB_CIRCLE) {
        // largest double s with fl(sqrt(s)) <= r0 (host sqrt: correctly rounded, as the device's is): start at r0^2, walk
        double t = a.ob_r0 * a.ob_r0;
        while (std::sqrt(t) > a.ob_r0) t = std::nextafter(t, 0.0);
        while (std::sqrt(std::nextafter(t, INFINITY)) <= a.ob_r0) t = std::nextafter(t, INFINITY);
        a.ob_in2 = t;
        // the doubles q next to r0 with |q - r0| <= 1e-15 (the difference of neighbours is exact), then their pre-images
        double qlo = a.ob_r0, qhi = a.ob_r0;
        while (std::fabs(std::nextafter(qlo, 0.0) - a.ob_r0) <= 1e-15) qlo = std::nextafter(qlo, 0.0);
        while (std::fabs(std::nextafter(qhi, INFINITY) - a.ob_r0) <= 1e-15) qhi = std::nextafter(qhi, INFINITY);
        double lo2 = qlo * qlo, hi2 = qhi * qhi;
        while (std::sqrt(lo2) >= qlo) lo2 = std::nextafter(lo2, 0.0);         // below the pre-image of qlo ...
        while (std::sqrt(lo2) < qlo) lo2 = std::nextafter(lo2, INFINITY);     // ... its smallest member
        while (std::sqrt(hi2) <= qhi) hi2 = std::nextafter(hi2, INFINITY);
        while (std::sqrt(hi2) > qhi) hi2 = std::nextafter(hi2, 0.0);          // the largest s with fl(sqrt(s)) <= qhi
        a.ob_lo2 = lo2; a.ob_hi2 = hi2;
    }
    for (int k = 0; k < kRadialMax; ++k) a.anchor_k[k] = 1;
    if (med->kind == kSinProd) {
        const double cmax = 1 + std::fabs(med->sp.A);
        a.anchor_k[0] = trig::anchor_period(med->sp.wx, a.dt, cmax);
        a.anchor_k[1] = trig::anchor_period(med->sp.wy, a.dt, cmax);
    } else if (med->kind == kRadial) {
        double cmax = 1;
        for (int k = 0; k < med->rd.K; ++k) cmax += std::fabs(med->rd.a[k]);
        for (int k = 0; k < med->rd.K; ++k) a.anchor_k[k] = trig::anchor_period(3.141592653589793, a.dt, cmax);
    }
    a.step_counter = ctx->d_steps;
    if (a.max_steps <= 0) a.max_steps = 1000000;
    if (a.num == 0) return RAY_OK;
    const int bt = block_threads(ctx), wpb = bt / 64;
    unsigned nblk;
    if (a.init) {
        // wave shape: a patch of 2 adjacent sources x 32 directions by default (narrower fans touch fewer
        // distinct cells per wave than 64 directions of one source: +3..7 % on every gridded workload,
        // profiles/r1/ablations.md); fewer directions than that -> more sources per wave.
        // RAY_OPT_TUNE_WAVE_DIRS = 64/32/16/8/4 overrides.
        const int v = ctx->tune_wave_dirs;
        const int opt_dshift = v == 64 ? 6 : v == 32 ? 5 : v == 16 ? 4 : v == 8 ? 3 : v == 4 ? 2 : 0;
        const int dshift = opt_dshift ? opt_dshift : pick_dshift(a.nd, a.nsrc, wpb);
        a.dshift = dshift;
        const int spb = wpb << (6 - dshift), dpw = 1 << dshift;      // sources per block, directions per wave
        a.ngroup = (a.nsrc + spb - 1) / spb;
        nblk = (unsigned)(((a.nd + dpw - 1) / dpw) * a.ngroup);
    } else nblk = (unsigned)((a.num + bt - 1) / bt);
    const dim3 grid(nblk), block(bt);
    const bool fused = ctx->arith == RAY_ARITH_FUSED;
    const size_t lds = nodes_lds_bytes(med, fused);
    if (lds > 160 * 1024) return fail(RAY_ERR_UNSUPPORTED, "grid too large for the LDS node table");
    hipEvent_t e0, e1;
    time_begin(ctx, e0, e1);
#define RAY_LAUNCH_SHOOT(MED, OB, AR, PATH, LDS) hipLaunchKernelGGL((k_shoot<MED, OB, AR, PATH>), grid, block, LDS, ctx->stream, a)
#define RAY_LAUNCH_BY_OB(MED, AR, PATH, LDS)                                  \
    do {                                                                      \
        if (obk == RAY_OB_NONE) RAY_LAUNCH_SHOOT(MED, kObNone, AR, PATH, LDS);  \
        else if (obk == RAY_OB_CIRCLE) RAY_LAUNCH_SHOOT(MED, kObCircle, AR, PATH, LDS); \
        else RAY_LAUNCH_SHOOT(MED, kObTrefoil, AR, PATH, LDS);                  \
    } while (0)
    if (a.path) {
        if (med->kind == kGrid) {
            if (fused) RAY_LAUNCH_BY_OB(kGrid, kFused, 1, lds);
            else RAY_LAUNCH_BY_OB(kGrid, kExact, 1, lds);
        } else if (med->kind == kRadial) {
            if (fused) RAY_LAUNCH_BY_OB(kRadial, kFused, 1, 0);
            else RAY_LAUNCH_BY_OB(kRadial, kExact, 1, 0);
        } else if (fused) RAY_LAUNCH_BY_OB(kSinProd, kFused, 1, 0);
        else RAY_LAUNCH_BY_OB(kSinProd, kExact, 1, 0);
    } else if (med->kind == kGrid) {
        if (fused) RAY_LAUNCH_BY_OB(kGrid, kFused, 0, lds);
        else RAY_LAUNCH_BY_OB(kGrid, kExact, 0, lds);
    } else if (med->kind == kRadial) {
        if (fused) RAY_LAUNCH_BY_OB(kRadial, kFused, 0, 0);
        else RAY_LAUNCH_BY_OB(kRadial, kExact, 0, 0);
    } else if (fused) RAY_LAUNCH_BY_OB(kSinProd, kFused, 0, 0);
    else RAY_LAUNCH_BY_OB(kSinProd, kExact, 0, 0);
#undef RAY_LAUNCH_BY_OB
#undef RAY_LAUNCH_SHOOT
    time_end(ctx, e0, e1);
    HIPCHK(hipGetLastError());
    ctx->rays += (uint64_t)a.num;
    ctx->launches += 1;
    return RAY_OK;
}

static int check_med(ray_ctx *ctx, const ray_medium *med)
{
    if (!med) return fail(RAY_ERR_INVALID, "null medium");
    if (med->ctx != ctx) return fail(RAY_ERR_INVALID, "medium belongs to another context");
    return RAY_OK;
}

static int check_params(const ray_shoot_params *p)
{
    if (!p) return fail(RAY_ERR_INVALID, "null params");
    if (p->ns < 1 || p->nd < 1) return fail(RAY_ERR_INVALID, "ns, nd must be >= 1");
    if (p->src_begin < 0 || p->src_end > p->ns || p->src_begin > p->src_end)
        return fail(RAY_ERR_INVALID, "need 0 <= src_begin <= src_end <= ns");
    if (!(p->dt > 0)) return fail(RAY_ERR_INVALID, "dt must be > 0");
    if (p->time_unit != RAY_TIME_SECONDS && p->time_unit != RAY_TIME_STEPS) return fail(RAY_ERR_INVALID, "bad time_unit");
    return RAY_OK;
}

// sources of a call: src_begin, src_begin + stride, ... below src_end (RAY_OPT_SOURCE_STRIDE of the context)
static int call_sources(const ray_ctx *ctx, const ray_shoot_params *prm)
{
    return (prm->src_end - prm->src_begin + ctx->src_stride - 1) / ctx->src_stride;
}

// sources [i0, i0 + cnt) of the call's own sources (local indices: source src_begin + i * stride), rows i0 * nd .. of m
static int shoot_block(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob, const ray_shoot_params *prm,
                       const InitTables *tab, double *m_dev, int64_t ld, int i0, int cnt,
                       int32_t *status, int32_t *nsteps, int32_t *nreflect)
{
    const int64_t r0 = (int64_t)i0 * prm->nd;
    ShootArgs a{};
    a.init = 1; a.nd = prm->nd; a.nsrc = cnt;
    a.src = tab->src.as<const double>() + i0; a.src_cos = tab->cs.as<const double>() + i0; a.src_sin = tab->sn.as<const double>() + i0;
    a.direct = tab->dir.as<const double>();
    a.half_pi = M_PI * 0.5;
    a.dt = prm->dt; a.time_unit = prm->time_unit; a.max_steps = prm->max_steps;
    a.m = m_dev + r0; a.num = (int64_t)cnt * prm->nd; a.ld = ld;
    a.status = status ? status + r0 : nullptr; a.nsteps = nsteps ? nsteps + r0 : nullptr; a.nreflect = nreflect ? nreflect + r0 : nullptr;
    return launch_shoot(ctx, med, ob, a);
}

static int shoot_to_exit_impl(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob,
                              const ray_shoot_params *prm, double *m_dev, int64_t ld,
                              int32_t *status, int32_t *nsteps, int32_t *nreflect)
{
    const int64_t num = (int64_t)call_sources(ctx, prm) * prm->nd;
    if (ld < num) return fail(RAY_ERR_INVALID, "ld smaller than the number of rows");
    InitTables *tab = nullptr;
    int rc = init_tables(ctx, prm, &tab);
    if (rc) return rc;
    return shoot_block(ctx, med, ob, prm, tab, m_dev, ld, 0, call_sources(ctx, prm), status, nsteps, nreflect);
}

int ray_shoot_to_exit_dev(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob,
                          const ray_shoot_params *prm, double *m_dev, int64_t ld,
                          int32_t *status_dev, int32_t *nsteps_dev, int32_t *nreflect_dev)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_med(ctx, med)) || (rc = check_params(prm))) return rc;
    if (!m_dev) return fail(RAY_ERR_INVALID, "null m");
    return shoot_to_exit_impl(ctx, med, ob, prm, m_dev, ld, status_dev, nsteps_dev, nreflect_dev);
}

// the largest double s with fl(sqrt(s)) <= r (r > 0): `norm(X) > r` on s = x^2 + y^2 without the square root
static double sqrt_le_threshold(double r)
{
    double t = r * r;
    while (std::sqrt(t) > r) t = std::nextafter(t, 0.0);
    while (std::sqrt(std::nextafter(t, INFINITY)) <= r) t = std::nextafter(t, INFINITY);
    return t;
}

static int copy_cols_h2d(ray_ctx *ctx, double *dst, int64_t ldd, const double *src, int64_t lds, int64_t rows, int c0, int ncol)
{
    if (ldd == rows && lds == rows) {                  // the columns abut on both sides: one flat copy
        HIPCHK(hipMemcpyAsync(dst + (size_t)c0 * ldd, src + (size_t)c0 * lds, (size_t)rows * 8 * (size_t)ncol, hipMemcpyHostToDevice, ctx->stream));
        return RAY_OK;
    }
    for (int c = c0; c < c0 + ncol; ++c)               // one flat copy per column (at most nine)
        HIPCHK(hipMemcpyAsync(dst + (size_t)c * ldd, src + (size_t)c * lds, (size_t)rows * 8, hipMemcpyHostToDevice, ctx->stream));
    return RAY_OK;
}
static int copy_cols_d2h(ray_ctx *ctx, double *dst, int64_t ldd, const double *src, int64_t lds, int64_t rows, int c0, int ncol)
{
    if (ldd == rows && lds == rows) {                  // the columns abut on both sides: one flat copy
        HIPCHK(hipMemcpyAsync(dst + (size_t)c0 * ldd, src + (size_t)c0 * lds, (size_t)rows * 8 * (size_t)ncol, hipMemcpyDeviceToHost, ctx->stream));
        return RAY_OK;
    }
    for (int c = c0; c < c0 + ncol; ++c)               // one flat copy per column (at most nine)
        HIPCHK(hipMemcpyAsync(dst + (size_t)c * ldd, src + (size_t)c * lds, (size_t)rows * 8, hipMemcpyDeviceToHost, ctx->stream));
    return RAY_OK;
}

// staging slots of the host-pointer entry points (grow-only, kept by the context: a caller that shoots the same
// sweep every iteration -- a blocking Julia ccall -- allocates on its first call only)
enum { kWsM = 12, kWsSt = 13, kWsNs = 14, kWsNr = 15, kWsX = 16, kWsObs = 17, kWsOut = 18, kWsMask = 19, kWsMis = 20,
       kWsPath = 21, kWsNpts = 22, kWsAux = 23 };

struct RayOutBufs {
    WsBuf st, ns, nr;
    explicit RayOutBufs(ray_ctx *c) : st(c, kWsSt), ns(c, kWsNs), nr(c, kWsNr) {}
    int alloc(int64_t num) {
        HIPCHK(st.alloc(sizeof(int32_t) * (size_t)num));
        HIPCHK(ns.alloc(sizeof(int32_t) * (size_t)num));
        HIPCHK(nr.alloc(sizeof(int32_t) * (size_t)num));
        return RAY_OK;
    }
    int fetch(ray_ctx *ctx, int64_t num, int32_t *status, int32_t *nsteps, int32_t *nreflect) {
        if (status) HIPCHK(hipMemcpyAsync(status, st.p, sizeof(int32_t) * (size_t)num, hipMemcpyDeviceToHost, ctx->stream));
        if (nsteps) HIPCHK(hipMemcpyAsync(nsteps, ns.p, sizeof(int32_t) * (size_t)num, hipMemcpyDeviceToHost, ctx->stream));
        if (nreflect) HIPCHK(hipMemcpyAsync(nreflect, nr.p, sizeof(int32_t) * (size_t)num, hipMemcpyDeviceToHost, ctx->stream));
        return RAY_OK;
    }
};

// the ring's slots (>= bytes each), its copy stream, events and workers; false when page-locked memory is not to be had
constexpr int kCopyWorkers = 6;
static bool ring_prepare(ray_ctx *ctx, size_t bytes, int nblocks)
{
    HostRing &r = ctx->ring;
    if (r.refused) return false;
    hipError_t e = hipSuccess;
    if (!r.copy_stream) {
        e = hipStreamCreateWithFlags(&r.copy_stream, hipStreamNonBlocking);
        for (int k = 0; k < HostRing::kSlots && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&r.filled[k], hipEventDisableTiming);
        if (e == hipSuccess && !r.pool) r.pool = new CopyPool(kCopyWorkers);
    }
    while (e == hipSuccess && (int)r.blk_done.size() < nblocks) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) r.blk_done.push_back(ev);
    }
    if (e == hipSuccess && r.cap < bytes) {
        for (int k = 0; k < HostRing::kSlots; ++k) {
            if (r.slot[k]) (void)hipHostFree(r.slot[k]);
            r.slot[k] = nullptr;
        }
        r.cap = 0;
        for (int k = 0; k < HostRing::kSlots && e == hipSuccess; ++k) e = hipHostMalloc(&r.slot[k], bytes, hipHostMallocDefault);
        if (e == hipSuccess) r.cap = bytes;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        r.release();
        r.refused = true;                       // e.g. no page-locked memory for this user: the plain copies still work
        return false;
    }
    return true;
}

// One block of result rows [r0, r0 + rows): which device arrays go to which host arrays, packed into a ring slot as
// [column c0 | ... | column 8 | status | nsteps | nreflect] (each `rows` long)
struct RowBlockIo {
    const double *dm; int64_t ldd;            // device m (num, 9)
    double *hm; int64_t ldh;                  // host m
    int c0;                                   // first column that is copied
    const int32_t *dint[3]; int32_t *hint[3]; // status / nsteps / nreflect (host pointer null: not wanted)
    size_t row_bytes() const
    {
        size_t b = (size_t)(9 - c0) * 8;
        for (int k = 0; k < 3; ++k) b += hint[k] ? 4 : 0;
        return b;
    }
};
static hipError_t ring_enqueue_block(ray_ctx *ctx, const RowBlockIo &io, int slot, int64_t r0, int64_t rows)
{
    HostRing &r = ctx->ring;
    char *dst = static_cast<char *>(r.slot[slot]);
    hipError_t e = hipSuccess;
    for (int c = io.c0; c < 9 && e == hipSuccess; ++c, dst += rows * 8)
        e = hipMemcpyAsync(dst, io.dm + (size_t)c * io.ldd + r0, (size_t)rows * 8, hipMemcpyDeviceToHost, r.copy_stream);
    for (int k = 0; k < 3 && e == hipSuccess; ++k)
        if (io.hint[k]) {
            e = hipMemcpyAsync(dst, io.dint[k] + r0, (size_t)rows * 4, hipMemcpyDeviceToHost, r.copy_stream);
            dst += rows * 4;
        }
    if (e == hipSuccess) e = hipEventRecord(r.filled[slot], r.copy_stream);
    return e;
}
static void ring_deliver_block(ray_ctx *ctx, const RowBlockIo &io, int slot, int64_t r0, int64_t rows)
{
    const char *src = static_cast<const char *>(ctx->ring.slot[slot]);
    std::vector<CopyPiece> pieces;
    for (int c = io.c0; c < 9; ++c, src += rows * 8) pieces.push_back(CopyPiece{io.hm + (size_t)c * io.ldh + r0, src, (size_t)rows * 8});
    for (int k = 0; k < 3; ++k)
        if (io.hint[k]) { pieces.push_back(CopyPiece{io.hint[k] + r0, src, (size_t)rows * 4}); src += rows * 4; }
    ctx->ring.pool->run(pieces);
}

int ray_shoot_to_exit(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob,
                      const ray_shoot_params *prm, double *m_host, int64_t ld,
                      int32_t *status, int32_t *nsteps, int32_t *nreflect)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_med(ctx, med)) || (rc = check_params(prm))) return rc;
    if (!m_host) return fail(RAY_ERR_INVALID, "null m");
    const int nsrc = call_sources(ctx, prm);
    const int64_t num = (int64_t)nsrc * prm->nd;
    if (ld < num) return fail(RAY_ERR_INVALID, "ld smaller than the number of rows");
    if (num == 0) return RAY_OK;
    WsBuf dm(ctx, kWsM);
    RayOutBufs ob_(ctx);
    HIPCHK(dm.alloc(sizeof(double) * 9 * (size_t)num));
    if ((rc = ob_.alloc(num))) return rc;
    const int c0 = ctx->skip_init_cols ? 4 : 0;             // RAY_OPT_SKIP_INIT_COLUMNS: columns 5..9 only
    RowBlockIo io{dm.as<double>(), num, m_host, ld, c0,
                  {ob_.st.as<int32_t>(), ob_.ns.as<int32_t>(), ob_.nr.as<int32_t>()}, {status, nsteps, nreflect}};
    // Large sweeps go out in blocks of sources through the page-locked ring: block k's DMA overlaps block k+1's kernel
    // and the host threads' copy of block k-1 into the caller's (pageable) array.  Blocks are whole workgroup rows of
    // sources (64: the XCD arcs of the launch stay intact) and about 24 MiB of results.
    constexpr size_t kSlotTarget = 24u << 20, kPipeMin = 8u << 20;
    int bsrc = (int)std::min<int64_t>(nsrc, std::max<int64_t>(1, (int64_t)(kSlotTarget / io.row_bytes()) / prm->nd));
    if (bsrc >= 64) bsrc &= ~63;
    const int nblk = (nsrc + bsrc - 1) / bsrc;
    const bool piped = (size_t)num * io.row_bytes() >= kPipeMin && nblk >= 2 &&
                       ring_prepare(ctx, (size_t)bsrc * prm->nd * io.row_bytes(), nblk);
    if (!piped) {
        rc = shoot_to_exit_impl(ctx, med, ob, prm, dm.as<double>(), num, ob_.st.as<int32_t>(), ob_.ns.as<int32_t>(), ob_.nr.as<int32_t>());
        if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
        if ((rc = copy_cols_d2h(ctx, m_host, ld, dm.as<double>(), num, num, c0, 9 - c0))) return rc;
        if ((rc = ob_.fetch(ctx, num, status, nsteps, nreflect))) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return RAY_OK;
    }
    InitTables *tab = nullptr;
    if ((rc = init_tables(ctx, prm, &tab))) return rc;
    HostRing &ring = ctx->ring;
    auto bail = [&](int code) {                            // nothing of this call stays in flight behind an error return
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ring.copy_stream);
        return code;
    };
    for (int k = 0; k < nblk; ++k) {                       // every block's kernel, back to back
        const int i0 = k * bsrc, cnt = std::min(bsrc, nsrc - i0);
        if ((rc = shoot_block(ctx, med, ob, prm, tab, dm.as<double>(), num, i0, cnt, ob_.st.as<int32_t>(), ob_.ns.as<int32_t>(), ob_.nr.as<int32_t>())))
            return bail(rc);
        if (hipEventRecord(ring.blk_done[(size_t)k], ctx->stream) != hipSuccess) return bail(fail(RAY_ERR_HIP, "ray_shoot_to_exit: hipEventRecord failed"));
    }
    for (int k = 0; k <= nblk; ++k) {
        if (k < nblk) {                                    // DMA of block k into slot k % 2 (block k - 2 has left it: delivered below, one turn ago)
            const int i0 = k * bsrc, cnt = std::min(bsrc, nsrc - i0);
            hipError_t e = hipStreamWaitEvent(ring.copy_stream, ring.blk_done[(size_t)k], 0);
            if (e == hipSuccess) e = ring_enqueue_block(ctx, io, k % HostRing::kSlots, (int64_t)i0 * prm->nd, (int64_t)cnt * prm->nd);
            if (e != hipSuccess) return bail(fail(RAY_ERR_HIP, std::string("ray_shoot_to_exit (ring): ") + hipGetErrorString(e)));
        }
        if (k >= 1) {                                      // block k - 1: wait for its DMA, copy it out while block k's DMA runs
            const int j = k - 1, i0 = j * bsrc, cnt = std::min(bsrc, nsrc - i0);
            if (hipEventSynchronize(ring.filled[j % HostRing::kSlots]) != hipSuccess) return bail(fail(RAY_ERR_HIP, "ray_shoot_to_exit (ring): hipEventSynchronize failed"));
            ring_deliver_block(ctx, io, j % HostRing::kSlots, (int64_t)i0 * prm->nd, (int64_t)cnt * prm->nd);
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RAY_OK;
}

int ray_shoot_states_dev(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob, double dt,
                         int32_t time_unit, int32_t max_steps, double *m_dev, int64_t num,
                         int64_t ld, int32_t *status_dev, int32_t *nsteps_dev, int32_t *nreflect_dev)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_med(ctx, med))) return rc;
    if (!m_dev || num < 0 || ld < num) return fail(RAY_ERR_INVALID, "bad m / num / ld");
    if (!(dt > 0)) return fail(RAY_ERR_INVALID, "dt must be > 0");
    if (time_unit != RAY_TIME_SECONDS && time_unit != RAY_TIME_STEPS) return fail(RAY_ERR_INVALID, "bad time_unit");
    ShootArgs a{};
    a.init = 0; a.nd = 1;
    a.dt = dt; a.time_unit = time_unit; a.max_steps = max_steps;
    a.m = m_dev; a.num = num; a.ld = ld;
    a.status = status_dev; a.nsteps = nsteps_dev; a.nreflect = nreflect_dev;
    return launch_shoot(ctx, med, ob, a);
}

int ray_shoot_states(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob, double dt,
                     int32_t time_unit, int32_t max_steps, double *m_host, int64_t num, int64_t ld,
                     int32_t *status, int32_t *nsteps, int32_t *nreflect)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!m_host || num < 0 || ld < num) return fail(RAY_ERR_INVALID, "bad m / num / ld");
    if (num == 0) return RAY_OK;
    WsBuf dm(ctx, kWsM);
    RayOutBufs ob_(ctx);
    HIPCHK(dm.alloc(sizeof(double) * 9 * (size_t)num));
    if ((rc = ob_.alloc(num))) return rc;
    if ((rc = copy_cols_h2d(ctx, dm.as<double>(), num, m_host, ld, num, 0, 4))) return rc;
    rc = ray_shoot_states_dev(ctx, med, ob, dt, time_unit, max_steps, dm.as<double>(), num, num,
                              ob_.st.as<int32_t>(), ob_.ns.as<int32_t>(), ob_.nr.as<int32_t>());
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    if ((rc = copy_cols_d2h(ctx, m_host, ld, dm.as<double>(), num, num, 4, 5))) return rc;
    if ((rc = ob_.fetch(ctx, num, status, nsteps, nreflect))) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RAY_OK;
}

// ---- ray paths (NEXT-4: what the reference's *Plot tracers collect, Obstacle.jl:98-221) ------------------
int ray_shoot_paths(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob, double dt, int32_t time_unit,
                    int32_t max_steps, double *m_host, int64_t num, int64_t ld, int32_t max_pts,
                    double *path_host, int32_t *npts, int32_t *status, int32_t *nsteps, int32_t *nreflect)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_med(ctx, med))) return rc;
    if (!m_host || !path_host || !npts || num < 0 || ld < num || max_pts < 1) return fail(RAY_ERR_INVALID, "bad m / path / npts / num / ld / max_pts");
    if (!(dt > 0)) return fail(RAY_ERR_INVALID, "dt must be > 0");
    if (time_unit != RAY_TIME_SECONDS && time_unit != RAY_TIME_STEPS) return fail(RAY_ERR_INVALID, "bad time_unit");
    if (num == 0) return RAY_OK;
    const size_t pbytes = sizeof(double) * 2 * (size_t)max_pts * (size_t)num;
    WsBuf dm(ctx, kWsM), dpath(ctx, kWsPath), dnp(ctx, kWsNpts);
    RayOutBufs ob_(ctx);
    HIPCHK(dm.alloc(sizeof(double) * 9 * (size_t)num));
    HIPCHK(dpath.alloc(pbytes));
    HIPCHK(dnp.alloc(sizeof(int32_t) * (size_t)num));
    if ((rc = ob_.alloc(num))) return rc;
    HIPCHK(hipMemsetAsync(dpath.p, 0, pbytes, ctx->stream));
    if ((rc = copy_cols_h2d(ctx, dm.as<double>(), num, m_host, ld, num, 0, 4))) return rc;
    ShootArgs a{};
    a.init = 0; a.nd = 1;
    a.dt = dt; a.time_unit = time_unit; a.max_steps = max_steps;
    a.m = dm.as<double>(); a.num = num; a.ld = num;
    a.status = ob_.st.as<int32_t>(); a.nsteps = ob_.ns.as<int32_t>(); a.nreflect = ob_.nr.as<int32_t>();
    a.path = dpath.as<double>(); a.npts = dnp.as<int32_t>(); a.max_pts = max_pts;
    rc = launch_shoot(ctx, med, ob, a);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    if ((rc = copy_cols_d2h(ctx, m_host, ld, dm.as<double>(), num, num, 4, 5))) return rc;
    HIPCHK(hipMemcpyAsync(path_host, dpath.p, pbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(npts, dnp.p, sizeof(int32_t) * (size_t)num, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = ob_.fetch(ctx, num, status, nsteps, nreflect))) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RAY_OK;
}

// ---- fixed-time trace -------------------------------------------------------------------------------
// The rows of a host `m` as fans: nd when they are runs of nd rows with one starting point each (what ScatterRelation
// returns and the reference's callers pass on: NonObstacle.jl:63, Obstacle.jl:275), else 0.  Looks at the first run and
// at the two ends of every other one: the answer only chooses the trace kernel's work mapping, never its results.
static int detect_fan(const double *m, int64_t num, int64_t ld)
{
    const double *x = m, *y = m + ld;
    int64_t nd = 1;
    while (nd < num && x[nd] == x[0] && y[nd] == y[0]) ++nd;
    if (nd < 2 || nd > INT32_MAX || num % nd) return 0;
    for (int64_t f = nd; f < num; f += nd) {
        const int64_t l = f + nd - 1;
        if (x[f] != x[l] || y[f] != y[l] || (x[f] == x[f - 1] && y[f] == y[f - 1])) return 0;
    }
    return (int)nd;
}

// fan: rows per fan for k_shoot's work mapping (RAY_OPT_FAN_DIRECTIONS or detect_fan), 0 = one lane per row in order
static int trace_dev_impl(ray_ctx *ctx, const ray_medium *med, const double *m_dev, int64_t num,
                          int64_t ld, double dt, int32_t mode, double *X_dev, int64_t ldx,
                          int32_t *status_dev, int32_t *nsteps_dev, double *path_dev, int32_t *npts_dev, int max_pts,
                          int fan)
{
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_med(ctx, med))) return rc;
    if (med->kind != kGrid) return fail(RAY_ERR_UNSUPPORTED, "fixed-time trace is defined on gridded media only (Obstacle.jl:275, NonObstacle.jl:141)");
    if (!m_dev || !X_dev || num < 0 || ld < num || ldx < num) return fail(RAY_ERR_INVALID, "bad m / X / num / ld");
    if (!(dt > 0)) return fail(RAY_ERR_INVALID, "dt must be > 0");
    if (mode != RAY_TRACE_SHORTEN_CEASE && mode != RAY_TRACE_OVERSHOOT) return fail(RAY_ERR_INVALID, "bad mode");
    if (num == 0) return RAY_OK;
    TraceArgs a{};
    a.med = medium_dev(med);
    a.m = m_dev; a.num = num; a.ld = ld; a.dt = dt; a.mode = mode;
    a.cease_r = 1 + 2 * (med->p[1] - med->p[0]);
    a.cease_s2 = sqrt_le_threshold(a.cease_r);
    a.max_steps = 10000000;
    a.X = X_dev; a.ldx = ldx; a.status = status_dev; a.nsteps = nsteps_dev;
    a.step_counter = ctx->d_steps;
    unsigned nblk = (unsigned)((num + block_threads(ctx) - 1) / block_threads(ctx));
    if (fan > 0 && num % fan == 0) {                        // the rows are fans of `fan` directions: k_shoot's mapping
        const int wpb = block_threads(ctx) / 64;
        a.nd = fan;
        a.nsrc = (int)(num / fan);
        const int dshift = pick_dshift(a.nd, a.nsrc, wpb);
        a.dshift = dshift;
        const int spb = wpb << (6 - dshift), dpw = 1 << dshift;
        a.ngroup = (a.nsrc + spb - 1) / spb;
        nblk = (unsigned)(((a.nd + dpw - 1) / dpw) * a.ngroup);
    }
    hipEvent_t e0, e1;
    time_begin(ctx, e0, e1);
    const dim3 tgrid(nblk), tblock(block_threads(ctx));
    if (path_dev) {
        a.path = path_dev; a.npts = npts_dev; a.max_pts = max_pts;
        if (ctx->arith == RAY_ARITH_FUSED) hipLaunchKernelGGL((k_trace<kFused, 1>), tgrid, tblock, nodes_lds_bytes(med, true), ctx->stream, a);
        else hipLaunchKernelGGL((k_trace<kExact, 1>), tgrid, tblock, nodes_lds_bytes(med, false), ctx->stream, a);
    } else if (ctx->arith == RAY_ARITH_FUSED) hipLaunchKernelGGL((k_trace<kFused, 0>), tgrid, tblock, nodes_lds_bytes(med, true), ctx->stream, a);
    else hipLaunchKernelGGL((k_trace<kExact, 0>), tgrid, tblock, nodes_lds_bytes(med, false), ctx->stream, a);
    time_end(ctx, e0, e1);
    HIPCHK(hipGetLastError());
    ctx->rays += (uint64_t)num;
    ctx->launches += 1;
    return RAY_OK;
}

int ray_trace_fixed_time_dev(ray_ctx *ctx, const ray_medium *med, const double *m_dev, int64_t num,
                             int64_t ld, double dt, int32_t mode, double *X_dev, int64_t ldx,
                             int32_t *status_dev, int32_t *nsteps_dev)
{
    return trace_dev_impl(ctx, med, m_dev, num, ld, dt, mode, X_dev, ldx, status_dev, nsteps_dev, nullptr, nullptr, 0,
                          ctx ? ctx->fan_nd : 0);
}

int ray_trace_fixed_time(ray_ctx *ctx, const ray_medium *med, const double *m_host, int64_t num,
                         int64_t ld, double dt, int32_t mode, double *X_host, int64_t ldx,
                         int32_t *status, int32_t *nsteps)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!m_host || !X_host || num < 0 || ld < num || ldx < num) return fail(RAY_ERR_INVALID, "bad m / X / num / ld");
    if (num == 0) return RAY_OK;
    WsBuf dm(ctx, kWsM), dX(ctx, kWsX);
    RayOutBufs ob_(ctx);
    HIPCHK(dm.alloc(sizeof(double) * 9 * (size_t)num));
    HIPCHK(dX.alloc(sizeof(double) * 4 * (size_t)num));
    if ((rc = ob_.alloc(num))) return rc;
    if ((rc = copy_cols_h2d(ctx, dm.as<double>(), num, m_host, ld, num, 0, 4))) return rc;
    if ((rc = copy_cols_h2d(ctx, dm.as<double>(), num, m_host, ld, num, 8, 1))) return rc;
    rc = trace_dev_impl(ctx, med, dm.as<double>(), num, num, dt, mode, dX.as<double>(), num,
                        ob_.st.as<int32_t>(), ob_.ns.as<int32_t>(), nullptr, nullptr, 0,
                        ctx->fan_nd > 0 ? ctx->fan_nd : detect_fan(m_host, num, ld));
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    if ((rc = copy_cols_d2h(ctx, X_host, ldx, dX.as<double>(), num, num, 0, 4))) return rc;
    if ((rc = ob_.fetch(ctx, num, status, nsteps, nullptr))) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RAY_OK;
}

// NonReflectionPlot's data (Obstacle.jl:222-274 without the plotting): the overshoot trace plus polylines
int ray_trace_paths(ray_ctx *ctx, const ray_medium *med, const double *m_host, int64_t num, int64_t ld,
                    double dt, int32_t mode, int32_t max_pts, double *path_host, int32_t *npts,
                    double *X_host, int64_t ldx, int32_t *status, int32_t *nsteps)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!m_host || !X_host || !path_host || !npts || num < 0 || ld < num || ldx < num || max_pts < 1)
        return fail(RAY_ERR_INVALID, "bad m / X / path / npts / num / ld / max_pts");
    if (num == 0) return RAY_OK;
    const size_t pbytes = sizeof(double) * 2 * (size_t)max_pts * (size_t)num;
    WsBuf dm(ctx, kWsM), dX(ctx, kWsX), dpath(ctx, kWsPath), dnp(ctx, kWsNpts);
    RayOutBufs ob_(ctx);
    HIPCHK(dm.alloc(sizeof(double) * 9 * (size_t)num));
    HIPCHK(dX.alloc(sizeof(double) * 4 * (size_t)num));
    HIPCHK(dpath.alloc(pbytes));
    HIPCHK(dnp.alloc(sizeof(int32_t) * (size_t)num));
    if ((rc = ob_.alloc(num))) return rc;
    HIPCHK(hipMemsetAsync(dpath.p, 0, pbytes, ctx->stream));
    HIPCHK(hipMemsetAsync(dnp.p, 0, sizeof(int32_t) * (size_t)num, ctx->stream));
    if ((rc = copy_cols_h2d(ctx, dm.as<double>(), num, m_host, ld, num, 0, 4))) return rc;
    if ((rc = copy_cols_h2d(ctx, dm.as<double>(), num, m_host, ld, num, 8, 1))) return rc;
    rc = trace_dev_impl(ctx, med, dm.as<double>(), num, num, dt, mode, dX.as<double>(), num,
                        ob_.st.as<int32_t>(), ob_.ns.as<int32_t>(), dpath.as<double>(), dnp.as<int32_t>(), max_pts,
                        ctx->fan_nd > 0 ? ctx->fan_nd : detect_fan(m_host, num, ld));
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    if ((rc = copy_cols_d2h(ctx, X_host, ldx, dX.as<double>(), num, num, 0, 4))) return rc;
    HIPCHK(hipMemcpyAsync(path_host, dpath.p, pbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(npts, dnp.p, sizeof(int32_t) * (size_t)num, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = ob_.fetch(ctx, num, status, nsteps, nullptr))) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RAY_OK;
}

// ---- complete forward operator: X, rho, Frechet matrix as device block rows (NEXT-1) -------------------
enum { kFmIndptr = 0, kFmIndices, kFmData, kFmUoff, kFmUcols, kFmUvals, kFmRowcnt, kFmBstart, kFmBcols, kFmBvals, kFmY, kFmArrays };
static_assert(kFmArrays == ray_ctx::kPool, "pool slots");
struct ray_frechet {
    ray_ctx *ctx = nullptr;
    int64_t num = 0, nnz = 0, U = 0;    // rays, entries that survive droptol, distinct (ray, node) pairs
    int N = 0;
    double droptol = -1;
    // device arrays, see kFm*: [indptr 4 num + 1 | indices nnz | data nnz] (indices / data only once an export
    // asked for them) and the block rows [uoff num + 1 | ucols U | uvals 4 U | rowcnt 4 num]
    void *arr[kFmArrays] = {};
    size_t cap[kFmArrays] = {};         // bytes behind them
    bool csr_built = false;
    bool bands_built = false;           // the band-major copy the scatter passes stream through (ray_frechet.h), made on first use
    int nbands() const { return (int)(((int64_t)N * N + kBandNodes - 1) / kBandNodes); }
    int64_t *bstart() { return static_cast<int64_t *>(arr[kFmBstart]); }
    double *ywork() { return static_cast<double *>(arr[kFmY]); }
    BlockRows band_rows() { return BlockRows{nullptr, static_cast<const int32_t *>(arr[kFmBcols]), static_cast<const double *>(arr[kFmBvals]), num, droptol}; }
    int64_t *indptr() { return static_cast<int64_t *>(arr[kFmIndptr]); }
    int32_t *indices() { return static_cast<int32_t *>(arr[kFmIndices]); }
    double *data() { return static_cast<double *>(arr[kFmData]); }
    int64_t *uoff() { return static_cast<int64_t *>(arr[kFmUoff]); }
    int32_t *ucols() { return static_cast<int32_t *>(arr[kFmUcols]); }
    double *uvals() { return static_cast<double *>(arr[kFmUvals]); }
    int32_t *rowcnt() { return static_cast<int32_t *>(arr[kFmRowcnt]); }
    BlockRows rows() { return BlockRows{uoff(), ucols(), uvals(), num, droptol}; }
};

// array `which` of f (size known): the context's parked one when it fits, else fresh memory
static hipError_t frechet_alloc(ray_frechet *f, int which, size_t bytes)
{
    if (f->arr[which] && f->cap[which] >= bytes) return hipSuccess;
    if (f->arr[which]) { (void)hipFree(f->arr[which]); f->arr[which] = nullptr; f->cap[which] = 0; }
    void *p = nullptr;
    size_t have = f->ctx->csr_pool[which] && f->ctx->csr_cap[which] >= bytes ? f->ctx->csr_cap[which] : (bytes ? bytes : 8);
    hipError_t e = csr_take(f->ctx, which, bytes, &p);
    if (e != hipSuccess) return e;
    f->cap[which] = have;
    f->arr[which] = p;
    return hipSuccess;
}

static void detach_frechet(ray_frechet *f)
{
    for (int k = 0; k < kFmArrays; ++k) {
        if (f->arr[k]) (void)hipFree(f->arr[k]);
        f->arr[k] = nullptr;
    }
    f->ctx = nullptr;
}

int ray_frechet_destroy(ray_frechet *f)
{
    if (!f) return RAY_OK;
    if (f->ctx) forget(f->ctx->frechets, f);
    if (f->ctx) {
        (void)hipSetDevice(f->ctx->device);
        (void)hipStreamSynchronize(f->ctx->stream);             // nothing of this handle is in flight after this
    }
    for (int k = 0; k < kFmArrays; ++k) {
        if (!f->arr[k]) continue;
        // park the larger of (this array, what is parked); free the other
        if (f->ctx && f->cap[k] > f->ctx->csr_cap[k]) {
            if (f->ctx->csr_pool[k]) (void)hipFree(f->ctx->csr_pool[k]);
            f->ctx->csr_pool[k] = f->arr[k];
            f->ctx->csr_cap[k] = f->cap[k];
        } else (void)hipFree(f->arr[k]);
    }
    delete f;
    return RAY_OK;
}

// out[0..n] = exclusive prefix sums of in[0..n) on the context's stream; *total_host after a sync
static int scan_counts(ray_ctx *ctx, const int32_t *in, int64_t n, int64_t *out, int64_t *total_host)
{
    const int64_t per = 256 * kScanItems;
    const int64_t nb = (n + per - 1) / per;
    DevBuf bs, tot;
    HIPCHK(bs.alloc(sizeof(int64_t) * (size_t)nb));
    HIPCHK(tot.alloc(sizeof(int64_t)));
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(256), 0, ctx->stream, in, n, bs.as<int64_t>());
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, ctx->stream, bs.as<int64_t>(), nb, tot.as<int64_t>());
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, ctx->stream, in, n, bs.as<int64_t>(),
                       tot.as<int64_t>(), out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(total_host, tot.p, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RAY_OK;
}

static int forward_operator_impl(ray_ctx *ctx, ray_medium *med, const double *m_host, int64_t num, int64_t ld, double dt,
                                 double droptol, double *X_host, int64_t ldx, double *rho_host, int32_t *status,
                                 int32_t *nsteps, ray_frechet *f)
{
    const int max_steps = 1000000;
    // Slab sizing on the host: a ray takes at most floor(T/dt) + 2 steps (NonObstacle.jl:152-157).
    // A wave steps until its longest ray is done, so the rays are regrouped before they are dealt to waves: by
    // step-count class (1/32 of the longest), longest class first, ray order kept inside a class (a class holds
    // runs of adjacent directions of a source: the wave stays a coherent fan).  64 consecutive rays of a 512-direction
    // fan span chords from grazing to diameter: 83 % live lanes in ray order; the classes give >= 95 %, and the
    // waves come out longest first, which is what fills the chip to the end of the launch.
    const int64_t nwave = (num + 63) / 64;
    std::vector<int32_t> nst((size_t)num);
    int cap_max = 0;
    for (int64_t r = 0; r < num; ++r) {
        const double T = m_host[r + 8 * ld];
        double n = T > 0 ? std::floor(T / dt) + 2 : 0;
        if (n > max_steps) n = max_steps;
        nst[(size_t)r] = (int32_t)n;
        cap_max = std::max(cap_max, (int)n);
    }
    const int cls_w = std::max(1, (cap_max + 31) / 32);
    std::vector<int32_t> perm((size_t)num);
    {   // counting sort by class, descending, stable
        std::vector<int64_t> start(34, 0);
        for (int64_t r = 0; r < num; ++r) ++start[(size_t)(32 - std::min(32, nst[(size_t)r] / cls_w)) + 1];
        for (int k = 1; k < 34; ++k) start[(size_t)k] += start[(size_t)k - 1];
        for (int64_t r = 0; r < num; ++r) {
            const int64_t s = start[(size_t)(32 - std::min(32, nst[(size_t)r] / cls_w))]++;
            perm[(size_t)s] = (int32_t)r;
        }
    }
    std::vector<int32_t> cap((size_t)nwave, 0);
    std::vector<int64_t> wave_base((size_t)nwave, 0);
    int64_t rows = 0;
    for (int64_t w = 0; w < nwave; ++w) {
        int n = 0;
        for (int64_t s = w * 64; s < num && s < (w + 1) * 64; ++s) n = std::max(n, (int)nst[(size_t)perm[(size_t)s]]);
        cap[(size_t)w] = n;
        wave_base[(size_t)w] = rows;
        rows += frechet_rows(n);
    }
    if (frechet_entries(cap_max) > kMergeMaxEntries)
        return fail(RAY_ERR_UNSUPPORTED, "ScatterForwardOperator: more than 1364 steps per ray (T/dt); the per-ray merge holds 16384 stencil entries");
    const size_t slots = (size_t)rows * 64;                  // trajectory records of this call
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    if (slots * (kTrajBytesPerRow + 12 * 36) + (size_t)num * 512 > free_b + ctx->ws_cap[0] + ctx->ws_cap[1] + ctx->ws_cap[9])
        return fail(RAY_ERR_NOMEM, "ScatterForwardOperator: the trajectory slab does not fit in device memory (sum of T/dt too large); split the rays");
    WsBuf dm(ctx, 2), dX(ctx, 3), drho(ctx, 4), dbase(ctx, 5), dcap(ctx, 6), ducnt(ctx, 7), dperm(ctx, 10);
    RayOutBufs ob_(ctx);
    void *ptraj = nullptr, *psrt = nullptr, *pcells = nullptr;
    HIPCHK(ctx_workspace(ctx, 0, (slots ? slots : 1) * kTrajDoubles * sizeof(double), &ptraj));
    HIPCHK(ctx_workspace(ctx, 1, (slots ? slots : 1) * kFrSlotsPerCall * sizeof(uint32_t), &psrt));
    HIPCHK(ctx_workspace(ctx, 9, (slots ? slots : 1) * sizeof(int32_t), &pcells));
    BorrowedBuf dtraj{ptraj}, dsrt{psrt}, dcells{pcells};
    HIPCHK(dm.alloc(sizeof(double) * 9 * (size_t)num));
    HIPCHK(dX.alloc(sizeof(double) * 4 * (size_t)num));
    HIPCHK(drho.alloc(sizeof(double) * 16 * (size_t)num));
    HIPCHK(dbase.alloc(sizeof(int64_t) * (size_t)nwave));
    HIPCHK(dcap.alloc(sizeof(int32_t) * (size_t)nwave));
    HIPCHK(dperm.alloc(sizeof(int32_t) * (size_t)num));
    HIPCHK(ducnt.alloc(sizeof(int32_t) * (size_t)num));
    HIPCHK(frechet_alloc(f, kFmUoff, sizeof(int64_t) * (size_t)(num + 1)));
    HIPCHK(frechet_alloc(f, kFmRowcnt, sizeof(int32_t) * 4 * (size_t)num));
    int rc;
    if ((rc = ob_.alloc(num)) || (rc = ensure_hess(med))) return rc;
    if ((rc = copy_cols_h2d(ctx, dm.as<double>(), num, m_host, ld, num, 0, 4))) return rc;
    if ((rc = copy_cols_h2d(ctx, dm.as<double>(), num, m_host, ld, num, 8, 1))) return rc;
    HIPCHK(hipMemcpyAsync(dbase.p, wave_base.data(), sizeof(int64_t) * (size_t)nwave, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dcap.p, cap.data(), sizeof(int32_t) * (size_t)nwave, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dperm.p, perm.data(), sizeof(int32_t) * (size_t)num, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));           // the host vectors above are read by those copies

    FrechetArgs a{};
    a.med = medium_dev(med);
    a.m = dm.as<double>(); a.num = num; a.ld = num; a.dt = dt; a.max_steps = max_steps;
    a.cease_r = 1 + 2 * (med->p[1] - med->p[0]);
    a.cease_s2 = sqrt_le_threshold(a.cease_r);
    a.wave_base = dbase.as<int64_t>(); a.wave_cap = dcap.as<int32_t>(); a.perm = dperm.as<int32_t>();
    a.traj = dtraj.as<double>(); a.cells = dcells.as<int32_t>();
    a.X = dX.as<double>(); a.ldx = num; a.rho = drho.as<double>();
    a.status = ob_.st.as<int32_t>(); a.nsteps = ob_.ns.as<int32_t>();
    a.step_counter = ctx->d_steps;
    const size_t lds = frechet_lds_bytes(med->N, ctx->arith != RAY_ARITH_FUSED);
    TimedSpan span(ctx);
    if (ctx->arith == RAY_ARITH_FUSED) hipLaunchKernelGGL(k_frechet<kFused>, dim3((unsigned)nwave), dim3(64), lds, ctx->stream, a);
    else hipLaunchKernelGGL(k_frechet<kExact>, dim3((unsigned)nwave), dim3(64), lds, ctx->stream, a);
    HIPCHK(hipGetLastError());
    // distinct nodes per ray: the records' stencil segments sorted by first node (ray_frechet.h); RAY_OPT_TUNE_MERGE_ENTRIES = 1
    // selects the older form that sorts the twelve entries of every record (kept for A/B and as a cross-check in the tests)
    const bool by_entries = ctx->tune_merge_entries != 0;
    int n2 = 256;
    while (n2 < (by_entries ? frechet_entries(cap_max) : frechet_segments(cap_max))) n2 <<= 1;
    // 32-bit sort keys when first node, id and (segments) length fit together; RAY_OPT_TUNE_SORT_KEY_BITS = 64 forces the wide ones
    int eshift = 0;
    while ((1 << eshift) < n2) ++eshift;
    const bool env_wide = ctx->tune_sort_key == 64;                      // RAY_OPT_TUNE_SORT_KEY_BITS
    const int idbits = eshift + (by_entries ? 0 : 2);
    const bool narrow_key = !env_wide && idbits < 31 && (uint64_t)med->N * (uint64_t)med->N < (1ull << (32 - idbits)) - 1;
    const size_t sort_lds = merge_sort_lds_bytes(n2, narrow_key ? 4 : 8) + (by_entries ? 0 : 2 * (size_t)n2);
#define RAY_LAUNCH_SORT(KERNEL)                                                                                                        \
    do {                                                                                                                               \
        if (sort_lds > 48 * 1024)                                                                                                      \
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds)); \
        hipLaunchKernelGGL((KERNEL), dim3((unsigned)num), dim3(kSortThreads), sort_lds, ctx->stream, dcells.as<int32_t>(),             \
                           dsrt.as<uint32_t>(), dbase.as<int64_t>(), dcap.as<int32_t>(), dperm.as<int32_t>(), ob_.ns.as<int32_t>(),    \
                           ducnt.as<int32_t>(), eshift, med->N);                                                                      \
    } while (0)
    if (by_entries) {
        if (narrow_key) RAY_LAUNCH_SORT((k_merge_sort<kSortThreads, uint32_t>));
        else RAY_LAUNCH_SORT((k_merge_sort<kSortThreads, uint64_t>));
    } else {
        if (narrow_key) RAY_LAUNCH_SORT((k_seg_sort<kSortThreads, uint32_t>));
        else RAY_LAUNCH_SORT((k_seg_sort<kSortThreads, uint64_t>));
    }
#undef RAY_LAUNCH_SORT
    HIPCHK(hipGetLastError());
    if ((rc = scan_counts(ctx, ducnt.as<int32_t>(), num, f->uoff(), &f->U))) return rc;
    HIPCHK(frechet_alloc(f, kFmUcols, sizeof(int32_t) * (size_t)(f->U ? f->U : 1)));
    HIPCHK(frechet_alloc(f, kFmUvals, sizeof(double) * 4 * (size_t)(f->U ? f->U : 1)));
    // the ray's records sit in LDS during the merge when they fit (224 rows of 144 B; RAY_OPT_TUNE_FRECHET_LDS_ROWS for the tests)
    int lds_rows = std::min(frechet_rows(cap_max), kMergeLdsMaxRows);
    if (ctx->tune_lds_rows >= 0) lds_rows = std::min(lds_rows, ctx->tune_lds_rows);
    // one workgroup per ray; RAY_OPT_TUNE_MERGE_GRID (tests, experiments) makes fewer workgroups loop over the rays instead
    const int64_t emit_grid = ctx->tune_merge_grid > 0 ? std::min<int64_t>(num, ctx->tune_merge_grid) : num;
    if (by_entries) {
        hipLaunchKernelGGL(k_merge_emit, dim3((unsigned)emit_grid), dim3(256), (size_t)lds_rows * kMergeLdsRowBytes, ctx->stream,
                           dsrt.as<uint32_t>(), dtraj.as<double>(), dbase.as<int64_t>(), dcap.as<int32_t>(), dperm.as<int32_t>(), ob_.ns.as<int32_t>(),
                           drho.as<double>(), f->uoff(), droptol, med->N, lds_rows, num, f->ucols(), f->uvals(), f->rowcnt());
    } else {
        const int max_rows = frechet_rows(cap_max);
        // four workgroups per CU (40 KB each) when that costs at most an eighth of the rows kept in LDS
        const int fit4 = (int)((40960 - 64 - (int64_t)max_rows * (kFrSlotsPerCall * 4)) / kSegLdsRowBytes);
        if (ctx->tune_lds_rows < 0 && fit4 >= lds_rows * 7 / 8) lds_rows = std::min(lds_rows, fit4);
        const size_t emit_lds = seg_emit_lds_bytes(lds_rows, max_rows);
        if (emit_lds > 48 * 1024)
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_seg_emit), hipFuncAttributeMaxDynamicSharedMemorySize, (int)emit_lds));
        hipLaunchKernelGGL(k_seg_emit, dim3((unsigned)emit_grid), dim3(256), emit_lds, ctx->stream,
                           dsrt.as<uint32_t>(), dtraj.as<double>(), dbase.as<int64_t>(), dcap.as<int32_t>(), dperm.as<int32_t>(), ob_.ns.as<int32_t>(),
                           drho.as<double>(), f->uoff(), droptol, med->N, lds_rows, max_rows, num, f->ucols(), f->uvals(), f->rowcnt());
    }
    HIPCHK(hipGetLastError());
    HIPCHK(frechet_alloc(f, kFmIndptr, sizeof(int64_t) * (size_t)(4 * num + 1)));
    if ((rc = scan_counts(ctx, f->rowcnt(), 4 * num, f->indptr(), &f->nnz))) return rc;
    span.end();
    ctx->rays += (uint64_t)num;
    ctx->launches += 1;
    if ((rc = copy_cols_d2h(ctx, X_host, ldx, dX.as<double>(), num, num, 0, 4))) return rc;
    if (rho_host) HIPCHK(hipMemcpyAsync(rho_host, drho.p, sizeof(double) * 16 * (size_t)num, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = ob_.fetch(ctx, num, status, nsteps, nullptr))) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RAY_OK;
}

// indices / data of the CSR form, built from the block rows when an export first asks for them
static int ensure_csr(ray_frechet *f)
{
    if (f->csr_built || f->num == 0) return RAY_OK;
    ray_ctx *ctx = f->ctx;
    HIPCHK(frechet_alloc(f, kFmIndices, sizeof(int32_t) * (size_t)(f->nnz ? f->nnz : 1)));
    HIPCHK(frechet_alloc(f, kFmData, sizeof(double) * (size_t)(f->nnz ? f->nnz : 1)));
    hipLaunchKernelGGL(k_csr_emit, dim3((unsigned)f->num), dim3(256), 0, ctx->stream, f->ucols(), f->uvals(),
                       f->uoff(), f->indptr(), f->droptol, f->indices(), f->data());
    HIPCHK(hipGetLastError());
    f->csr_built = true;
    return RAY_OK;
}

int ray_forward_operator(ray_ctx *ctx, ray_medium *med, const double *m_host, int64_t num, int64_t ld, double dt,
                         double droptol, double *X_host, int64_t ldx, double *rho_host, int32_t *status,
                         int32_t *nsteps, ray_frechet **out)
{
    if (!out) return fail(RAY_ERR_INVALID, "null out");
    *out = nullptr;
    int rc = use(ctx);
    if (rc) return rc;
    if ((rc = check_med(ctx, med))) return rc;
    if (med->kind != kGrid) return fail(RAY_ERR_UNSUPPORTED, "ScatterForwardOperator takes a gridded medium (NonObstacle.jl:189)");
    if (med->N < 6) return fail(RAY_ERR_INVALID, "the Frechet path needs N >= 6 (hess table, NonObstacle.jl:212)");
    if (!m_host || !X_host || num < 0 || ld < num || ldx < num) return fail(RAY_ERR_INVALID, "bad m / X / num / ld");
    if (num > 0x7fffffff / 4) return fail(RAY_ERR_INVALID, "too many rays for one call (4 num must fit in int32)");
    if (!(dt > 0)) return fail(RAY_ERR_INVALID, "dt must be > 0");
    ray_frechet *f = new ray_frechet;
    f->ctx = ctx; f->num = num; f->N = med->N; f->droptol = droptol;
    ctx->frechets.push_back(f);
    if (num == 0) {
        if (frechet_alloc(f, kFmIndptr, sizeof(int64_t)) != hipSuccess || frechet_alloc(f, kFmUoff, sizeof(int64_t)) != hipSuccess ||
            hipMemsetAsync(f->indptr(), 0, sizeof(int64_t), ctx->stream) != hipSuccess ||
            hipMemsetAsync(f->uoff(), 0, sizeof(int64_t), ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            ray_frechet_destroy(f);
            return fail(RAY_ERR_HIP, "ray_forward_operator: allocation failed");
        }
        *out = f;
        return RAY_OK;
    }
    // The trajectory slab and the sort's hand-over need 180 B per ray-step: rays go through in chunks (whole
    // waves) that keep them within a budget, and the block rows are concatenated (ray-ordered, chunks contiguous).
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { ray_frechet_destroy(f); return fail(RAY_ERR_HIP, "hipMemGetInfo failed"); }
    const double env_mb = (double)ctx->tune_slab_mb;                      // RAY_OPT_TUNE_FRECHET_SLAB_MB: tests force small chunks with it
    const double budget = env_mb > 0 ? env_mb * 1048576.0 : std::min((double)free_b / 4, 16.0 * 1073741824.0);
    std::vector<int64_t> cuts{0};
    {
        double acc = 0;
        for (int64_t w0 = 0; w0 < num; w0 += 64) {
            double tmax = 0;
            for (int64_t r = w0; r < num && r < w0 + 64; ++r) tmax = std::max(tmax, m_host[r + 8 * ld] > 0 ? m_host[r + 8 * ld] : 0.0);
            const double bytes = tmax > 0 ? (std::floor(tmax / dt) + 3) * 64 * (double)kTrajBytesPerRow : 0.0;
            if (acc > 0 && acc + bytes > budget) { cuts.push_back(w0); acc = 0; }
            acc += bytes;
        }
        cuts.push_back(num);
    }
    if (cuts.size() == 2) {
        rc = forward_operator_impl(ctx, med, m_host, num, ld, dt, droptol, X_host, ldx, rho_host, status, nsteps, f);
        if (rc) {
            (void)hipStreamSynchronize(ctx->stream);
            ray_frechet_destroy(f);
            return rc;
        }
        *out = f;
        return RAY_OK;
    }
    std::vector<ray_frechet *> pieces;
    auto drop = [&]() { for (ray_frechet *q : pieces) ray_frechet_destroy(q); ray_frechet_destroy(f); };
    int64_t nnz = 0, U = 0;
    for (size_t c = 0; c + 1 < cuts.size(); ++c) {
        const int64_t r0 = cuts[c], n = cuts[c + 1] - r0;
        ray_frechet *q = new ray_frechet;
        q->ctx = ctx; q->num = n; q->N = med->N; q->droptol = droptol;
        ctx->frechets.push_back(q);
        pieces.push_back(q);
        rc = forward_operator_impl(ctx, med, m_host + r0, n, ld, dt, droptol, X_host + r0, ldx,
                                   rho_host ? rho_host + 16 * r0 : nullptr, status ? status + r0 : nullptr,
                                   nsteps ? nsteps + r0 : nullptr, q);
        if (rc) { (void)hipStreamSynchronize(ctx->stream); drop(); return rc; }
        nnz += q->nnz;
        U += q->U;
    }
    hipError_t e = frechet_alloc(f, kFmIndptr, sizeof(int64_t) * (size_t)(4 * num + 1));
    if (e == hipSuccess) e = frechet_alloc(f, kFmUoff, sizeof(int64_t) * (size_t)(num + 1));
    if (e == hipSuccess) e = frechet_alloc(f, kFmRowcnt, sizeof(int32_t) * 4 * (size_t)num);
    if (e == hipSuccess) e = frechet_alloc(f, kFmUcols, sizeof(int32_t) * (size_t)(U ? U : 1));
    if (e == hipSuccess) e = frechet_alloc(f, kFmUvals, sizeof(double) * 4 * (size_t)(U ? U : 1));
    int64_t at = 0, atU = 0;
    for (size_t c = 0; e == hipSuccess && c < pieces.size(); ++c) {
        ray_frechet *q = pieces[c];
        const int last = c + 1 == pieces.size() ? 1 : 0;                          // the last piece also brings the end markers
        const int64_t rows = 4 * q->num + last, rays = q->num + last;
        hipLaunchKernelGGL(k_indptr_shift, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream,
                           f->indptr() + 4 * cuts[c], q->indptr(), rows, at);
        hipLaunchKernelGGL(k_indptr_shift, dim3((unsigned)((rays + 255) / 256)), dim3(256), 0, ctx->stream,
                           f->uoff() + cuts[c], q->uoff(), rays, atU);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(f->rowcnt() + 4 * cuts[c], q->rowcnt(), sizeof(int32_t) * 4 * (size_t)q->num, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess && q->U) e = hipMemcpyAsync(f->ucols() + atU, q->ucols(), sizeof(int32_t) * (size_t)q->U, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess && q->U) e = hipMemcpyAsync(f->uvals() + 4 * atU, q->uvals(), sizeof(double) * 4 * (size_t)q->U, hipMemcpyDeviceToDevice, ctx->stream);
        at += q->nnz;
        atU += q->U;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        const std::string msg = std::string("ray_forward_operator (chunk assembly): ") + hipGetErrorString(e);
        drop();
        return fail(e == hipErrorOutOfMemory ? RAY_ERR_NOMEM : RAY_ERR_HIP, msg);
    }
    for (ray_frechet *q : pieces) ray_frechet_destroy(q);
    f->nnz = nnz;
    f->U = U;
    *out = f;
    return RAY_OK;
}

int ray_frechet_shape(ray_frechet *f, int64_t *n_rows, int64_t *n_cols, int64_t *nnz)
{
    if (!f) return fail(RAY_ERR_INVALID, "null handle");
    if (n_rows) *n_rows = 4 * f->num;
    if (n_cols) *n_cols = (int64_t)f->N * f->N;
    if (nnz) *nnz = f->nnz;
    return RAY_OK;
}

int ray_frechet_export_csr(ray_frechet *f, int64_t *indptr, int32_t *indices, double *data)
{
    if (!f || !indptr) return fail(RAY_ERR_INVALID, "null argument");
    if (f->nnz && (!indices || !data)) return fail(RAY_ERR_INVALID, "null indices / data");
    int rc = use(f->ctx);
    if (rc) return rc;
    if ((rc = ensure_csr(f))) return rc;
    HIPCHK(hipMemcpyAsync(indptr, f->indptr(), sizeof(int64_t) * (size_t)(4 * f->num + 1), hipMemcpyDeviceToHost, f->ctx->stream));
    if (f->nnz) {
        HIPCHK(hipMemcpyAsync(indices, f->indices(), sizeof(int32_t) * (size_t)f->nnz, hipMemcpyDeviceToHost, f->ctx->stream));
        HIPCHK(hipMemcpyAsync(data, f->data(), sizeof(double) * (size_t)f->nnz, hipMemcpyDeviceToHost, f->ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(f->ctx->stream));
    return RAY_OK;
}

int ray_frechet_csr_dev(ray_frechet *f, const int64_t **indptr, const int32_t **indices, const double **data)
{
    if (!f) return fail(RAY_ERR_INVALID, "null handle");
    int rc = use(f->ctx);
    if (rc) return rc;
    if ((indices || data) && (rc = ensure_csr(f))) return rc;           // converted on the context's stream
    if (indptr) *indptr = f->indptr();
    if (indices) *indices = f->indices();
    if (data) *data = f->data();
    return RAY_OK;
}

int ray_frechet_block_rows_dev(ray_frechet *f, int64_t *n_entries, const int64_t **uoff, const int32_t **ucols, const double **uvals)
{
    if (!f) return fail(RAY_ERR_INVALID, "null handle");
    if (n_entries) *n_entries = f->U;
    if (uoff) *uoff = f->uoff();
    if (ucols) *ucols = f->ucols();
    if (uvals) *uvals = f->uvals();
    return RAY_OK;
}

// ---- NEXT-2: passes of the reconstruction loop over M (NonObstacle.jl:340-367) ------------------------
static int check_frechet(ray_frechet *f)
{
    if (!f) return fail(RAY_ERR_INVALID, "null handle");
    return use(f->ctx);
}
static unsigned wave_rows_grid(int64_t rows) { return (unsigned)((rows + 3) / 4); }

int ray_frechet_dofs_dev(ray_frechet *f, const double *fidelity_dev, double *dofs_dev)
{
    int rc = check_frechet(f);
    if (rc) return rc;
    if (!fidelity_dev || !dofs_dev) return fail(RAY_ERR_INVALID, "null argument");
    if (f->num == 0) return RAY_OK;
    hipLaunchKernelGGL(k_br_dofs, dim3(wave_rows_grid(f->num)), dim3(256), 0, f->ctx->stream, f->rows(), fidelity_dev, dofs_dev);
    HIPCHK(hipGetLastError());
    return RAY_OK;
}

int ray_frechet_matvec_dev(ray_frechet *f, const double *x_dev, double *y_dev)
{
    int rc = check_frechet(f);
    if (rc) return rc;
    if (!x_dev || !y_dev) return fail(RAY_ERR_INVALID, "null argument");
    if (f->num == 0) return RAY_OK;
    hipLaunchKernelGGL(k_br_matvec, dim3(wave_rows_grid(f->num)), dim3(256), 0, f->ctx->stream, f->rows(), x_dev, y_dev);
    HIPCHK(hipGetLastError());
    return RAY_OK;
}

// the band-major copy of the scatter passes (ray_frechet.h): built on the first pass that needs it, kept with the matrix
static int ensure_bands(ray_frechet *f)
{
    if (f->bands_built || f->num == 0) return RAY_OK;
    ray_ctx *ctx = f->ctx;
    const int nb = f->nbands();
    const int64_t cells = (int64_t)nb * f->num;
    DevBuf boff, blen;
    HIPCHK(boff.alloc(sizeof(int32_t) * (size_t)(cells + f->num)));
    HIPCHK(blen.alloc(sizeof(int32_t) * (size_t)cells));
    HIPCHK(frechet_alloc(f, kFmBstart, sizeof(int64_t) * (size_t)(cells + 1)));
    HIPCHK(frechet_alloc(f, kFmBcols, sizeof(int32_t) * (size_t)(f->U ? f->U : 1)));
    HIPCHK(frechet_alloc(f, kFmBvals, sizeof(double) * 4 * (size_t)(f->U ? f->U : 1)));
    HIPCHK(frechet_alloc(f, kFmY, sizeof(double) * 4 * (size_t)f->num));
    hipLaunchKernelGGL(k_br_band_offsets, dim3(wave_rows_grid(f->num)), dim3(256), 0, ctx->stream, f->rows(), nb, boff.as<int32_t>());
    hipLaunchKernelGGL(k_br_band_lengths, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream, boff.as<int32_t>(), cells, f->num, blen.as<int32_t>());
    HIPCHK(hipGetLastError());
    int64_t total = 0;
    int rc = scan_counts(ctx, blen.as<int32_t>(), cells, f->bstart(), &total);
    if (rc) return rc;
    if (total != f->U) return fail(RAY_ERR_HIP, "band copy: the pieces do not add up to the block rows (columns of a ray not sorted?)");
    hipLaunchKernelGGL(k_br_band_copy, dim3(wave_rows_grid(f->num)), dim3(256), 0, ctx->stream, f->rows(), boff.as<int32_t>(), f->bstart(), nb,
                       static_cast<int32_t *>(f->arr[kFmBcols]), static_cast<double *>(f->arr[kFmBvals]));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));          // boff / blen are released on return
    f->bands_built = true;
    return RAY_OK;
}

static int band_scatter(int mode, ray_frechet *f, const double *coef, const uint8_t *ray_mask_dev, double *out_dev)
{
    const size_t nn = (size_t)f->N * f->N;
    HIPCHK(hipMemsetAsync(out_dev, 0, sizeof(double) * nn, f->ctx->stream));
    if (f->num == 0) return RAY_OK;
    int rc = ensure_bands(f);
    if (rc) return rc;
    const size_t lds = sizeof(double) * kBandNodes;   // 64 KB of dynamic LDS: above the default limit of a launch
    const void *fn = mode == 0 ? reinterpret_cast<const void *>(&k_br_band_scatter<0>) : reinterpret_cast<const void *>(&k_br_band_scatter<1>);
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid((unsigned)((f->num + kBandRays - 1) / kBandRays), (unsigned)f->nbands());
    if (mode == 0)
        hipLaunchKernelGGL(k_br_band_scatter<0>, grid, dim3(kBandThreads), lds, f->ctx->stream, f->band_rows(), f->bstart(), coef, ray_mask_dev, (int64_t)nn, out_dev);
    else
        hipLaunchKernelGGL(k_br_band_scatter<1>, grid, dim3(kBandThreads), lds, f->ctx->stream, f->band_rows(), f->bstart(), coef, ray_mask_dev, (int64_t)nn, out_dev);
    HIPCHK(hipGetLastError());
    return RAY_OK;
}

int ray_frechet_matvec_t_dev(ray_frechet *f, const double *v_dev, const uint8_t *ray_mask_dev, double *b_dev)
{
    int rc = check_frechet(f);
    if (rc) return rc;
    if (!v_dev || !b_dev) return fail(RAY_ERR_INVALID, "null argument");
    return band_scatter(0, f, v_dev, ray_mask_dev, b_dev);
}

int ray_frechet_normal_matvec_dev(ray_frechet *f, const double *x_dev, const uint8_t *ray_mask_dev, double *w_dev)
{
    int rc = check_frechet(f);
    if (rc) return rc;
    if (!x_dev || !w_dev || x_dev == w_dev) return fail(RAY_ERR_INVALID, "null argument / x and w must differ");
    if (f->num) {
        if ((rc = ensure_bands(f))) return rc;
        hipLaunchKernelGGL(k_br_matvec, dim3(wave_rows_grid(f->num)), dim3(256), 0, f->ctx->stream, f->rows(), x_dev, f->ywork());
        HIPCHK(hipGetLastError());
    }
    return band_scatter(0, f, f->ywork(), ray_mask_dev, w_dev);
}

int ray_frechet_normal_diag_dev(ray_frechet *f, const uint8_t *ray_mask_dev, double *d_dev)
{
    int rc = check_frechet(f);
    if (rc) return rc;
    if (!d_dev) return fail(RAY_ERR_INVALID, "null argument");
    return band_scatter(1, f, nullptr, ray_mask_dev, d_dev);
}

int ray_frechet_fidelity_update_dev(ray_frechet *f, const double *x_dev, const double *mismatch_dev,
                                    const uint8_t *ray_mask_dev, double rejection, double decay,
                                    double *fidelity_dev, double *residual_dev)
{
    int rc = check_frechet(f);
    if (rc) return rc;
    if (!x_dev || !mismatch_dev || !fidelity_dev) return fail(RAY_ERR_INVALID, "null argument");
    if (f->num == 0) return RAY_OK;
    hipLaunchKernelGGL(k_br_fidelity, dim3(wave_rows_grid(f->num)), dim3(256), 0, f->ctx->stream, f->rows(), x_dev, mismatch_dev,
                       ray_mask_dev, rejection, decay, fidelity_dev, residual_dev);
    HIPCHK(hipGetLastError());
    return RAY_OK;
}

// host-pointer forms: stage, run, fetch
struct Staged {
    DevBuf d;
    int up(ray_ctx *ctx, const void *src, size_t bytes)
    {
        HIPCHK(d.alloc(bytes));
        if (src && bytes) HIPCHK(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return RAY_OK;
    }
    int down(ray_ctx *ctx, void *dst, size_t bytes)
    {
        if (dst && bytes) HIPCHK(hipMemcpyAsync(dst, d.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return RAY_OK;
    }
};

int ray_frechet_dofs(ray_frechet *f, const double *fidelity, double *dofs)
{
    int rc = check_frechet(f);
    if (rc) return rc;
    if (!fidelity || !dofs) return fail(RAY_ERR_INVALID, "null argument");
    const size_t n2 = (size_t)f->N * f->N;
    Staged fi, out;
    if ((rc = fi.up(f->ctx, fidelity, n2 * 8)) || (rc = out.up(f->ctx, nullptr, (size_t)f->num * 8))) return rc;
    if ((rc = ray_frechet_dofs_dev(f, fi.d.as<double>(), out.d.as<double>()))) return rc;
    if ((rc = out.down(f->ctx, dofs, (size_t)f->num * 8))) return rc;
    HIPCHK(hipStreamSynchronize(f->ctx->stream));
    return RAY_OK;
}

int ray_frechet_matvec(ray_frechet *f, const double *x, double *y)
{
    int rc = check_frechet(f);
    if (rc) return rc;
    if (!x || !y) return fail(RAY_ERR_INVALID, "null argument");
    Staged xi, out;
    if ((rc = xi.up(f->ctx, x, (size_t)f->N * f->N * 8)) || (rc = out.up(f->ctx, nullptr, (size_t)f->num * 32))) return rc;
    if ((rc = ray_frechet_matvec_dev(f, xi.d.as<double>(), out.d.as<double>()))) return rc;
    if ((rc = out.down(f->ctx, y, (size_t)f->num * 32))) return rc;
    HIPCHK(hipStreamSynchronize(f->ctx->stream));
    return RAY_OK;
}

int ray_frechet_matvec_t(ray_frechet *f, const double *v, const uint8_t *ray_mask, double *b)
{
    int rc = check_frechet(f);
    if (rc) return rc;
    if (!v || !b) return fail(RAY_ERR_INVALID, "null argument");
    Staged vi, mi, out;
    if ((rc = vi.up(f->ctx, v, (size_t)f->num * 32)) || (rc = out.up(f->ctx, nullptr, (size_t)f->N * f->N * 8))) return rc;
    if (ray_mask && (rc = mi.up(f->ctx, ray_mask, (size_t)f->num))) return rc;
    if ((rc = ray_frechet_matvec_t_dev(f, vi.d.as<double>(), ray_mask ? mi.d.as<uint8_t>() : nullptr, out.d.as<double>()))) return rc;
    if ((rc = out.down(f->ctx, b, (size_t)f->N * f->N * 8))) return rc;
    HIPCHK(hipStreamSynchronize(f->ctx->stream));
    return RAY_OK;
}

int ray_frechet_normal_matvec(ray_frechet *f, const double *x, const uint8_t *ray_mask, double *w)
{
    int rc = check_frechet(f);
    if (rc) return rc;
    if (!x || !w) return fail(RAY_ERR_INVALID, "null argument");
    const size_t n2 = (size_t)f->N * f->N;
    Staged xi, mi, out;
    if ((rc = xi.up(f->ctx, x, n2 * 8)) || (rc = out.up(f->ctx, nullptr, n2 * 8))) return rc;
    if (ray_mask && (rc = mi.up(f->ctx, ray_mask, (size_t)f->num))) return rc;
    if ((rc = ray_frechet_normal_matvec_dev(f, xi.d.as<double>(), ray_mask ? mi.d.as<uint8_t>() : nullptr, out.d.as<double>()))) return rc;
    if ((rc = out.down(f->ctx, w, n2 * 8))) return rc;
    HIPCHK(hipStreamSynchronize(f->ctx->stream));
    return RAY_OK;
}

int ray_frechet_normal_diag(ray_frechet *f, const uint8_t *ray_mask, double *d)
{
    int rc = check_frechet(f);
    if (rc) return rc;
    if (!d) return fail(RAY_ERR_INVALID, "null argument");
    const size_t n2 = (size_t)f->N * f->N;
    Staged mi, out;
    if ((rc = out.up(f->ctx, nullptr, n2 * 8))) return rc;
    if (ray_mask && (rc = mi.up(f->ctx, ray_mask, (size_t)f->num))) return rc;
    if ((rc = ray_frechet_normal_diag_dev(f, ray_mask ? mi.d.as<uint8_t>() : nullptr, out.d.as<double>()))) return rc;
    if ((rc = out.down(f->ctx, d, n2 * 8))) return rc;
    HIPCHK(hipStreamSynchronize(f->ctx->stream));
    return RAY_OK;
}

int ray_frechet_fidelity_update(ray_frechet *f, const double *x, const double *mismatch, const uint8_t *ray_mask,
                                double rejection, double decay, double *fidelity, double *residual)
{
    int rc = check_frechet(f);
    if (rc) return rc;
    if (!x || !mismatch || !fidelity) return fail(RAY_ERR_INVALID, "null argument");
    const size_t n2 = (size_t)f->N * f->N;
    Staged xi, mm, mi, fi, res;
    if ((rc = xi.up(f->ctx, x, n2 * 8)) || (rc = mm.up(f->ctx, mismatch, (size_t)f->num * 32)) ||
        (rc = fi.up(f->ctx, fidelity, n2 * 8)))
        return rc;
    if (ray_mask && (rc = mi.up(f->ctx, ray_mask, (size_t)f->num))) return rc;
    if (residual && (rc = res.up(f->ctx, nullptr, (size_t)f->num * 32))) return rc;
    if ((rc = ray_frechet_fidelity_update_dev(f, xi.d.as<double>(), mm.d.as<double>(), ray_mask ? mi.d.as<uint8_t>() : nullptr,
                                              rejection, decay, fi.d.as<double>(), residual ? res.d.as<double>() : nullptr)))
        return rc;
    if ((rc = fi.down(f->ctx, fidelity, n2 * 8))) return rc;
    if (residual && (rc = res.down(f->ctx, residual, (size_t)f->num * 32))) return rc;
    HIPCHK(hipStreamSynchronize(f->ctx->stream));
    return RAY_OK;
}

// ---- NEXT-3: ray classification ----------------------------------------------------------------------
int ray_classify_dev(ray_ctx *ctx, const double *m_dev, int64_t ns, int64_t nd, int64_t ld, uint8_t *ortho_dev,
                     double *arg_dev, int32_t *lo_dev, int32_t *hi_dev)
{
    int rc = use(ctx);
    if (rc) return rc;
    const int64_t num = ns * nd;
    if (!m_dev || !arg_dev || !lo_dev || !hi_dev || ns < 0 || nd < 0 || ld < num) return fail(RAY_ERR_INVALID, "bad classify arguments");
    if (num == 0) return RAY_OK;
    hipLaunchKernelGGL(k_classify_rays, dim3((unsigned)((num + 255) / 256)), dim3(256), 0, ctx->stream, m_dev, num, ld,
                       ortho_dev, arg_dev);
    hipLaunchKernelGGL(k_unbroken, dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, ctx->stream, arg_dev, ns, nd, lo_dev, hi_dev);
    HIPCHK(hipGetLastError());
    return RAY_OK;
}

int ray_classify(ray_ctx *ctx, const double *m_host, int64_t ns, int64_t nd, int64_t ld, uint8_t *ortho,
                 double *arg, int32_t *lo, int32_t *hi)
{
    int rc = use(ctx);
    if (rc) return rc;
    const int64_t num = ns * nd;
    if (!m_host || !lo || !hi || ns < 0 || nd < 0 || ld < num) return fail(RAY_ERR_INVALID, "bad classify arguments");
    if (num == 0) return RAY_OK;
    DevBuf dm, dor, darg, dlo, dhi;
    HIPCHK(dm.alloc(sizeof(double) * 8 * (size_t)num));
    HIPCHK(dor.alloc((size_t)num));
    HIPCHK(darg.alloc(sizeof(double) * (size_t)num));
    HIPCHK(dlo.alloc(sizeof(int32_t) * (size_t)ns));
    HIPCHK(dhi.alloc(sizeof(int32_t) * (size_t)ns));
    if ((rc = copy_cols_h2d(ctx, dm.as<double>(), num, m_host, ld, num, 0, 8))) return rc;
    if ((rc = ray_classify_dev(ctx, dm.as<double>(), ns, nd, num, dor.as<uint8_t>(), darg.as<double>(), dlo.as<int32_t>(), dhi.as<int32_t>()))) return rc;
    if (ortho) HIPCHK(hipMemcpyAsync(ortho, dor.p, (size_t)num, hipMemcpyDeviceToHost, ctx->stream));
    if (arg) HIPCHK(hipMemcpyAsync(arg, darg.p, sizeof(double) * (size_t)num, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(lo, dlo.p, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(hi, dhi.p, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RAY_OK;
}

#ifdef RAY_TIMELINE
int ray_debug_timeline(unsigned long long *out, int nblocks)
{
    if (hipDeviceSynchronize() != hipSuccess) return RAY_ERR_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ray_timeline), sizeof(unsigned long long) * 3 * (size_t)nblocks) != hipSuccess) return RAY_ERR_HIP;
    return RAY_OK;
}
#endif

#ifdef RAY_TIMELINE
int ray_debug_fr_timeline(unsigned long long *out, int nblocks)
{
    if (hipDeviceSynchronize() != hipSuccess) return RAY_ERR_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fr_timeline), sizeof(unsigned long long) * 3 * (size_t)nblocks) != hipSuccess) return RAY_ERR_HIP;
    return RAY_OK;
}
#endif

#ifdef RAY_STATS
// instrumentation build only (scripts/): look-up statistics of the cell cache since the last call
int ray_debug_stats(unsigned long long out[12])
{
    unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ray_stats), sizeof(z)) != hipSuccess) return RAY_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ray_stats), z, sizeof(z)) != hipSuccess) return RAY_ERR_HIP;
    return RAY_OK;
}
#endif

#ifdef RAY_FR_PHASES
// instrumentation build only: summed phase cycles of k_merge_emit since the last call
int ray_debug_fr_phases(unsigned long long out[16])
{
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fr_phase), sizeof(z)) != hipSuccess) return RAY_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fr_phase), z, sizeof(z)) != hipSuccess) return RAY_ERR_HIP;
    return RAY_OK;
}
#endif

// ---- misfit ---------------------------------------------------------------------------------------
static int misfit_launch(ray_ctx *ctx, const double *m_dev, const double *obs_dev, int64_t num, int64_t ld,
                         double *out_dev, uint8_t *mask_dev, double perc_thresh, double *mismatch_dev, double *partial)
{
    if (!m_dev || !obs_dev || !out_dev || num < 0 || ld < num) return fail(RAY_ERR_INVALID, "bad misfit arguments");
    if (mismatch_dev && (reinterpret_cast<uintptr_t>(mismatch_dev) & 15)) return fail(RAY_ERR_INVALID, "mismatch must be 16-byte aligned");
    // every one of the kMisfitBlocks blocks writes both of its partials (zeros when it has no rays), so the
    // scratch needs no clearing between calls
    hipLaunchKernelGGL(k_misfit_partial, dim3(kMisfitBlocks), dim3(256), 0, ctx->stream, m_dev, obs_dev, num, ld,
                       perc_thresh, mask_dev, mismatch_dev, partial);
    hipLaunchKernelGGL(k_misfit_final, dim3(1), dim3(256), 0, ctx->stream, partial, out_dev);
    HIPCHK(hipGetLastError());
    return RAY_OK;
}

int ray_misfit_dev(ray_ctx *ctx, const double *m_dev, const double *obs_dev, int64_t num, int64_t ld,
                   double *out_dev, uint8_t *mask_dev, double perc_thresh, double *mismatch_dev)
{
    int rc = use(ctx);
    if (rc) return rc;
    return misfit_launch(ctx, m_dev, obs_dev, num, ld, out_dev, mask_dev, perc_thresh, mismatch_dev, ctx->d_partial);
}

// The allocation-free variant: same kernels, partial sums in this device's copy of g_misfit_scratch.
// Calls nothing that allocates (no hipMalloc / hipFree, no event creation).
int ray_misfit_static_dev(ray_ctx *ctx, const double *m_dev, const double *obs_dev, int64_t num, int64_t ld,
                          double *out_dev, uint8_t *mask_dev, double perc_thresh, double *mismatch_dev)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (num > RAY_MISFIT_STATIC_MAX_RAYS)
        return fail(RAY_ERR_UNSUPPORTED, "ray_misfit_static_dev: num exceeds RAY_MISFIT_STATIC_MAX_RAYS (2^28 rays per call)");
    double *scratch = ctx->device < kMaxDevices ? g_static_partial[ctx->device].load(std::memory_order_acquire) : nullptr;
    if (!scratch) return fail(RAY_ERR_HIP, "ray_misfit_static_dev: the static scratch of this device was not resolved");
    return misfit_launch(ctx, m_dev, obs_dev, num, ld, out_dev, mask_dev, perc_thresh, mismatch_dev, scratch);
}

int ray_misfit(ray_ctx *ctx, const double *m_host, const double *obs_host, int64_t num, int64_t ld,
               double out[2], uint8_t *mask, double perc_thresh, double *mismatch)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!m_host || !obs_host || !out || num < 0 || ld < num) return fail(RAY_ERR_INVALID, "bad misfit arguments");
    WsBuf dm(ctx, kWsM), dobs(ctx, kWsObs), dout(ctx, kWsOut), dmask(ctx, kWsMask), dmis(ctx, kWsMis);
    HIPCHK(dm.alloc(sizeof(double) * 9 * (size_t)num));
    HIPCHK(dobs.alloc(sizeof(double) * 9 * (size_t)num));
    HIPCHK(dout.alloc(sizeof(double) * 2));
    if (mask) HIPCHK(dmask.alloc((size_t)num));
    if (mismatch) HIPCHK(dmis.alloc(sizeof(double) * 4 * (size_t)num));
    if (num > 0) {
        if ((rc = copy_cols_h2d(ctx, dm.as<double>(), num, m_host, ld, num, 4, 4))) return rc;
        if ((rc = copy_cols_h2d(ctx, dobs.as<double>(), num, obs_host, ld, num, 4, 4))) return rc;
    }
    rc = ray_misfit_dev(ctx, dm.as<double>(), dobs.as<double>(), num, num > 0 ? num : 1, dout.as<double>(),
                        mask ? dmask.as<uint8_t>() : nullptr, perc_thresh, mismatch ? dmis.as<double>() : nullptr);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    HIPCHK(hipMemcpyAsync(out, dout.p, sizeof(double) * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (mask && num) HIPCHK(hipMemcpyAsync(mask, dmask.p, (size_t)num, hipMemcpyDeviceToHost, ctx->stream));
    if (mismatch && num) HIPCHK(hipMemcpyAsync(mismatch, dmis.p, sizeof(double) * 4 * (size_t)num, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return RAY_OK;
}

int ray_ctx_set_option(ray_ctx *ctx, int option, int value)
{
    if (!ctx) return fail(RAY_ERR_INVALID, "null context");
    if (option == RAY_OPT_SKIP_INIT_COLUMNS) { ctx->skip_init_cols = value != 0; return RAY_OK; }
    if (option == RAY_OPT_FAN_DIRECTIONS) {
        if (value < 0) return fail(RAY_ERR_INVALID, "fan directions must be >= 0");
        ctx->fan_nd = value;
        return RAY_OK;
    }
    if (option == RAY_OPT_SOURCE_STRIDE) {
        if (value < 1) return fail(RAY_ERR_INVALID, "source stride must be >= 1");
        ctx->src_stride = value;
        return RAY_OK;
    }
    if (option == RAY_OPT_TUNE_BLOCK_THREADS) {
        if (value != 64 && value != 128 && value != 256) return fail(RAY_ERR_INVALID, "block threads: 64, 128 or 256");
        ctx->tune_block = value;
        return RAY_OK;
    }
    if (option == RAY_OPT_TUNE_WAVE_DIRS) {
        if (value != 0 && value != 4 && value != 8 && value != 16 && value != 32 && value != 64) return fail(RAY_ERR_INVALID, "wave directions: 0 (automatic), 4, 8, 16, 32 or 64");
        ctx->tune_wave_dirs = value;
        return RAY_OK;
    }
    if (option == RAY_OPT_TUNE_SORT_KEY_BITS) {
        if (value != 0 && value != 64) return fail(RAY_ERR_INVALID, "sort key bits: 0 (automatic) or 64");
        ctx->tune_sort_key = value;
        return RAY_OK;
    }
    if (option == RAY_OPT_TUNE_FRECHET_LDS_ROWS) { ctx->tune_lds_rows = value < 0 ? -1 : value; return RAY_OK; }
    if (option == RAY_OPT_TUNE_MERGE_GRID) { ctx->tune_merge_grid = value < 0 ? 0 : value; return RAY_OK; }
    if (option == RAY_OPT_TUNE_FRECHET_SLAB_MB) { ctx->tune_slab_mb = value < 0 ? 0 : value; return RAY_OK; }
    if (option == RAY_OPT_TUNE_MERGE_ENTRIES) { ctx->tune_merge_entries = value != 0; return RAY_OK; }
    return fail(RAY_ERR_INVALID, "unknown option");
}

// page-lock a caller-owned host array so that the host-pointer entry points copy at the full PCIe rate.
// The runtime wants whole pages: the page-aligned span around [ptr, ptr + bytes) is what gets registered
// (remembered per ptr for ray_host_unregister).
static std::vector<std::pair<void *, void *>> g_registered;      // (caller's ptr, registered base)
int ray_host_register(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return fail(RAY_ERR_INVALID, "ray_host_register: null pointer / zero size");
    const uintptr_t page = 4096, lo = reinterpret_cast<uintptr_t>(ptr) & ~(page - 1);
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(ptr) + bytes + page - 1) & ~(page - 1);
    HIPCHK(hipHostRegister(reinterpret_cast<void *>(lo), hi - lo, hipHostRegisterDefault));
    g_registered.emplace_back(ptr, reinterpret_cast<void *>(lo));
    return RAY_OK;
}

int ray_host_unregister(void *ptr)
{
    for (size_t k = 0; k < g_registered.size(); ++k)
        if (g_registered[k].first == ptr) {
            void *base = g_registered[k].second;
            g_registered[k] = g_registered.back();
            g_registered.pop_back();
            HIPCHK(hipHostUnregister(base));
            return RAY_OK;
        }
    return fail(RAY_ERR_INVALID, "ray_host_unregister: not a pointer ray_host_register was given");
}

int ray_ctx_set_arith(ray_ctx *ctx, int arith)
{
    if (!ctx) return fail(RAY_ERR_INVALID, "null context");
    if (arith != RAY_ARITH_EXACT && arith != RAY_ARITH_FUSED) return fail(RAY_ERR_INVALID, "bad arith");
    ctx->arith = arith;
    return RAY_OK;
}

// ---- counters ---------------------------------------------------------------------------------------
int ray_ctx_enable_timing(ray_ctx *ctx, int on)
{
    if (!ctx) return fail(RAY_ERR_INVALID, "null context");
    ctx->timing = on != 0;
    return RAY_OK;
}

int ray_get_counters(ray_ctx *ctx, ray_counters *out)
{
    int rc = use(ctx);
    if (rc) return rc;
    if (!out) return fail(RAY_ERR_INVALID, "null out");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    drain_timing(ctx);
    unsigned long long s = 0;
    HIPCHK(hipMemcpyAsync(&s, ctx->d_steps, sizeof(s), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    out->rays = ctx->rays;
    out->ray_steps = s;
    out->launches = ctx->launches;
    out->kernel_ms = ctx->kernel_ms;
    return RAY_OK;
}

int ray_reset_counters(ray_ctx *ctx)
{
    int rc = use(ctx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    drain_timing(ctx);
    HIPCHK(hipMemsetAsync(ctx->d_steps, 0, sizeof(unsigned long long), ctx->stream));   // ordered before the next launch
    ctx->rays = ctx->launches = 0;
    ctx->kernel_ms = 0;
    return RAY_OK;
}

} // extern "C"

#include "ray_group.h"
