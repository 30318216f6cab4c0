// ray_device.h -- gfx950 device code of the forward ray-shooting core (one thread = one ray).
//
// Arithmetic contract ("exact" build, the default): compiled with -ffp-contract=off and no
// fast-math, every +,-,* below is one IEEE binary64 rounding in the order the reference writes
// it, so gridded results are bit-identical to an un-fused CPU evaluation of the same Julia
// expressions.  Two places use fma on purpose, both producing the correctly rounded quotient
// that the reference's `/` produces, at 3 instructions instead of the ~11 of a full IEEE
// division sequence (validated against `/` on 2.4e9 operands, tests/test_host_math.py):
//   - x / 6.0         (RK4 combination, NonObstacle.jl:76,171; Obstacle.jl:38,317)
//   - (x - p[1]) / h  (cell index,      NonObstacle.jl:33-34)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ray {

constexpr int kStExited = 0x01, kStStepCap = 0x02, kStBisectCap = 0x04, kStOob = 0x08,
              kStCeased = 0x10;
constexpr int kBisectCap = 200;

// ---- gridded medium in HBM ------------------------------------------------------------------
// One 96-byte record per cell (I,J), I,J = 1..N-1, at index (I-1) + (J-1)*(N-1):
//   [ eval[I,J,1:4] | grad[I,J,1:4] | grad[I,J,5:8] ]      (NonObstacle.jl:192-193 tables)
// so one right-hand side touches one or two 128-byte lines instead of 12 planes.
struct GridDev {
    const double *rec;   // (N-1)*(N-1)*12
    const double *p;     // N nodes, linspace(-ext, ext, N)
    int N;
    double p1;           // p[1] = -ext
    double h;            // p[2] - p[1]            (NonObstacle.jl:32)
    double rh;           // RN(1/h), for the Markstein quotient
};
struct SinProdDev { double A, wx, wy, gx, gy; };

enum MediumKind { kSinProd = 0, kGrid = 1 };
enum ObstacleKind { kObNone = 0, kObCircle = 1 };

// correctly rounded a/b given rb = RN(1/b) (Markstein); see header comment
__device__ __forceinline__ double div_rn(double a, double b, double rb)
{
    double q0 = a * rb;
    double r = __builtin_fma(-b, q0, a);
    return __builtin_fma(r, rb, q0);
}
__device__ __forceinline__ double div6(double a)
{
    return div_rn(a, 6.0, 0.16666666666666666);
}

// I-1 for I = Int64(floor((x - p[1])/h)) + 1, clamped to the cells whose grad entry exists
// (reference: undefined behaviour outside; here flagged)           NonObstacle.jl:33-34
__device__ __forceinline__ int cell0(double x, const GridDev &g, int &oob)
{
    double q = __builtin_floor(div_rn(x - g.p1, g.h, g.rh));
    double qc = __builtin_fmin(__builtin_fmax(q, 0.0), (double)(g.N - 3));
    oob |= (qc != q);
    return (int)qc;
}

// a2 / a3(H): DiscreteHamilton                                     NonObstacle.jl:30-42
__device__ __forceinline__ void rhs_grid(const GridDev &g, const double *pt, const double X[4], double H[4], int &oob)
{
    const int i = cell0(X[0], g, oob);
    const int j = cell0(X[1], g, oob);
    const double2 *r = reinterpret_cast<const double2 *>(g.rec) +
                       ((size_t)i + (size_t)j * (size_t)(g.N - 1)) * 6;
    const double2 e01 = r[0], e23 = r[1], a01 = r[2], a23 = r[3], b01 = r[4], b23 = r[5];
    const double dx = X[0] - pt[i], dy = X[1] - pt[j];
    const double dxdy = dx * dy;
    const double tau = X[2] * X[2] + X[3] * X[3];
    const double c = ((e01.x + dx * e01.y) + dy * e23.x) + dxdy * e23.y;
    const double gcX = ((a01.x + dx * a01.y) + dy * a23.x) + dxdy * a23.y;
    const double gcY = ((b01.x + dx * b01.y) + dy * b23.x) + dxdy * b23.y;
    const double c2 = c * c;
    H[0] = c2 * X[2];
    H[1] = c2 * X[3];
    H[2] = ((-gcX) * c) * tau;
    H[3] = ((-gcY) * c) * tau;
}

// z' * eval[I,J,:] alone (c at the sensor for the initial slowness)  NonObstacle.jl:39,62
__device__ __forceinline__ double speed_grid(const GridDev &g, const double *pt, double x, double y, int &oob)
{
    const int i = cell0(x, g, oob);
    const int j = cell0(y, g, oob);
    const double2 *r = reinterpret_cast<const double2 *>(g.rec) +
                       ((size_t)i + (size_t)j * (size_t)(g.N - 1)) * 6;
    const double2 e01 = r[0], e23 = r[1];
    const double dx = x - pt[i], dy = y - pt[j];
    return ((e01.x + dx * e01.y) + dy * e23.x) + (dx * dy) * e23.y;
}

// a1: Hamilton with the analytic sin-product medium                  NonObstacle.jl:1-4
__device__ __forceinline__ void rhs_sinprod(const SinProdDev &s, const double X[4], double H[4])
{
    double sx, cx, sy, cy;
    sincos(s.wx * X[0], &sx, &cx);
    sincos(s.wy * X[1], &sy, &cy);
    const double speed = 1 + s.A * sx * sy;
    const double g0 = s.gx * (cx * sy);
    const double g1 = s.gy * (sx * cy);
    const double tau = X[2] * X[2] + X[3] * X[3];
    const double s2 = speed * speed;
    H[0] = s2 * X[2];
    H[1] = s2 * X[3];
    H[2] = ((-tau) * g0) * speed;
    H[3] = ((-tau) * g1) * speed;
}
__device__ __forceinline__ double speed_sinprod(const SinProdDev &s, double x, double y)
{
    return 1 + s.A * sin(s.wx * x) * sin(s.wy * y);
}

struct MediumDev {
    GridDev g;
    SinProdDev sp;
};

template <int MED>
__device__ __forceinline__ void rhs(const MediumDev &m, const double *pt, const double X[4], double H[4], int &oob)
{
    if (MED == kGrid) rhs_grid(m.g, pt, X, H, oob);
    else rhs_sinprod(m.sp, X, H);
}

// classical RK4, a6/a7 form: k = H*dt; X + k1/2; ...; /6.0       NonObstacle.jl:72-76
template <int MED>
__device__ __forceinline__ void rk4_kdt(const MediumDev &m, const double *pt, double X[4], double dt, int &oob)
{
    double H[4], Y[4], acc[4];
    rhs<MED>(m, pt, X, H, oob);
#pragma unroll
    for (int k = 0; k < 4; ++k) { double kk = H[k] * dt; acc[k] = kk; Y[k] = X[k] + kk * 0.5; }
    rhs<MED>(m, pt, Y, H, oob);
#pragma unroll
    for (int k = 0; k < 4; ++k) { double kk = H[k] * dt; acc[k] = acc[k] + 2 * kk; Y[k] = X[k] + kk * 0.5; }
    rhs<MED>(m, pt, Y, H, oob);
#pragma unroll
    for (int k = 0; k < 4; ++k) { double kk = H[k] * dt; acc[k] = acc[k] + 2 * kk; Y[k] = X[k] + kk; }
    rhs<MED>(m, pt, Y, H, oob);
#pragma unroll
    for (int k = 0; k < 4; ++k) { double kk = H[k] * dt; X[k] = X[k] + div6(acc[k] + kk); }
}

// a8/a9 form: k = H; X + k/2*dt; X += (k1+2k2+2k3+k4)*dt/6.0     NonObstacle.jl:161-164,171
template <int MED>
__device__ __forceinline__ void rk4_k(const MediumDev &m, const double *pt, double X[4], double dt, int &oob)
{
    double H[4], Y[4], acc[4];
    rhs<MED>(m, pt, X, H, oob);
#pragma unroll
    for (int k = 0; k < 4; ++k) { acc[k] = H[k]; Y[k] = X[k] + (H[k] * 0.5) * dt; }
    rhs<MED>(m, pt, Y, H, oob);
#pragma unroll
    for (int k = 0; k < 4; ++k) { acc[k] = acc[k] + 2 * H[k]; Y[k] = X[k] + (H[k] * 0.5) * dt; }
    rhs<MED>(m, pt, Y, H, oob);
#pragma unroll
    for (int k = 0; k < 4; ++k) { acc[k] = acc[k] + 2 * H[k]; Y[k] = X[k] + H[k] * dt; }
    rhs<MED>(m, pt, Y, H, oob);
#pragma unroll
    for (int k = 0; k < 4; ++k) X[k] = X[k] + div6((acc[k] + H[k]) * dt);
}

__device__ __forceinline__ double norm2(double a, double b) { return __builtin_sqrt(a * a + b * b); }

__device__ __forceinline__ unsigned wave_sum(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

} // namespace ray
