// ray_frechet.h -- NEXT-1 of SURVEY.md section 8f: the Jacobian rho and the Frechet matrix M of
// ScatterForwardOperator (NonObstacle.jl:6-28 DiscreteHamilton with Hessian, :107-139 dV,
// :141-187 ChunkProcessing!, :212-218 hess table, :242 sparse + droptol).
//
// The reference builds, per ray and per RK4 step, three 4 x N^2 sparse matrices (O(N^2) each) and a
// dense 4num x N^2 SharedArray that it then compresses.  Here the sparse matrix is built on the device:
//   k_frechet      one thread integrates (X, rho) of one ray and EMITS the 12-node stencil
//                  contributions Theta = inv(rho) dV w of every trajectory node into its ray's own
//                  contiguous stretch of the slab (8-byte keys, 32-byte values): the merge kernels
//                  read a ray's entries as one stream;
//   k_merge_sort   one workgroup per ray sorts that ray's (node, emission order) keys in LDS and
//                  ranks the distinct nodes;
//   k_merge_emit   sums each node's contributions in emission order (the order of the reference's
//                  Phi += Theta), applies the final rho (M rows = rho * Phi, :184) and counts the
//                  entries that survive droptol per matrix row;
//   k_csr_emit     writes the CSR arrays (rows ray-major like reshape(M, 4num, N^2), columns sorted).
// The reference goes through LAPACK inv / BLAS gemm here, so parity is by tolerance; the X
// trajectory itself is bit-identical to k_trace (same expressions).
#pragma once
#include "ray_device.h"

namespace ray {

constexpr int kFrSlotsPerCall = 12;     // nodes touched by one dV evaluation

// hess records (NonObstacle.jl:212-218): [hess[I,J,1:4] | hess[I,J,5:8] | hess[I,J,9:12]] per cell,
// same cell indexing as the eval/grad records; zero where the reference never writes.
__global__ __launch_bounds__(256) void k_build_hess(const double *__restrict__ rec, int N, double dx,
                                                    double *__restrict__ hrec)
{
    const int n1 = N - 1;
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t)n1 * n1) return;
    const int i = (int)(id % n1), j = (int)(id / n1);            // cell (I,J) = (i+1, j+1)
    double out[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) out[k] = 0.0;
    if (i >= 2 && i <= N - 4 && j >= 2 && j <= N - 4) {          // I,J in 3..N-3
        const double two_dx = 2 * dx;
        const double *e = rec + (size_t)(id + 1) * 12, *w = rec + (size_t)(id - 1) * 12;
        const double *n = rec + (size_t)(id + n1) * 12, *s = rec + (size_t)(id - n1) * 12;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            out[k] = (e[4 + k] - w[4 + k]) / two_dx;             // grad[I+1,J,1:4] - grad[I-1,J,1:4]
            out[4 + k] = (n[4 + k] - s[4 + k]) / two_dx;         // grad[I,J+1,1:4] - grad[I,J-1,1:4]
            out[8 + k] = (n[8 + k] - s[8 + k]) / two_dx;         // grad[I,J+1,5:8] - grad[I,J-1,5:8]
        }
    }
    double2 *r = reinterpret_cast<double2 *>(hrec) + (size_t)id * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) r[k] = make_double2(out[2 * k], out[2 * k + 1]);
}

struct PointEval {
    int i, j;                 // 0-based cell
    double dx, dy, c, gcX, gcY, tau;
};

// table look-up at X (per-lane loads), H in the chosen flavour          NonObstacle.jl:30-42
template <int AR>
__device__ __forceinline__ void point_eval(const GridDev &g, const double X[4], double H[4], PointEval &q, int &oob)
{
    OobMax om = 0;
    double u = 0, v = 0;                                    // kFused: in-cell offsets in cell units (ray_device.h)
    q.i = cell_of<AR>(X[0], g, om, g.nb, u);
    q.j = cell_of<AR>(X[1], g, om, g.nb, v);
    oob |= oob_hit(g, om);
    const unsigned cid = mad24((unsigned)q.j, (unsigned)(g.N - 1), (unsigned)q.i);
    const double2 *src = reinterpret_cast<const double2 *>(AR == kFused ? g.recf : g.rec) + (size_t)cid * 6;
    double2 r[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) r[k] = src[k];
    if (AR == kFused) {
        h_from_record<AR>(X, u, v, r, H, q.c, q.gcX, q.gcY, q.tau);
        q.dx = u * g.h;                                     // physical offsets for the hess look-up and dV
        q.dy = v * g.h;
    } else {
        const double *pt = lds_nodes();
        q.dx = X[0] - pt[q.i];
        q.dy = X[1] - pt[q.j];
        h_from_record<AR>(X, q.dx, q.dy, r, H, q.c, q.gcX, q.gcY, q.tau);
    }
}

__device__ __forceinline__ double bil(double dx, double dy, double dxdy, double2 a, double2 b)
{
    return ((a.x + dx * a.y) + dy * b.x) + dxdy * b.y;           // z' * [a.x a.y b.x b.y]
}

// a3: H and Jm = dH/dX (row-major 4x4).                           NonObstacle.jl:6-28
// M = [2*c*X[3:4]*g'  c^2*eye(2); -(c*h + g*g')*tau  -2*c*g*X[3:4]']
// Jm from an evaluated point (the hess part of a3)
__device__ __forceinline__ void jac_from_point(const GridDev &g, const double X[4], const PointEval &q, double Jm[16], int &oob)
{
    // hess exists for I,J <= N-3 only: clamp + flag (reference: out-of-bounds read)
    const int ih = q.i < g.N - 4 ? q.i : g.N - 4, jh = q.j < g.N - 4 ? q.j : g.N - 4;
    oob |= (ih != q.i) | (jh != q.j);
    const double2 *hs = reinterpret_cast<const double2 *>(g.hrec) + ((size_t)ih + (size_t)jh * (size_t)(g.N - 1)) * 6;
    const double dxdy = q.dx * q.dy;
    const double hXX = bil(q.dx, q.dy, dxdy, hs[0], hs[1]);
    const double hXY = bil(q.dx, q.dy, dxdy, hs[2], hs[3]);
    const double hYY = bil(q.dx, q.dy, dxdy, hs[4], hs[5]);
    const double c = q.c, tau = q.tau, c2 = c * c;
    const double gv[2] = {q.gcX, q.gcY};
    const double hv[2][2] = {{hXX, hXY}, {hXY, hYY}};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            Jm[a * 4 + b] = ((2 * c) * X[2 + a]) * gv[b];
            Jm[a * 4 + 2 + b] = (a == b) ? c2 : 0.0;
            Jm[(2 + a) * 4 + b] = (-(c * hv[a][b] + gv[a] * gv[b])) * tau;
            Jm[(2 + a) * 4 + 2 + b] = ((-2 * c) * gv[a]) * X[2 + b];
        }
}

template <int AR>
__device__ __forceinline__ void hamilton_jac(const GridDev &g, const double X[4], double H[4], double Jm[16], int &oob)
{
    PointEval q;
    point_eval<AR>(g, X, H, q, oob);
    jac_from_point(g, X, q, Jm, oob);
}

// One RK4 stage of the Jacobian equation, column by column so that only four entries of V = Jm * Rin are
// alive at a time: acc (+)= wa * V, Rout = base + c * V.  Rout may be Rin (column j of Rin is consumed
// before it is overwritten).  Element for element the same operations as mat4_mul + the two updates.
template <int MODE>   // 0: acc = V (stage 1); 1: acc += 2 V (stages 2, 3); 2: Rout = base + c * (acc + V) (stage 4)
__device__ __forceinline__ void rho_stage(const double Jm[16], const double Rin[16], const double base[16],
                                          double acc[16], double Rout[16], double c)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double t = Jm[i * 4] * Rin[j];
#pragma unroll
            for (int k = 1; k < 4; ++k) t = __builtin_fma(Jm[i * 4 + k], Rin[k * 4 + j], t);
            v[i] = t;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = i * 4 + j;
            if (MODE == 0) { acc[e] = v[i]; Rout[e] = __builtin_fma(v[i], c, base[e]); }
            else if (MODE == 1) { acc[e] = __builtin_fma(2.0, v[i], acc[e]); Rout[e] = __builtin_fma(v[i], c, base[e]); }
            else Rout[e] = __builtin_fma(acc[e] + v[i], c, base[e]);
        }
    }
}

__device__ __forceinline__ void mat4_mul(const double A[16], const double B[16], double C[16])
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = A[i * 4] * B[j];
#pragma unroll
            for (int k = 1; k < 4; ++k) s = __builtin_fma(A[i * 4 + k], B[k * 4 + j], s);
            C[i * 4 + j] = s;
        }
}

// inverse of a well-conditioned 4x4 (rho is a perturbation of a symplectic matrix, det ~ 1) by the
// adjugate through 2x2 sub-determinants; the reference calls LAPACK inv (NonObstacle.jl:158).
__device__ __forceinline__ void mat4_inv(const double m[16], double inv[16])
{
    const double s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    const double s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const double c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    const double c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    const double id = 1.0 / det;
    inv[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * id;
    inv[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * id;
    inv[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * id;
    inv[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * id;
    inv[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * id;
    inv[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * id;
    inv[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * id;
    inv[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * id;
    inv[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * id;
    inv[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * id;
    inv[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * id;
    inv[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * id;
    inv[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * id;
    inv[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * id;
    inv[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * id;
    inv[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * id;
}

struct FrechetArgs {
    MediumDev med;
    const double *m;            // (num, 9): initial states and T
    int64_t num, ld;
    double dt;
    int max_steps;
    double cease_r;
    const int64_t *wave_base;   // slot-row offset of every wave (a wave owns 64 x rows entries from 64 * base)
    const int32_t *wave_cap;    // step capacity of every wave (rows = 12 * (cap + 1) entries per ray)
    int64_t *keys;              // [wave][lane][row] node index (0-based, first index x) or -1
    double *vals;               // [wave][lane][row][4]
    double *X;                  // (num, 4) end states
    int64_t ldx;
    double *rho;                // (num, 16) final Jacobian, row-major
    int32_t *status, *nsteps;
    unsigned long long *step_counter;
};

constexpr uint64_t kKeyDead = ~0ull;    // sorted key layout: node << 40 | emission index << 20 | rank
constexpr int kMergeMaxEntries = 16384; // per ray: 128 KiB of LDS keys

__host__ __device__ inline int frechet_entries(int steps) { return steps > 0 ? (steps + 1) * kFrSlotsPerCall : 0; }

// first slab entry of ray `lane` of a wave whose rays hold frechet_entries(cap) entries each
__device__ __forceinline__ int64_t ray_slab0(int64_t wave_base, int cap, int lane)
{
    return wave_base * 64 + (int64_t)lane * frechet_entries(cap);
}

// Theta = inv(rho) * dV(X) * w for the 12 stencil nodes of X's cell -> entries [12 it, 12 it + 12) of
// this lane's ray.  dV: NonObstacle.jl:107-139.  (ALL 64 lanes of the one-wave workgroup together.)
// A ray's entries are contiguous in the slab, so lane-wise stores would touch 64 lines per
// instruction: the wave transposes four entries per lane at a time through LDS (2 KB of keys, 8 KB
// of values) and stores them as 16-byte pieces, consecutive lanes on consecutive pieces of one ray
// (values: 8 lanes = one ray's 128 bytes).  Finished lanes' pieces are not stored: nobody reads them.
constexpr int kFrStageBytes = 64 * 4 * 8 + 64 * 4 * 32;
__host__ __device__ inline size_t frechet_lds_bytes(int N) { return (((size_t)N * 8 + 15) & ~(size_t)15) + kFrStageBytes; }

template <int AR>
__device__ __forceinline__ void emit_theta(const FrechetArgs &a, const double X[4], const PointEval &q, int o,
                                           const double rho[16], double w,
                                           bool live, int64_t wave_slab0, int rows, int it, unsigned lane, int &oob)
{
    double inv[16];
    mat4_inv(rho, inv);
    const GridDev &g = a.med.g;
    const double h = g.h;
    const double u = q.dx / h, v = q.dy / h;
    const double phi[4] = {(1 - u) * (1 - v), u * (1 - v), (1 - u) * v, u * v};
    const double psi[4] = {u * (1 - v), -(1 - u) * (1 - v), u * v, -(1 - u) * v};
    const double gam[4] = {(1 - u) * v, u * v, -(1 - u) * (1 - v), -u * (1 - v)};
    const double ct2h = q.c * q.tau / (2 * h);
    const int N = g.N;
    // node offsets relative to (i, j): cell nodes, x-neighbours (row 3), y-neighbours (row 4)
    const int di[12] = {0, 1, 0, 1, 2, -1, 2, -1, 0, 1, 0, 1};
    const int dj[12] = {0, 0, 1, 1, 0, 0, 1, 1, 2, 2, -1, -1};
    char *stage = ray_smem + (((size_t)N * 8 + 15) & ~(size_t)15);
    int64_t *skeys = reinterpret_cast<int64_t *>(stage);                    // [lane][4]
    double2 *svals = reinterpret_cast<double2 *>(stage + 64 * 4 * 8);       // [lane][4][2]
    const unsigned long long lm = __ballot(live);
#pragma unroll
    for (int grp = 0; grp < 3; ++grp) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = grp * 4 + kk;
            const int ni = q.i + di[k], nj = q.j + dj[k];
            const bool ok = ni >= 0 && ni < N && nj >= 0 && nj < N;
            o |= !ok;
            double th[4];
            if (k < 4) {
                const double A0 = 2 * q.c * X[2] * phi[k], A1 = 2 * q.c * X[3] * phi[k];
                const double A2 = -q.tau * q.gcX * phi[k] + ct2h * psi[k];
                const double A3 = -q.tau * q.gcY * phi[k] + ct2h * gam[k];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    th[r] = (((inv[r * 4] * A0 + inv[r * 4 + 1] * A1) + inv[r * 4 + 2] * A2) + inv[r * 4 + 3] * A3) * w;
            } else if (k < 8) {
                const double A2 = -ct2h * psi[k - 4];
#pragma unroll
                for (int r = 0; r < 4; ++r) th[r] = (inv[r * 4 + 2] * A2) * w;
            } else {
                const double A3 = -ct2h * gam[k - 8];
#pragma unroll
                for (int r = 0; r < 4; ++r) th[r] = (inv[r * 4 + 3] * A3) * w;
            }
            skeys[lane * 4 + kk] = ok ? (int64_t)(ni + nj * N) : (int64_t)-1;
            svals[(lane * 4 + kk) * 2] = make_double2(th[0], th[1]);
            svals[(lane * 4 + kk) * 2 + 1] = make_double2(th[2], th[3]);
        }
        __syncthreads();
        const int64_t e0 = (int64_t)it * kFrSlotsPerCall + grp * 4;          // first entry of this group in a ray
#pragma unroll
        for (int t = 0; t < 2; ++t) {                                         // keys: 128 pieces, 2 per ray
            const unsigned qq = t * 64 + lane, ray = qq >> 1, part = qq & 1;
            if ((lm >> ray) & 1) {
                const double2 piece = reinterpret_cast<const double2 *>(skeys)[qq];
                *reinterpret_cast<double2 *>(a.keys + wave_slab0 + (int64_t)ray * rows + e0 + part * 2) = piece;
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {                                         // values: 512 pieces, 8 per ray
            const unsigned qq = t * 64 + lane, ray = qq >> 3, part = qq & 7;
            if ((lm >> ray) & 1) {
                const double2 piece = svals[qq];
                *reinterpret_cast<double2 *>(a.vals + (wave_slab0 + (int64_t)ray * rows + e0) * 4 + part * 2) = piece;
            }
        }
        __syncthreads();
    }
    if (live) oob |= o;
}

// ChunkProcessing! for one ray per thread                         NonObstacle.jl:146-185
// The reference adds Theta twice per step (:158 before, :174 after) -- the "after" point of step n and
// the "before" point of step n+1 are the same (X, rho), so one emission per trajectory node carries
// both weights (trapezoid rule; the cease term :178 joins the last node's weight).
template <int AR>
__global__ __launch_bounds__(64) void k_frechet(const FrechetArgs a)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = r < a.num;
    const unsigned lane = threadIdx.x & 63;
    const int64_t wave = r >> 6;
    {   // node table into LDS (the look-ups read it from there)
        double *p_lds = reinterpret_cast<double *>(ray_smem);
        for (int k = threadIdx.x; k < a.med.g.N; k += blockDim.x) p_lds[k] = a.med.g.p[k];
        __syncthreads();
    }
    const int cap = a.wave_cap[wave];
    const int64_t wave_slab0 = ray_slab0(a.wave_base[wave], cap, 0);
    const int rows = frechet_entries(cap);
    double X[4] = {0, 0, 1, 0}, T = 0;
    double rho[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    int steps = 0, st = 0, oob = 0, it = 0;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = a.m[r + k * a.ld];
        T = a.m[r + 8 * a.ld];
        if (!(X[0] == X[0]) || !(X[1] == X[1])) oob = 1;
    }
    double t = 0, dt = a.dt, wp = 0.0;
    bool active = valid && t < T, fin = false;
    for (;;) {
        if (active && (it >= cap || it >= a.max_steps)) { st |= kStStepCap; active = false; fin = true; }
        const bool anyact = __any(active);
        if (!anyact && !__any(fin)) break;
        double dtl = 0.0;
        if (active) {
            if ((t + dt) > T) dt = T - t;                                    // :153-156
            t = t + dt;                                                      // :157
            dtl = dt;
        }
        // the trajectory node X is evaluated once: for its emission and for RK4 stage 1
        double k1[4];
        PointEval q0;
        int o0 = 0;
        point_eval<AR>(a.med.g, X, k1, q0, o0);
        emit_theta<AR>(a, X, q0, o0, rho, wp + dtl * 0.5, active || fin, wave_slab0, rows, it, lane, oob);
        fin = false;
        if (!anyact) break;                                                  // that was the last node
        // RK4 for X (bit-identical to k_trace) and rho                       :161-172
        double H[4], Y[4], accx[4], Jm[16], R[16], accr[16];
        int o = o0;
        const double hdt = 0.5 * dtl;
        jac_from_point(a.med.g, X, q0, Jm, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) { accx[k] = k1[k]; Y[k] = (AR == kFused) ? __builtin_fma(k1[k], hdt, X[k]) : X[k] + k1[k] * hdt; }
        rho_stage<0>(Jm, rho, rho, accr, R, hdt);
        hamilton_jac<AR>(a.med.g, Y, H, Jm, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) { accx[k] = __builtin_fma(2.0, H[k], accx[k]); Y[k] = (AR == kFused) ? __builtin_fma(H[k], hdt, X[k]) : X[k] + H[k] * hdt; }
        rho_stage<1>(Jm, R, rho, accr, R, hdt);
        hamilton_jac<AR>(a.med.g, Y, H, Jm, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) { accx[k] = __builtin_fma(2.0, H[k], accx[k]); Y[k] = (AR == kFused) ? __builtin_fma(H[k], dtl, X[k]) : X[k] + H[k] * dtl; }
        rho_stage<1>(Jm, R, rho, accr, R, dtl);
        hamilton_jac<AR>(a.med.g, Y, H, Jm, o);
        const double dt6 = dtl * 0.16666666666666666;
        if (active) {                          // a finished lane keeps (X, rho) untouched whatever its stages evaluated to
#pragma unroll
            for (int k = 0; k < 4; ++k)
                X[k] = (AR == kFused) ? __builtin_fma(accx[k] + H[k], dt6, X[k]) : X[k] + div6((accx[k] + H[k]) * dtl);
            rho_stage<2>(Jm, R, rho, accr, rho, dt6);
        }
        ++it;
        wp = 0.0;
        if (active) {
            steps = it;
            wp = dtl * 0.5;                                                  // :174-175
            const double s = X[0] * X[0] + X[1] * X[1];
            oob |= o | !(s == s);
            const double c2 = a.cease_r * a.cease_r;
            bool out;
            if (s > c2 * 1.0000000001) out = true;
            else if (s < c2 * 0.9999999999) out = false;
            else out = __builtin_sqrt(s) > a.cease_r;
            if (out) { wp = wp + (T - t); t = t + (T - t); st |= kStCeased; }   // :177-181
            active = t < T;
            fin = !active;
        }
    }
    if (oob) st |= kStOob;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a.X[r + k * a.ldx] = X[k];
#pragma unroll
        for (int k = 0; k < 16; ++k) a.rho[r * 16 + k] = rho[k];
        a.status[r] = st;
        a.nsteps[r] = steps;
    }
    const unsigned wsum = wave_sum((unsigned)steps);
    if (lane == 0 && a.step_counter) atomicAdd(a.step_counter, (unsigned long long)wsum);
}

// ---- merge: distinct nodes of one ray --------------------------------------------------------------
// One workgroup per ray.  Keys (node, emission index) are sorted in LDS (bitonic, n2 = next power of
// two); every entry gets the rank of its node among the ray's distinct nodes and is written back in
// place as node << 40 | emission << 20 | rank; ucount[ray] = number of distinct nodes.
// The sort is LDS-bandwidth bound, so the compare-exchange passes with stride < 16 -- the last four of
// every stage, and all of the first four stages -- run in registers on 16-key chunks: one LDS round
// trip instead of four.  Keys sit padded by one per 16 (a thread's chunk is 136 bytes from its
// neighbour's: conflict-free 8-byte reads).
// chunk = keys a thread sorts in registers = padding period: 16 eight-byte keys, 32 four-byte keys
__host__ __device__ inline size_t merge_sort_lds_bytes(int n2, size_t key_bytes)
{
    const int chunk = key_bytes == 4 ? 32 : 16;
    return (size_t)(n2 + n2 / chunk) * key_bytes;
}

// Sort keys.  64-bit: node << 40 | emission << 20.  32-bit (chosen by the host when the node index and
// the emission index fit together: N^2 < 2^(32 - eshift), n2 <= 2^eshift): node << eshift | emission --
// half the LDS traffic, one-instruction compares, min/max exchanges.
template <typename K> struct SortKey;
template <> struct SortKey<uint64_t> {
    static constexpr int kChunk = 16, kShift = 4;
    static __device__ __forceinline__ uint64_t dead() { return kKeyDead; }
    static __device__ __forceinline__ uint64_t pack(int64_t node, int e, int) { return (uint64_t)node << 40 | (uint64_t)e << 20; }
    static __device__ __forceinline__ uint64_t node(uint64_t k, int) { return k >> 40; }
    static __device__ __forceinline__ uint64_t out(uint64_t k, int, int rank) { return k | (uint64_t)rank; }
};
template <> struct SortKey<uint32_t> {
    static constexpr int kChunk = 32, kShift = 5;
    static __device__ __forceinline__ uint32_t dead() { return 0xffffffffu; }
    static __device__ __forceinline__ uint32_t pack(int64_t node, int e, int es) { return (uint32_t)node << es | (uint32_t)e; }
    static __device__ __forceinline__ uint64_t node(uint32_t k, int es) { return k >> es; }
    static __device__ __forceinline__ uint64_t out(uint32_t k, int es, int rank)
    {
        return (uint64_t)(k >> es) << 40 | (uint64_t)(k & ((1u << es) - 1u)) << 20 | (uint64_t)rank;
    }
};

// strides min(size/2, chunk/2) .. 1 of bitonic stage `size` on the chunk that starts at key c0
template <typename K>
__device__ __forceinline__ void sort_chunk_stage(K v[SortKey<K>::kChunk], int c0, int size)
{
    constexpr int kSortChunk = SortKey<K>::kChunk;
#pragma unroll
    for (int stride = kSortChunk / 2; stride > 0; stride >>= 1) {
        if (stride < size) {
#pragma unroll
            for (int p = 0; p < kSortChunk / 2; ++p) {
                const int lo = ((p & ~(stride - 1)) << 1) | (p & (stride - 1)), hi = lo + stride;
                const bool up = ((c0 + lo) & size) == 0;
                const K x = v[lo], y = v[hi];
                const K mn = x < y ? x : y, mx = x < y ? y : x;
                v[lo] = up ? mn : mx;
                v[hi] = up ? mx : mn;
            }
        }
    }
}

template <int BT, typename K>
__global__ __launch_bounds__(BT) void k_merge_sort(int64_t *__restrict__ keys, const int64_t *__restrict__ wave_base,
                                                    const int32_t *__restrict__ wave_cap,
                                                    const int32_t *__restrict__ nsteps, int32_t *__restrict__ ucount, int eshift)
{
    typedef SortKey<K> SK;
    constexpr int kSortChunk = SK::kChunk;
    auto spad = [](int i) { return i + (i >> SK::kShift); };
    K *sk = reinterpret_cast<K *>(ray_smem);
    __shared__ int s_cnt[BT];
    const int64_t ray = blockIdx.x;
    const int tid = threadIdx.x;
    const int ne = frechet_entries(nsteps[ray]);
    if (ne == 0) {
        if (tid == 0) ucount[ray] = 0;
        return;
    }
    int n2 = 256;
    while (n2 < ne) n2 <<= 1;
    int64_t *rk = keys + ray_slab0(wave_base[ray >> 6], wave_cap[ray >> 6], (int)(ray & 63));
    for (int e = tid; e < n2; e += BT) {
        int64_t k = -1;
        if (e < ne) k = rk[e];
        sk[spad(e)] = k < 0 ? SK::dead() : SK::pack(k, e, eshift);
    }
    __syncthreads();
    const int nchunk = n2 / kSortChunk;
    for (int c = tid; c < nchunk; c += BT) {                       // stages 2, 4, 8, 16 entirely in registers
        K v[kSortChunk];
#pragma unroll
        for (int j = 0; j < kSortChunk; ++j) v[j] = sk[c * (kSortChunk + 1) + j];
#pragma unroll
        for (int size = 2; size <= kSortChunk; size <<= 1) sort_chunk_stage<K>(v, c * kSortChunk, size);
#pragma unroll
        for (int j = 0; j < kSortChunk; ++j) sk[c * (kSortChunk + 1) + j] = v[j];
    }
    __syncthreads();
    for (int size = 2 * kSortChunk; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride >= kSortChunk; stride >>= 1) {
            const int sm = stride - 1;                                   // strides are powers of two
            for (int t = tid; t < (n2 >> 1); t += BT) {
                const int lo = ((t & ~sm) << 1) | (t & sm), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const K x = sk[spad(lo)], y = sk[spad(hi)];
                if ((x > y) == up) { sk[spad(lo)] = y; sk[spad(hi)] = x; }
            }
            __syncthreads();
        }
        for (int c = tid; c < nchunk; c += BT) {
            K v[kSortChunk];
#pragma unroll
            for (int j = 0; j < kSortChunk; ++j) v[j] = sk[c * (kSortChunk + 1) + j];
            sort_chunk_stage<K>(v, c * kSortChunk, size);
#pragma unroll
            for (int j = 0; j < kSortChunk; ++j) sk[c * (kSortChunk + 1) + j] = v[j];
        }
        __syncthreads();
    }
    // rank of each entry's node: heads counted per contiguous chunk, scanned over the BT threads
    const int cs = n2 / BT, e0 = tid * cs;
    int heads = 0;
    for (int e = e0; e < e0 + cs; ++e) {
        const K k = sk[spad(e)];
        heads += k != SK::dead() && (e == 0 || SK::node(sk[spad(e - 1)], eshift) != SK::node(k, eshift));
    }
    s_cnt[tid] = heads;
    __syncthreads();
    for (int off = 1; off < BT; off <<= 1) {
        const int v = tid >= off ? s_cnt[tid - off] : 0;
        __syncthreads();
        s_cnt[tid] += v;
        __syncthreads();
    }
    int rank = s_cnt[tid] - heads - 1;              // rank of the last head before this chunk
    if (tid == BT - 1) ucount[ray] = s_cnt[BT - 1];
    for (int e = e0; e < e0 + cs; ++e) {
        const K k = sk[spad(e)];
        if (k == SK::dead()) { if (e < ne) rk[e] = -1; continue; }
        rank += (e == 0 || SK::node(sk[spad(e - 1)], eshift) != SK::node(k, eshift));
        rk[e] = (int64_t)SK::out(k, eshift, rank);
    }
}

// sums every distinct node's contributions in emission order, applies rho, writes the ray's block row
// (ucols, uvals[4]) at uoff[ray] and the per-matrix-row entry counts that survive droptol.
__global__ __launch_bounds__(256) void k_merge_emit(const int64_t *__restrict__ keys, const double *__restrict__ vals,
                                                    const int64_t *__restrict__ wave_base, const int32_t *__restrict__ wave_cap,
                                                    const int32_t *__restrict__ nsteps,
                                                    const double *__restrict__ rho, const int64_t *__restrict__ uoff,
                                                    double droptol, int32_t *__restrict__ ucols,
                                                    double *__restrict__ uvals, int32_t *__restrict__ rowcnt)
{
    __shared__ int s_rows[4];
    const int64_t ray = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid < 4) s_rows[tid] = 0;
    __syncthreads();
    const int ne = frechet_entries(nsteps[ray]);
    const int64_t slab0 = ray_slab0(wave_base[ray >> 6], wave_cap[ray >> 6], (int)(ray & 63));
    const int64_t *rk = keys + slab0;
    const double *rv = vals + slab0 * 4;
    const int64_t off = uoff[ray];
    double R[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) R[k] = rho[ray * 16 + k];
    int cnt[4] = {0, 0, 0, 0};
    for (int e = tid; e < ne; e += 256) {
        const uint64_t k = (uint64_t)rk[e];
        if (k == kKeyDead) continue;
        const uint64_t node = k >> 40;
        if (e > 0 && ((uint64_t)rk[e - 1] >> 40) == node) continue;                          // not a head
        double s[4] = {0, 0, 0, 0};
        uint64_t k2 = k;
        int e2 = e;
        bool first = true;
        do {
            const double2 *p2 = reinterpret_cast<const double2 *>(rv + (size_t)((k2 >> 20) & 0xFFFFF) * 4);
            const double2 lo = p2[0], hi = p2[1];
            if (first) { s[0] = lo.x; s[1] = lo.y; s[2] = hi.x; s[3] = hi.y; first = false; }
            else { s[0] += lo.x; s[1] += lo.y; s[2] += hi.x; s[3] += hi.y; }                 // Phi += Theta (:159,175)
            ++e2;
            k2 = e2 < ne ? (uint64_t)rk[e2] : kKeyDead;
        } while (k2 != kKeyDead && (k2 >> 40) == node);
        const int64_t u = off + (int64_t)(k & 0xFFFFF);
        double o4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                                       // rho * Phi (:184)
            o4[i] = ((R[i * 4] * s[0] + R[i * 4 + 1] * s[1]) + R[i * 4 + 2] * s[2]) + R[i * 4 + 3] * s[3];
            cnt[i] += (droptol < 0) || (__builtin_fabs(o4[i]) > droptol);                   // droptol! (:242)
        }
        ucols[u] = (int32_t)node;
        double2 *dst = reinterpret_cast<double2 *>(uvals + u * 4);
        dst[0] = make_double2(o4[0], o4[1]);
        dst[1] = make_double2(o4[2], o4[3]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (cnt[i]) atomicAdd(&s_rows[i], cnt[i]);
    __syncthreads();
    if (tid < 4) rowcnt[ray * 4 + tid] = s_rows[tid];
}

// CSR rows 4*ray + r from the ray's block row; one wave per (ray, r), order kept (columns sorted).
__global__ __launch_bounds__(256) void k_csr_emit(const int32_t *__restrict__ ucols, const double *__restrict__ uvals,
                                                  const int64_t *__restrict__ uoff, const int64_t *__restrict__ indptr,
                                                  double droptol, int32_t *__restrict__ indices, double *__restrict__ data)
{
    const int64_t ray = blockIdx.x;
    const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t u0 = uoff[ray], u1 = uoff[ray + 1];
    int64_t w = indptr[ray * 4 + r];
    for (int64_t u = u0; u < u1; u += 64) {
        const bool in = u + lane < u1;
        const double v = in ? uvals[(u + lane) * 4 + r] : 0.0;
        const bool keep = in && ((droptol < 0) || (__builtin_fabs(v) > droptol));
        const uint64_t mask = __ballot(keep);
        if (keep) {
            const int64_t pos = w + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            indices[pos] = ucols[u + lane];
            data[pos] = v;
        }
        w += __popcll(mask);
    }
}

// ---- exclusive scan of int32 counts into int64 offsets (out has n + 1 entries) -----------------------
constexpr int kScanItems = 16;       // per thread; a block of 256 covers 4096 counts

__global__ __launch_bounds__(256) void k_scan_sums(const int32_t *__restrict__ in, int64_t n, int64_t *__restrict__ bsum)
{
    __shared__ long long s[256];
    const int64_t b0 = (int64_t)blockIdx.x * 256 * kScanItems + (int64_t)threadIdx.x * kScanItems;
    long long t = 0;
    for (int k = 0; k < kScanItems; ++k)
        if (b0 + k < n) t += in[b0 + k];
    s[threadIdx.x] = t;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(256) void k_scan_blocks(int64_t *__restrict__ bsum, int64_t nb, int64_t *__restrict__ total)
{
    __shared__ long long s[256];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < nb; c0 += 256) {
        const int64_t i = c0 + threadIdx.x;
        const long long v = i < nb ? bsum[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const long long x = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += x;
            __syncthreads();
        }
        if (i < nb) bsum[i] = carry + s[threadIdx.x] - v;        // exclusive
        __syncthreads();
        if (threadIdx.x == 255) carry += s[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const int32_t *__restrict__ in, int64_t n, const int64_t *__restrict__ bsum,
                                                    const int64_t *__restrict__ total, int64_t *__restrict__ out)
{
    __shared__ long long s[256];
    const int64_t b0 = (int64_t)blockIdx.x * 256 * kScanItems + (int64_t)threadIdx.x * kScanItems;
    long long loc[kScanItems], t = 0;
    for (int k = 0; k < kScanItems; ++k) {
        loc[k] = t;
        if (b0 + k < n) t += in[b0 + k];
    }
    s[threadIdx.x] = t;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const long long x = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
        __syncthreads();
        s[threadIdx.x] += x;
        __syncthreads();
    }
    const long long pre = bsum[blockIdx.x] + s[threadIdx.x] - t;
    for (int k = 0; k < kScanItems; ++k)
        if (b0 + k < n) out[b0 + k] = pre + loc[k];
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

// indptr of a row block appended behind `add` earlier non-zeros (chunked assembly)
__global__ __launch_bounds__(256) void k_indptr_shift(int64_t *__restrict__ dst, const int64_t *__restrict__ src, int64_t n, int64_t add)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + add;
}

// ---- NEXT-2: the reconstruction loop's passes over M (NonObstacle.jl:340-367) on the device CSR ----------
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// dofs[j] = nnz(M[4j-3,:]) - sum(fidelity[find(M[4j-3,:])])        NonObstacle.jl:340-342; one wave per ray
__global__ __launch_bounds__(256) void k_csr_dofs(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                  const double *__restrict__ fidelity, int64_t num, double *__restrict__ dofs)
{
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= num) return;
    const int lane = threadIdx.x & 63;
    const int64_t b = indptr[ray * 4], e = indptr[ray * 4 + 1];
    double s = 0;
    for (int64_t k = b + lane; k < e; k += 64) s += fidelity[indices[k]];
    s = wave_sum_f64(s);
    if (lane == 0) dofs[ray] = (double)(e - b) - s;
}

// y = M x; one wave per row
__global__ __launch_bounds__(256) void k_csr_matvec(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                    const double *__restrict__ data, const double *__restrict__ x,
                                                    int64_t nrows, double *__restrict__ y)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int lane = threadIdx.x & 63;
    double s = 0;
    for (int64_t k = indptr[row] + lane; k < indptr[row + 1]; k += 64) s = __builtin_fma(data[k], x[indices[k]], s);
    s = wave_sum_f64(s);
    if (lane == 0) y[row] = s;
}

// b += M' (mask .* v): the rows of the selected rays scattered into b (NonObstacle.jl:354); fp64 atomics
__global__ __launch_bounds__(256) void k_csr_matvec_t(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                      const double *__restrict__ data, const double *__restrict__ v,
                                                      const uint8_t *__restrict__ ray_mask, int64_t nrows, double *__restrict__ b)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    if (ray_mask && !ray_mask[row >> 2]) return;
    const int lane = threadIdx.x & 63;
    const double vr = v[row];
    for (int64_t k = indptr[row] + lane; k < indptr[row + 1]; k += 64) unsafeAtomicAdd(&b[indices[k]], data[k] * vr);
}

// The normal operator of the selected rays without forming it (the reference forms MtM = M[order,Idx]' * M[order,Idx],
// NonObstacle.jl:353; at N = 1024 that product is out of reach, a Krylov solve needs only its action):
//   w += M[order,:]' (M[order,:] x), one wave per row: dot with x, then scatter row * dot (the row comes back from L2)
__global__ __launch_bounds__(256) void k_csr_normal_matvec(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                           const double *__restrict__ data, const double *__restrict__ x,
                                                           const uint8_t *__restrict__ ray_mask, int64_t nrows, double *__restrict__ w)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    if (ray_mask && !ray_mask[row >> 2]) return;
    const int lane = threadIdx.x & 63;
    const int64_t b = indptr[row], e = indptr[row + 1];
    double s = 0;
    for (int64_t k = b + lane; k < e; k += 64) s = __builtin_fma(data[k], x[indices[k]], s);
    s = wave_sum_f64(s);
    for (int64_t k = b + lane; k < e; k += 64) unsafeAtomicAdd(&w[indices[k]], data[k] * s);
}

//   d[j] += sum over the selected rows of M[k,j]^2 = diag(MtM): the Jacobi preconditioner of that solve
__global__ __launch_bounds__(256) void k_csr_normal_diag(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                         const double *__restrict__ data, const uint8_t *__restrict__ ray_mask,
                                                         int64_t nrows, double *__restrict__ d)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    if (ray_mask && !ray_mask[row >> 2]) return;
    const int lane = threadIdx.x & 63;
    for (int64_t k = indptr[row] + lane; k < indptr[row + 1]; k += 64) unsafeAtomicAdd(&d[indices[k]], data[k] * data[k]);
}

__device__ __forceinline__ void atomic_max_f64(double *addr, double v)
{
    unsigned long long *a = reinterpret_cast<unsigned long long *>(addr);
    unsigned long long old = *a;
    while (__longlong_as_double((long long)old) < v) {
        const unsigned long long seen = atomicCAS(a, old, (unsigned long long)__double_as_longlong(v));
        if (seen == old) break;
        old = seen;
    }
}

// residual[k] = |M[k,:] x - mismatch[k]| for the rows of the selected rays; where residual < rejection,
// fidelity[find(M[k,:])] = max(fidelity, 1 - decay * residual)     NonObstacle.jl:359-367
__global__ __launch_bounds__(256) void k_csr_fidelity(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                      const double *__restrict__ data, const double *__restrict__ x,
                                                      const double *__restrict__ mismatch, const uint8_t *__restrict__ ray_mask,
                                                      double rejection, double decay, int64_t nrows,
                                                      double *__restrict__ fidelity, double *__restrict__ residual)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int lane = threadIdx.x & 63;
    if (ray_mask && !ray_mask[row >> 2]) {
        if (residual && lane == 0) residual[row] = 0.0;
        return;
    }
    const int64_t b = indptr[row], e = indptr[row + 1];
    double s = 0;
    for (int64_t k = b + lane; k < e; k += 64) s = __builtin_fma(data[k], x[indices[k]], s);
    s = wave_sum_f64(s);
    const double res = __builtin_fabs(s - mismatch[row]);
    if (residual && lane == 0) residual[row] = res;
    if (res < rejection) {
        const double f = 1 - decay * res;
        for (int64_t k = b + lane; k < e; k += 64) atomic_max_f64(&fidelity[indices[k]], f);
    }
}

} // namespace ray
