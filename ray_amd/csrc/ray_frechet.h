// ray_frechet.h -- NEXT-1 of SURVEY.md section 8f: the Jacobian rho and the Frechet matrix M of
// ScatterForwardOperator (NonObstacle.jl:6-28 DiscreteHamilton with Hessian, :107-139 dV,
// :141-187 ChunkProcessing!, :212-218 hess table, :242 sparse + droptol).
//
// The reference builds, per ray and per RK4 step, three 4 x N^2 sparse matrices (O(N^2) each) and a
// dense 4num x N^2 SharedArray that it then compresses.  Here the matrix is built on the device:
//   k_frechet      one thread integrates (X, rho) of one ray and leaves ONE 128-byte record per trajectory
//                  node in its ray's own contiguous stretch of the trajectory slab -- the three 4-vectors
//                  P, Q, S that Theta = inv(rho) dV w factors into, the in-cell offsets and the cell -- plus
//                  the cell alone in a compact [wave][step][lane] array for the sort;
//   k_seg_sort     one workgroup per ray sorts the four stencil SEGMENTS of every record -- runs of consecutive nodes,
//                  derived from the cell -- by first node in LDS, counts the distinct nodes (what the segments own) and
//                  hands over 2 bytes per segment (its id in sorted order);
//   k_seg_emit     one distinct node per thread: rebuilds the Theta columns that reach it from their records (held in
//                  LDS), sums them in sorted order, applies the final rho (M rows = rho * Phi, :184) and writes the
//                  ray's BLOCK ROW: one column index and the four values of rows 4 ray .. 4 ray + 3 per distinct node
//                  (36 B instead of 48 B of CSR); it also counts, per matrix row, the entries that survive droptol.
//   (k_merge_sort / k_merge_emit: the round-2 form of the same two steps, sorting the twelve entries of every record;
//   kept behind RAY_OPT_TUNE_MERGE_ENTRIES as a cross-check.)
// The block rows are the device-resident form of M: the loop passes (k_br_*) apply droptol on the fly.
//   k_csr_emit     converts to CSR (rows ray-major like reshape(M, 4num, N^2), columns sorted) on export only.
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
    double2 hs[6];            // the cell's hess record
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
    {   // the hess record of the same cell, asked for with the first one: both gathers in flight together
        // (hess exists for I,J <= N-3 only: clamp here, flag in jac_from_point)
        const int ih = q.i < g.N - 4 ? q.i : g.N - 4, jh = q.j < g.N - 4 ? q.j : g.N - 4;
        const double2 *hs = reinterpret_cast<const double2 *>(g.hrec) + ((size_t)ih + (size_t)jh * (size_t)(g.N - 1)) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) q.hs[k] = hs[k];
    }
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
    const double2 *hs = q.hs;
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

// One RK4 stage of the Jacobian equation V = Jm * Rin (Rin = rho at stage 1, the stage argument R afterwards):
//   acc (+)= wa * V, R = rho + c * V, and at stage 4 rho = rho + c * (acc + V)                      NonObstacle.jl:161-172
// with rho and the RK4 accumulator in LDS (k_frechet: two waves per SIMD need the 64 VGPRs the two
// 4x4 arrays would take).  Element pair (2m, 2m+1) of lane l sits at (m * 64 + l) * 16 from `rho_a` / `acc_a`
// (absolute LDS byte addresses of this lane's column: + m * 1024), so every access is a conflict-free b128.
// Two columns at a time, each column's four entries of V consumed before R's column is overwritten; element for
// element the operations of a 4x4 product and the two updates.  R (the stage argument) stays in registers.
__device__ __forceinline__ void lds_st_d2(unsigned addr, double x, double y)
{
    lds_d2v v;
    v.x = x; v.y = y;
    *RAY_LDS_PTR(lds_d2v, addr) = v;
}
template <int MODE>   // 0: acc = V, R = rho + c V; 1: acc += 2 V, R = rho + c V; 2: rho = rho + c (acc + V) where `write`
__device__ __forceinline__ void rho_stage_lds(const double Jm[16], double R[16], unsigned rho_a, unsigned acc_a, double c, bool write)
{
#pragma unroll
    for (int jp = 0; jp < 4; jp += 2) {
        double2 rb[4];                                      // rho rows 0..3, columns jp, jp + 1
#pragma unroll
        for (int k = 0; k < 4; ++k) rb[k] = lds_ld_d2(rho_a + (unsigned)((k * 4 + jp) / 2) * 1024u);
        double v[4][2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = jp + cc;
                double t;
                if (MODE == 0) {
                    t = Jm[i * 4] * (cc ? rb[0].y : rb[0].x);
#pragma unroll
                    for (int k = 1; k < 4; ++k) t = __builtin_fma(Jm[i * 4 + k], cc ? rb[k].y : rb[k].x, t);
                } else {
                    t = Jm[i * 4] * R[j];
#pragma unroll
                    for (int k = 1; k < 4; ++k) t = __builtin_fma(Jm[i * 4 + k], R[k * 4 + j], t);
                }
                v[i][cc] = t;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned off = (unsigned)((i * 4 + jp) / 2) * 1024u;
            if (MODE == 0) {
                lds_st_d2(acc_a + off, v[i][0], v[i][1]);
                R[i * 4 + jp] = __builtin_fma(v[i][0], c, rb[i].x);
                R[i * 4 + jp + 1] = __builtin_fma(v[i][1], c, rb[i].y);
            } else if (MODE == 1) {
                const double2 ac = lds_ld_d2(acc_a + off);
                lds_st_d2(acc_a + off, __builtin_fma(2.0, v[i][0], ac.x), __builtin_fma(2.0, v[i][1], ac.y));
                R[i * 4 + jp] = __builtin_fma(v[i][0], c, rb[i].x);
                R[i * 4 + jp + 1] = __builtin_fma(v[i][1], c, rb[i].y);
            } else {
                const double2 ac = lds_ld_d2(acc_a + off);
                if (write) lds_st_d2(rho_a + off, __builtin_fma(ac.x + v[i][0], c, rb[i].x), __builtin_fma(ac.y + v[i][1], c, rb[i].y));
            }
        }
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
    double cease_s2;            // the largest x^2 + y^2 whose rounded root is <= cease_r
    const int64_t *wave_base;   // record-row offset of every wave (a wave owns 64 x rows records from 64 * base)
    const int32_t *wave_cap;    // step capacity of every wave (rows = cap + 1 records per ray)
    const int32_t *perm;        // lane l of wave w integrates ray perm[64 w + l] (rays regrouped by step count, longest first)
    double *traj;               // [wave][lane][row][16] trajectory records (kTrajDoubles)
    int32_t *cells;             // [wave][row][lane] cell of every trajectory node: i | j << 16
    double *X;                  // (num, 4) end states
    int64_t ldx;
    double *rho;                // (num, 16) final Jacobian, row-major
    int32_t *status, *nsteps;
    unsigned long long *step_counter;
};

// Trajectory record: Theta (4 x 12) = inv(rho) dV(X) w is a combination of three 4-vectors with the
// twelve bilinear weights of the node's cell (NonObstacle.jl:107-139 has only those three column patterns):
//   P = inv * [2c px, 2c py, -tau gcX, -tau gcY] * w,  Q = inv[:,2] * (c tau / 2h) * w,  S = inv[:,3] * (c tau / 2h) * w
//   cell nodes k = 0..3:  phi_k P + psi_k Q + gam_k S;  x-neighbours: -psi_k Q;  y-neighbours: -gam_k S
// [ P | Q | S | u v cell - ]: 128 bytes per node instead of 12 x 40 bytes of (key, Theta column).
constexpr int kTrajDoubles = 16;
constexpr int kTrajBytesPerRow = kTrajDoubles * 8 + kFrSlotsPerCall * 4 + 4;   // record + sorted entries + cell
constexpr uint32_t kSrtDead = 0xffffffffu;  // sorted entry: rank << 15 | head << 14 | emission index
constexpr int kMergeMaxEntries = 16384;     // per ray (emission index: 14 bits; 64 KiB of 32-bit LDS keys)
constexpr uint64_t kKeyDead = ~0ull;        // wide sort key: node << 40 | emission index << 20

__host__ __device__ inline int frechet_rows(int steps) { return steps > 0 ? steps + 1 : 0; }
__host__ __device__ inline int frechet_entries(int steps) { return frechet_rows(steps) * kFrSlotsPerCall; }

// first record row of ray `lane` of a wave whose rays hold frechet_rows(cap) records each
__device__ __forceinline__ int64_t ray_row0(int64_t wave_base, int cap, int lane)
{
    return wave_base * 64 + (int64_t)lane * frechet_rows(cap);
}

// stencil node k of cell (i, j): cell nodes, x-neighbours (row 3 of dV), y-neighbours (row 4)
//   di = {0, 1, 0, 1, 2, -1, 2, -1, 0, 1, 0, 1},  dj = {0, 0, 1, 1, 0, 0, 1, 1, 2, 2, -1, -1}
// packed two bits each as offset + 1 (register constants: no table look-up by a runtime k)
__device__ __forceinline__ int stencil_di(int k) { return (int)((0x993399u >> (2 * k)) & 3u) - 1; }
__device__ __forceinline__ int stencil_dj(int k) { return (int)((0x0fa5a5u >> (2 * k)) & 3u) - 1; }

// One trajectory record per live lane -> row `it` of the lane's ray (ALL 64 lanes of the one-wave workgroup
// together).  A ray's records are contiguous in the slab, so lane-wise stores would touch 64 lines per
// instruction: the wave transposes through LDS (64 x 144 B, rows padded by 16 B against bank conflicts) and
// stores 16-byte pieces, 8 consecutive lanes on the 8 pieces of one ray's 128-byte record = one full line.
// Finished lanes' records are not stored: nobody reads them.
// LDS of the one-wave workgroup: [ node table, padded to 16 B | rho: 8 planes of 64 x 16 B | RK4 accumulator of rho,
// same shape ] -- the transposition's staging area lies over the accumulator (dead between stage 4 of a step and
// stage 1 of the next, which is when a node is emitted) and 1 KB beyond it.
constexpr int kFrStageBytes = 64 * 144;
constexpr int kFrRhoBytes = 8 * 64 * 16;
__host__ __device__ inline size_t frechet_nodes_bytes(int N, bool nodes) { return nodes ? ((size_t)N * 8 + 15) & ~(size_t)15 : 0; }   // kExact only
__host__ __device__ inline size_t frechet_lds_bytes(int N, bool nodes) { return frechet_nodes_bytes(N, nodes) + kFrRhoBytes + kFrStageBytes; }

template <int AR>
__device__ __forceinline__ void emit_node(const FrechetArgs &a, const double X[4], const PointEval &q, int o,
                                          unsigned rho_a, double w,
                                          bool live, int64_t wave_base, int64_t wave_row0, int rows, int it, unsigned lane, int &oob)
{
    double rho[16], inv[16];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const double2 p2 = lds_ld_d2(rho_a + (unsigned)m * 1024u);
        rho[2 * m] = p2.x; rho[2 * m + 1] = p2.y;
    }
    mat4_inv(rho, inv);
    const GridDev &g = a.med.g;
    const double h = g.h;
    const double u = q.dx / h, v = q.dy / h;
    const double ctw = (q.c * q.tau / (2 * h)) * w;
    const double a0 = (2 * q.c * X[2]) * w, a1 = (2 * q.c * X[3]) * w, a2 = (-q.tau * q.gcX) * w, a3 = (-q.tau * q.gcY) * w;
    const int N = g.N;
    // the stencil reaches i-1 .. i+2, j-1 .. j+2: nodes off the grid are dropped and flagged
    o |= (q.i < 1) | (q.i + 2 >= N) | (q.j < 1) | (q.j + 2 >= N);
    double2 *stage = reinterpret_cast<double2 *>(ray_smem + frechet_nodes_bytes(N, AR == kExact) + kFrRhoBytes);   // [lane][9]
    double2 *mine = stage + lane * 9;
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        const double p0 = ((inv[r * 4] * a0 + inv[r * 4 + 1] * a1) + inv[r * 4 + 2] * a2) + inv[r * 4 + 3] * a3;
        const double p1 = ((inv[r * 4 + 4] * a0 + inv[r * 4 + 5] * a1) + inv[r * 4 + 6] * a2) + inv[r * 4 + 7] * a3;
        mine[r / 2] = make_double2(p0, p1);
        mine[2 + r / 2] = make_double2(inv[r * 4 + 2] * ctw, inv[r * 4 + 6] * ctw);
        mine[4 + r / 2] = make_double2(inv[r * 4 + 3] * ctw, inv[r * 4 + 7] * ctw);
    }
    const int cellv = q.i | (q.j << 16);
    mine[6] = make_double2(u, v);
    mine[7] = make_double2(__longlong_as_double((long long)cellv), 0.0);
    a.cells[(wave_base + it) * 64 + lane] = live ? cellv : -1;
    const unsigned long long lm = __ballot(live);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; ++t) {                                         // 512 pieces, 8 per ray
        const unsigned qq = t * 64 + lane, ray = qq >> 3, part = qq & 7;
        if ((lm >> ray) & 1) {
            const double2 piece = stage[ray * 9 + part];
            *reinterpret_cast<double2 *>(a.traj + (wave_row0 + (int64_t)ray * rows + it) * kTrajDoubles + part * 2) = piece;
        }
    }
    __syncthreads();
    if (live) oob |= o;
}

// ChunkProcessing! for one ray per thread                         NonObstacle.jl:146-185
// The reference adds Theta twice per step (:158 before, :174 after) -- the "after" point of step n and
// the "before" point of step n+1 are the same (X, rho), so one emission per trajectory node carries
// both weights (trapezoid rule; the cease term :178 joins the last node's weight).
#ifdef RAY_TIMELINE
__device__ unsigned long long g_fr_timeline[3 * 65536];   // instrumentation build: start / end clock, hardware id per workgroup
#endif
template <int AR>
__global__ __launch_bounds__(64, 2) void k_frechet(const FrechetArgs a)
{
#ifdef RAY_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < 65536) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_fr_timeline[3 * blockIdx.x] = wall_clock64();
        g_fr_timeline[3 * blockIdx.x + 2] = hw;
    }
#endif
    const int64_t wave = blockIdx.x;
    const unsigned lane = threadIdx.x & 63;
    const bool valid = wave * 64 + lane < a.num;
    const int64_t r = valid ? a.perm[wave * 64 + lane] : 0;
    if (AR == kExact) {   // node table into LDS (the look-ups read it from there; kFused takes its offsets from the cell coordinate)
        double *p_lds = reinterpret_cast<double *>(ray_smem);
        for (int k = threadIdx.x; k < a.med.g.N; k += blockDim.x) p_lds[k] = a.med.g.p[k];
        __syncthreads();
    }
    const int cap = a.wave_cap[wave];
    const int64_t wave_base = a.wave_base[wave], wave_row0 = ray_row0(wave_base, cap, 0);
    const int rows = frechet_rows(cap);
    double X[4] = {0, 0, 1, 0}, T = 0;
    const unsigned rho_a = (unsigned)(uintptr_t)(lptr_t)ray_smem + (unsigned)frechet_nodes_bytes(a.med.g.N, AR == kExact) + lane * 16u;
    const unsigned acc_a = rho_a + kFrRhoBytes;
#pragma unroll
    for (int m = 0; m < 8; ++m) lds_st_d2(rho_a + (unsigned)m * 1024u, (m == 0 || m == 5) ? 1.0 : 0.0, (m == 2 || m == 7) ? 1.0 : 0.0);
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
        emit_node<AR>(a, X, q0, o0, rho_a, wp + dtl * 0.5, active || fin, wave_base, wave_row0, rows, it, lane, oob);
        fin = false;
        if (!anyact) break;                                                  // that was the last node
        // RK4 for X (bit-identical to k_trace) and rho                       :161-172
        double H[4], Y[4], accx[4], Jm[16], R[16];
        int o = o0;
        const double hdt = 0.5 * dtl;
        jac_from_point(a.med.g, X, q0, Jm, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) { accx[k] = k1[k]; Y[k] = (AR == kFused) ? __builtin_fma(k1[k], hdt, X[k]) : X[k] + k1[k] * hdt; }
        rho_stage_lds<0>(Jm, R, rho_a, acc_a, hdt, true);
        hamilton_jac<AR>(a.med.g, Y, H, Jm, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) { accx[k] = __builtin_fma(2.0, H[k], accx[k]); Y[k] = (AR == kFused) ? __builtin_fma(H[k], hdt, X[k]) : X[k] + H[k] * hdt; }
        rho_stage_lds<1>(Jm, R, rho_a, acc_a, hdt, true);
        hamilton_jac<AR>(a.med.g, Y, H, Jm, o);
#pragma unroll
        for (int k = 0; k < 4; ++k) { accx[k] = __builtin_fma(2.0, H[k], accx[k]); Y[k] = (AR == kFused) ? __builtin_fma(H[k], dtl, X[k]) : X[k] + H[k] * dtl; }
        rho_stage_lds<1>(Jm, R, rho_a, acc_a, dtl, true);
        hamilton_jac<AR>(a.med.g, Y, H, Jm, o);
        const double dt6 = dtl * 0.16666666666666666;
        if (active) {                          // a finished lane keeps (X, rho) untouched whatever its stages evaluated to
#pragma unroll
            for (int k = 0; k < 4; ++k)
                X[k] = (AR == kFused) ? __builtin_fma(accx[k] + H[k], dt6, X[k]) : X[k] + div6((accx[k] + H[k]) * dtl);
        }
        rho_stage_lds<2>(Jm, R, rho_a, acc_a, dt6, active);
        ++it;
        wp = 0.0;
        if (active) {
            steps = it;
            wp = dtl * 0.5;                                                  // :174-175
            const double s = X[0] * X[0] + X[1] * X[1];
            oob |= o | !(s == s);
            if (s > a.cease_s2) { wp = wp + (T - t); t = t + (T - t); st |= kStCeased; }   // :177-181 (norm > 1 + 2h without the root)
            active = t < T;
            fin = !active;
        }
    }
    if (oob) st |= kStOob;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a.X[r + k * a.ldx] = X[k];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const double2 p2 = lds_ld_d2(rho_a + (unsigned)m * 1024u);
            reinterpret_cast<double2 *>(a.rho + r * 16)[m] = p2;
        }
        a.status[r] = st;
        a.nsteps[r] = steps;
    }
    const unsigned wsum = wave_sum((unsigned)steps);
    if (lane == 0 && a.step_counter) atomicAdd(a.step_counter, (unsigned long long)wsum);
#ifdef RAY_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < 65536) g_fr_timeline[3 * blockIdx.x + 1] = wall_clock64();
#endif
}

#ifdef RAY_FR_PHASES
// instrumentation build: summed cycles per phase, k_merge_emit [0] workgroups, [1..4] meta / copy / tiles / tail;
// k_merge_sort [8] workgroups, [9..12] meta + cells / keys + first 16 / rounds / ranking
__device__ unsigned long long g_fr_phase[16];
#define RAY_FR_MARK(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_fr_phase[k], t_ - t_mark); t_mark = t_; } } while (0)
#else
#define RAY_FR_MARK(k) do { } while (0)
#endif

// ---- merge: distinct nodes of one ray --------------------------------------------------------------
// One workgroup (two waves) per ray.  Keys (node, emission index) are sorted in LDS (bitonic, n2 = next power
// of two); every entry gets the rank of its node among the ray's distinct nodes and goes back to memory as
// rank << 15 | head << 14 | emission index; ucount[ray] = number of distinct nodes.
// The network runs in ROUNDS: a thread takes SIXTEEN keys whose indices differ in four bits b..b+3, does the
// compare-exchanges of up to four consecutive strides in registers and puts them back -- one LDS round trip per
// four strides (17 rounds after the in-register start at n2 = 2048; the instruction count per key is what bounds
// this kernel).  All-ascending form of the network: the first pass of a stage pairs i with its MIRROR image in
// the block (i ^ (size - 1)), the rest are plain half-cleaners, so an exchange is a min / max pair without a
// direction.  Keys sit padded by one per 16: the sixteen-key groups of consecutive threads then fall on distinct
// banks for every b.
constexpr int kSortThreads = 128;
__host__ __device__ inline size_t merge_sort_lds_bytes(int n2, size_t key_bytes)
{
    return (size_t)(n2 + n2 / 16) * key_bytes;
}

// Sort keys.  64-bit: node << 40 | emission << 20.  32-bit (chosen by the host when the node index and
// the emission index fit together: N^2 < 2^(32 - eshift), n2 <= 2^eshift): node << eshift | emission --
// half the LDS traffic, one-instruction compares, min/max exchanges.
template <typename K> struct SortKey;
template <> struct SortKey<uint64_t> {
    static __device__ __forceinline__ uint64_t dead() { return kKeyDead; }
    static __device__ __forceinline__ uint64_t pack(int64_t node, int e, int) { return (uint64_t)node << 40 | (uint64_t)e << 20; }
    static __device__ __forceinline__ uint64_t node(uint64_t k, int) { return k >> 40; }
    static __device__ __forceinline__ uint32_t emission(uint64_t k, int) { return (uint32_t)(k >> 20) & 0xFFFFFu; }
};
template <> struct SortKey<uint32_t> {
    static __device__ __forceinline__ uint32_t dead() { return 0xffffffffu; }
    static __device__ __forceinline__ uint32_t pack(int64_t node, int e, int es) { return (uint32_t)node << es | (uint32_t)e; }
    static __device__ __forceinline__ uint64_t node(uint32_t k, int es) { return k >> es; }
    static __device__ __forceinline__ uint32_t emission(uint32_t k, int es) { return k & ((1u << es) - 1u); }
};

// ascending compare-exchange: v_min_u32 + v_max_u32 for the narrow keys
template <typename K>
__device__ __forceinline__ void cmp_exchange(K &x, K &y)
{
    const K mn = x < y ? x : y, mx = x < y ? y : x;
    x = mn;
    y = mx;
}

// half-cleaners of the strides 2^(nstr-1) .. 1 (in units of the group's own spacing) on sixteen keys
template <typename K>
__device__ __forceinline__ void sort_group_strides(K v[16], int nstr)
{
#pragma unroll
    for (int sb = 3; sb >= 0; --sb) {
        if (sb < nstr) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int lo = ((p >> sb) << (sb + 1)) | (p & ((1 << sb) - 1));
                cmp_exchange<K>(v[lo], v[lo | (1 << sb)]);
            }
        }
    }
}

// stages 2, 4, 8, 16 on sixteen consecutive keys
template <typename K>
__device__ __forceinline__ void sort_sixteen(K v[16])
{
#pragma unroll
    for (int p = 0; p < 16; p += 2) cmp_exchange<K>(v[p], v[p + 1]);                 // size 2
#pragma unroll
    for (int q = 0; q < 16; q += 4) { cmp_exchange<K>(v[q], v[q + 3]); cmp_exchange<K>(v[q + 1], v[q + 2]); }   // size 4: mirror
    sort_group_strides<K>(v, 1);
#pragma unroll
    for (int q = 0; q < 16; q += 8)                                                   // size 8: mirror
#pragma unroll
        for (int p = 0; p < 4; ++p) cmp_exchange<K>(v[q + p], v[q + 7 - p]);
    sort_group_strides<K>(v, 2);
#pragma unroll
    for (int p = 0; p < 8; ++p) cmp_exchange<K>(v[p], v[15 - p]);                     // size 16: mirror
    sort_group_strides<K>(v, 3);
}

template <int BT, typename K>
__global__ __launch_bounds__(BT) void k_merge_sort(const int32_t *__restrict__ cells, uint32_t *__restrict__ srt,
                                                    const int64_t *__restrict__ wave_base,
                                                    const int32_t *__restrict__ wave_cap, const int32_t *__restrict__ perm,
                                                    const int32_t *__restrict__ nsteps, int32_t *__restrict__ ucount, int eshift, int N)
{
    typedef SortKey<K> SK;
    auto spad = [](int i) { return i + (i >> 4); };
    K *sk = reinterpret_cast<K *>(ray_smem);
    __shared__ int s_cnt[BT / 64];
    __shared__ int s_cell[kMergeMaxEntries / kFrSlotsPerCall + 3];
    const int tid = threadIdx.x;
#ifdef RAY_FR_PHASES
    unsigned long long t_mark = __builtin_readcyclecounter();
    if (tid == 0) atomicAdd(&g_fr_phase[8], 1ull);
#endif
    const int64_t slot = blockIdx.x;                    // where k_frechet integrated the ray: wave slot >> 6, lane slot & 63
    const int64_t ray = perm[slot];                     // (workgroups in slot order: the 64 rays of a wave share their cell lines)
    const int nrow = frechet_rows(nsteps[ray]);
    const int ne = nrow * kFrSlotsPerCall;
    if (ne == 0) {
        if (tid == 0) ucount[ray] = 0;
        return;
    }
    int n2 = 256;
    while (n2 < ne) n2 <<= 1;
    RAY_FR_MARK(13);
    const int64_t wb = wave_base[slot >> 6];
    const int lane = (int)(slot & 63);
    uint32_t *rk = srt + ray_row0(wb, wave_cap[slot >> 6], lane) * kFrSlotsPerCall;
    for (int r = tid; r < nrow + 2; r += BT) s_cell[r] = r < nrow ? cells[(wb + r) * 64 + lane] : 0;
    __syncthreads();
    RAY_FR_MARK(9);
    const int ngroup = n2 >> 4;
    for (int t = tid; t < ngroup; t += BT) {                       // keys of sixteen consecutive entries; stages 2 .. 16
        K v[16];
        // sixteen consecutive entries lie in at most three records: slot k0 + j of record r0, or the next ones'
        const int e0 = t * 16;
        const int r0 = e0 < ne ? (int)(((unsigned)e0 * 43691u) >> 19) : 0, k0 = e0 - r0 * kFrSlotsPerCall;   // e0 / 12 for e0 < 2^16
        const int cva = s_cell[r0], cvb = s_cell[r0 + 1], cvc = s_cell[r0 + 2];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int kk = k0 + j;
            const int ro = kk >= 2 * kFrSlotsPerCall ? 2 : kk >= kFrSlotsPerCall ? 1 : 0;
            const int k = kk - ro * kFrSlotsPerCall;
            const int cv = ro == 2 ? cvc : ro == 1 ? cvb : cva;
            const int ni = (cv & 0xffff) + stencil_di(k), nj = (cv >> 16) + stencil_dj(k);
            const bool ok = e0 + j < ne && (unsigned)ni < (unsigned)N && (unsigned)nj < (unsigned)N;
            const K key = SK::pack((int64_t)ni + (int64_t)nj * N, e0 + j, eshift);
            v[j] = ok ? key : SK::dead();
        }
        sort_sixteen<K>(v);
#pragma unroll
        for (int j = 0; j < 16; ++j) sk[t * 17 + j] = v[j];
    }
    __syncthreads();
    RAY_FR_MARK(10);
    for (int top0 = 4; (1 << top0) < n2; ++top0) {                 // stage size = 2^(top0 + 1)
        {   // first round: the mirror pass and the strides 2^(top0-1) .. 2^(top0-3); a thread's eight upper keys sit at
            // the complemented low bits, which puts every mirror pair (j, 15 - j) into one thread
            const int b = top0 - 3, lm = (1 << b) - 1;
            unsigned d[16];                                         // uniform: spad(j << b) in bytes
#pragma unroll
            for (int j = 0; j < 16; ++j) d[j] = (unsigned)sizeof(K) * (unsigned)spad(j << b);
            for (int t = tid; t < ngroup; t += BT) {
                const int hi = (t >> b) << (b + 4), lo = t & lm;
                const unsigned a_lo = (unsigned)sizeof(K) * (unsigned)spad(hi | lo), a_hi = (unsigned)sizeof(K) * (unsigned)spad(hi | (lo ^ lm));
                char *base = reinterpret_cast<char *>(sk);
                K v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<K *>(base + (j < 8 ? a_lo : a_hi) + d[j]);
#pragma unroll
                for (int p = 0; p < 8; ++p) cmp_exchange<K>(v[p], v[15 - p]);
                sort_group_strides<K>(v, 3);
#pragma unroll
                for (int j = 0; j < 16; ++j) *reinterpret_cast<K *>(base + (j < 8 ? a_lo : a_hi) + d[j]) = v[j];
            }
            __syncthreads();
        }
        for (int top = top0 - 4; top >= 0;) {                      // half-cleaners 2^top .. 2^b in one round
            const int b = top >= 3 ? top - 3 : 0;
            unsigned d[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) d[j] = (unsigned)sizeof(K) * (unsigned)spad(j << b);
            for (int t = tid; t < ngroup; t += BT) {
                const unsigned a0 = (unsigned)sizeof(K) * (unsigned)spad(((t >> b) << (b + 4)) | (t & ((1 << b) - 1)));
                char *base = reinterpret_cast<char *>(sk);
                K v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<K *>(base + a0 + d[j]);
                sort_group_strides<K>(v, top - b + 1);
#pragma unroll
                for (int j = 0; j < 16; ++j) *reinterpret_cast<K *>(base + a0 + d[j]) = v[j];
            }
            __syncthreads();
            top = b - 1;
        }
    }
    RAY_FR_MARK(11);
    // rank of each entry's node: heads counted per contiguous chunk of n2 / BT keys, scanned over the BT threads
    // (inclusive scan inside the wave by shuffles, the wave totals through LDS)
    const int cs = n2 / BT, e0 = tid * cs;
    const K before = e0 > 0 ? sk[spad(e0 - 1)] : SK::dead();       // dead sorts last: never equal to a live key's node
    int heads = 0;
    {
        uint64_t pn = e0 > 0 && before != SK::dead() ? SK::node(before, eshift) : ~0ull;
        for (int e = e0; e < e0 + cs; ++e) {
            const K k = sk[spad(e)];
            const uint64_t nd = SK::node(k, eshift);
            heads += k != SK::dead() && nd != pn;
            pn = nd;
        }
    }
    int incl = heads;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if ((tid & 63) >= off) incl += v;
    }
    if ((tid & 63) == 63) s_cnt[tid >> 6] = incl;
    __syncthreads();
    int wave_before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < BT / 64; ++w) {
        const int c = s_cnt[w];
        wave_before += w < (tid >> 6) ? c : 0;
        total += c;
    }
    int rank = wave_before + incl - heads - 1;      // rank of the last head before this chunk
    if (tid == 0) ucount[ray] = total;
    // The words replace the keys in LDS (every thread has read the key before its chunk by now: barrier above), then
    // go to memory with consecutive lanes on consecutive words: a thread storing its own chunk word by word touches
    // 64 different lines per instruction, and those partial-line stores were 40 % of this kernel's time.
    uint32_t *sw = reinterpret_cast<uint32_t *>(ray_smem);
    constexpr int kw = (int)(sizeof(K) / 4);          // a key's slot holds its word in the first four bytes
    {
        uint64_t pn = e0 > 0 && before != SK::dead() ? SK::node(before, eshift) : ~0ull;
        for (int e = e0; e < e0 + cs; ++e) {
            const K k = sk[spad(e)];
            uint32_t w = kSrtDead;
            if (k != SK::dead()) {
                const uint64_t nd = SK::node(k, eshift);
                const bool head = nd != pn;
                pn = nd;
                rank += head;
                w = (uint32_t)rank << 15 | (uint32_t)head << 14 | SK::emission(k, eshift);
            }
            sw[spad(e) * kw] = w;
        }
    }
    __syncthreads();
    for (int e = tid; e < ne; e += BT) rk[e] = sw[spad(e) * kw];
    RAY_FR_MARK(12);
}

// Theta column of emission e = 12 row + k from the ray's trajectory records (see kTrajDoubles)
// (`stride`: doubles between consecutive records: kTrajDoubles in the slab, kTrajLdsDoubles in the LDS copy)
// The LDS copy is NOT padded: 128-byte rows put every row's piece on one of two bank groups, but the merge is bound by
// its instruction count and by occupancy, not by LDS (SQ_LDS_BANK_CONFLICT is 1 % of the SIMD cycles), and the 16 bytes
// per row are what fits a fourth workgroup on a CU: 5.45 -> 4.97 ms (a 136-byte stride breaks the b128 alignment: 3x slower)
constexpr int kTrajLdsDoubles = kTrajDoubles;
__device__ __forceinline__ void theta_of_record(const double2 *__restrict__ rec, unsigned k, double th[4]);
__device__ __forceinline__ void theta_of_entry(const double *__restrict__ recs, int stride, unsigned e, int N, double th[4], int &node)
{
    const unsigned r = (e * 43691u) >> 19, k = e - r * (unsigned)kFrSlotsPerCall;
    const double2 *rec = reinterpret_cast<const double2 *>(recs + (size_t)r * stride);
    const int cv = (int)__double_as_longlong(rec[7].x);
    node = ((cv & 0xffff) + stencil_di((int)k)) + ((cv >> 16) + stencil_dj((int)k)) * N;
    theta_of_record(rec, k, th);
}
// Theta column of stencil node k from one trajectory record
__device__ __forceinline__ void theta_of_record(const double2 *__restrict__ rec, unsigned k, double th[4])
{
    const double2 uv = rec[6];
    const double u = uv.x, v = uv.y, a = 1 - u, b = 1 - v;
    const unsigned kk = k & 3u;
    // t = {ab, ub, av, uv}: phi_kk = t[kk], psi_kk = +-t[kk ^ 1] (minus for odd kk), gam_kk = +-t[kk ^ 2] (minus for kk >= 2)
    const double fx0 = (kk & 1u) ? u : a, fx1 = (kk & 1u) ? a : u;         // x factor of t[kk], of t[kk ^ 1]
    const double fy0 = (kk & 2u) ? v : b, fy1 = (kk & 2u) ? b : v;         // y factor of t[kk], of t[kk ^ 2]
    const double phi = fx0 * fy0;
    const double psi = (kk & 1u) ? -(fx1 * fy0) : fx1 * fy0;
    const double gam = (kk & 2u) ? -(fx0 * fy1) : fx0 * fy1;
    if (k < 4u) {
        const double2 p0 = rec[0], p1 = rec[1], q0 = rec[2], q1 = rec[3], s0 = rec[4], s1 = rec[5];
        // (phi P + psi Q) + gam S, multiply-adds contracted (M is defined up to rounding: the reference forms it through
        // LAPACK / BLAS products)
        th[0] = __builtin_fma(gam, s0.x, __builtin_fma(psi, q0.x, phi * p0.x));
        th[1] = __builtin_fma(gam, s0.y, __builtin_fma(psi, q0.y, phi * p0.y));
        th[2] = __builtin_fma(gam, s1.x, __builtin_fma(psi, q1.x, phi * p1.x));
        th[3] = __builtin_fma(gam, s1.y, __builtin_fma(psi, q1.y, phi * p1.y));
    } else {
        const bool xs = k < 8u;
        const double2 c0 = xs ? rec[2] : rec[4], c1 = xs ? rec[3] : rec[5];
        const double f = xs ? -psi : -gam;
        th[0] = f * c0.x; th[1] = f * c0.y; th[2] = f * c1.x; th[3] = f * c1.y;
    }
}

// Tiles of 256 sorted entries: every thread rebuilds ONE entry's Theta column (no divergence: heads and
// followers cost the same), the tile's columns go to LDS, then each head adds up its run from there (a run
// that leaves the tile is finished from the records directly), applies rho and stores its node.
// LDS = true: the ray's records (144-byte rows) and its sorted words were copied to dynamic shared memory;
// the loop then holds no global load, so nothing in it waits for the stores of the tile before (loads and
// stores share one in-order counter on gfx9: a load behind a store waits for the store's acknowledgement).
template <bool LDS>
__device__ __forceinline__ void emit_tiles(const uint32_t *__restrict__ rk_g, const double *__restrict__ recs_g, int nrow, int ne, int N,
                                           const double R[16], double droptol, int64_t off, int32_t *__restrict__ ucols,
                                           double *__restrict__ uvals, int cnt[4], double2 *s_th, uint32_t *s_w)
{
    const int tid = threadIdx.x;
    const double *recs = LDS ? reinterpret_cast<const double *>(ray_smem) : recs_g;
    const uint32_t *rk = LDS ? reinterpret_cast<const uint32_t *>(ray_smem + (size_t)nrow * kTrajLdsDoubles * 8) : rk_g;
    const int stride = LDS ? kTrajLdsDoubles : kTrajDoubles;
    for (int t0 = 0; t0 < ne; t0 += 256) {
        const int e = t0 + tid;
        const uint32_t w = e < ne ? rk[e] : kSrtDead;
        double s[4] = {0, 0, 0, 0};
        int node = 0;
        if (w != kSrtDead) theta_of_entry(recs, stride, w & 0x3FFFu, N, s, node);
        if (!LDS) s_w[tid] = w;
        s_th[tid * 2] = make_double2(s[0], s[1]);
        s_th[tid * 2 + 1] = make_double2(s[2], s[3]);
        __syncthreads();
        unsigned kept = 0;                                                                   // bit i: row i keeps this head's entry
        if (w != kSrtDead && (w & 0x4000u)) {                                                // a head
            for (int e2 = e + 1; e2 < ne; ++e2) {
                const int q = e2 - t0, qc = q < 256 ? q : 255;
                const uint32_t w2 = (LDS || q >= 256) ? rk[e2] : s_w[q];
                const double2 lo = s_th[qc * 2], hi = s_th[qc * 2 + 1];                      // read with the word: one LDS latency per follower
                if (w2 == kSrtDead || (w2 & 0x4000u)) break;
                double th[4];
                if (q < 256) {
                    th[0] = lo.x; th[1] = lo.y; th[2] = hi.x; th[3] = hi.y;
                } else {
                    int n2_;
                    theta_of_entry(recs, stride, w2 & 0x3FFFu, N, th, n2_);
                }
                s[0] += th[0]; s[1] += th[1]; s[2] += th[2]; s[3] += th[3];                  // Phi += Theta (:159,175)
            }
            const int64_t u = off + (int64_t)(w >> 15);
            double o4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {                                                   // rho * Phi (:184)
                o4[i] = __builtin_fma(R[i * 4 + 3], s[3], __builtin_fma(R[i * 4 + 2], s[2], __builtin_fma(R[i * 4 + 1], s[1], R[i * 4] * s[0])));
                kept |= (unsigned)((droptol < 0) || (__builtin_fabs(o4[i]) > droptol)) << i;     // droptol! (:242)
            }
            ucols[u] = (int32_t)node;
            double2 *dst = reinterpret_cast<double2 *>(uvals + u * 4);
            dst[0] = make_double2(o4[0], o4[1]);
            dst[1] = make_double2(o4[2], o4[3]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) cnt[i] += __popcll(__ballot((kept >> i) & 1u));          // the WAVE's counts, in every lane
        __syncthreads();
    }
}

// sums every distinct node's contributions in emission order, applies rho, writes the ray's block row
// (ucols, uvals[4]) at uoff[ray] and the per-matrix-row entry counts that survive droptol.
// Rays of at most `lds_rows` records go through the LDS form of emit_tiles (dynamic shared memory:
// lds_rows * kMergeLdsRowBytes), longer ones read records and sorted words in place.
constexpr int kMergeLdsRowBytes = kTrajLdsDoubles * 8 + kFrSlotsPerCall * 4;
constexpr int kMergeLdsMaxRows = 224;           // 42 KiB; 7 record pieces and 11 sorted words per thread in flight
__global__ __launch_bounds__(256) void k_merge_emit(const uint32_t *__restrict__ srt, const double *__restrict__ traj,
                                                    const int64_t *__restrict__ wave_base, const int32_t *__restrict__ wave_cap,
                                                    const int32_t *__restrict__ perm, const int32_t *__restrict__ nsteps,
                                                    const double *__restrict__ rho, const int64_t *__restrict__ uoff,
                                                    double droptol, int N, int lds_rows, int64_t num, int32_t *__restrict__ ucols,
                                                    double *__restrict__ uvals, int32_t *__restrict__ rowcnt)
{
    __shared__ int s_rows[4];
    __shared__ double2 s_th[256 * 2];            // the tile's Theta columns
    __shared__ uint32_t s_w[256];                // the tile's sorted words (form without the LDS copy)
    const int tid = threadIdx.x;
    for (int64_t slot = blockIdx.x; slot < num; slot += gridDim.x) {
    const int64_t ray = perm[slot];
#ifdef RAY_FR_PHASES
    unsigned long long t_mark = __builtin_readcyclecounter();
    if (tid == 0) atomicAdd(&g_fr_phase[0], 1ull);
#endif
    if (tid < 4) s_rows[tid] = 0;
    const int nrow = frechet_rows(nsteps[ray]);
    const int ne = nrow * kFrSlotsPerCall;
    const int64_t row0 = ray_row0(wave_base[slot >> 6], wave_cap[slot >> 6], (int)(slot & 63));
    const uint32_t *rk = srt + row0 * kFrSlotsPerCall;
    const double *recs = traj + row0 * kTrajDoubles;
    const int64_t off = uoff[ray];
    double R[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) R[k] = rho[ray * 16 + k];
    int cnt[4] = {0, 0, 0, 0};
#ifdef RAY_FR_PHASES
    if (R[0] == 1.2345e300 || off < 0 || ne < 0) return;      // the loads above are complete when the mark is taken
#endif
    RAY_FR_MARK(1);
    if (nrow <= lds_rows) {
        // every load of the copy is issued before the first one is waited for (fixed trip counts: a loop with a
        // bound known only at run time comes out as load - wait - store per iteration, one memory latency each)
        double2 *l2 = reinterpret_cast<double2 *>(ray_smem);
        const double2 *g2 = reinterpret_cast<const double2 *>(recs);
        uint32_t *lw = reinterpret_cast<uint32_t *>(ray_smem + (size_t)nrow * kTrajLdsDoubles * 8);
        constexpr int kRecLoads = kMergeLdsMaxRows * (kTrajDoubles / 2) / 256, kWordLoads = (kMergeLdsMaxRows * kFrSlotsPerCall + 255) / 256;
        double2 rv[kRecLoads];
        uint32_t wv[kWordLoads];
        const int nrec2 = nrow * (kTrajDoubles / 2);
#pragma unroll
        for (int k = 0; k < kRecLoads; ++k) rv[k] = tid + 256 * k < nrec2 ? g2[tid + 256 * k] : make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < kWordLoads; ++k) wv[k] = tid + 256 * k < ne ? rk[tid + 256 * k] : kSrtDead;
#pragma unroll
        for (int k = 0; k < kRecLoads; ++k) {
            const int q = tid + 256 * k;
            if (q < nrec2) l2[(q >> 3) * (kTrajLdsDoubles / 2) + (q & 7)] = rv[k];
        }
#pragma unroll
        for (int k = 0; k < kWordLoads; ++k)
            if (tid + 256 * k < ne) lw[tid + 256 * k] = wv[k];
        __syncthreads();
        RAY_FR_MARK(2);
        emit_tiles<true>(rk, recs, nrow, ne, N, R, droptol, off, ucols, uvals, cnt, s_th, s_w);
    } else {
        __syncthreads();
        emit_tiles<false>(rk, recs, nrow, ne, N, R, droptol, off, ucols, uvals, cnt, s_th, s_w);
    }
    RAY_FR_MARK(3);
    // cnt holds the WAVE's counts (ballots): one LDS atomic per wave and row -- 256 lanes adding to one LDS
    // word are served one lane at a time and kept the LDS pipeline of the whole CU busy for 17k cycles per ray
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (cnt[i]) atomicAdd(&s_rows[i], cnt[i]);
    }
    __syncthreads();
    if (tid < 4) rowcnt[ray * 4 + tid] = s_rows[tid];
    __syncthreads();
    RAY_FR_MARK(4);
    }
}

// ---- merge by stencil SEGMENTS ---------------------------------------------------------------------------
// In node order (i + j N) the twelve stencil nodes of a cell (i, j) are four runs of consecutive nodes: row j - 1:
// i, i + 1; rows j and j + 1: i - 1 .. i + 2; row j + 2: i, i + 1.  Sorting the 4 segments of every record by their
// first node -- a third of the keys, a quarter of the network at the usual sizes -- orders everything the merge needs:
// a segment OWNS the nodes it covers that no earlier segment (in sorted order) reaches, owned nodes come out in
// increasing order, and the other contributions to an owned node sit in the segments right behind its owner.
//   k_seg_sort   keys (first node, record * 4 + segment, length - 1); the network of k_merge_sort; the number of
//                distinct nodes of the ray = what the segments own; hands over the sorted segment ids (2 bytes each)
//   k_seg_emit   records and sorted ids in LDS; first node and length of every segment from its record's cell; the
//                running maximum of the ends and the owned counts scanned over the workgroup; then every thread
//                walks its stretch of segments: each owned node sums its contributions (sorted order: first node,
//                then emission), applies rho, writes column and values.
// Segments are clipped to the grid (a cell on the grid's rim); a segment wholly outside is dead and sorts last.
constexpr uint16_t kSegDead = 0xffffu;
__host__ __device__ inline int frechet_segments(int steps) { return frechet_rows(steps) * 4; }

__device__ __forceinline__ void stencil_segment(int cv, int sg, int N, int &start, int &len, int &di0)
{
    const int i = cv & 0xffff, j = cv >> 16;
    const int row = j - 1 + sg;
    const bool wide = sg == 1 || sg == 2;
    const int lo = wide ? i - 1 : i, hi = wide ? i + 2 : i + 1;
    di0 = lo - i;
    const int clo = lo > 0 ? lo : 0, chi = hi < N - 1 ? hi : N - 1;
    len = ((unsigned)row < (unsigned)N && chi >= clo) ? chi - clo + 1 : 0;
    start = clo + row * N;
}
// stencil index k of the node at x-offset di (-1 .. 2) in segment sg: a nibble table (see stencil_di / stencil_dj)
__device__ __forceinline__ unsigned stencil_k_of(int sg, int di)
{
    return (unsigned)(0x0980632741050ba0ull >> (4 * (sg * 4 + di + 1))) & 15u;
}

template <typename K> struct SegKey;
template <> struct SegKey<uint64_t> {
    static __device__ __forceinline__ uint64_t dead() { return kKeyDead; }
    static __device__ __forceinline__ uint64_t pack(int start, int q, int len, int) { return (uint64_t)start << 40 | (uint64_t)q << 20 | (uint64_t)(len - 1); }
    static __device__ __forceinline__ int start(uint64_t k, int) { return (int)(k >> 40); }
    static __device__ __forceinline__ int q(uint64_t k, int) { return (int)(k >> 20) & 0xFFFFF; }
    static __device__ __forceinline__ int len(uint64_t k) { return (int)(k & 3u) + 1; }
};
template <> struct SegKey<uint32_t> {                 // first node << (es + 2) | id << 2 | length - 1
    static __device__ __forceinline__ uint32_t dead() { return 0xffffffffu; }
    static __device__ __forceinline__ uint32_t pack(int start, int q, int len, int es) { return (uint32_t)start << (es + 2) | (uint32_t)q << 2 | (uint32_t)(len - 1); }
    static __device__ __forceinline__ int start(uint32_t k, int es) { return (int)(k >> (es + 2)); }
    static __device__ __forceinline__ int q(uint32_t k, int es) { return (int)(k >> 2) & ((1 << es) - 1); }
    static __device__ __forceinline__ int len(uint32_t k) { return (int)(k & 3u) + 1; }
};

// the bitonic network of k_merge_sort on n2 keys in LDS whose aligned groups of sixteen are sorted already
template <int BT, typename K>
__device__ __forceinline__ void bitonic_from_sixteen(K *sk, int n2, int tid)
{
    auto spad = [](int i) { return i + (i >> 4); };
    const int ngroup = n2 >> 4;
    for (int top0 = 4; (1 << top0) < n2; ++top0) {
        {
            const int b = top0 - 3, lm = (1 << b) - 1;
            unsigned d[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) d[j] = (unsigned)sizeof(K) * (unsigned)spad(j << b);
            for (int t = tid; t < ngroup; t += BT) {
                const int hi = (t >> b) << (b + 4), lo = t & lm;
                const unsigned a_lo = (unsigned)sizeof(K) * (unsigned)spad(hi | lo), a_hi = (unsigned)sizeof(K) * (unsigned)spad(hi | (lo ^ lm));
                char *base = reinterpret_cast<char *>(sk);
                K v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<K *>(base + (j < 8 ? a_lo : a_hi) + d[j]);
#pragma unroll
                for (int p = 0; p < 8; ++p) cmp_exchange<K>(v[p], v[15 - p]);
                sort_group_strides<K>(v, 3);
#pragma unroll
                for (int j = 0; j < 16; ++j) *reinterpret_cast<K *>(base + (j < 8 ? a_lo : a_hi) + d[j]) = v[j];
            }
            __syncthreads();
        }
        for (int top = top0 - 4; top >= 0;) {
            const int b = top >= 3 ? top - 3 : 0;
            unsigned d[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) d[j] = (unsigned)sizeof(K) * (unsigned)spad(j << b);
            for (int t = tid; t < ngroup; t += BT) {
                const unsigned a0 = (unsigned)sizeof(K) * (unsigned)spad(((t >> b) << (b + 4)) | (t & ((1 << b) - 1)));
                char *base = reinterpret_cast<char *>(sk);
                K v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<K *>(base + a0 + d[j]);
                sort_group_strides<K>(v, top - b + 1);
#pragma unroll
                for (int j = 0; j < 16; ++j) *reinterpret_cast<K *>(base + a0 + d[j]) = v[j];
            }
            __syncthreads();
            top = b - 1;
        }
    }
}

// exclusive running sum of one int over the threads of a workgroup (BT / 64 waves)
template <int BT>
__device__ __forceinline__ int block_excl_sum(int v, int *s_w, int tid, int &total)
{
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if ((tid & 63) >= off) incl += o;
    }
    if ((tid & 63) == 63) s_w[tid >> 6] = incl;
    __syncthreads();
    int before = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < BT / 64; ++w) { const int c = s_w[w]; before += w < (tid >> 6) ? c : 0; total += c; }
    __syncthreads();
    return before + incl - v;
}

template <int BT, typename K>
__global__ __launch_bounds__(BT) void k_seg_sort(const int32_t *__restrict__ cells, uint32_t *__restrict__ srt,
                                                  const int64_t *__restrict__ wave_base,
                                                  const int32_t *__restrict__ wave_cap, const int32_t *__restrict__ perm,
                                                  const int32_t *__restrict__ nsteps, int32_t *__restrict__ ucount, int eshift, int N)
{
    typedef SegKey<K> SK;
    auto spad = [](int i) { return i + (i >> 4); };
    K *sk = reinterpret_cast<K *>(ray_smem);
    __shared__ int s_w[BT / 64];
    __shared__ int s_cell[kMergeMaxEntries / kFrSlotsPerCall + 4];
    const int tid = threadIdx.x;
    const int64_t slot = blockIdx.x;
    const int64_t ray = perm[slot];
    const int nrow = frechet_rows(nsteps[ray]);
    const int nseg = nrow * 4;
    if (nseg == 0) {
        if (tid == 0) ucount[ray] = 0;
        return;
    }
    int n2 = 256;
    while (n2 < nseg) n2 <<= 1;
    const int64_t wb = wave_base[slot >> 6];
    const int lane = (int)(slot & 63);
    uint16_t *rk = reinterpret_cast<uint16_t *>(srt + ray_row0(wb, wave_cap[slot >> 6], lane) * kFrSlotsPerCall);
    for (int r = tid; r < nrow + 3; r += BT) s_cell[r] = r < nrow ? cells[(wb + r) * 64 + lane] : 0;
    __syncthreads();
    const int ngroup = n2 >> 4;
    for (int t = tid; t < ngroup; t += BT) {                       // sixteen keys = the segments of four records
        K v[16];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * t + rr;
            const int cv = r < nrow ? s_cell[r] : 0;
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                int start, len, di0;
                stencil_segment(cv, sg, N, start, len, di0);
                v[4 * rr + sg] = (r < nrow && len > 0) ? SK::pack(start, 4 * r + sg, len, eshift) : SK::dead();
            }
        }
        sort_sixteen<K>(v);
#pragma unroll
        for (int j = 0; j < 16; ++j) sk[t * 17 + j] = v[j];
    }
    __syncthreads();
    bitonic_from_sixteen<BT, K>(sk, n2, tid);
    // what the segments own: a node is new when no earlier segment reaches it.  The running maximum of the ends needs no
    // scan over the workgroup: a segment is at most four nodes long, so only those that start within three nodes of this
    // stretch's first can reach into it -- a short look-back.
    const int cs = n2 / BT, e0 = tid * cs;
    int pme = 0;
    {
        const K k0 = sk[spad(e0)];
        if (k0 != SK::dead()) {
            const int lim = SK::start(k0, eshift) - 3;
            for (int e = e0 - 1; e >= 0; --e) {
                const K k = sk[spad(e)];
                const int st = SK::start(k, eshift);
                if (st < lim) break;
                const int en = st + SK::len(k);
                pme = pme > en ? pme : en;
            }
        }
    }
    int own = 0;
    for (int e = e0; e < e0 + cs; ++e) {
        const K k = sk[spad(e)];
        if (k != SK::dead()) {
            const int st = SK::start(k, eshift), en = st + SK::len(k);
            const int a = st > pme ? st : pme;
            own += en > a ? en - a : 0;
            pme = pme > en ? pme : en;
        }
    }
    int total;
    (void)block_excl_sum<BT>(own, s_w, tid, total);
    if (tid == 0) ucount[ray] = total;
    // the sorted ids leave through LDS (a stretch of its own behind the launch's key capacity 2^eshift), consecutive
    // lanes on consecutive halfwords
    uint16_t *sw = reinterpret_cast<uint16_t *>(ray_smem + merge_sort_lds_bytes(1 << eshift, sizeof(K)));
    for (int e = e0; e < e0 + cs; ++e) {
        const K k = sk[spad(e)];
        sw[e] = k != SK::dead() ? (uint16_t)SK::q(k, eshift) : kSegDead;
    }
    __syncthreads();
    for (int e = tid; e < nseg; e += BT) rk[e] = sw[e];
}


#ifndef RAY_SEG_WINDOW
#define RAY_SEG_WINDOW 4
#endif
constexpr int kSegWindow = RAY_SEG_WINDOW;                                     // segments behind an owner looked at without a loop
constexpr int kSegLdsRowBytes = kTrajLdsDoubles * 8;             // the record copy; + 48 B per row of the longest ray for the segments
__host__ __device__ inline size_t seg_emit_lds_bytes(int lds_rows, int max_rows)
{
    return (size_t)lds_rows * kSegLdsRowBytes + (size_t)max_rows * (4 * 4 + 4 * 2 + kFrSlotsPerCall * 2) + 16;
}
// Theta column of node n as segment (id q: record q >> 2, stencil segment q & 3) sees it
__device__ __forceinline__ void theta_of_segment_node(const double *__restrict__ recs, int stride, unsigned q, int n, int N, double th[4])
{
    const int sg = (int)(q & 3u);
    const double2 *rec = reinterpret_cast<const double2 *>(recs + (size_t)(q >> 2) * stride);
    const int cv = (int)__double_as_longlong(rec[7].x);
    const int di = n - (int)mad24((unsigned)((cv >> 16) - 1 + sg), (unsigned)N, (unsigned)(cv & 0xffff));
    theta_of_record(rec, stencil_k_of(sg, di), th);
}
__global__ __launch_bounds__(256) void k_seg_emit(const uint32_t *__restrict__ srt, const double *__restrict__ traj,
                                                  const int64_t *__restrict__ wave_base, const int32_t *__restrict__ wave_cap,
                                                  const int32_t *__restrict__ perm, const int32_t *__restrict__ nsteps,
                                                  const double *__restrict__ rho, const int64_t *__restrict__ uoff,
                                                  double droptol, int N, int lds_rows, int max_rows, int64_t num, int32_t *__restrict__ ucols,
                                                  double *__restrict__ uvals, int32_t *__restrict__ rowcnt)
{
    __shared__ int s_rows[4];
    __shared__ int s_w[4];
    const int tid = threadIdx.x;
    // per sorted segment: first node << 3 | length (length 0: dead) and its id; per distinct node, in node order: its owner
    // (sorted index of the first segment that reaches it) << 2 | its offset in that segment
    uint32_t *s_seg = reinterpret_cast<uint32_t *>(ray_smem + (size_t)lds_rows * kSegLdsRowBytes);
    uint16_t *s_q = reinterpret_cast<uint16_t *>(s_seg + (size_t)max_rows * 4);
    uint16_t *s_own = s_q + (size_t)max_rows * 4;
    constexpr uint32_t kDeadSeg = 0xfffffff8u;
    for (int64_t slot = blockIdx.x; slot < num; slot += gridDim.x) {
    const int64_t ray = perm[slot];
    if (tid < 4) s_rows[tid] = 0;
    const int nrow = frechet_rows(nsteps[ray]);
    const int nseg = nrow * 4;
    const int64_t row0 = ray_row0(wave_base[slot >> 6], wave_cap[slot >> 6], (int)(slot & 63));
    const uint16_t *rk = reinterpret_cast<const uint16_t *>(srt + row0 * kFrSlotsPerCall);
    const double *recs_g = traj + row0 * kTrajDoubles;
    const int64_t off = uoff[ray];
    double R[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) R[k] = rho[ray * 16 + k];
    const bool in_lds = nrow <= lds_rows;
    if (in_lds) {
        // every load of the copy is issued before the first one is waited for (fixed trip counts)
        double2 *l2 = reinterpret_cast<double2 *>(ray_smem);
        const double2 *g2 = reinterpret_cast<const double2 *>(recs_g);
        constexpr int kRecLoads = kMergeLdsMaxRows * (kTrajDoubles / 2) / 256, kIdLoads = (kMergeLdsMaxRows * 2 + 255) / 256;
        double2 rv[kRecLoads];
        uint32_t wv[kIdLoads];
        const int nrec2 = nrow * (kTrajDoubles / 2), nid2 = nrow * 2;          // ids two by two (a row's four are 8 bytes)
        const uint32_t *rk2 = reinterpret_cast<const uint32_t *>(rk);
#pragma unroll
        for (int k = 0; k < kRecLoads; ++k) rv[k] = tid + 256 * k < nrec2 ? g2[tid + 256 * k] : make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < kIdLoads; ++k) wv[k] = tid + 256 * k < nid2 ? rk2[tid + 256 * k] : 0xffffffffu;
#pragma unroll
        for (int k = 0; k < kRecLoads; ++k) {
            const int q = tid + 256 * k;
            if (q < nrec2) l2[(q >> 3) * (kTrajLdsDoubles / 2) + (q & 7)] = rv[k];
        }
#pragma unroll
        for (int k = 0; k < kIdLoads; ++k)
            if (tid + 256 * k < nid2) reinterpret_cast<uint32_t *>(s_q)[tid + 256 * k] = wv[k];
    } else {
        for (int e = tid; e < nseg; e += 256) s_q[e] = rk[e];
    }
    __syncthreads();
    const double *recs = in_lds ? reinterpret_cast<const double *>(ray_smem) : recs_g;
    const int stride = in_lds ? kTrajLdsDoubles : kTrajDoubles;
    // first node and length of every sorted segment, from its record's cell
    for (int e = tid; e < nseg; e += 256) {
        const unsigned q = s_q[e];
        uint32_t w = kDeadSeg;
        if (q != kSegDead) {
            const int cv = (int)__double_as_longlong(recs[(size_t)(q >> 2) * stride + 14]);
            int start, len, di0;
            stencil_segment(cv, (int)(q & 3u), N, start, len, di0);
            if (len > 0) w = (uint32_t)start << 3 | (uint32_t)len;
        }
        s_seg[e] = w;
    }
    __syncthreads();
    // what every segment owns: the nodes no earlier segment reaches (running maximum of the ends: a look-back over the
    // segments that start within three nodes, as in k_seg_sort), ranked by a running sum
    const int cs = (nseg + 255) / 256, e0 = tid * cs < nseg ? tid * cs : nseg, e1 = e0 + cs < nseg ? e0 + cs : nseg;
    int pme = 0;
    if (e0 < e1 && (s_seg[e0] & 7u)) {
        const int lim = (int)(s_seg[e0] >> 3) - 3;
        for (int e = e0 - 1; e >= 0; --e) {
            const uint32_t w = s_seg[e];
            const int st = (int)(w >> 3);
            if (st < lim) break;
            const int en = st + (int)(w & 7u);
            pme = pme > en ? pme : en;
        }
    }
    int own = 0;
    {
        int pm = pme;
        for (int e = e0; e < e1; ++e) {
            const uint32_t w = s_seg[e];
            if (w & 7u) {
                const int st = (int)(w >> 3), en = st + (int)(w & 7u);
                const int a = st > pm ? st : pm;
                own += en > a ? en - a : 0;
                pm = pm > en ? pm : en;
            }
        }
    }
    int total;
    int rank = block_excl_sum<256>(own, s_w, tid, total);
    {
        int pm = pme;
        for (int e = e0; e < e1; ++e) {
            const uint32_t w = s_seg[e];
            if (w & 7u) {
                const int st = (int)(w >> 3), en = st + (int)(w & 7u);
                for (int n = st > pm ? st : pm; n < en; ++n) s_own[rank++] = (uint16_t)(e << 2 | (n - st));
                pm = pm > en ? pm : en;
            }
        }
    }
    __syncthreads();
    // one distinct node per thread, in node order: its owner's Theta column and those of the segments right behind it
    int cnt[4] = {0, 0, 0, 0};
    for (int x = tid; x < total; x += 256) {
        const unsigned ow = s_own[x];
        const int js = (int)(ow >> 2);
        const uint32_t w = s_seg[js];
        const int n = (int)(w >> 3) + (int)(ow & 3u);
        uint32_t wn[kSegWindow];
#pragma unroll
        for (int t = 0; t < kSegWindow; ++t) wn[t] = js + 1 + t < nseg ? s_seg[js + 1 + t] : kDeadSeg;
        double s4[4];
        theta_of_segment_node(recs, stride, s_q[js], n, N, s4);
        bool more = true;
        unsigned cover = 0;                                           // which of the window's segments reach n
#pragma unroll
        for (int t = 0; t < kSegWindow; ++t) {
            const int sj = (int)(wn[t] >> 3);
            more = more && sj <= n;
            cover |= (unsigned)(more && sj + (int)(wn[t] & 7u) > n) << t;
        }
        while (cover) {                                               // in sorted order; as many passes as the busiest lane needs
            const int t = __builtin_ctz(cover);
            cover &= cover - 1;
            double th[4];
            theta_of_segment_node(recs, stride, s_q[js + 1 + t], n, N, th);
            s4[0] += th[0]; s4[1] += th[1]; s4[2] += th[2]; s4[3] += th[3];              // Phi += Theta (:159,175)
        }
        for (int j = js + 1 + kSegWindow; more && j < nseg; ++j) {        // rare: more than kSegWindow segments start within reach
            const uint32_t wj = s_seg[j];
            const int sj = (int)(wj >> 3);
            if (sj > n) break;
            if (sj + (int)(wj & 7u) <= n) continue;
            double th[4];
            theta_of_segment_node(recs, stride, s_q[j], n, N, th);
            s4[0] += th[0]; s4[1] += th[1]; s4[2] += th[2]; s4[3] += th[3];
        }
        double o4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                                   // rho * Phi (:184)
            o4[i] = __builtin_fma(R[i * 4 + 3], s4[3], __builtin_fma(R[i * 4 + 2], s4[2], __builtin_fma(R[i * 4 + 1], s4[1], R[i * 4] * s4[0])));
            cnt[i] += (droptol < 0) || (__builtin_fabs(o4[i]) > droptol);                    // droptol! (:242)
        }
        const int64_t u = off + x;
        ucols[u] = (int32_t)n;
        double2 *dst = reinterpret_cast<double2 *>(uvals + u * 4);
        dst[0] = make_double2(o4[0], o4[1]);
        dst[1] = make_double2(o4[2], o4[3]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = cnt[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if ((tid & 63) == 0 && c) atomicAdd(&s_rows[i], c);
    }
    __syncthreads();
    if (tid < 4) rowcnt[ray * 4 + tid] = s_rows[tid];
    __syncthreads();
    }
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

// The passes below read the BLOCK ROWS: per ray `ray` the distinct nodes uoff[ray] .. uoff[ray+1], each with
// its column and the values of matrix rows 4 ray .. 4 ray + 3.  An entry belongs to the matrix when it
// survives droptol (NonObstacle.jl:242), decided on the fly: kept(v).  One wave per ray.
struct BlockRows {
    const int64_t *uoff;
    const int32_t *ucols;
    const double *uvals;
    int64_t num;
    double droptol;
    __device__ __forceinline__ bool kept(double v) const { return droptol < 0 || __builtin_fabs(v) > droptol; }
    __device__ __forceinline__ void load(int64_t u, double v[4]) const
    {
        const double2 *p2 = reinterpret_cast<const double2 *>(uvals + u * 4);
        const double2 lo = p2[0], hi = p2[1];
        v[0] = kept(lo.x) ? lo.x : 0.0; v[1] = kept(lo.y) ? lo.y : 0.0;
        v[2] = kept(hi.x) ? hi.x : 0.0; v[3] = kept(hi.y) ? hi.y : 0.0;
    }
};

// dofs[j] = nnz(M[4j-3,:]) - sum(fidelity[find(M[4j-3,:])])        NonObstacle.jl:340-342
__global__ __launch_bounds__(256) void k_br_dofs(const BlockRows m, const double *__restrict__ fidelity, double *__restrict__ dofs)
{
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= m.num) return;
    const int lane = threadIdx.x & 63;
    double s = 0;
    for (int64_t u = m.uoff[ray] + lane; u < m.uoff[ray + 1]; u += 64)
        if (m.kept(m.uvals[u * 4])) s += 1.0 - fidelity[m.ucols[u]];
    s = wave_sum_f64(s);
    if (lane == 0) dofs[ray] = s;
}

// the four dot products M[4 ray + r, :] x of one ray (all lanes hold them)
__device__ __forceinline__ void br_dots(const BlockRows &m, int64_t b, int64_t e, int lane, const double *__restrict__ x, double s[4])
{
    s[0] = s[1] = s[2] = s[3] = 0.0;
    for (int64_t u = b + lane; u < e; u += 64) {
        double v[4];
        m.load(u, v);
        const double xv = x[m.ucols[u]];
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = __builtin_fma(v[r], xv, s[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = wave_sum_f64(s[r]);
}

// y = M x
__global__ __launch_bounds__(256) void k_br_matvec(const BlockRows m, const double *__restrict__ x, double *__restrict__ y)
{
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= m.num) return;
    const int lane = threadIdx.x & 63;
    double s[4];
    br_dots(m, m.uoff[ray], m.uoff[ray + 1], lane, x, s);
    if (lane == 0) {
        double2 *dst = reinterpret_cast<double2 *>(y + ray * 4);
        dst[0] = make_double2(s[0], s[1]);
        dst[1] = make_double2(s[2], s[3]);
    }
}

// ---- the passes that end in a sum over rays: b = M' v, w = M' (M x), d = diag(M' M) --------------------------------
// The node range is cut into BANDS of kBandNodes consecutive columns (64 KB of fp64: one LDS tile); a ray's columns are
// sorted, so its entries of one band are one contiguous piece of its block row -- found once per matrix by binary
// search (k_br_band_offsets: `boff`, band-major) -- and a workgroup (band, chunk of rays) adds the pieces of its rays
// into the tile with LDS atomics, then adds the tile's non-zero words to the result: 256 consecutive doubles per
// instruction, and only (ray chunks) x N^2 of them instead of one global atomic per entry.  Sums are order-free (1e-12).
#ifndef RAY_BAND_ABLATE
#define RAY_BAND_ABLATE 0
#endif
constexpr int kBandNodes = 8192;
constexpr int kBandThreads = 1024;               // 16 waves share a tile: two workgroups = 32 waves per CU
constexpr int kBandRays = 4096;                  // rays per chunk; (ray in chunk, column in band) pack into 12 + 13 bits of the copy's column word

// boff[b * num + ray] = first entry of `ray` (relative to uoff[ray]) whose column is >= b * kBandNodes, b = 0 .. nb
__global__ __launch_bounds__(256) void k_br_band_offsets(const BlockRows m, int nb, int32_t *__restrict__ boff)
{
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= m.num) return;
    const int lane = threadIdx.x & 63;
    const int64_t u0 = m.uoff[ray], n = m.uoff[ray + 1] - u0;
    for (int b = lane; b <= nb; b += 64) {
        const int64_t key = (int64_t)b * kBandNodes;
        int64_t lo = 0, hi = n;                                   // first position with column >= key
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)m.ucols[u0 + mid] < key) lo = mid + 1; else hi = mid;
        }
        boff[(int64_t)b * m.num + ray] = (int32_t)lo;
    }
}

// blen[b * num + ray] = entries of `ray` in band b (the scan of these is where the band-major copy keeps them)
__global__ __launch_bounds__(256) void k_br_band_lengths(const int32_t *__restrict__ boff, int64_t n, int64_t num, int32_t *__restrict__ blen)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) blen[i] = boff[i + num] - boff[i];
}

// The BAND-MAJOR copy of the block rows: band after band, inside a band ray after ray, every (band, ray) piece at
// bstart[b * num + ray].  Read in that order the scatter passes stream through memory, where the ray-major rows give
// them one ~1 KB piece per ray and band (measured: 3.5 TB/s against 5.8 TB/s for the same bytes read in storage order).
__global__ __launch_bounds__(256) void k_br_band_copy(const BlockRows m, const int32_t *__restrict__ boff, const int64_t *__restrict__ bstart,
                                                      int nb, int32_t *__restrict__ bcols, double *__restrict__ bvals)
{
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= m.num) return;
    const int lane = threadIdx.x & 63;
    const int64_t u0 = m.uoff[ray];
    for (int b = 0; b < nb; ++b) {
        const int s = boff[(int64_t)b * m.num + ray], e = boff[(int64_t)(b + 1) * m.num + ray];
        if (e <= s) continue;
        const int64_t d0 = bstart[(int64_t)b * m.num + ray];
        for (int k = s + lane; k < e; k += 64) {
            const double2 *src = reinterpret_cast<const double2 *>(m.uvals + (u0 + k) * 4);
            double2 *dst = reinterpret_cast<double2 *>(bvals + (d0 + k - s) * 4);
            dst[0] = src[0];
            dst[1] = src[1];
            // the copy's column word: (ray within its chunk of kBandRays) << 13 | (column within the band)
            bcols[d0 + k - s] = (int32_t)(((unsigned)(ray % kBandRays) << 13) | (unsigned)(m.ucols[u0 + k] - (int64_t)b * kBandNodes));
        }
    }
}

// MODE 0: out += M[rays,:]' * coef[rays] (coef: 4 per ray -- v for the gradient, M x for the normal operator)
// MODE 1: out += sum of the squares of the rays' entries (diag(M' M))
template <int MODE>
__global__ __launch_bounds__(kBandThreads) void k_br_band_scatter(const BlockRows m, const int64_t *__restrict__ bstart,
                                                                  const double *__restrict__ coef, const uint8_t *__restrict__ ray_mask,
                                                                  int64_t nn, double *__restrict__ out)
{
    double *acc = reinterpret_cast<double *>(ray_smem);
    // bands from the middle of the grid outwards: the rays live in the unit disk, so the central bands hold the most
    // entries and the ones near the rim of the grid none -- longest first, the light ones fill the tail of the launch
    const int nbd = gridDim.y, mid = nbd >> 1, hh = ((int)blockIdx.y + 1) >> 1;
    const int b = ((int)blockIdx.y & 1) ? mid - hh : mid + hh;
    const int tid = threadIdx.x;
    const int64_t col0 = (int64_t)b * kBandNodes;
    for (int k = tid; k < kBandNodes; k += kBandThreads) acc[k] = 0.0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * kBandRays, r1 = r0 + kBandRays < m.num ? r0 + kBandRays : m.num;
    // `m` here is the BAND-MAJOR copy (uvals in band order, ucols = packed (ray in chunk, column in band); uoff unused).
    // The workgroup's entries are ONE contiguous stretch of it: a flat, fully coalesced stream, four loads in flight per
    // thread; an entry's ray comes with it, its coefficients from the L1/L2 (neighbouring entries share their ray).
    const int64_t *bs = bstart + (int64_t)b * m.num;
    const int64_t lo = bs[r0], hi = bs[r1];
#ifndef RAY_BAND_Q
#define RAY_BAND_Q 4
#endif
    constexpr int Q = RAY_BAND_Q;
    for (int64_t base = lo + tid; base < hi; base += (int64_t)Q * kBandThreads) {
        unsigned w[Q];
        double2 va[Q], vb[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t u = base + (int64_t)q * kBandThreads;
            const int64_t uc = u < hi ? u : lo;                   // clamped: every load is issued, the surplus is dropped below
            w[q] = (unsigned)m.ucols[uc];
            const double2 *p2 = reinterpret_cast<const double2 *>(m.uvals + uc * 4);
            va[q] = p2[0]; vb[q] = p2[1];
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int64_t u = base + (int64_t)q * kBandThreads;
            if (u >= hi) continue;
            const int64_t ray = r0 + (w[q] >> 13);
            const double v0 = m.kept(va[q].x) ? va[q].x : 0.0, v1 = m.kept(va[q].y) ? va[q].y : 0.0;
            const double v2 = m.kept(vb[q].x) ? vb[q].x : 0.0, v3 = m.kept(vb[q].y) ? vb[q].y : 0.0;
            if (ray_mask && !ray_mask[ray]) continue;
            double t;
            if (MODE == 0) {
                const double2 *cp = reinterpret_cast<const double2 *>(coef + ray * 4);
                const double2 c01 = cp[0], c23 = cp[1];
                t = ((v0 * c01.x + v1 * c01.y) + v2 * c23.x) + v3 * c23.y;
            } else t = ((v0 * v0 + v1 * v1) + v2 * v2) + v3 * v3;
#if RAY_BAND_ABLATE == 1
            acc[w[q] & (unsigned)(kBandNodes - 1)] = t;
#else
            atomicAdd(&acc[w[q] & (unsigned)(kBandNodes - 1)], t);      // LDS: ds_add_f64
#endif
        }
    }
    __syncthreads();
    for (int k = tid; k < kBandNodes; k += kBandThreads) {
        const double t = acc[k];
        if (t != 0.0 && col0 + k < nn) unsafeAtomicAdd(&out[col0 + k], t);
    }
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
// (a node takes the largest of its accepted rows' values in one atomic max)
__global__ __launch_bounds__(256) void k_br_fidelity(const BlockRows m, const double *__restrict__ x,
                                                     const double *__restrict__ mismatch, const uint8_t *__restrict__ ray_mask,
                                                     double rejection, double decay,
                                                     double *__restrict__ fidelity, double *__restrict__ residual)
{
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= m.num) return;
    const int lane = threadIdx.x & 63;
    if (ray_mask && !ray_mask[ray]) {
        if (residual && lane < 4) residual[ray * 4 + lane] = 0.0;
        return;
    }
    const int64_t b = m.uoff[ray], e = m.uoff[ray + 1];
    double s[4], f[4];
    br_dots(m, b, e, lane, x, s);
    bool any = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double res = __builtin_fabs(s[r] - mismatch[ray * 4 + r]);
        if (residual && lane == r) residual[ray * 4 + r] = res;
        const bool acc = res < rejection;
        f[r] = acc ? 1 - decay * res : -__builtin_inf();
        any |= acc;
    }
    if (!any) return;
    for (int64_t u = b + lane; u < e; u += 64) {
        const double2 *p2 = reinterpret_cast<const double2 *>(m.uvals + u * 4);
        const double2 lo = p2[0], hi = p2[1];
        double best = -__builtin_inf();
        if (m.kept(lo.x)) best = f[0];
        if (m.kept(lo.y) && f[1] > best) best = f[1];
        if (m.kept(hi.x) && f[2] > best) best = f[2];
        if (m.kept(hi.y) && f[3] > best) best = f[3];
        if (best > -__builtin_inf()) atomic_max_f64(&fidelity[m.ucols[u]], best);
    }
}

} // namespace ray
