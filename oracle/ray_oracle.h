/*
 * ray_oracle.h -- CPU restatement of the forward ray-shooting path of lowrank/ray.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (ray_amd/, include/) may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * PARITY STATUS: "parity unpinned" by the reference.  The reference is Julia-0.6-only
 * code with no tests, no golden vectors and no runnable interpreter in this image
 * (SURVEY.md section 8c).  This restatement follows the reference's operation order line by
 * line (citations at each function) and is pinned by (1) an independent pure-Python
 * restatement (oracle/ray_oracle.py) that must agree bit for bit, (2) closed-form
 * known-answer tests (tests/test_oracle_kat.py), (3) committed golden fixtures generated
 * by the Python restatement (tests/golden/).
 *
 * Array conventions (same as the reference):
 *   m, s     : double (num, 9) COLUMN-major with leading dimension ld  -> m[r + k*ld]
 *              columns [x0 y0 px0 py0 x1 y1 px1 py1 t]
 *   c        : double (N, N) column-major, first index = x           (NonObstacle.jl:102,306)
 *   eval     : double (N-1, N-1, 4) column-major                      (NonObstacle.jl:192)
 *   grad     : double (N-2, N-2, 8) column-major                      (NonObstacle.jl:193)
 */
#ifndef RAY_ORACLE_H
#define RAY_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status bits (shared vocabulary with include/ray_amd.h; values restated, not included) */
#define ORC_ST_EXITED      0x01  /* left the unit disk, exit refined by bisection            */
#define ORC_ST_STEP_CAP    0x02  /* max_steps reached (reference would loop forever)          */
#define ORC_ST_BISECT_CAP  0x04  /* a bisection hit its iteration cap                         */
#define ORC_ST_OOB         0x08  /* a table lookup fell outside the valid cell range, clamped */
#define ORC_ST_CEASED      0x10  /* a9 only: ray left 1+2h and was ceased                     */

#define ORC_BISECT_CAP 200

/* analytic medium family  c = 1 + A sin(wx x) sin(wy y),
 * grad c = [gx cos(wx x) sin(wy y), gy sin(wx x) cos(wy y)]
 * NonObstacleDemo.jl:12,23 -> A=0.3 wx=wy=1.5*pi gx=gy=0.45*pi
 * ObstacleDemo.jl:12,18    -> A=0.2 wx=2*pi wy=pi gx=2*(0.2*pi) gy=0.2*pi            */
typedef struct { double A, wx, wy, gx, gy; } orc_sinprod;

/* obstacle family: kind 0 none, 1 circle r - r0 (ObstacleDemo.jl:20-34 as shipped),
 * 2 trefoil r - (r0 - rho_ob cos 3 theta) with the normal of ObstacleDemo.jl:31, which carries a rho of
 * its own (the commented alternate, :23-24 rho = 0.3 and :30-31 rho = 0.2: the "cavity")       */
typedef struct {
    int kind; double r0; double rho_ob, rho_n;
    int trig;              /* trefoil only: 0 = atan2/cos/sin of libm as written, 1 = x/r, y/r and the
                              triple-angle identities (what the HIP kernels evaluate; ray_oracle.c) */
} orc_obstacle;
double orc_ob_val(const orc_obstacle *ob, double x, double y);
void   orc_ob_normal(const orc_obstacle *ob, double x, double y, double n[2]);

/* second analytic family, the commented alternates of the demos (NonObstacleDemo.jl:7-11,15-22 with four
 * terms; ObstacleDemo.jl:9-11,15-17 with two):  c = 1 + sum_k a_k sin(pi r_k),  r_k = |(x,y) - (x_k,y_k)|,
 * grad c = sum_k a_k pi cos(pi r_k)/r_k * [(x - x_k), (y - y_k)]                                        */
#define ORC_RADIAL_MAX 4
typedef struct { int K; double a[ORC_RADIAL_MAX], x[ORC_RADIAL_MAX], y[ORC_RADIAL_MAX]; } orc_radial;

typedef struct {
    int kind;              /* 0 = sinprod analytic, 1 = grid tables, 2 = radial sine sum */
    orc_sinprod sp;
    int N; double ext;     /* grid */
    const double *eval, *grad, *p;
    int arith;             /* 0 = as written (un-fused), 1 = "fused" flavour (ray_oracle.c): grid -- contracted DiscreteHamilton
                            * and RK4 combination; analytic kinds (with trig = 1) -- contracted Hamilton, sin/cos anchors
                            * shared by consecutive steps, the same RK4 combination */
    orc_radial rad;
    int trig;              /* sin/cos of the analytic media and of the initial directions: 0 = the host libm (the
                              reference calls Julia's sin/cos), 1 = the "reproducible" flavour (ray_oracle.c: a fixed
                              sequence of +,-,*,fma -- what the HIP kernels evaluate, the same bits on any machine) */
} orc_medium;

void   orc_linspace(double a, double b, int n, double *out);
void   orc_q4(const double *c, int N, int I, int J, double dx, double q[4]);
void   orc_build_tables(const double *c, int N, double ext, double *eval, double *grad);
int    orc_discrete_hamilton(const double X[4], const double *eval, const double *grad,
                             const double *p, int N, double H[4]);
int    orc_discrete_hamilton_fused(const double X[4], const double *eval, const double *grad,
                                   const double *p, int N, double H[4]);
void   orc_hamilton_sinprod(const double X[4], const orc_sinprod *sp, double H[4]);
void   orc_hamilton_radial(const double X[4], const orc_radial *rd, double H[4]);

/* initial boundary phase-space grid, rows for sources [src_begin, src_end) */
/* the reproducible flavour's building blocks, exported for the CPU tests that measure them against long double */
void   orc_rep_sincos(double a, double *s, double *c);
void   orc_rep_shift(double s0, double c0, double d, double *s, double *c);

void   orc_init_states(const orc_medium *med, int ns, int nd, double dir_lo, double dir_hi,
                       int src_begin, int src_end, double *m, int64_t ld);

/* a6 / a7: shoot rows [0,num) of m from m[:,1:4]; obstacle may be kind 0.
 * time_unit: 0 = seconds (a6: t += dt), 1 = steps (a7: t += 1).                      */
void   orc_shoot(const orc_medium *med, const orc_obstacle *ob, double dt, int time_unit,
                 int max_steps, double *m, int64_t num, int64_t ld,
                 int32_t *status, int32_t *nsteps, int32_t *nreflect, int nthreads);

/* a8 (mode 1: overshoot) / a9 X-only (mode 0: shorten last step + cease) */
void   orc_trace_fixed_time(const double *eval, const double *grad, const double *p, int N,
                            double dt, int mode, int arith, const double *m_in, int64_t num, int64_t ld,
                            double *X_out, int64_t ldx, int32_t *status, int32_t *nsteps,
                            int nthreads);

/* NEXT-1: hess table, and the complete ScatterForwardOperator (X, rho, dense M) -- ray_oracle.c */
void   orc_build_hess(const double *grad, int N, double ext, double *hess);
void   orc_forward_operator(const double *eval, const double *grad, const double *hess, const double *p, int N,
                            double timeStep, int arith, const double *m_in, int64_t num, int64_t ld,
                            double *X_out, int64_t ldx, double *rho_out, double *M_dense,
                            int32_t *status, int32_t *nsteps, int nthreads);

/* a11 */
void   orc_misfit(const double *m, const double *obs, int64_t num, int64_t ld,
                  double *mismatch /* 4*num or NULL */, double out[2] /* sum mismatch^2, sum target^2 */,
                  uint8_t *mask /* num or NULL */, double thresh);

#ifdef __cplusplus
}
#endif
#endif
