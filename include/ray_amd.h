/*
 * ray_amd.h -- C ABI of libray_amd.so: the MI355X (gfx950) forward ray-shooting core that
 * replaces the per-ray loops of lowrank/ray (reference citations are file:line under the
 * reference tree; SURVEY.md section 8 is the scope table).
 *
 * The reference has no FFI: its boundary is the set of Julia call sites
 *   ScatterRelation            NonObstacle.jl:54    (called NonObstacleDemo.jl:32)
 *   ScatterRelationObstacle    Obstacle.jl:11       (called ObstacleDemo.jl:43,306)
 *   ScatterForwardOperator     NonObstacle.jl:189   (called NonObstacle.jl:333, ObstacleDemo.jl:199)
 *   NonReflectionTrace         Obstacle.jl:275
 *   misfit / objective / perc  NonObstacle.jl:287,336,372; ObstacleDemo.jl:202,219-226,252
 * A Julia shim (julia/RayAMD.jl) and a Python mirror (ray_amd/) keep those names and ccall /
 * ctypes into the entry points below.  INTEGRATION.md shows the bindings.
 *
 * Conventions
 *   - every function returns int: 0 = RAY_OK, negative = error class; ray_last_error() gives
 *     a thread-local message.  No exceptions cross the ABI, no callbacks into the caller.
 *   - arrays are caller-owned, fp64, COLUMN-major exactly as the reference lays them out:
 *       m, s : (num, 9) with leading dimension ld -> element (r,k) at m[r + k*ld],
 *              columns [x0 y0 px0 py0 x1 y1 px1 py1 t]      (NonObstacle.jl:58,63,95-96)
 *       c    : (N, N), first index = x                        (NonObstacle.jl:102,306)
 *   - plain entry points take HOST pointers and are synchronous (results valid on return,
 *     which is what a blocking Julia ccall needs).  *_dev entry points take DEVICE pointers,
 *     enqueue on the context's stream and return immediately; ray_ctx_synchronize() waits.
 *   - one ray_ctx per GPU; a ctx and its media must be used from one thread at a time.
 *   - function-valued arguments of the reference (c, grad c, ob, grad ob closures) cannot
 *     cross a C ABI onto a GPU: media and obstacles are closed parametric families that cover
 *     what the reference's demos ship, plus the gridded medium.
 */
#ifndef RAY_AMD_H
#define RAY_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAY_OK               0
#define RAY_ERR_INVALID     -1   /* bad argument / shape */
#define RAY_ERR_HIP         -2   /* HIP runtime error (message holds hipGetErrorString) */
#define RAY_ERR_NOMEM       -3
#define RAY_ERR_UNSUPPORTED -4

/* per-ray status word (bit set).  The reference has no such word: a trapped ray loops forever
 * (NonObstacle.jl:70, Obstacle.jl:27), bisections are uncapped (NonObstacle.jl:85,
 * Obstacle.jl:48,77) and out-of-range cells are undefined behaviour under @inbounds.        */
#define RAY_ST_EXITED      0x01  /* left the unit disk; exit point refined by bisection       */
#define RAY_ST_STEP_CAP    0x02  /* max_steps reached                                         */
#define RAY_ST_BISECT_CAP  0x04  /* a bisection hit RAY_BISECT_CAP iterations                 */
#define RAY_ST_OOB         0x08  /* a table lookup left the valid cell range and was clamped  */
#define RAY_ST_CEASED      0x10  /* fixed-time trace: left radius 1+2h and was ceased         */
#define RAY_BISECT_CAP     200

typedef struct ray_ctx    ray_ctx;     /* one per GPU */
typedef struct ray_medium ray_medium;  /* wavespeed model resident in HBM */

/* ---- context ------------------------------------------------------------------------- */
/* 101 since ray_ctx_set_stream takes (ctx, mode, stream); 100 had (ctx, stream) -- a caller built against the older
 * header must check the version before it passes three arguments.                                               */
int         ray_version(void);
/* "hip <runtime version> <path>; rccl <version> <path>": the HIP runtime and the RCCL the dynamic loader bound this
 * library's calls to in THIS process.  A process that loaded another copy first (torch bundles its own libamdhip64 /
 * librccl) runs the library -- ray_group's collectives included -- on that copy: a multi-GPU job wants ONE RCCL, so
 * check this string where the job starts (bench.py prints it in its JSON line).  No
 * counterpart in the reference (Julia's Distributed has no such ambiguity).                                       */
int         ray_runtime_info(char *buf, size_t len);
const char *ray_last_error(void);
int         ray_device_count(int *n);
int         ray_ctx_create(int device, ray_ctx **out);
/* also releases the device memory of every medium / ray_frechet handle still alive on the context and
 * detaches them: their own destroy calls stay valid afterwards (and free the host struct only), every
 * other call on them reports RAY_ERR_INVALID.                                                         */
int         ray_ctx_destroy(ray_ctx *ctx);
/* The stream the *_dev entry points enqueue on (and the host-pointer ones stage through).
 *   RAY_STREAM_OWN       the context's own hipStreamNonBlocking stream (the default); hip_stream is ignored
 *   RAY_STREAM_BORROWED  the caller's hipStream_t exactly as given: NULL is the legacy default (null)
 *                        stream, so work the caller orders on that stream (torch's default stream, an RCCL
 *                        collective issued on it) is ordered with the library's kernels
 * Waits for the stream being left before switching.                                                    */
#define RAY_STREAM_OWN      0
#define RAY_STREAM_BORROWED 1
int         ray_ctx_set_stream(ray_ctx *ctx, int mode, void *hip_stream);
int         ray_ctx_synchronize(ray_ctx *ctx);
/* frees the device memory the context keeps between calls (ray_forward_operator's grow-only workspaces
 * and the parked CSR arrays of the last destroyed ray_frechet); the next call allocates again.        */
int         ray_ctx_release_workspaces(ray_ctx *ctx);
/* Arithmetic flavour of the right-hand sides (Hamilton NonObstacle.jl:1-4, DiscreteHamilton :30-42) and the RK4
 * combinations (:72-76,161-171).  The reference evaluates them under @fastmath, so its own results are
 * defined only up to multiply-add contraction:
 *   RAY_ARITH_EXACT  every operator one IEEE rounding in the reference's order (= un-fused code); analytic
 *                    media: sin/cos of a step's first stage in full, stages 2-4 advanced from it
 *   RAY_ARITH_FUSED  multiply-adds contracted (default; ~1.5x fewer fp64 instructions; differs from EXACT by
 *                    rounding only: ~1e-12 relative on exit states, bound 1e-10); analytic media: one full
 *                    sin/cos evaluation serves several steps, the stages are advanced from it
 * Each flavour is bit-identical to the oracle's flavour of the same name.  Exit/reflection bisections and
 * the misfit are identical in both.                                                                       */
#define RAY_ARITH_EXACT 0
#define RAY_ARITH_FUSED 1
int         ray_ctx_set_arith(ray_ctx *ctx, int arith);

/* Options of the host-pointer entry points.
 *   RAY_OPT_SKIP_INIT_COLUMNS  ray_shoot_to_exit copies back columns 5..9 only (the caller already holds the
 *                              initial states or does not need them): 44 % fewer bytes over PCIe             */
#define RAY_OPT_SKIP_INIT_COLUMNS 1
/*   RAY_OPT_SOURCE_STRIDE      s >= 1 (default 1): ray_shoot_to_exit(_dev) shoots the sources src_begin, src_begin + s,
 *                              ... below src_end; m has ceil((src_end - src_begin) / s) * nd rows, row i*nd + j is
 *                              source src_begin + i*s.  Rank r of n with (src_begin, src_end, s) = (r, ns, n) takes a
 *                              COMB of the source circle instead of an arc: every rank then sees every part of the
 *                              medium and the ranks' ray-step counts agree to a fraction of a percent, where
 *                              contiguous theta-blocks differ by 16 % on the shipped medium (no counterpart in the
 *                              reference, whose workers take whatever @parallel hands them).                      */
#define RAY_OPT_SOURCE_STRIDE 2
/*   RAY_OPT_FAN_DIRECTIONS     nd >= 0 (default 0 = unknown): a hint for the fixed-time trace entry points: the rows of m
 *                              are fans of nd consecutive directions per source (row i*nd + j), as ScatterRelation lays
 *                              them out (NonObstacle.jl:63).  The trace kernel then walks them like the shooting
 *                              kernel does -- waves of 2 adjacent sources x 32 directions, longest chords first, XCD
 *                              arcs -- instead of 64 consecutive rows per wave: same results, row for row and bit for
 *                              bit, 1.5x the rate on a 2048 x 2048 sweep.  Ignored when num is not a multiple of nd.
 *                              With the option at 0 the host-pointer entry points (ray_trace_fixed_time,
 *                              ray_trace_paths) look for such fans in m themselves -- runs of equal length with one
 *                              starting point each -- so the reference's own call NonReflectionTrace(c, m, ext, dt)
 *                              (Obstacle.jl:275) needs no hint; the _dev entry points take m as it is told.         */
#define RAY_OPT_FAN_DIRECTIONS 3
/*   RAY_OPT_TUNE_*             kernel shapes for experiments and tests; results never depend on them.  They are options of
 *                              the context on purpose: the library reads NO environment variable, so nothing outside the
 *                              caller's own calls changes what its process launches.
 *                                BLOCK_THREADS 64 | 128 | 256 (default)   workgroup size of the integration kernels
 *                                WAVE_DIRS 0 (automatic) | 4 .. 64        directions per wavefront of the fan mapping
 *                                SORT_KEY_BITS 0 (automatic) | 64         key width of the Frechet merge's sort
 *                                FRECHET_LDS_ROWS -1 (default) | n        longest ray whose records the merge keeps in LDS
 *                                MERGE_GRID 0 (one per ray) | n           workgroups of the merge's emit kernel
 *                                FRECHET_SLAB_MB 0 (a quarter of free memory) | n   trajectory slab budget per chunk of rays
 *                                MERGE_ENTRIES 0 (default) | 1            1: the Frechet merge sorts the twelve entries of every record
 *                                                                         (round-2 form, kept as a cross-check) instead of its four segments */
#define RAY_OPT_TUNE_BLOCK_THREADS    4
#define RAY_OPT_TUNE_WAVE_DIRS        5
#define RAY_OPT_TUNE_SORT_KEY_BITS    6
#define RAY_OPT_TUNE_FRECHET_LDS_ROWS 7
#define RAY_OPT_TUNE_MERGE_GRID       8
#define RAY_OPT_TUNE_FRECHET_SLAB_MB  9
#define RAY_OPT_TUNE_MERGE_ENTRIES    10
int         ray_ctx_set_option(ray_ctx *ctx, int option, int value);
/* The host-pointer entry points stage through device buffers the context keeps (grow-only; released by
 * ray_ctx_release_workspaces / ray_ctx_destroy) and copy to / from the caller's arrays.  From pageable memory
 * those copies run at a fraction of the PCIe rate: a caller that keeps its m / s arrays alive across calls (the
 * reference's SharedArrays live for the whole reconstruction, NonObstacle.jl:58,196) can page-lock them once. */
int         ray_host_register(void *ptr, size_t bytes);
int         ray_host_unregister(void *ptr);

/* out[k] = linspace(a, b, n)[k+1] exactly as the library (and Julia's TwicePrecision ranges,
 * NonObstacle.jl:55-56,191) evaluates it: the correctly rounded a + k(b-a)/(n-1).  Host only;
 * lets a caller fill c[i,j] = waveSpeed(p[i], p[j]) (NonObstacle.jl:306) on the same nodes.   */
int         ray_linspace(double a, double b, int n, double *out);

/* ---- media --------------------------------------------------------------------------- */
/* Gridded medium.  Replaces the eval/grad SharedArray tables and their build loops
 * (NonObstacle.jl:191-210, dup. Obstacle.jl:224-241, 277-294) and Q4 (NonObstacle.jl:103-105).
 * c is copied; the caller keeps ownership.  N >= 4.                                          */
int ray_medium_create_grid(ray_ctx *ctx, const double *c_host, int N, double ext, ray_medium **out);
/* new nodal values, same N and ext: re-upload and rebuild the tables (one reconstruction
 * iteration, NonObstacle.jl:333 -> :191-210).  _dev takes a device pointer, asynchronous.   */
int ray_medium_update_grid(ray_medium *med, const double *c_host);
int ray_medium_update_grid_dev(ray_medium *med, const double *c_dev);
/* copy the tables back in the reference's layout: eval (N-1,N-1,4), grad (N-2,N-2,8),
 * p (N) = linspace(-ext,ext,N); any pointer may be NULL.  For parity tests of the build.    */
int ray_medium_export_tables(ray_medium *med, double *eval_host, double *grad_host, double *p_host);
/* Analytic family  c = 1 + A sin(wx x) sin(wy y),  grad c = [gx cos(wx x) sin(wy y),
 * gy sin(wx x) cos(wy y)].  NonObstacleDemo.jl:12,23 = (0.3, 1.5pi, 1.5pi, 0.45pi, 0.45pi);
 * ObstacleDemo.jl:12,18 = (0.2, 2pi, pi, 2(0.2pi), 0.2pi).                                   */
int ray_medium_create_sinprod(ray_ctx *ctx, double A, double wx, double wy, double gx, double gy,
                              ray_medium **out);
/* Second analytic family, the commented alternates of the demos:  c = 1 + sum_k a_k sin(pi r_k),
 * r_k = |(x,y) - (x_k,y_k)|,  grad c = sum_k a_k pi cos(pi r_k)/r_k [(x - x_k), (y - y_k)], K <= 4.
 * NonObstacleDemo.jl:7-11,15-22 = a (0.2,0.4,0.2,0.3), centres (0.5,0.2) (-0.4,-0.3) (0.3,-0.4) (-0.2,0.1);
 * ObstacleDemo.jl:9-11,15-17 = a (0.4,0.4), centres (0.5,0.2) (-0.4,-0.3).                            */
int ray_medium_create_radial(ray_ctx *ctx, int K, const double *a, const double *xk, const double *yk,
                             ray_medium **out);
int ray_medium_destroy(ray_medium *med);

/* obstacle family (zero level set + normal).  CIRCLE: ob = sqrt(x^2+y^2) - r0, n = (x,y)/r
 * renormalised (ObstacleDemo.jl:20-34 as shipped).  TREFOIL, the commented alternate of the same
 * lines (the "cavity"): ob = r - (r0 - rho cos 3 theta) with rho = reserved[0] (ObstacleDemo.jl:23-24:
 * r0 = 0.4, rho = 0.3) and n = [x,y]/r - 3 rho' sin(3 theta)/r [-y,x]/r renormalised with
 * rho' = reserved[1] (:30-31: 0.2 -- the reference's normal carries a constant of its own, kept).  */
#define RAY_OB_NONE    0
#define RAY_OB_CIRCLE  1
#define RAY_OB_TREFOIL 2
typedef struct { int32_t kind; int32_t pad_; double r0; double reserved[2]; } ray_obstacle;

#define RAY_TIME_SECONDS 0   /* t += dt, exit correction t -= (1-mid)*dt  (NonObstacle.jl:77,90) */
#define RAY_TIME_STEPS   1   /* t += 1,  exit correction t -= (1-mid)     (Obstacle.jl:39,53)    */

typedef struct {
    int32_t ns, nd;             /* boundary phase-space grid: ns sources x nd directions        */
    int32_t src_begin, src_end; /* this call shoots sources [src_begin, src_end) (theta-block)  */
    double  dir_lo, dir_hi;     /* direct = linspace(dir_lo, dir_hi, nd+2)[2:nd+1]              */
    double  dt;                 /* RK4 step                                                    */
    int32_t time_unit;          /* RAY_TIME_SECONDS | RAY_TIME_STEPS                           */
    int32_t max_steps;          /* <=0 selects the default 1000000                             */
} ray_shoot_params;

/* ---- shooting to the boundary: ScatterRelation (NonObstacle.jl:54-99) and
 *      ScatterRelationObstacle (Obstacle.jl:11-97; pass a CIRCLE obstacle and RAY_TIME_STEPS).
 * m has (src_end-src_begin)*nd rows; row (i-src_begin)*nd + j is source i, direction j
 * (NonObstacle.jl:63).  All nine columns are written.  status/nsteps/nreflect (int32 per row)
 * may each be NULL.                                                                           */
int ray_shoot_to_exit(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob,
                      const ray_shoot_params *prm, double *m_host, int64_t ld,
                      int32_t *status, int32_t *nsteps, int32_t *nreflect);
int ray_shoot_to_exit_dev(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob,
                          const ray_shoot_params *prm, double *m_dev, int64_t ld,
                          int32_t *status_dev, int32_t *nsteps_dev, int32_t *nreflect_dev);
/* same loops from explicit initial states m[:,1:4] (the per-ray part, NonObstacle.jl:67-97 /
 * Obstacle.jl:23-95); columns 5..9 are written.                                               */
int ray_shoot_states(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob, double dt,
                     int32_t time_unit, int32_t max_steps, double *m_host, int64_t num, int64_t ld,
                     int32_t *status, int32_t *nsteps, int32_t *nreflect);
int ray_shoot_states_dev(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob, double dt,
                         int32_t time_unit, int32_t max_steps, double *m_dev, int64_t num,
                         int64_t ld, int32_t *status_dev, int32_t *nsteps_dev,
                         int32_t *nreflect_dev);

/* ---- fixed-time tracing on a gridded medium ------------------------------------------------
 * RAY_TRACE_SHORTEN_CEASE : the X-trajectory of ChunkProcessing! (NonObstacle.jl:146-157,
 *                           161-164,171,177-181,185) = the `s` output of ScatterForwardOperator
 * RAY_TRACE_OVERSHOOT     : NonReflectionTrace (Obstacle.jl:306-326)
 * Reads m[:,1:4] and T = m[:,9]; writes X_out (num,4) column-major with leading dim ldx.      */
#define RAY_TRACE_SHORTEN_CEASE 0
#define RAY_TRACE_OVERSHOOT     1
int ray_trace_fixed_time(ray_ctx *ctx, const ray_medium *med, const double *m_host, int64_t num,
                         int64_t ld, double dt, int32_t mode, double *X_host, int64_t ldx,
                         int32_t *status, int32_t *nsteps);
int ray_trace_fixed_time_dev(ray_ctx *ctx, const ray_medium *med, const double *m_dev, int64_t num,
                             int64_t ld, double dt, int32_t mode, double *X_dev, int64_t ldx,
                             int32_t *status_dev, int32_t *nsteps_dev);

/* ---- complete ScatterForwardOperator (NonObstacle.jl:189-243; SURVEY.md section 8f NEXT-1) -----------
 * Per ray: the traced end state X (= the `s` output, identical to RAY_TRACE_SHORTEN_CEASE), the
 * Jacobian rho (NonObstacle.jl:149,166-172; rho_host: num x 16 row-major, may be NULL) and its
 * 4 x N^2 block rho*Phi of the Frechet matrix M (NonObstacle.jl:158-159,174-184; dV :107-139,
 * DiscreteHamilton with Hessian :6-28, hess table :212-218).
 * The reference assembles M densely (4num x N^2 SharedArray) and compresses it with
 * sparse()/droptol! (:242).  Here M is assembled sparse on the device and handed over as CSR:
 *   row 4*ray + component (0-based; = reshape(M, 4num, N^2) at :242), column i + j*N (0-based node,
 *   first index x), columns sorted within a row, duplicates summed in emission order, entries with
 *   |v| <= droptol removed (the reference passes 1e-12; droptol < 0 keeps every stencil entry).
 * A Julia caller gets the reference's SparseMatrixCSC with
 *   copy(transpose(SparseMatrixCSC(N^2, 4num, indptr .+ 1, indices .+ 1, data)))  (julia/RayAMD.jl).
 * The trajectory nodes pass through a device slab (one 128-byte record per ray-step plus 52 B of sort
 * hand-over); when that would exceed a quarter of the free memory (or 16 GB) the rays go through in
 * chunks and the pieces are concatenated -- the result is identical.
 * On the device M lives as BLOCK ROWS -- per ray its distinct nodes, each with the column index and the
 * values of the ray's four matrix rows (36 B per node instead of 48 B of CSR; droptol applied by the
 * readers) -- which is what the loop passes below read; the CSR arrays are made from them when
 * ray_frechet_export_csr / ray_frechet_csr_dev first ask.
 * Limits: 4*num < 2^31; at most 1364 steps per ray (T/dt).  Parity is by tolerance: the reference
 * goes through LAPACK inv / BLAS gemm for rho and Phi.                                           */
typedef struct ray_frechet ray_frechet;
int ray_forward_operator(ray_ctx *ctx, ray_medium *med, const double *m_host, int64_t num, int64_t ld,
                         double dt, double droptol, double *X_host, int64_t ldx, double *rho_host,
                         int32_t *status, int32_t *nsteps, ray_frechet **out);
int ray_frechet_shape(ray_frechet *f, int64_t *n_rows, int64_t *n_cols, int64_t *nnz);
/* indptr: 4num + 1, indices / data: nnz (host buffers) */
int ray_frechet_export_csr(ray_frechet *f, int64_t *indptr, int32_t *indices, double *data);
/* the same three arrays in HBM (valid until ray_frechet_destroy; indices / data are converted on the
 * context's stream by the first call that asks for them: pass NULL for both to get indptr alone) */
int ray_frechet_csr_dev(ray_frechet *f, const int64_t **indptr, const int32_t **indices, const double **data);
/* The block rows where they live in HBM (no counterpart in the reference, whose M is a host CSC): ray j owns
 * entries uoff[j] .. uoff[j+1] (uoff: num + 1); entry u is node ucols[u] (sorted within a ray) with
 * M[4j + r, ucols[u]] = uvals[4u + r], r = 0..3, kept when |value| > droptol (all when droptol < 0).  */
int ray_frechet_block_rows_dev(ray_frechet *f, int64_t *n_entries, const int64_t **uoff, const int32_t **ucols,
                               const double **uvals);
/* Waits for the context's stream, then releases the handle.  Its device arrays are parked on the
 * context (the largest of each kind) and handed to the next ray_forward_operator that fits in them, and
 * the operator's intermediates live in grow-only workspaces of the context: a reconstruction loop that
 * builds M every iteration (NonObstacle.jl:331-333) allocates nothing after the first one.  All of it is
 * released by ray_ctx_destroy (or earlier by ray_ctx_release_workspaces).                              */
int ray_frechet_destroy(ray_frechet *f);

/* ---- passes of the reconstruction loop over M (NonObstacle.jl:340-367; SURVEY.md section 8f NEXT-2) ----
 * The normal-equation solve A\b (CHOLMOD, :353-356) stays with the caller; these are the O(nnz)
 * passes around it, run on the device block rows.  Vector lengths: fidelity / x / b = N^2 (node order
 * i + j*N), v / mismatch / y / residual = 4 num (ray-major, like `mismatch` at :336), dofs / ray_mask =
 * num.  ray_mask[j] != 0 selects the four rows of ray j (the reference's `order`, :345-349; NULL = all).
 *   dofs[j]   = nnz(M[4j-3,:]) - sum(fidelity[find(M[4j-3,:])])                     (:340-342)
 *   b         = M[order,:]' * v[order]            (take b[Idx] for :354; fp64 atomics: 1e-12)
 *   y         = M * x
 *   fidelity_update: residual = |M[k,:] x - mismatch[k]| on the selected rows (0 elsewhere; x must be
 *               zero outside Idx to reproduce :359), and where residual < rejection
 *               fidelity[find(M[k,:])] = max(fidelity, 1 - decay*residual)           (:362-367)
 *   normal_matvec: w = M[order,:]' * (M[order,:] * x) -- the action of the reference's MtM (:353: it forms
 *               M[order,Idx]' * M[order,Idx] and factorises it, which a 1024^2 grid does not allow; a Krylov
 *               solve of (MtM + penalty R) x = b needs only this action and normal_diag)       (fp64 atomics)
 *   normal_diag:   d[j] = sum over the selected rows of M[k,j]^2 = diag(MtM)
 * `_dev` forms take device pointers and are asynchronous on the context's stream.              */
int ray_frechet_dofs(ray_frechet *f, const double *fidelity, double *dofs);
int ray_frechet_dofs_dev(ray_frechet *f, const double *fidelity_dev, double *dofs_dev);
int ray_frechet_matvec(ray_frechet *f, const double *x, double *y);
int ray_frechet_matvec_dev(ray_frechet *f, const double *x_dev, double *y_dev);
int ray_frechet_matvec_t(ray_frechet *f, const double *v, const uint8_t *ray_mask, double *b);
int ray_frechet_matvec_t_dev(ray_frechet *f, const double *v_dev, const uint8_t *ray_mask_dev, double *b_dev);
int ray_frechet_normal_matvec(ray_frechet *f, const double *x, const uint8_t *ray_mask, double *w);
int ray_frechet_normal_matvec_dev(ray_frechet *f, const double *x_dev, const uint8_t *ray_mask_dev, double *w_dev);
int ray_frechet_normal_diag(ray_frechet *f, const uint8_t *ray_mask, double *d);
int ray_frechet_normal_diag_dev(ray_frechet *f, const uint8_t *ray_mask_dev, double *d_dev);
int ray_frechet_fidelity_update(ray_frechet *f, const double *x, const double *mismatch,
                                const uint8_t *ray_mask, double rejection, double decay,
                                double *fidelity, double *residual);
int ray_frechet_fidelity_update_dev(ray_frechet *f, const double *x_dev, const double *mismatch_dev,
                                    const uint8_t *ray_mask_dev, double rejection, double decay,
                                    double *fidelity_dev, double *residual_dev);

/* ---- ray paths (ScatterRelationObstaclePlot, Obstacle.jl:98-221, without the plotting; SURVEY.md 8f
 * NEXT-4).  As ray_shoot_states, and additionally the polyline `res` of every ray: its start, the
 * position after every step, each reflection point (twice, as the reference appends R and then the
 * reflected X), the exit point (Obstacle.jl:114,142,175,211).  path_host: [ray][point][x,y], max_pts
 * points per ray; npts[ray] = points the ray produced (> max_pts: the tail was dropped).             */
int ray_shoot_paths(ray_ctx *ctx, const ray_medium *med, const ray_obstacle *ob, double dt,
                    int32_t time_unit, int32_t max_steps, double *m_host, int64_t num, int64_t ld,
                    int32_t max_pts, double *path_host, int32_t *npts, int32_t *status,
                    int32_t *nsteps, int32_t *nreflect);

/* NonReflectionPlot (Obstacle.jl:222-274, without the plotting): the fixed-time trace of
 * ray_trace_fixed_time plus the polyline `res` of every ray -- its start and the position after every
 * step (Obstacle.jl:256,266).  Layout of path_host / npts as in ray_shoot_paths.                        */
int ray_trace_paths(ray_ctx *ctx, const ray_medium *med, const double *m_host, int64_t num, int64_t ld,
                    double dt, int32_t mode, int32_t max_pts, double *path_host, int32_t *npts,
                    double *X_host, int64_t ldx, int32_t *status, int32_t *nsteps);

/* ---- ray classification (Utility.jl:1-39; ObstacleDemo.jl:48-67; SURVEY.md section 8f NEXT-3) -----------
 * m: (ns*nd, >=8 columns), rows source-major as ScatterRelation(Obstacle) writes them.
 *   ortho[r] = similarity(m[r,1:2], m[r,5:6]) > 0.995 && similarity(m[r,3:4], m[r,7:8]) < -0.995
 *              (the `orthoIndex` filter, ObstacleDemo.jl:48-49; may be NULL)
 *   arg      = alignment(atan2.(m[.,6], m[.,5])) per source (Utility.jl:5-13; ObstacleDemo.jl:59-61);
 *              num doubles, workspace and output (host form: may be NULL)
 *   lo, hi   = derivativeCheck(arg) per source, 1-BASED like the reference (Utility.jl:15-39): the
 *              unbroken rays of source s are rows 1..lo and hi..nd of its block (ObstacleDemo.jl:63-64).
 * atan2 / cos are the device library's (<= 2 ulp from the host's): thresholds are compared on those. */
int ray_classify(ray_ctx *ctx, const double *m_host, int64_t ns, int64_t nd, int64_t ld, uint8_t *ortho,
                 double *arg, int32_t *lo, int32_t *hi);
int ray_classify_dev(ray_ctx *ctx, const double *m_dev, int64_t ns, int64_t nd, int64_t ld,
                     uint8_t *ortho_dev, double *arg_dev, int32_t *lo_dev, int32_t *hi_dev);

/* ---- misfit (NonObstacle.jl:287,336,372; ObstacleDemo.jl:202,219-226,252) ------------------
 * out[0] = sum_r,k (m[r,4+k]-obs[r,4+k])^2, out[1] = sum_r,k m[r,4+k]^2  (objective =
 * sqrt(out[0]/out[1])); mask[r] = |m[r,5:8]-obs[r,5:8]| / |m[r,5:8]| < perc_thresh;
 * mismatch[4r+k] ray-major as the reference's reshape of the transposed difference.
 * mask and mismatch may be NULL.  Sums are reduced in a fixed, run-to-run reproducible tree
 * order (differs from the reference's serial order by rounding only).                         */
int ray_misfit(ray_ctx *ctx, const double *m_host, const double *obs_host, int64_t num, int64_t ld,
               double out[2], uint8_t *mask, double perc_thresh, double *mismatch);
int ray_misfit_dev(ray_ctx *ctx, const double *m_dev, const double *obs_dev, int64_t num,
                   int64_t ld, double *out_dev, uint8_t *mask_dev, double perc_thresh,
                   double *mismatch_dev);

/* ---- node-wide fan-out: ONE host process drives every GPU of the node -----------------------------------------
 * Replaces what the reference's callers get from addprocs(3) (NonObstacleDemo.jl:1, ObstacleDemo.jl:1),
 * `@sync @parallel` (NonObstacle.jl:59; Obstacle.jl:16), the per-pid remotecall_wait of ScatterForwardOperator
 * (NonObstacle.jl:236-240) and the serial misfit sum on the master (NonObstacle.jl:336,372): a single call uses
 * every worker.  A ray_group holds one ray_ctx per listed device; a call launches every member's share at once
 * (member k of G: the comb of sources k, k + G, ... -- equal work per member whatever the medium's asymmetry)
 * and lands each member's rows in the caller's column-major m;
 * nothing is exchanged between GPUs except ONE ncclAllReduce(sum, f64) of {sum mismatch^2, sum target^2} per
 * ray_group_misfit (RCCL: xGMI on the box).  `devices` may name ONE device n times (rehearsal on a one-GPU box;
 * RCCL refuses two ranks on a device): such a group folds the members' sums with a kernel on that device instead.
 * Host-pointer entry points, synchronous like the single-context ones.                                        */
typedef struct ray_group        ray_group;
typedef struct ray_group_medium ray_group_medium;     /* one replica of the medium per member */
int ray_group_create(const int *devices, int n, ray_group **out);
int ray_group_destroy(ray_group *g);                  /* also releases its media (their destroy stays valid) */
int ray_group_size(const ray_group *g, int *n, int *uses_rccl);
int ray_group_set_arith(ray_group *g, int arith);
int ray_group_medium_create_grid(ray_group *g, const double *c_host, int N, double ext, ray_group_medium **out);
int ray_group_medium_create_sinprod(ray_group *g, double A, double wx, double wy, double gx, double gy,
                                    ray_group_medium **out);
int ray_group_medium_update_grid(ray_group_medium *med, const double *c_host);
int ray_group_medium_destroy(ray_group_medium *med);
/* ScatterRelation / ScatterRelationObstacle (NonObstacle.jl:54-99; Obstacle.jl:11-97) for sources
 * [prm->src_begin, prm->src_end) split over the members; arguments as ray_shoot_to_exit.  m_host may be NULL:
 * the slabs then stay on the devices only (forward-only sweeps that want the misfit alone).                  */
int ray_group_shoot_to_exit(ray_group *g, const ray_group_medium *med, const ray_obstacle *ob,
                            const ray_shoot_params *prm, double *m_host, int64_t ld,
                            int32_t *status, int32_t *nsteps, int32_t *nreflect);
/* The fixed-time traces of a reconstruction iteration over the members: NonReflectionTrace (Obstacle.jl:275-329,
 * mode RAY_TRACE_OVERSHOOT) and the trajectory part of ScatterForwardOperator (NonObstacle.jl:141-187, mode
 * RAY_TRACE_SHORTEN_CEASE), which the reference deals to its workers pid by pid (NonObstacle.jl:236-240).
 * Arguments as ray_trace_fixed_time; nd = rows per source when m is laid out as ScatterRelation returns it (0: the
 * library looks for such fans itself).  Fans are dealt to the members as ray_group_shoot_to_exit deals sources
 * (member k of G: fans k, k + G, ...), rows in any other order in G contiguous blocks.  X_host may be NULL.  Every
 * member keeps its rows and their traced states resident, so ray_group_set_observation + ray_group_misfit apply
 * to the traces as they do to a shoot.                                                                          */
int ray_group_trace_fixed_time(ray_group *g, const ray_group_medium *med, const double *m_host, int64_t num,
                               int64_t ld, int32_t nd, double dt, int32_t mode, double *X_host, int64_t ldx,
                               int32_t *status, int32_t *nsteps);
/* ScatterForwardOperator (NonObstacle.jl:189-243) over the members, as the reference spreads it over its workers
 * (:236-240): rays dealt as ray_group_trace_fixed_time deals them (nd as there), every member building the block rows
 * of its rays concurrently (one host thread per member: the operator exchanges step counts with the host half way).
 * Arguments as ray_forward_operator, without rho.  The result is one ordinary ray_frechet per member:
 *   ray_group_frechet_member      member k's handle -- every ray_frechet_* pass applies to it -- and its rays: `count`
 *                                 runs of nd rays, run i = the caller's rays row0 + i*stride .. + nd
 *   ray_group_frechet_export_csr  the whole M on the host in the caller's row order, i.e. what the reference's master
 *                                 holds when its workers return (:242); sizes from ray_group_frechet_shape          */
typedef struct ray_group_frechet ray_group_frechet;
int ray_group_forward_operator(ray_group *g, ray_group_medium *med, const double *m_host, int64_t num, int64_t ld,
                               int32_t nd, double dt, double droptol, double *X_host, int64_t ldx, int32_t *status,
                               int32_t *nsteps, ray_group_frechet **out);
int ray_group_frechet_shape(ray_group_frechet *f, int64_t *n_rows, int64_t *n_cols, int64_t *nnz);
int ray_group_frechet_member(ray_group_frechet *f, int k, ray_frechet **member, int64_t *row0, int32_t *count,
                             int32_t *nd, int64_t *stride);
int ray_group_frechet_export_csr(ray_group_frechet *f, int64_t *indptr, int32_t *indices, double *data);
/* The two passes of the loop that end in a sum over all rays, with ONE all-reduce of N^2 doubles over the members
 * (RCCL; members sharing a device: a fold kernel in member order) -- the gradient sums of the sharded loop:
 *   matvec_t         b = M[order,:]' * v[order]                  (NonObstacle.jl:354; v: 4 num, ray-major)
 *   normal_matvec    w = M[order,:]' * (M[order,:] * x)          (the action of :353's MtM; x: N^2)
 * v / ray_mask are host arrays in the caller's row order (ray_mask: num entries or NULL = all), b / w host, N^2. */
int ray_group_frechet_matvec_t(ray_group_frechet *f, const double *v, const uint8_t *ray_mask, double *b);
int ray_group_frechet_normal_matvec(ray_group_frechet *f, const double *x, const uint8_t *ray_mask, double *w);
/* valid before and after ray_group_destroy (which detaches the handle: every other call on it then reports
 * RAY_ERR_INVALID instead of touching freed memory)                                                             */
int ray_group_frechet_destroy(ray_group_frechet *f);
/* the observation (NonObstacle.jl:336), rows as the LAST ray_group_shoot_to_exit / ray_group_trace_fixed_time laid
 * them out; columns 5..8 are uploaded once, sharded like m, and stay resident                                  */
int ray_group_set_observation(ray_group *g, const double *obs_host, int64_t num, int64_t ld);
/* misfit of the resident slabs of the last shoot against the resident observation; out / mask / mismatch as
 * ray_misfit (mask, mismatch may be NULL); sums to 1e-12 of the single-context ones (order of the all-reduce)  */
int ray_group_misfit(ray_group *g, double out[2], uint8_t *mask, double perc_thresh, double *mismatch);
/* (ray_group_get_counters, below: rays, ray-steps, launches summed over the members; kernel_ms = the slowest
 * member's, as they run concurrently)                                                                        */

/* Allocation-free variant of ray_misfit_dev.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot allocate -- code under HIP graph capture, tight serving loops that call the misfit every iteration,
 * harnesses that forbid hipMalloc.  Same kernels and bit-identical results; the scratch (2 x 1024 partial sums,
 * 16 KB) is a file-scope __device__ array of the library, one copy per GPU, resolved once per device at
 * ray_ctx_create.  Calls no hipMalloc / hipFree / event creation.  Limit: num <= RAY_MISFIT_STATIC_MAX_RAYS
 * (RAY_ERR_UNSUPPORTED beyond).  Every call on a GPU shares that GPU's scratch: calls that overlap on one GPU
 * (two streams, two contexts on one device) race; calls on one stream are ordered and need no clearing.       */
#define RAY_MISFIT_STATIC_MAX_RAYS ((int64_t)1 << 28)
int ray_misfit_static_dev(ray_ctx *ctx, const double *m_dev, const double *obs_dev, int64_t num,
                          int64_t ld, double *out_dev, uint8_t *mask_dev, double perc_thresh,
                          double *mismatch_dev);

/* ---- counters (replaces the reference's tic/toq table, NonObstacle.jl:332-334) ------------- */
typedef struct {
    uint64_t rays;        /* rays integrated since reset                         */
    uint64_t ray_steps;   /* RK4 steps taken by those rays                       */
    uint64_t launches;    /* integration kernel launches                         */
    double   kernel_ms;   /* summed HIP-event time of those launches (timing on) */
} ray_counters;
int ray_ctx_enable_timing(ray_ctx *ctx, int on);   /* record HIP events around each launch */
int ray_get_counters(ray_ctx *ctx, ray_counters *out);   /* synchronizes the stream */
int ray_reset_counters(ray_ctx *ctx);
int ray_group_get_counters(ray_group *g, ray_counters *out);

#ifdef __cplusplus
}
#endif
#endif /* RAY_AMD_H */
