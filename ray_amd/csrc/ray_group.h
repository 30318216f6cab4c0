// ray_group.h -- node-wide fan-out behind the C ABI (included at the end of ray_amd.hip).
//
// The reference spreads one ScatterRelation / ScatterForwardOperator call over its workers itself:
// addprocs(3) (NonObstacleDemo.jl:1, ObstacleDemo.jl:1), `@sync @parallel` (NonObstacle.jl:59, Obstacle.jl:16),
// per-pid remotecall_wait (NonObstacle.jl:236-240), data through SharedArrays, and sums the misfit serially
// on the master (NonObstacle.jl:336,372).  Here ONE host process drives every GPU of the node: a ray_group
// holds one ray_ctx per device, a call launches every member's share asynchronously -- member k of G takes the
// COMB of sources k, k + G, ... (RAY_OPT_SOURCE_STRIDE of its context): every member sweeps the whole source
// circle, so their ray-step counts agree whatever the medium's asymmetry, where contiguous arcs differ by 16 %
// on the shipped medium and the call would last as long as the slowest -- each member's rows land in the
// caller's column-major m (nd consecutive rows per source, G nd rows apart), and the only
// exchange is ONE ncclAllReduce(sum, f64) of {sum mismatch^2, sum target^2} over RCCL (xGMI on the box).
// A group may name one device several times (rehearsal on a one-GPU box): RCCL refuses two ranks on one
// device, so such a group folds the members' partial sums with a one-block kernel on that device, in member
// order -- same device memory, no collective needed.
#pragma once
#include <rccl/rccl.h>
#include <memory>
#include <thread>

struct GroupMember {
    ray_ctx *ctx = nullptr;
    int device = 0;
    // share of the last ray_group_shoot_to_exit: `count` sources first, first + G, ... of the call, nd rows each
    // (rows = count * nd, device ld = max(rows, 1)); in the caller's arrays source i of the share starts at row
    // (first + i * G) * nd
    int64_t rows = 0;
    int64_t row0 = 0;           // first row of the share in the caller's arrays
    int first = 0, count = 0, nd = 0, G = 1;
    double *d_m = nullptr;      size_t m_cap = 0;
    int32_t *d_int = nullptr;   size_t int_cap = 0;     // status | nsteps | nreflect, rows each
    double *d_obs = nullptr;    size_t obs_cap = 0;
    int64_t obs_rows = -1;
    uint8_t *d_mask = nullptr;  size_t mask_cap = 0;
    double *d_mis = nullptr;    size_t mis_cap = 0;
    double *d_sums = nullptr;                           // [0..2) own sums, [2..4) reduced
    // the group's passes over M: this member's input vector (4 per ray, or N^2), ray mask and N^2 result
    double *d_gv = nullptr;     size_t gv_cap = 0;
    uint8_t *d_gmask = nullptr; size_t gmask_cap = 0;
    double *d_gb = nullptr;     size_t gb_cap = 0;
    hipEvent_t done = nullptr;
};

struct ray_group_frechet;
struct ray_group {
    std::vector<GroupMember> mem;
    bool use_rccl = false;
    std::vector<ncclComm_t> comms;
    const double **d_ptrs = nullptr;                    // fold path: the members' d_sums, on device mem[0].device
    const double **d_bptrs = nullptr;                   // fold path: the members' d_gb
    std::vector<ray_group_medium *> media;
    std::vector<ray_group_frechet *> frechets;          // live handles: detached (g = nullptr) by ray_group_destroy
    // which RCCL this process bound (ray_group_info): "rccl <version> <path of the library that holds ncclAllReduce>"
    std::string rccl_info;
};

// the Frechet matrix of one ray_group_forward_operator: one ordinary ray_frechet per member over its share of the rays
struct GroupShare { int64_t row0 = 0, rows = 0; int count = 0, nd = 0, G = 1; };
struct ray_group_frechet {
    ray_group *g = nullptr;
    std::vector<ray_frechet *> f;
    std::vector<GroupShare> share;
    int64_t num = 0;
    int N = 0;
};

struct ray_group_medium {
    ray_group *g = nullptr;
    std::vector<ray_medium *> med;                      // one per member, tables replicated
};

// fold path (all members on one device): out[0..2) = sum over members of their {sum mismatch^2, sum target^2},
// in member order (fixed -> reproducible)
__global__ void k_group_fold(const double *const *parts, int n, double *out)
{
    if (threadIdx.x < 2) {
        double s = parts[0][threadIdx.x];
        for (int k = 1; k < n; ++k) s = s + parts[k][threadIdx.x];
        out[threadIdx.x] = s;
    }
}

// fold path for the N^2-long results: out[i] = sum over members of parts[k][i], in member order
__global__ void k_group_fold_vec(const double *const *parts, int n, int64_t len, double *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double s = parts[0][i];
    for (int k = 1; k < n; ++k) s = s + parts[k][i];
    out[i] = s;
}

static hipError_t grow(void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) *cap = bytes;
    return e;
}

static void group_sync_all(ray_group *g)
{
    for (GroupMember &m : g->mem) {
        if (!m.ctx) continue;                           // ray_group_create's failure path: members not built yet
        (void)hipSetDevice(m.device);
        (void)hipStreamSynchronize(m.ctx->stream);
    }
}

#define NCCLCHK(expr)                                                                          \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess)                                                                 \
            return fail(RAY_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_));      \
    } while (0)

// One row-indexed array of the caller (w bytes per row: a column of m, status, mask, 4 doubles of mismatch ...)
// <-> the member's packed copy of its share: `count` pieces of nd rows, G nd rows apart on the host side.
static hipError_t comb_copy(const GroupMember &m, void *host, void *dev, size_t w, bool to_host)
{
    if (m.count == 0) return hipSuccess;
    char *h = static_cast<char *>(host) + (size_t)m.row0 * w;
    const size_t piece = (size_t)m.nd * w, hpitch = piece * (size_t)m.G;
    if (m.G == 1)                                       // a group of one: the share is the whole array, one flat copy
        return to_host ? hipMemcpyAsync(h, dev, piece * (size_t)m.count, hipMemcpyDeviceToHost, m.ctx->stream)
                       : hipMemcpyAsync(dev, h, piece * (size_t)m.count, hipMemcpyHostToDevice, m.ctx->stream);
    return to_host ? hipMemcpy2DAsync(h, hpitch, dev, piece, piece, (size_t)m.count, hipMemcpyDeviceToHost, m.ctx->stream)
                   : hipMemcpy2DAsync(dev, piece, h, hpitch, piece, (size_t)m.count, hipMemcpyHostToDevice, m.ctx->stream);
}

extern "C" {

int ray_group_destroy(ray_group *g)
{
    if (!g) return RAY_OK;
    group_sync_all(g);
    for (ray_group_medium *gm : g->media) {             // detach: their own destroy frees the struct only
        for (ray_medium *m : gm->med) ray_medium_destroy(m);
        gm->med.clear();
        gm->g = nullptr;
    }
    // the members' ray_frechet handles are detached by ray_ctx_destroy below (their destroy stays valid); the group
    // handle over them loses its group: ray_group_frechet_destroy stays valid, every other call reports RAY_ERR_INVALID
    for (ray_group_frechet *gf : g->frechets) gf->g = nullptr;
    for (ncclComm_t c : g->comms) (void)ncclCommDestroy(c);
    for (GroupMember &m : g->mem) {
        if (!m.ctx) continue;                           // never built (ray_group_create's failure path): owns nothing
        (void)hipSetDevice(m.device);
        if (m.d_m) (void)hipFree(m.d_m);
        if (m.d_int) (void)hipFree(m.d_int);
        if (m.d_obs) (void)hipFree(m.d_obs);
        if (m.d_mask) (void)hipFree(m.d_mask);
        if (m.d_mis) (void)hipFree(m.d_mis);
        if (m.d_sums) (void)hipFree(m.d_sums);
        if (m.d_gv) (void)hipFree(m.d_gv);
        if (m.d_gmask) (void)hipFree(m.d_gmask);
        if (m.d_gb) (void)hipFree(m.d_gb);
        if (m.done) (void)hipEventDestroy(m.done);
        ray_ctx_destroy(m.ctx);
    }
    (void)hipGetLastError();                            // nothing of the tear-down is left for a later launch check to trip over
    if (g->d_ptrs) { (void)hipSetDevice(g->mem[0].device); (void)hipFree(g->d_ptrs); }
    if (g->d_bptrs) { (void)hipSetDevice(g->mem[0].device); (void)hipFree(g->d_bptrs); }
    delete g;
    return RAY_OK;
}

int ray_group_create(const int *devices, int n, ray_group **out)
{
    if (!out) return fail(RAY_ERR_INVALID, "null out");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return fail(RAY_ERR_INVALID, "ray_group_create: 1..64 devices");
    ray_group *g = new ray_group;
    g->mem.resize((size_t)n);
    bool distinct = true;
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < k; ++j) distinct = distinct && devices[k] != devices[j];
    for (int k = 0; k < n; ++k) {
        GroupMember &m = g->mem[(size_t)k];
        m.device = devices[k];
        int rc = ray_ctx_create(devices[k], &m.ctx);
        if (rc) { const std::string msg = g_err; ray_group_destroy(g); return fail(rc, msg); }
        hipError_t e = hipMalloc((void **)&m.d_sums, 4 * sizeof(double));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m.done, hipEventDisableTiming);
        if (e != hipSuccess) { ray_group_destroy(g); return fail(RAY_ERR_HIP, std::string("ray_group_create: ") + hipGetErrorString(e)); }
    }
    if (distinct) {
        // single-process multi-GPU communicator: one rank per device, collectives issued between
        // ncclGroupStart / ncclGroupEnd from this thread
        g->comms.resize((size_t)n);
        ncclResult_t r = ncclCommInitAll(g->comms.data(), n, devices);
        if (r != ncclSuccess) {
            g->comms.clear();
            ray_group_destroy(g);
            return fail(RAY_ERR_HIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
        }
        g->use_rccl = true;
    } else {
        for (int k = 1; k < n; ++k)
            if (devices[k] != devices[0]) {
                ray_group_destroy(g);
                return fail(RAY_ERR_UNSUPPORTED, "ray_group_create: a device may repeat only if ALL members share it (one-GPU rehearsal)");
            }
        std::vector<const double *> ptrs((size_t)n);
        for (int k = 0; k < n; ++k) ptrs[(size_t)k] = g->mem[(size_t)k].d_sums;
        hipError_t e = hipSetDevice(devices[0]);
        if (e == hipSuccess) e = hipMalloc((void **)&g->d_ptrs, sizeof(double *) * (size_t)n);
        if (e == hipSuccess) e = hipMemcpy(g->d_ptrs, ptrs.data(), sizeof(double *) * (size_t)n, hipMemcpyHostToDevice);
        if (e != hipSuccess) { ray_group_destroy(g); return fail(RAY_ERR_HIP, std::string("ray_group_create: ") + hipGetErrorString(e)); }
    }
    *out = g;
    return RAY_OK;
}

int ray_group_size(const ray_group *g, int *n, int *uses_rccl)
{
    if (!g) return fail(RAY_ERR_INVALID, "null group");
    if (n) *n = (int)g->mem.size();
    if (uses_rccl) *uses_rccl = g->use_rccl ? 1 : 0;
    return RAY_OK;
}

int ray_group_set_arith(ray_group *g, int arith)
{
    if (!g) return fail(RAY_ERR_INVALID, "null group");
    for (GroupMember &m : g->mem) {
        int rc = ray_ctx_set_arith(m.ctx, arith);
        if (rc) return rc;
    }
    return RAY_OK;
}

// ---- media: one replica per member ----------------------------------------------------------------------
static int group_medium_finish(ray_group *g, ray_group_medium *gm, int rc, ray_group_medium **out)
{
    if (rc) {
        const std::string msg = g_err;
        for (ray_medium *m : gm->med) ray_medium_destroy(m);
        delete gm;
        return fail(rc, msg);
    }
    g->media.push_back(gm);
    *out = gm;
    return RAY_OK;
}

int ray_group_medium_create_grid(ray_group *g, const double *c_host, int N, double ext, ray_group_medium **out)
{
    if (!out) return fail(RAY_ERR_INVALID, "null out");
    *out = nullptr;
    if (!g) return fail(RAY_ERR_INVALID, "null group");
    ray_group_medium *gm = new ray_group_medium;
    gm->g = g;
    int rc = RAY_OK;
    for (GroupMember &m : g->mem) {
        ray_medium *med = nullptr;
        if ((rc = ray_medium_create_grid(m.ctx, c_host, N, ext, &med))) break;
        gm->med.push_back(med);
    }
    return group_medium_finish(g, gm, rc, out);
}

int ray_group_medium_create_sinprod(ray_group *g, double A, double wx, double wy, double gx, double gy, ray_group_medium **out)
{
    if (!out) return fail(RAY_ERR_INVALID, "null out");
    *out = nullptr;
    if (!g) return fail(RAY_ERR_INVALID, "null group");
    ray_group_medium *gm = new ray_group_medium;
    gm->g = g;
    int rc = RAY_OK;
    for (GroupMember &m : g->mem) {
        ray_medium *med = nullptr;
        if ((rc = ray_medium_create_sinprod(m.ctx, A, wx, wy, gx, gy, &med))) break;
        gm->med.push_back(med);
    }
    return group_medium_finish(g, gm, rc, out);
}

int ray_group_medium_update_grid(ray_group_medium *gm, const double *c_host)
{
    if (!gm || !gm->g) return fail(RAY_ERR_INVALID, "null / detached group medium");
    for (ray_medium *m : gm->med) {
        int rc = ray_medium_update_grid(m, c_host);
        if (rc) return rc;
    }
    return RAY_OK;
}

int ray_group_medium_destroy(ray_group_medium *gm)
{
    if (!gm) return RAY_OK;
    if (gm->g) {
        forget(gm->g->media, gm);
        for (ray_medium *m : gm->med) ray_medium_destroy(m);
    }
    delete gm;
    return RAY_OK;
}

// ---- ScatterRelation / ScatterRelationObstacle over the whole node -----------------------------------------
int ray_group_shoot_to_exit(ray_group *g, const ray_group_medium *gm, const ray_obstacle *ob, const ray_shoot_params *prm,
                            double *m_host, int64_t ld, int32_t *status, int32_t *nsteps, int32_t *nreflect)
{
    if (!g) return fail(RAY_ERR_INVALID, "null group");
    if (!gm || gm->g != g) return fail(RAY_ERR_INVALID, "group medium belongs to another group");
    int rc = check_params(prm);
    if (rc) return rc;
    const int G = (int)g->mem.size(), nsrc = prm->src_end - prm->src_begin;
    const int64_t num = (int64_t)nsrc * prm->nd;
    if (m_host && ld < num) return fail(RAY_ERR_INVALID, "ld smaller than the number of rows");
    // 1. every member's launch, back to back (asynchronous: the GPUs run concurrently)
    for (int k = 0; k < G; ++k) {
        GroupMember &m = g->mem[(size_t)k];
        m.first = k; m.G = G; m.nd = prm->nd;
        m.row0 = (int64_t)k * prm->nd;
        m.count = k < nsrc ? (nsrc - k + G - 1) / G : 0;
        m.rows = (int64_t)m.count * prm->nd;
        if ((rc = use(m.ctx))) { group_sync_all(g); return rc; }
        const size_t rows1 = (size_t)std::max<int64_t>(m.rows, 1);
        hipError_t he = grow((void **)&m.d_m, &m.m_cap, rows1 * 9 * sizeof(double));
        if (he == hipSuccess) he = grow((void **)&m.d_int, &m.int_cap, rows1 * 3 * sizeof(int32_t));
        if (he != hipSuccess) { group_sync_all(g); return fail(he == hipErrorOutOfMemory ? RAY_ERR_NOMEM : RAY_ERR_HIP, std::string("ray_group_shoot_to_exit: ") + hipGetErrorString(he)); }
        if (m.rows == 0) continue;
        ray_shoot_params sub = *prm;
        sub.src_begin = prm->src_begin + k;
        m.ctx->src_stride = G;
        rc = shoot_to_exit_impl(m.ctx, gm->med[(size_t)k], ob, &sub, m.d_m, m.rows, m.d_int, m.d_int + m.rows, m.d_int + 2 * m.rows);
        if (rc) { group_sync_all(g); return rc; }
    }
    // 2. every member's share into the caller's column-major arrays
    for (int k = 0; k < G; ++k) {
        GroupMember &m = g->mem[(size_t)k];
        if (m.rows == 0) continue;
        hipError_t he = hipSetDevice(m.device);
        for (int c = 0; c < 9 && m_host && he == hipSuccess; ++c)
            he = comb_copy(m, m_host + (size_t)c * (size_t)ld, m.d_m + (size_t)c * (size_t)m.rows, sizeof(double), true);
        if (he == hipSuccess && status) he = comb_copy(m, status, m.d_int, sizeof(int32_t), true);
        if (he == hipSuccess && nsteps) he = comb_copy(m, nsteps, m.d_int + m.rows, sizeof(int32_t), true);
        if (he == hipSuccess && nreflect) he = comb_copy(m, nreflect, m.d_int + 2 * m.rows, sizeof(int32_t), true);
        if (he != hipSuccess) { group_sync_all(g); return fail(RAY_ERR_HIP, std::string("ray_group_shoot_to_exit: ") + hipGetErrorString(he)); }
    }
    // 3. results valid on return
    for (GroupMember &m : g->mem) {
        HIPCHK(hipSetDevice(m.device));
        HIPCHK(hipStreamSynchronize(m.ctx->stream));
    }
    return RAY_OK;
}

// ---- the fixed-time traces of a reconstruction iteration over the whole node ---------------------------------------
// NonReflectionTrace (Obstacle.jl:275-329, mode OVERSHOOT) / the X part of ScatterForwardOperator (NonObstacle.jl:
// 141-187, mode SHORTEN_CEASE), which the reference hands to its workers pid by pid (NonObstacle.jl:236-240).
// The rows of m are dealt like ray_group_shoot_to_exit deals them when they are fans (nd rows per source, given or
// recognised in m): member k of G takes the fans k, k + G, ...; rows in any other order go in G contiguous blocks.
// Every member keeps its share resident (columns 1-4 and 9 of m, the traced state in columns 5-8), so that
// ray_group_set_observation + ray_group_misfit apply to the traced states exactly as they do to a shoot.
int ray_group_trace_fixed_time(ray_group *g, const ray_group_medium *gm, const double *m_host, int64_t num, int64_t ld,
                               int32_t nd, double dt, int32_t mode, double *X_host, int64_t ldx, int32_t *status,
                               int32_t *nsteps)
{
    if (!g) return fail(RAY_ERR_INVALID, "null group");
    if (!gm || gm->g != g) return fail(RAY_ERR_INVALID, "group medium belongs to another group");
    if (!m_host || num < 0 || ld < num || nd < 0 || (X_host && ldx < num)) return fail(RAY_ERR_INVALID, "bad m / X / num / ld / nd");
    const int G = (int)g->mem.size();
    int fan = nd > 0 ? (num % nd == 0 ? nd : 0) : detect_fan(m_host, num, ld);
    const int64_t nfan = fan ? num / fan : 0, block = (num + G - 1) / G;
    int rc;
    for (int k = 0; k < G; ++k) {
        GroupMember &m = g->mem[(size_t)k];
        if (fan) {
            m.first = k; m.G = G; m.nd = fan; m.row0 = (int64_t)k * fan;
            m.count = k < nfan ? (int)((nfan - k + G - 1) / G) : 0;
            m.rows = (int64_t)m.count * fan;
        } else {
            m.row0 = std::min<int64_t>((int64_t)k * block, num);
            m.rows = std::min<int64_t>(block, num - m.row0);
            if (m.rows > INT32_MAX) { group_sync_all(g); return fail(RAY_ERR_INVALID, "ray_group_trace_fixed_time: more than 2^31 rows per member"); }
            m.first = k; m.G = 1; m.nd = (int)m.rows; m.count = m.rows ? 1 : 0;
        }
        if ((rc = use(m.ctx))) { group_sync_all(g); return rc; }
        const size_t rows1 = (size_t)std::max<int64_t>(m.rows, 1);
        hipError_t he = grow((void **)&m.d_m, &m.m_cap, rows1 * 9 * sizeof(double));
        if (he == hipSuccess) he = grow((void **)&m.d_int, &m.int_cap, rows1 * 3 * sizeof(int32_t));
        if (m.rows == 0 || he != hipSuccess) {
            if (he != hipSuccess) { group_sync_all(g); return fail(he == hipErrorOutOfMemory ? RAY_ERR_NOMEM : RAY_ERR_HIP, std::string("ray_group_trace_fixed_time: ") + hipGetErrorString(he)); }
            continue;
        }
        for (int c = 0; c < 9 && he == hipSuccess; ++c)
            if (c < 4 || c == 8)
                he = comb_copy(m, const_cast<double *>(m_host) + (size_t)c * (size_t)ld, m.d_m + (size_t)c * (size_t)m.rows, sizeof(double), false);
        if (he != hipSuccess) { group_sync_all(g); return fail(RAY_ERR_HIP, std::string("ray_group_trace_fixed_time: ") + hipGetErrorString(he)); }
        rc = trace_dev_impl(m.ctx, gm->med[(size_t)k], m.d_m, m.rows, m.rows, dt, mode, m.d_m + 4 * (size_t)m.rows, m.rows,
                            m.d_int, m.d_int + m.rows, nullptr, nullptr, 0, fan);
        if (rc) { group_sync_all(g); return rc; }
    }
    for (int k = 0; k < G; ++k) {
        GroupMember &m = g->mem[(size_t)k];
        if (m.rows == 0) continue;
        hipError_t he = hipSetDevice(m.device);
        for (int c = 0; c < 4 && X_host && he == hipSuccess; ++c)
            he = comb_copy(m, X_host + (size_t)c * (size_t)ldx, m.d_m + (size_t)(4 + c) * (size_t)m.rows, sizeof(double), true);
        if (he == hipSuccess && status) he = comb_copy(m, status, m.d_int, sizeof(int32_t), true);
        if (he == hipSuccess && nsteps) he = comb_copy(m, nsteps, m.d_int + m.rows, sizeof(int32_t), true);
        if (he != hipSuccess) { group_sync_all(g); return fail(RAY_ERR_HIP, std::string("ray_group_trace_fixed_time: ") + hipGetErrorString(he)); }
    }
    for (GroupMember &m : g->mem) {
        HIPCHK(hipSetDevice(m.device));
        HIPCHK(hipStreamSynchronize(m.ctx->stream));
    }
    return RAY_OK;
}

// ---- ScatterForwardOperator over the whole node (NonObstacle.jl:189-243; its workers: :236-240) ------------------------
// Rays are dealt as ray_group_trace_fixed_time deals them.  ray_forward_operator talks to the host in the middle of
// its work (step counts come back for the grouping of rays into waves), so every member runs on a host thread of its
// own; each works from a packed copy of its rows and returns an ordinary ray_frechet over them.
static void group_share_rows(const GroupShare &sh, int64_t r, int64_t *global_first, int64_t *run)
{
    const int64_t p = r / sh.nd;                       // piece (fan) of the share that local ray r starts
    *global_first = sh.row0 + p * (int64_t)sh.G * sh.nd + (r - p * sh.nd);
    *run = sh.nd - (r - p * sh.nd);
}

int ray_group_frechet_destroy(ray_group_frechet *gf)
{
    if (!gf) return RAY_OK;
    int rc = RAY_OK;
    if (gf->g) forget(gf->g->frechets, gf);
    for (ray_frechet *f : gf->f)
        if (f) { const int r = ray_frechet_destroy(f); if (r) rc = r; }
    delete gf;
    return rc;
}

int ray_group_forward_operator(ray_group *g, ray_group_medium *gm, const double *m_host, int64_t num, int64_t ld,
                               int32_t nd, double dt, double droptol, double *X_host, int64_t ldx, int32_t *status,
                               int32_t *nsteps, ray_group_frechet **out)
{
    if (!out) return fail(RAY_ERR_INVALID, "null out");
    *out = nullptr;
    if (!g) return fail(RAY_ERR_INVALID, "null group");
    if (!gm || gm->g != g) return fail(RAY_ERR_INVALID, "group medium belongs to another group");
    if (!m_host || !X_host || num < 0 || ld < num || ldx < num || nd < 0) return fail(RAY_ERR_INVALID, "bad m / X / num / ld / nd");
    const int G = (int)g->mem.size();
    const int fan = nd > 0 ? (num % nd == 0 ? nd : 0) : detect_fan(m_host, num, ld);
    const int64_t nfan = fan ? num / fan : 0, block = (num + G - 1) / G;
    ray_group_frechet *gf = new ray_group_frechet;
    gf->g = g; gf->num = num; gf->N = gm->med.empty() ? 0 : gm->med[0]->N;
    gf->f.assign((size_t)G, nullptr);
    gf->share.resize((size_t)G);
    for (int k = 0; k < G; ++k) {
        GroupShare &sh = gf->share[(size_t)k];
        if (fan) {
            sh.G = G; sh.nd = fan; sh.row0 = (int64_t)k * fan;
            sh.count = k < nfan ? (int)((nfan - k + G - 1) / G) : 0;
            sh.rows = (int64_t)sh.count * fan;
        } else {
            sh.row0 = std::min<int64_t>((int64_t)k * block, num);
            sh.rows = std::min<int64_t>(block, num - sh.row0);
            sh.G = 1; sh.nd = (int)std::max<int64_t>(sh.rows, 1); sh.count = sh.rows ? 1 : 0;
        }
    }
    std::vector<int> rcs((size_t)G, RAY_OK);
    std::vector<std::string> errs((size_t)G);
    auto work = [&](int k) {
        const GroupShare &sh = gf->share[(size_t)k];
        const int64_t n = sh.rows, ldk = std::max<int64_t>(n, 1);
        // not value-initialised: columns 5-8 of the packed rows are never read, X / status / steps are written by the call
        std::unique_ptr<double[]> mk(new double[(size_t)ldk * 9]), Xk(new double[(size_t)ldk * 4]);
        std::unique_ptr<int32_t[]> stk(new int32_t[(size_t)ldk]), nsk(new int32_t[(size_t)ldk]);
        for (int p = 0; p < sh.count; ++p) {
            const int64_t src = sh.row0 + (int64_t)p * sh.G * sh.nd, dst = (int64_t)p * sh.nd;
            for (int c = 0; c < 9; ++c)
                if (c < 4 || c == 8) std::memcpy(&mk[(size_t)(dst + c * ldk)], m_host + src + (int64_t)c * ld, sizeof(double) * (size_t)sh.nd);
        }
        rcs[(size_t)k] = ray_forward_operator(g->mem[(size_t)k].ctx, gm->med[(size_t)k], mk.get(), n, ldk, dt, droptol,
                                               Xk.get(), ldk, nullptr, stk.get(), nsk.get(), &gf->f[(size_t)k]);
        if (rcs[(size_t)k]) { errs[(size_t)k] = ray_last_error(); return; }
        for (int p = 0; p < sh.count; ++p) {
            const int64_t dst = sh.row0 + (int64_t)p * sh.G * sh.nd, src = (int64_t)p * sh.nd;
            for (int c = 0; c < 4; ++c) std::memcpy(X_host + dst + (int64_t)c * ldx, &Xk[(size_t)(src + c * ldk)], sizeof(double) * (size_t)sh.nd);
            if (status) std::memcpy(status + dst, &stk[(size_t)src], sizeof(int32_t) * (size_t)sh.nd);
            if (nsteps) std::memcpy(nsteps + dst, &nsk[(size_t)src], sizeof(int32_t) * (size_t)sh.nd);
        }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < G; ++k) th.emplace_back(work, k);
    work(0);
    for (std::thread &t : th) t.join();
    for (int k = 0; k < G; ++k)
        if (rcs[(size_t)k]) {
            const int rc = rcs[(size_t)k];
            const std::string msg = "ray_group_forward_operator (member " + std::to_string(k) + "): " + errs[(size_t)k];
            ray_group_frechet_destroy(gf);
            return fail(rc, msg);
        }
    g->frechets.push_back(gf);
    *out = gf;
    return RAY_OK;
}

int ray_group_frechet_shape(ray_group_frechet *gf, int64_t *n_rows, int64_t *n_cols, int64_t *nnz)
{
    if (!gf) return fail(RAY_ERR_INVALID, "null handle");
    if (n_rows) *n_rows = 4 * gf->num;
    if (n_cols) *n_cols = (int64_t)gf->N * gf->N;
    if (nnz) {
        *nnz = 0;
        for (ray_frechet *f : gf->f) *nnz += f ? f->nnz : 0;
    }
    return RAY_OK;
}

// member k's part: an ordinary handle (every ray_frechet_* pass applies to it) over `count` runs of nd rays, run i being
// the caller's rays row0 + i * stride .. + nd
int ray_group_frechet_member(ray_group_frechet *gf, int k, ray_frechet **f, int64_t *row0, int32_t *count, int32_t *nd,
                             int64_t *stride)
{
    if (!gf || k < 0 || k >= (int)gf->f.size()) return fail(RAY_ERR_INVALID, "bad handle / member");
    const GroupShare &sh = gf->share[(size_t)k];
    if (f) *f = gf->f[(size_t)k];
    if (row0) *row0 = sh.row0;
    if (count) *count = sh.count;
    if (nd) *nd = sh.nd;
    if (stride) *stride = (int64_t)sh.G * sh.nd;
    return RAY_OK;
}

// the whole M = droptol!(sparse(M), 1e-12) (NonObstacle.jl:242) on the host, rows in the caller's order: what the
// reference's master holds after its workers return
int ray_group_frechet_export_csr(ray_group_frechet *gf, int64_t *indptr, int32_t *indices, double *data)
{
    if (!gf || !indptr) return fail(RAY_ERR_INVALID, "null argument");
    int64_t total = 0;
    for (ray_frechet *f : gf->f) total += f ? f->nnz : 0;
    if (total && (!indices || !data)) return fail(RAY_ERR_INVALID, "null indices / data");
    const size_t G = gf->f.size();
    std::vector<std::vector<int64_t>> ip(G);
    std::vector<std::vector<int32_t>> ix(G);
    std::vector<std::vector<double>> dv(G);
    indptr[0] = 0;
    for (size_t k = 0; k < G; ++k) {                    // row lengths into indptr[row + 1]
        ray_frechet *f = gf->f[k];
        const GroupShare &sh = gf->share[k];
        if (!f || sh.rows == 0) continue;
        ip[k].resize((size_t)(4 * sh.rows + 1));
        ix[k].resize((size_t)std::max<int64_t>(f->nnz, 1));
        dv[k].resize((size_t)std::max<int64_t>(f->nnz, 1));
        const int rc = ray_frechet_export_csr(f, ip[k].data(), ix[k].data(), dv[k].data());
        if (rc) return rc;
        for (int64_t r = 0; r < sh.rows; r += sh.nd) {
            int64_t g0, run;
            group_share_rows(sh, r, &g0, &run);
            for (int64_t q = 0; q < 4 * (int64_t)sh.nd; ++q) indptr[4 * g0 + q + 1] = ip[k][(size_t)(4 * r + q + 1)] - ip[k][(size_t)(4 * r + q)];
        }
    }
    for (int64_t q = 0; q < 4 * gf->num; ++q) indptr[q + 1] += indptr[q];
    for (size_t k = 0; k < G; ++k) {
        const GroupShare &sh = gf->share[k];
        if (ip[k].empty()) continue;
        for (int64_t r = 0; r < sh.rows; r += sh.nd) {
            int64_t g0, run;
            group_share_rows(sh, r, &g0, &run);
            const int64_t a = ip[k][(size_t)(4 * r)], b = ip[k][(size_t)(4 * (r + sh.nd))];
            if (b > a) {
                std::memcpy(indices + indptr[4 * g0], &ix[k][(size_t)a], sizeof(int32_t) * (size_t)(b - a));
                std::memcpy(data + indptr[4 * g0], &dv[k][(size_t)a], sizeof(double) * (size_t)(b - a));
            }
        }
    }
    return RAY_OK;
}

// ---- the passes of the loop that end in a sum over all rays: ONE all-reduce of N^2 doubles -------------------------------
// b = M[order,:]' * v[order] (the gradient of NonObstacle.jl:354, `normal` = 0) and w = M[order,:]' * (M[order,:] * x)
// (the action of :353's MtM, `normal` = 1): every member runs the pass over its rays into its own N^2 vector, the
// vectors are summed across the members (ncclAllReduce over RCCL; members on one device: a fold kernel, member order),
// and the caller gets the sum.  in: v (4 num, ray-major, caller's row order) or x (N^2); ray_mask: num or NULL.
static int group_frechet_reduce(ray_group_frechet *gf, const double *in, const uint8_t *ray_mask, double *out_host, int normal)
{
    if (!gf || !in || !out_host) return fail(RAY_ERR_INVALID, "null argument");
    ray_group *g = gf->g;
    if (!g) return fail(RAY_ERR_INVALID, "the group of this ray_group_frechet was destroyed: only ray_group_frechet_destroy is valid on it");
    const int G = (int)g->mem.size();
    const size_t nn = (size_t)gf->N * (size_t)gf->N;
    int rc;
    std::vector<double> vk;
    std::vector<uint8_t> mk;
    for (int k = 0; k < G; ++k) {
        GroupMember &m = g->mem[(size_t)k];
        const GroupShare &sh = gf->share[(size_t)k];
        if ((rc = use(m.ctx))) { group_sync_all(g); return rc; }
        const size_t rows1 = (size_t)std::max<int64_t>(sh.rows, 1);
        hipError_t he = grow((void **)&m.d_gv, &m.gv_cap, sizeof(double) * (normal ? nn : 4 * rows1));
        if (he == hipSuccess) he = grow((void **)&m.d_gmask, &m.gmask_cap, rows1);
        if (he == hipSuccess) he = grow((void **)&m.d_gb, &m.gb_cap, sizeof(double) * nn);
        if (he == hipSuccess && normal) he = hipMemcpyAsync(m.d_gv, in, sizeof(double) * nn, hipMemcpyHostToDevice, m.ctx->stream);
        if (he == hipSuccess && (!normal || ray_mask) && sh.rows) {
            // the member's rays out of the caller's order (pageable staging: synchronous before the vectors are reused)
            if (!normal) vk.resize(4 * (size_t)sh.rows);
            if (ray_mask) mk.resize((size_t)sh.rows);
            for (int p = 0; p < sh.count; ++p) {
                const int64_t src = sh.row0 + (int64_t)p * sh.G * sh.nd, dst = (int64_t)p * sh.nd;
                if (!normal) std::memcpy(&vk[4 * (size_t)dst], in + 4 * src, sizeof(double) * 4 * (size_t)sh.nd);
                if (ray_mask) std::memcpy(&mk[(size_t)dst], ray_mask + src, (size_t)sh.nd);
            }
            if (!normal) he = hipMemcpy(m.d_gv, vk.data(), sizeof(double) * 4 * (size_t)sh.rows, hipMemcpyHostToDevice);
            if (he == hipSuccess && ray_mask) he = hipMemcpy(m.d_gmask, mk.data(), (size_t)sh.rows, hipMemcpyHostToDevice);
        }
        if (he != hipSuccess) { group_sync_all(g); return fail(he == hipErrorOutOfMemory ? RAY_ERR_NOMEM : RAY_ERR_HIP, std::string("ray_group_frechet pass: ") + hipGetErrorString(he)); }
        ray_frechet *f = gf->f[(size_t)k];
        rc = normal ? ray_frechet_normal_matvec_dev(f, m.d_gv, ray_mask ? m.d_gmask : nullptr, m.d_gb)
                    : ray_frechet_matvec_t_dev(f, m.d_gv, ray_mask ? m.d_gmask : nullptr, m.d_gb);
        if (rc) { group_sync_all(g); return rc; }
    }
    GroupMember &m0 = g->mem[0];
    if (g->use_rccl) {
        NCCLCHK(ncclGroupStart());
        for (int k = 0; k < G; ++k) {
            GroupMember &m = g->mem[(size_t)k];
            ncclResult_t r = ncclAllReduce(m.d_gb, m.d_gb, nn, ncclDouble, ncclSum, g->comms[(size_t)k], m.ctx->stream);
            if (r != ncclSuccess) { (void)ncclGroupEnd(); group_sync_all(g); return fail(RAY_ERR_HIP, std::string("ncclAllReduce: ") + ncclGetErrorString(r)); }
        }
        NCCLCHK(ncclGroupEnd());
    } else if (G > 1) {
        HIPCHK(hipSetDevice(m0.device));
        std::vector<const double *> ptrs((size_t)G);
        for (int k = 0; k < G; ++k) ptrs[(size_t)k] = g->mem[(size_t)k].d_gb;
        if (!g->d_bptrs) HIPCHK(hipMalloc((void **)&g->d_bptrs, sizeof(double *) * (size_t)G));
        HIPCHK(hipMemcpy(g->d_bptrs, ptrs.data(), sizeof(double *) * (size_t)G, hipMemcpyHostToDevice));
        for (int k = 1; k < G; ++k) {
            HIPCHK(hipEventRecord(g->mem[(size_t)k].done, g->mem[(size_t)k].ctx->stream));
            HIPCHK(hipStreamWaitEvent(m0.ctx->stream, g->mem[(size_t)k].done, 0));
        }
        // member 0's own vector is both a part and the result: each element is read (all parts) before it is written
        hipLaunchKernelGGL(k_group_fold_vec, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, m0.ctx->stream, g->d_bptrs, G, (int64_t)nn, m0.d_gb);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipSetDevice(m0.device));
    HIPCHK(hipMemcpyAsync(out_host, m0.d_gb, sizeof(double) * nn, hipMemcpyDeviceToHost, m0.ctx->stream));
    for (GroupMember &m : g->mem) {
        HIPCHK(hipSetDevice(m.device));
        HIPCHK(hipStreamSynchronize(m.ctx->stream));
    }
    return RAY_OK;
}

int ray_group_frechet_matvec_t(ray_group_frechet *gf, const double *v, const uint8_t *ray_mask, double *b)
{
    return group_frechet_reduce(gf, v, ray_mask, b, 0);
}

int ray_group_frechet_normal_matvec(ray_group_frechet *gf, const double *x, const uint8_t *ray_mask, double *w)
{
    return group_frechet_reduce(gf, x, ray_mask, w, 1);
}

// the observation the misfit is taken against (NonObstacle.jl:336 `observation`), sharded like the last shoot
// and kept resident: the rows of every column go to the member that owns their source
int ray_group_set_observation(ray_group *g, const double *obs_host, int64_t num, int64_t ld)
{
    if (!g || !obs_host) return fail(RAY_ERR_INVALID, "null group / observation");
    int64_t total = 0;
    for (GroupMember &m : g->mem) total += m.rows;
    if (num != total || ld < num) return fail(RAY_ERR_INVALID, "ray_group_set_observation: num must equal the rows of the last ray_group_shoot_to_exit / ray_group_trace_fixed_time");
    for (GroupMember &m : g->mem) {
        int rc = use(m.ctx);
        if (rc) { group_sync_all(g); return rc; }
        const size_t rows1 = (size_t)std::max<int64_t>(m.rows, 1);
        hipError_t he = grow((void **)&m.d_obs, &m.obs_cap, rows1 * 9 * sizeof(double));
        if (he != hipSuccess) { group_sync_all(g); return fail(RAY_ERR_HIP, std::string("ray_group_set_observation: ") + hipGetErrorString(he)); }
        m.obs_rows = m.rows;
        for (int c = 4; c < 8 && he == hipSuccess; ++c)
            he = comb_copy(m, const_cast<double *>(obs_host) + (size_t)c * (size_t)ld, m.d_obs + (size_t)c * (size_t)m.rows, sizeof(double), false);
        if (he != hipSuccess) { group_sync_all(g); return fail(RAY_ERR_HIP, std::string("ray_group_set_observation: ") + hipGetErrorString(he)); }
    }
    for (GroupMember &m : g->mem) {
        HIPCHK(hipSetDevice(m.device));
        HIPCHK(hipStreamSynchronize(m.ctx->stream));
    }
    return RAY_OK;
}

// misfit of the resident slabs against the resident observation: per-member reduction kernels, then ONE
// all-reduce of the two sums (NonObstacle.jl:336,372; ObstacleDemo.jl:219-226 for the mask)
int ray_group_misfit(ray_group *g, double out[2], uint8_t *mask, double perc_thresh, double *mismatch)
{
    if (!g || !out) return fail(RAY_ERR_INVALID, "null group / out");
    const int G = (int)g->mem.size();
    int rc;
    for (GroupMember &m : g->mem) {
        if (m.obs_rows != m.rows) return fail(RAY_ERR_INVALID, "ray_group_misfit: call ray_group_set_observation after the shoot it belongs to");
        if ((rc = use(m.ctx))) { group_sync_all(g); return rc; }
        const size_t rows1 = (size_t)std::max<int64_t>(m.rows, 1);
        hipError_t he = hipSuccess;
        if (mask) he = grow((void **)&m.d_mask, &m.mask_cap, rows1);
        if (he == hipSuccess && mismatch) he = grow((void **)&m.d_mis, &m.mis_cap, rows1 * 4 * sizeof(double));
        if (he != hipSuccess) { group_sync_all(g); return fail(RAY_ERR_HIP, std::string("ray_group_misfit: ") + hipGetErrorString(he)); }
        rc = ray_misfit_dev(m.ctx, m.d_m, m.d_obs, m.rows, std::max<int64_t>(m.rows, 1), m.d_sums, mask ? m.d_mask : nullptr,
                            perc_thresh, mismatch ? m.d_mis : nullptr);
        if (rc) { group_sync_all(g); return rc; }
    }
    if (g->use_rccl) {
        NCCLCHK(ncclGroupStart());
        for (int k = 0; k < G; ++k) {
            GroupMember &m = g->mem[(size_t)k];
            ncclResult_t r = ncclAllReduce(m.d_sums, m.d_sums + 2, 2, ncclDouble, ncclSum, g->comms[(size_t)k], m.ctx->stream);
            if (r != ncclSuccess) { (void)ncclGroupEnd(); group_sync_all(g); return fail(RAY_ERR_HIP, std::string("ncclAllReduce: ") + ncclGetErrorString(r)); }
        }
        NCCLCHK(ncclGroupEnd());
    } else {
        GroupMember &m0 = g->mem[0];
        HIPCHK(hipSetDevice(m0.device));
        for (int k = 1; k < G; ++k) {
            HIPCHK(hipEventRecord(g->mem[(size_t)k].done, g->mem[(size_t)k].ctx->stream));
            HIPCHK(hipStreamWaitEvent(m0.ctx->stream, g->mem[(size_t)k].done, 0));
        }
        hipLaunchKernelGGL(k_group_fold, dim3(1), dim3(64), 0, m0.ctx->stream, g->d_ptrs, G, m0.d_sums + 2);
        HIPCHK(hipGetLastError());
    }
    {
        GroupMember &m0 = g->mem[0];
        HIPCHK(hipSetDevice(m0.device));
        HIPCHK(hipMemcpyAsync(out, m0.d_sums + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, m0.ctx->stream));
    }
    for (GroupMember &m : g->mem) {
        if (m.rows == 0) continue;
        HIPCHK(hipSetDevice(m.device));
        if (mask) HIPCHK(comb_copy(m, mask, m.d_mask, 1, true));
        if (mismatch) HIPCHK(comb_copy(m, mismatch, m.d_mis, 4 * sizeof(double), true));
    }
    for (GroupMember &m : g->mem) {
        HIPCHK(hipSetDevice(m.device));
        HIPCHK(hipStreamSynchronize(m.ctx->stream));
    }
    return RAY_OK;
}

int ray_group_get_counters(ray_group *g, ray_counters *out)
{
    if (!g || !out) return fail(RAY_ERR_INVALID, "null group / out");
    ray_counters acc{};
    for (GroupMember &m : g->mem) {
        ray_counters c{};
        int rc = ray_get_counters(m.ctx, &c);
        if (rc) return rc;
        acc.rays += c.rays; acc.ray_steps += c.ray_steps; acc.launches += c.launches;
        acc.kernel_ms = std::max(acc.kernel_ms, c.kernel_ms);     // the members run concurrently
    }
    *out = acc;
    return RAY_OK;
}

} // extern "C"
