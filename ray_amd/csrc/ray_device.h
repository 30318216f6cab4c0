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
// One bit pattern differs from IEEE division: a zero dividend of negative sign gives +0.0 (the residual
// fma(-b, -0, -0) is +0); equal in value, and a zero momentum component stays zero either way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ray_trig.h"

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
    const double *recf;  // the same records in CELL units for the kFused flavour: coefficient k of each
                         // bilinear quadruple times [1, h, h, h*h] (one rounding each), so that the in-cell
                         // offsets enter as u = dx/h, v = dy/h in [0,1) -- see cell_of<kFused>
    const double *hrec;  // (N-1)*(N-1)*12 hess records (Frechet path only; may be null)
    const double *p;     // N nodes, linspace(-ext, ext, N)
    int N;
    double p1;           // p[1] = -ext
    double h;            // p[2] - p[1]            (NonObstacle.jl:32)
    double rh;           // RN(1/h), for the Markstein quotient
    double nb;           // RN(-p1*rh): kFused cell index = floor(fma(x, rh, nb))
};
struct SinProdDev { double A, wx, wy, gx, gy; };
constexpr int kRadialMax = 4;
struct RadialDev { int K; int pad_; double a[kRadialMax], x[kRadialMax], y[kRadialMax]; };

enum MediumKind { kSinProd = 0, kGrid = 1, kRadial = 2 };
enum ObstacleKind { kObNone = 0, kObCircle = 1, kObTrefoil = 2 };

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

// ---- per-thread access context for the gridded medium -----------------------------------------
// LDS carve-up per workgroup (dynamic shared memory, see grid_lds_bytes()):
//   [ p[0..N) node coordinates, padded to 512 B | per wave: 128 cell tags | per wave: 64 x 96-byte
//     record slots | per wave: 64 cell ids ]
// The node table leaves the vector-memory path entirely; the record slots are the landing zone of
// the wave-cooperative record fetch below, and the tags make them a small cell cache: a look-up at
// a nearby position (the next RK4 stage moves a ray by ~1e-4 of a cell) finds its record there
// without a new fetch.
constexpr int kRunSlots = 64;                       // record slots per wave: one per lane, never overflows
#ifndef RAY_SLOT_PIECES
#define RAY_SLOT_PIECES 6                           // 16-byte pieces per record slot: 6 = packed 96-byte slots; 7 = 112-byte
#endif                                              // slots (the seventh piece is never loaded): 16 bank positions instead of 8
constexpr int kSlotPieces = RAY_SLOT_PIECES;
constexpr unsigned kSlotBytes = 16u * kSlotPieces;
constexpr int kPasses = kSlotPieces;                // global_load_lds passes that fill them (P x 1 KiB = 64 slots x 16 P bytes)
constexpr int kWaveSlotBytes = kPasses * 1024;
#ifndef RAY_FETCH_PRIO
#define RAY_FETCH_PRIO 3                            // wave priority from the start of a fetching right-hand side to its DMA issue (0: off)
#endif
#ifndef RAY_TAG_ENTRIES
#define RAY_TAG_ENTRIES 128
#endif
constexpr int kTagEntries = RAY_TAG_ENTRIES;        // per wave, 4 bytes each: (cell id << 6) | slot, or kTagEmpty
constexpr unsigned kTagEmpty = 0xffffffffu;
static_assert(kTagEntries == 128 || kTagEntries == 64, "tags are cleared by one store per lane");
constexpr int kWaveLdsBytes = kWaveSlotBytes + 64 * 4 + kTagEntries * 4;

struct GridAcc {
    unsigned wrec;           // LDS byte offset of this wave's record slots
    unsigned wrec_v;         // the same in a VGPR (addend of the per-lane slot address mad)
    double nb_v;             // g.nb in VGPRs: fma(x, rh, nb) may read one SGPR operand only
    unsigned wcells;         // LDS byte offset of this wave's cell-id slots
    unsigned wtags;          // LDS byte offset of this wave's cell tags (512-byte aligned)
    unsigned lane;
    unsigned role_run4[3];   // loader role slot s (shift 0,4,2): 4 * ((lane + shift) / 6) ...
    unsigned role_poff[3];   // ... and 16 * ((lane + shift) % 6); see fetch_records()
};

// What a lane remembers of its last fetch_records(): the cell and where its record sits in the slots.
// The look-up that follows a fetch (RK4 stage 3 after 2, the next stage 1 after 4) is for the SAME cell
// in most lanes, and when that holds for the whole wave it reads the slot again without the tags.
struct RecState {
    int cid = -1;
    unsigned addr = 0;
};

// kFused keeps no node table (its in-cell offsets come from the fractional part of the cell coordinate)
__host__ __device__ inline size_t grid_lds_bytes(int N, int block, bool nodes = true)
{
    return (nodes ? (((size_t)N * 8 + 511) & ~(size_t)511) : 0) + (size_t)(block / 64) * kWaveLdsBytes;
}

typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

extern __shared__ __attribute__((aligned(16))) char ray_smem[];   // [ p | record slots | cell ids ]

__device__ __forceinline__ const double *lds_nodes() { return reinterpret_cast<const double *>(ray_smem); }

// LDS accesses by ABSOLUTE byte address (the dynamic segment's own base folded into the wave offsets once,
// at set-up): `ray_smem + offset` costs a v_add_u32 of the base -- a link-time 0 -- on every access.
typedef double lds_d2v __attribute__((ext_vector_type(2)));
typedef unsigned lds_u2v __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
#define RAY_LDS_PTR(T, addr) ((__attribute__((address_space(3))) T *)(unsigned)(addr))      // 32-bit LDS pointers
#else
#define RAY_LDS_PTR(T, addr) ((__attribute__((address_space(3))) T *)(uintptr_t)(addr))     // host pass: parsed only
#endif
__device__ __forceinline__ double2 lds_ld_d2(unsigned addr)
{
    const lds_d2v v = *RAY_LDS_PTR(const lds_d2v, addr);
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ unsigned lds_ld_u32(unsigned addr) { return *RAY_LDS_PTR(const unsigned, addr); }
__device__ __forceinline__ void lds_st_u32(unsigned addr, unsigned v) { *RAY_LDS_PTR(unsigned, addr) = v; }
__device__ __forceinline__ void lds_st_u2(unsigned addr, unsigned x, unsigned y)
{
    lds_u2v v;
    v.x = x; v.y = y;
    *RAY_LDS_PTR(lds_u2v, addr) = v;
}

template <bool NODES>
__device__ __forceinline__ GridAcc grid_acc_setup(const GridDev &g)
{
    GridAcc a;
    if (NODES) {
        double *p_lds = reinterpret_cast<double *>(ray_smem);
        for (int k = threadIdx.x; k < g.N; k += blockDim.x) p_lds[k] = g.p[k];
    }
    const unsigned base = (unsigned)(uintptr_t)(lptr_t)ray_smem;
    const unsigned poff = base + (NODES ? (((unsigned)g.N * 8u + 511u) & ~511u) : 0u);
    // wave-uniform by construction; readfirstlane tells the compiler so (LDS bases live in SGPRs)
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    a.wtags = poff + wave * (kTagEntries * 4);
    a.wrec = poff + nw * (kTagEntries * 4) + wave * kWaveSlotBytes;
    a.wcells = poff + nw * (kTagEntries * 4 + kWaveSlotBytes) + wave * (64 * 4);
    a.lane = threadIdx.x & 63;
    a.wrec_v = a.wrec;
    a.nb_v = g.nb;
    asm volatile("" : "+v"(a.wrec_v), "+v"(a.nb_v));       // opaque: not re-materialised from SGPRs per use
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        const unsigned v = a.lane + (kSlotPieces == 6 ? (sl == 0 ? 0u : sl == 1 ? 4u : 2u) : (unsigned)sl);
        const unsigned run = kSlotPieces == 6 ? (v * 171u) >> 10 : (v * 147u) >> 10;              // v / 6, v / 7
        a.role_run4[sl] = run * 4u;
        a.role_poff[sl] = (v - run * (unsigned)kSlotPieces) * 16u;
    }
    static_assert(kSlotPieces == 6 || kSlotPieces == 7, "slot pieces");
    // the cell-id slots are read speculatively by the loader lanes: keep them valid cell ids
    lds_st_u32(a.wcells + a.lane * 4u, 0u);
    if (kTagEntries == 128) lds_st_u2(a.wtags + a.lane * 8u, kTagEmpty, kTagEmpty);
    else lds_st_u32(a.wtags + a.lane * 4u, kTagEmpty);
    __syncthreads();
    return a;
}

// Cell index I-1 for I = Int64(floor((x - p[1])/h)) + 1 (NonObstacle.jl:33-34), clamped to the
// cells whose grad entry exists (reference: undefined behaviour outside; here flagged): cell_of().
// v_cvt_i32_f64 saturates and maps NaN to 0; a NaN position is flagged by the callers' own
// per-step check, so the flag word equals the CPU restatement's in every case.
// Wave-cooperative fetch of the 96-byte cell records (ALL 64 lanes must call this together).
// Lanes are consecutive directions of one source, so along the wavefront the cell index changes
// only every few lanes: a lane is a "leader" when its cell differs from its left neighbour's.
// Each run of equal cells is fetched ONCE, as 16-byte pieces by global_load_lds (in pass t lane l
// loads piece (64t+l)%6 of run (64t+l)/6 straight into LDS), then every lane reads its run's record
// back with 6 ds_read_b128.  Per right-hand side this moves ~17 x 96 B through the L1 instead of
// 64 x 96 B; the L1 tag rate, not HBM or the VALU, bounds the per-lane gather form (profiles/).
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c)   // v_mad_u32_u24
{
    return __umul24(a, b) + c;
}

// Loader roles.  In pass t lane l handles piece q = 64t + l: run q/6, 16-byte piece q%6.  Because
// 64 = 6*10 + 4, q = 6*base_t + sh_t + l with sh_t = 4t mod 6 = 0,4,2,0,4,2 and base_t = 0,10,21,32,
// 42,53: only three role patterns exist (slot t%3), the pass enters as a constant run offset.
__device__ __forceinline__ constexpr unsigned pass_shift(int t) { return (64u * t) % (unsigned)kSlotPieces; }
__device__ __forceinline__ constexpr unsigned pass_base(int t) { return (64u * t - (64u * t) % (unsigned)kSlotPieces) / (unsigned)kSlotPieces; }

// role of this lane in pass T: run (times 4) and byte offset of its piece; 6-piece slots keep the three patterns in
// registers, 7-piece slots (seven patterns) compute them (v / 7 = v * 147 >> 10 for v < 77)
template <int T>
__device__ __forceinline__ void pass_role(const GridAcc &a, unsigned &run4, unsigned &poff)
{
    if (kSlotPieces == 6) { run4 = a.role_run4[T % 3]; poff = a.role_poff[T % 3]; return; }
    if (T < 2) { run4 = a.role_run4[T]; poff = a.role_poff[T]; return; }           // the common two passes: kept
    unsigned l = a.lane;
    asm volatile("" : "+v"(l));                      // computed where it is used (rare passes), not hoisted into live registers
    const unsigned v = l + pass_shift(T), run = (v * 147u) >> 10;
    run4 = run * 4u;
    poff = (v - run * 7u) * 16u;
}

template <int T>
__device__ __forceinline__ unsigned read_cell_of_pass(const GridAcc &a)
{
    unsigned run4, poff;
    pass_role<T>(a, run4, poff);
    return lds_ld_u32(a.wcells + run4 + 4u * pass_base(T));
}
template <int T>
__device__ __forceinline__ void dma_pass(const char *rec, const GridAcc &a, unsigned c, unsigned npiece)
{
    unsigned run4, poff;
    pass_role<T>(a, run4, poff);
    if (a.lane + 64u * T < npiece && (kSlotPieces == 6 || poff < 96u))
        __builtin_amdgcn_global_load_lds((gptr_t)(rec + mad24(c, 96u, poff)),
                                         (lptr_t)RAY_LDS_PTR(char, a.wrec + T * 1024u), 16, 0, 0);
}

#ifdef RAY_STATS
__device__ unsigned long long g_ray_stats[12];   // [4] fetches, [5..9] fetches with > 21/32/42/48/53 runs, [10] sum of runs   // [0] wave look-ups, [1] with >= 1 miss, [2] missing lanes, [3] wide-fan fetches, [11] look-ups served without the tags
#endif

// Tag slot of cell (i, j): (i + 37 j) mod 128 -- distinct for the cells of one wavefront (a compact arc
// of at most a few tens of cells); a collision only costs the loser a miss.
__device__ __forceinline__ unsigned tag_addr(const GridAcc &a, int i, int j)
{
    unsigned t;                                   // spelled out: the compiler narrows __umul24 by a constant
    asm("v_mad_u32_u24 %0, %1, 37, %2" : "=v"(t) : "v"(j), "v"(i));   // to a quarter-rate v_mul_lo_u32
    unsigned addr;                                // and + lshl_add; left alone the compiler shifts first: 4 ops
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(addr) : "v"(t & (unsigned)(kTagEntries - 1)), "s"(a.wtags));
    return addr;
}

__device__ __forceinline__ void fetch_records(const char *rec, const GridAcc &a, int i, int j, int cid, double2 out[6], RecState &rs)
{
#ifdef RAY_ABLATE_NOFETCH
    asm volatile("" :: "v"(cid));
    out[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int k = 1; k < 6; ++k) out[k] = make_double2(0.0, 0.0);
    asm volatile("" : "+v"(out[0].x), "+v"(out[0].y), "+v"(out[1].x), "+v"(out[1].y), "+v"(out[2].x), "+v"(out[2].y));
    asm volatile("" : "+v"(out[3].x), "+v"(out[3].y), "+v"(out[4].x), "+v"(out[4].y), "+v"(out[5].x), "+v"(out[5].y));
    return;
#endif
    // left neighbour's cell (wave_shr:1; lane 0 keeps ~cid, i.e. is always a leader)
    const int left = __builtin_amdgcn_update_dpp(~cid, cid, 0x138, 0xf, 0xf, false);
    const bool leader = left != cid;
    const unsigned long long mask = __ballot(leader);
    const unsigned mlo = (unsigned)mask, mhi = (unsigned)(mask >> 32);
    const unsigned npiece = (unsigned)kSlotPieces * (unsigned)(__builtin_popcount(mlo) + __builtin_popcount(mhi));   // uniform
    // run index of this lane = leaders at or below it, minus one = set bits of (mask >> 1) below it
    // (bit 0 is always set)
    const unsigned long long m1 = mask >> 1;
    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u));
#ifdef RAY_STATS
    if (a.lane == 0) {
        const unsigned nr = npiece / 6u;
        atomicAdd(&g_ray_stats[4], 1ull);
        atomicAdd(&g_ray_stats[10], (unsigned long long)nr);
        if (nr > 21) atomicAdd(&g_ray_stats[5], 1ull);
        if (nr > 32) atomicAdd(&g_ray_stats[6], 1ull);
        if (nr > 42) atomicAdd(&g_ray_stats[7], 1ull);
        if (nr > 48) atomicAdd(&g_ray_stats[8], 1ull);
        if (nr > 53) atomicAdd(&g_ray_stats[9], 1ull);
    }
#endif
    // the slots are about to be overwritten: forget what the tags said about them
    if (kTagEntries == 128) lds_st_u2(a.wtags + a.lane * 8u, kTagEmpty, kTagEmpty);
    else lds_st_u32(a.wtags + a.lane * 4u, kTagEmpty);
    if (leader) {
        lds_st_u32(a.wcells + rank * 4u, (unsigned)cid);
        lds_st_u32(tag_addr(a, i, j), ((unsigned)cid << 6) | rank);
    }
    // passes 0 and 1 (<= 21 runs, the common case): both cell-id reads first, then both DMA loads
    unsigned c0 = read_cell_of_pass<0>(a), c1 = read_cell_of_pass<1>(a);
    asm volatile("" : "+v"(c0), "+v"(c1));                                   // keep the reads ahead of the masks
    dma_pass<0>(rec, a, c0, npiece);
    dma_pass<1>(rec, a, c1, npiece);
    if (npiece > 128u) {                                                     // wide fans: up to 4 more passes
        unsigned c2 = read_cell_of_pass<2>(a), c3 = read_cell_of_pass<3>(a);
        asm volatile("" : "+v"(c2), "+v"(c3));
        dma_pass<2>(rec, a, c2, npiece);
        dma_pass<3>(rec, a, c3, npiece);
        if (npiece > 256u) {
#ifdef RAY_STATS
            if (a.lane == 0) atomicAdd(&g_ray_stats[3], 1ull);
#endif
            unsigned c4 = read_cell_of_pass<4>(a), c5 = read_cell_of_pass<5>(a);
            asm volatile("" : "+v"(c4), "+v"(c5));
            dma_pass<4>(rec, a, c4, npiece);
            dma_pass<5>(rec, a, c5, npiece);
            if (kSlotPieces == 7 && npiece > 384u) {
                unsigned c6 = read_cell_of_pass<6>(a);
                asm volatile("" : "+v"(c6));
                dma_pass<6>(rec, a, c6, npiece);
            }
        }
    }
    if (RAY_FETCH_PRIO) asm volatile("s_setprio 0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RAY_ABLATE_NOREAD
    asm volatile("" :: "v"(rank));
    out[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int k = 1; k < 6; ++k) out[k] = make_double2(0.0, 0.0);
    asm volatile("" : "+v"(out[0].x), "+v"(out[0].y), "+v"(out[1].x), "+v"(out[1].y), "+v"(out[2].x), "+v"(out[2].y));
    asm volatile("" : "+v"(out[3].x), "+v"(out[3].y), "+v"(out[4].x), "+v"(out[4].y), "+v"(out[5].x), "+v"(out[5].y));
    return;
#endif
    rs.cid = cid;
    rs.addr = mad24(rank, kSlotBytes, a.wrec_v);
    asm volatile("" : "+v"(rs.addr));            // computed once: the look-up after this fetch reuses the register
#pragma unroll
    for (int k = 0; k < 6; ++k) out[k] = lds_ld_d2(rs.addr + 16u * k);
}

// Look-up through the tags (ALL 64 lanes together): the record of cell `cid` if the last
// fetch_records() of this wave brought it in -- for this lane or for any other -- else this lane
// gathers it from memory itself.  Same bytes either way: the path taken never shows in the results.
__device__ __forceinline__ void lookup_records(const char *rec, const GridAcc &a, int i, int j, int cid, double2 out[6], RecState &rs)
{
#if defined(RAY_ABLATE_NOFETCH) || defined(RAY_ABLATE_NOCACHE)
    fetch_records(rec, a, i, j, cid, out, rs);
    return;
#endif
    if (__builtin_amdgcn_ballot_w64(cid != rs.cid) == 0) {           // every lane still in the cell it fetched
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = lds_ld_d2(rs.addr + 16u * k);
#ifdef RAY_STATS
        if (a.lane == 0) { atomicAdd(&g_ray_stats[0], 1ull); atomicAdd(&g_ray_stats[11], 1ull); }
#endif
        return;
    }
    const unsigned tag = lds_ld_u32(tag_addr(a, i, j));
    const unsigned d = tag - ((unsigned)cid << 6);           // = slot on a hit, >= 64 otherwise (kTagEmpty included)
    const bool hit = d < 64u;
    const unsigned slot = d & 63u;                           // a missing lane reads some slot and overwrites it below
    const unsigned saddr = mad24(slot, kSlotBytes, a.wrec_v);
#pragma unroll
    for (int k = 0; k < 6; ++k) out[k] = lds_ld_d2(saddr + 16u * k);
#ifdef RAY_STATS
    {
        const unsigned long long mm = __builtin_amdgcn_ballot_w64(!hit);
        if (a.lane == 0) {
            atomicAdd(&g_ray_stats[0], 1ull);
            if (mm) { atomicAdd(&g_ray_stats[1], 1ull); atomicAdd(&g_ray_stats[2], (unsigned long long)__popcll(mm)); }
        }
    }
#endif
    if (__builtin_amdgcn_ballot_w64(!hit) != 0) {
        if (!hit) {
            const double2 *src = reinterpret_cast<const double2 *>(rec + (size_t)cid * 96u);
#pragma unroll
            for (int k = 0; k < 6; ++k) out[k] = src[k];
        }
    }
}

// ---- arithmetic flavours of the gridded right-hand side and RK4 combination -----------------
// kExact : every operator of the reference's expressions is one IEEE rounding, in the reference's
//          order (bit-identical to the un-fused CPU restatement, oracle/ray_oracle.c).  fma is used
//          only where it provably returns the same bits: x + k*0.5, acc + 2*k (the product is exact)
//          and the two Markstein quotients (header comment).
// kFused : the same formulas evaluated the way a compiler is free to under the reference's @fastmath
//          (NonObstacle.jl:1,6,30,54) -- algebraically identical, rounded differently:
//            cell coordinate  q = fma(x, RN(1/h), RN(-p1/h));  cell = trunc(q);  in-cell offset in CELL units
//                             u = q - floor(q)  (v_fract_f64) instead of dx = x - p[I]: no node table
//            records          in cell units (GridDev::recf: coefficients times [1, h, h, h*h]), so that
//                             t0 + t1 dx + t2 dy + t3 dx dy = fma(u, fma(v, t3', t1'), fma(v, t2', t0))
//            H*(dt/2) folded into the stage update, tau = fma(py, py, px*px), H[2:3] = (-g) * (c*tau).
//          Differs from kExact by rounding only (~1e-12 relative on exits, bound 1e-10 with identical
//          events, tests/test_gpu_parity.py) and is itself bit-identical to the oracle's "fused"
//          flavour, which spells out the same operations (oracle/ray_oracle.c).
enum Arith { kExact = 0, kFused = 1 };

// Out-of-grid tracking: a running UNSIGNED maximum of the raw cell indices (a negative index wraps to
// a huge value), turned into the flag by oob_hit() once per step: one v_max3_u32 per look-up
// instead of two compares and their mask algebra.
typedef unsigned OobMax;
__device__ __forceinline__ int oob_hit(const GridDev &g, OobMax m) { return m > (unsigned)(g.N - 3); }

template <int AR>
__device__ __forceinline__ int cell_of(double x, const GridDev &g, OobMax &oob, double nb, double &u)
{
    // kExact: floor of the correctly rounded quotient.  kFused: one fma, and the conversion's own
    // truncation stands in for floor (equal for q >= 0; q in (-1,0) lands in cell 0 un-flagged,
    // which the oracle's fused flavour defines the same way); u = fract(q) = q - floor(q), clamped
    // below 1 by the instruction (v_fract_f64), is the in-cell offset in cell units.
    const double q = (AR == kFused) ? __builtin_fma(x, g.rh, nb)
                                    : __builtin_floor(div_rn(x - g.p1, g.h, g.rh));
    int iraw, i;                                 // v_cvt truncates and saturates; (int)q would be UB out of range
    asm("v_cvt_i32_f64 %0, %1" : "=v"(iraw) : "v"(q));
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(i) : "v"(iraw), "s"(g.N - 3));
    if (AR == kFused) asm("v_fract_f64 %0, %1" : "=v"(u) : "v"(q));
    oob = (unsigned)iraw > oob ? (unsigned)iraw : oob;
    return i;
}
template <int AR>
__device__ __forceinline__ int cell_of(double x, const GridDev &g, OobMax &oob)
{
    double u;
    return cell_of<AR>(x, g, oob, g.nb, u);
}

// H = [c^2 p ; -grad c * c * tau] from one cell record and the in-cell offsets (NonObstacle.jl:37-42),
// in the chosen arithmetic flavour; also returns c, dc/dx, dc/dy, tau for the Jacobian path.
// kExact: (dx, dy) physical offsets and a record of GridDev::rec; kFused: (dx, dy) = (u, v) in cell units and a
// record of GridDev::recf.
template <int AR>
__device__ __forceinline__ void h_from_record(const double X[4], double dx, double dy, const double2 r[6],
                                              double H[4], double &c, double &gcX, double &gcY, double &tau)
{
    if (AR == kFused) {
        // t0 + t1 dx + t2 dy + t3 dx dy = fma(dx, fma(dy, t3, t1), fma(dy, t2, t0))
        tau = __builtin_fma(X[3], X[3], X[2] * X[2]);
        c = __builtin_fma(dx, __builtin_fma(dy, r[1].y, r[0].y), __builtin_fma(dy, r[1].x, r[0].x));
        gcX = __builtin_fma(dx, __builtin_fma(dy, r[3].y, r[2].y), __builtin_fma(dy, r[3].x, r[2].x));
        gcY = __builtin_fma(dx, __builtin_fma(dy, r[5].y, r[4].y), __builtin_fma(dy, r[5].x, r[4].x));
        const double c2 = c * c, w = c * tau;
        H[0] = c2 * X[2];
        H[1] = c2 * X[3];
        H[2] = (-gcX) * w;
        H[3] = (-gcY) * w;
    } else {
        const double dxdy = dx * dy;
        tau = X[2] * X[2] + X[3] * X[3];
        c = ((r[0].x + dx * r[0].y) + dy * r[1].x) + dxdy * r[1].y;
        gcX = ((r[2].x + dx * r[2].y) + dy * r[3].x) + dxdy * r[3].y;
        gcY = ((r[4].x + dx * r[4].y) + dy * r[5].x) + dxdy * r[5].y;
        const double c2 = c * c;
        H[0] = c2 * X[2];
        H[1] = c2 * X[3];
        H[2] = ((-gcX) * c) * tau;
        H[3] = ((-gcY) * c) * tau;
    }
}

// a2 / a3(H): DiscreteHamilton                                     NonObstacle.jl:30-42
// ALL 64 lanes must be converged here (wave-cooperative record fetch).
// FETCH: wave-cooperative fetch of this position's cells (and tags for them); LOOKUP: served from the
// cells of the previous FETCH where possible.  RK4 stage positions pair up -- stage 3 sits ~dt^2 from
// stage 2, the next step's stage 1 ~dt^2 from stage 4 -- so stages 2 and 4 FETCH, stages 3 and 1 LOOKUP.
enum RecMode { kFetch = 0, kLookup = 1 };

// `lm`: ballot of the lanes that still integrate, taken once per step by the caller.  A finished lane sits
// frozen at its exit point, far from the fan, and would add a run of its own to every fetch until the whole
// wave is done: it looks up the first live lane's cell instead.  What it computes from that record is dropped
// (the callers take a step's result for live lanes only).
template <int AR, int MODE = kFetch>
__device__ __forceinline__ void rhs_grid(const GridDev &g, const GridAcc &a, const double X[4], double H[4], OobMax &oob,
                                         unsigned long long lm, RecState &rs)
{
    // A wave on its way to a fetch goes first: its memory requests leave early and the waves of a SIMD drift out of step,
    // so that one computes while another waits (cell indices -> leaders -> DMA issue at priority 3, back to 0 before the
    // wait; the wait and the read-back at raised priority cost 3 %: profiles/r3).  -1.9 % kernel time.
    if (RAY_FETCH_PRIO && MODE == kFetch) asm volatile("s_setprio %0" :: "n"(RAY_FETCH_PRIO));
    double dx, dy;                                                   // kFused: set here, in cell units
    int i = cell_of<AR>(X[0], g, oob, a.nb_v, dx);
    int j = cell_of<AR>(X[1], g, oob, a.nb_v, dy);
#ifndef RAY_ABLATE_NOGHOST
    asm("" : "+s"(lm));               // compare afresh (s_cmp_lg_u64): a shared result comes back through a VGPR
    if (lm != ~0ull) {                                                // wave-uniform: bit l = lane l is live
        const int first = __builtin_ctzll(lm | (1ull << 63));
        const int i0 = __builtin_amdgcn_readlane(i, first), j0 = __builtin_amdgcn_readlane(j, first);
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(i) : "v"(i0), "v"(i), "s"(lm));   // live ? own : first live lane's
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(j) : "v"(j0), "v"(j), "s"(lm));
    }
#endif
    const int cid = (int)mad24((unsigned)j, (unsigned)(g.N - 1), (unsigned)i);
    double pi_ = 0, pj_ = 0;
    if (AR == kExact) {
        const double *pt = lds_nodes();
        pi_ = pt[i]; pj_ = pt[j];
    }
    const char *rec = reinterpret_cast<const char *>(AR == kFused ? g.recf : g.rec);
    double2 r[6];
#ifdef RAY_ABLATE_HOT
    fetch_records(rec, a, i, j, (cid & 0xff) + 5000, r, rs);   // same run structure, 256-record footprint: always L1-hot
#else
    if (MODE == kLookup) lookup_records(rec, a, i, j, cid, r, rs);
    else fetch_records(rec, a, i, j, cid, r, rs);
#endif
    if (AR == kExact) { dx = X[0] - pi_; dy = X[1] - pj_; }
    double c, gcX, gcY, tau;
    h_from_record<AR>(X, dx, dy, r, H, c, gcX, gcY, tau);
}

// z' * eval[I,J,:] alone (c at the sensor for the initial slowness)  NonObstacle.jl:39,62
// per-lane loads (called under divergence, once per ray); as-written arithmetic in both flavours
__device__ __forceinline__ double speed_grid(const GridDev &g, const double *pt, double x, double y, int &oob)
{
    OobMax om = 0;
    const int i = cell_of<kExact>(x, g, om);
    const int j = cell_of<kExact>(y, g, om);
    oob |= oob_hit(g, om);
    const double2 *r = reinterpret_cast<const double2 *>(g.rec) +
                       ((size_t)i + (size_t)j * (size_t)(g.N - 1)) * 6;
    const double2 e01 = r[0], e23 = r[1];
    const double dx = x - pt[i], dy = y - pt[j];
    return ((e01.x + dx * e01.y) + dy * e23.x) + (dx * dy) * e23.y;
}

// sin and cos as the analytic media and the initial directions take them: ray_trig.h's reproducible evaluation
// inside its range, the platform's beyond (|a| >= 2^18, inf, NaN: nothing the unit disk produces)
__device__ __forceinline__ void sincos_rep(double a, double &sn, double &cs)
{
    if (trig::in_range(a)) trig::sincos_bounded(a, sn, cs);
    else sincos(a, &sn, &cs);
}
// the same out of line: the rare branches of the stepping loop call it instead of carrying a copy each
__device__ __attribute__((noinline)) double2 sincos_rep_call(double a)
{
    double2 r;
    sincos_rep(a, r.x, r.y);
    return r;
}
__device__ __forceinline__ double sin_rep(double a)
{
    double sn, cs;
    sincos_rep(a, sn, cs);
    return sn;
}

// sin/cos of an ANCHOR evaluation, from which the stages are advanced (trig::shift): their arguments differ from the
// anchor's by w * (Y - X_anchor) = O(w dt c).  The difference of the two arguments is taken as computed, so each stage
// still evaluates sin/cos of its own argument w * Y, to 1 ulp of 1, one advance away from a full evaluation.
//   kExact: the anchor is the step's own first stage (every step evaluates it in full).
//   kFused: an anchor serves K consecutive steps (trig::anchor_period; `anchor` = the wave-uniform "this step re-anchors"
//           bit of the argument), the difference is one fma, the advance is trig::shift_fused.  Stage 3 sits O(dt^2) from
//           stage 2 and the next step's stage 1 O(dt^3) from stage 4: where the two differences from the anchor are
//           within trig::kTinyMax of each other the stage follows from the stage BEFORE it by trig::shift_tiny (half the
//           operations; never more than two advances away from a full evaluation: stages 2 and 4 always start at the
//           anchor).  A lane whose argument has moved beyond kShiftMax evaluates in full and -- at a first stage --
//           re-anchors itself.  Every choice is the lane's own: the oracle's scalar restatement takes the same ones.
struct TrigBase {
    double a[kRadialMax], s[kRadialMax], c[kRadialMax];   // per argument: x and y of the sin-product, r_k of the radial sum
    double pd[kRadialMax], ps[kRadialMax], pc[kRadialMax]; // kFused: the stage before -- its difference from the anchor, sin, cos
};
template <int STAGE, int AR>
__device__ __forceinline__ void sincos_stage(double w, double x, bool anchor, TrigBase &tb, int k, double &sn, double &cs)
{
    double &base_a = tb.a[k], &base_s = tb.s[k], &base_c = tb.c[k];
    if (STAGE == 0 && (AR == kExact || anchor)) {
        const double a = w * x;
        if (trig::in_range(a)) trig::sincos_bounded(a, sn, cs);
        else { const double2 r = sincos_rep_call(a); sn = r.x; cs = r.y; }
        base_a = a; base_s = sn; base_c = cs;
        if (AR == kFused) { tb.pd[k] = 0.0; tb.ps[k] = sn; tb.pc[k] = cs; }
    } else if (AR == kExact) {
        const double d = w * x - base_a;
        if (__builtin_fabs(d) <= trig::kShiftMax) trig::shift(base_s, base_c, d, sn, cs);
        else {
            const double a = w * x;
            const double2 r = sincos_rep_call(a); sn = r.x; cs = r.y;
            if (STAGE == 0) { base_a = a; base_s = sn; base_c = cs; }
        }
    } else {
        double d = __builtin_fma(w, x, -base_a);
        const double dl = d - tb.pd[k];
        if ((STAGE == 0 || STAGE == 2) && __builtin_fabs(dl) <= trig::kTinyMax) {
            trig::shift_tiny(tb.ps[k], tb.pc[k], dl, sn, cs);
        } else if (__builtin_fabs(d) <= trig::kShiftMax) {
            trig::shift_fused(base_s, base_c, d, sn, cs);
        } else {
            const double a = w * x;
            const double2 r = sincos_rep_call(a); sn = r.x; cs = r.y;
            if (STAGE == 0) { base_a = a; base_s = sn; base_c = cs; d = 0.0; }
        }
        if (STAGE == 1 || STAGE == 3) { tb.pd[k] = d; tb.ps[k] = sn; tb.pc[k] = cs; }   // read by the stage after
    }
}

// a1: Hamilton with the analytic sin-product medium                  NonObstacle.jl:1-4
// kFused (the reference's Hamilton is @fastmath): speed = fma(A sx, sy, 1), tau = fma(py, py, px px),
// H[2:3] = (-g cos sin) * (speed tau) with the sign folded into the constant.
template <int STAGE, int AR>
__device__ __forceinline__ void rhs_sinprod(const SinProdDev &s, const double X[4], double H[4], TrigBase &tb, unsigned anchor)
{
    double sx, cx, sy, cy;
    sincos_stage<STAGE, AR>(s.wx, X[0], anchor & 1u, tb, 0, sx, cx);
    sincos_stage<STAGE, AR>(s.wy, X[1], anchor & 2u, tb, 1, sy, cy);
    if (AR == kFused) {
        const double speed = __builtin_fma(s.A * sx, sy, 1.0);
        const double tau = __builtin_fma(X[3], X[3], X[2] * X[2]);
        const double s2 = speed * speed, w = speed * tau;
        H[0] = s2 * X[2];
        H[1] = s2 * X[3];
        H[2] = ((-s.gx) * w) * (cx * sy);
        H[3] = ((-s.gy) * w) * (sx * cy);
        return;
    }
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
    return 1 + s.A * sin_rep(s.wx * x) * sin_rep(s.wy * y);
}

// a1 with the commented alternate media (NonObstacleDemo.jl:7-11,15-22; ObstacleDemo.jl:9-11,15-17), as written:
//   c      = 1.0 + a1 * sin(pi * r1) + a2 * sin(pi * r2) + ...             r_k = sqrt((x-x_k)^2 + (y-y_k)^2)
//   grad c = a1 * pi * cos(pi * r1)/r1 * [(x-x1), (y-y1)] + ...
// kFused: the sums as fma chains, r^2 = fma(dy, dy, dx dx), tau and H[2:3] as in rhs_sinprod; the division stays one.
template <int STAGE, int AR>
__device__ __forceinline__ void rhs_radial(const RadialDev &rd, const double X[4], double H[4], TrigBase &tb, unsigned anchor)
{
    constexpr double kPi = 3.141592653589793;
    double speed = 1.0, g0 = 0, g1 = 0;
#pragma unroll
    for (int k = 0; k < kRadialMax; ++k)
        if (k < rd.K) {
            const double dx = X[0] - rd.x[k], dy = X[1] - rd.y[k];
            const double r = __builtin_sqrt(AR == kFused ? __builtin_fma(dy, dy, dx * dx) : dx * dx + dy * dy);
            double sn, cs;
            sincos_stage<STAGE, AR>(kPi, r, anchor & (1u << k), tb, k, sn, cs);
            const double f = ((rd.a[k] * kPi) * cs) / r;
            if (AR == kFused) {
                speed = __builtin_fma(rd.a[k], sn, speed);
                g0 = __builtin_fma(f, dx, g0); g1 = __builtin_fma(f, dy, g1);
            } else {
                speed = speed + rd.a[k] * sn;
                if (k == 0) { g0 = f * dx; g1 = f * dy; }
                else { g0 = g0 + f * dx; g1 = g1 + f * dy; }
            }
        }
    const double s2 = speed * speed;
    if (AR == kFused) {
        const double tau = __builtin_fma(X[3], X[3], X[2] * X[2]);
        const double w = speed * tau;
        H[0] = s2 * X[2];
        H[1] = s2 * X[3];
        H[2] = (-g0) * w;
        H[3] = (-g1) * w;
        return;
    }
    const double tau = X[2] * X[2] + X[3] * X[3];
    H[0] = s2 * X[2];
    H[1] = s2 * X[3];
    H[2] = ((-tau) * g0) * speed;
    H[3] = ((-tau) * g1) * speed;
}
__device__ __forceinline__ double speed_radial(const RadialDev &rd, double x, double y)
{
    constexpr double kPi = 3.141592653589793;
    double speed = 1.0;
    for (int k = 0; k < rd.K; ++k) {
        const double dx = x - rd.x[k], dy = y - rd.y[k];
        speed = speed + rd.a[k] * sin_rep(kPi * __builtin_sqrt(dx * dx + dy * dy));
    }
    return speed;
}

struct MediumDev {
    GridDev g;
    SinProdDev sp;
    RadialDev rd;
};

template <int MED, int AR, int MODE = kFetch, int STAGE = 0>
__device__ __forceinline__ void rhs(const MediumDev &m, const GridAcc &pt, const double X[4], double H[4], OobMax &oob,
                                    unsigned long long lm, RecState &rs, TrigBase &tb, unsigned anchor)
{
    if (MED == kGrid) rhs_grid<AR, MODE>(m.g, pt, X, H, oob, lm, rs);
    else if (MED == kRadial) rhs_radial<STAGE, AR>(m.rd, X, H, tb, anchor);
    else rhs_sinprod<STAGE, AR>(m.sp, X, H, tb, anchor);
}

// fma with a destination of its own (v_fma_f64): the stage update Y = X + H*hdt must leave X alive, and
// the compiler otherwise picks the two-address v_fmac_f64 plus a 64-bit copy of X per component.
__device__ __forceinline__ double fma_vop3(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// One classical RK4 step X -> Xn with step dt (the callers keep X for lanes that were not live).
// FORM 0 = a6/a7: k = H*dt; X + k1/2; (k1+2k2+2k3+k4)/6.0         NonObstacle.jl:72-76
// FORM 1 = a8/a9: k = H;    X + k/2*dt; (k1+2k2+2k3+k4)*dt/6.0    NonObstacle.jl:161-164,171
// kFused (both forms, they are the same formula): Y = fma(H, dt/2, X), Xn = fma(sum, dt*RN(1/6), X)
// `tb`, `anchor`: analytic media only -- the anchors of the sin/cos advances (they outlive the step in kFused) and the
// wave-uniform bit mask of the arguments that re-anchor in this step (sincos_stage).
template <int MED, int AR, int FORM>
__device__ __forceinline__ void rk4(const MediumDev &m, const GridAcc &pt, const double X[4], double Xn[4], double dt,
                                    unsigned long long lm, OobMax &oob, RecState &rs, TrigBase &tb, unsigned anchor)
{
    double H[4], Y[4], acc[4];
    if (AR == kFused) {
        const double hdt = 0.5 * dt, dt6 = dt * 0.16666666666666666;
        rhs<MED, AR, kLookup, 0>(m, pt, X, H, oob, lm, rs, tb, anchor);
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[k] = H[k]; Y[k] = fma_vop3(H[k], hdt, X[k]); }
        rhs<MED, AR, kFetch, 1>(m, pt, Y, H, oob, lm, rs, tb, anchor);
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[k] = __builtin_fma(2.0, H[k], acc[k]); Y[k] = fma_vop3(H[k], hdt, X[k]); }
        rhs<MED, AR, kLookup, 2>(m, pt, Y, H, oob, lm, rs, tb, anchor);
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[k] = __builtin_fma(2.0, H[k], acc[k]); Y[k] = fma_vop3(H[k], dt, X[k]); }
        rhs<MED, AR, kFetch, 3>(m, pt, Y, H, oob, lm, rs, tb, anchor);
#pragma unroll
        for (int k = 0; k < 4; ++k) Xn[k] = __builtin_fma(acc[k] + H[k], dt6, X[k]);
    } else if (FORM == 0) {
        rhs<MED, AR, kLookup, 0>(m, pt, X, H, oob, lm, rs, tb, anchor);
#pragma unroll
        for (int k = 0; k < 4; ++k) { double kk = H[k] * dt; acc[k] = kk; Y[k] = __builtin_fma(kk, 0.5, X[k]); }
        rhs<MED, AR, kFetch, 1>(m, pt, Y, H, oob, lm, rs, tb, anchor);
#pragma unroll
        for (int k = 0; k < 4; ++k) { double kk = H[k] * dt; acc[k] = __builtin_fma(2.0, kk, acc[k]); Y[k] = __builtin_fma(kk, 0.5, X[k]); }
        rhs<MED, AR, kLookup, 2>(m, pt, Y, H, oob, lm, rs, tb, anchor);
#pragma unroll
        for (int k = 0; k < 4; ++k) { double kk = H[k] * dt; acc[k] = __builtin_fma(2.0, kk, acc[k]); Y[k] = X[k] + kk; }
        rhs<MED, AR, kFetch, 3>(m, pt, Y, H, oob, lm, rs, tb, anchor);
#pragma unroll
        for (int k = 0; k < 4; ++k) { double kk = H[k] * dt; Xn[k] = X[k] + div6(acc[k] + kk); }
    } else {
        const double hdt = 0.5 * dt;              // (H/2)*dt == RN(H*(dt/2)): halving is exact
        rhs<MED, AR, kLookup, 0>(m, pt, X, H, oob, lm, rs, tb, anchor);
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[k] = H[k]; Y[k] = X[k] + H[k] * hdt; }
        rhs<MED, AR, kFetch, 1>(m, pt, Y, H, oob, lm, rs, tb, anchor);
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[k] = __builtin_fma(2.0, H[k], acc[k]); Y[k] = X[k] + H[k] * hdt; }
        rhs<MED, AR, kLookup, 2>(m, pt, Y, H, oob, lm, rs, tb, anchor);
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[k] = __builtin_fma(2.0, H[k], acc[k]); Y[k] = X[k] + H[k] * dt; }
        rhs<MED, AR, kFetch, 3>(m, pt, Y, H, oob, lm, rs, tb, anchor);
#pragma unroll
        for (int k = 0; k < 4; ++k) Xn[k] = X[k] + div6((acc[k] + H[k]) * dt);
    }
}

__device__ __forceinline__ double norm2(double a, double b) { return __builtin_sqrt(a * a + b * b); }

__device__ __forceinline__ unsigned wave_sum(unsigned v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

} // namespace ray
