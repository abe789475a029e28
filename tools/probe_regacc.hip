// tools/probe_regacc.hip — a probe, not part of the engine: the panel product with the row group's accumulators in the
// REGISTER FILE instead of LDS.
//
// Why: C2 (10M rows x 32 uniform-random columns) is bound by L2 line operations; the gathers of x cost (1 - e^-L)/L lines
// per entry, L = entries of a row group per 128-byte line of x = rows x 32 x 16 / ncol.  With the accumulators in LDS a
// group holds 20000 rows (L = 1.02: 0.63 lines per entry).  The register file of a CU is 512 KiB, but a lane cannot index
// its registers with a run-time value, so a product cannot be added "to the register of its row" where it is formed.
// Here the products of a chunk (8192 entries, ordered by x line) are added into a COMPACT staging array in LDS - one slot
// per distinct row of the chunk, numbered in the order (wavefront, register, lane) of the rows' owners - and after the
// chunk's barrier every wavefront collects its rows' slots register by register: for register r (a compile-time index,
// the loop is unrolled) a 64-bit mask says which of its 64 lanes' rows occur in the chunk; lane l reads slot
// base + popcount(mask below l).  8 wavefronts x 77 registers x 64 lanes = 39424 rows per group (L = 2.02: 0.43 lines per
// entry), one round of 254 workgroups, x swept once per XCD.
//   probe_regacc [rows] [per_row] [ncol]      defaults 10000000 32 rows
// Prints the kernel's time per product and the largest |y - y_ref| / (|A||x|).
// Build (here, no GPU needed): hipcc -O3 -std=c++17 --offload-arch=gfx950 -o build/tools/probe_regacc tools/probe_regacc.hip
//   switches: -DSUPER=1 (-DMASKS_VEC=0) -DSUBSTEPS=2 -DSTREAM_FIRST=0 -DGATHER_AHEAD=0 -DCBLOCK=6 -DSTREAM_NT=0, timing experiments -DNO_COLLECT=1 -DNO_GATHER=1
//   -DNO_ADDS=1 -DCOLLECT_NOLDS=1 -DGATHER_SHAPE=1|2.  Counters: tools/pmc_probe_regacc.sh.
// RESULT (profiles/r06_probe_regacc.txt): correct, 23 % fewer L2 operations than the engine's panel kernel, and slower (C2 1.19 ms
// against 1.114): with 154 accumulator registers per lane too little is left for the loads in flight.  Not part of the engine.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do                                                                                 \
    {                                                                                  \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess)                                                          \
        {                                                                              \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

using u32x2 = unsigned __attribute__((ext_vector_type(2)));
using f64x2 = double __attribute__((ext_vector_type(2)));

constexpr int kWaves   = 8;
constexpr int kThreads = kWaves * 64;
#ifndef KREG
#define KREG 77
#endif
constexpr int kReg     = KREG;                  // accumulators per lane
constexpr int kWords   = kWaves * kReg;         // 64-bit masks per chunk
constexpr int kGroup   = kWords * 64;           // rows per group
#ifndef SUPER
#define SUPER 0  // 1: chunks of 16384 entries and ONE staging array (14336 slots), the collection after the chunk instead of beside the next
#endif
constexpr int kChunk   = SUPER ? 16384 : 8192;  // entries per chunk
constexpr int kPairs   = kChunk / 2 / kThreads; // pair blocks per chunk (8 / 16): a lane's 16 / 32 entries
constexpr int kSlotBits = SUPER ? 14 : 13;      // slot < 8192 / 14336
constexpr int kEPL     = kChunk / kThreads;     // entries per lane and chunk
constexpr int kStage   = SUPER ? 14336 : kChunk; // slots of one staging array
constexpr int kWbStride = 16;                   // int32 per chunk: first slot of every wavefront, total at [8] (read through the scalar cache)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double unit(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0; }

__global__ void gen_kernel(int64_t nnz, int k, int ncol, int* col, double* val, uint64_t* key, unsigned* idx)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x)
    {
        const uint64_t h   = mix((uint64_t)e * 2 + 1);
        const int      c   = (int)((h >> 32) * (uint64_t)ncol >> 32);
        const int64_t  row = e / k;
        col[e]             = c;
        val[e]             = unit(mix(h));
        key[e]             = ((uint64_t)(row / kGroup) << 32) | (unsigned)c;
        idx[e]             = (unsigned)e;
    }
}
__global__ void genx_kernel(int n, double* x)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = unit(mix(0xABCDEF00ull + i));
}
__global__ void ref_kernel(int nrow, int k, const int* col, const double* val, const double* x, double* y, double* mag)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    double s = 0.0, m = 0.0;
    for (int j = 0; j < k; ++j)
    {
        const double p = val[(int64_t)r * k + j] * x[col[(int64_t)r * k + j]];
        s += p;
        m += fabs(p);
    }
    y[r]   = s;
    mag[r] = m;
}

// ---- layout: one workgroup per chunk -----------------------------------------------------------------------------
// sorted entry s of the chunk (s = u * 512 + t) lies in pair block u / 2, lane t, half u & 1
__global__ __launch_bounds__(kThreads) void chunk_build_kernel(int k, int64_t nnz, const int* __restrict__ choff_of_group, const int* __restrict__ group_of_chunk,
                                                               const uint64_t* __restrict__ skey, const unsigned* __restrict__ sidx,
                                                               const double* __restrict__ val, unsigned* __restrict__ pw, double* __restrict__ pv,
                                                               uint64_t* __restrict__ masks, int* __restrict__ wbase, int* __restrict__ cbase,
                                                               int* __restrict__ err)
{
    __shared__ unsigned bits[kWords * 2];
    __shared__ int      pre[kWords + 1];
    __shared__ int      s_cb;
    const int     c    = blockIdx.x;
    const int     g    = group_of_chunk[c];
    const int     t    = threadIdx.x;
    const int64_t gbeg = (int64_t)g * kGroup * k;
    const int64_t gend = min(nnz, (int64_t)(g + 1) * kGroup * k);
    const int64_t cbeg = gbeg + (int64_t)(c - choff_of_group[g]) * kChunk;
    for (int i = t; i < kWords * 2; i += kThreads) bits[i] = 0;
    if (t == 0) s_cb = (int)((unsigned)skey[cbeg]) & ~15;
    __syncthreads();
    int    loc[kEPL], cc[kEPL];
    double vv[kEPL];
#pragma unroll
    for (int u = 0; u < kEPL; ++u)
    {
        const int64_t e = cbeg + u * kThreads + t;
        if (e < gend)
        {
            const unsigned id = sidx[e];
            loc[u]            = (int)(id / (unsigned)k) - g * kGroup;
            cc[u]             = (int)(unsigned)skey[e];
            vv[u]             = val[id];
            atomicOr(&bits[loc[u] >> 5], 1u << (loc[u] & 31));
        }
        else
        {
            loc[u] = -1;
            cc[u]  = s_cb;
            vv[u]  = 0.0;
        }
    }
    __syncthreads();
    if (t == 0)
    {
        int run = 0;
        for (int i = 0; i < kWords; ++i)
        {
            pre[i] = run;
            run += __popc(bits[2 * i]) + __popc(bits[2 * i + 1]);
        }
        pre[kWords] = run;
        if (run > kStage) atomicExch(err, 2);
    }
    __syncthreads();
    const int cb = s_cb;
    for (int i = t; i < kWords; i += kThreads) masks[(size_t)c * kWords + i] = ((uint64_t)bits[2 * i + 1] << 32) | bits[2 * i];
    if (t <= kWaves) wbase[(size_t)c * kWbStride + t] = pre[t * kReg];
    if (t == 0) cbase[c] = cb;
#pragma unroll
    for (int u = 0; u < kEPL; ++u)
    {
        unsigned word = 0;
        if (loc[u] >= 0)
        {
            const int      wd   = loc[u] >> 6, b = loc[u] & 63;
            const uint64_t m    = ((uint64_t)bits[2 * wd + 1] << 32) | bits[2 * wd];
            const int      slot = pre[wd] + __popcll(m & ((1ull << b) - 1ull));
            const unsigned rel  = (unsigned)(cc[u] - cb);
            if (rel >= (1u << (32 - kSlotBits))) atomicExch(err, 1);
            word = (rel << kSlotBits) | (unsigned)slot;
        }
        const size_t at = (((size_t)c * kPairs + (u >> 1)) * kThreads + t) * 2 + (u & 1);
        pw[at]          = word;
        pv[at]          = vv[u];
    }
}

// ---- the product -------------------------------------------------------------------------------------------------
#ifndef NO_COLLECT
#define NO_COLLECT 0  // timing experiments (wrong sums): 1 = the products are never collected
#endif
#ifndef NO_GATHER
#define NO_GATHER 0   // 1 = x[0] instead of the gathers
#endif
#ifndef NO_ADDS
#define NO_ADDS 0       // timing experiment: the products are summed in a register instead of added into LDS
#endif
#ifndef COLLECT_NOLDS
#define COLLECT_NOLDS 0 // timing experiment: the collection does its arithmetic without reading the staging array
#endif
#ifndef MASKS_VEC
#define MASKS_VEC 1      // (SUPER) the masks of a chunk come through a vector load issued at the top of the chunk
#endif
#ifndef GATHER_SHAPE
#define GATHER_SHAPE 0
#endif
#ifndef STREAM_FIRST
#define STREAM_FIRST 1  // the stream of sub-step s + 3 is requested before the gathers of s + 1 (best: profiles/r06_probe_regacc.txt)
#endif
#ifndef STREAM_NT
#define STREAM_NT 1
#endif
#ifndef GATHER_AHEAD
#define GATHER_AHEAD 1
#endif
#ifndef SUBSTEPS
#define SUBSTEPS 4
#endif
constexpr int kSub = SUBSTEPS;       // sub-steps per chunk: a lane has 16 / kSub entries in flight between its loads and its LDS adds
constexpr int kPS  = kPairs / kSub;  // pair blocks per sub-step
struct Raw
{
    u32x2 w[kPS];
    f64x2 v[kPS];
};

__device__ __forceinline__ void load_raw(Raw& R, const u32x2* __restrict__ pw, const f64x2* __restrict__ pv, int c, int sub, int t)
{
    const size_t at = ((size_t)c * kPairs + sub * kPS) * kThreads + t;
#pragma unroll
    for (int j = 0; j < kPS; ++j)
    {
        R.w[j] = STREAM_NT ? __builtin_nontemporal_load(pw + at + (size_t)j * kThreads) : pw[at + (size_t)j * kThreads];
        R.v[j] = STREAM_NT ? __builtin_nontemporal_load(pv + at + (size_t)j * kThreads) : pv[at + (size_t)j * kThreads];
    }
}
__device__ __forceinline__ void gather(const Raw& R, const double* __restrict__ x, int cb, double (&xv)[2 * kPS])
{
#pragma unroll
    for (int j = 0; j < kPS; ++j)
    {
        // (uniform base + 32-bit byte offset: one address register per gather)
        if (NO_GATHER)
        {
            xv[2 * j]     = (double)(R.w[j].x >> kSlotBits);
            xv[2 * j + 1] = (double)(R.w[j].y >> kSlotBits);
            continue;
        }
        // GATHER_SHAPE (timing experiments, wrong sums): 1 = every lane of a wavefront reads the address of its first lane's entry,
        // 2 = the lanes read 64 consecutive doubles from there (one 512-byte run)
        unsigned o0 = R.w[j].x >> kSlotBits, o1 = R.w[j].y >> kSlotBits;
        if (GATHER_SHAPE >= 1)
        {
            o0 = __builtin_amdgcn_readfirstlane(o0);
            o1 = __builtin_amdgcn_readfirstlane(o1);
        }
        if (GATHER_SHAPE == 2)
        {
            o0 += threadIdx.x & 63;
            o1 += threadIdx.x & 63;
        }
        xv[2 * j]     = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(x) + (((unsigned)cb + o0) << 3));
        xv[2 * j + 1] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(x) + (((unsigned)cb + o1) << 3));
    }
}
__device__ __forceinline__ void adds(const Raw& R, const double (&xv)[2 * kPS], double* sb, double& g_sink)
{
#pragma unroll
    for (int j = 0; j < kPS; ++j)
    {
        if (NO_ADDS)
        {
            g_sink += R.v[j].x * xv[2 * j] + R.v[j].y * xv[2 * j + 1] + (double)(R.w[j].x & 1u);
            continue;
        }
        atomicAdd(&sb[R.w[j].x & ((1u << kSlotBits) - 1u)], R.v[j].x * xv[2 * j]);
        atomicAdd(&sb[R.w[j].y & ((1u << kSlotBits) - 1u)], R.v[j].y * xv[2 * j + 1]);
    }
}
// registers R0 .. R1-1 of this wavefront collect their rows' sums of a chunk: mk[] are the registers' masks (loaded by the
// caller ahead of time: a scalar load is ~300 clocks away), `base` runs along
#ifndef CBLOCK
#define CBLOCK 10
#endif

// the same masks out of two vector registers (lane l holds mask l, and mask 64 + l): requested by a VECTOR load a whole chunk
// ahead - a scalar load of a line nobody has touched takes ~2 us (HBM), and the collection used to wait for four of them in turn
template <int R0, int R1>
__device__ __forceinline__ void masks_from_lanes(uint64_t (&mk)[R1 - R0], uint64_t v0, uint64_t v1)
{
#pragma unroll
    for (int r = R0; r < R1; ++r)
    {
        const uint64_t v  = r < 64 ? v0 : v1;
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, r & 63), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), r & 63);
        mk[r - R0]        = ((uint64_t)hi << 32) | lo;
    }
}
template <int R0, int R1>
__device__ __forceinline__ void load_masks(uint64_t (&mk)[R1 - R0], const uint64_t* __restrict__ mp)
{
#pragma unroll
    for (int r = R0; r < R1; ++r) mk[r - R0] = mp[r];
}
template <int R0, int R1>
__device__ __forceinline__ void collect(double (&acc)[kReg], const uint64_t (&mk)[R1 - R0], const double* sprev, int& base)
{
#pragma unroll
    for (int b0 = R0; b0 < R1; b0 += CBLOCK)
    {
        double v[CBLOCK];
#pragma unroll
        for (int q = 0; q < CBLOCK; ++q)
            if (b0 + q < R1)
            {
                const uint64_t m   = mk[b0 + q - R0];
                const unsigned cnt = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                v[q]               = COLLECT_NOLDS ? (double)cnt : sprev[base + (int)cnt];
                base += __builtin_popcountll(m);
            }
        __builtin_amdgcn_sched_barrier(0);  // the block's LDS reads go out together ...
#pragma unroll
        for (int q = 0; q < CBLOCK; ++q)
            if (b0 + q < R1)
            {
                acc[b0 + q] += __builtin_amdgcn_inverse_ballot_w64(mk[b0 + q - R0]) ? v[q] : 0.0;
                // (the adds are pure arithmetic: without a use that is ordered with the fence the instruction selector is free to
                // place them blocks later, masks held - and spilled - until then)
                asm volatile("" : "+v"(acc[b0 + q]));
            }
        __builtin_amdgcn_sched_barrier(0);  // ... and are added before the next block's
    }
}

__global__ __launch_bounds__(kThreads) void regacc_kernel(const int* __restrict__ choff, int ngroups, int nrow, int zero_chunk,
                                                          const uint64_t* __restrict__ masks, const int* __restrict__ wbase,
                                                          const int* __restrict__ cbase, const u32x2* __restrict__ pw,
                                                          const f64x2* __restrict__ pv, const double* __restrict__ x, double* __restrict__ y)
{
    extern __shared__ double stg[];  // 2 x kChunk slots + 64 of slack (lanes without a row read past their wavefront's range)
    const int t    = threadIdx.x;
    const int lane = t & 63;
    const int w    = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x)
    {
        double acc[kReg];
        double sink = 0.0;
#pragma unroll
        for (int r = 0; r < kReg; ++r) acc[r] = 0.0;
        for (int i = t; i < (SUPER ? 1 : 2) * kStage + 64; i += kThreads) stg[i] = 0.0;
        __syncthreads();
        const int c0 = choff[g], c1 = choff[g + 1];
        // four register sets in turn: the stream of sub-step s + 3 is requested while s is multiplied (HBM latency is ~2 us:
        // a CU needs ~40 KB of the stream in flight)
#if SUPER
        // SUPER: a chunk is 16384 entries = four steps of 8 per lane; the products of the whole chunk go into ONE staging array,
        // then barrier - collection (77 registers, once per 16384 entries) - barrier.  The first step's gathers and the second
        // step's stream of the NEXT chunk are requested before the collection and are in flight beside it.
        static_assert(kPS * kSub == kPairs && kSub == 4, "four steps of kPS pair blocks");
        {
            Raw A, B;
            double xa[2 * kPS], xb[2 * kPS];
            load_raw(A, pw, pv, c0, 0, t);
            gather(A, x, cbase[c0], xa);
            load_raw(B, pw, pv, c0, 1, t);
            for (int c = c0; c < c1; ++c)
            {
                const int cb = cbase[c];
                const int cn = c + 1 < c1 ? c + 1 : c;
                static_assert(kReg <= 128, "two mask registers per lane");
                const uint64_t* mpv = masks + (size_t)c * kWords + (size_t)w * kReg;
                uint64_t        vm0 = 0, vm1 = 0;
                if (MASKS_VEC)
                {
                    vm0 = mpv[lane < kReg ? lane : 0];
                    vm1 = mpv[64 + lane < kReg ? 64 + lane : 0];
                }
                // step 0: xa (requested before the last collection), B holds step 1's stream
                __builtin_amdgcn_sched_barrier(0);
                adds(A, xa, stg, sink);
                __builtin_amdgcn_sched_barrier(0);
                gather(B, x, cb, xb);
                __builtin_amdgcn_sched_barrier(0);
                load_raw(A, pw, pv, c, 2, t);
                __builtin_amdgcn_sched_barrier(0);
                adds(B, xb, stg, sink);  // step 1
                __builtin_amdgcn_sched_barrier(0);
                gather(A, x, cb, xa);
                __builtin_amdgcn_sched_barrier(0);
                load_raw(B, pw, pv, c, 3, t);
                __builtin_amdgcn_sched_barrier(0);
                adds(A, xa, stg, sink);  // step 2
                __builtin_amdgcn_sched_barrier(0);
                gather(B, x, cb, xb);
                __builtin_amdgcn_sched_barrier(0);
                load_raw(A, pw, pv, cn, 0, t);
                __builtin_amdgcn_sched_barrier(0);
                adds(B, xb, stg, sink);  // step 3
                __builtin_amdgcn_sched_barrier(0);
                gather(A, x, cbase[cn], xa);  // the next chunk's step 0 (the last chunk gathers itself once more: unused)
                __builtin_amdgcn_sched_barrier(0);
                load_raw(B, pw, pv, cn, 1, t);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();  // every product of chunk c is in the staging array
                {
                    const uint64_t* mp   = masks + (size_t)c * kWords + (size_t)w * kReg;
                    const int       lo   = wbase[(size_t)c * kWbStride + w], hi = wbase[(size_t)c * kWbStride + w + 1];
                    int             base = lo;
#define PART(S)                                                                                   \
    {                                                                                             \
        uint64_t mk[kReg * ((S) + 1) / 4 - kReg * (S) / 4];                                       \
        if (MASKS_VEC)                                                                            \
            masks_from_lanes<kReg * (S) / 4, kReg * ((S) + 1) / 4>(mk, vm0, vm1);                 \
        else                                                                                      \
            load_masks<kReg * (S) / 4, kReg * ((S) + 1) / 4>(mk, mp);                             \
        if (!NO_COLLECT) collect<kReg * (S) / 4, kReg * ((S) + 1) / 4>(acc, mk, stg, base);       \
    }
                    PART(0) PART(1) PART(2) PART(3)
#undef PART
                    for (int i = lo + lane; i < hi; i += 64) stg[i] = 0.0;
                }
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();  // ... and collected and cleared before chunk c + 1 adds into it
            }
        }
#else
#if SUBSTEPS == 2
        // two sub-steps of 8 entries per lane, two register sets: a wait for gathers also waits for every OLDER load (vmcnt
        // counts in order), so what matters is how long ago the stream in front of them was requested - here a whole half chunk
        Raw A, B;
        load_raw(A, pw, pv, c0, 0, t);
#else
        Raw A, B, C, D;
        static_assert(kSub == 4, "four sub-steps per chunk");
        load_raw(A, pw, pv, c0, 0, t);
        load_raw(B, pw, pv, c0, 1, t);
        load_raw(C, pw, pv, c0, 2, t);
#endif
#if GATHER_AHEAD && SUBSTEPS == 4
        double xa[2 * kPS], xb[2 * kPS];
        gather(A, x, cbase[c0], xa);
#endif
        for (int c = c0; c < c1; ++c)
        {
            double*         sb    = stg + (c & 1) * kChunk;
            double*         sprev = stg + ((c + 1) & 1) * kChunk;
            const int       cp    = c > c0 ? c - 1 : zero_chunk;  // the chunk collected during this one (first: an empty record)
            const uint64_t* mp    = masks + (size_t)cp * kWords + (size_t)w * kReg;
            const int       lo    = wbase[(size_t)cp * kWbStride + w], hi = wbase[(size_t)cp * kWbStride + w + 1];
            const int       cb    = cbase[c];
            const int       cn    = c + 1 < c1 ? c + 1 : c;  // (the last chunk re-reads itself: no load behind a branch)
            int             base  = lo;
            // the gathers of sub-step s are in flight while the stream of s + 3 is requested and a quarter of the PREVIOUS
            // chunk's sums is collected into the registers
#if SUBSTEPS == 2
#define SUBSTEP(CUR, NXT, S)                                                                                   \
    {                                                                                                          \
        double   xv[2 * kPS];                                                                                  \
        uint64_t mk[kReg * ((S) + 1) / kSub - kReg * (S) / kSub];                                              \
        gather(CUR, x, cb, xv);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        load_raw(NXT, pw, pv, (S) == 0 ? c : cn, ((S) + 1) % kSub, t);                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        load_masks<kReg * (S) / kSub, kReg * ((S) + 1) / kSub>(mk, mp);                                        \
        if (!NO_COLLECT) collect<kReg * (S) / kSub, kReg * ((S) + 1) / kSub>(acc, mk, sprev, base);            \
        if ((S) + 1 == kSub)                                                                                   \
            for (int i = lo + lane; i < hi; i += 64) sprev[i] = 0.0;                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        adds(CUR, xv, sb, sink);                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
            SUBSTEP(A, B, 0)
            SUBSTEP(B, A, 1)
#undef SUBSTEP
#elif !GATHER_AHEAD
#define SUBSTEP(CUR, NXT, S)                                                                                   \
    {                                                                                                          \
        double   xv[2 * kPS];                                                                                  \
        uint64_t mk[kReg * ((S) + 1) / kSub - kReg * (S) / kSub];                                              \
        load_masks<kReg * (S) / kSub, kReg * ((S) + 1) / kSub>(mk, mp);                                        \
        gather(CUR, x, cb, xv);                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        load_raw(NXT, pw, pv, (S) == 0 ? c : cn, ((S) + 3) % kSub, t);                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (!NO_COLLECT) collect<kReg * (S) / kSub, kReg * ((S) + 1) / kSub>(acc, mk, sprev, base);            \
        if ((S) + 1 == kSub)                                                                                   \
            for (int i = lo + lane; i < hi; i += 64) sprev[i] = 0.0; /* this wavefront's slots, for chunk c + 1 */ \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        adds(CUR, xv, sb, sink);                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
            SUBSTEP(A, D, 0)
            SUBSTEP(B, A, 1)
            SUBSTEP(C, B, 2)
            SUBSTEP(D, C, 3)
#undef SUBSTEP
#else
            // GATHER_AHEAD: the gathers of sub-step s + 1 go out before sub-step s waits for its own (requested one sub-step
            // ago): two sub-steps' gathers in flight per lane; XC holds the values sub-step s multiplies, XN receives s + 1's
            const int cbn = cbase[cn];
#define SUBSTEP(CUR, NEXT, TGT, XC, XN, S)                                                                     \
    {                                                                                                          \
        uint64_t mk[kReg * ((S) + 1) / kSub - kReg * (S) / kSub];                                              \
        load_masks<kReg * (S) / kSub, kReg * ((S) + 1) / kSub>(mk, mp);                                        \
        if (STREAM_FIRST) load_raw(TGT, pw, pv, (S) == 0 ? c : cn, ((S) + 3) % kSub, t);                       \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        gather(NEXT, x, (S) + 1 < kSub ? cb : cbn, XN);                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (!STREAM_FIRST) load_raw(TGT, pw, pv, (S) == 0 ? c : cn, ((S) + 3) % kSub, t);                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (!NO_COLLECT) collect<kReg * (S) / kSub, kReg * ((S) + 1) / kSub>(acc, mk, sprev, base);            \
        if ((S) + 1 == kSub)                                                                                   \
            for (int i = lo + lane; i < hi; i += 64) sprev[i] = 0.0;                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        adds(CUR, XC, sb, sink);                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
            SUBSTEP(A, B, D, xa, xb, 0)
            SUBSTEP(B, C, A, xb, xa, 1)
            SUBSTEP(C, D, B, xa, xb, 2)
            SUBSTEP(D, A, C, xb, xa, 3)
#undef SUBSTEP
#endif
            __syncthreads();
        }
        {
            const int       cp    = c1 - 1;
            const double*   sprev = stg + (cp & 1) * kChunk;
            const uint64_t* mp    = masks + (size_t)cp * kWords + (size_t)w * kReg;
            int             base  = wbase[(size_t)cp * kWbStride + w];
#define TAIL(S)                                                                            \
    {                                                                                      \
        uint64_t mk[kReg * ((S) + 1) / kSub - kReg * (S) / kSub];                          \
        load_masks<kReg * (S) / kSub, kReg * ((S) + 1) / kSub>(mk, mp);                    \
        collect<kReg * (S) / kSub, kReg * ((S) + 1) / kSub>(acc, mk, sprev, base);         \
    }
#if SUBSTEPS == 2
            TAIL(0) TAIL(1)
#else
            TAIL(0) TAIL(1) TAIL(2) TAIL(3)
#endif
#undef TAIL
        }
#endif  // SUPER
        if (NO_ADDS) acc[0] += sink;
        // (y is padded to whole groups: no bounds test, so that the loads of a block of registers go out together)
        double* yw = y + (size_t)g * kGroup + (size_t)w * kReg * 64 + lane;
#pragma unroll
        for (int r0 = 0; r0 < kReg; r0 += 11)
        {
            double yo[11];
#pragma unroll
            for (int q = 0; q < 11; ++q)
                if (r0 + q < kReg) yo[q] = yw[(r0 + q) * 64];
#pragma unroll
            for (int q = 0; q < 11; ++q)
                if (r0 + q < kReg) yw[(r0 + q) * 64] = yo[q] + acc[r0 + q];
        }
        __syncthreads();
    }
}

__global__ void err_kernel(int n, const double* y, const double* yr, const double* mag, double* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double e = fabs(y[i] - yr[i]) / fmax(mag[i], 1e-300);
    // (positive doubles order like their bit patterns)
    atomicMax((unsigned long long*)out, (unsigned long long)__double_as_longlong(e));
}

int main(int argc, char** argv)
{
    const int     nrow = argc > 1 ? atoi(argv[1]) : 10000000;
    const int     k    = argc > 2 ? atoi(argv[2]) : 32;
    const int     ncol = argc > 3 ? atoi(argv[3]) : nrow;
    const int64_t nnz  = (int64_t)nrow * k;
    if (nnz >= ((int64_t)1 << 32)) return fprintf(stderr, "too many entries for this probe\n"), 1;
    printf("probe_regacc: %d rows x %d per row over %d columns; %d wavefronts x %d registers x 64 lanes = %d rows per group, chunks of %d\n", nrow, k, ncol,
           kWaves, kReg, kGroup, kChunk);
    int *     col;
    double *  val, *x, *y, *yr, *mag;
    uint64_t *key, *skey;
    unsigned *idx, *sidx;
    CK(hipMalloc(&col, nnz * 4));
    CK(hipMalloc(&val, nnz * 8));
    CK(hipMalloc(&key, nnz * 8));
    CK(hipMalloc(&skey, nnz * 8));
    CK(hipMalloc(&idx, nnz * 4));
    CK(hipMalloc(&sidx, nnz * 4));
    CK(hipMalloc(&x, (size_t)ncol * 8));
    CK(hipMalloc(&y, ((size_t)nrow + kGroup) * 8));  // padded to whole groups
    CK(hipMalloc(&yr, (size_t)nrow * 8));
    CK(hipMalloc(&mag, (size_t)nrow * 8));
    gen_kernel<<<4096, 256>>>(nnz, k, ncol, col, val, key, idx);
    genx_kernel<<<1024, 256>>>(ncol, x);
    ref_kernel<<<(nrow + 255) / 256, 256>>>(nrow, k, col, val, x, yr, mag);
    CK(hipDeviceSynchronize());
    const int ngroups = (nrow + kGroup - 1) / kGroup;
    int       gbits   = 1;
    while ((1 << gbits) < ngroups) ++gbits;
    {
        size_t tmp_bytes = 0;
        CK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key, skey, idx, sidx, (int)nnz, 0, 32 + gbits));
        void* tmp;
        CK(hipMalloc(&tmp, tmp_bytes));
        CK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key, skey, idx, sidx, (int)nnz, 0, 32 + gbits));
        CK(hipDeviceSynchronize());
        CK(hipFree(tmp));
    }
    CK(hipFree(key));
    CK(hipFree(idx));
    std::vector<int> choff(ngroups + 1, 0), gofc;
    for (int g = 0; g < ngroups; ++g)
    {
        const int64_t cnt = std::min<int64_t>(nnz, (int64_t)(g + 1) * kGroup * k) - (int64_t)g * kGroup * k;
        const int     nch = (int)((cnt + kChunk - 1) / kChunk);
        choff[g + 1]      = choff[g] + nch;
        for (int i = 0; i < nch; ++i) gofc.push_back(g);
    }
    const int nchunks = choff[ngroups];
    int *     d_choff, *d_gofc, *cbase, *err;
    unsigned* pw;
    double*   pv;
    uint64_t* masks;
    int*      wbase;
    CK(hipMalloc(&d_choff, sizeof(int) * choff.size()));
    CK(hipMalloc(&d_gofc, sizeof(int) * gofc.size()));
    CK(hipMemcpy(d_choff, choff.data(), sizeof(int) * choff.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_gofc, gofc.data(), sizeof(int) * gofc.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&cbase, sizeof(int) * (nchunks + 1)));
    CK(hipMalloc(&err, sizeof(int)));
    CK(hipMemset(err, 0, sizeof(int)));
    CK(hipMalloc(&pw, (size_t)nchunks * kChunk * 4));
    CK(hipMalloc(&pv, (size_t)nchunks * kChunk * 8));
    CK(hipMalloc(&masks, (size_t)(nchunks + 1) * kWords * 8));
    CK(hipMalloc(&wbase, (size_t)(nchunks + 1) * kWbStride * 4));
    CK(hipMemset(masks, 0, (size_t)(nchunks + 1) * kWords * 8));  // record `nchunks` stays empty: what the first chunk of a group collects
    CK(hipMemset(wbase, 0, (size_t)(nchunks + 1) * kWbStride * 4));
    chunk_build_kernel<<<nchunks, kThreads>>>(k, nnz, d_choff, d_gofc, skey, sidx, val, pw, pv, masks, wbase, cbase, err);
    CK(hipDeviceSynchronize());
    int h_err = 0;
    CK(hipMemcpy(&h_err, err, sizeof(int), hipMemcpyDeviceToHost));
    if (h_err) return fprintf(stderr, h_err == 2 ? "a chunk holds more distinct rows than the staging array has slots\n" : "a chunk spans more columns than its packed words hold\n"), 1;
    CK(hipFree(skey));
    CK(hipFree(sidx));
    const double stream_gb = ((double)nchunks * kChunk * 12 + (double)nchunks * (kWords * 8 + kWbStride * 4 + 4)) / 1e9;
    printf("layout: %d groups, %d chunks, %.3f GB streamed per product (%.3f GB = 12 bytes per stored entry)\n", ngroups, nchunks, stream_gb,
           (double)nnz * 12 / 1e9);

    const size_t lds = sizeof(double) * ((SUPER ? 1 : 2) * kStage + 64);
    CK(hipFuncSetAttribute((const void*)regacc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = std::min(ngroups, 256);
    CK(hipMemset(y, 0, ((size_t)nrow + kGroup) * 8));
    regacc_kernel<<<grid, kThreads, lds>>>(d_choff, ngroups, nrow, nchunks, masks, wbase, cbase, (const u32x2*)pw, (const f64x2*)pv, x, y);
    CK(hipDeviceSynchronize());
    double* d_e;
    CK(hipMalloc(&d_e, 8));
    CK(hipMemset(d_e, 0, 8));
    err_kernel<<<(nrow + 255) / 256, 256>>>(nrow, y, yr, mag, d_e);
    double h_e = 0;
    CK(hipMemcpy(&h_e, d_e, 8, hipMemcpyDeviceToHost));
    printf("max |y - y_ref| / (|A||x|) = %.3e\n", h_e);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
    {
        for (int i = 0; i < 3; ++i) regacc_kernel<<<grid, kThreads, lds>>>(d_choff, ngroups, nrow, nchunks, masks, wbase, cbase, (const u32x2*)pw, (const f64x2*)pv, x, y);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) regacc_kernel<<<grid, kThreads, lds>>>(d_choff, ngroups, nrow, nchunks, masks, wbase, cbase, (const u32x2*)pw, (const f64x2*)pv, x, y);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 20;
        const double alg = (double)nnz * 12 + (double)(nrow + 1) * 4 + (double)ncol * 8 + (double)nrow * 16;
        printf("regacc_kernel: %.4f ms per product = %.1f GFLOP/s, %.3f of the 8 TB/s roofline on %.3f GB algorithmic\n", ms, 2.0 * nnz / ms / 1e6,
               alg / (ms * 1e-3) / 8e12, alg / 1e9);
    }
    return 0;
}
