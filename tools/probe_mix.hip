// tools/probe_mix.hip — can one CU overlap an HBM stream with L2-served divergent gathers?
// One 1024-thread workgroup per CU (the panel kernel's shape).  Per step and lane: U divergent 8-byte gathers from
// a table that fits every XCD's L2 (one 128-byte line per lane and load) and U 12-byte streamed entries (4 B + 8 B,
// nontemporal, from a buffer far larger than the caches).
//   mode 0  gathers only          mode 1  stream only
//   mode 2  phased: gathers(b), stream(b+1), use gathers(b), use stream(b+1)     (the two-stage pipeline)
//   mode 3  ring: gathers(b+1), stream(b+3), use gathers(b)                      (stream kept two steps ahead)
//   mode 4  as 3 with the stream FIRST in each step
// If mode 3 runs at max(mode 0, mode 1) the vector memory pipe overlaps the two; at their sum it does not.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mixu(uint32_t z)
{
    z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16; return z;
}
template <int U, int MODE>
__global__ __launch_bounds__(1024) void k(const double* __restrict__ tab, uint32_t lines, const uint32_t* __restrict__ sw,
                                          const double* __restrict__ sv, size_t per_wg, int steps, double* __restrict__ out)
{
    const uint32_t* pw = sw + (size_t)blockIdx.x * per_wg;
    const double*   pv = sv + (size_t)blockIdx.x * per_wg;
    const unsigned  tid = threadIdx.x;
    double acc = 0.0;
    uint32_t seed = blockIdx.x * 1024u + tid;
    auto gather = [&](double (&x)[U], int b) {
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const uint32_t l = (uint32_t)(((uint64_t)mixu(seed + (uint32_t)(b * U + u) * 2654435761u) * lines) >> 32);
            x[u] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(tab) + ((size_t)l << 7));
        }
    };
    auto stream = [&](uint32_t (&w)[U], double (&v)[U], int b) {
        const size_t off = (size_t)b * U * 1024;
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            w[u] = __builtin_nontemporal_load(pw + off + u * 1024 + tid);
            v[u] = __builtin_nontemporal_load(pv + off + u * 1024 + tid);
        }
    };
    auto use_g = [&](const double (&x)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += x[u];
    };
    auto use_s = [&](const uint32_t (&w)[U], const double (&v)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u] + (double)w[u];
    };
    if constexpr (MODE == 7 || MODE == 8)
    {
        // roles split by WAVEFRONT inside every workgroup: NSW wavefronts only stream (all of the workgroup's entries),
        // the others only gather (all of its gathers): same work per CU as mode 0 plus mode 1
        constexpr int NSW = MODE == 7 ? 4 : 8, NGW = 16 - NSW;
        const int wave = tid >> 6, lane = tid & 63;
        if (wave < NSW)
        {
            // 16 wavefronts' worth of 64-entry rows per (step, u): this wavefront takes rows wave, wave + NSW, ...
            for (int b = 0; b < steps; ++b)
            {
                const size_t off = (size_t)b * U * 1024;
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < 16 / NSW; ++r)
                    {
                        const unsigned e = u * 1024 + (wave + r * NSW) * 64 + lane;
                        acc += (double)__builtin_nontemporal_load(pw + off + e) + __builtin_nontemporal_load(pv + off + e);
                    }
            }
        }
        else
        {
            const int total = steps * 16 / NGW + 1;  // gather steps per gathering wavefront
            double xa[U], xb[U];
            gather(xa, 0);
            for (int b = 0; b < total; b += 2)
            {
                gather(xb, b + 1); use_g(xa);
                gather(xa, b + 2); use_g(xb);
            }
            use_g(xa);
        }
    }
    else if constexpr (MODE == 5 || MODE == 6)
    {
        // roles split by CU inside every XCD: half of the workgroups only gather, half only stream, each twice the
        // steps, so the chip does the same work as mode 0 plus mode 1.  MODE 6: the gather half alone (others idle).
        const bool streamer = (blockIdx.x >> 3) & 1;
        if (!streamer)
        {
            double xa[U], xb[U];
            gather(xa, 0);
            for (int b = 0; b < 2 * steps; b += 2)
            {
                gather(xb, b + 1); use_g(xa);
                gather(xa, b + 2); use_g(xb);
            }
            use_g(xa);
        }
        else if (MODE == 5)
        {
            // this workgroup's share and its neighbour's (the gather-only workgroup 8 below)
            const uint32_t* p2w = sw + (size_t)(blockIdx.x - 8) * per_wg;
            const double*   p2v = sv + (size_t)(blockIdx.x - 8) * per_wg;
            for (int half = 0; half < 2; ++half)
            {
                const uint32_t* qw = half ? p2w : pw;
                const double*   qv = half ? p2v : pv;
                for (int b = 0; b < steps; ++b)
                {
                    const size_t off = (size_t)b * U * 1024;
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        acc += (double)__builtin_nontemporal_load(qw + off + u * 1024 + tid) + __builtin_nontemporal_load(qv + off + u * 1024 + tid);
                }
            }
        }
    }
    else if constexpr (MODE == 0)
    {
        double xa[U], xb[U];
        gather(xa, 0);
        for (int b = 0; b < steps; b += 2)
        {
            gather(xb, b + 1); use_g(xa);
            gather(xa, b + 2); use_g(xb);
        }
        use_g(xa);
    }
    else if constexpr (MODE == 1)
    {
        uint32_t wa[U], wb[U]; double va[U], vb[U];
        stream(wa, va, 0);
        for (int b = 0; b < steps; b += 2)
        {
            stream(wb, vb, b + 1); use_s(wa, va);
            stream(wa, va, b + 2); use_s(wb, vb);
        }
        use_s(wa, va);
    }
    else if constexpr (MODE == 2)
    {
        uint32_t w[U]; double v[U], x[U];
        stream(w, v, 0);
        for (int b = 0; b < steps; ++b)
        {
            use_s(w, v);          // "unpack": the stream of this step must be back
            gather(x, b);
            stream(w, v, b + 1);
            use_g(x);             // waits for the gathers only (issued before the stream)
        }
    }
    else
    {
        uint32_t w[4][U]; double v[4][U], x[2][U];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < U; ++u) { w[s][u] = 0; v[s][u] = 0.0; }
#pragma unroll
        for (int u = 0; u < U; ++u) x[0][u] = x[1][u] = 0.0;
        for (int b0 = 0; b0 < steps; b0 += 4)
        {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
            {
                const int b = b0 + kk;
                use_s(w[(kk + 1) % 4], v[(kk + 1) % 4]);  // the entries of step b+1 (streamed two steps ago)
                if constexpr (MODE == 4) stream(w[(kk + 3) % 4], v[(kk + 3) % 4], b + 3);
                gather(x[(kk + 1) % 2], b + 1);
                if constexpr (MODE == 3) stream(w[(kk + 3) % 4], v[(kk + 3) % 4], b + 3);
                use_g(x[kk % 2]);
            }
        }
    }
    if (acc == 123.456) out[blockIdx.x] = acc;
}
template <int U, int MODE>
float run(const double* tab, uint32_t lines, const uint32_t* sw, const double* sv, size_t per_wg, int steps, double* out)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<U, MODE>), dim3(256), dim3(1024), 0, 0, tab, lines, sw, sv, per_wg, steps, out); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<U, MODE>), dim3(256), dim3(1024), 0, 0, tab, lines, sw, sv, per_wg, steps, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 3;
}
template <int U> void all(const double* tab, uint32_t lines, const uint32_t* sw, const double* sv, size_t per_wg, double* out)
{
    const int steps = (int)(per_wg / (U * 1024)) - 8;  // look-ahead stays inside the workgroup's share
    const double ent = (double)steps * U * 1024 * 256;
    const float t0 = run<U, 0>(tab, lines, sw, sv, per_wg, steps, out), t1 = run<U, 1>(tab, lines, sw, sv, per_wg, steps, out);
    const float t2 = run<U, 2>(tab, lines, sw, sv, per_wg, steps, out), t3 = run<U, 3>(tab, lines, sw, sv, per_wg, steps, out);
    const float t4 = run<U, 4>(tab, lines, sw, sv, per_wg, steps, out);
    const float t7 = run<U, 7>(tab, lines, sw, sv, per_wg, steps, out), t8 = run<U, 8>(tab, lines, sw, sv, per_wg, steps, out);
    const float t5 = run<U, 5>(tab, lines, sw, sv, per_wg, steps, out), t6 = run<U, 6>(tab, lines, sw, sv, per_wg, steps, out);
    printf("U=%d, %d steps/WG, table %u lines: gathers only %.3f ms (%.0f Glines/s) | stream only %.3f ms (%.2f TB/s) | phased %.3f | ring %.3f | ring stream-first %.3f | sum %.3f max %.3f | roles split by CU (16 gather + 16 stream CUs per XCD, same total work) %.3f, its gather half alone %.3f | roles split by wavefront: 4 stream + 12 gather %.3f, 8 + 8 %.3f\n",
           U, steps, lines, t0, ent / t0 / 1e6, t1, ent * 12 / t1 / 1e9, t2, t3, t4, t0 + t1, t0 > t1 ? t0 : t1, t5, t6, t7, t8);
}
int main()
{
    const size_t per_wg = (size_t)1 << 20;           // entries per workgroup: 12 MiB of stream each, 3 GiB in all
    uint32_t* sw; double *sv, *tab, *out;
    CK(hipMalloc(&sw, per_wg * 256 * 4)); CK(hipMalloc(&sv, per_wg * 256 * 8));
    CK(hipMemset(sw, 0, per_wg * 256 * 4)); CK(hipMemset(sv, 0, per_wg * 256 * 8));
    CK(hipMalloc(&out, 4096));
    for (uint32_t lines : {8192u, 16384u})           // 1 MiB, 2 MiB of table: resident in every XCD's 4 MiB L2
    {
        CK(hipMalloc(&tab, (size_t)lines * 128)); CK(hipMemset(tab, 0, (size_t)lines * 128));
        all<4>(tab, lines, sw, sv, per_wg, out);
        all<8>(tab, lines, sw, sv, per_wg, out);
        CK(hipFree(tab));
    }
    return 0;
}
