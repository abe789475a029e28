// tools/probe_hybrid.hip — do gathers on the scalar memory path (s_load) ADD to the vector path's rate, or do both
// draw on the same L2 budget?  Per workgroup of 4 wavefronts, `nscalar` of them gather from an L2-resident table with
// wave-uniform scalar loads, the others with divergent vector loads; both rates are reported.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t lcg(uint32_t s) { return s * 1664525u + 1013904223u; }
__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k(const double* __restrict__ t, uint32_t mask, int Gv, int Gs, int nscalar, double* __restrict__ out)
{
    const int wave = threadIdx.x >> 6;
    double acc = 0.0;
    if (wave < nscalar)
    {
        uint32_t s = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + wave) * 2654435761u + 12345u);
        for (int g = 0; g < Gs; g += 16)
        {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { s = lcg(s); v[u] = t[(s >> 8) & mask]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
    }
    else
    {
        const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
        for (int g = 0; g < Gv; g += 4)
        {
            const uint64_t r0 = mix(gid * 1315423911ull + g), r1 = mix(r0);
            acc += t[(r0 >> 32) & mask] + t[r0 & mask] + t[(r1 >> 32) & mask] + t[r1 & mask];
        }
    }
    if (acc == 123.456) out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main()
{
    const uint32_t doubles = 1u << 18;
    double *t, *out; CK(hipMalloc(&t, doubles * 8)); CK(hipMemset(t, 0, doubles * 8)); CK(hipMalloc(&out, 64 << 20));
    const int blocks = 8192;
    for (int nscalar : {0, 1, 2, 4})
    {
        // size the two loops so that both kinds of wavefront run about equally long (rates from the single-path probes)
        const int Gv = 64, Gs = nscalar ? 1536 : 0;
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, t, doubles - 1, Gv, Gs, nscalar, out); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, t, doubles - 1, Gv, Gs, nscalar, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
        const double nv = (double)blocks * (4 - nscalar) * 64 * Gv, ns = (double)blocks * nscalar * Gs;
        printf("%d scalar + %d vector wavefronts per workgroup: %.3f ms  vector %.1f Ggather/s  scalar %.1f Gload/s  total %.1f G/s\n", nscalar,
               4 - nscalar, ms, nv / ms / 1e6, ns / ms / 1e6, (nv + ns) / ms / 1e6);
    }
    return 0;
}
