// tools/probe_occ.hip — gather rate from an L2-resident table at the panel kernel's occupancy: ONE 1024-thread workgroup
// per CU (156 KB of LDS claimed), U independent gathers in flight per lane.  Compare with probe_gather (8 blocks of
// 256 threads per CU: 258 Ggather/s).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
template <int U>
__global__ __launch_bounds__(1024) void k(const double* __restrict__ t, uint32_t mask, int G, double* __restrict__ out)
{
    extern __shared__ double lds[];
    if (threadIdx.x == 0) lds[0] = 0.0;
    const uint64_t gid = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    double acc = 0.0;
    for (int g = 0; g < G; g += U)
    {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; u += 2)
        {
            const uint64_t r = mix(gid * 1315423911ull + g + u);
            v[u]     = t[(r >> 32) & mask];
            v[u + 1] = t[r & mask];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc == 123.456) out[gid] = acc + lds[0];
}
template <int U> void run(const double* t, uint32_t mask, double* out)
{
    const int G = 512, blocks = 256 * 4;  // four generations of one workgroup per CU
    CK(hipFuncSetAttribute((const void*)k<U>, hipFuncAttributeMaxDynamicSharedMemorySize, 156000));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<U>, dim3(blocks), dim3(1024), 156000, 0, t, mask, G, out); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<U>, dim3(blocks), dim3(1024), 156000, 0, t, mask, G, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    const double n = (double)blocks * 1024 * G;
    printf("16 wavefronts per CU, %2d gathers in flight per lane: %.3f ms  %.1f Ggather/s\n", U, ms, n / ms / 1e6);
}
int main()
{
    const uint32_t doubles = 1u << 18;
    double *t, *out; CK(hipMalloc(&t, doubles * 8)); CK(hipMemset(t, 0, doubles * 8)); CK(hipMalloc(&out, 64 << 20));
    run<2>(t, doubles - 1, out); run<4>(t, doubles - 1, out); run<8>(t, doubles - 1, out); run<16>(t, doubles - 1, out); run<32>(t, doubles - 1, out);
    return 0;
}
