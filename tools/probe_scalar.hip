// tools/probe_scalar.hip — can the SCALAR memory path (s_load through the scalar data cache) carry random 8-byte
// gathers of an L2-resident table, next to the vector path?  Every wavefront issues wave-uniform loads from random
// addresses (16 independent per iteration); compare with probe_gather's 265 Ggather/s on the vector path.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t lcg(uint32_t s) { return s * 1664525u + 1013904223u; }

__global__ __launch_bounds__(256) void k_scalar(const double* __restrict__ t, uint32_t mask, int G, double* __restrict__ out)
{
    // everything below is wave-uniform: the compiler keeps it in SGPRs and emits s_load_dwordx2
    uint32_t s = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 12345u);
    double   acc = 0.0;
    for (int g = 0; g < G; g += 16)
    {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
        {
            s    = lcg(s);
            v[u] = t[(s >> 8) & mask];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    if (acc == 123.456) out[blockIdx.x] = acc;
}

int main()
{
    const uint32_t doubles = 1u << 18;  // 2 MB table
    double *t, *out; CK(hipMalloc(&t, doubles * 8)); CK(hipMemset(t, 0, doubles * 8)); CK(hipMalloc(&out, 1 << 20));
    for (int waves_per_cu : {4, 8, 16, 32})
    {
        const int blocks = 256 * waves_per_cu / 4 * 4;  // 4 waves per block, several generations
        const int G = 4096;
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, t, doubles - 1, G, out); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, t, doubles - 1, G, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
        const double n = (double)blocks * 4 * G;  // one load per wavefront per step
        printf("scalar gathers, %d blocks x 4 waves: %.3f ms  %.2f Gload/s chip  %.3f loads/clk/CU @2.4GHz\n", blocks, ms, n / ms / 1e6,
               n / ms / 1e6 / 256 / 2.4);
    }
    return 0;
}
