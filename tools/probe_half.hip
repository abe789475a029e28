// tools/probe_half.hip — is the unit a divergent gather pays for a 128-byte line or a 64-byte half line?
// Pairs of lanes read the same random 128-byte line of an L2-resident table, either the same 64-byte half
// (slots 0,1) or different halves (slots 0,8).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
template <int OTHER_SLOT>
__global__ __launch_bounds__(256) void k(const double* __restrict__ t, uint32_t lines, int G, double* __restrict__ out)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t grp = gid / 2;
    const uint32_t slot = (gid & 1) ? OTHER_SLOT : 0;
    double acc = 0.0;
    for (int g = 0; g < G; g += 4)
    {
        const uint64_t r0 = mix(grp * 1315423911ull + g), r1 = mix(r0);
        const uint32_t l0 = (uint32_t)(((r0 >> 32) * lines) >> 32), l1 = (uint32_t)(((r0 & 0xffffffffu) * (uint64_t)lines) >> 32);
        const uint32_t l2 = (uint32_t)(((r1 >> 32) * lines) >> 32), l3 = (uint32_t)(((r1 & 0xffffffffu) * (uint64_t)lines) >> 32);
        acc += t[l0 * 16 + slot] + t[l1 * 16 + slot] + t[l2 * 16 + slot] + t[l3 * 16 + slot];
    }
    if (acc == 123.456) out[gid] = acc;
}
template <int S> float run(const double* t, uint32_t lines, int G, double* out, int blocks)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<S>, dim3(blocks), dim3(256), 0, 0, t, lines, G, out); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<S>, dim3(blocks), dim3(256), 0, 0, t, lines, G, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5;
}
int main()
{
    const uint32_t doubles = 262144, lines = doubles / 16;
    const int G = 64, blocks = 8192;
    double *t, *out; CK(hipMalloc(&t, doubles * 8)); CK(hipMemset(t, 0, doubles * 8)); CK(hipMalloc(&out, 64 << 20));
    const double n = (double)blocks * 256 * G;
    float a = run<1>(t, lines, G, out, blocks), b = run<8>(t, lines, G, out, blocks);
    printf("pairs in the same 64-byte half : %.3f ms  %.1f Ggather/s\n", a, n / a / 1e6);
    printf("pairs in different 64-byte halves of one 128-byte line: %.3f ms  %.1f Ggather/s\n", b, n / b / 1e6);
    return 0;
}
