// tools/probe_gather.hip — micro-probes of the memory system behaviour that bounds SpMV on gfx950.
//
//   probe_gather gather  <table_doubles> <gathers_per_lane>   random 8-byte gathers from a table
//   probe_gather stream  <bytes>                              nontemporal 16 B/lane streaming read
//   probe_gather ldsadd  <slots> <adds_per_lane>              random ds_add_f64 into an LDS table
//
// Built by `make tools`; run on the GPU box only.  Not part of the engine.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                      \
    do                                                                             \
    {                                                                              \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess)                                                       \
        {                                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                 \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// MODE 0: plain loads, 1: nontemporal, 2: agent-scope relaxed atomic load (sc1, bypasses L1)
template <int MODE>
__device__ __forceinline__ double gload(const double* p)
{
    if constexpr (MODE == 1) return __builtin_nontemporal_load(p);
    if constexpr (MODE == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// every lane: G gathers, 4 independent in flight, indices uniform over [0,T)
template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const double* __restrict__ table, uint32_t T, int G,
                                                     double* __restrict__ out)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    double         acc = 0.0;
    for (int g = 0; g < G; g += 4)
    {
        const uint64_t r0 = mix(gid * 1315423911ull + g);
        const uint64_t r1 = mix(r0);
        const uint32_t i0 = (uint32_t)(((r0 >> 32) * T) >> 32);
        const uint32_t i1 = (uint32_t)(((r0 & 0xffffffffu) * (uint64_t)T) >> 32);
        const uint32_t i2 = (uint32_t)(((r1 >> 32) * T) >> 32);
        const uint32_t i3 = (uint32_t)(((r1 & 0xffffffffu) * (uint64_t)T) >> 32);
        const double   a = gload<MODE>(table + i0), b = gload<MODE>(table + i1), c = gload<MODE>(table + i2), d = gload<MODE>(table + i3);
        acc += a + b + c + d;
    }
    if (acc == 123.456) out[gid] = acc;  // never true; keeps the loads alive
}

typedef double f64x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void stream_kernel(const f64x2* __restrict__ src, size_t n, double* __restrict__ out)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    {
        const f64x2 v = __builtin_nontemporal_load(src + i);
        acc += v.x + v.y;
    }
    if (acc == 123.456) out[0] = acc;
}

__global__ __launch_bounds__(256) void ldsadd_kernel(int slots, int adds, double* __restrict__ out)
{
    extern __shared__ double acc[];
    for (int i = threadIdx.x; i < slots; i += 256) acc[i] = 0.0;
    __syncthreads();
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (int g = 0; g < adds; g += 2)
    {
        const uint64_t r  = mix(gid * 2654435761ull + g);
        const uint32_t i0 = (uint32_t)(((r >> 32) * (uint64_t)slots) >> 32);
        const uint32_t i1 = (uint32_t)(((r & 0xffffffffu) * (uint64_t)slots) >> 32);
        atomicAdd(&acc[i0], 1.0);
        atomicAdd(&acc[i1], 0.5);
    }
    __syncthreads();
    double s = 0.0;
    for (int i = threadIdx.x; i < slots; i += 256) s += acc[i];
    if (s == 123.456) out[gid] = s;
}

static float time_it(void (*launch)(void*), void* arg, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch(arg);  // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch(arg);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

struct GArgs
{
    const double* table;
    uint32_t      T;
    int           G;
    double*       out;
    int           blocks;
    int           mode;
};

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    double* out;
    CK(hipMalloc(&out, 64 << 20));
    if (!strcmp(argv[1], "gather"))
    {
        // sweep table sizes unless one is given
        const int G      = argc > 3 ? atoi(argv[3]) : 64;
        const int blocks = 256 * 8 * 4;  // 4 generations of full occupancy
        size_t    sizes[] = {32u << 10, 128u << 10, 256u << 10, 512u << 10, 1u << 20, 2u << 20, 4u << 20, 10u << 20, 20u << 20, 80u << 20};
        for (size_t s : sizes)
        {
            if (argc > 2 && atol(argv[2]) > 0 && (size_t)atol(argv[2]) != s) continue;
            double* table;
            CK(hipMalloc(&table, s * 8));
            CK(hipMemset(table, 0, s * 8));
            for (int mode = 0; mode < 3; ++mode)
            {
            GArgs ga{table, (uint32_t)s, G, out, blocks, mode};
            auto  launch = [](void* p) {
                GArgs* g = (GArgs*)p;
                if (g->mode == 0) hipLaunchKernelGGL(gather_kernel<0>, dim3(g->blocks), dim3(256), 0, 0, g->table, g->T, g->G, g->out);
                if (g->mode == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(g->blocks), dim3(256), 0, 0, g->table, g->T, g->G, g->out);
                if (g->mode == 2) hipLaunchKernelGGL(gather_kernel<2>, dim3(g->blocks), dim3(256), 0, 0, g->table, g->T, g->G, g->out);
            };
            float        ms = time_it(launch, &ga, 5);
            const double n  = (double)blocks * 256 * G;
            printf("mode %d (0 plain 1 nt 2 sc1) ", mode);
            printf("gather table %8.2f MB (%9zu doubles): %8.3f ms  %7.1f Ggather/s  (%6.2f TB/s at 64B, %6.2f TB/s at 128B)\n",
                   s * 8 / 1048576.0, s, ms, n / ms / 1e6, n * 64 / ms / 1e9, n * 128 / ms / 1e9);
            }
            CK(hipFree(table));
        }
    }
    else if (!strcmp(argv[1], "stream"))
    {
        const size_t bytes = argc > 2 ? (size_t)atoll(argv[2]) : (size_t)4 << 30;
        f64x2*       src;
        CK(hipMalloc(&src, bytes));
        CK(hipMemset(src, 0, bytes));
        struct SA { const f64x2* s; size_t n; double* o; } sa{src, bytes / 16, out};
        auto launch = [](void* p) {
            SA* a = (SA*)p;
            hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, 0, a->s, a->n, a->o);
        };
        float ms = time_it(launch, &sa, 5);
        printf("stream %zu MB nontemporal 16B/lane: %.3f ms  %.2f TB/s\n", bytes >> 20, ms, bytes / ms / 1e9);
    }
    else if (!strcmp(argv[1], "ldsadd"))
    {
        const int slots = argc > 2 ? atoi(argv[2]) : 16384;
        const int adds  = argc > 3 ? atoi(argv[3]) : 256;
        CK(hipFuncSetAttribute((const void*)ldsadd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, slots * 8));
        struct LA { int s, a; double* o; } la{slots, adds, out};
        auto launch = [](void* p) {
            LA* a = (LA*)p;
            hipLaunchKernelGGL(ldsadd_kernel, dim3(256), dim3(256), a->s * 8, 0, a->s, a->a, a->o);
        };
        float        ms = time_it(launch, &la, 5);
        const double n  = 256.0 * 256 * adds;
        printf("ldsadd %d slots, %d adds/lane, 256 WG x 256 thr: %.3f ms  %.1f Gadd/s chip, %.2f adds/clk/CU @2.4GHz\n", slots,
               adds, ms, n / ms / 1e6, n / 256 / (ms * 1e-3 * 2.4e9));
    }
    return 0;
}
