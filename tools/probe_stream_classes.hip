// tools/probe_stream_classes.hip — does a pure READ stream care which pieces of device memory it comes from?
// The two-phase product's stream between the phases does (three classes of memory; gigabytes of one class under a
// scattered-write stream are ~10 % slower together).  C3's ELL product reads 2 GB of values and varies 0.33-0.40 ms by box
// and allocation.  Here: 12 allocations of 1 GB; a kernel shaped like ell_diag_kernel_x2 over TILED values (workgroup b reads
// a contiguous 256 KB tile, 64 steps of 4 KB, 4 in flight) takes tile b from piece b % P — timed for every single piece
// (P = 1), every pair (P = 2) and a sample of triples (P = 3).
//   hipcc -O3 --offload-arch=gfx950 tools/probe_stream_classes.hip -o tools/bin/probe_stream_classes && tools/bin/probe_stream_classes
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                      \
    do                                                             \
    {                                                              \
        hipError_t e = (x);                                        \
        if (e != hipSuccess)                                       \
        {                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

typedef double f64x2 __attribute__((ext_vector_type(2)));
struct pieces
{
    const double* p[4];
};
constexpr int kTile = 512, kSlots = 64;  // a tile: 512 rows x 64 slots x 8 B = 256 KB

__global__ __launch_bounds__(256) void read_tiles(pieces tab, int P, double* __restrict__ y)
{
    const int     piece = blockIdx.x % P, tile = blockIdx.x / P;
    const double* v     = tab.p[piece] + (size_t)tile * kTile * kSlots + 2 * threadIdx.x;
    f64x2         acc{0.0, 0.0};
    for (int s0 = 0; s0 < kSlots; s0 += 4)
    {
        f64x2 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(v + (size_t)(s0 + u) * kTile));
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += t[u];
    }
    *reinterpret_cast<f64x2*>(y + (size_t)blockIdx.x * kTile + 2 * threadIdx.x) = acc;
}

int main(int argc, char** argv)
{
    const int    npieces = argc > 1 ? atoi(argv[1]) : 12;
    const size_t bytes   = (size_t)1 << 30;
    const int    tiles   = (int)(bytes / (kTile * kSlots * 8));  // 4096 tiles per piece
    std::vector<double*> pc((size_t)npieces);
    for (auto& p : pc)
    {
        CK(hipMalloc(&p, bytes));
        CK(hipMemset(p, 0, bytes));
    }
    double* y;
    CK(hipMalloc(&y, sizeof(double) * (size_t)3 * tiles * kTile));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](std::vector<int> sel) {
        pieces tab{};
        for (size_t i = 0; i < sel.size(); ++i) tab.p[i] = pc[(size_t)sel[i]];
        const int P = (int)sel.size();
        float     best = 1e30f;
        for (int rep = 0; rep < 3; ++rep)
        {
            hipLaunchKernelGGL(read_tiles, dim3((unsigned)(tiles * P)), dim3(256), 0, 0, tab, P, y);
            CK(hipEventRecord(e0));
            for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(read_tiles, dim3((unsigned)(tiles * P)), dim3(256), 0, 0, tab, P, y);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 4);
        }
        CK(hipGetLastError());
        return (double)P * bytes / best / 1e9;  // TB/s... GB per ms = TB/s
    };
    printf("%d pieces of 1 GB; rate of the read stream in TB/s\nsingle pieces:", npieces);
    for (int i = 0; i < npieces; ++i) printf(" %.2f", time({i}));
    printf("\npairs (row i, column j > i):\n");
    double lo = 1e9, hi = 0;
    for (int i = 0; i < npieces; ++i)
    {
        printf("  %2d:", i);
        for (int j = 0; j < npieces; ++j)
        {
            if (j <= i)
            {
                printf("     ");
                continue;
            }
            const double r = time({i, j});
            lo = std::min(lo, r);
            hi = std::max(hi, r);
            printf(" %.2f", r);
        }
        printf("\n");
    }
    printf("pairs: %.2f .. %.2f TB/s\ntriples (i, i+%d, i+%d):", lo, hi, npieces / 3, 2 * (npieces / 3));
    for (int i = 0; i < npieces / 3; ++i) printf(" %.2f", time({i, i + npieces / 3, i + 2 * (npieces / 3)}));
    printf("\n");
    return 0;
}
