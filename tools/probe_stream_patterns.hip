// tools/probe_stream_patterns.hip — does it matter to HBM how 256 workgroups divide a streamed buffer between them?
//   front    every workgroup reads its slice of ONE moving front (grid-stride: what probe_gather's `stream` does)
//   chunked  every workgroup streams a contiguous chunk of its own (what the two-phase reduce phase and the panel kernel do)
//   blocked  chunked in the large, interleaved in blocks of B bytes: workgroup w reads blocks w, w + 256, ... (B = 16 KB .. 1 MB)
// One workgroup of 1024 threads per CU, U 16-byte loads per lane in flight, nontemporal.  Run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/probe_stream_patterns.hip -o tools/bin/probe_stream_patterns && tools/bin/probe_stream_patterns
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                          \
    do                                                                 \
    {                                                                  \
        hipError_t e = (x);                                            \
        if (e != hipSuccess)                                           \
        {                                                              \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));     \
            exit(1);                                                   \
        }                                                              \
    } while (0)

typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int T = 1024, U = 8;

// block_pairs: pairs (16 B) per block; workgroup w reads blocks w, w + G, w + 2G, ...
__global__ __launch_bounds__(T) void stream_blocks(const f64x2* __restrict__ src, size_t npairs, size_t block_pairs, double* __restrict__ out)
{
    double       acc     = 0.0;
    const size_t nblocks = npairs / block_pairs;
    for (size_t b = blockIdx.x; b < nblocks; b += gridDim.x)
    {
        const f64x2* p = src + b * block_pairs;
        for (size_t i0 = 0; i0 < block_pairs; i0 += (size_t)T * U)
        {
            f64x2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                const size_t i = i0 + (size_t)u * T + threadIdx.x;
                v[u]           = i < block_pairs ? __builtin_nontemporal_load(p + i) : f64x2{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
        }
    }
    if (acc == 123.456) out[0] = acc;
}

int main(int argc, char** argv)
{
    const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)3 << 30;
    f64x2*       src;
    double*      out;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(src, 0, bytes));
    const size_t npairs = bytes / 16;
    hipEvent_t   e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char* name, size_t block_pairs) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep)
        {
            hipLaunchKernelGGL(stream_blocks, dim3(256), dim3(T), 0, 0, src, npairs, block_pairs, out);
            CK(hipEventRecord(e0));
            for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(stream_blocks, dim3(256), dim3(T), 0, 0, src, npairs, block_pairs, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms / 3 < best ? ms / 3 : best;
        }
        printf("%-44s %.4f ms  %.2f TB/s\n", name, best, (double)(npairs / block_pairs * block_pairs) * 16 / best / 1e9);
    };
    printf("%zu MB, 256 workgroups x 1024 threads, %d x 16 B per lane in flight, nontemporal loads\n", bytes >> 20, U);
    run("front (blocks of one set = 128 KB)", (size_t)T * U);
    run("blocks of 256 KB", (size_t)16384);
    run("blocks of 1 MB", (size_t)65536);
    run("blocks of 4 MB", (size_t)262144);
    run("chunked (one contiguous chunk per workgroup)", npairs / 256);
    return 0;
}
