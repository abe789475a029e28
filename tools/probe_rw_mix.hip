// tools/probe_rw_mix.hip — what does HBM give a kernel that reads AND writes large streams at once?
// Phase A of the two-phase product reads 4.3 GB and writes 2.6 GB in 1.18 ms: exactly what the read rate (7 TB/s) and
// the write rate (4.5 TB/s) of this chip cost ONE AFTER THE OTHER.  Is that the memory system (reads and writes share
// one budget) or the kernel (each workgroup alternates)?  Variants, all 256 workgroups x 1024 threads, 16-byte accesses:
//   read            R loads per lane per step, nothing written
//   write           W stores per lane per step
//   mix R:W         every lane loads R and stores W pieces per step (a copy with ratio R:W)
//   split R:W       workgroups are readers or writers (CUs split by ratio), same total bytes
// Run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/probe_rw_mix.hip -o tools/bin/probe_rw_mix && tools/bin/probe_rw_mix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

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
constexpr int T = 1024;

// every workgroup owns a contiguous chunk of src (R pieces per lane and step) and of dst (W pieces per lane and step)
template <int R, int W, bool NT_STORE>
__global__ __launch_bounds__(T) void mix_kernel(const f64x2* __restrict__ src, f64x2* __restrict__ dst, size_t steps, double* __restrict__ out)
{
    const size_t rchunk = steps * (size_t)T * R, wchunk = steps * (size_t)T * W;
    const f64x2* s = src + blockIdx.x * rchunk;
    f64x2*       d = dst + blockIdx.x * wchunk;
    f64x2        acc{0.0, 0.0};
    f64x2        v[2][R > 0 ? R : 1];
    if constexpr (R > 0)
    {
#pragma unroll
        for (int k = 0; k < R; ++k) v[0][k] = __builtin_nontemporal_load(s + (size_t)k * T + threadIdx.x);
    }
    for (size_t i = 0; i < steps; i += 2)
    {
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
            const size_t j = i + h;
            if constexpr (R > 0)
            {
                const size_t jn = j + 1 < steps ? j + 1 : j;
#pragma unroll
                for (int k = 0; k < R; ++k) v[h ^ 1][k] = __builtin_nontemporal_load(s + (jn * R + k) * T + threadIdx.x);
#pragma unroll
                for (int k = 0; k < R; ++k) acc += v[h][k];
            }
            if constexpr (W > 0)
            {
#pragma unroll
                for (int k = 0; k < W; ++k)
                {
                    f64x2 o = acc;
                    o.x += (double)k;
                    if constexpr (NT_STORE)
                        __builtin_nontemporal_store(o, d + (j * W + k) * T + threadIdx.x);
                    else
                        d[(j * W + k) * T + threadIdx.x] = o;
                }
            }
        }
    }
    if (acc.x == 123.456) out[0] = acc.y;
}

// readers and writers are different workgroups: blocks [0, nread) read, the rest write
template <bool NT_STORE>
__global__ __launch_bounds__(T) void split_kernel(const f64x2* __restrict__ src, f64x2* __restrict__ dst, int nread, size_t rsteps, size_t wsteps,
                                                  double* __restrict__ out)
{
    constexpr int U = 6;
    if ((int)blockIdx.x < nread)
    {
        const f64x2* s = src + blockIdx.x * rsteps * (size_t)T * U;
        f64x2        acc{0.0, 0.0};
        for (size_t i = 0; i < rsteps; ++i)
        {
            f64x2 v[U];
#pragma unroll
            for (int k = 0; k < U; ++k) v[k] = __builtin_nontemporal_load(s + (i * U + k) * T + threadIdx.x);
#pragma unroll
            for (int k = 0; k < U; ++k) acc += v[k];
        }
        if (acc.x == 123.456) out[0] = acc.y;
    }
    else
    {
        f64x2* d = dst + (blockIdx.x - nread) * wsteps * (size_t)T * U;
        for (size_t i = 0; i < wsteps; ++i)
        {
#pragma unroll
            for (int k = 0; k < U; ++k)
            {
                f64x2 o{(double)i, (double)k};
                if constexpr (NT_STORE)
                    __builtin_nontemporal_store(o, d + (i * U + k) * T + threadIdx.x);
                else
                    d[(i * U + k) * T + threadIdx.x] = o;
            }
        }
    }
}

int main(int argc, char** argv)
{
    const size_t gb    = argc > 1 ? (size_t)atoll(argv[1]) : 4;
    const size_t bytes = gb << 30;
    f64x2 *      src, *dst;
    double*      out;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&dst, bytes));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(src, 0, bytes));
    CK(hipMemset(dst, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t pieces = bytes / 16;
    auto         time   = [&](auto launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep)
        {
            launch();
            CK(hipEventRecord(e0));
            for (int k = 0; k < 3; ++k) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms / 3 < best ? ms / 3 : best;
        }
        CK(hipGetLastError());
        return best;
    };
    printf("%zu GB buffers, 256 workgroups x 1024 threads, 16-byte accesses, loads nontemporal\n", gb);
    printf("%-34s %9s %9s %9s %9s\n", "variant", "ms", "read TB/s", "write TB/s", "sum TB/s");
#define MIX(R, W, NT)                                                                                                                      \
    {                                                                                                                                      \
        const size_t per   = (size_t)256 * T * ((R) > (W) ? (R) : (W));                                                                     \
        const size_t steps = (pieces / per) & ~(size_t)1;                                                                                   \
        const float  ms    = time([&] { hipLaunchKernelGGL((mix_kernel<R, W, NT>), dim3(256), dim3(T), 0, 0, src, dst, steps, out); });    \
        const double rb = (double)steps * 256 * T * (R) * 16, wb = (double)steps * 256 * T * (W) * 16;                                      \
        char         name[64];                                                                                                             \
        snprintf(name, sizeof name, "mix %d:%d%s", R, W, (NT) ? " nt stores" : "");                                                         \
        printf("%-34s %9.4f %9.2f %9.2f %9.2f\n", name, ms, rb / ms / 1e9, wb / ms / 1e9, (rb + wb) / ms / 1e9);                            \
    }
    MIX(6, 0, false)
    MIX(0, 6, false)
    MIX(0, 6, true)
    MIX(3, 3, false)
    MIX(3, 3, true)
    MIX(5, 3, false)
    MIX(5, 3, true)
    MIX(6, 2, false)
    MIX(6, 2, true)
    MIX(7, 1, false)
    MIX(7, 1, true)
    for (int nread : {128, 160, 176, 192})
        for (int nt = 0; nt < 2; ++nt)
        {
            // total bytes in ratio 5:3 whatever the split of the workgroups
            const size_t unit   = (size_t)T * 6;
            const size_t rtotal = pieces / unit / 2, wtotal = rtotal * 3 / 5;  // steps in all
            const size_t rsteps = rtotal / nread, wsteps = wtotal / (256 - nread);
            const float  ms     = time([&] {
                if (nt)
                    hipLaunchKernelGGL((split_kernel<true>), dim3(256), dim3(T), 0, 0, src, dst, nread, rsteps, wsteps, out);
                else
                    hipLaunchKernelGGL((split_kernel<false>), dim3(256), dim3(T), 0, 0, src, dst, nread, rsteps, wsteps, out);
            });
            const double rb = (double)rsteps * nread * unit * 16, wb = (double)wsteps * (256 - nread) * unit * 16;
            char         name[64];
            snprintf(name, sizeof name, "split %d readers / %d writers%s", nread, 256 - nread, nt ? " nt" : "");
            printf("%-34s %9.4f %9.2f %9.2f %9.2f\n", name, ms, rb / ms / 1e9, wb / ms / 1e9, (rb + wb) / ms / 1e9);
        }
    return 0;
}
