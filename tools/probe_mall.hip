// tools/probe_mall.hip — what the 256 MiB Infinity Cache does for streams: read, write and write-then-read rates of
// a buffer of S bytes swept repeatedly (16 B per lane, grid-stride), S from 32 MiB to 2 GiB.  Answers, for the
// design of a two-phase product (expand x into the entry order, then reduce): do written lines stay on-die for
// a reader that follows shortly, and at what rate are they served?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using f64x2 = double __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(256) void rd(const f64x2* __restrict__ p, size_t n, double* __restrict__ out)
{
    f64x2 acc = {0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc += NT ? __builtin_nontemporal_load(p + i) : p[i];
    if (acc.x + acc.y == 123.456) out[0] = acc.x;
}
template <bool NT>
__global__ __launch_bounds__(256) void wr(f64x2* __restrict__ p, size_t n, double v)
{
    const f64x2 val = {v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    {
        if (NT) __builtin_nontemporal_store(val, p + i); else p[i] = val;
    }
}
int main()
{
    const size_t maxb = (size_t)2048 << 20;
    f64x2* buf; double* out;
    CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&out, 64)); CK(hipMemset(buf, 0, maxb));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grid = 256 * 8, reps = 10;
    printf("%8s %12s %12s %12s %12s %14s %14s\n", "MiB", "read GB/s", "read nt", "write GB/s", "write nt", "wr->rd: wr", "wr->rd: rd");
    for (size_t mb : {32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048})
    {
        const size_t bytes = mb << 20, n = bytes / 16;
        float ms; double r[6];
        for (int mode = 0; mode < 4; ++mode)
        {
            for (int i = 0; i < reps + 2; ++i)
            {
                if (i == 2) CK(hipEventRecord(a));
                if (mode == 0) hipLaunchKernelGGL(rd<false>, dim3(grid), dim3(256), 0, 0, buf, n, out);
                if (mode == 1) hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(256), 0, 0, buf, n, out);
                if (mode == 2) hipLaunchKernelGGL(wr<false>, dim3(grid), dim3(256), 0, 0, buf, n, 1.0);
                if (mode == 3) hipLaunchKernelGGL(wr<true>, dim3(grid), dim3(256), 0, 0, buf, n, 1.0);
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            r[mode] = (double)bytes * reps / ms / 1e6;
        }
        // alternate write and read of the same buffer; time the writes and the reads apart
        float tw = 0.f, tr = 0.f;
        hipEvent_t c; CK(hipEventCreate(&c));
        for (int i = 0; i < reps + 1; ++i)
        {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(wr<false>, dim3(grid), dim3(256), 0, 0, buf, n, (double)i);
            CK(hipEventRecord(b));
            hipLaunchKernelGGL(rd<true>, dim3(grid), dim3(256), 0, 0, buf, n, out);
            CK(hipEventRecord(c)); CK(hipEventSynchronize(c));
            float x, y; CK(hipEventElapsedTime(&x, a, b)); CK(hipEventElapsedTime(&y, b, c));
            if (i) { tw += x; tr += y; }
        }
        r[4] = (double)bytes * reps / tw / 1e6; r[5] = (double)bytes * reps / tr / 1e6;
        printf("%8zu %12.0f %12.0f %12.0f %12.0f %14.0f %14.0f\n", mb, r[0], r[1], r[2], r[3], r[4], r[5]);
    }
    return 0;
}
