// tools/probe_host_write_vram.hip - can the CPU store straight into device memory on this platform (large BAR), and what does
// it cost against the other ways of handing a small host vector to a kernel?  For spmv_apply_host (abi.hip): x of a small
// matrix (C1: 80 KB) has to reach device memory before the product can gather from it.
//   hipcc -O2 --offload-arch=gfx950 tools/probe_host_write_vram.hip -o tools/bin/probe_host_write_vram
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <vector>

static sigjmp_buf g_jmp;
static void       on_segv(int) { siglongjmp(g_jmp, 1); }
static double     now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void sum_kernel(const double* __restrict__ p, int n, double* __restrict__ out)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) *out = s;
}
__global__ void copy_kernel(double* __restrict__ dst, const double* __restrict__ src, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

int main()
{
    const int           n = 10000;  // C1's x
    std::vector<double> h((size_t)n);
    for (int i = 0; i < n; ++i) h[(size_t)i] = 1.0 + i * 1e-3;
    double want = 0.0;
    for (double v : h) want += v;
    double* out = nullptr;
    hipHostMalloc((void**)&out, 64, hipHostMallocMapped);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    struct Kind
    {
        const char* name;
        int         how;
    } kinds[] = {{"hipMalloc", 0}, {"hipExtMallocWithFlags(hipDeviceMallocFinegrained)", 1}, {"hipExtMallocWithFlags(hipDeviceMallocUncached)", 2},
                 {"hipMallocManaged + hipMemAdviseSetCoarseGrain-less (default)", 3}};
    for (const Kind& k : kinds)
    {
        double*    d = nullptr;
        hipError_t e = k.how == 0   ? hipMalloc((void**)&d, sizeof(double) * n)
                       : k.how == 1 ? hipExtMallocWithFlags((void**)&d, sizeof(double) * n, hipDeviceMallocFinegrained)
                       : k.how == 2 ? hipExtMallocWithFlags((void**)&d, sizeof(double) * n, hipDeviceMallocUncached)
                                    : hipMallocManaged((void**)&d, sizeof(double) * n);
        if (e != hipSuccess)
        {
            printf("%-70s allocation failed: %s\n", k.name, hipGetErrorString(e));
            (void)hipGetLastError();
            continue;
        }
        hipMemset(d, 0, sizeof(double) * n);
        hipDeviceSynchronize();
        if (sigsetjmp(g_jmp, 1))
        {
            printf("%-70s NOT host-writable (signal on the first store)\n", k.name);
            continue;  // (leaks d: the probe exits soon)
        }
        // three rounds of: the CPU stores new values into the allocation, a kernel sums what it sees there.  Round 0 follows a
        // hipMemset (zeros may sit in the L2s); rounds 1 and 2 follow a kernel that has just read the previous values.
        // flush: 0 = sfence only, 1 = sfence + a write to the HDP flush register (hipDeviceAttributeHdpMemFlushCntl)
        unsigned* hdp = nullptr;
        (void)hipDeviceGetAttribute((int*)&hdp, hipDeviceAttributeHdpMemFlushCntl, 0);
        bool ok = true;
        for (int flush = 0; flush < 2; ++flush)
            for (int round = 0; round < 3; ++round)
            {
                for (int i = 0; i < n; ++i) h[(size_t)i] = 1.0 + i * 1e-3 + round + 10 * flush;
                double w = 0.0;
                for (double v : h) w += v;
                memcpy(d, h.data(), sizeof(double) * n);  // the CPU stores into the allocation
                __sync_synchronize();
                if (flush && hdp) *(volatile unsigned*)hdp = 1u;
                *out = -1.0;
                hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(64), 0, st, d, n, out);
                hipStreamSynchronize(st);
                const bool good = fabs(*out - w) <= 1e-9 * fabs(w);
                printf("    %s flush %d round %d: kernel sum %.6f, expected %.6f %s\n", k.name, flush, round, *out, w, good ? "ok" : "STALE / WRONG");
                ok = ok && good;
            }
        double     t0 = now_us();
        const int  reps = 200;
        for (int r = 0; r < reps; ++r)
        {
            h[0] += 1.0;
            memcpy(d, h.data(), sizeof(double) * n);
        }
        __sync_synchronize();
        const double us = (now_us() - t0) / reps;
        printf("%-70s host-writable, the kernel saw %s, CPU memcpy of %d KB into it: %.2f us\n", k.name, ok ? "the stores" : "SOMETHING ELSE", n * 8 / 1024, us);
        hipFree(d);
    }
    // the alternatives
    double *dx = nullptr, *pin = nullptr, *pin_dev = nullptr;
    hipMalloc((void**)&dx, sizeof(double) * n);
    hipHostMalloc((void**)&pin, sizeof(double) * n, hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&pin_dev, pin, 0);
    const int reps = 200;
    double    t0   = now_us();
    for (int r = 0; r < reps; ++r) hipMemcpy(dx, h.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    printf("hipMemcpy H2D from pageable memory, %d KB: %.2f us per call\n", n * 8 / 1024, (now_us() - t0) / reps);
    t0 = now_us();
    for (int r = 0; r < reps; ++r)
    {
        memcpy(pin, h.data(), sizeof(double) * n);
        hipLaunchKernelGGL(copy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dx, pin_dev, n);
        while (hipStreamQuery(st) == hipErrorNotReady) {}
    }
    printf("memcpy into pinned memory + copy kernel + poll, %d KB: %.2f us per call\n", n * 8 / 1024, (now_us() - t0) / reps);
    t0 = now_us();
    for (int r = 0; r < reps; ++r)
    {
        hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(64), 0, st, dx, pin_dev, 64);
        while (hipStreamQuery(st) == hipErrorNotReady) {}
    }
    printf("one tiny kernel + poll (launch + completion latency): %.2f us per call\n", (now_us() - t0) / reps);
    t0 = now_us();
    for (int r = 0; r < reps; ++r)
    {
        hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(64), 0, st, dx, pin_dev, 64);
        hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(64), 0, st, dx + 64, pin_dev, 64);
        while (hipStreamQuery(st) == hipErrorNotReady) {}
    }
    printf("two tiny kernels back to back + poll: %.2f us per call\n", (now_us() - t0) / reps);
    t0 = now_us();
    for (int r = 0; r < reps; ++r)
    {
        hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(64), 0, st, dx, pin_dev, 64);
        hipStreamSynchronize(st);
    }
    printf("one tiny kernel + hipStreamSynchronize: %.2f us per call\n", (now_us() - t0) / reps);
    return 0;
}
