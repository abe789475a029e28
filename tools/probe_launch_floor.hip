// tools/probe_launch_floor.hip - what does ONE small product cost to hand to the GPU and get back, by which mechanism?
// For the reference's call shape (CSRMatrixMatVector(A, x, y) with host vectors, main.cpp:56-59): C1's kernel takes 3 us,
// a call through spmv_apply_host 26 us (round 5).  This probe measures the mechanisms, with a kernel that does nothing, so that
// the engine builds on the one that pays:
//   (a) launch + hipStreamSynchronize                      (b) launch + hipStreamQuery spin (what spmv_apply_host does)
//   (c) launch, the kernel writes a flag in pinned host memory, the host spins on the flag (no HIP call after the launch)
//   (d) hipGraphLaunch of the captured kernel + flag        (e) a PERSISTENT kernel fed through a mailbox: the host bumps a
//   sequence number in pinned host memory, one lane polls it, the workgroups are released through a device-memory word,
//   "work", a grid-wide count-down, the last workgroup writes the acknowledgement to pinned host memory; round trip on the host.
// Every spinning loop on the device has two exits every wave reaches: a stop word and a wall-clock limit.
//   hipcc -O2 --offload-arch=gfx950 tools/probe_launch_floor.hip -o tools/bin/probe_launch_floor
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                                                  \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
        {                                                                                      \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);          \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

__global__ void null_kernel() {}
__global__ void flag_kernel(volatile unsigned long long* host_flag, unsigned long long v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
    {
        __threadfence_system();
        *host_flag = v;
    }
}

struct mailbox  // pinned host memory, one line each way
{
    volatile unsigned long long seq;  // host -> device: bumped per request; ~0ull = stop
    unsigned long long          pad0[15];
    volatile unsigned long long ack;  // device -> host
    unsigned long long          pad1[15];
};
struct devwords  // device memory
{
    unsigned long long go;    // the request the workgroups are released for
    unsigned long long left;  // workgroups still working on it
    unsigned long long exit_;
};

// wall_clock64(): the 100 MHz constant clock
__global__ __launch_bounds__(256) void persistent_kernel(mailbox* mb, devwords* dw, double* work, int work_n, unsigned long long idle_limit_ticks,
                                                        unsigned long long hard_limit_ticks)
{
    const unsigned long long t_start = wall_clock64();
    unsigned long long       last_active = t_start, seen = 0;
    __shared__ unsigned long long s_go;
    for (;;)
    {
        // ---- wait for a request: workgroup 0 polls the host's word, everybody else the device word it sets
        if (threadIdx.x == 0)
        {
            unsigned long long got = seen;
            for (;;)
            {
                const unsigned long long now = wall_clock64();
                if (now - last_active > idle_limit_ticks || now - t_start > hard_limit_ticks)
                {
                    got = ~0ull;
                    break;
                }
                if (blockIdx.x == 0)
                {
                    const unsigned long long s = __hip_atomic_load((unsigned long long*)&mb->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (s != seen)
                    {
                        got = s;
                        __hip_atomic_store(&dw->left, (unsigned long long)gridDim.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&dw->go, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                else
                {
                    const unsigned long long g = __hip_atomic_load(&dw->go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (g != seen)
                    {
                        got = g;
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (got == ~0ull && blockIdx.x == 0) __hip_atomic_store(&dw->go, ~0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // stop / timeout: release the others
            s_go = got;
        }
        __syncthreads();
        const unsigned long long req = s_go;
        __syncthreads();
        if (req == ~0ull) return;
        seen = req;
        // ---- "work": touch a little device memory, like a 3 us product would
        for (int i = blockIdx.x * 256 + threadIdx.x; i < work_n; i += gridDim.x * 256) work[i] += 1.0;
        __syncthreads();
        // ---- count down; the last workgroup acknowledges to the host
        if (threadIdx.x == 0)
        {
            __threadfence();
            if (__hip_atomic_fetch_add(&dw->left, ~0ull /* -1 */, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 1)
            {
                __threadfence_system();
                __hip_atomic_store((unsigned long long*)&mb->ack, req, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            last_active = wall_clock64();
        }
    }
}

static void report(const char* what, std::vector<double>& us)
{
    std::sort(us.begin(), us.end());
    printf("%-92s median %6.2f us   min %6.2f   p90 %6.2f\n", what, us[us.size() / 2], us[0], us[us.size() * 9 / 10]);
}

int main()
{
    constexpr int kCalls = 2000;
    hipStream_t   st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    volatile unsigned long long* flag = nullptr;
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped));
    unsigned long long* dflag = nullptr;
    CK(hipHostGetDevicePointer((void**)&dflag, (void*)flag, 0));
    *flag = 0;
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(null_kernel, dim3(313), dim3(256), 0, st);
    CK(hipStreamSynchronize(st));
    std::vector<double> us(kCalls);
    for (int i = 0; i < kCalls; ++i)
    {
        const double t0 = now_us();
        hipLaunchKernelGGL(null_kernel, dim3(313), dim3(256), 0, st);
        CK(hipStreamSynchronize(st));
        us[i] = now_us() - t0;
    }
    report("(a) launch of an empty kernel (313 workgroups) + hipStreamSynchronize", us);
    for (int i = 0; i < kCalls; ++i)
    {
        const double t0 = now_us();
        hipLaunchKernelGGL(null_kernel, dim3(313), dim3(256), 0, st);
        while (hipStreamQuery(st) == hipErrorNotReady) {}
        us[i] = now_us() - t0;
    }
    report("(b) launch + hipStreamQuery spin (spmv_apply_host today)", us);
    for (int i = 0; i < kCalls; ++i)
    {
        const unsigned long long v = (unsigned long long)i + 1;
        const double             t0 = now_us();
        hipLaunchKernelGGL(flag_kernel, dim3(313), dim3(256), 0, st, dflag, v);
        while (*flag != v) {}
        us[i] = now_us() - t0;
    }
    CK(hipStreamSynchronize(st));
    report("(c) launch, the kernel writes a word of pinned host memory, the host spins on it", us);
    {
        hipGraph_t     g;
        hipGraphExec_t ge;
        static unsigned long long gv = 0;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(flag_kernel, dim3(313), dim3(256), 0, st, dflag, 0xabcdefull);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < kCalls; ++i)
        {
            *flag           = 0;
            const double t0 = now_us();
            CK(hipGraphLaunch(ge, st));
            while (*flag != 0xabcdefull) {}
            us[i] = now_us() - t0;
        }
        (void)gv;
        CK(hipStreamSynchronize(st));
        report("(d) hipGraphLaunch of the captured kernel, host spins on the word it writes", us);
        (void)hipGraphExecDestroy(ge);
        (void)hipGraphDestroy(g);
    }
    // ---- (e) persistent kernel
    mailbox* mb = nullptr;
    CK(hipHostMalloc((void**)&mb, sizeof(mailbox), hipHostMallocMapped));
    mailbox* dmb = nullptr;
    CK(hipHostGetDevicePointer((void**)&dmb, mb, 0));
    devwords* dw = nullptr;
    CK(hipMalloc((void**)&dw, sizeof(devwords)));
    double* work = nullptr;
    const int work_n = 20000;  // C1: x and y
    CK(hipMalloc((void**)&work, sizeof(double) * work_n));
    CK(hipMemset(work, 0, sizeof(double) * work_n));
    for (int grid : {1, 64, 256})
    {
        memset((void*)mb, 0, sizeof(mailbox));
        CK(hipMemset(dw, 0, sizeof(devwords)));
        CK(hipDeviceSynchronize());
        // idle limit 20 ms here (the probe sleeps between nothing), hard limit 5 s: the kernel ends by itself whatever the host does
        hipLaunchKernelGGL(persistent_kernel, dim3(grid), dim3(256), 0, st, dmb, dw, work, work_n, 2000000ull, 500000000ull);
        CK(hipGetLastError());
        bool ok = true;
        for (int i = 0; i < kCalls && ok; ++i)
        {
            const unsigned long long v  = (unsigned long long)i + 1;
            const double             t0 = now_us();
            __atomic_store_n(&mb->seq, v, __ATOMIC_RELEASE);
            while (__atomic_load_n(&mb->ack, __ATOMIC_ACQUIRE) != v)
                if (now_us() - t0 > 2e6)
                {
                    printf("    (e) grid %d: request %d was not acknowledged within 2 s\n", grid, i);
                    ok = false;
                    break;
                }
            us[i] = now_us() - t0;
        }
        __atomic_store_n(&mb->seq, ~0ull, __ATOMIC_RELEASE);  // stop
        CK(hipStreamSynchronize(st));
        char what[160];
        snprintf(what, sizeof(what), "(e) persistent kernel, %3d workgroups: host word -> release -> 160 KB touched -> count-down -> host word", grid);
        if (ok) report(what, us);
    }
    // the idle exit: start it, send nothing, it must be gone within its limit
    memset((void*)mb, 0, sizeof(mailbox));
    CK(hipMemset(dw, 0, sizeof(devwords)));
    CK(hipDeviceSynchronize());
    const double t0 = now_us();
    hipLaunchKernelGGL(persistent_kernel, dim3(256), dim3(256), 0, st, dmb, dw, work, work_n, 100000ull /* 1 ms */, 500000000ull);
    CK(hipStreamSynchronize(st));
    printf("idle exit: a persistent kernel with a 1 ms idle limit that is sent nothing ended by itself after %.2f ms\n", (now_us() - t0) / 1e3);
    return 0;
}
