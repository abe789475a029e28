// tools/probe_vm_remap.hip — after hipMemUnmap + hipMemMap of ANOTHER physical piece at the SAME virtual address, do kernels
// reach the new piece, or the old one through a stale translation?  (Round 4's placement search saw identical timings for
// every window re-mapped inside one reservation and a product that then disagreed with the search: this is the check.)
//   piece A mapped at V: kernel fills V with 1.0.  Unmap.  Piece B mapped at V: kernel fills V with 2.0.
//   A mapped at another address W: what does it hold?  1.0 = correct; 2.0 = the second kernel wrote through a stale translation.
// Also aliasing: the same piece mapped at two addresses at once.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_vm_remap.hip -o tools/bin/probe_vm_remap && tools/bin/probe_vm_remap
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                \
    do                                                                       \
    {                                                                        \
        hipError_t e = (x);                                                  \
        if (e != hipSuccess)                                                 \
        {                                                                    \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); \
            exit(1);                                                         \
        }                                                                    \
    } while (0)

__global__ void fill(double* p, size_t n, double v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void count(const double* p, size_t n, double v, unsigned long long* out)
{
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] == v;
    atomicAdd(out, c);
}

int main(int argc, char** argv)
{
    const size_t piece = (argc > 1 ? (size_t)atoll(argv[1]) : 256) << 20;
    hipMemAllocationProp prop{};
    prop.type          = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id   = 0;
    size_t gran        = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu KB, piece %zu MB\n", gran >> 10, piece >> 20);
    hipMemGenericAllocationHandle_t A, B;
    CK(hipMemCreate(&A, piece, &prop, 0));
    CK(hipMemCreate(&B, piece, &prop, 0));
    void *V = nullptr, *W = nullptr, *W2 = nullptr;
    CK(hipMemAddressReserve(&V, piece, (size_t)1 << 30, nullptr, 0));
    CK(hipMemAddressReserve(&W, piece, (size_t)1 << 30, nullptr, 0));
    CK(hipMemAddressReserve(&W2, piece, (size_t)1 << 30, nullptr, 0));
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags    = hipMemAccessFlagsProtReadWrite;
    const size_t        n = piece / 8;
    unsigned long long* d_cnt;
    CK(hipMalloc(&d_cnt, 8));
    auto how_many = [&](void* p, double v) {
        CK(hipMemset(d_cnt, 0, 8));
        hipLaunchKernelGGL(count, dim3(1024), dim3(256), 0, 0, (const double*)p, n, v, d_cnt);
        unsigned long long h = 0;
        CK(hipMemcpy(&h, d_cnt, 8, hipMemcpyDeviceToHost));
        return h;
    };
    for (int round = 0; round < 3; ++round)
    {
        CK(hipMemMap(V, piece, 0, A, 0));
        CK(hipMemSetAccess(V, piece, &acc, 1));
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (double*)V, n, 1.0);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(V, piece));
        CK(hipMemMap(V, piece, 0, B, 0));
        CK(hipMemSetAccess(V, piece, &acc, 1));
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (double*)V, n, 2.0);
        CK(hipDeviceSynchronize());
        const unsigned long long v2 = how_many(V, 2.0);
        CK(hipMemUnmap(V, piece));
        CK(hipMemMap(W, piece, 0, A, 0));
        CK(hipMemSetAccess(W, piece, &acc, 1));
        CK(hipMemMap(W2, piece, 0, B, 0));
        CK(hipMemSetAccess(W2, piece, &acc, 1));
        const unsigned long long a1 = how_many(W, 1.0), a2 = how_many(W, 2.0), b2 = how_many(W2, 2.0);
        printf("round %d: through V after the re-map: %llu of %zu entries read 2.0; piece A holds %llu x 1.0 and %llu x 2.0 (stale writes); piece B holds %llu x 2.0  -> %s\n",
               round, v2, n, a1, a2, b2, a2 == 0 && b2 == n ? "re-mapping reaches the new piece" : "STALE TRANSLATION");
        CK(hipMemUnmap(W, piece));
        CK(hipMemUnmap(W2, piece));
    }
    // the same with the reservation given back in between: reserve, map A, fill, unmap, FREE; reserve again, map B, fill
    for (int round = 0; round < 3; ++round)
    {
        void* R = nullptr;
        CK(hipMemAddressReserve(&R, piece, (size_t)1 << 30, nullptr, 0));
        CK(hipMemMap(R, piece, 0, A, 0));
        CK(hipMemSetAccess(R, piece, &acc, 1));
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (double*)R, n, 3.0);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(R, piece));
        CK(hipMemAddressFree(R, piece));
        void* R2 = nullptr;
        CK(hipMemAddressReserve(&R2, piece, (size_t)1 << 30, nullptr, 0));
        CK(hipMemMap(R2, piece, 0, B, 0));
        CK(hipMemSetAccess(R2, piece, &acc, 1));
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (double*)R2, n, 4.0);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(R2, piece));
        CK(hipMemAddressFree(R2, piece));
        CK(hipMemMap(W, piece, 0, A, 0));
        CK(hipMemSetAccess(W, piece, &acc, 1));
        CK(hipMemMap(W2, piece, 0, B, 0));
        CK(hipMemSetAccess(W2, piece, &acc, 1));
        printf("freed and reserved again (%p then %p%s): piece A holds %llu x 3.0 and %llu x 4.0; piece B holds %llu x 4.0\n", R, R2, R == R2 ? ", the same address" : "",
               how_many(W, 3.0), how_many(W, 4.0), how_many(W2, 4.0));
        CK(hipMemUnmap(W, piece));
        CK(hipMemUnmap(W2, piece));
    }
    // and when the first piece is RELEASED before its address is used again: reserve, map C, fill, unmap, release C, free;
    // reserve again (same address), map D, fill: does D hold the data?
    for (int round = 0; round < 3; ++round)
    {
        hipMemGenericAllocationHandle_t C, D;
        CK(hipMemCreate(&C, piece, &prop, 0));
        void* R = nullptr;
        CK(hipMemAddressReserve(&R, piece, (size_t)1 << 30, nullptr, 0));
        CK(hipMemMap(R, piece, 0, C, 0));
        CK(hipMemSetAccess(R, piece, &acc, 1));
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (double*)R, n, 6.0);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(R, piece));
        CK(hipMemRelease(C));
        CK(hipMemAddressFree(R, piece));
        CK(hipMemCreate(&D, piece, &prop, 0));
        void* R2 = nullptr;
        CK(hipMemAddressReserve(&R2, piece, (size_t)1 << 30, nullptr, 0));
        CK(hipMemMap(R2, piece, 0, D, 0));
        CK(hipMemSetAccess(R2, piece, &acc, 1));
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (double*)R2, n, 7.0 + round);
        CK(hipDeviceSynchronize());
        const unsigned long long via = how_many(R2, 7.0 + round);
        CK(hipMemUnmap(R2, piece));
        void* R3 = nullptr;  // a never-used address for reading D back
        CK(hipMemAddressReserve(&R3, piece, (size_t)1 << 30, nullptr, 0));
        CK(hipMemMap(R3, piece, 0, D, 0));
        CK(hipMemSetAccess(R3, piece, &acc, 1));
        printf("first piece released before its address (%p, then %p) is used again: through the address %llu, piece D itself holds %llu of %zu\n", R, R2, via,
               how_many(R3, 7.0 + round), n);
        CK(hipMemUnmap(R3, piece));
        CK(hipMemRelease(D));
        CK(hipMemAddressFree(R2, piece));
        // R3 is kept reserved: never used again
    }
    // a piece moved between two NEVER-USED addresses: does the second address reach it?
    for (int round = 0; round < 2; ++round)
    {
        hipMemGenericAllocationHandle_t E;
        CK(hipMemCreate(&E, piece, &prop, 0));
        void *X1 = nullptr, *X2 = nullptr;
        CK(hipMemAddressReserve(&X1, piece, (size_t)1 << 30, nullptr, 0));
        CK(hipMemAddressReserve(&X2, piece, (size_t)1 << 30, nullptr, 0));
        CK(hipMemMap(X1, piece, 0, E, 0));
        CK(hipMemSetAccess(X1, piece, &acc, 1));
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (double*)X1, n, 8.0);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(X1, piece));
        CK(hipMemMap(X2, piece, 0, E, 0));
        CK(hipMemSetAccess(X2, piece, &acc, 1));
        const unsigned long long got = how_many(X2, 8.0);
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (double*)X2, n, 9.0);
        CK(hipDeviceSynchronize());
        printf("a piece moved from one never-used address (%p) to another (%p): the second reads %llu x 8.0 of %zu, then %llu x 9.0 after a fill\n", X1, X2, got, n,
               how_many(X2, 9.0));
        CK(hipMemUnmap(X2, piece));
        CK(hipMemRelease(E));  // X1, X2 stay reserved
    }
    // aliasing: A at W and at W2 at once
    hipError_t e = hipMemMap(W, piece, 0, A, 0);
    if (e == hipSuccess) e = hipMemSetAccess(W, piece, &acc, 1);
    hipError_t e2 = e == hipSuccess ? hipMemMap(W2, piece, 0, A, 0) : e;
    if (e2 == hipSuccess) e2 = hipMemSetAccess(W2, piece, &acc, 1);
    if (e2 == hipSuccess)
    {
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, (double*)W, n, 5.0);
        CK(hipDeviceSynchronize());
        printf("aliasing: one piece at two addresses: the second address reads %llu x 5.0 of %zu\n", how_many(W2, 5.0), n);
    }
    else
        printf("aliasing refused: %s\n", hipGetErrorString(e2));
    return 0;
}
