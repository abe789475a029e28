// engine.hpp — process-wide glue between the reference-shaped C++ API (include/arm_spmv_compat.hpp) and the
// C ABI (include/spmv_abi.h).  Host-side plumbing only: contexts per GPU, a cache that remembers which
// host container already lives on which GPU, and pooled device vectors for the x / y hand-over.
#pragma once

#include <cstdint>
#include <map>
#include <vector>

#include "spmv_abi.h"

namespace armspmv
{
[[noreturn]] void die(const char* where);  // prints spmv_last_error() and exit(1)s, like src/data_io.cpp:53-75
inline void       check(int rc, const char* where)
{
    if (rc != SPMV_OK) die(where);
}

class Engine
{
public:
    static Engine& get();

    int       ngpus();
    spmv_ctx* ctx(int device);  // created on first use

    // device copy of a host container, keyed by its `values` pointer; `make` uploads on a miss
    template <class Make>
    spmv_mat* cached(const void* key, int device, Make make)
    {
        auto it = cache_.find({key, device});
        if (it != cache_.end()) return it->second;
        spmv_mat* m = make(ctx(device));
        cache_[{key, device}] = m;
        return m;
    }
    void adopt(const void* key, int device, spmv_mat* m);  // conversions hand their result to the cache
    void invalidate(const void* key);                     // all devices

    // y += A*x with host vectors: upload x and y, apply, download y (synchronous)
    void apply_host(int device, const spmv_mat* A, const double* x, int64_t nx, double* y, int64_t ny);

    spmv_vec* pooled(int device, int slot, int64_t n);  // a device vector of exactly n entries, reused across calls

    ~Engine();

private:
    Engine() = default;
    std::vector<spmv_ctx*>                               ctxs_;
    std::map<std::pair<const void*, int>, spmv_mat*>     cache_;
    std::map<std::tuple<int, int, int64_t>, spmv_vec*>   pool_;
    int                                                  ngpus_ = -1;
};
}  // namespace armspmv
