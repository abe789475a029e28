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

    // Device copy of a host container, keyed by its `values` pointer and checked against a fingerprint of the
    // container (dimensions, array addresses, a sample of the array contents: fingerprint() in compat.cpp).  The
    // reference's products read the public host arrays on every call; a cached copy must not outlive an edit, a
    // re-pointed `values` or an address that was freed and handed out again: on a mismatch the copy is rebuilt.
    // `owner` = address of the container object, so that spmv_compat_invalidate(&A) works as well as (A.values).
    template <class Make>
    spmv_mat* cached(const void* key, int device, uint64_t fingerprint, const void* owner, Make make)
    {
        auto it = cache_.find({key, device});
        std::vector<unsigned char> plan;  // what the stale copy had decided (kernel, layout, tuned parameters: spmv_mat_get_plan)
        if (it != cache_.end())
        {
            if (it->second.fingerprint == fingerprint) return it->second.mat;
            // stale: the host arrays changed under the key.  The container is the same object with edited contents: its
            // replacement is built under the old copy's PLAN (no timing launches; the same kernel as before the edit) - a plan
            // that no longer fits the edited matrix is dropped by the engine, which then selects afresh
            int64_t len = 0;
            if (spmv_mat_get_plan(it->second.mat, nullptr, &len) == SPMV_OK && len > 0)
            {
                plan.resize((size_t)len);
                if (spmv_mat_get_plan(it->second.mat, plan.data(), &len) != SPMV_OK) plan.clear();
            }
            spmv_mat_destroy(it->second.mat);
            cache_.erase(it);
        }
        spmv_ctx* c = ctx(device);
        if (!plan.empty()) (void)spmv_ctx_set_plan(c, plan.data(), (int64_t)plan.size());
        spmv_mat* m = make(c);
        if (!plan.empty()) (void)spmv_ctx_set_plan(c, nullptr, 0);
        cache_[{key, device}] = Entry{m, fingerprint, owner};
        return m;
    }
    // conversions hand their result to the cache (fingerprint 0 = not known yet: taken on the first product)
    void adopt(const void* key, int device, spmv_mat* m, uint64_t fingerprint, const void* owner);
    void invalidate(const void* key_or_owner);  // all devices; matches the `values` pointer or the container address

    // y += A*x with host vectors: upload x and y, apply, download y (synchronous)
    void apply_host(int device, const spmv_mat* A, const double* x, int64_t nx, double* y, int64_t ny);

    spmv_vec* pooled(int device, int slot, int64_t n);  // a device vector of exactly n entries, reused across calls

    ~Engine();

private:
    Engine() = default;
    std::vector<spmv_ctx*>                               ctxs_;
    struct Entry
    {
        spmv_mat*   mat;
        uint64_t    fingerprint;
        const void* owner;
    };
    std::map<std::pair<const void*, int>, Entry>         cache_;
    std::map<std::tuple<int, int, int64_t>, spmv_vec*>   pool_;
    int                                                  ngpus_ = -1;
};
}  // namespace armspmv
