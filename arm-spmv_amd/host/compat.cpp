// compat.cpp — implementation of include/arm_spmv_compat.hpp over the C ABI (include/spmv_abi.h).
//
// What runs where:
//   * products (y += A*x), BLAS-1, COO->CSR and COO->ELL conversion:  GPU, through libspmv_hip.so
//   * container bookkeeping (new[]/delete[], deep copies), COO->CSC and CSR->DIA conversion (formats that
//     SURVEY.md 8f ranks "next"), sharding arithmetic of the Numa drivers:  host, in this file
// Citations are to the reference tree (src/..., include/...).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <tuple>
#include <vector>

#include "arm_spmv_compat.hpp"
#include "engine.hpp"

using armspmv::check;
using armspmv::Engine;

// =====================================================================================================
// Engine
// =====================================================================================================
namespace armspmv
{
void die(const char* where)
{
    printf("*** spmv engine error in %s: %s ***\n", where, spmv_last_error());
    fflush(stdout);
    exit(1);
}

Engine& Engine::get()
{
    static Engine e;
    return e;
}

int Engine::ngpus()
{
    if (ngpus_ < 0)
    {
        int n = 0;
        check(spmv_device_count(&n), "spmv_device_count");
        if (n <= 0)
        {
            printf("*** no HIP device visible: this build of the arm-spmv API has no CPU path ***\n");
            exit(1);
        }
        ngpus_ = n;
        ctxs_.assign((size_t)n, nullptr);
    }
    return ngpus_;
}

spmv_ctx* Engine::ctx(int device)
{
    ngpus();
    if (!ctxs_[(size_t)device]) check(spmv_ctx_create(device, &ctxs_[(size_t)device]), "spmv_ctx_create");
    return ctxs_[(size_t)device];
}

void Engine::adopt(const void* key, int device, spmv_mat* m, uint64_t fingerprint, const void* owner)
{
    auto it = cache_.find({key, device});
    if (it != cache_.end()) spmv_mat_destroy(it->second.mat);
    cache_[{key, device}] = Entry{m, fingerprint, owner};
}

void Engine::invalidate(const void* key)
{
    if (!key) return;
    for (auto it = cache_.begin(); it != cache_.end();)
    {
        if (it->first.first == key || it->second.owner == key)
        {
            spmv_mat_destroy(it->second.mat);
            it = cache_.erase(it);
        }
        else
            ++it;
    }
}

spmv_vec* Engine::pooled(int device, int slot, int64_t n)
{
    auto key = std::make_tuple(device, slot, n);
    auto it  = pool_.find(key);
    if (it != pool_.end()) return it->second;
    spmv_vec* v = nullptr;
    check(spmv_vec_create(ctx(device), n, &v), "spmv_vec_create");
    pool_[key] = v;
    return v;
}

void Engine::apply_host(int device, const spmv_mat* A, const double* x, int64_t nx, double* y, int64_t ny)
{
    spmv_mat_info info;
    check(spmv_mat_get_info(A, &info), "spmv_mat_get_info");
    if (info.ncol == nx && info.nrow == ny)
    {
        // the engine's own hand-over of host vectors: staged through pinned memory the GPU reads itself when they are small
        // (three launches, no hipMemcpy), asynchronous copies when they are large (spmv_apply_host, include/spmv_abi.h)
        check(spmv_apply_host(ctx(device), A, x, y), "spmv_apply_host");
        return;
    }
    // (vectors longer than the matrix needs - the reference never checks, its loops read what they index: src/mat_vec.cpp:46-52)
    spmv_vec* dx = pooled(device, 0, nx);
    spmv_vec* dy = pooled(device, 1, ny);
    check(spmv_vec_upload(dx, 0, nx, x), "spmv_vec_upload(x)");
    check(spmv_vec_upload(dy, 0, ny, y), "spmv_vec_upload(y)");
    check(spmv_apply(ctx(device), A, dx, dy), "spmv_apply");
    check(spmv_vec_download(dy, 0, ny, y), "spmv_vec_download(y)");
}

Engine::~Engine()
{
    for (auto& kv : cache_) spmv_mat_destroy(kv.second.mat);
    for (auto& kv : pool_) spmv_vec_destroy(kv.second);
    for (spmv_ctx* c : ctxs_) spmv_ctx_destroy(c);
}
}  // namespace armspmv

void spmv_compat_invalidate(const void* p) { Engine::get().invalidate(p); }

namespace
{
template <class T>
T* clone(const T* src, size_t n)
{
    T* dst = new T[n > 0 ? n : 1];
    if (n) std::memcpy(dst, src, n * sizeof(T));
    return dst;
}

template <class T>
void drop(T*& p)
{
    delete[] p;
    p = 0;
}

// Fingerprint of a host container for the device-copy cache (engine.hpp): dimensions, array addresses and up to
// 2048 (small arrays: 256) evenly spaced elements of every array (first and last included).  Cheap enough for every call (a few
// thousand reads next to the PCIe copies of x and y); catches a re-pointed or re-allocated container and edits that
// touch a sampled element.  An in-place edit that dodges every sample needs spmv_compat_invalidate().
inline uint64_t fp_mix(uint64_t h, uint64_t v)
{
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    return h * 0xBF58476D1CE4E5B9ull;
}
template <class T>
uint64_t fp_array(uint64_t h, const T* p, size_t n)
{
    h = fp_mix(h, (uint64_t)(uintptr_t)p);
    h = fp_mix(h, (uint64_t)n);
    if (!p || n == 0) return h;
    // (2048 samples of arrays with a million elements and more; 256 of smaller ones, where the hand-over of x and y is tens of
    // microseconds and three times 2048 scattered reads would be a fifth of the call)
    const size_t samples = std::min<size_t>(n, n >= ((size_t)1 << 20) ? 2048 : 256);
    for (size_t k = 0; k < samples; ++k)
    {
        const size_t i = samples > 1 ? (size_t)(((unsigned __int128)k * (n - 1)) / (samples - 1)) : 0;
        uint64_t     bits = 0;
        std::memcpy(&bits, p + i, sizeof(T));
        h = fp_mix(h, bits);
    }
    return h;
}
uint64_t fingerprint(const COOMatrix& A)
{
    uint64_t h = fp_mix(fp_mix(fp_mix(1, (uint64_t)A.nrow), (uint64_t)A.ncol), (uint64_t)A.nnz);
    const size_t n = A.nnz > 0 ? (size_t)A.nnz : 0;
    return fp_array(fp_array(fp_array(h, A.row_ind, n), A.col_ind, n), A.values, n);
}
uint64_t fingerprint(const CSRMatrix& A)
{
    uint64_t h = fp_mix(fp_mix(2, (uint64_t)A.nrow), (uint64_t)A.ncol);
    const size_t n = (A.row_ptr && A.nrow >= 0) ? (size_t)std::max(A.row_ptr[A.nrow], 0) : 0;
    return fp_array(fp_array(fp_array(h, A.row_ptr, A.row_ptr ? (size_t)A.nrow + 1 : 0), A.col_ind, n), A.values, n);
}
uint64_t fingerprint(const CSCMatrix& A)
{
    uint64_t h = fp_mix(fp_mix(3, (uint64_t)A.nrow), (uint64_t)A.ncol);
    const size_t n = (A.col_ptr && A.ncol >= 0) ? (size_t)std::max(A.col_ptr[A.ncol], 0) : 0;
    return fp_array(fp_array(fp_array(h, A.col_ptr, A.col_ptr ? (size_t)A.ncol + 1 : 0), A.row_ind, n), A.values, n);
}
uint64_t fingerprint(const ELLMatrix& A)
{
    uint64_t h = fp_mix(fp_mix(fp_mix(fp_mix(4, (uint64_t)A.nrow), (uint64_t)A.ncol), (uint64_t)A.nnz), (uint64_t)A.nonzeros_in_row);
    const size_t n = (size_t)std::max(A.nrow, 0) * (size_t)std::max(A.nonzeros_in_row, 0);
    return fp_array(fp_array(h, A.col_ind, n), A.values, n);
}
uint64_t fingerprint(const DIAMatrix& A)
{
    uint64_t h = fp_mix(fp_mix(fp_mix(5, (uint64_t)A.nrow), (uint64_t)A.ncol), (uint64_t)A.ndiags);
    const size_t n = (size_t)std::max(A.nrow, 0) * (size_t)std::max(A.ndiags, 0);
    return fp_array(fp_array(h, A.offsets, (size_t)std::max(A.ndiags, 0)), A.values, n);
}

// packed diagonal, entries in COO encounter order (src/matrix.cpp:146-153); never read by a product
double* pack_diagonal(const COOMatrix& A)
{
    double* d = new double[A.nrow > 0 ? A.nrow : 1];
    int     n = 0;
    for (int k = 0; k < A.nnz; ++k)
        if (A.row_ind[k] == A.col_ind[k] && n < A.nrow) d[n++] = A.values[k];
    return d;
}

spmv_mat* device_coo(const COOMatrix& A, int device = 0)
{
    return Engine::get().cached(A.values, device, fingerprint(A), &A, [&](spmv_ctx* c) {
        spmv_mat* m = nullptr;
        check(spmv_coo_upload(c, A.nrow, A.ncol, A.nnz, A.row_ind, A.col_ind, A.values, &m), "spmv_coo_upload");
        return m;
    });
}
}  // namespace

// =====================================================================================================
// Vector (include/vector.h:4-26, src/vector.cpp)
// =====================================================================================================
Vector::Vector() : size(0), values(0) {}
Vector::Vector(int n, double* v) : size(n), values(v) {}
Vector::Vector(const Vector& x) : size(x.size), values(clone(x.values, (size_t)x.size)) {}
Vector::~Vector() { delete[] values; }

Vector& Vector::operator=(double a)
{
    std::fill(values, values + size, a);
    return *this;
}

Vector& Vector::operator=(const Vector& x)
{
    if (this != &x)
    {
        Resize(x.size);
        std::copy(x.values, x.values + size, values);
    }
    return *this;
}

void Vector::Free()
{
    drop(values);
    size = 0;
}

void Vector::Resize(int n)
{
    delete[] values;
    size   = n;
    values = new double[n > 0 ? n : 1];
}

void Vector::Fill(double a) const { std::fill(values, values + size, a); }

void Vector::FillRandom() const
{
    for (double *p = values, *e = values + size; p != e; ++p) *p = (double)rand() / RAND_MAX;  // src/vector.cpp:65-69
}

void Vector::Copy(const Vector& x) const { std::copy(x.values, x.values + size, values); }

void Vector::Scale(double a) const
{
    for (int i = 0; i < size; ++i) values[i] *= a;
}

void Vector::Shift(double a) const
{
    for (int i = 0; i < size; ++i) values[i] += a;
}

void Vector::AddScaled(double a, const Vector& x) const
{
    for (int i = 0; i < size; ++i) values[i] += a * x.values[i];
}

void Vector::Add2Scaled(double a, const Vector& x, double b, const Vector& y) const
{
    for (int i = 0; i < size; ++i) values[i] += a * x.values[i] + b * y.values[i];
}

bool checkVector(const Vector& x, const Vector& y)
{
    if (x.size != y.size) return false;
    for (int i = 0; i < x.size; ++i)
        if (std::fabs(x.values[i] - y.values[i]) > 1e-6) return false;  // src/vector.cpp:161-171
    return true;
}

// =====================================================================================================
// COOMatrix (include/matrix.h:7-25)
// =====================================================================================================
COOMatrix::COOMatrix() : nrow(0), ncol(0), nnz(0), row_ind(0), col_ind(0), values(0) {}
COOMatrix::COOMatrix(int n, int m, int z, int* r, int* c, double* v) : nrow(n), ncol(m), nnz(z), row_ind(r), col_ind(c), values(v) {}
COOMatrix::COOMatrix(const COOMatrix& A)
    : nrow(A.nrow), ncol(A.ncol), nnz(A.nnz), row_ind(clone(A.row_ind, (size_t)A.nnz)), col_ind(clone(A.col_ind, (size_t)A.nnz)),
      values(clone(A.values, (size_t)A.nnz))
{
}
COOMatrix::~COOMatrix() { Free(); }

COOMatrix& COOMatrix::operator=(const COOMatrix& A)
{
    if (this == &A) return *this;
    Free();
    nrow    = A.nrow;
    ncol    = A.ncol;
    nnz     = A.nnz;
    row_ind = clone(A.row_ind, (size_t)nnz);
    col_ind = clone(A.col_ind, (size_t)nnz);
    values  = clone(A.values, (size_t)nnz);
    return *this;
}

void COOMatrix::Free()
{
    Engine::get().invalidate(values);
    drop(row_ind);
    drop(col_ind);
    drop(values);
    nrow = ncol = nnz = 0;
}

// =====================================================================================================
// CSRMatrix (include/matrix.h:27-47)
// =====================================================================================================
CSRMatrix::CSRMatrix() : nrow(0), ncol(0), row_ptr(0), col_ind(0), values(0), diagonal(0) {}
CSRMatrix::CSRMatrix(int n, int m, int* rp, int* ci, double* v, double* d) : nrow(n), ncol(m), row_ptr(rp), col_ind(ci), values(v), diagonal(d) {}

static void csr_copy_from(CSRMatrix& dst, const CSRMatrix& A)
{
    const size_t nnz = A.row_ptr ? (size_t)A.row_ptr[A.nrow] : 0;
    dst.nrow         = A.nrow;
    dst.ncol         = A.ncol;
    dst.row_ptr      = clone(A.row_ptr, (size_t)A.nrow + 1);
    dst.col_ind      = clone(A.col_ind, nnz);
    dst.values       = clone(A.values, nnz);
    dst.diagonal     = A.diagonal ? clone(A.diagonal, (size_t)A.nrow) : 0;
}

// COO -> CSR on the GPU: histogram, prefix sum, stable placement (spmv_coo_to_csr), then the arrays come back
// to the host because the reference exposes them as public fields.  The device copy stays cached for the products.
static void csr_from_coo(CSRMatrix& dst, const COOMatrix& A)
{
    Engine&   E   = Engine::get();
    spmv_mat* coo = device_coo(A);
    spmv_mat* csr = nullptr;
    check(spmv_coo_to_csr(E.ctx(0), coo, &csr), "spmv_coo_to_csr");
    dst.nrow     = A.nrow;
    dst.ncol     = A.ncol;
    dst.row_ptr  = new int[(size_t)A.nrow + 1];
    dst.col_ind  = new int[A.nnz > 0 ? A.nnz : 1];
    dst.values   = new double[A.nnz > 0 ? A.nnz : 1];
    dst.diagonal = pack_diagonal(A);
    check(spmv_mat_download(csr, dst.row_ptr, dst.col_ind, dst.values), "spmv_mat_download(csr)");
    E.adopt(dst.values, 0, csr, fingerprint(dst), &dst);
}

CSRMatrix::CSRMatrix(const CSRMatrix& A) { csr_copy_from(*this, A); }
CSRMatrix::CSRMatrix(const COOMatrix& A) { csr_from_coo(*this, A); }
CSRMatrix::~CSRMatrix() { Free(); }

CSRMatrix& CSRMatrix::operator=(const CSRMatrix& A)
{
    if (this == &A) return *this;
    Free();
    csr_copy_from(*this, A);
    return *this;
}

CSRMatrix& CSRMatrix::operator=(const COOMatrix& A)
{
    Free();
    csr_from_coo(*this, A);  // follows the constructor; the reference's operator= leaves row_ptr[nrow] unset (src/matrix.cpp:217-220)
    return *this;
}

void CSRMatrix::Free()
{
    Engine::get().invalidate(values);
    drop(row_ptr);
    drop(col_ind);
    drop(values);
    drop(diagonal);
    nrow = ncol = 0;
}

// =====================================================================================================
// CSCMatrix (include/matrix.h:49-68) — conversion on the host (format ranked "next")
// =====================================================================================================
CSCMatrix::CSCMatrix() : nrow(0), ncol(0), row_ind(0), col_ptr(0), values(0) {}
CSCMatrix::CSCMatrix(int n, int m, int* ri, int* cp, double* v) : nrow(n), ncol(m), row_ind(ri), col_ptr(cp), values(v) {}

static void csc_copy_from(CSCMatrix& dst, const CSCMatrix& A)
{
    const size_t nnz = A.col_ptr ? (size_t)A.col_ptr[A.ncol] : 0;
    dst.nrow         = A.nrow;
    dst.ncol         = A.ncol;
    dst.col_ptr      = clone(A.col_ptr, (size_t)A.ncol + 1);
    dst.row_ind      = clone(A.row_ind, nnz);
    dst.values       = clone(A.values, nnz);
}

// stable counting sort by column: same arrays as the reference's backward scatter (src/matrix.cpp:305-324)
static void csc_from_coo(CSCMatrix& dst, const COOMatrix& A)
{
    dst.nrow    = A.nrow;
    dst.ncol    = A.ncol;
    dst.col_ptr = new int[(size_t)A.ncol + 1]();
    dst.row_ind = new int[A.nnz > 0 ? A.nnz : 1];
    dst.values  = new double[A.nnz > 0 ? A.nnz : 1];
    for (int k = 0; k < A.nnz; ++k) dst.col_ptr[A.col_ind[k] + 1]++;
    for (int j = 0; j < A.ncol; ++j) dst.col_ptr[j + 1] += dst.col_ptr[j];
    std::vector<int> next(dst.col_ptr, dst.col_ptr + A.ncol);
    for (int k = 0; k < A.nnz; ++k)
    {
        const int at    = next[(size_t)A.col_ind[k]]++;
        dst.row_ind[at] = A.row_ind[k];
        dst.values[at]  = A.values[k];
    }
}

CSCMatrix::CSCMatrix(const CSCMatrix& A) { csc_copy_from(*this, A); }
CSCMatrix::CSCMatrix(const COOMatrix& A)
{
    csc_from_coo(*this, A);
    spmv_compat_prefetch(*this);  // (set-up belongs to the constructor, as with CSRMatrix(COO) and ELLMatrix(COO): main.cpp:74 is outside the timed loop)
}
CSCMatrix::~CSCMatrix() { Free(); }

CSCMatrix& CSCMatrix::operator=(const CSCMatrix& A)
{
    if (this == &A) return *this;
    Free();
    csc_copy_from(*this, A);
    return *this;
}

CSCMatrix& CSCMatrix::operator=(const COOMatrix& A)
{
    Free();
    csc_from_coo(*this, A);
    spmv_compat_prefetch(*this);
    return *this;
}

void CSCMatrix::Free()
{
    Engine::get().invalidate(values);
    drop(row_ind);
    drop(col_ptr);
    drop(values);
    nrow = ncol = 0;
}

// =====================================================================================================
// ELLMatrix (include/matrix.h:70-92)
// =====================================================================================================
ELLMatrix::ELLMatrix() : nrow(0), ncol(0), nnz(0), nonzeros_in_row(0), col_ind(0), values(0), diagonal(0) {}
ELLMatrix::ELLMatrix(int n, int m, int z, int k, int* ci, double* v, double* d)
    : nrow(n), ncol(m), nnz(z), nonzeros_in_row(k), col_ind(ci), values(v), diagonal(d)
{
}

static void ell_copy_from(ELLMatrix& dst, const ELLMatrix& A)
{
    const size_t total  = (size_t)A.nrow * (size_t)A.nonzeros_in_row;
    dst.nrow            = A.nrow;
    dst.ncol            = A.ncol;
    dst.nnz             = A.nnz;
    dst.nonzeros_in_row = A.nonzeros_in_row;
    dst.col_ind         = clone(A.col_ind, total);
    dst.values          = clone(A.values, total);
    dst.diagonal        = A.diagonal ? clone(A.diagonal, (size_t)A.nrow) : 0;
}

static void ell_from_coo(ELLMatrix& dst, const COOMatrix& A)
{
    Engine&   E   = Engine::get();
    spmv_mat* coo = device_coo(A);
    spmv_mat* ell = nullptr;
    check(spmv_coo_to_ell(E.ctx(0), coo, &ell), "spmv_coo_to_ell");
    spmv_mat_info info;
    check(spmv_mat_get_info(ell, &info), "spmv_mat_get_info");
    const size_t total  = (size_t)A.nrow * (size_t)info.ell_k;
    dst.nrow            = A.nrow;
    dst.ncol            = A.ncol;
    dst.nnz             = A.nnz;
    dst.nonzeros_in_row = info.ell_k;
    dst.col_ind         = new int[total > 0 ? total : 1];
    dst.values          = new double[total > 0 ? total : 1];
    dst.diagonal        = pack_diagonal(A);
    check(spmv_mat_download(ell, nullptr, dst.col_ind, dst.values), "spmv_mat_download(ell)");
    E.adopt(dst.values, 0, ell, fingerprint(dst), &dst);
}

ELLMatrix::ELLMatrix(const ELLMatrix& A) { ell_copy_from(*this, A); }
ELLMatrix::ELLMatrix(const COOMatrix& A) { ell_from_coo(*this, A); }
ELLMatrix::~ELLMatrix() { Free(); }

ELLMatrix& ELLMatrix::operator=(const ELLMatrix& A)
{
    if (this == &A) return *this;
    Free();
    ell_copy_from(*this, A);
    return *this;
}

ELLMatrix& ELLMatrix::operator=(const COOMatrix& A)
{
    Free();
    ell_from_coo(*this, A);
    return *this;
}

void ELLMatrix::Free()
{
    Engine::get().invalidate(values);
    drop(col_ind);
    drop(values);
    drop(diagonal);
    nrow = ncol = nnz = nonzeros_in_row = 0;
}

// =====================================================================================================
// DIAMatrix (include/matrix.h:117-138) — conversion on the host (format ranked "next")
// =====================================================================================================
DIAMatrix::DIAMatrix() : nnz(0), nrow(0), ncol(0), ndiags(0), offsets(0), values(0) {}
DIAMatrix::DIAMatrix(int n, int m, int nd, int* off, double* v) : nnz(0), nrow(n), ncol(m), ndiags(nd), offsets(off), values(v) {}

static void dia_copy_from(DIAMatrix& dst, const DIAMatrix& A)
{
    dst.nnz     = A.nnz;
    dst.nrow    = A.nrow;
    dst.ncol    = A.ncol;
    dst.ndiags  = A.ndiags;
    dst.offsets = clone(A.offsets, (size_t)A.ndiags);
    dst.values  = clone(A.values, (size_t)A.nrow * (size_t)A.ndiags);
}

// Diagonal discovery + row-major fill (src/matrix.cpp:673-726).  Diagonal id of entry (i,j) is j - i; the
// offsets come out ascending; a later duplicate of the same (i,j) overwrites the earlier one, as in the
// reference (:721).  The occupancy table has nrow+ncol slots (the reference's has one fewer and overruns
// for an entry at (0, ncol-1)).
static void dia_from_csr(DIAMatrix& dst, const CSRMatrix& A)
{
    const int         span = A.nrow + A.ncol;
    std::vector<int>  slot((size_t)span, -1);
    std::vector<char> used((size_t)span, 0);
    for (int i = 0; i < A.nrow; ++i)
        for (int jj = A.row_ptr[i]; jj < A.row_ptr[i + 1]; ++jj) used[(size_t)(A.col_ind[jj] - i + A.nrow)] = 1;
    int nd = 0;
    for (int s = 0; s < span; ++s)
        if (used[(size_t)s]) slot[(size_t)s] = nd++;
    dst.nnz     = A.row_ptr[A.nrow];
    dst.nrow    = A.nrow;
    dst.ncol    = A.ncol;
    dst.ndiags  = nd;
    dst.offsets = new int[nd > 0 ? nd : 1];
    dst.values  = new double[(size_t)A.nrow * (size_t)nd + 1]();
    for (int s = 0; s < span; ++s)
        if (used[(size_t)s]) dst.offsets[slot[(size_t)s]] = s - A.nrow;
    for (int i = 0; i < A.nrow; ++i)
        for (int jj = A.row_ptr[i]; jj < A.row_ptr[i + 1]; ++jj)
            dst.values[(size_t)i * nd + slot[(size_t)(A.col_ind[jj] - i + A.nrow)]] = A.values[jj];
}

DIAMatrix::DIAMatrix(const DIAMatrix& A) { dia_copy_from(*this, A); }
DIAMatrix::DIAMatrix(const CSRMatrix& A)
{
    dia_from_csr(*this, A);
    spmv_compat_prefetch(*this);
}
DIAMatrix::~DIAMatrix() { Free(); }

DIAMatrix& DIAMatrix::operator=(const DIAMatrix& A)
{
    if (this == &A) return *this;
    Free();
    dia_copy_from(*this, A);
    return *this;
}

DIAMatrix& DIAMatrix::operator=(const CSRMatrix& A)
{
    Free();
    dia_from_csr(*this, A);
    spmv_compat_prefetch(*this);
    return *this;
}

void DIAMatrix::Free()
{
    Engine::get().invalidate(values);
    drop(offsets);
    drop(values);
}

// =====================================================================================================
// y += A*x (include/mat_vec.h:7-11).  The matrix is uploaded on first use and cached; x and y cross PCIe.
// =====================================================================================================
void COOMatirxMatVector(const COOMatrix& A, const Vector& x, Vector& y)
{
    Engine::get().apply_host(0, device_coo(A), x.values, x.size, y.values, y.size);
}

static spmv_mat* device_csr(const CSRMatrix& A, int device = 0)
{
    return Engine::get().cached(A.values, device, fingerprint(A), &A, [&](spmv_ctx* c) {
        spmv_mat* m = nullptr;
        check(spmv_csr_upload(c, A.nrow, A.ncol, A.row_ptr, A.col_ind, A.values, &m), "spmv_csr_upload");
        return m;
    });
}

void CSRMatrixMatVector(const CSRMatrix& A, const Vector& x, Vector& y)
{
    Engine::get().apply_host(0, device_csr(A), x.values, x.size, y.values, y.size);
}

static spmv_mat* device_csc(const CSCMatrix& A)
{
    return Engine::get().cached(A.values, 0, fingerprint(A), &A, [&](spmv_ctx* c) {
        spmv_mat* m = nullptr;
        check(spmv_csc_upload(c, A.nrow, A.ncol, A.col_ptr, A.row_ind, A.values, &m), "spmv_csc_upload");
        return m;
    });
}

void CSCMatrixMatVector(const CSCMatrix& A, const Vector& x, Vector& y)
{
    Engine::get().apply_host(0, device_csc(A), x.values, x.size, y.values, y.size);
}

static spmv_mat* device_ell(const ELLMatrix& A)
{
    return Engine::get().cached(A.values, 0, fingerprint(A), &A, [&](spmv_ctx* c) {
        spmv_mat* m = nullptr;
        check(spmv_ell_upload(c, A.nrow, A.ncol, A.nonzeros_in_row, A.nnz, A.col_ind, A.values, &m), "spmv_ell_upload");
        return m;
    });
}

void ELLMatrixMatVector(const ELLMatrix& A, const Vector& x, Vector& y)
{
    Engine::get().apply_host(0, device_ell(A), x.values, x.size, y.values, y.size);
}

static spmv_mat* device_dia(const DIAMatrix& A)
{
    return Engine::get().cached(A.values, 0, fingerprint(A), &A, [&](spmv_ctx* c) {
        spmv_mat* m = nullptr;
        check(spmv_dia_upload(c, A.nrow, A.ncol, A.ndiags, A.offsets, A.values, &m), "spmv_dia_upload");
        return m;
    });
}

void DIAMatrixMatVector(const DIAMatrix& A, const Vector& x, Vector& y)
{
    Engine::get().apply_host(0, device_dia(A), x.values, x.size, y.values, y.size);
}

// =====================================================================================================
// "Numa" drivers (include/mat_vec.h:13-17, src/mat_vec.cpp:148-484): row-range shards, one per "thread",
// shard i on GPU i % ngpus (reference: NUMA node i % numanodes).  Same protocol as the reference:
//   build shards + a full x replica per device + zeroed local y   (outside the timed region, :240-268)
//   NTESTS x { launch every shard ; wait for all }                (timed, :270-282)
//   print "### <FMT> NUMA GFLOPS = %.5f"                          (:285)
// Afterwards the local y slices are copied into y (the reference only does that for DIA, :474-477).
// =====================================================================================================
namespace
{
int    g_numa_reps    = 50;  // NTESTS, src/mat_vec.cpp:201
double g_last_numa_ms = 0.0;
// How the sharded drivers cut the rows: 0 = equal rows, the last shard takes the remainder (the reference: src/mat_vec.cpp:151,
// :233,:245-246 - the default), 1 = by entries (spmv_partition_rows_balanced: every shard about the same number of stored
// entries; SURVEY.md 8e's option for skewed matrices).  -1 = not set by a call: the environment's SPMV_COMPAT_PARTITION
// ("nnz" / "entries" / "1" -> by entries, anything else -> rows), read at every driver call.
int    g_partition    = -1;
double g_last_slowest_shard_ms = 0.0, g_last_shard_imbalance = 0.0;

bool partition_by_entries()
{
    if (g_partition >= 0) return g_partition == 1;
    const char* e = getenv("SPMV_COMPAT_PARTITION");
    return e && (!strcmp(e, "nnz") || !strcmp(e, "entries") || !strcmp(e, "1"));
}

struct Shard
{
    int       device = 0;
    int64_t   row0 = 0, row1 = 0;
    spmv_mat* mat = nullptr;
    spmv_vec* y   = nullptr;
};

// runs the timed loop over ready-made shards, prints the line, copies y back, frees the shards
void run_shards(const char* fmt_name, std::vector<Shard>& shards, const Vector& x, Vector& y, double flops_per_apply)
{
    Engine& E = Engine::get();
    // One full replica of x per device in use (src/mat_vec.cpp:257,266), assembled ON the devices: device k receives
    // only its own slice of x from the host, the rest arrives from the other devices through spmv_comm_allgather (RCCL
    // or peer copies over xGMI).  SPMV_COMPAT_X_PARTS=n (tests) cuts x into n slices even on one GPU, so that the
    // exchange runs there too (several participants on one device).
    std::vector<int> devices;
    for (Shard& s : shards)
        if (std::find(devices.begin(), devices.end(), s.device) == devices.end()) devices.push_back(s.device);
    const char* env_parts = getenv("SPMV_COMPAT_X_PARTS");
    const int   parts     = std::max((int)devices.size(), env_parts ? atoi(env_parts) : 0);
    std::vector<spmv_ctx*> pctx((size_t)parts);
    std::vector<spmv_vec*> xrep((size_t)parts, nullptr);
    std::vector<int64_t>   xoff((size_t)parts + 1, 0);
    for (int p = 0; p < parts; ++p)
    {
        pctx[(size_t)p] = E.ctx(devices[(size_t)p % devices.size()]);
        int64_t b = 0, e = 0;
        check(spmv_partition_rows(x.size, parts, p, &b, &e), "spmv_partition_rows(x)");
        xoff[(size_t)p]     = b;
        xoff[(size_t)p + 1] = e;
        check(spmv_vec_create(pctx[(size_t)p], x.size, &xrep[(size_t)p]), "spmv_vec_create(x replica)");
        check(spmv_vec_upload(xrep[(size_t)p], b, e - b, x.values + b), "spmv_vec_upload(x slice)");
    }
    spmv_comm* comm = nullptr;
    check(spmv_comm_create(pctx.data(), parts, &comm), "spmv_comm_create");
    check(spmv_comm_allgather(comm, xrep.data(), xoff.data()), "spmv_comm_allgather(x)");
    auto replica_of = [&](int device) {
        for (int p = 0; p < parts; ++p)
            if (devices[(size_t)p % devices.size()] == device) return xrep[(size_t)p];
        return xrep[0];
    };
    for (Shard& s : shards)
    {
        check(spmv_vec_create(E.ctx(s.device), s.row1 - s.row0, &s.y), "spmv_vec_create(y shard)");
        check(spmv_vec_fill(s.y, 0.0), "spmv_vec_fill");  // memset(Y, 0), src/mat_vec.cpp:267
    }
    for (Shard& s : shards) check(spmv_sync(E.ctx(s.device)), "spmv_sync");
    for (int p = 0; p < parts; ++p) check(spmv_sync(pctx[(size_t)p]), "spmv_sync");

    // The reference's protocol: every repetition is joined before the next starts (src/mat_vec.cpp:272-281 creates and
    // joins its threads in every repetition), so launch latency and the imbalance between the shards of one repetition
    // are part of the printed figure, which stays comparable with the reference's own line.
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < g_numa_reps; ++k)
    {
        for (Shard& s : shards) check(spmv_apply(E.ctx(s.device), s.mat, replica_of(s.device), s.y), "spmv_apply(shard)");
        for (Shard& s : shards) check(spmv_sync(E.ctx(s.device)), "spmv_sync");
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // same expression as src/mat_vec.cpp:284 (milliseconds per repetition; the "+ secs/1000" term is the reference's)
    const double t_avg = (secs * 1000.0 + secs / 1000.0) / g_numa_reps;
    g_last_numa_ms     = secs * 1000.0 / g_numa_reps;
    printf("### %s NUMA GFLOPS = %.5f\n", fmt_name, flops_per_apply / t_avg / 1e6);
    // Beside it: the same products with every repetition queued on its device's stream and ONE wait at the end (the devices
    // work through their queues side by side, no launch latency between repetitions) - into scratch vectors, so that y
    // holds exactly the reference protocol's sums.
    {
        std::vector<spmv_vec*> scratch(shards.size(), nullptr);
        for (size_t i = 0; i < shards.size(); ++i)
        {
            check(spmv_vec_create(E.ctx(shards[i].device), shards[i].row1 - shards[i].row0, &scratch[i]), "spmv_vec_create(scratch y)");
            check(spmv_vec_fill(scratch[i], 0.0), "spmv_vec_fill");
        }
        for (Shard& s : shards) check(spmv_sync(E.ctx(s.device)), "spmv_sync");
        const auto q0 = std::chrono::steady_clock::now();
        for (int k = 0; k < g_numa_reps; ++k)
            for (size_t i = 0; i < shards.size(); ++i)
                check(spmv_apply(E.ctx(shards[i].device), shards[i].mat, replica_of(shards[i].device), scratch[i]), "spmv_apply(shard)");
        for (Shard& s : shards) check(spmv_sync(E.ctx(s.device)), "spmv_sync");
        const double qsecs = std::chrono::duration<double>(std::chrono::steady_clock::now() - q0).count();
        printf("### %s NUMA GFLOPS, all repetitions queued and one wait = %.5f\n", fmt_name, flops_per_apply / (qsecs * 1000.0 / g_numa_reps) / 1e6);
        // How even the partition is: stored entries per shard (max / mean) and every shard's own product time (HIP events, 5
        // products each, one shard at a time).  With one device per shard the step of the job is its SLOWEST shard; on fewer
        // devices the shards of a device run one after the other and only the sum shows in the line above.
        double  slowest = 0.0, sum_ms = 0.0;
        int64_t most = 0, total = 0;
        for (size_t i = 0; i < shards.size(); ++i)
        {
            spmv_mat_info info;
            check(spmv_mat_get_info(shards[i].mat, &info), "spmv_mat_get_info(shard)");
            const int64_t stored = info.format == SPMV_FMT_ELL || info.format == SPMV_FMT_DIA ? (int64_t)info.nrow * info.ell_k : info.nnz;
            most = std::max(most, stored);
            total += stored;
            double ms = 0.0;
            if (info.nrow > 0)
                check(spmv_apply_timed(E.ctx(shards[i].device), shards[i].mat, replica_of(shards[i].device), scratch[i], 5, &ms), "spmv_apply_timed(shard)");
            slowest = std::max(slowest, ms);
            sum_ms += ms;
        }
        g_last_slowest_shard_ms = slowest;
        g_last_shard_imbalance  = total > 0 ? (double)most * (double)shards.size() / (double)total : 1.0;
        printf("### %s NUMA shards = %zu, partition by %s: stored entries per shard max / mean = %.3f; slowest shard %.4f ms, all shards one after the other %.4f ms\n",
               fmt_name, shards.size(), partition_by_entries() ? "entries" : "rows", g_last_shard_imbalance, slowest, sum_ms);
        for (spmv_vec* v : scratch) spmv_vec_destroy(v);
    }

    // y: the shards' slices are concatenated on the first device (device-to-device / peer copies), one copy to the host
    spmv_vec* yfull = nullptr;
    check(spmv_vec_create(pctx[0], y.size, &yfull), "spmv_vec_create(y)");
    for (Shard& s : shards)
    {
        check(spmv_vec_copy(yfull, s.row0, s.y, 0, s.row1 - s.row0), "spmv_vec_copy(y shard)");
    }
    check(spmv_vec_download(yfull, 0, y.size, y.values), "spmv_vec_download(y)");
    spmv_vec_destroy(yfull);
    for (Shard& s : shards)
    {
        spmv_vec_destroy(s.y);
        spmv_mat_destroy(s.mat);
    }
    spmv_comm_destroy(comm);
    for (spmv_vec* v : xrep)
        if (v) spmv_vec_destroy(v);
    shards.clear();
}

// row_ptr64: offsets of the rows (nrow + 1 entries) when the caller has them - the by-entries partition is cut from them;
// nullptr (formats that store the same number of slots for every row: ELL, DIA) or the default mode: equal rows
std::vector<Shard> plan_shards(int64_t nrow, int nthreads, const int64_t* row_ptr64 = nullptr)
{
    Engine&            E = Engine::get();
    std::vector<Shard> shards((size_t)std::max(nthreads, 1));
    std::vector<int64_t> bounds;
    if (row_ptr64 && partition_by_entries())
    {
        bounds.resize(shards.size() + 1);
        check(spmv_partition_rows_balanced(nrow, row_ptr64, (int)shards.size(), bounds.data()), "spmv_partition_rows_balanced");
    }
    for (int i = 0; i < (int)shards.size(); ++i)
    {
        shards[(size_t)i].device = i % E.ngpus();
        if (!bounds.empty())
        {
            shards[(size_t)i].row0 = bounds[(size_t)i];
            shards[(size_t)i].row1 = bounds[(size_t)i + 1];
        }
        else
            check(spmv_partition_rows(nrow, (int)shards.size(), i, &shards[(size_t)i].row0, &shards[(size_t)i].row1), "spmv_partition_rows");
    }
    return shards;
}
}  // namespace

void   spmv_compat_set_numa_reps(int reps) { g_numa_reps = reps > 0 ? reps : 1; }
double spmv_compat_last_numa_ms(void) { return g_last_numa_ms; }
// (a prefetch is an optimisation: where no device is visible - a reader used on a build machine - it does nothing, and the
// first product says what is missing)
static bool device_visible()
{
    int n = 0;
    return spmv_device_count(&n) == SPMV_OK && n > 0;
}
void spmv_compat_prefetch(const COOMatrix& A)
{
    if (A.values && A.nnz > 0 && device_visible()) (void)device_coo(A);
}
void spmv_compat_prefetch(const CSCMatrix& A)
{
    if (A.values && A.col_ptr && device_visible()) (void)device_csc(A);
}
void spmv_compat_prefetch(const DIAMatrix& A)
{
    if (A.values && A.ndiags > 0 && device_visible()) (void)device_dia(A);
}

void   spmv_compat_set_partition(int by_entries) { g_partition = by_entries < 0 ? -1 : (by_entries ? 1 : 0); }
double spmv_compat_last_slowest_shard_ms(void) { return g_last_slowest_shard_ms; }
double spmv_compat_last_shard_imbalance(void) { return g_last_shard_imbalance; }

void CSRMatrixMatVectorNuma(const CSRMatrix& A, const Vector& x, Vector& y, int nthreads)
{
    Engine&              E      = Engine::get();
    std::vector<int64_t> rp64(A.row_ptr, A.row_ptr + A.nrow + 1);
    std::vector<Shard>   shards = plan_shards(A.nrow, nthreads, rp64.data());
    // The first shard selects its kernel (AUTO: a measurement); the others are built under ITS plan - the reference builds every
    // shard the same way (src/mat_vec.cpp:240-268), and shards that draw different kernels would differ in their last bits and
    // in their step times for no reason the user can see
    std::vector<unsigned char> plan;
    for (Shard& s : shards)  // rebased row_ptr, global columns (src/mat_vec.cpp:250-265) — done inside the ABI call
    {
        spmv_ctx* c = E.ctx(s.device);
        if (!plan.empty()) (void)spmv_ctx_set_plan(c, plan.data(), (int64_t)plan.size());
        check(spmv_csr_upload_shard(c, s.row0, s.row1, A.ncol, rp64.data(), A.col_ind, A.values, &s.mat), "spmv_csr_upload_shard");
        if (!plan.empty())
        {
            (void)spmv_ctx_set_plan(c, nullptr, 0);
            // a two-phase shard built under a plan was built without the search over where its product stream lies in THIS device's
            // memory (no part of a plan): it runs now, within the engine's default budget
            spmv_mat_info inf;
            if (spmv_mat_get_info(s.mat, &inf) == SPMV_OK && inf.kernel == SPMV_CSR_TWOPHASE)
                check(spmv_mat_set_param(s.mat, "twophase_choose_pieces", 1), "twophase_choose_pieces");
        }
        else
        {
            int64_t len = 0;
            if (spmv_mat_get_plan(s.mat, nullptr, &len) == SPMV_OK && len > 0)
            {
                plan.resize((size_t)len);
                if (spmv_mat_get_plan(s.mat, plan.data(), &len) != SPMV_OK) plan.clear();
            }
        }
    }
    run_shards("CSR", shards, x, y, 2.0 * (double)A.row_ptr[A.nrow]);
}

void COOMatrixMatVectorNuma(const COOMatrix& A, const Vector& x, Vector& y, int nthreads)
{
    // The reference scans a row-sorted COO for each thread's range (src/mat_vec.cpp:170-183) and is wrong for
    // unsorted input; here every entry goes to the shard that owns its row, in file order.
    Engine&                          E      = Engine::get();
    std::vector<int64_t>             rp64;
    if (partition_by_entries())  // entries per row of the file, whatever its order (the reference's scan assumes row-sorted input, :170-183)
    {
        rp64.assign((size_t)A.nrow + 1, 0);
        for (int k = 0; k < A.nnz; ++k) ++rp64[(size_t)A.row_ind[k] + 1];
        for (int i = 0; i < A.nrow; ++i) rp64[(size_t)i + 1] += rp64[(size_t)i];
    }
    std::vector<Shard>               shards = plan_shards(A.nrow, nthreads, rp64.empty() ? nullptr : rp64.data());
    std::vector<std::vector<int>>    rows(shards.size()), cols(shards.size());
    std::vector<std::vector<double>> vals(shards.size());
    // owner of a row = the planned shard whose range holds it (with more shards than rows all but the last are empty,
    // src/mat_vec.cpp:233: rows_per_thread = nrow / nthreads = 0, and the last one takes every row)
    std::vector<int64_t> first(shards.size());
    for (size_t s = 0; s < shards.size(); ++s) first[s] = shards[s].row0;
    auto owner = [&](int64_t r) {
        size_t s = (size_t)(std::upper_bound(first.begin(), first.end(), r) - first.begin()) - 1;
        while (s + 1 < shards.size() && r >= shards[s].row1) ++s;  // empty ranges share their first row with the next one
        return s;
    };
    for (int k = 0; k < A.nnz; ++k)
    {
        const size_t s = owner(A.row_ind[k]);
        rows[s].push_back(A.row_ind[k] - (int)shards[s].row0);  // local row, as the thread body does (:500)
        cols[s].push_back(A.col_ind[k]);
        vals[s].push_back(A.values[k]);
    }
    for (size_t s = 0; s < shards.size(); ++s)
        check(spmv_coo_upload(E.ctx(shards[s].device), (int)(shards[s].row1 - shards[s].row0), A.ncol, (int64_t)rows[s].size(),
                              rows[s].data(), cols[s].data(), vals[s].data(), &shards[s].mat),
              "spmv_coo_upload(shard)");
    run_shards("COO", shards, x, y, 2.0 * (double)A.nnz);
}

void ELLMatrixMatVectorNuma(const ELLMatrix& A, const Vector& x, Vector& y, int nthreads)
{
    // Row shards that stay column-major with the shard's own stride.  (The reference slices the column-major
    // array as if it were row-major, src/mat_vec.cpp:394-395; the intent — shard the rows — is what is built.)
    Engine&             E      = Engine::get();
    std::vector<Shard>  shards = plan_shards(A.nrow, nthreads);
    const int           K      = A.nonzeros_in_row;
    std::vector<int>    sc;
    std::vector<double> sv;
    for (Shard& s : shards)
    {
        const int64_t rows = s.row1 - s.row0;
        sc.resize((size_t)(rows * K));
        sv.resize((size_t)(rows * K));
        for (int k = 0; k < K; ++k)
        {
            std::memcpy(sc.data() + (size_t)k * rows, A.col_ind + (size_t)k * A.nrow + s.row0, (size_t)rows * sizeof(int));
            std::memcpy(sv.data() + (size_t)k * rows, A.values + (size_t)k * A.nrow + s.row0, (size_t)rows * sizeof(double));
        }
        check(spmv_ell_upload(E.ctx(s.device), (int)rows, A.ncol, K, rows * K, sc.data(), sv.data(), &s.mat), "spmv_ell_upload(shard)");
    }
    // the reference's ELL driver counts padded slots (src/mat_vec.cpp:415)
    run_shards("ELL", shards, x, y, 2.0 * (double)A.nrow * (double)K);
}

void CSCMatrixMatVectorNuma(const CSCMatrix& A, const Vector& x, Vector& y, int nthreads)
{
    // COLUMN-range shards, as the reference cuts them (src/mat_vec.cpp:299-337): shard i holds columns [c0, c1) with a
    // rebased col_ptr, ITS SLICE OF x ONLY (:329 — no replica, no exchange) and a private full-length Y (:330).  The
    // reference leaves the partial Y vectors where they are; here they are summed on the first device (device-to-device
    // / peer copies) and y comes back once — the reduction a column partition needs.
    Engine&   E = Engine::get();
    const int n = std::max(nthreads, 1);
    struct ColShard
    {
        int       device = 0;
        int64_t   c0 = 0, c1 = 0;
        spmv_mat* mat = nullptr;
        spmv_vec *x = nullptr, *y = nullptr;
    };
    std::vector<ColShard> shards((size_t)n);
    std::vector<int>      sub_ptr;
    std::vector<int64_t>  cbounds;
    if (partition_by_entries())  // columns cut so that every shard stores about the same number of entries
    {
        std::vector<int64_t> cp64(A.col_ptr, A.col_ptr + A.ncol + 1);
        cbounds.resize((size_t)n + 1);
        check(spmv_partition_rows_balanced(A.ncol, cp64.data(), n, cbounds.data()), "spmv_partition_rows_balanced(columns)");
    }
    for (int i = 0; i < n; ++i)
    {
        ColShard& s = shards[(size_t)i];
        s.device    = i % E.ngpus();
        if (!cbounds.empty())
        {
            s.c0 = cbounds[(size_t)i];
            s.c1 = cbounds[(size_t)i + 1];
        }
        else
            check(spmv_partition_rows(A.ncol, n, i, &s.c0, &s.c1), "spmv_partition_rows(columns)");  // equal columns, the last takes the rest (:302,315)
        const int     base = A.col_ptr[s.c0];
        const int64_t cols = s.c1 - s.c0;
        sub_ptr.resize((size_t)cols + 1);
        for (int64_t j = 0; j <= cols; ++j) sub_ptr[(size_t)j] = A.col_ptr[s.c0 + j] - base;  // :333-336
        spmv_ctx* c = E.ctx(s.device);
        check(spmv_csc_upload(c, A.nrow, (int)cols, sub_ptr.data(), A.row_ind + base, A.values + base, &s.mat), "spmv_csc_upload(shard)");
        check(spmv_vec_create(c, cols, &s.x), "spmv_vec_create(x slice)");
        check(spmv_vec_upload(s.x, 0, cols, x.values + s.c0), "spmv_vec_upload(x slice)");
        check(spmv_vec_create(c, A.nrow, &s.y), "spmv_vec_create(partial y)");
        check(spmv_vec_fill(s.y, 0.0), "spmv_vec_fill");
    }
    for (ColShard& s : shards) check(spmv_sync(E.ctx(s.device)), "spmv_sync");
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < g_numa_reps; ++k)
    {
        for (ColShard& s : shards) check(spmv_apply(E.ctx(s.device), s.mat, s.x, s.y), "spmv_apply(shard)");
        for (ColShard& s : shards) check(spmv_sync(E.ctx(s.device)), "spmv_sync");
    }
    const double secs  = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double t_avg = (secs * 1000.0 + secs / 1000.0) / g_numa_reps;  // src/mat_vec.cpp:353
    g_last_numa_ms     = secs * 1000.0 / g_numa_reps;
    printf("### CSC NUMA GFLOPS = %.5f\n", 2.0 * (double)A.col_ptr[A.ncol] / t_avg / 1e6);
    // y = sum of the partial vectors, in shard order, on the first shard's device
    spmv_ctx* c0  = E.ctx(shards[0].device);
    spmv_vec* tmp = nullptr;
    if (n > 1) check(spmv_vec_create(c0, A.nrow, &tmp), "spmv_vec_create(reduction buffer)");
    for (int i = 1; i < n; ++i)
    {
        check(spmv_vec_copy(tmp, 0, shards[(size_t)i].y, 0, A.nrow), "spmv_vec_copy(partial y)");
        check(spmv_axpby(c0, 1.0, shards[0].y, 1.0, tmp, shards[0].y), "spmv_axpby(partial y)");
    }
    check(spmv_vec_download(shards[0].y, 0, A.nrow, y.values), "spmv_vec_download(y)");
    if (tmp) spmv_vec_destroy(tmp);
    for (ColShard& s : shards)
    {
        spmv_vec_destroy(s.x);
        spmv_vec_destroy(s.y);
        spmv_mat_destroy(s.mat);
    }
}

void DIAMatrixMatVectorNuma(const DIAMatrix& A, const Vector& x, Vector& y, int nthreads)
{
    // Row shards (src/mat_vec.cpp:428-484: rows_per_thread rows of the row-major diagonal array per thread, :449-452).
    // A shard's local row i is global row row0 + i, so its diagonals sit at offsets + row0 against the full x; the
    // column bound stays the global one (the reference checks `j < nrow` with the global nrow, :572 / :140).
    Engine&            E      = Engine::get();
    std::vector<Shard> shards = plan_shards(A.nrow, nthreads);
    std::vector<int>   off((size_t)std::max(A.ndiags, 1));
    for (Shard& s : shards)
    {
        for (int d = 0; d < A.ndiags; ++d) off[(size_t)d] = A.offsets[d] + (int)s.row0;
        check(spmv_dia_upload(E.ctx(s.device), (int)(s.row1 - s.row0), A.ncol, A.ndiags, off.data(),
                              A.values + (size_t)s.row0 * (size_t)A.ndiags, &s.mat), "spmv_dia_upload(shard)");
        check(spmv_mat_set_param(s.mat, "dia_col_bound", std::min(A.nrow, A.ncol)), "spmv_mat_set_param(dia_col_bound)");
    }
    run_shards("DIA", shards, x, y, 2.0 * (double)A.nnz);
}

// =====================================================================================================
// BLAS-1 (include/vec_vec.h:6-7)
// =====================================================================================================
double vec_dot(const Vector& x, const Vector& y)
{
    Engine&   E  = Engine::get();
    spmv_vec* dx = E.pooled(0, 0, x.size);
    spmv_vec* dy = E.pooled(0, 1, x.size);
    check(spmv_vec_upload(dx, 0, x.size, x.values), "spmv_vec_upload");
    check(spmv_vec_upload(dy, 0, x.size, y.values), "spmv_vec_upload");
    double r = 0.0;
    check(spmv_dot(E.ctx(0), dx, dy, &r), "spmv_dot");
    return r;
}

void vec_axpby(double alpha, const Vector& x, double beta, const Vector& y, const Vector& w)
{
    Engine&       E  = Engine::get();
    const int64_t n  = w.size;  // the reference sizes the loop by w (src/vec_vec.cpp:33)
    spmv_vec*     dx = E.pooled(0, 0, n);
    spmv_vec*     dy = E.pooled(0, 1, n);
    spmv_vec*     dw = E.pooled(0, 2, n);
    // alpha == 0 never reads x and beta == 0 never reads y (src/vec_vec.cpp:38-53): do not even upload them
    if (alpha != 0) check(spmv_vec_upload(dx, 0, n, x.values), "spmv_vec_upload");
    if (beta != 0) check(spmv_vec_upload(dy, 0, n, y.values), "spmv_vec_upload");
    check(spmv_axpby(E.ctx(0), alpha, dx, beta, dy, dw), "spmv_axpby");
    check(spmv_vec_download(dw, 0, n, w.values), "spmv_vec_download");
}
