// arm_spmv_compat.hpp — source-compatible C++ face of the reference (Layer 1 of the drop-in boundary).
//
// The reference's API is ten C++ free functions over classes with public raw-pointer fields
// (reference include/mat_vec.h:7-17, include/matrix.h:7-138, include/vector.h:4-26,
// include/vec_vec.h:6-7, include/data_io.h:9-15, include/mytime.h:4).  This header re-declares those
// classes and functions — same names (typo `COOMatirxMatVector` included), same public fields in the
// same order, same constructor signatures — so that the reference's main.cpp compiles against it
// unchanged (`make -C arm-spmv_amd/host ref-main` does exactly that).  The implementation
// (arm-spmv_amd/host/compat.cpp) is new: containers still own host arrays (callers read the fields),
// but every product, conversion and BLAS-1 call goes through the C ABI of include/spmv_abi.h to the HIP
// kernels; there is no host implementation of the arithmetic in the shim.
//
// The forwarding headers in include/compat/ (matrix.h, vector.h, mat_vec.h, vec_vec.h, data_io.h,
// mytime.h, mmio.h) all include this file, so `-Iinclude/compat` replaces the reference's `-I include`.
//
// Differences a caller can observe (all documented in INTEGRATION.md):
//   * each *MatVector call copies x and y to the GPU and y back (the signature hands over host
//     memory); the matrix itself is uploaded once and cached per container (keyed by its `values`
//     pointer and re-checked on every call against a fingerprint: dimensions, array addresses and a sample of the
//     array contents).  An in-place edit that misses every sampled element needs spmv_compat_invalidate(A.values)
//     — or (&A) — before the next product.
//   * the `...Numa` drivers shard over GPUs instead of NUMA nodes: `nthreads` = number of shards,
//     shard i lives on GPU i % ngpus (reference: node i % numanodes, src/mat_vec.cpp:242).  They
//     print the same `### <FMT> NUMA GFLOPS = %.5f` line.  Unlike the reference (which drops the
//     per-thread Y slices, src/mat_vec.cpp:287-296) they copy the result into y.
//   * errors from the engine print the message and exit(1), like the reference's I/O errors
//     (src/data_io.cpp:53-75).
#ifndef ARM_SPMV_COMPAT_HPP
#define ARM_SPMV_COMPAT_HPP

#include <stdio.h>
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------------
// Vector — reference include/vector.h:4-26.  16-byte object {int size; double* values}, new[]-owned.
// ---------------------------------------------------------------------------------------------------
class Vector
{
public:
    int     size;
    double* values;

    Vector();
    Vector(int n, double* values);  // adopts the pointer (deleted in the destructor), as the reference does
    Vector(const Vector& x);
    ~Vector();
    Vector& operator=(double a);
    Vector& operator=(const Vector& x);

    void Free();
    void Resize(int n);
    void Fill(double a) const;
    void FillRandom() const;  // rand()/RAND_MAX, unseeded (src/vector.cpp:65-69)
    void Copy(const Vector& x) const;
    void Scale(double a) const;
    void Shift(double a) const;
    void AddScaled(double a, const Vector& x) const;
    void Add2Scaled(double a, const Vector& x, double b, const Vector& y) const;
};

bool checkVector(const Vector& x, const Vector& y);  // |x_i - y_i| <= 1e-6 for all i (src/vector.cpp:161-171)

// ---------------------------------------------------------------------------------------------------
// Sparse containers — reference include/matrix.h.  Field order is part of the contract.
// ---------------------------------------------------------------------------------------------------
class COOMatrix  // include/matrix.h:7-25: 0-based, file order, duplicates allowed
{
public:
    int nrow;
    int ncol;
    int nnz;

    int*    row_ind;
    int*    col_ind;
    double* values;

    COOMatrix();
    COOMatrix(int n, int m, int nnz, int* row_ind, int* col_ind, double* values);
    COOMatrix(const COOMatrix& A);
    ~COOMatrix();
    COOMatrix& operator=(const COOMatrix& A);

    void Free();
};

class CSRMatrix  // include/matrix.h:27-47: no nnz field, use row_ptr[nrow]
{
public:
    int nrow;
    int ncol;

    int*    row_ptr;
    int*    col_ind;
    double* values;
    double* diagonal;  // packed diagonal entries in COO encounter order (src/matrix.cpp:146-153)

    CSRMatrix();
    CSRMatrix(int n, int m, int* row_ptr, int* col_ind, double* values, double* diagonal);
    CSRMatrix(const CSRMatrix& A);
    CSRMatrix(const COOMatrix& A);  // stable counting sort by row — runs on the GPU (spmv_coo_to_csr)
    ~CSRMatrix();
    CSRMatrix& operator=(const CSRMatrix& A);
    CSRMatrix& operator=(const COOMatrix& A);

    void Free();
};

class CSCMatrix  // include/matrix.h:49-68
{
public:
    int nrow;
    int ncol;

    int*    row_ind;
    int*    col_ptr;
    double* values;

    CSCMatrix();
    CSCMatrix(int n, int m, int* row_ind, int* col_ptr, double* values);
    CSCMatrix(const CSCMatrix& A);
    CSCMatrix(const COOMatrix& A);
    ~CSCMatrix();
    CSCMatrix& operator=(const CSCMatrix& A);
    CSCMatrix& operator=(const COOMatrix& A);

    void Free();
};

class ELLMatrix  // include/matrix.h:70-92: COLUMN-major, element (row i, slot k) at i + k*nrow
{
public:
    int nrow;
    int ncol;
    int nnz;
    int nonzeros_in_row;

    int*    col_ind;
    double* values;
    double* diagonal;

    ELLMatrix();
    ELLMatrix(int n, int m, int nnz, int nonzeros_in_row, int* col_ind, double* values, double* diagonal);
    ELLMatrix(const ELLMatrix& A);
    ELLMatrix(const COOMatrix& A);  // runs on the GPU (spmv_coo_to_ell); no stack VLA, any nrow
    ~ELLMatrix();
    ELLMatrix& operator=(const ELLMatrix& A);
    ELLMatrix& operator=(const COOMatrix& A);

    void Free();
};

class DIAMatrix  // include/matrix.h:117-138: row-major, (row i, diagonal d) at i*ndiags + d
{
public:
    int nnz;
    int nrow;
    int ncol;
    int ndiags;

    int*    offsets;
    double* values;

    DIAMatrix();
    DIAMatrix(int n, int m, int ndiags, int* offsets, double* values);
    DIAMatrix(const DIAMatrix& A);
    DIAMatrix(const CSRMatrix& A);
    ~DIAMatrix();
    DIAMatrix& operator=(const DIAMatrix& A);
    DIAMatrix& operator=(const CSRMatrix& A);

    void Free();
};

// ---------------------------------------------------------------------------------------------------
// y += A*x — reference include/mat_vec.h:7-17.  `nthreads` of the Numa drivers = number of GPU shards.
// ---------------------------------------------------------------------------------------------------
void COOMatirxMatVector(const COOMatrix& A, const Vector& x, Vector& y);
void CSRMatrixMatVector(const CSRMatrix& A, const Vector& x, Vector& y);
void CSCMatrixMatVector(const CSCMatrix& A, const Vector& x, Vector& y);
void ELLMatrixMatVector(const ELLMatrix& A, const Vector& x, Vector& y);
void DIAMatrixMatVector(const DIAMatrix& A, const Vector& x, Vector& y);

void COOMatrixMatVectorNuma(const COOMatrix& A, const Vector& x, Vector& y, int nthreads);
void CSRMatrixMatVectorNuma(const CSRMatrix& A, const Vector& x, Vector& y, int nthreads);
void CSCMatrixMatVectorNuma(const CSCMatrix& A, const Vector& x, Vector& y, int nthreads);
void ELLMatrixMatVectorNuma(const ELLMatrix& A, const Vector& x, Vector& y, int nthreads);
void DIAMatrixMatVectorNuma(const DIAMatrix& A, const Vector& x, Vector& y, int nthreads);

// ---------------------------------------------------------------------------------------------------
// BLAS-1 — reference include/vec_vec.h:6-7 (`w` is const& yet written, as in the reference)
// ---------------------------------------------------------------------------------------------------
double vec_dot(const Vector& x, const Vector& y);
void   vec_axpby(double alpha, const Vector& x, double beta, const Vector& y, const Vector& w);

// ---------------------------------------------------------------------------------------------------
// I/O and timer — reference include/data_io.h:9-15, include/mytime.h:4
// ---------------------------------------------------------------------------------------------------
void VectorRead(const char* filename, Vector& x);
void VectorWrite(const char* filename, const Vector& x);
void COOMatrixRead(const char* filename, COOMatrix& A);
void CSRMatrixRead(const char* filename, CSRMatrix& A);
void CSCMatrixRead(const char* filename, CSCMatrix& A);
void ELLMatrixRead(const char* filename, ELLMatrix& A);

double mytimer(void);  // seconds since the first call; the first call returns 0.0 (src/mytime.cpp:6-18)

// ---------------------------------------------------------------------------------------------------
// Additions (not in the reference)
// ---------------------------------------------------------------------------------------------------
// Forget the cached device copy of a container whose arrays were edited in place.
void spmv_compat_invalidate(const void* container_values_pointer);
// Upload a container's device copy NOW (it is cached; the products find it): what the reader and the converting constructors do
// at their end, so that a container's set-up - upload, analysis, kernel selection - falls where the reference does its own set-up,
// before the timed loop (main.cpp:34,64,74,84,94), not into the first product.
void spmv_compat_prefetch(const COOMatrix& A);
void spmv_compat_prefetch(const CSCMatrix& A);
void spmv_compat_prefetch(const DIAMatrix& A);
// Repetitions the Numa drivers time (reference: NTESTS = 50, src/mat_vec.cpp:201).  For tests.
void spmv_compat_set_numa_reps(int reps);
// Milliseconds per application measured by the last Numa driver call (device-resident, HIP events).
double spmv_compat_last_numa_ms(void);
// How the ...MatVectorNuma drivers cut their shards: 0 = equal rows, the last shard takes the remainder - the reference's split
// (src/mat_vec.cpp:151,:233,:245-246) and the default; 1 = by stored entries (spmv_partition_rows_balanced; CSC: columns cut
// by entries; ELL / DIA store the same number of slots for every row and keep equal rows); -1 = back to the environment's
// SPMV_COMPAT_PARTITION=rows|nnz.  The drivers print `### <FMT> NUMA shards = ...` with the entries per shard (max / mean)
// and the slowest shard's own product time - the step of a job with one GPU per shard.
void   spmv_compat_set_partition(int by_entries);
double spmv_compat_last_slowest_shard_ms(void);
double spmv_compat_last_shard_imbalance(void);

#endif  // ARM_SPMV_COMPAT_HPP
