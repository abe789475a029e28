// ref_wrap.cpp — extern "C" doorway into the REAL reference, for the oracle only.
//
// TEST INFRASTRUCTURE.  Compiled by oracle/Makefile together with the reference's own sources
// (taken where they lie under $(REF)/src, never copied) into oracle/_ref/libarmspmv_ref.so.
// It lets tests/ and tests/golden/make_golden.py call the reference's C++ functions on plain
// arrays, and lets bench.py time the reference's OpenMP CSR loop as the cpu_baseline
// ("kind": "reference").  Nothing in the product path links or loads this library.
//
// The reference's containers take ownership of the pointers handed to their field constructors
// and delete[] them in their destructors (src/matrix.cpp:31-39, src/vector.cpp:21-25), so every
// wrapper detaches the borrowed pointers before the container goes out of scope.
#include <cstdint>
#include <cstring>

#include "mat_vec.h"
#include "matrix.h"
#include "vec_vec.h"
#include "vector.h"

namespace
{
struct BorrowedVec
{
    Vector v;
    BorrowedVec(int n, const double* p) : v(n, const_cast<double*>(p)) {}
    ~BorrowedVec() { v.values = 0; v.size = 0; }
};

void detach(COOMatrix& A) { A.row_ind = 0; A.col_ind = 0; A.values = 0; }
void detach(CSRMatrix& A) { A.row_ptr = 0; A.col_ind = 0; A.values = 0; A.diagonal = 0; }
void detach(CSCMatrix& A) { A.row_ind = 0; A.col_ptr = 0; A.values = 0; }
void detach(ELLMatrix& A) { A.col_ind = 0; A.values = 0; A.diagonal = 0; }
}  // namespace

extern "C" {

// ---- include/mat_vec.h:7-11 -----------------------------------------------------------------
void ref_coo_spmv(int nrow, int ncol, int nnz, const int* row, const int* col, const double* val,
                  const double* x, double* y)
{
    COOMatrix   A(nrow, ncol, nnz, const_cast<int*>(row), const_cast<int*>(col), const_cast<double*>(val));
    BorrowedVec bx(ncol, x), by(nrow, y);
    COOMatirxMatVector(A, bx.v, by.v);
    detach(A);
}

void ref_csr_spmv(int nrow, int ncol, const int* row_ptr, const int* col, const double* val,
                  const double* x, double* y)
{
    CSRMatrix   A(nrow, ncol, const_cast<int*>(row_ptr), const_cast<int*>(col), const_cast<double*>(val), 0);
    BorrowedVec bx(ncol, x), by(nrow, y);
    CSRMatrixMatVector(A, bx.v, by.v);
    detach(A);
}

void ref_csc_spmv(int nrow, int ncol, const int* col_ptr, const int* row, const double* val,
                  const double* x, double* y)
{
    // field ctor argument order is (n, m, row_ind, col_ptr, values) in include/matrix.h:59
    CSCMatrix A;
    A.nrow    = nrow;
    A.ncol    = ncol;
    A.col_ptr = const_cast<int*>(col_ptr);
    A.row_ind = const_cast<int*>(row);
    A.values  = const_cast<double*>(val);
    BorrowedVec bx(ncol, x), by(nrow, y);
    CSCMatrixMatVector(A, bx.v, by.v);
    detach(A);
}

void ref_ell_spmv(int nrow, int ncol, int nnz, int k, const int* col, const double* val,
                  const double* x, double* y)
{
    ELLMatrix   A(nrow, ncol, nnz, k, const_cast<int*>(col), const_cast<double*>(val), 0);
    BorrowedVec bx(ncol, x), by(nrow, y);
    ELLMatrixMatVector(A, bx.v, by.v);
    detach(A);
}

void ref_dia_spmv(int nrow, int ncol, int ndiags, const int* offsets, const double* val,
                  const double* x, double* y)
{
    DIAMatrix A;
    A.nrow    = nrow;
    A.ncol    = ncol;
    A.ndiags  = ndiags;
    A.offsets = const_cast<int*>(offsets);
    A.values  = const_cast<double*>(val);
    BorrowedVec bx(ncol, x), by(nrow, y);
    DIAMatrixMatVector(A, bx.v, by.v);
    A.offsets = 0;  // DIAMatrix::Free() delete[]s both (src/matrix.cpp:787-798)
    A.values  = 0;
}

// ---- converting constructors (src/matrix.cpp:115-154, :295-325, :450-500, :673-726) ------------
// The caller sizes the outputs: row_ptr[nrow+1], col/val[nnz], diagonal[nrow].
void ref_coo_to_csr(int nrow, int ncol, int nnz, const int* row, const int* col, const double* val,
                    int* out_row_ptr, int* out_col, double* out_val)
{
    COOMatrix A(nrow, ncol, nnz, const_cast<int*>(row), const_cast<int*>(col), const_cast<double*>(val));
    {
        CSRMatrix B(A);
        std::memcpy(out_row_ptr, B.row_ptr, sizeof(int) * (size_t)(nrow + 1));
        std::memcpy(out_col, B.col_ind, sizeof(int) * (size_t)nnz);
        std::memcpy(out_val, B.values, sizeof(double) * (size_t)nnz);
    }
    detach(A);
}

void ref_coo_to_csc(int nrow, int ncol, int nnz, const int* row, const int* col, const double* val,
                    int* out_col_ptr, int* out_row, double* out_val)
{
    COOMatrix A(nrow, ncol, nnz, const_cast<int*>(row), const_cast<int*>(col), const_cast<double*>(val));
    {
        CSCMatrix C(A);
        std::memcpy(out_col_ptr, C.col_ptr, sizeof(int) * (size_t)(ncol + 1));
        std::memcpy(out_row, C.row_ind, sizeof(int) * (size_t)nnz);
        std::memcpy(out_val, C.values, sizeof(double) * (size_t)nnz);
    }
    detach(A);
}

// Two-step: first call with out_col == NULL returns K; second call fills nrow*K entries.
// (The constructor keeps an int[nrow] VLA on the stack, src/matrix.cpp:457 — small nrow only.)
int ref_coo_to_ell(int nrow, int ncol, int nnz, const int* row, const int* col, const double* val,
                   int* out_col, double* out_val)
{
    COOMatrix A(nrow, ncol, nnz, const_cast<int*>(row), const_cast<int*>(col), const_cast<double*>(val));
    int       k;
    {
        ELLMatrix D(A);
        k = D.nonzeros_in_row;
        if (out_col)
        {
            std::memcpy(out_col, D.col_ind, sizeof(int) * (size_t)nrow * (size_t)k);
            std::memcpy(out_val, D.values, sizeof(double) * (size_t)nrow * (size_t)k);
        }
    }
    detach(A);
    return k;
}

// First call with out_offsets == NULL returns ndiags; second call fills offsets[ndiags] and
// values[nrow*ndiags].  DIAMatrix mallocs what its Free() delete[]s (src/matrix.cpp:698-699 vs
// :789-798); the wrapper frees the buffers itself with free() and detaches to stay defined.
int ref_csr_to_dia(int nrow, int ncol, const int* row_ptr, const int* col, const double* val,
                   int* out_offsets, double* out_val)
{
    CSRMatrix A(nrow, ncol, const_cast<int*>(row_ptr), const_cast<int*>(col), const_cast<double*>(val), 0);
    int       nd;
    {
        DIAMatrix E(A);
        nd = E.ndiags;
        if (out_offsets)
        {
            std::memcpy(out_offsets, E.offsets, sizeof(int) * (size_t)nd);
            std::memcpy(out_val, E.values, sizeof(double) * (size_t)nrow * (size_t)nd);
        }
        free(E.offsets);
        free(E.values);
        E.offsets = 0;
        E.values  = 0;
    }
    detach(A);
    return nd;
}

// ---- include/vec_vec.h:6-7 --------------------------------------------------------------------
double ref_dot(int n, const double* x, const double* y)
{
    BorrowedVec bx(n, x), by(n, y);
    return vec_dot(bx.v, by.v);
}

void ref_axpby(int n, double alpha, const double* x, double beta, const double* y, double* w)
{
    BorrowedVec bx(n, x), by(n, y), bw(n, w);
    vec_axpby(alpha, bx.v, beta, by.v, bw.v);
}

// ---- the NUMA driver (src/mat_vec.cpp:230-297): prints its own GFLOPS line, discards Y ------------
void ref_csr_spmv_numa(int nrow, int ncol, const int* row_ptr, const int* col, const double* val,
                       const double* x, double* y, int nthreads)
{
    CSRMatrix   A(nrow, ncol, const_cast<int*>(row_ptr), const_cast<int*>(col), const_cast<double*>(val), 0);
    BorrowedVec bx(ncol, x), by(nrow, y);
    CSRMatrixMatVectorNuma(A, bx.v, by.v, nthreads);
    detach(A);
}

}  // extern "C"
