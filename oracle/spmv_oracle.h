/*
 * spmv_oracle.h — CPU restatement of the arm-spmv hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity checker for the HIP engine.  It is never linked into, loaded by
 * or called from the product path (libspmv_hip.so, the C++ compat shim, the harness).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Every function restates one loop of the reference (file:line given per function) in plain C,
 * single-threaded, same operation order, so that it is bit-identical to the reference compiled
 * on this host with its own flags and OMP_NUM_THREADS=1.  Pinning: tests/test_oracle_vs_ref.py
 * compares it with oracle/_ref/libarmspmv_ref.so (the reference's own sources compiled where
 * they lie) and tests/golden/ holds vectors produced by that reference build.
 *
 * Two arithmetic flavours are exported for every multiply-add loop:
 *   orc_<name>      : separate multiply and add, -ffp-contract=off  (what g++ -O2 emits on x86-64,
 *                     i.e. what oracle/_ref computes; pinned bitwise)
 *   orc_<name>_fma  : fused multiply-add                            (what g++ -O2 emits on the
 *                     reference's native aarch64, where -ffp-contract=fast turns `s += a*b`
 *                     into fmadd; also what the HIP kernels compute)
 */
#ifndef SPMV_ORACLE_H
#define SPMV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- y += A*x ------------------------------------------------------------------------- */
/* src/mat_vec.cpp:32-40  (COOMatirxMatVector, serial order, duplicates summed) */
void orc_coo_spmv(int64_t nnz, const int32_t* row, const int32_t* col, const double* val,
                  const double* x, double* y);
void orc_coo_spmv_fma(int64_t nnz, const int32_t* row, const int32_t* col, const double* val,
                      const double* x, double* y);
/* src/mat_vec.cpp:57-65  (CSRMatrixMatVector: private sum from 0.0, one += per row) */
void orc_csr_spmv(int32_t nrow, const int32_t* row_ptr, const int32_t* col, const double* val,
                  const double* x, double* y);
void orc_csr_spmv_fma(int32_t nrow, const int32_t* row_ptr, const int32_t* col,
                      const double* val, const double* x, double* y);
/* same loop with `#pragma omp parallel for` as in src/mat_vec.cpp:54-57 (cpu_baseline "port") */
void orc_csr_spmv_omp(int32_t nrow, const int32_t* row_ptr, const int32_t* col,
                      const double* val, const double* x, double* y);
/* first-touch copy of a CSR matrix, x and a zeroed y inside an OpenMP team with the product's static row schedule
 * (cpu_baseline leg: BASELINE.md section 4 "first-touch initialisation") */
void orc_csr_first_touch_copy(int32_t nrow, int64_t ncol, const int32_t* row_ptr, const int32_t* col, const double* val,
                              const double* x, int32_t* d_row_ptr, int32_t* d_col, double* d_val, double* d_x, double* d_y);
/* NUMA-driver protocol with persistent pinned workers (src/mat_vec.cpp:230-297); returns ms per repetition */
double orc_csr_spmv_sharded(int32_t nrow, int32_t ncol, const int32_t* row_ptr, const int32_t* col, const double* val,
                            const double* x, double* y, int32_t nshards, int32_t reps);
/* src/mat_vec.cpp:82-93  (CSCMatrixMatVector, serial) */
void orc_csc_spmv(int32_t ncol, const int32_t* col_ptr, const int32_t* row, const double* val,
                  const double* x, double* y);
void orc_csc_spmv_fma(int32_t ncol, const int32_t* col_ptr, const int32_t* row,
                      const double* val, const double* x, double* y);
/* src/mat_vec.cpp:107-118 (ELLMatrixMatVector: k outer, i inner, column-major i + k*nrow) */
void orc_ell_spmv(int32_t nrow, int32_t k, const int32_t* col, const double* val,
                  const double* x, double* y);
void orc_ell_spmv_fma(int32_t nrow, int32_t k, const int32_t* col, const double* val,
                      const double* x, double* y);
/* src/mat_vec.cpp:135-145 (DIAMatrixMatVector; bound check against nrow as in :140) */
void orc_dia_spmv(int32_t nrow, int32_t ndiags, const int32_t* offsets, const double* val,
                  const double* x, double* y);
void orc_dia_spmv_fma(int32_t nrow, int32_t ndiags, const int32_t* offsets, const double* val,
                      const double* x, double* y);

/* ---- format conversion ---------------------------------------------------------------- */
/* src/matrix.cpp:115-154 CSRMatrix(const COOMatrix&): histogram, inclusive scan, backward stable
 * scatter.  diagonal[] is packed in COO encounter order (matrix.cpp:146-153); returns its count. */
int32_t orc_coo_to_csr(int32_t nrow, int64_t nnz, const int32_t* row, const int32_t* col,
                       const double* val, int32_t* row_ptr, int32_t* out_col, double* out_val,
                       double* diagonal);
/* src/matrix.cpp:295-325 CSCMatrix(const COOMatrix&) */
void orc_coo_to_csc(int32_t ncol, int64_t nnz, const int32_t* row, const int32_t* col,
                    const double* val, int32_t* col_ptr, int32_t* out_row, double* out_val);
/* src/matrix.cpp:456-470: K = longest row */
int32_t orc_coo_max_row_nnz(int32_t nrow, int64_t nnz, const int32_t* row);
/* src/matrix.cpp:472-489 ELLMatrix(const COOMatrix&): zero padded (col 0, val 0.0), column-major,
 * slot order = COO order.  out arrays hold nrow*k entries.  (Heap counters instead of the
 * reference's stack VLA at :457, so nrow is not limited by the stack.) */
void orc_coo_to_ell(int32_t nrow, int32_t k, int64_t nnz, const int32_t* row,
                    const int32_t* col, const double* val, int32_t* out_col, double* out_val);
/* src/matrix.cpp:673-726 DIAMatrix(const CSRMatrix&): pass 1 returns ndiags and fills
 * offsets (ascending); pass 2 fills row-major values (later duplicate overwrites, :721).
 * The scratch map has one more slot than the reference's (which overruns for entry (0,ncol-1)). */
int32_t orc_csr_count_diags(int32_t nrow, int32_t ncol, const int32_t* row_ptr,
                            const int32_t* col, int32_t* offsets /* may be NULL */);
void orc_csr_to_dia(int32_t nrow, int32_t ncol, const int32_t* row_ptr, const int32_t* col,
                    const double* val, int32_t ndiags, const int32_t* offsets, double* out_val);

/* ---- row-range sharding (src/mat_vec.cpp:233,245-246,250-251,260-263) -------------------- */
/* equal rows, last part takes the remainder */
void orc_partition_rows(int64_t nrow, int32_t nparts, int32_t part, int64_t* begin,
                        int64_t* end);
/* rebased row_ptr of rows [begin,end): sub_row_ptr[j] = row_ptr[begin+j] - row_ptr[begin] */
void orc_csr_shard_row_ptr(const int32_t* row_ptr, int64_t begin, int64_t end,
                           int32_t* sub_row_ptr);

/* ---- BLAS-1 (src/vec_vec.cpp:15-29, :31-94) ------------------------------------------- */
double orc_dot(int64_t n, const double* x, const double* y);
double orc_dot_fma(int64_t n, const double* x, const double* y);
void   orc_axpby(int64_t n, double alpha, const double* x, double beta, const double* y,
                 double* w);
void   orc_axpby_fma(int64_t n, double alpha, const double* x, double beta, const double* y,
                     double* w);

/* ---- parity-gate helper (not a reference function) -------------------------- */
/* s[i] = sum_j |a_ij| * |x_j| for CSR */
void orc_csr_abs_row_sums(int32_t nrow, const int32_t* row_ptr, const int32_t* col,
                          const double* val, const double* x, double* s);

/* ---- the sweep the `diagonal // for SymGS` fields were reserved for (include/matrix.h:36,81) --------------- */
/* The reference holds the fields and no sweep, so there is no reference loop to restate and nothing of the
 * reference's to pin this against (PARITY UNPINNED; the tests add properties: the exact solution is a fixed point, a
 * triangular matrix is solved by one sweep, preconditioned CG needs fewer iterations).  The definition is the textbook
 * one, rows in the matrix's own order, entries of a row left to right, duplicates of the diagonal entry summed:
 *   forward  i = 0..n-1, then backward i = n-1..0:   x_i = (b_i - sum_{j != i} a_ij x_j) / a_ii   with the newest x.
 * Returns 0, or 1 + the first row without a diagonal (x is then untouched from that row of the first sweep on). */
int32_t orc_symgs(int32_t n, const int32_t* row_ptr, const int32_t* col, const double* val, const double* b, double* x,
                  int32_t sweeps);
/* the same sweep over the rows in the sequence order[0..n-1] (forward) and back (order == NULL: 0..n-1) */
int32_t orc_symgs_ordered(int32_t n, const int32_t* row_ptr, const int32_t* col, const double* val, const double* b,
                          double* x, int32_t sweeps, const int32_t* order);
/* the engine's default sequence: greedy colouring in row order (colour[i] = smallest colour no coupled row j < i has),
 * rows by (colour, row).  Returns the number of colours. */
int32_t orc_greedy_colour_order(int32_t n, const int32_t* row_ptr, const int32_t* col, int32_t* colour, int32_t* order);

#ifdef __cplusplus
}
#endif
#endif
