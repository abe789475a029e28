/*
 * spmv_oracle.c — CPU restatement of the arm-spmv hot path.  TEST INFRASTRUCTURE ONLY
 * (see spmv_oracle.h: never part of the product path).
 *
 * Compiled twice by oracle/Makefile:
 *   plain   : -ffp-contract=off            -> orc_<name>       (+ conversions, sharding)
 *   -DORC_FMA -mfma                        -> orc_<name>_fma   (arithmetic loops only)
 * Each function cites the reference loop it follows (paths relative to the reference root).
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* pthread_setaffinity_np, CPU_SET */
#endif
#include "spmv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORC_FMA
#define ORC_NAME(n) n##_fma
#define MULADD(acc, a, b) fma((a), (b), (acc))
#else
#define ORC_NAME(n) n
#define MULADD(acc, a, b) ((acc) + (a) * (b))
#endif

/* src/mat_vec.cpp:32-40 */
void ORC_NAME(orc_coo_spmv)(int64_t nnz, const int32_t* row, const int32_t* col,
                            const double* val, const double* x, double* y)
{
    for (int64_t i = 0; i < nnz; i++)
        y[row[i]] = MULADD(y[row[i]], val[i], x[col[i]]);
}

/* src/mat_vec.cpp:57-65 */
void ORC_NAME(orc_csr_spmv)(int32_t nrow, const int32_t* row_ptr, const int32_t* col,
                            const double* val, const double* x, double* y)
{
    for (int32_t i = 0; i < nrow; i++)
    {
        double sum = 0.0;
        for (int32_t j = row_ptr[i]; j < row_ptr[i + 1]; j++)
            sum = MULADD(sum, val[j], x[col[j]]);
        y[i] += sum;
    }
}

/* src/mat_vec.cpp:82-93: `value = val*x` in one statement, `y[row] += value` in the next.  The plain flavour
 * (separate multiply and add: x86-64, and any build where the add is an `omp atomic`) is the pinned one; the _fma
 * flavour fuses the two, which is what g++ -O2 may emit for the serial loop on aarch64 (-ffp-contract=fast works
 * on the whole expression tree, across the two statements) and what the engine's row-grouped CSC path computes. */
void ORC_NAME(orc_csc_spmv)(int32_t ncol, const int32_t* col_ptr, const int32_t* row,
                            const double* val, const double* x, double* y)
{
    for (int32_t i = 0; i < ncol; i++)
        for (int32_t j = col_ptr[i]; j < col_ptr[i + 1]; j++)
            y[row[j]] = MULADD(y[row[j]], val[j], x[i]);
}

/* src/mat_vec.cpp:107-118 */
void ORC_NAME(orc_ell_spmv)(int32_t nrow, int32_t k, const int32_t* col, const double* val,
                            const double* x, double* y)
{
    for (int32_t s = 0; s < k; s++)
        for (int32_t i = 0; i < nrow; i++)
        {
            size_t at = (size_t)i + (size_t)s * (size_t)nrow;
            y[i]      = MULADD(y[i], val[at], x[col[at]]);
        }
}

/* src/mat_vec.cpp:135-145 */
void ORC_NAME(orc_dia_spmv)(int32_t nrow, int32_t ndiags, const int32_t* offsets,
                            const double* val, const double* x, double* y)
{
    for (int32_t i = 0; i < nrow; ++i)
        for (int32_t d = 0; d < ndiags; ++d)
        {
            int32_t j = i + offsets[d];
            if (j >= 0 && j < nrow)
                y[i] = MULADD(y[i], val[(size_t)i * ndiags + d], x[j]);
        }
}

/* src/vec_vec.cpp:21-28 (serial order) */
double ORC_NAME(orc_dot)(int64_t n, const double* x, const double* y)
{
    double result = 0.0;
    for (int64_t i = 0; i < n; ++i)
        result = MULADD(result, x[i], y[i]);
    return result;
}

/* src/vec_vec.cpp:38-93: the seven branches; alpha==0 never reads x, beta==0 never reads y */
void ORC_NAME(orc_axpby)(int64_t n, double alpha, const double* x, double beta, const double* y,
                         double* w)
{
    if (alpha == 0)
        for (int64_t i = 0; i < n; ++i) w[i] = beta * y[i];
    else if (beta == 0)
        for (int64_t i = 0; i < n; ++i) w[i] = alpha * x[i];
    else if (alpha == 1)
        for (int64_t i = 0; i < n; ++i) w[i] = MULADD(x[i], beta, y[i]);
    else if (alpha == -1)
        for (int64_t i = 0; i < n; ++i) w[i] = MULADD(-x[i], beta, y[i]);
    else if (beta == 1)
        for (int64_t i = 0; i < n; ++i) w[i] = MULADD(y[i], alpha, x[i]);
    else if (beta == -1)
        for (int64_t i = 0; i < n; ++i) w[i] = MULADD(-y[i], alpha, x[i]);
    else
#ifdef ORC_FMA
        /* aarch64 g++ -O2 contracts a*x + b*y to fmadd(a, x, b*y) */
        for (int64_t i = 0; i < n; ++i) w[i] = fma(alpha, x[i], beta * y[i]);
#else
        for (int64_t i = 0; i < n; ++i) w[i] = alpha * x[i] + beta * y[i];
#endif
}

#ifndef ORC_FMA
/* ======================= everything below: integer work, one flavour ===================== */

/* src/mat_vec.cpp:54-65 with the OpenMP pragma kept (cpu_baseline "port" leg) */
void orc_csr_spmv_omp(int32_t nrow, const int32_t* row_ptr, const int32_t* col,
                      const double* val, const double* x, double* y)
{
#pragma omp parallel for
    for (int32_t i = 0; i < nrow; i++)
    {
        double sum = 0.0;
        for (int32_t j = row_ptr[i]; j < row_ptr[i + 1]; j++)
            sum += val[j] * x[col[j]];
        y[i] += sum;
    }
}

/* First-touch placement for the OpenMP baseline ("first-touch initialisation"): copies a CSR
 * matrix and its vectors into destination arrays the caller has allocated but never written, inside an OpenMP team with
 * the SAME static row schedule as the product loop above (src/mat_vec.cpp:54-57: `omp parallel for` over rows, default
 * static schedule) — so every page of a row's entries and of its y lands on the NUMA node of the thread that will
 * multiply that row.  x is gathered from by every thread: its pages are spread over the team in equal stretches. */
void orc_csr_first_touch_copy(int32_t nrow, int64_t ncol, const int32_t* row_ptr, const int32_t* col, const double* val,
                              const double* x, int32_t* d_row_ptr, int32_t* d_col, double* d_val, double* d_x, double* d_y)
{
#pragma omp parallel
    {
#pragma omp for schedule(static) nowait
        for (int32_t i = 0; i < nrow; i++)
        {
            d_row_ptr[i] = row_ptr[i];
            for (int32_t j = row_ptr[i]; j < row_ptr[i + 1]; j++)
            {
                d_col[j] = col[j];
                d_val[j] = val[j];
            }
            d_y[i] = 0.0;
        }
#pragma omp for schedule(static)
        for (int64_t c = 0; c < ncol; c++) d_x[c] = x[c];
    }
    d_row_ptr[nrow] = row_ptr[nrow];
}

/* src/mat_vec.cpp:230-297 + :507-530, the NUMA driver's protocol with PERSISTENT workers (cpu_baseline "port" leg,
 * BASELINE.md section 4): equal-row shards (last takes the remainder, :245-246), every shard a private copy of its
 * rebased row_ptr (:260-263), its col/val slices and a FULL replica of x (:257,:266), a local y slice (:258,:267) —
 * all allocated and first-touched by the worker that owns them, which is pinned to one of the allowed CPUs (the
 * reference pins to a NUMA node with numa_run_on_node, :511).  Differences from the reference, both deliberate: the
 * workers live across the repetitions (the reference re-creates its pthreads in every one, :274-281, and measures
 * mostly that), and the y slices are copied back (the reference drops them, :287-296).
 * Returns milliseconds per repetition (reps timed between two barriers), < 0 on failure. */
#include <pthread.h>
#include <sched.h>
#include <time.h>

typedef struct
{
    int                shard, nshards, reps, cpu;
    int32_t            nrow, ncol, r0, r1;
    const int32_t *    row_ptr, *col;
    const double *     val, *x;
    double*            y;
    pthread_barrier_t* bar;
    int                failed;
} orc_shard_job;

static void* orc_shard_worker(void* arg)
{
    orc_shard_job* j = (orc_shard_job*)arg;
    if (j->cpu >= 0)
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(j->cpu, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    const int32_t rows = j->r1 - j->r0;
    const int64_t e0 = j->row_ptr[j->r0], e1 = j->row_ptr[j->r1];
    int32_t* rp  = (int32_t*)malloc(sizeof(int32_t) * ((size_t)rows + 1));
    int32_t* col = (int32_t*)malloc(sizeof(int32_t) * (size_t)(e1 - e0 > 0 ? e1 - e0 : 1));
    double*  val = (double*)malloc(sizeof(double) * (size_t)(e1 - e0 > 0 ? e1 - e0 : 1));
    double*  x   = (double*)malloc(sizeof(double) * (size_t)(j->ncol > 0 ? j->ncol : 1));
    double*  y   = (double*)malloc(sizeof(double) * (size_t)(rows > 0 ? rows : 1));
    j->failed    = !(rp && col && val && x && y);
    if (!j->failed)
    {
        for (int32_t i = 0; i <= rows; ++i) rp[i] = (int32_t)(j->row_ptr[j->r0 + i] - e0);
        memcpy(col, j->col + e0, sizeof(int32_t) * (size_t)(e1 - e0));
        memcpy(val, j->val + e0, sizeof(double) * (size_t)(e1 - e0));
        memcpy(x, j->x, sizeof(double) * (size_t)j->ncol);
        memset(y, 0, sizeof(double) * (size_t)rows);
    }
    pthread_barrier_wait(j->bar); /* shards built */
    pthread_barrier_wait(j->bar); /* clock started by the caller */
    if (!j->failed)
        for (int r = 0; r < j->reps; ++r)
            for (int32_t i = 0; i < rows; ++i) /* src/mat_vec.cpp:517-527 */
            {
                double sum = 0.0;
                for (int32_t k = rp[i]; k < rp[i + 1]; ++k) sum += val[k] * x[col[k]];
                y[i] += sum;
            }
    pthread_barrier_wait(j->bar); /* all repetitions done */
    if (!j->failed)
        for (int32_t i = 0; i < rows; ++i) j->y[j->r0 + i] += y[i];
    free(rp);
    free(col);
    free(val);
    free(x);
    free(y);
    return 0;
}

double orc_csr_spmv_sharded(int32_t nrow, int32_t ncol, const int32_t* row_ptr, const int32_t* col, const double* val,
                            const double* x, double* y, int32_t nshards, int32_t reps)
{
    if (nshards < 1 || reps < 1) return -1.0;
    cpu_set_t allowed;
    int       cpus[1024], ncpu = 0;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE && ncpu < 1024; ++c)
            if (CPU_ISSET(c, &allowed)) cpus[ncpu++] = c;
    pthread_barrier_t bar;
    if (pthread_barrier_init(&bar, 0, (unsigned)nshards + 1) != 0) return -1.0;
    orc_shard_job* jobs = (orc_shard_job*)calloc((size_t)nshards, sizeof(orc_shard_job));
    pthread_t*     thr  = (pthread_t*)calloc((size_t)nshards, sizeof(pthread_t));
    const int32_t  per  = nrow / nshards;
    int            started = 0;
    for (int s = 0; jobs && thr && s < nshards; ++s)
    {
        orc_shard_job* j = &jobs[s];
        j->shard = s, j->nshards = nshards, j->reps = reps, j->cpu = ncpu ? cpus[s % ncpu] : -1;
        j->nrow = nrow, j->ncol = ncol, j->r0 = s * per, j->r1 = s == nshards - 1 ? nrow : (s + 1) * per;
        j->row_ptr = row_ptr, j->col = col, j->val = val, j->x = x, j->y = y, j->bar = &bar;
        if (pthread_create(&thr[s], 0, orc_shard_worker, j) != 0) break;
        ++started;
    }
    double ms = -1.0;
    if (started == nshards)
    {
        struct timespec t0, t1;
        pthread_barrier_wait(&bar);
        clock_gettime(CLOCK_MONOTONIC, &t0);
        pthread_barrier_wait(&bar);
        pthread_barrier_wait(&bar);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        ms = ((double)(t1.tv_sec - t0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-6) / reps;
        for (int s = 0; s < nshards; ++s)
        {
            pthread_join(thr[s], 0);
            if (jobs[s].failed) ms = -1.0;
        }
    }
    /* (a failed pthread_create leaves the started workers at the first barrier: not recoverable here, and
     * never seen with the handful of shards the baseline uses) */
    pthread_barrier_destroy(&bar);
    free(jobs);
    free(thr);
    return ms;
}

/* src/matrix.cpp:125-153 */
int32_t orc_coo_to_csr(int32_t nrow, int64_t nnz, const int32_t* row, const int32_t* col,
                       const double* val, int32_t* row_ptr, int32_t* out_col, double* out_val,
                       double* diagonal)
{
    for (int32_t i = 0; i <= nrow; ++i) row_ptr[i] = 0;
    for (int64_t k = 0; k < nnz; ++k) ++row_ptr[row[k]];
    for (int32_t i = 0; i < nrow; ++i) row_ptr[i + 1] += row_ptr[i];
    for (int64_t k = nnz - 1; k >= 0; --k)
    {
        int32_t at  = --row_ptr[row[k]];
        out_col[at] = col[k];
        out_val[at] = val[k];
    }
    int32_t count = 0;
    if (diagonal)
        for (int64_t k = 0; k < nnz; ++k)
            if (row[k] == col[k]) diagonal[count++] = val[k];
    return count;
}

/* src/matrix.cpp:305-324 */
void orc_coo_to_csc(int32_t ncol, int64_t nnz, const int32_t* row, const int32_t* col,
                    const double* val, int32_t* col_ptr, int32_t* out_row, double* out_val)
{
    for (int32_t j = 0; j <= ncol; ++j) col_ptr[j] = 0;
    for (int64_t k = 0; k < nnz; ++k) ++col_ptr[col[k]];
    for (int32_t j = 0; j < ncol; ++j) col_ptr[j + 1] += col_ptr[j];
    for (int64_t k = nnz - 1; k >= 0; --k)
    {
        int32_t at  = --col_ptr[col[k]];
        out_row[at] = row[k];
        out_val[at] = val[k];
    }
}

/* src/matrix.cpp:456-470 */
int32_t orc_coo_max_row_nnz(int32_t nrow, int64_t nnz, const int32_t* row)
{
    int32_t* cnt = (int32_t*)calloc((size_t)nrow > 0 ? (size_t)nrow : 1, sizeof(int32_t));
    for (int64_t k = 0; k < nnz; ++k) cnt[row[k]]++;
    int32_t mx = 0;
    for (int32_t i = 0; i < nrow; ++i) mx = cnt[i] > mx ? cnt[i] : mx;
    free(cnt);
    return mx;
}

/* src/matrix.cpp:472-489 */
void orc_coo_to_ell(int32_t nrow, int32_t k, int64_t nnz, const int32_t* row,
                    const int32_t* col, const double* val, int32_t* out_col, double* out_val)
{
    size_t   total = (size_t)nrow * (size_t)k;
    int32_t* cnt   = (int32_t*)calloc((size_t)nrow > 0 ? (size_t)nrow : 1, sizeof(int32_t));
    for (int64_t e = 0; e < nnz; ++e) cnt[row[e]]++;
    memset(out_col, 0, total * sizeof(int32_t));
    for (size_t e = 0; e < total; ++e) out_val[e] = 0.0;
    for (int64_t e = nnz - 1; e >= 0; --e)
    {
        int32_t r  = row[e];
        int32_t s  = --cnt[r];
        size_t  at = (size_t)r + (size_t)s * (size_t)nrow;
        out_col[at] = col[e];
        out_val[at] = val[e];
    }
    free(cnt);
}

/* src/matrix.cpp:675-709: map index (nrow - i) + j, offsets[d] = n - nrow, ascending */
int32_t orc_csr_count_diags(int32_t nrow, int32_t ncol, const int32_t* row_ptr,
                            const int32_t* col, int32_t* offsets)
{
    size_t   span = (size_t)nrow + (size_t)ncol; /* reference: nrow+ncol-1 (one short) */
    int32_t* map  = (int32_t*)calloc(span, sizeof(int32_t));
    int32_t  nd   = 0;
    for (int32_t i = 0; i < nrow; ++i)
        for (int32_t jj = row_ptr[i]; jj < row_ptr[i + 1]; ++jj)
        {
            size_t at = (size_t)(nrow - i) + (size_t)col[jj];
            if (!map[at])
            {
                map[at] = 1;
                nd++;
            }
        }
    if (offsets)
    {
        int32_t d = 0;
        for (size_t n = 0; n < span; ++n)
            if (map[n]) offsets[d++] = (int32_t)n - nrow;
    }
    free(map);
    return nd;
}

/* src/matrix.cpp:711-723 */
void orc_csr_to_dia(int32_t nrow, int32_t ncol, const int32_t* row_ptr, const int32_t* col,
                    const double* val, int32_t ndiags, const int32_t* offsets, double* out_val)
{
    size_t   span = (size_t)nrow + (size_t)ncol;
    int32_t* map  = (int32_t*)calloc(span, sizeof(int32_t));
    for (int32_t d = 0; d < ndiags; ++d) map[(size_t)(offsets[d] + nrow)] = d;
    for (size_t e = 0; e < (size_t)nrow * (size_t)ndiags; ++e) out_val[e] = 0.0;
    for (int32_t i = 0; i < nrow; ++i)
        for (int32_t jj = row_ptr[i]; jj < row_ptr[i + 1]; ++jj)
        {
            size_t at = (size_t)(nrow - i) + (size_t)col[jj];
            out_val[(size_t)i * ndiags + map[at]] = val[jj];
        }
    free(map);
}

/* src/mat_vec.cpp:233,245-246 */
void orc_partition_rows(int64_t nrow, int32_t nparts, int32_t part, int64_t* begin,
                        int64_t* end)
{
    int64_t per = nrow / nparts;
    *begin      = (int64_t)part * per;
    *end        = (part == nparts - 1) ? nrow : *begin + per;
}

/* src/mat_vec.cpp:260-263 */
void orc_csr_shard_row_ptr(const int32_t* row_ptr, int64_t begin, int64_t end,
                           int32_t* sub_row_ptr)
{
    int32_t base = row_ptr[begin];
    for (int64_t j = 0; j <= end - begin; j++) sub_row_ptr[j] = row_ptr[begin + j] - base;
}

void orc_csr_abs_row_sums(int32_t nrow, const int32_t* row_ptr, const int32_t* col,
                          const double* val, const double* x, double* s)
{
    for (int32_t i = 0; i < nrow; i++)
    {
        double acc = 0.0;
        for (int32_t j = row_ptr[i]; j < row_ptr[i + 1]; j++)
            acc += fabs(val[j]) * fabs(x[col[j]]);
        s[i] = acc;
    }
}

/* no reference loop (see spmv_oracle.h): the textbook symmetric Gauss-Seidel sweep in row order */
static int32_t symgs_row(int32_t i, const int32_t* row_ptr, const int32_t* col, const double* val, const double* b, double* x)
{
    double sum = b[i], diag = 0.0;
    for (int32_t j = row_ptr[i]; j < row_ptr[i + 1]; j++)
    {
        if (col[j] == i)
            diag += val[j];
        else
            sum -= val[j] * x[col[j]];
    }
    if (diag == 0.0) return 1;
    x[i] = sum / diag;
    return 0;
}

int32_t orc_symgs_ordered(int32_t n, const int32_t* row_ptr, const int32_t* col, const double* val, const double* b,
                          double* x, int32_t sweeps, const int32_t* order)
{
    for (int32_t s = 0; s < sweeps; s++)
    {
        for (int32_t k = 0; k < n; k++)
        {
            const int32_t i = order ? order[k] : k;
            if (symgs_row(i, row_ptr, col, val, b, x)) return 1 + i;
        }
        for (int32_t k = n - 1; k >= 0; k--)
        {
            const int32_t i = order ? order[k] : k;
            if (symgs_row(i, row_ptr, col, val, b, x)) return 1 + i;
        }
    }
    return 0;
}

int32_t orc_symgs(int32_t n, const int32_t* row_ptr, const int32_t* col, const double* val, const double* b, double* x,
                  int32_t sweeps)
{
    return orc_symgs_ordered(n, row_ptr, col, val, b, x, sweeps, 0);
}

/* colour[i] = smallest colour none of the rows j < i coupled to i (a_ij stored) has: the sequential greedy colouring
 * the engine's multicolour sweep order is defined by; order[] = rows by (colour, row).  Returns the number of colours. */
int32_t orc_greedy_colour_order(int32_t n, const int32_t* row_ptr, const int32_t* col, int32_t* colour, int32_t* order)
{
    int32_t ncolours = 0;
    for (int32_t i = 0; i < n; i++)
    {
        int32_t c = 0;
        for (;;)
        {
            int32_t clash = 0;
            for (int32_t j = row_ptr[i]; j < row_ptr[i + 1] && !clash; j++)
                if (col[j] < i && colour[col[j]] == c) clash = 1;
            if (!clash) break;
            c++;
        }
        colour[i] = c;
        if (c + 1 > ncolours) ncolours = c + 1;
    }
    int32_t k = 0;
    for (int32_t c = 0; c < ncolours; c++)
        for (int32_t i = 0; i < n; i++)
            if (colour[i] == c) order[k++] = i;
    return ncolours;
}
#endif /* !ORC_FMA */
