/*
 * spmv_abi.h — the drop-in boundary of the MI355X SpMV engine (libspmv_hip.so).
 *
 * Plain C ABI: opaque handles, raw pointers, sizes.  No C++ types, no torch types.  Everything
 * the reference's hot path does (`y += A*x` for its storage formats, the format conversions, the
 * BLAS-1 helpers and the row-range sharding of its NUMA drivers) is reachable from here; the
 * C++ classes of the reference (include/matrix.h, include/vector.h, include/mat_vec.h in the
 * reference tree) are re-created as a thin source-compatible shim over this ABI in
 * include/arm_spmv_compat.hpp, so the reference's main.cpp recompiles unchanged.
 *
 * Conventions
 *   - every function returns 0 on success, a negative spmv_status otherwise; the message of the
 *     last failure on the calling thread is spmv_last_error().  (The reference has no error
 *     convention: ops return void, I/O failures printf + exit(1), src/data_io.cpp:53-75.)
 *   - values are fp64, indices int32 (as in the reference, include/matrix.h:9-16); counts that can
 *     exceed 2^31 across shards (nnz) are int64 in this ABI.
 *   - semantics are ACCUMULATING: spmv_apply computes y += A*x (src/mat_vec.cpp:39,64,91,116,142);
 *     the caller zeroes y (main.cpp:55,65,85).
 *   - host arrays passed to *_upload stay owned by the caller and may be freed on return; device
 *     pointers passed to *_wrap_device are borrowed and must outlive the handle.
 *   - work is queued on the context's HIP stream and is asynchronous; spmv_sync() waits.
 *   - threading: like the reference's entry points (synchronous, one caller), a context and the
 *     handles made from it are driven by ONE host thread at a time; different contexts (e.g. one
 *     per GPU) may be driven by different threads concurrently.  A matrix handle carries run-time
 *     state (scratch of the two-phase kernel, the pace guard of the panel kernel), so do not apply the
 *     same handle from two streams at once.
 *   - the library is HIP-only.  There is no CPU fallback: without a usable GPU spmv_ctx_create
 *     fails with SPMV_ERR_NO_DEVICE and nothing else can be called.
 */
#ifndef SPMV_ABI_H
#define SPMV_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPMV_ABI_VERSION 1

typedef enum spmv_status
{
    SPMV_OK             = 0,
    SPMV_ERR_INVALID    = -1, /* bad argument (null handle, negative size, shape mismatch) */
    SPMV_ERR_NO_DEVICE  = -2, /* no HIP device / device index out of range */
    SPMV_ERR_HIP        = -3, /* a HIP runtime call failed (message has hipGetErrorString) */
    SPMV_ERR_ALLOC      = -4, /* device or host allocation failed */
    SPMV_ERR_UNSUPPORTED = -5 /* valid request the engine does not implement */
} spmv_status;

typedef enum spmv_format
{
    SPMV_FMT_COO = 0, /* include/matrix.h:7-25  (reference) */
    SPMV_FMT_CSR = 1, /* include/matrix.h:27-47 */
    SPMV_FMT_CSC = 2, /* include/matrix.h:49-68 */
    SPMV_FMT_ELL = 3, /* include/matrix.h:70-92, column-major: (row i, slot k) at i + k*nrow */
    SPMV_FMT_DIA = 4  /* include/matrix.h:117-138, row-major: (row i, diag d) at i*ndiags + d */
} spmv_format;

/* CSR kernel selection (spmv_mat_set_kernel).  AUTO (csrc/select.hip) is decided when the handle is
 * created - one-off, outside every timed region, like the reference's shard construction (src/mat_vec.cpp:240-268):
 *   the model   statistics of the matrix: fewer than 1.5M entries -> VECTOR (or LDSWIN for narrow bands; PANEL when a few hub
 *               rows dwarf the mean: they would serialise on one lane group); larger -> PANEL, or TWOPHASE when the sweeps of x
 *               the panel kernel would make (8 XCDs x rounds x 8 * ncol bytes) outweigh the 16 extra bytes per entry of the two
 *               phases, or VECTOR when the rows are long contiguous runs (dense blocks of 32 and more: x is read coalesced);
 *   the trial   handles of 64K to 8M entries - where a product takes microseconds and the candidates lie within a factor of a
 *               few - TIME their candidates (PANEL, VECTOR, LDSWIN where the windows fit, SCALAR for short even rows, the
 *               model's pick): 1 warm-up + 2 x 4 products each on zeroed scratch vectors (8 * (nrow + ncol) bytes, freed
 *               before the call returns), a candidate 3x behind after its first product is dropped at once, the fastest
 *               stays, layouts built for the others are freed.  Larger handles keep the model's pick without a launch, unless
 *               a statistic casts doubt on it: rows that are contiguous runs (then VECTOR and PANEL are timed), the model's
 *               TWOPHASE below 64M entries (PANEL is timed beside it).  At ANY size: a longest row with 1/128 of the entries
 *               adds SEGSCAN, rows 32x the mean (and >= 4096; from 8M entries on >= 1/512 of the entries) add SPLIT - from 8M
 *               entries on at two thresholds -, (nearly) equal rows with local columns add ELL (the kernels' own comments
 *               below).  The panel layout itself times a cut for more rounds of workgroups where skewed rows leave its
 *               busiest row group far above the mean ("panel_rounds").  COO and ELL handles time their own
 *               kernel(s) against a row-grouped CSR copy of themselves - which picks ITS kernel the same way - where that copy
 *               is a candidate (below).  "panel_trial" 0 / SPMV_PANEL_TRIAL=0: no timing launch anywhere, the model alone.
 *   what was timed is on record: spmv_mat_get_param "select_candidates", "select_us_vector" / "_ldswin" / "_scalar" / "_panel" /
 *   "_twophase" (microseconds per product, 0 = not timed; COO / ELL: "_vector" the format's own kernel, "_panel" the copy; ELL
 *   also "_variant1" one row per lane, "_variant2" two rows per lane reading every index), "rowgrouped_kernel" (the kernel the
 *   copy of a COO / ELL / CSC handle runs, 0 = no copy in use), "ell_variant", "contiguous_permille"; CSR handles also
 *   "select_us_segscan" / "_split" / "_split_low" / "_ell", "min_row_entries", for kernel ELL "ell_copy_slots",
 *   "ell_copy_diagonal_slots", "ell_copy_variant", and for kernel SPLIT "split_row_threshold" (get: in effect; set: 0 = default, read at the
 *   next spmv_mat_set_kernel), "split_mode" (likewise; get: the mode in effect), "split_long_rows", "split_long_entries",
 *   "split_inner_kernel" (what the short rows' copy runs), "split_long_kernel" / "split_virtual_rows" (mode 2).
 *   Audit on stencils, dense blocks, R-MAT graphs, rectangles, permutations, arrow and dense-row shapes, power laws, bands and meshes:
 *   tools/sweep_structures.py, tools/fuzz_medium.py, profiles/r05_sweep_structures_*, profiles/r05_fuzz_medium_sizes.txt.
 *
 * Order of the additions (all within the parity tolerance of 1e-10):
 *   SCALAR              the reference's own order (left to right inside a row): bit-identical to its fma flavour;
 *   VECTOR, LDSWIN      a fixed tree per row: deterministic, the same bits on every call;
 *   PANEL, TWOPHASE     products are added into per-row accumulators in LDS with ds_add_f64 in ARRIVAL order: two calls
 *                       on the same data may differ in the last bits (the reference's CSR loop is deterministic per row,
 *                       src/mat_vec.cpp:57-65; its COO and CSC loops are not: `omp atomic`, :36-39, :88-91).  Callers that
 *                       need run-to-run identical bits select VECTOR (spmv_mat_set_kernel(A, SPMV_CSR_VECTOR, 0));
 *   SEGSCAN             a fixed tree inside a wavefront's 512 entries; rows that cross into another wavefront's entries are joined by
 *                       atomic adds on y in arrival order (like the COO scan);
 *   ELL                 the reference's ELL order (slot after slot into the row's sum): deterministic;
 *   SPLIT               the long rows: mode 1 a fixed tree per chunk of 4096 entries, chunks joined by atomic adds in arrival order;
 *                       mode 2 as the virtual rows' kernel, then a fixed order over a row's partial sums; the other rows: as the
 *                       inner kernel. */
typedef enum spmv_csr_kernel
{
    SPMV_CSR_AUTO     = 0,
    SPMV_CSR_VECTOR   = 1, /* 2^k lanes per row, ds_swizzle/DPP segment reduction */
    SPMV_CSR_LDSWIN   = 2, /* row blocks whose x window is staged in LDS (banded matrices) */
    SPMV_CSR_SCALAR   = 3, /* one lane per row, strictly left-to-right (bitwise = oracle _fma) */
    SPMV_CSR_PANEL    = 4, /* row groups x column panels: x gathered from L2, y accumulated in LDS */
    SPMV_CSR_TWOPHASE = 5, /* x-stationary expand + y-stationary reduce (x far larger than the rows held: C5 shards) */
    SPMV_CSR_SEGSCAN  = 6, /* the entries in row order, 512 per wavefront whatever row they belong to, joined by a segmented scan
                              (the COO kernel over a row index per entry, 4 bytes per entry on top of the CSR arrays): a matrix whose
                              longest rows hold a large share of the entries - arrow shapes, a few dense rows - where every other CSR
                              kernel leaves that row to ONE wavefront or workgroup (1M entries in one row: 1.26 ms there, 0.045 here).
                              CSR handles only; AUTO times it where the longest row exceeds 1/128 of the entries */
    SPMV_CSR_SPLIT    = 7, /* the rows of "split_row_threshold" entries and more (default: a sixteenth of the longest row, at least
                              4096) leave the matrix; every other row goes into a copy without them, which picks its own kernel
                              ("split_inner_kernel"; the panel layout as a rule).  The long rows ("split_mode": 0 by their density):
                              1 = in chunks of 4096 entries over the handle's own arrays, one workgroup and one atomic add on y each
                              (dense rows: an entry per 2 columns); 2 = dealt out to virtual rows of 64 entries (entry j of a row to
                              virtual row j mod V) that form a CSR matrix with a handle and a kernel of its own ("split_long_kernel",
                              "split_virtual_rows"), its product summed per long row in a fixed order (long AND sparse rows: power
                              laws, graph hubs).  1M rows x 32 + one dense row: panel 1.26 ms, scan 0.47, split 0.11; 1M rows of
                              min(500000, 8/u) entries: panel 0.64, split 0.29.  CSR handles only; AUTO times it where the longest
                              row is >= 4096 and 32x the mean */
    SPMV_CSR_ELL      = 8  /* an ELL copy of the handle (column-major slots, padded to the longest row with value 0.0 and the row's own
                              last column; the copy's kernels leave that padding OUT of the sums - its slots are marked in the
                              copy's index stream - so the copy is the CSR matrix in non-finite arithmetic too) with the ELL kernels -
                              diagonal slots recognised, one or two rows per lane, the DIA-order copy, timed - for
                              matrices of (nearly) equal rows: stencils, bands, block diagonals (tridiagonal, 8M rows: panel 0.087
                              ms, this 0.056; a band of 33: 0.163 / 0.105).  12 bytes per slot on top of the CSR arrays.  CSR
                              handles only; AUTO times it where the padding stays below a quarter, no row is empty and the columns are local
                              (the x window of 256 rows within 2 MB, or all of x within 4 MB) */
} spmv_csr_kernel;

/* Tuning bits for spmv_mat_set_flags (speed only; results stay within the parity tolerance). */
#define SPMV_FLAG_DPP_REDUCE 1u /* CSR vector kernel: DPP row shifts instead of ds_swizzle for the <=16-lane steps */
#define SPMV_FLAG_XCD_REMAP  2u /* CSR vector kernel: each XCD walks one contiguous eighth of the rows */
#define SPMV_FLAG_ELL_READ_COLUMNS 8u /* ELL kernel: read the column indices even where the slots were found to be diagonals (A/B switch) */
#define SPMV_FLAG_DIA_GLOBAL_X 4u /* DIA kernel: read x from global memory even when the offsets lie in a band (A/B switch) */

typedef struct spmv_ctx spmv_ctx; /* one HIP device + one stream */
typedef struct spmv_vec spmv_vec; /* device-resident fp64 vector         (reference: class Vector) */
typedef struct spmv_mat spmv_mat; /* device-resident sparse matrix/shard (reference: XMatrix) */

typedef struct spmv_mat_info
{
    int32_t format;   /* spmv_format */
    int32_t nrow;     /* rows held by this handle (a shard holds its own rows only) */
    int32_t ncol;     /* columns = length of x (always global: x is a full replica, mat_vec.cpp:257) */
    int32_t ell_k;    /* ELL: slots per row (nonzeros_in_row); DIA: ndiags; else 0 */
    int64_t nnz;      /* stored nonzeros (ELL: true nnz given at creation, not nrow*k) */
    int64_t row_begin; /* first global row of this shard (0 for an unsharded matrix) */
    int32_t max_row_nnz;
    int32_t kernel;   /* CSR: the spmv_csr_kernel in effect.  COO / ELL / CSC: 1 = the format's own kernel (segmented scan / lanes
                         over rows / scatter over columns), 4 = the product runs from the handle's row-grouped CSR copy, whichever
                         kernel that copy runs (spmv_mat_get_param "rowgrouped_kernel") */
    int32_t lanes_per_row; /* CSR vector kernel: lanes cooperating on one row */
    int32_t sorted_rows;   /* COO: 1 if row indices are non-decreasing */
    int64_t device_bytes;  /* bytes of device memory owned by the handle */
} spmv_mat_info;

/* ---- library / context ---------------------------------------------------------------------- */
int         spmv_abi_version(void);
const char* spmv_last_error(void);
int         spmv_device_count(int* count);
/* Creates a context on `device` with its own non-blocking stream. */
int spmv_ctx_create(int device, spmv_ctx** out);
/* Same, but queues all work on a caller-owned hipStream_t (e.g. torch's current stream). */
int spmv_ctx_create_on_stream(int device, void* hip_stream, spmv_ctx** out);
int spmv_ctx_destroy(spmv_ctx* ctx);
int spmv_sync(spmv_ctx* ctx);
int spmv_ctx_device(const spmv_ctx* ctx, int* device);
/* The context's start-up probe of workgroup placement: *round_robin = 1 when the workgroups of a launch are dealt round-robin
 * over 8 XCDs (b and b + 8 share an XCD and its L2; read from the XCC_ID hardware register by 2048 workgroups when the context
 * was created), 0 when not (a partitioned device, another dispatch order), -1 when the probe could not run; *xcds_seen = distinct
 * XCD ids seen.  HIP promises no placement: layouts that lean on it for speed - the COO scan over one column bin per XCD
 * ("coo_column_bins") - are built only when it is 1 (otherwise the scan runs in place and the library says so once on stderr);
 * nothing leans on it for correctness.  Either pointer may be NULL. */
int spmv_ctx_xcd_round_robin(spmv_ctx* ctx, int32_t* round_robin, int32_t* xcds_seen);
/* What a context found out about its platform when it was created (read-only):
 *   "host_stores"      1 = spmv_apply_host stores a small x straight into device memory from the CPU (large BAR reported, a
 *                      self-check passed at creation - CPU stores summed by a kernel, twice - and SPMV_HOST_STORES is not 0)
 *   "xcd_round_robin", "xcds_seen"   as spmv_ctx_xcd_round_robin
 *   "trial_arena_bytes"  bytes of the arena that timing launches of AUTO take their scratch vectors from (0 until a trial ran) */
int spmv_ctx_get_param(const spmv_ctx* ctx, const char* name, int64_t* value);
/* free and total device memory in bytes (sizing shards for 288 GB of HBM; checking that handles give memory back) */
int spmv_ctx_mem_info(spmv_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes);

/* ---- vectors  (reference: include/vector.h:4-26 {int size; double* values}) --------------------- */
int spmv_vec_create(spmv_ctx* ctx, int64_t n, spmv_vec** out);
int spmv_vec_wrap_device(spmv_ctx* ctx, int64_t n, double* device_ptr, spmv_vec** out);
int spmv_vec_destroy(spmv_vec* v);
int spmv_vec_size(const spmv_vec* v, int64_t* n);
int spmv_vec_device_ptr(const spmv_vec* v, double** device_ptr);
/* copy host[0..n) -> v[offset..offset+n)  /  v[offset..offset+n) -> host[0..n); both synchronous */
int spmv_vec_upload(spmv_vec* v, int64_t offset, int64_t n, const double* host);
int spmv_vec_download(const spmv_vec* v, int64_t offset, int64_t n, double* host);
int spmv_vec_fill(spmv_vec* v, double a); /* Vector::Fill, src/vector.cpp:59-63 */
/* dst[dst_offset..+n) = src[src_offset..+n) between vectors of ANY two contexts (same GPU or peers over xGMI), queued on
 * the destination's stream behind the work already queued on the source's.  The sharded drivers assemble y with it
 * (the reference leaves the per-thread Y slices where they are, src/mat_vec.cpp:287-296; DIA copies them, :474-477). */
int spmv_vec_copy(spmv_vec* dst, int64_t dst_offset, const spmv_vec* src, int64_t src_offset, int64_t n);

/* ---- exchange between the contexts (GPUs) of ONE process -------------------------------------------------
 * Replaces the x replication of the reference's sharded drivers — memcpy(p[i].X, x.values, ...) per NUMA node,
 * src/mat_vec.cpp:257,266 — by an all-gather on the devices: participant i holds slice [offsets[i], offsets[i+1]) of x
 * in its own (full-length) vector; afterwards every participant's vector holds all of [0, offsets[n]).  Asynchronous:
 * ordered by events behind the work queued on the participants' streams, and the streams continue behind it.
 * Transport: RCCL (ncclCommInitAll; one in-place ncclAllGather when the slices are equal, else a group of broadcasts, one
 * per slice; loaded with dlopen and called through rccl.h's own prototypes; fresh communicators pass a self-check first) when
 * there are two or more participants, each with a GPU of its own, else — or with SPMV_COMM=peer — concurrent
 * hipMemcpyPeerAsync pulls, one stream per peer link.  SPMV_COMM=rccl: RCCL or SPMV_ERR_*, also with ONE participant (a
 * one-GPU box exercising the RCCL calls).  Participants may share a GPU (several shards on one device): then the copies
 * are device-to-device.  The calling thread's current HIP device is left as it was found.
 * (One process per GPU instead: arm-spmv_amd/dist.py does the same exchange through torch.distributed.) */
typedef struct spmv_comm spmv_comm;
int         spmv_comm_create(spmv_ctx* const* ctxs, int32_t n, spmv_comm** out);
void        spmv_comm_destroy(spmv_comm* comm);
const char* spmv_comm_backend(const spmv_comm* comm); /* "rccl" or "peer-copy" */
int         spmv_comm_allgather(spmv_comm* comm, spmv_vec* const* vecs, const int64_t* offsets /* n + 1 */);

/* ---- matrices ------------------------------------------------------------------------------ */
/* CSR (include/matrix.h:27-47).  nnz = row_ptr[nrow]. */
int spmv_csr_upload(spmv_ctx* ctx, int32_t nrow, int32_t ncol, const int32_t* row_ptr,
                    const int32_t* col_ind, const double* values, spmv_mat** out);
int spmv_csr_wrap_device(spmv_ctx* ctx, int32_t nrow, int32_t ncol, const int32_t* d_row_ptr,
                         const int32_t* d_col_ind, const double* d_values, spmv_mat** out);
/* One row-range shard [row_begin,row_end) of a host CSR matrix whose offsets may exceed int32:
 * row_ptr64 is the GLOBAL 64-bit offset array; the shard keeps a rebased int32 row_ptr and global
 * column indices, exactly as src/mat_vec.cpp:250-265 builds NumaNode4CSR. */
int spmv_csr_upload_shard(spmv_ctx* ctx, int64_t row_begin, int64_t row_end, int32_t ncol,
                          const int64_t* row_ptr64, const int32_t* col_ind, const double* values,
                          spmv_mat** out);
/* COO (include/matrix.h:7-25): file order, unsorted and duplicate entries allowed (summed). */
int spmv_coo_upload(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int64_t nnz, const int32_t* row_ind,
                    const int32_t* col_ind, const double* values, spmv_mat** out);
int spmv_coo_wrap_device(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int64_t nnz,
                         const int32_t* d_row_ind, const int32_t* d_col_ind, const double* d_values,
                         spmv_mat** out);
/* ELL (include/matrix.h:70-92): column-major nrow*k arrays, padding col 0 / val 0.0
 * (src/matrix.cpp:473-474).  nnz = true nonzero count (used for the flop count only). */
int spmv_ell_upload(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t k, int64_t nnz,
                    const int32_t* col_ind, const double* values, spmv_mat** out);
int spmv_ell_wrap_device(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t k, int64_t nnz,
                         const int32_t* d_col_ind, const double* d_values, spmv_mat** out);
/* CSC (include/matrix.h:49-68) and DIA (include/matrix.h:117-138). */
int spmv_csc_upload(spmv_ctx* ctx, int32_t nrow, int32_t ncol, const int32_t* col_ptr,
                    const int32_t* row_ind, const double* values, spmv_mat** out);
int spmv_dia_upload(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t ndiags,
                    const int32_t* offsets, const double* values, spmv_mat** out);

int spmv_mat_destroy(spmv_mat* m);
int spmv_mat_get_info(const spmv_mat* m, spmv_mat_info* info);
/* Structural check for untrusted input (the products themselves, like the reference's loops, never check an index):
 * every row/column index inside its range, offset arrays non-decreasing from 0 to the entry count.
 * SPMV_ERR_INVALID + message if not.  One pass over the index arrays; synchronous. */
int spmv_mat_validate(const spmv_mat* m);
/* Force a CSR kernel (and, for VECTOR, lanes_per_row in {1,2,4,...,64}; 0 = keep auto choice).
 * COO, CSC and ELL handles take AUTO, VECTOR (the format's own kernel: segmented scan / atomic scatter / one lane per row;
 * for ELL lanes_per_row 1 or 2 picks the one- or two-rows-per-lane variant; for a large COO handle whose x is beyond an
 * XCD's L2 the scan runs over a copy of the entries in column bins - "coo_column_bins" below) or PANEL (regroup by row now and
 * run the panel kernel on that copy).  AUTO for COO: the scan over the entries as they are against a copy grouped by row
 * (duplicates and the order inside a row kept) that picks its own CSR kernel - timed from 64K entries on, the copy by the
 * model from 1.5M on.  AUTO for ELL: its own variants timed from 64K slots on; the row-grouped copy (every slot, padding
 * included: the sums and the reference's 0.0 * x[0] stay) is a candidate where one lane per row cannot work - at most 65536
 * rows of 16 slots and more, or rows whose blocks of 256 span more than 16 columns per row (scattered) and whose slots are not
 * diagonals.  AUTO for CSC: the scatter over the columns (one atomic on y per entry) against the copy grouped by row, timed
 * from 64K entries on, the copy by the model from 1.5M on. */
int spmv_mat_set_kernel(spmv_mat* m, int32_t kernel, int32_t lanes_per_row);
int spmv_mat_set_flags(spmv_mat* m, uint32_t flags);
/* Named parameters.  None is needed in normal use: what is left at its default is chosen when the layout is built, by
 * timing a handful of launches (about 0.1 s for 320M entries; SPMV_PANEL_TRIAL=0 or "panel_trial" 0: no launches, the
 * choice that wins on scattered columns).  Panel-kernel parameters take effect with the next spmv_mat_set_kernel:
 *   "panel_rows"     rows per group (0 = choose, entry-balanced; at most 20000)
 *   "panel_rounds"   0 (default) = the fewest groups the cap of 20000 rows allows, a whole number of rounds of 256 workgroups; where
 *                    skewed rows leave the busiest CU more than 1.15x the mean, a cut for 2-4 rounds is timed against it (get:
 *                    "panel_rounds" what was built, "panel_rounds_us_one" / "_more" the timing); 1 = never, k = groups for k rounds
 *   "panel_width"    columns per panel (0 = 131072; at most 524288, rounded down to whole 128-byte lines of x)
 *   "panel_sort"     1 = bucket the entries of a panel by 128-byte line of x (default), 0 = leave unordered
 *   "panel_aos"      entry layout: 4 = 12-byte packed entries, slices of 1024 stored in interleaved pairs and read with 8- and
 *                    16-byte loads (default; falls back to 0 where padding would outweigh it), 3 = the same without the
 *                    pairing (4-/8-byte loads), 0 = three arrays (14 bytes)
 *   "panel_unroll"   chunk = unroll x 1024 entries: 2, 4 or 8 (0 = by trial; 16 existed through round 3 - every instance spilled
 *                    registers - and now runs 8)
 *   "panel_pipe"     order of a chunk's memory instructions: 0 = no pipelining, 1 = next chunk's stream first,
 *                    2 = this chunk's gathers first (-1 = by trial)
 *   "panel_sync"     how the 16 wavefronts of a workgroup are kept in the same chunk: 0 = not at all, 1 = workgroup barrier at
 *                    the top of every chunk, 3 = barrier between a chunk's loads and its LDS adds (-1 = by trial; DESIGN.md 4.2:
 *                    this is what keeps the workgroups of an XCD in step on scattered columns).  (2, a split barrier through an LDS
 *                    counter, was measured no better than 3 and runs 3 since round 5.)
 *                    Round 1's clock throttle and what hung on it - "panel_pace_ns", "panel_guard", "panel_stagger", the trace build
 *                    "panel_trace", the A/B switch "panel_legacy" - were deleted in round 5 with the kernel that carried them
 *                    inside its chunk loop (the headline's schedule had come to depend on that dead code: DESIGN.md 4.2).
 *   "panel_keep_csr" 0 = release col_ind / values of a CSR handle whose product runs from the panel or two-phase layout or from its ELL copy
 *                    (memory 2x -> 1x the matrix; download, other kernels, re-builds and conversions are then refused)
 *   "panel_trial"    1 / 0 = timing launches when the layout is built, yes / no (-1 = environment, default yes)
 *   "dia_col_bound"  DIA handles: columns >= this are skipped (row shards keep the bound of the whole matrix)
 *   "twophase_panel_cols"   two-phase CSR kernel: columns of x per panel (<= 20000, even); takes effect at the next
 *                    spmv_mat_set_kernel(SPMV_CSR_TWOPHASE)
 *   "twophase_placement_budget_mb", "twophase_choose_pieces"   two-phase CSR kernel, TRANSIENT MEMORY.  The stream of products
 *                    between the two phases (8 bytes per padded entry: 2.6 GB for 320M entries) lives in pieces of 1 GB,
 *                    and where those lie in the device's PHYSICAL memory decides a tenth of the product's time (DESIGN.md
 *                    4.7; nothing can be asked of the allocator).  When the layout is built - by spmv_mat_set_kernel(TWOPHASE),
 *                    or by the upload / generator calls when AUTO picks this kernel - the engine therefore allocates
 *                    budget / 1 GB more pieces than the stream needs, times configurations of pieces (3 products each,
 *                    ~0.3 s in all), keeps the fastest and FREES EVERY OTHER PIECE BEFORE THE CALL RETURNS.  While the call
 *                    runs it holds: the layout + budget (default 8192 MB, never more than a quarter of the free device
 *                    memory) + two scratch vectors (8 ncol + 8 nrow bytes).  Afterwards: the layout, its product stream
 *                    rounded up to whole gigabytes.  "twophase_placement_budget_mb": -1 = default (or the environment's
 *                    SPMV_TP_PLACEMENT_BUDGET_MB), 0 = no search and no timing launches ("panel_trial" 0 /
 *                    SPMV_PANEL_TRIAL=0 do the same), otherwise megabytes.  PRECEDENCE: a value >= 0 set on the handle wins
 *                    over the environment variable, which is consulted only while the handle's value is -1 (and only when a
 *                    layout is built or the search is run again, never on a product's path); "panel_trial" 0 /
 *                    SPMV_PANEL_TRIAL=0 override both (no search); it applies to the next build, or at once with
 *                    "twophase_choose_pieces" (any value): run the search again on the built layout.  Streams below 512 MB
 *                    are never searched.  The outcome is reported by spmv_mat_get_param (below); a timing launch that fails
 *                    is an error of the call, not a silent fallback.
 *                    Round 5: (a) a pool of more than 24 extra pieces is SAMPLED (every k-th piece, ~16 per slot): the classes of
 *                    physical memory come in runs of gigabytes to tens of gigabytes, so a 64 GB pool reaches as far with a
 *                    quarter of the timing launches; (b) "panel_keep_csr" 0 on a two-phase handle first offers the whole
 *                    gigabytes inside the col_ind / values arrays it is about to release to the search (no allocation, ~0.1 s):
 *                    an array that ends up under the stream stays with the layout, the pieces it replaced are freed
 *                    ("twophase_pieces_carved"; "twophase_offer_csr_copy" 0 switches the offer off).
 *   "twophase_rotate"   1 (default): workgroup b of the expand phase starts b / 256 of the way through its panels
 *   "twophase_only", "twophase_realloc", "twophase_pool_alloc", "twophase_pool_config"   experiments, refused unless
 *                    SPMV_EXPERIMENTS=1 is in the environment: run phase A (1) or B (2) alone - THE PRODUCT IS THEN WRONG -,
 *                    move streams of the built layout to fresh allocations (bits 1 products, 2 values, 4 columns, 8 rows,
 *                    16 table), hold a pool of pieces and put any of them under the product stream (10 bits per slot):
 *                    tools/tune_twophase.py, tools/probe_twophase_{moves,pairs,classes,rotate}.py
 *   "ell_tiled_values"  ELL handles whose slots were found to be diagonals, takes effect at once: 1 = keep a copy of the values
 *                    in tiles of 512 rows, (tile * k + slot) * 512 + row, and multiply from it - a workgroup then reads one
 *                    contiguous stretch instead of k stretches nrow * 8 bytes apart; same bits.  8 bytes per slot of device
 *                    memory for 1-7 % of the product's time (C3), so never made unasked.  0 = drop it.
 *   "coo_column_bins"   COO handles, takes effect at once: the segmented scan (kernel VECTOR) runs over a COPY of the entries in
 *                    8 x value column bins, the bins of one XCD after the other, so that each XCD gathers x from a slice that
 *                    stays in its L2 (kernels_coo.hip; C4: 0.76 ms against 1.81).  -1 = as many bins as keep a slice within
 *                    2 MB (what spmv_mat_set_kernel(VECTOR) does by itself when x is beyond 3 MB and the handle has >= 2M
 *                    entries), 0 = drop the copy: the scan reads the handle's own arrays in their order, 1..8 = bins per
 *                    XCD.  Costs 16 bytes per entry of device memory (+ up to 64 x 2048 entries of padding) and, while it is
 *                    built, 8 bytes per entry more.  Needs fewer than 2^31 - 131072 entries.  A row is spread over up to that
 *                    many runs, each ending in an atomic on y: results differ from the scan in place by rounding only.
 *   "symgs_order"    sweep order of spmv_symgs / SPMV_PRECOND_SYMGS: 1 multicolour (default), 0 the matrix's own row order
 *   "panel_two_per_cu"   0 = never two workgroups per CU (default 1: two when their accumulators fit the LDS twice) */
int spmv_mat_set_param(spmv_mat* m, const char* name, int64_t value);
/* What is in effect: "panel_rows", "panel_width", "panel_sort", "panel_groups", "panel_layout", "panel_unroll",
 * "panel_pipe", "panel_sync", "select_candidates" / "select_us_<kernel>" (what AUTO timed, microseconds per product; 0 = not timed),
 * "select_rounds" (rounds the last trial took until its candidates' minima stood still), "adds_into_y_with_atomics" (1: the product
 * adds into y with device atomics - the COO scan, the CSC scatter, CSR under SEGSCAN or SPLIT's chunks, or a copy that runs one of
 * those; spmv_apply_host then stages y in device memory),
 * "panel_bytes", "panel_keep_csr", "device_bytes", "window_max_span", "window_avg_span", "twophase_panel_cols",
 * "twophase_padded" (entries of the two-phase layout with its padding), "twophase_pieces" (1 GB pieces of its product stream),
 * "twophase_pieces_carved" (how many of them lie inside allocations taken over from the released CSR copy),
 * "ell_tiled_values" (1: the ELL product reads its values from the copy in tiles), "coo_column_bins" (bins of the copy the COO
 * segmented scan runs over, 0: none), "coo_bins_padded" (its entries with the padding),
 * "twophase_placement_budget_mb", "twophase_placements_timed" (configurations of pieces timed by the search, 0 = no search ran),
 * "twophase_placement_spread" (time as built / time kept in 1/1000, both re-timed in turn when the search is over; a
 * configuration that does not hold up there is dropped for the pieces as built, so never < 1000), "twophase_pieces_exchanged", "ell_diagonal_slots" (1: the slots of an ELL
 * handle were found to be diagonals and conforming rows read no column index), "symgs_order", "symgs_colours",
 * "symgs_levels_forward", "symgs_levels_backward", "symgs_launches", "symgs_bytes", "symgs_fused" (1: the colouring is proper and
 * the sweep takes one launch per colour). */
int spmv_mat_get_param(const spmv_mat* m, const char* name, int64_t* value);
/* ---- plans: a handle's set-up decisions as plain data (plan.hip) ------------------------------------------------------------
 * AUTO is a measurement (spmv_mat_set_kernel above): which kernel a handle runs, in which layout, with which chunk size,
 * barrier placement, split threshold or ELL variant is found by timing candidates when the handle is created.  Consequences:
 * NON-DETERMINISM - two handles of one matrix, or two ranks holding statistically identical shards, may end on different
 * kernels (candidates within 2 % of each other swap places between two trials) and with them on different last bits of y (every
 * kernel stays within the parity tolerance; the order of a row's additions differs) and different step times - and set-up
 * time (C2: 0.2 s with the timing launches, 0.05 s without).  The reference builds its shards once, the same way every time
 * (src/mat_vec.cpp:240-268).  Three ways to get that back:
 *   - spmv_mat_set_kernel with an explicit kernel, or SPMV_PANEL_TRIAL=0 / "panel_trial" 0: the model alone, no launches;
 *   - a PLAN: spmv_mat_get_plan writes what a handle decided - for itself and for the copies it runs from - into a POD blob
 *     (16-byte header + 128 bytes per handle; pass buf = NULL to learn the size in *len); spmv_mat_set_plan builds exactly
 *     that kernel and layout on another handle of the same format, with NO timing launch;
 *   - spmv_ctx_set_plan: every handle of the plan's format created on the context afterwards (uploads, wraps, generators,
 *     conversions, spmv_csr_extract_rows) takes the plan instead of selecting; buf = NULL clears it.
 * A plan holds decisions, nothing about the matrix: row cuts, column panels and slot descriptors are derived again from the
 * handle's own arrays, so the plan of one shard fits a shard of another size (arm-spmv_amd/dist.py broadcasts rank 0's).  It
 * does not hold where a two-phase product stream lies in a device's physical memory: run "twophase_choose_pieces" afterwards
 * where that search is wanted.  A plan that does not fit the handle's matrix (an LDS window too wide, an ELL copy of a matrix
 * with an empty row) is an error of spmv_mat_set_plan; the handle then selects by itself.  Blobs are validated (magic, version,
 * sizes, ids, child indices) before anything is read from them. */
int spmv_mat_get_plan(const spmv_mat* m, void* buf, int64_t* len);
int spmv_mat_set_plan(spmv_mat* m, const void* buf, int64_t len);
int spmv_ctx_set_plan(spmv_ctx* ctx, const void* buf, int64_t len);
/* The checks spmv_mat_set_plan / spmv_ctx_set_plan run on a blob, by themselves (no device, no handle: a plan received from
 * another rank or read from a file can be looked at anywhere): SPMV_OK and the root's format, the root's kernel and the number
 * of nodes (any pointer may be NULL), or SPMV_ERR_INVALID with the reason in spmv_last_error(). */
int spmv_plan_check(const void* buf, int64_t len, int32_t* format, int32_t* kernel, int32_t* nodes);
/* Copy the arrays of a handle back to the host (any pointer may be NULL to skip it).
 *   CSR: a=row_ptr[nrow+1]  b=col_ind[nnz]      v=values[nnz]
 *   COO: a=row_ind[nnz]     b=col_ind[nnz]      v=values[nnz]
 *   ELL: a=NULL             b=col_ind[nrow*k]   v=values[nrow*k]
 *   CSC: a=col_ptr[ncol+1]  b=row_ind[nnz]      v=values[nnz]
 *   DIA: a=offsets[ndiags]  b=NULL              v=values[nrow*ndiags] */
int spmv_mat_download(const spmv_mat* m, int32_t* a, int32_t* b, double* v);
/* Device pointers of the same three arrays (borrowed; NULL where the format has none). */
int spmv_mat_device_ptrs(const spmv_mat* m, const int32_t** a, const int32_t** b, const double** v);

/* ---- the hot path: y += A*x  (include/mat_vec.h:7-11) ------------------------------------------ */
/* x must have ncol entries, y nrow entries (the shard's rows).  Asynchronous. */
int spmv_apply(spmv_ctx* ctx, const spmv_mat* A, const spmv_vec* x, spmv_vec* y);
/* `reps` back-to-back applications between two HIP events on the context's stream (the reference's
 * NUM_TEST loop, main.cpp:56-59).  Returns the mean milliseconds per application.  Synchronous. */
int spmv_apply_timed(spmv_ctx* ctx, const spmv_mat* A, const spmv_vec* x, spmv_vec* y, int32_t reps,
                     double* ms_per_apply);

/* y_host += A * x_host with the caller's HOST vectors, synchronous - the reference's own call shape (include/mat_vec.h:7-11:
 * every CSRMatrixMatVector(A, x, y) hands over host arrays; main.cpp:56-59 does it 50 times), as ONE entry point so that the
 * hand-over can be done the cheapest way for its size.  Vectors of up to 1 MB together: x by CPU stores straight into device
 * memory where the platform has a large BAR and the context's self-check of such stores passed (SPMV_HOST_STORES=0: through a
 * pinned staging buffer and one more launch); y stays on the HOST where the kernel that runs forms a row's sum before it adds it
 * to y (row-parallel, panel and two-phase CSR: the kernel writes the sums into the pinned buffer, the host adds them to y while
 * it copies out - the same IEEE addition, bit for bit), is updated in place in the pinned buffer over the host link where the
 * accumulator starts at y_i (ELL, DIA, scalar and LDS-window CSR), and goes through a device buffer for the kernels that add
 * into y with atomics (spmv_mat_get_param "adds_into_y_with_atomics"; two more launches).  The call ends in hipStreamSynchronize.
 * Larger vectors: asynchronous copies.  x_host has ncol entries, y_host nrow.  The caller's arrays are neither registered nor
 * mapped (they may be freed or moved between calls).  C1 (10000 x 10000 x 16): 22 us per call, of which the kernel takes 3 - the
 * rest is one launch and its completion (10.5 us for an empty kernel on this platform), 80 KB each way and the runtime; never
 * part of a throughput figure: resident vectors (spmv_apply) are what the roofline numbers are measured with. */
int spmv_apply_host(spmv_ctx* ctx, const spmv_mat* A, const double* x_host, double* y_host);

/* ---- BLAS-1 (include/vec_vec.h:6-7) ---------------------------------------------------------- */
int spmv_dot(spmv_ctx* ctx, const spmv_vec* x, const spmv_vec* y, double* result); /* synchronous */
int spmv_axpby(spmv_ctx* ctx, double alpha, const spmv_vec* x, double beta, const spmv_vec* y,
               spmv_vec* w);

/* ---- the solver step around the product ------------------------------------ */
/* The reference ships vec_dot / vec_axpby (src/vec_vec.cpp) and `diagonal // for SymGS` fields
 * (include/matrix.h:36,81) for a Krylov loop it never calls; these two entry points are that loop,
 * device-resident.  Not present in the reference's API.
 *
 * spmv_apply_dot: y = A*x (overwrite != 0) or y += A*x, and *dot = sum_i w_i * y_i over the UPDATED y,
 *   computed in the product's own write-back where the kernel supports it (panel kernel), else by a
 *   pass behind it.  w has nrow entries and may be x itself (square A).  Synchronous (returns the scalar).
 * spmv_cg: conjugate gradients for symmetric positive definite A (square): solves A*x = b starting from
 *   the x passed in, until ||r|| <= rel_tol * ||b|| or max_iter iterations.  alpha/beta stay on the
 *   device; the host reads the residual every check_every iterations (>= 1).  *iters = iterations run,
 *   *rel_resid = ||r|| / ||b|| of the recurrence at the last check.  SPMV_ERR_INVALID if p.Ap <= 0.
 *   precond: spmv_precond (Jacobi uses the diagonal the reference's containers carry "for SymGS", matrix.h:36). */
int spmv_apply_dot(spmv_ctx* ctx, const spmv_mat* A, const spmv_vec* x, spmv_vec* y, int32_t overwrite,
                   const spmv_vec* w, double* dot);
enum spmv_precond
{
    SPMV_PRECOND_NONE   = 0,
    SPMV_PRECOND_JACOBI = 1, /* z = D^-1 r, D = diag(A) read from a CSR handle (zero diagonal: SPMV_ERR_INVALID) */
    SPMV_PRECOND_SYMGS  = 2  /* z = one symmetric Gauss-Seidel sweep on A z = r from z = 0 (spmv_symgs below; the handle is
                                set up on first use although it is passed as const) */
};
int spmv_cg(spmv_ctx* ctx, const spmv_mat* A, const spmv_vec* b, spmv_vec* x, int32_t max_iter,
            double rel_tol, int32_t check_every, int32_t precond, int32_t* iters, double* rel_resid);
/* spmv_symgs: `sweeps` symmetric Gauss-Seidel sweeps on A*x = b, x updated in place: forward over the rows in sweep
 *   order with the newest x, then backward — the sweep the reference's `diagonal // for SymGS` fields were reserved
 *   for (include/matrix.h:36,81) and that it never wrote.  A: CSR handle holding the whole square matrix with a non-zero
 *   diagonal (duplicates of a diagonal entry are summed); symmetric or not.
 *   Sweep order (spmv_mat_set_param "symgs_order", before the first use or between uses):
 *     1 (default) multicolour: the greedy colouring in row order (colour(i) = smallest colour no coupled row j < i has),
 *       rows swept colour by colour, ascending row index inside a colour — rows of a colour are solved together, a
 *       sweep is a handful of launches (two colours, 6 launches for a 7-point Laplacian);
 *     0 the matrix's own row order i = 0..n-1, exactly — as parallel as the matrix allows (the rows are solved in
 *       dependency levels: ~480 levels, ~790 launches for a 7-point Laplacian on 160^3 points; a band matrix is sequential).
 *   spmv_symgs_order returns the sequence (order[k] = the k-th row of a forward sweep), so that a host implementation can
 *   repeat the sweep number by number.  Either way the result is exact for that order (only the sums inside a row are
 *   taken by several lanes).
 *   spmv_symgs_setup (also run by the first spmv_symgs / spmv_cg(SPMV_PRECOND_SYMGS) on the handle): colours the rows,
 *   splits A into L + D + U by sweep order on the device, finds the dependency levels of both parts and fixes a launch
 *   schedule; synchronous; kept in the handle (about the size of the matrix again).  A sweep is then t = b - U*x,
 *   (L+D)*x = t level by level, t = b - L*x, (D+U)*x = t level by level — asynchronous.  spmv_mat_get_param
 *   "symgs_colours" / "symgs_levels_forward" / "symgs_levels_backward" / "symgs_launches" (per sweep) / "symgs_bytes".
 *   SPMV_ERR_UNSUPPORTED if a dependency chain exceeds 2^18 rows. */
int spmv_symgs_setup(spmv_ctx* ctx, spmv_mat* A);
int spmv_symgs_order(spmv_ctx* ctx, const spmv_mat* A, int32_t* order /* host, nrow entries */);
int spmv_symgs(spmv_ctx* ctx, spmv_mat* A, const spmv_vec* b, spmv_vec* x, int32_t sweeps);

/* ---- format conversion on the device (src/matrix.cpp:115-154, :450-500) -------------------------- */
/* Both keep the COO order of the entries inside each row (stable), like the reference's backward
 * scatter, so the result is identical to the reference's arrays, not merely equivalent. */
int spmv_coo_to_csr(spmv_ctx* ctx, const spmv_mat* coo, spmv_mat** out_csr);
int spmv_coo_to_ell(spmv_ctx* ctx, const spmv_mat* coo, spmv_mat** out_ell);
int spmv_csr_to_ell(spmv_ctx* ctx, const spmv_mat* csr, spmv_mat** out_ell);
/* Split a CSR handle (a row shard) by column range, on the device: `inside` holds the entries with a column in
 * [col_begin, col_end), REBASED to 0 (it multiplies the caller's own slice of x: ncol = col_end - col_begin);
 * `outside` holds the others with their global columns.  A*x = inside*x[col_begin:col_end] + outside*x, rows and the
 * order inside each row kept.  For the sharded solver step: the inside product needs no exchange and can run while
 * the x all-gather is in flight (no counterpart in the reference).
 * Row bookkeeping: `outside` keeps the shard's row_begin.  `inside` reports row_begin = csr.row_begin - col_begin (see
 * spmv_mat_get_info): because its columns are rebased, local row i meets its own diagonal at column row_begin + i of
 * `inside` - the offset the Jacobi diagonal and the Gauss-Seidel sweep of the sharded solver use.  For the usual call
 * (col_begin = the shard's first row) that is 0: a square block starting at (0, 0).  It is NEGATIVE when col_begin lies
 * beyond the shard's first row (rows before col_begin have no diagonal inside the block). */
int spmv_csr_split_columns(spmv_ctx* ctx, const spmv_mat* csr, int32_t col_begin, int32_t col_end,
                           spmv_mat** out_inside, spmv_mat** out_outside);

/* ---- row-range sharding (src/mat_vec.cpp:233,245-246) ------------------------------------------- */
/* Equal rows per part, the last part takes the remainder.  Pure host arithmetic. */
int spmv_partition_rows(int64_t nrow, int32_t nparts, int32_t part, int64_t* row_begin,
                        int64_t* row_end);
/* Entry-balanced alternative ("an nnz-balanced split as an option for C4-like skew"): bounds[nparts+1],
 * bounds[p] = first row of part p; the boundary of part p is the row boundary whose offset lies nearest to p / nparts of the
 * entries (rows are never split, so a part can exceed its share by at most half its longest boundary row).  The reference
 * partitions by equal rows only (src/mat_vec.cpp:151,233): that stays the default of every driver; this is the option. */
int spmv_partition_rows_balanced(int64_t nrow, const int64_t* row_ptr64, int32_t nparts,
                                 int64_t* bounds);
/* The same for a DEVICE-RESIDENT handle: bounds[nparts+1] over its rows (over its COLUMNS for CSC, which the reference shards
 * by column, src/mat_vec.cpp:299-337).  balance_entries = 0: equal rows, the last part takes the remainder (the reference's
 * split); 1: by entries - CSR / CSC from the offset array, COO from a histogram of its row indices taken on the device (any
 * entry order); ELL and DIA store the same number of slots for every row, so equal rows are equal work and are what they
 * get either way.  Synchronous; the offsets cross to the host (4 bytes per row), the entries do not. */
int spmv_mat_partition_rows(const spmv_mat* m, int32_t nparts, int32_t balance_entries, int64_t* bounds);
/* Rows [row_begin, row_end) of a device-resident CSR handle (local row numbers of that handle) as a shard of their own on
 * dst_ctx - the same GPU or a peer over xGMI: rebased int32 row_ptr, global column indices (src/mat_vec.cpp:250-265), entries
 * copied device to device, analysed like an uploaded shard; its row_begin = the source's row_begin + row_begin.  With
 * spmv_mat_partition_rows this shards a matrix that was generated, converted or uploaded once without a second trip through
 * the host.  Synchronous.  The source must still hold its CSR arrays (panel_keep_csr = 1). */
int spmv_csr_extract_rows(spmv_ctx* dst_ctx, const spmv_mat* csr, int64_t row_begin, int64_t row_end, spmv_mat** out);

/* ---- synthetic inputs (counter-based splitmix64) ------------- */
/* CSR shard with exactly k entries in each of rows [row_begin,row_end) of a (nrow_global x ncol)
 * matrix.  band == 0: columns uniform over [0,ncol); band > 0: uniform in a window of `band`
 * columns centred on the diagonal (wrapping).  values U(-1,1). */
int spmv_gen_csr_uniform(spmv_ctx* ctx, int64_t row_begin, int64_t row_end, int32_t ncol, int32_t k,
                         int32_t band, uint64_t seed, spmv_mat** out);
/* ELL with k slots per row, circulant band col = (i + d - k/2) mod ncol, values U(-1,1). */
int spmv_gen_ell_banded(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t k, uint64_t seed,
                        spmv_mat** out);
/* Row-sorted COO with power-law row lengths min(max_len, floor(8/u)), u ~ U(0,1], uniform columns. */
/* DIA (row-major, reference layout) with k diagonals at offsets d - k/2, same value draws as spmv_gen_ell_banded. */
int spmv_gen_dia_banded(spmv_ctx* ctx, int32_t nrow, int32_t k, uint64_t seed, spmv_mat** out);
int spmv_gen_coo_powerlaw(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t max_len, uint64_t seed,
                          spmv_mat** out);
/* The same distribution of row lengths taken at its quantiles (u_i = (i + 1) / nrow) instead of drawn: the rows come SORTED BY
 * LENGTH, the longest first - every heavy row at one end, the positional skew an equal-rows partition handles worst (the first
 * of 8 equal-rows shards of N = 2M holds 43 % of the entries).  Columns and values as above. */
int spmv_gen_coo_powerlaw_sorted(spmv_ctx* ctx, int32_t nrow, int32_t ncol, int32_t max_len, uint64_t seed,
                                 spmv_mat** out);
/* v[i] = U(0,1) drawn from (seed, global index index_offset + i). */
int spmv_gen_vec_uniform(spmv_ctx* ctx, spmv_vec* v, int64_t index_offset, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* SPMV_ABI_H */
