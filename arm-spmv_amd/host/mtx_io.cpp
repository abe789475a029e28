// mtx_io.cpp — the harness I/O of the reference, re-implemented: Matrix Market coordinate reader, the plain-text
// vector files and the wall-clock timer (reference src/data_io.cpp, src/mytime.cpp; the banner rules are those of
// the NIST Matrix Market format the reference parses with its vendored mmio.c).
//
// Behaviour kept identical to the reference's COOMatrixRead (src/data_io.cpp:45-105) because tools scrape it:
//   * the progress lines and `### ROW=%d, COL=%d, NNZ=%d`
//   * only `matrix coordinate complex` is refused (:66-71); `symmetric` files are NOT expanded and `pattern`
//     files are read with the same three-field format (a pattern file therefore mis-parses, as it does there)
//   * entries are read as "%d %d %lg", 1-based -> 0-based (:83-88); failures print and exit(1)
// Opt-in (off by default so that the default output equals the reference's):
//   SPMV_MTX_PATTERN=1    a `pattern` file is read as "%d %d" per entry with value 1.0
//   SPMV_MTX_SYMMETRIC=1  `symmetric` / `hermitian` / `skew-symmetric` files are expanded: every off-diagonal entry
//                         (i, j, v) is followed by its mirror (j, i, v) (skew: -v); NNZ in the `###` line is the
//                         expanded count
//   SPMV_MTX_CACHE=1      binary cache beside the file (`<file>.spmvbin`: header + the three arrays as parsed, i.e.
//                         after the two options above); used when its recorded size / mtime of the .mtx and the
//                         options match, rewritten otherwise.  Same output lines either way.
#include <algorithm>
#include <cctype>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>
#include <sys/time.h>

#include "arm_spmv_compat.hpp"
#include "mm_banner.h"

// ---------------------------------------------------------------------------------------------------------
// Matrix Market banner / size line
// ---------------------------------------------------------------------------------------------------------
static std::string lower(std::string s)
{
    for (char& c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}

int mm_banner_read(FILE* fp, mm_banner* out)
{
    char line[1100];
    if (!fgets(line, sizeof(line), fp)) return MM_BANNER_PREMATURE_EOF;
    char tag[64], object[64], layout[64], field[64], symmetry[64];
    if (sscanf(line, "%63s %63s %63s %63s %63s", tag, object, layout, field, symmetry) != 5) return MM_BANNER_PREMATURE_EOF;
    if (strcmp(tag, "%%MatrixMarket") != 0) return MM_BANNER_NO_HEADER;
    const std::string o = lower(object), l = lower(layout), f = lower(field), s = lower(symmetry);
    if (o != "matrix") return MM_BANNER_UNSUPPORTED;
    out->is_matrix = 1;
    if (l == "coordinate")
        out->is_sparse = 1;
    else if (l == "array")
        out->is_sparse = 0;
    else
        return MM_BANNER_UNSUPPORTED;
    if (f == "real")
        out->field = 'R';
    else if (f == "complex")
        out->field = 'C';
    else if (f == "pattern")
        out->field = 'P';
    else if (f == "integer")
        out->field = 'I';
    else
        return MM_BANNER_UNSUPPORTED;
    if (s == "general")
        out->symmetry = 'G';
    else if (s == "symmetric")
        out->symmetry = 'S';
    else if (s == "hermitian")
        out->symmetry = 'H';
    else if (s == "skew-symmetric")
        out->symmetry = 'K';
    else
        return MM_BANNER_UNSUPPORTED;
    snprintf(out->text, sizeof(out->text), "%s %s %s %s", o.c_str(), l.c_str(), f.c_str(), s.c_str());
    return 0;
}

int mm_size_read(FILE* fp, int* rows, int* cols, int* entries)
{
    char line[1100];
    *rows = *cols = *entries = 0;
    do
    {
        if (!fgets(line, sizeof(line), fp)) return MM_BANNER_PREMATURE_EOF;
    } while (line[0] == '%');  // comment lines
    for (;;)
    {
        if (sscanf(line, "%d %d %d", rows, cols, entries) == 3) return 0;
        if (!fgets(line, sizeof(line), fp)) return MM_BANNER_PREMATURE_EOF;  // blank lines before the size line
    }
}

// ---------------------------------------------------------------------------------------------------------
// Parallel entry parser (the fscanf loop is the wall-clock bottleneck for real files).
// Same semantics as `fscanf("%d %d %lg\n")` repeated nz times: the rest of the file is a stream of
// whitespace-separated tokens, entry k = tokens 3k, 3k+1, 3k+2 (line boundaries do not matter — which is also why a
// `pattern` file mis-parses exactly as it does in the reference).  The remainder of the file is read in one go, cut
// into slices at whitespace, tokens are counted per slice, and every thread converts its own tokens in place.
// Returns false (nothing consumed) when it declines: small files, one thread, or a read problem.
// ---------------------------------------------------------------------------------------------------------
static inline bool is_space(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }

static bool env_on(const char* name)
{
    const char* e = getenv(name);
    return e && *e && strcmp(e, "0") != 0;
}

// fields: tokens per entry (3, or 2 for a pattern file read with SPMV_MTX_PATTERN=1: the value is 1.0)
static bool parse_entries_parallel(FILE* fp, int nz, int* ii, int* jj, double* vv, int fields)
{
    const char* env      = getenv("SPMV_MTX_THREADS");
    unsigned    nthreads = env ? (unsigned)atoi(env) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (nthreads <= 1 || nz < (1 << 16)) return false;
    const long here = ftell(fp);
    if (here < 0 || fseek(fp, 0, SEEK_END) != 0) return false;
    const long end = ftell(fp);
    if (end < here || fseek(fp, here, SEEK_SET) != 0) return false;
    std::vector<char> buf((size_t)(end - here) + 1);
    if (fread(buf.data(), 1, (size_t)(end - here), fp) != (size_t)(end - here))
    {
        fseek(fp, here, SEEK_SET);
        return false;
    }
    buf.back() = '\0';
    const size_t len = buf.size() - 1;

    // slice boundaries at the start of a token
    std::vector<size_t> cut(nthreads + 1, len);
    cut[0] = 0;
    for (unsigned t = 1; t < nthreads; ++t)
    {
        size_t p = len / nthreads * t;
        while (p < len && !is_space(buf[p])) ++p;  // finish the token we landed in
        cut[t] = p;
    }
    // pass 1: tokens per slice
    std::vector<long long> first(nthreads + 1, 0);
    auto count = [&](unsigned t) {
        long long n = 0;
        bool      in = false;
        for (size_t p = cut[t]; p < cut[t + 1]; ++p)
        {
            const bool sp = is_space(buf[p]);
            if (!sp && !in) ++n;
            in = !sp;
        }
        first[t + 1] = n;
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthreads; ++t) pool.emplace_back(count, t);
    for (auto& th : pool) th.join();
    for (unsigned t = 0; t < nthreads; ++t) first[t + 1] += first[t];
    if (first[nthreads] < (long long)fields * nz)
    {
        printf("*** Matrix Market file ends after %lld of %d entries ***\n", first[nthreads] / fields, nz);
        exit(1);
    }
    // pass 2: convert
    std::vector<int> bad(nthreads, 0);
    auto convert = [&](unsigned t) {
        long long tok = first[t];
        size_t    p   = cut[t];
        while (p < cut[t + 1] && tok < (long long)fields * nz)
        {
            while (p < cut[t + 1] && is_space(buf[p])) ++p;
            if (p >= cut[t + 1]) break;
            char*           stop = nullptr;
            const long long k    = tok / fields;
            switch (tok % fields)
            {
                case 0: ii[k] = (int)strtol(&buf[p], &stop, 10) - 1; break;
                case 1:
                    jj[k] = (int)strtol(&buf[p], &stop, 10) - 1;
                    if (fields == 2) vv[k] = 1.0;
                    break;
                default: vv[k] = strtod(&buf[p], &stop); break;
            }
            if (stop == &buf[p]) bad[t] = 1;  // not a number: fscanf would have stopped here
            while (p < cut[t + 1] && !is_space(buf[p])) ++p;
            ++tok;
        }
    };
    pool.clear();
    for (unsigned t = 0; t < nthreads; ++t) pool.emplace_back(convert, t);
    for (auto& th : pool) th.join();
    for (unsigned t = 0; t < nthreads; ++t)
        if (bad[t])
        {
            printf("*** Matrix Market file holds a non-numeric token among its entries ***\n");
            exit(1);
        }
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// Binary cache (SPMV_MTX_CACHE=1)
// ---------------------------------------------------------------------------------------------------------
struct CacheHeader
{
    char      magic[8];  // "SPMVBIN1"
    long long src_size, src_mtime;
    int       nrow, ncol, nz, options;  // options: bit 0 pattern read as pairs, bit 1 symmetric expanded
};

static bool cache_load(const std::string& path, const struct stat& src, int options, int* nrow, int* ncol, int* nz, int** ii,
                       int** jj, double** vv)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    CacheHeader h;
    bool        ok = fread(&h, sizeof(h), 1, f) == 1 && memcmp(h.magic, "SPMVBIN1", 8) == 0 && h.src_size == (long long)src.st_size &&
              h.src_mtime == (long long)src.st_mtime && h.options == options && h.nz >= 0 && h.nrow >= 0 && h.ncol >= 0;
    if (ok)
    {
        const size_t n = (size_t)(h.nz > 0 ? h.nz : 1);
        *ii            = new int[n];
        *jj            = new int[n];
        *vv            = new double[n];
        ok = fread(*ii, sizeof(int), (size_t)h.nz, f) == (size_t)h.nz && fread(*jj, sizeof(int), (size_t)h.nz, f) == (size_t)h.nz &&
             fread(*vv, sizeof(double), (size_t)h.nz, f) == (size_t)h.nz;
        if (!ok)
        {
            delete[] *ii;
            delete[] *jj;
            delete[] *vv;
        }
        *nrow = h.nrow;
        *ncol = h.ncol;
        *nz   = h.nz;
    }
    fclose(f);
    return ok;
}

static void cache_store(const std::string& path, const struct stat& src, int options, int nrow, int ncol, int nz, const int* ii,
                        const int* jj, const double* vv)
{
    const std::string tmp = path + ".tmp";
    FILE*             f   = fopen(tmp.c_str(), "wb");
    if (!f) return;  // read-only directory: no cache, no complaint
    CacheHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "SPMVBIN1", 8);
    h.src_size  = (long long)src.st_size;
    h.src_mtime = (long long)src.st_mtime;
    h.nrow      = nrow;
    h.ncol      = ncol;
    h.nz        = nz;
    h.options   = options;
    const bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(ii, sizeof(int), (size_t)nz, f) == (size_t)nz &&
                    fwrite(jj, sizeof(int), (size_t)nz, f) == (size_t)nz && fwrite(vv, sizeof(double), (size_t)nz, f) == (size_t)nz;
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
}

// ---------------------------------------------------------------------------------------------------------
// COO / CSR / CSC / ELL readers (include/data_io.h:12-15)
// ---------------------------------------------------------------------------------------------------------
void COOMatrixRead(const char* filename, COOMatrix& A)
{
    printf("\tOpening matrix market file\n");
    FILE* fp = fopen(filename, "r");
    if (!fp)
    {
        printf("***Failed to open MatrixMarket file %s ***\n", filename);
        exit(1);
    }
    printf("\tReading MatrixMarket banner\n");
    mm_banner banner;
    if (mm_banner_read(fp, &banner) != 0)
    {
        printf("*** Could not process Matrix Market banner ***\n");
        exit(1);
    }
    if (banner.field == 'C' && banner.is_matrix && banner.is_sparse)
    {
        printf("Sorry, this application does not support ");
        printf("Market Market type: [%s]\n", banner.text);
        exit(1);
    }
    printf("\tReading sparse matrix size...");
    int nrow, ncol, nz;
    if (mm_size_read(fp, &nrow, &ncol, &nz) != 0) exit(1);

    const bool pattern = banner.field == 'P' && env_on("SPMV_MTX_PATTERN");
    const bool expand  = banner.symmetry != 'G' && env_on("SPMV_MTX_SYMMETRIC");
    const int  options = (pattern ? 1 : 0) | (expand ? 2 : 0);
    struct stat src;
    const bool        use_cache  = env_on("SPMV_MTX_CACHE") && stat(filename, &src) == 0;
    const std::string cache_path = std::string(filename) + ".spmvbin";
    if (use_cache)
    {
        int *   ci = nullptr, *cj = nullptr, cn = 0, cm = 0, cz = 0;
        double* cv = nullptr;
        if (cache_load(cache_path, src, options, &cn, &cm, &cz, &ci, &cj, &cv))
        {
            fclose(fp);
            printf("\tAllocating memory for matrix\n");
            printf("\tReading matrix entries from file\n");
            printf("### ROW=%d, COL=%d, NNZ=%d\n", cn, cm, cz);
            A.Free();
            A.nrow    = cn;
            A.ncol    = cm;
            A.nnz     = cz;
            A.row_ind = ci;
            A.col_ind = cj;
            A.values  = cv;
            return;
        }
    }
    if (nz < 0 || (expand && nz > INT32_MAX / 2))
    {
        printf("*** Matrix Market size line: %d entries cannot be held ***\n", nz);
        exit(1);
    }
    printf("\tAllocating memory for matrix\n");
    const size_t room = (size_t)(nz > 0 ? nz : 1) * (expand ? 2 : 1);  // mirrors of a symmetric file
    int*    ii = new int[room];
    int*    jj = new int[room];
    double* vv = new double[room];

    printf("\tReading matrix entries from file\n");
    if (!parse_entries_parallel(fp, nz, ii, jj, vv, pattern ? 2 : 3))
    {
        // small file, or threads disabled (SPMV_MTX_THREADS=1): the reference's loop (src/data_io.cpp:83-88)
        for (int k = 0; k < nz; ++k)
        {
            vv[k]        = 1.0;
            const int got = pattern ? fscanf(fp, "%d %d\n", &ii[k], &jj[k]) + 1 : fscanf(fp, "%d %d %lg\n", &ii[k], &jj[k], &vv[k]);
            if (got != 3)
            {
                printf("*** Matrix Market file ends after %d of %d entries ***\n", k, nz);
                exit(1);
            }
            --ii[k];
            --jj[k];
        }
    }
    fclose(fp);
    if (expand)
    {
        // in place, back to front: entry k lands at k + (off-diagonal entries before k), its mirror right after it
        int off = 0;
        for (int k = 0; k < nz; ++k) off += ii[k] != jj[k];
        const double sign = banner.symmetry == 'K' ? -1.0 : 1.0;
        int          w    = nz + off;
        for (int k = nz - 1; k >= 0; --k)
        {
            const int    i = ii[k], j = jj[k];
            const double v = vv[k];
            if (i != j)
            {
                --w;
                ii[w] = j;
                jj[w] = i;
                vv[w] = sign * v;
            }
            --w;
            ii[w] = i;
            jj[w] = j;
            vv[w] = v;
        }
        nz += off;
    }
    printf("### ROW=%d, COL=%d, NNZ=%d\n", nrow, ncol, nz);
    if (use_cache) cache_store(cache_path, src, options, nrow, ncol, nz, ii, jj, vv);

    A.Free();
    A.nrow    = nrow;
    A.ncol    = ncol;
    A.nnz     = nz;
    A.row_ind = ii;
    A.col_ind = jj;
    A.values  = vv;
    spmv_compat_prefetch(A);  // the device copy now: a container's set-up belongs to the reader, not to the first timed product
}

void CSRMatrixRead(const char* filename, CSRMatrix& A)
{
    COOMatrix B;
    COOMatrixRead(filename, B);
    A = B;
}

void CSCMatrixRead(const char* filename, CSCMatrix& A)
{
    COOMatrix B;
    COOMatrixRead(filename, B);
    A = B;
}

void ELLMatrixRead(const char* filename, ELLMatrix& A)
{
    COOMatrix B;
    COOMatrixRead(filename, B);
    A = B;
}

// ---------------------------------------------------------------------------------------------------------
// Vector files: first line n, then one value per line (src/data_io.cpp:10-40).  The writer keeps the reference's
// "%20.16g" (16 significant digits: not round-trip exact — do not use these files for golden vectors).
// ---------------------------------------------------------------------------------------------------------
void VectorRead(const char* filename, Vector& x)
{
    FILE* fp = fopen(filename, "r");
    if (!fp)
    {
        printf("***Failed to open vector file %s ***\n", filename);
        exit(1);
    }
    int n = 0;
    if (fscanf(fp, "%d", &n) != 1 || n < 0) n = 0;
    double* v = new double[n > 0 ? n : 1];
    for (int i = 0; i < n; ++i)
        if (fscanf(fp, "%lg", &v[i]) != 1) v[i] = 0.0;
    fclose(fp);
    x.Free();
    x.size   = n;
    x.values = v;
}

void VectorWrite(const char* filename, const Vector& x)
{
    FILE* fp = fopen(filename, "w");
    if (!fp)
    {
        printf("***Failed to open vector file %s ***\n", filename);
        exit(1);
    }
    fprintf(fp, "%d", x.size);
    for (int i = 0; i < x.size; ++i) fprintf(fp, "\n%20.16g", x.values[i]);
    fclose(fp);
}

// ---------------------------------------------------------------------------------------------------------
// Timer: seconds since the first call, which itself returns 0.0 (src/mytime.cpp:6-18)
// ---------------------------------------------------------------------------------------------------------
double mytimer(void)
{
    static bool   started = false;
    static timeval origin;
    timeval        now;
    gettimeofday(&now, 0);
    if (!started)
    {
        started = true;
        origin  = now;
        return 0.0;
    }
    return (double)(now.tv_sec - origin.tv_sec) + (double)(now.tv_usec - origin.tv_usec) / 1e6;
}
