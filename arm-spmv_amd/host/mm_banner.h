/* mm_banner.h — the two pieces of the NIST Matrix Market text format the harness needs: the banner line
 * ("%%MatrixMarket matrix coordinate real general") and the size line.  Own implementation (mtx_io.cpp);
 * the reference vendors NIST's mmio.c/.h for the same job (src/mmio.cpp:109-229). */
#ifndef MM_BANNER_H
#define MM_BANNER_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

enum
{
    MM_BANNER_PREMATURE_EOF = 12,
    MM_BANNER_NO_HEADER     = 14,
    MM_BANNER_UNSUPPORTED   = 15
};

typedef struct mm_banner
{
    int  is_matrix; /* object == matrix */
    int  is_sparse; /* coordinate (1) or array (0) */
    char field;     /* 'R'eal 'C'omplex 'P'attern 'I'nteger */
    char symmetry;  /* 'G'eneral 'S'ymmetric 'H'ermitian s'K'ew */
    char text[80];  /* lower-cased "matrix coordinate real general" for messages */
} mm_banner;

int mm_banner_read(FILE* fp, mm_banner* out);                      /* 0 on success */
int mm_size_read(FILE* fp, int* rows, int* cols, int* entries);    /* skips comment lines; 0 on success */

#ifdef __cplusplus
}
#endif
#endif
