// mmio.h -- the two MatrixMarket header readers the S-BLAS API exposes (banner + coordinate size line),
// written for this build; same observable behaviour as the NIST routines the reference bundles
// (reference mmio.h:254-337 and :339-367).  The bulk loader lives in libsblas_hip.so (sblas_mm_read_*).
#ifndef SBLAS_AMD_MMIO_H
#define SBLAS_AMD_MMIO_H

#include <ctype.h>
#include <stdio.h>
#include <string.h>

typedef char MM_typecode[4]; // [0] object 'M', [1] format 'C'/'A', [2] field 'R'/'C'/'P'/'I', [3] symmetry 'G'/'S'/'H'/'K'

#define MM_MAX_LINE_LENGTH 1025
#define MM_PREMATURE_EOF 12
#define MM_NO_HEADER 14
#define MM_UNSUPPORTED_TYPE 15

#define mm_is_matrix(t) ((t)[0] == 'M')
#define mm_is_sparse(t) ((t)[1] == 'C')
#define mm_is_coordinate(t) ((t)[1] == 'C')
#define mm_is_dense(t) ((t)[1] == 'A')
#define mm_is_array(t) ((t)[1] == 'A')
#define mm_is_complex(t) ((t)[2] == 'C')
#define mm_is_real(t) ((t)[2] == 'R')
#define mm_is_pattern(t) ((t)[2] == 'P')
#define mm_is_integer(t) ((t)[2] == 'I')
#define mm_is_symmetric(t) ((t)[3] == 'S')
#define mm_is_general(t) ((t)[3] == 'G')
#define mm_is_skew(t) ((t)[3] == 'K')
#define mm_is_hermitian(t) ((t)[3] == 'H')

inline int mm_read_banner(FILE *f, MM_typecode *matcode)
{
    char line[MM_MAX_LINE_LENGTH], w[5][64];
    (*matcode)[0] = (*matcode)[1] = (*matcode)[2] = ' ';
    (*matcode)[3] = 'G';
    if (!fgets(line, sizeof line, f)) return MM_PREMATURE_EOF;
    if (sscanf(line, "%63s %63s %63s %63s %63s", w[0], w[1], w[2], w[3], w[4]) != 5) return MM_PREMATURE_EOF;
    for (int i = 1; i < 5; ++i)
        for (char *p = w[i]; *p; ++p) *p = (char)tolower((unsigned char)*p);
    if (strncmp(w[0], "%%MatrixMarket", 14) != 0) return MM_NO_HEADER;
    if (strcmp(w[1], "matrix") != 0) return MM_UNSUPPORTED_TYPE;
    (*matcode)[0] = 'M';
    if (!strcmp(w[2], "coordinate")) (*matcode)[1] = 'C';
    else if (!strcmp(w[2], "array")) (*matcode)[1] = 'A';
    else return MM_UNSUPPORTED_TYPE;
    if (!strcmp(w[3], "real")) (*matcode)[2] = 'R';
    else if (!strcmp(w[3], "complex")) (*matcode)[2] = 'C';
    else if (!strcmp(w[3], "pattern")) (*matcode)[2] = 'P';
    else if (!strcmp(w[3], "integer")) (*matcode)[2] = 'I';
    else return MM_UNSUPPORTED_TYPE;
    if (!strcmp(w[4], "general")) (*matcode)[3] = 'G';
    else if (!strcmp(w[4], "symmetric")) (*matcode)[3] = 'S';
    else if (!strcmp(w[4], "hermitian")) (*matcode)[3] = 'H';
    else if (!strcmp(w[4], "skew-symmetric")) (*matcode)[3] = 'K';
    else return MM_UNSUPPORTED_TYPE;
    return 0;
}

inline int mm_read_mtx_crd_size(FILE *f, int *M, int *N, int *nz)
{
    char line[MM_MAX_LINE_LENGTH];
    *M = *N = *nz = 0;
    do {
        if (!fgets(line, sizeof line, f)) return MM_PREMATURE_EOF;
    } while (line[0] == '%');
    if (sscanf(line, "%d %d %d", M, N, nz) == 3) return 0;
    for (;;) {
        const int got = fscanf(f, "%d %d %d", M, N, nz);
        if (got == EOF) return MM_PREMATURE_EOF;
        if (got == 3) return 0;
    }
}

#endif
