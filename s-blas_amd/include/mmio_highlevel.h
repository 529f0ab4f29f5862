// mmio_highlevel.h -- mmio_info / mmio_data with the reference's signatures (mmio_highlevel.h:7, :130),
// forwarding to the single-pass loader of libsblas_hip.so.  Return 0 on success, negative otherwise.
#ifndef SBLAS_AMD_MMIO_HIGHLEVEL_H
#define SBLAS_AMD_MMIO_HIGHLEVEL_H

#include "mmio.h"
#include "sblas_hip.h"

inline int mmio_info(int *m, int *n, int *nnz, int *isSymmetric, const char *filename)
{
    int32_t r = 0, c = 0, z = 0, s = 0;
    const int rc = sblas_mm_read_info(filename, &r, &c, &z, &s);
    if (rc != SBLAS_OK) {
        printf("Error loading matrix file.\n");
        return -1;
    }
    *m = r;
    *n = c;
    *nnz = z;
    *isSymmetric = s;
    return 0;
}

inline int mmio_data(int *csrRowPtr, int *csrColIdx, double *csrVal, const char *filename)
{
    const int rc = sblas_mm_read_csr(filename, csrRowPtr, csrColIdx, csrVal);
    if (rc != SBLAS_OK) {
        printf("Error loading matrix file.\n");
        return -1;
    }
    return 0;
}

#endif
