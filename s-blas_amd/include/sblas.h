// sblas.h -- umbrella header of the S-BLAS API on MI355X (reference sblas.h:18-19).
#ifndef SBLAS_AMD_SBLAS_H
#define SBLAS_AMD_SBLAS_H

#include "spmm.h"
#include "spmv.h"

#endif
