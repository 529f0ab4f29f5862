// sblas.h -- umbrella header of the S-BLAS API on MI355X: pulls in the SpMV and SpMM operators (and through them the
// containers of matrix.h).  Mirrors the role of the reference's sblas.h:18-19.
#pragma once
#include "spmv.h"
#include "spmm.h"
