// TEST INFRASTRUCTURE ONLY.  Thin extern "C" shim around the reference's OWN loader, compiled
// from the sources where they lie (-I/root/reference); nothing is copied into this repo.
// mmio.h / mmio_highlevel.h are plain ANSI C and build with g++ as they stand.
// Output goes to oracle/_ref/ (git-ignored, travels to the GPU box as a built .so).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "mmio.h"
#include "mmio_highlevel.h"

extern "C" int ref_mm_info(const char *path, int *m, int *n, int *nnz, int *sym)
{
    return mmio_info(m, n, nnz, sym, path);
}
extern "C" int ref_mm_data(const char *path, int *rowptr, int *colidx, double *val)
{
    return mmio_data(rowptr, colidx, val, path);
}
