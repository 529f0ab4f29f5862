// config.h -- compile-time settings of the S-BLAS API layer (MI355X build).
// Same knob names as the reference's config.h:19-27 so that code written against it keeps compiling; each one
// may be overridden on the compiler command line (-DERROR_BAR=1e-6 ...).
#pragma once

#ifndef NUM_THREADS_PER_BLK
#define NUM_THREADS_PER_BLK 256 // threads per workgroup of the generic element-wise kernels (kernel.h)
#endif

#ifndef WARP_SIZE
#define WARP_SIZE 64 // gfx950 wavefront width (the reference's 32 is an NVIDIA warp)
#endif

#ifndef RAND_INIT_SEED
#define RAND_INIT_SEED 211 // srand() seed of the dense initialisers (bit-identical B / x to the reference)
#endif

#ifndef ERROR_BAR
#define ERROR_BAR (1e-3) // absolute tolerance of check_equal, the reference's own acceptance bar
#endif

#ifndef SBLAS_NO_ERROR_CHECK
#define CUDA_ERROR_CHECK // keep API / kernel error checking on (utility.h)
#endif
