// config.h -- compile-time settings of the S-BLAS API layer (MI355X build).
// Mirrors the knobs of the reference's config.h:19-27; WARP_SIZE is the CDNA wavefront width.
#ifndef SBLAS_AMD_CONFIG_H
#define SBLAS_AMD_CONFIG_H

#define CUDA_ERROR_CHECK            // keep API/kernel error checking on (utility.h)
#define ERROR_BAR (1e-3)            // absolute tolerance of check_equal (reference config.h:21)
#define RAND_INIT_SEED 211          // srand() seed of the dense initialisers (reference config.h:23)
#define WARP_SIZE 64                // gfx950 wavefront (the reference's 32 is an NVIDIA warp)
#define NUM_THREADS_PER_BLK 256

#endif
