// kernel.h -- generic element-wise device kernels of the API layer.
// The fp64 hot path does not use these (it calls sblas_hip_axpby_f64 in libsblas_hip.so); they serve the
// other DataType instantiations of DenseMatrix/DenseVector::plusDense*GPU (reference kernel.h:18-38).
#ifndef SBLAS_AMD_KERNEL_H
#define SBLAS_AMD_KERNEL_H

#include <hip/hip_runtime.h>

// vec = vec * beta + val
template <typename IdxType, typename DataType>
__global__ void denseVector_plusEqual_scalar(DataType *vec, DataType val, DataType beta, IdxType n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n; i += stride)
        vec[i] = vec[i] * beta + val;
}

// vec0 = vec0 * beta + vec1 * alpha
template <typename IdxType, typename DataType>
__global__ void denseVector_plusEqual_denseVector(DataType *vec0, const DataType *vec1, DataType alpha,
                                                  DataType beta, IdxType n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n; i += stride)
        vec0[i] = vec0[i] * beta + vec1[i] * alpha;
}

#endif
