// Ordered split-K slab reduction shared by the GEMM translation units (no RDC: one static copy per TU).
#pragma once
#include "fx_common.h"

// C[m,n] (+)= sum_z slab[z][m][n] + bias[n]   -- fixed summation order => deterministic results
static __global__ void fx_reduce_slabs_kernel(float* __restrict__ C, const float* __restrict__ slabs,
                                              const float* __restrict__ bias, int M, int N, long ldc, int splitk,
                                              long slab_stride, int accumulate) {
  const long total = (long)M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / N), n = (int)(i % N);
    float s = 0.f;
    for (int z = 0; z < splitk; ++z) s += slabs[(long)z * slab_stride + i];
    if (bias) s += bias[n];
    float* dst = C + (long)m * ldc + n;
    if (accumulate) s += *dst;
    *dst = s;
  }
}
