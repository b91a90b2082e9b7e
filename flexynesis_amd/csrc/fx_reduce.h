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
    // 8 slab loads in flight at a time (same summation order): one dependent load per iteration made this kernel
    // pure memory latency, ~1 us per slab
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= splitk; z += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = slabs[(long)(z + j) * slab_stride + i];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = slabs[(long)min(z + j, splitk - 1) * slab_stride + i];   // clamped: loads stay unconditional
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (z + j < splitk) ? v[j] : 0.f;
    }
    if (bias) s += bias[n];
    float* dst = C + (long)m * ldc + n;
    if (accumulate) s += *dst;
    *dst = s;
  }
}
