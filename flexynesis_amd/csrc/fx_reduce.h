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

// Four consecutive outputs per thread (N, ldc, slab_stride multiples of 4; 16-byte aligned bases): 16-byte loads, and
// 32-bit index arithmetic instead of a 64-bit division per element.  Same summation order as the scalar kernel.
static __global__ void fx_reduce_slabs4_kernel(float* __restrict__ C, const float* __restrict__ slabs,
                                               const float* __restrict__ bias, int M, int N, long ldc, int splitk,
                                               long slab_stride, int accumulate) {
  const unsigned total4 = (unsigned)(((long)M * N) >> 2);
  const unsigned n4 = (unsigned)N >> 2;
  for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += gridDim.x * blockDim.x) {
    const unsigned m = q / n4, n = (q - m * n4) << 2;
    const float* src = slabs + ((long)q << 2);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 4 <= splitk; z += 4) {
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(src + (long)(z + j) * slab_stride);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
    }
    {
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(src + (long)min(z + j, splitk - 1) * slab_stride);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float k = (z + j < splitk) ? 1.f : 0.f;
        s.x += k * v[j].x; s.y += k * v[j].y; s.z += k * v[j].z; s.w += k * v[j].w;
      }
    }
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + n);
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    float4* dst = reinterpret_cast<float4*>(C + (long)m * ldc + n);
    if (accumulate) {
      const float4 p = *dst;
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    *dst = s;
  }
}

static inline void fx_launch_reduce_slabs(float* C, const float* slabs, const float* bias, int M, int N, long ldc, int splitk,
                                          long slab_stride, int accumulate, hipStream_t stream) {
  const long total = (long)M * N;
  const bool v4 = (N % 4 == 0) && (ldc % 4 == 0) && (slab_stride % 4 == 0) && total < (1L << 31) &&
                  ((((uintptr_t)C) | ((uintptr_t)slabs) | ((uintptr_t)bias)) & 15) == 0;
  if (v4) {
    const long t4 = total >> 2;
    const int blocks = (int)((t4 + 255) / 256 > 2048 ? 2048 : (t4 + 255) / 256);
    hipLaunchKernelGGL(fx_reduce_slabs4_kernel, dim3(blocks), dim3(256), 0, stream, C, slabs, bias, M, N, ldc, splitk,
                       slab_stride, accumulate);
  } else {
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(fx_reduce_slabs_kernel, dim3(blocks), dim3(256), 0, stream, C, slabs, bias, M, N, ldc, splitk,
                       slab_stride, accumulate);
  }
}
