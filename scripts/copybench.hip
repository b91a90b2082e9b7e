// HBM read+write ceiling probe: variants of a plain device-to-device copy (1 GiB -> 1 GiB), to quote next to the fused
// dW+Adam kernel's achieved rate.   hipcc --offload-arch=gfx950 -O3 scripts/copybench.hip -o /tmp/copybench && /tmp/copybench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void copy_flat(f4* __restrict__ d, const f4* __restrict__ s, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) { if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i); else d[i] = s[i]; }
}
// each block owns a contiguous 16 KB chunk, 4 loads in flight per thread
template <bool NT>
__global__ __launch_bounds__(256) void copy_chunk4(f4* __restrict__ d, const f4* __restrict__ s, long n4) {
  const long b = (long)blockIdx.x * 1024 + threadIdx.x;
  f4 v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) if (b + 256 * j < n4) v[j] = NT ? __builtin_nontemporal_load(s + b + 256 * j) : s[b + 256 * j];
#pragma unroll
  for (int j = 0; j < 4; ++j) if (b + 256 * j < n4) { if (NT) __builtin_nontemporal_store(v[j], d + b + 256 * j); else d[b + 256 * j] = v[j]; }
}
// grid-stride, UNROLL loads in flight
template <bool NT, int U>
__global__ __launch_bounds__(256) void copy_stride(f4* __restrict__ d, const f4* __restrict__ s, long n4) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = NT ? __builtin_nontemporal_load(s + i + j * stride) : s[i + j * stride];
#pragma unroll
    for (int j = 0; j < U; ++j) { if (NT) __builtin_nontemporal_store(v[j], d + i + j * stride); else d[i + j * stride] = v[j]; }
  }
  for (; i < n4; i += stride) d[i] = s[i];
}
// persistent blocks, each walking its own contiguous slice in 16 KB steps
template <bool NT>
__global__ __launch_bounds__(256) void copy_slice(f4* __restrict__ d, const f4* __restrict__ s, long n4) {
  const long per = (n4 / gridDim.x + 1023) / 1024 * 1024;
  const long lo = (long)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
  for (long b = lo + threadIdx.x; b < hi; b += 1024) {
    f4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (b + 256 * j < hi) v[j] = NT ? __builtin_nontemporal_load(s + b + 256 * j) : s[b + 256 * j];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (b + 256 * j < hi) { if (NT) __builtin_nontemporal_store(v[j], d + b + 256 * j); else d[b + 256 * j] = v[j]; }
  }
}

int main() {
  const long n = 1L << 28, n4 = n / 4;
  float *a, *b;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int it = 10;
    for (int i = 0; i < it; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.1f us  %6.3f TB/s (read+write)\n", name, ms / it * 1e3, 2.0 * n * 4 / (ms / it * 1e-3) / 1e12);
  };
  f4 *d = (f4*)b; const f4* s = (const f4*)a;
  run("hipMemcpyDtoD", [&] { CK(hipMemcpyAsync(b, a, n * 4, hipMemcpyDeviceToDevice, 0)); });
  run("flat 1 x 16 B / thread", [&] { hipLaunchKernelGGL(copy_flat<false>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, d, s, n4); });
  run("flat 1 x 16 B / thread, nt", [&] { hipLaunchKernelGGL(copy_flat<true>, dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, d, s, n4); });
  run("chunk 4 x 16 B / thread", [&] { hipLaunchKernelGGL(copy_chunk4<false>, dim3((unsigned)(n4 / 1024)), dim3(256), 0, 0, d, s, n4); });
  run("chunk 4 x 16 B / thread, nt", [&] { hipLaunchKernelGGL(copy_chunk4<true>, dim3((unsigned)(n4 / 1024)), dim3(256), 0, 0, d, s, n4); });
  for (int g : {1024, 2048, 4096, 8192}) {
    char nm[96];
    snprintf(nm, 96, "grid-stride x4, %d blocks", g);
    run(nm, [&] { hipLaunchKernelGGL((copy_stride<false, 4>), dim3(g), dim3(256), 0, 0, d, s, n4); });
    snprintf(nm, 96, "grid-stride x4, %d blocks, nt", g);
    run(nm, [&] { hipLaunchKernelGGL((copy_stride<true, 4>), dim3(g), dim3(256), 0, 0, d, s, n4); });
    snprintf(nm, 96, "grid-stride x8, %d blocks", g);
    run(nm, [&] { hipLaunchKernelGGL((copy_stride<false, 8>), dim3(g), dim3(256), 0, 0, d, s, n4); });
    snprintf(nm, 96, "slices, %d blocks", g);
    run(nm, [&] { hipLaunchKernelGGL(copy_slice<false>, dim3(g), dim3(256), 0, 0, d, s, n4); });
    snprintf(nm, 96, "slices, %d blocks, nt", g);
    run(nm, [&] { hipLaunchKernelGGL(copy_slice<true>, dim3(g), dim3(256), 0, 0, d, s, n4); });
  }
  return 0;
}
