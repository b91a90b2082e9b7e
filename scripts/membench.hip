// HBM access-pattern probe: does a 64-row x 128-byte strided tile stream (the wide forward GEMM's W operand)
// run slower than a blocked (8 KB contiguous per step) stream of the same bytes?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ W, float* out, int H, int F, int splitk, int steps, int rowbytes_per_step) {
  const int tid = threadIdx.x;
  int lin = blockIdx.x;
  const int z = lin % splitk; lin /= splitk;
  const int tn = lin;
  const int n0 = tn * 64, k_begin = z * steps * 32;
  float4 acc = make_float4(0, 0, 0, 0);
  float4 r[DEPTH][2];
  auto addr = [&](int kt, int i) -> const float4* {
    if (MODE == 0) {  // strided: row n = tid>>3 (+32), 128 B per row per step
      const int n = (tid >> 3) + 32 * i, k4 = tid & 7;
      long nn = n0 + n; if (nn >= H) nn = H - 1;
      return (const float4*)(W + nn * (long)F + k_begin + kt * 32 + 4 * k4);
    } else if (MODE == 1) {  // blocked: 8 KB contiguous per step per workgroup
      const long blk = ((long)blockIdx.x * steps + kt) * 2048;  // floats
      return (const float4*)(W + blk % ((long)H * F - 4096) + (tid + 256 * i) * 4);
    } else {  // MODE 2: strided, 256 B per row per step, 32 rows (same bytes)
      const int n = (tid >> 4) + 16 * i, k4 = tid & 15;
      long nn = n0 / 2 + n; if (nn >= H) nn = H - 1;
      return (const float4*)(W + nn * (long)F + (k_begin + kt * 32) * 2 % (F - 64) + 4 * k4);
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) { r[d][0] = *addr(d, 0); r[d][1] = *addr(d, 1); }
  for (int kt = 0; kt < steps; kt += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const float4 a = r[d][0], b = r[d][1];
      r[d][0] = *addr(kt + d + DEPTH, 0);
      r[d][1] = *addr(kt + d + DEPTH, 1);
      acc.x += a.x + b.x; acc.y += a.y + b.y; acc.z += a.z + b.z; acc.w += a.w + b.w;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

int main() {
  const int H = 5000, F = 20000, splitk = 8, steps = 79;
  float *W, *W2, *out;
  CK(hipMalloc(&W, (size_t)H * F * 4 + (1 << 22))); CK(hipMalloc(&W2, (size_t)H * F * 4 + (1 << 22))); CK(hipMalloc(&out, 1 << 20));
  CK(hipMemset(W, 0, (size_t)H * F * 4)); CK(hipMemset(W2, 0, (size_t)H * F * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 79 * splitk;
  auto run = [&](const char* name, auto kern) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (i & 1) ? W : W2, out, H, F, splitk, steps, 128);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int n = 20;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (i & 1) ? W : W2, out, H, F, splitk, steps, 128);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid * steps * 8192;
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms / n * 1e3, bytes / (ms / n * 1e-3) / 1e12);
  };
  run("strided 64 rows x 128 B, depth 2", probe<0, 2>);
  run("strided 64 rows x 128 B, depth 4", probe<0, 4>);
  run("strided 64 rows x 128 B, depth 8", probe<0, 8>);
  run("blocked 8 KB contiguous, depth 2", probe<1, 2>);
  run("blocked 8 KB contiguous, depth 4", probe<1, 4>);
  run("strided 32 rows x 256 B, depth 2", probe<2, 2>);
  run("strided 32 rows x 256 B, depth 4", probe<2, 4>);
  return 0;
}
