// Does reading the first columns of W / m / v while HBM is idle (the step's narrow-kernel tail) make the dW+Adam kernel's first
// tiles hit in the Infinity Cache?  Persistent-run Adam pattern of scripts/adamprobe.hip (79 row blocks x S = 6 runs over
// [5000, 20000] x 3 arrays), timed alone after (a) a 1 GB flush, (b) flush + a prefetch of the first P column tiles of every row.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void adam_runs(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V, int H, int F, long ld, int S) {
  constexpr int R = 64, C = 128, UPR = 32, PER = 4;
  const int tiles_m = (H + R - 1) / R, tiles_n = (F + C - 1) / C;
  const int tm = blockIdx.x % tiles_m, c = blockIdx.x / tiles_m;
  for (int tn = c; tn < tiles_n; tn += S) {
    f4 p[PER], m[PER], v[PER];
    long off[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = threadIdx.x + 512 * i, r = u / UPR, c4 = u % UPR;
      const int row = tm * R + r, col = tn * C + 4 * c4;
      off[i] = (row < H && col < F) ? ((long)row * ld + col) / 4 : -1;
      if (off[i] >= 0) {
        p[i] = __builtin_nontemporal_load((const f4*)W + off[i]);
        m[i] = __builtin_nontemporal_load((const f4*)M + off[i]);
        v[i] = __builtin_nontemporal_load((const f4*)V + off[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (off[i] < 0) continue;
      f4 po, mo, vo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = 1e-3f;
        const float m2 = m[i][j] + (g - m[i][j]) * 0.1f;
        const float v2 = v[i][j] * 0.999f + 0.001f * g * g;
        po[j] = p[i][j] - 1e-3f * (m2 / (sqrtf(v2) + 1e-8f));
        mo[j] = m2; vo[j] = v2;
      }
      __builtin_nontemporal_store(po, (f4*)W + off[i]);
      __builtin_nontemporal_store(mo, (f4*)M + off[i]);
      __builtin_nontemporal_store(vo, (f4*)V + off[i]);
    }
  }
}

// reads columns [0, cols) of every row of the three arrays (plain loads), one workgroup per 8 rows
__global__ __launch_bounds__(256) void prefetch_cols(const float* __restrict__ W, const float* __restrict__ M, const float* __restrict__ V, int H, int cols, long ld, float* sink) {
  float acc = 0.f;
  for (int r = blockIdx.x * 8; r < min(H, blockIdx.x * 8 + 8); ++r)
    for (int c4 = threadIdx.x; c4 < cols / 4; c4 += 256) {
      const long o = ((long)r * ld) / 4 + c4;
      const f4 a = ((const f4*)W)[o], b = ((const f4*)M)[o], c = ((const f4*)V)[o];
      acc += a[0] + b[1] + c[2];
    }
  if (acc == 12345.678f) sink[0] = acc;
}

__global__ void flush_k(const f4* __restrict__ p, long n, float* sink) {
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += p[i][0];
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const int H = 5000, F = 20000; const long ld = 20000;
  float *W, *M, *V, *big, *sink;
  const size_t bytes = (size_t)H * ld * 4, bigb = 1ull << 30;
  CK(hipMalloc(&W, bytes)); CK(hipMalloc(&M, bytes)); CK(hipMalloc(&V, bytes)); CK(hipMalloc(&big, bigb)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(W, 0, bytes)); CK(hipMemset(M, 0, bytes)); CK(hipMemset(V, 0, bytes)); CK(hipMemset(big, 0, bigb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int tiles : {0, 0, 6, 12, 24, 48, 0}) {
    float tot = 0.f; const int it = 6;
    for (int i = 0; i < it + 1; ++i) {
      hipLaunchKernelGGL(flush_k, dim3(2048), dim3(256), 0, 0, (const f4*)big, (long)(bigb / 16), sink);
      if (tiles) hipLaunchKernelGGL(prefetch_cols, dim3((H + 7) / 8), dim3(256), 0, 0, W, M, V, H, tiles * 128, ld, sink);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(adam_runs, dim3(79 * 6), dim3(512), 0, 0, W, M, V, H, F, ld, 6);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (i) tot += ms;
    }
    printf("prefetch of the first %2d column tiles (%6.1f MB): Adam-pattern kernel %7.1f us\n", tiles, 3.0 * H * tiles * 512 / 1e6, tot / it * 1e3);
  }
  return 0;
}
