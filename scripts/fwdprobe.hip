// What limits the wide forward kernel (3.3 TB/s)?  Replays its exact global-load pattern (512 threads, W tile
// 128 rows x 128 B fp32 + X hi/lo tiles 128 rows x 64 B bf16 per K-step, 40 N-tiles x 12 K-slices) with the
// on-chip work added stage by stage:  0 = W only, 1 = W + X, 2 = + LDS stash + barrier, 3 = + ds_read of the
// fragments (no MFMA), 4 = + MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int LEVEL, int XT>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ W, const unsigned short* __restrict__ Xh,
                                             const unsigned short* __restrict__ Xl, float* out, int H, int F, int splitk, int steps) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[2][4 * 128 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int lin = blockIdx.x;
  const int z = lin % splitk; lin /= splitk;
  const int n0 = lin * 128, k_begin = z * steps * 32;
  const int wn = tid >> 3, k4 = tid & 7;
  long r0 = n0 + wn, r1 = n0 + wn + 64; if (r0 >= H) r0 = H - 1; if (r1 >= H) r1 = H - 1;
  const float* w0 = W + r0 * (long)F + k_begin + 4 * k4;
  const float* w1 = W + r1 * (long)F + k_begin + 4 * k4;
  const int xr = tid >> 2, xc = tid & 3;
  // XT: tile-major X ([K/32][128 rows][32]) -> one K-step tile is 8 KB contiguous instead of 128 half cache lines
  const unsigned short* xh = XT ? Xh + ((long)(k_begin / 32) * 128 + xr) * 32 + 8 * xc : Xh + (long)xr * F + k_begin + 8 * xc;
  const unsigned short* xl = XT ? Xl + ((long)(k_begin / 32) * 128 + xr) * 32 + 8 * xc : Xl + (long)xr * F + k_begin + 8 * xc;
  const int xstep = XT ? 128 * 32 : 32;
  u32x4 a0, a1, b0, b1, c0, c1, d0, d1;
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  float sink = 0.f;
#define LD(A, B, C, D, kt) { A = *(const u32x4*)(w0 + (kt) * 32); B = *(const u32x4*)(w1 + (kt) * 32); \
    if (LEVEL >= 1) { C = *(const u32x4*)(xh + (kt) * xstep); D = *(const u32x4*)(xl + (kt) * xstep); } }
#define USE(A, B, C, D, buf) { \
    if (LEVEL >= 2) { \
      *(u32x4*)&smem[buf][tid * 8] = A; *(u32x4*)&smem[buf][4096 + tid * 8] = B; \
      *(u32x4*)&smem[buf][8192 + tid * 8] = C; *(u32x4*)&smem[buf][12288 + tid * 8] = D; \
    } else { sink += __builtin_bit_cast(float, A.x ^ B.y) ; if (LEVEL >= 1) sink += __builtin_bit_cast(float, C.x ^ D.y); } }
#define COMP(buf) { if (LEVEL >= 3) { \
      const int ar = (wid & 1) * 64 + (lane & 31), br = (wid >> 1) * 32 + (lane & 31), kh = lane >> 5; \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { \
        const bf16x8 ah0 = *(const bf16x8*)&smem[buf][(ar * 4 + ((2 * ks + kh) ^ ((ar >> 1) & 3))) * 8]; \
        const bf16x8 al0 = *(const bf16x8*)&smem[buf][4096 + (ar * 4 + ((2 * ks + kh) ^ ((ar >> 1) & 3))) * 8]; \
        const bf16x8 ah1 = *(const bf16x8*)&smem[buf][((ar + 32) * 4 + ((2 * ks + kh) ^ (((ar + 32) >> 1) & 3))) * 8]; \
        const bf16x8 al1 = *(const bf16x8*)&smem[buf][4096 + ((ar + 32) * 4 + ((2 * ks + kh) ^ (((ar + 32) >> 1) & 3))) * 8]; \
        const bf16x8 bh = *(const bf16x8*)&smem[buf][8192 + (br * 4 + ((2 * ks + kh) ^ ((br >> 1) & 3))) * 8]; \
        const bf16x8 bl = *(const bf16x8*)&smem[buf][12288 + (br * 4 + ((2 * ks + kh) ^ ((br >> 1) & 3))) * 8]; \
        if (LEVEL >= 4) { \
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh, acc1, 0, 0, 0); \
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl, acc1, 0, 0, 0); \
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc1, 0, 0, 0); \
        } else { sink += (float)ah0[0] + (float)al0[1] + (float)ah1[2] + (float)al1[3] + (float)bh[4] + (float)bl[5]; } \
      } } }
  LD(a0, a1, c0, c1, 0);
  LD(b0, b1, d0, d1, 1);
  USE(a0, a1, c0, c1, 0);
  if (LEVEL >= 2) __syncthreads();
  for (int kt = 0; kt < steps; kt += 2) {
    LD(a0, a1, c0, c1, kt + 2);
    COMP(0);
    USE(b0, b1, d0, d1, 1);
    if (LEVEL >= 2) __syncthreads();
    LD(b0, b1, d0, d1, kt + 3);
    COMP(1);
    USE(a0, a1, c0, c1, 0);
    if (LEVEL >= 2) __syncthreads();
  }
  for (int i = 0; i < 16; ++i) sink += acc0[i] + acc1[i];
  if (sink == 12345.678f) out[blockIdx.x] = sink;
}

int main() {
  const int H = 5000, F = 20000, splitk = 12, steps = 52;
  float *W, *W2, *out; unsigned short *Xh, *Xl;
  CK(hipMalloc(&W, (size_t)H * F * 4 + (1 << 22))); CK(hipMalloc(&W2, (size_t)H * F * 4 + (1 << 22))); CK(hipMalloc(&out, 1 << 20));
  CK(hipMalloc(&Xh, (size_t)128 * F * 2 + (1 << 20))); CK(hipMalloc(&Xl, (size_t)128 * F * 2 + (1 << 20)));
  CK(hipMemset(W, 0, (size_t)H * F * 4)); CK(hipMemset(W2, 0, (size_t)H * F * 4));
  CK(hipMemset(Xh, 0, (size_t)128 * F * 2)); CK(hipMemset(Xl, 0, (size_t)128 * F * 2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 40 * splitk;
  auto run = [&](const char* name, auto kern) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, (i & 1) ? W : W2, Xh, Xl, out, H, F, splitk, steps);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int n = 20;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, (i & 1) ? W : W2, Xh, Xl, out, H, F, splitk, steps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid * steps * 16384;
    printf("%-44s %8.1f us  %6.2f TB/s of W\n", name, ms / n * 1e3, bytes / (ms / n * 1e-3) / 1e12);
  };
  run("0: W only", (probe<0, 0>));
  run("1: W + X(L2) row-major", (probe<1, 0>));
  run("1: W + X(L2) tile-major", (probe<1, 1>));
  run("2: + LDS stash + barrier  (row-major)", (probe<2, 0>));
  run("2: + LDS stash + barrier  (tile-major)", (probe<2, 1>));
  run("4: + ds_reads + MFMA (row-major)", (probe<4, 0>));
  run("4: + ds_reads + MFMA (tile-major)", (probe<4, 1>));
  return 0;
}
