// What does v_mfma_f32_32x32x16_bf16 sustain in the shapes the split-bf16 kernels use?  One workgroup per CU, 4 or 8 waves,
// NACC accumulators visited round-robin (distance between dependent MFMAs = NACC) or in (a0, a1) x 3 pairs, with or without a
// ds_read_b128 per MFMA.   hipcc --offload-arch=gfx950 -O3 scripts/mfmaprobe.hip -o build_tmp/mfmaprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int MODE, int LDS>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[LDS ? 32768 : 8];
  const int tid = threadIdx.x;
  if (LDS) for (int i = tid; i < 32768; i += blockDim.x) smem[i] = (__bf16)(0.001f * (i & 255));
  __syncthreads();
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  bf16x8 fa, fb;
#pragma unroll
  for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(0.01f * (tid + j)); fb[j] = (__bf16)(0.02f * (tid - j)); }
  const __bf16* base = smem + (tid & 63) * 8;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {                         // round-robin: dependent distance NACC
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
          if (LDS) fa = *reinterpret_cast<const bf16x8*>(base + ((r * NACC + a) & 31) * 512);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[a], 0, 0, 0);
        }
    } else {                                 // pairs: (a0, a1) x 3, then (a2, a3) x 3 ...
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int a = 0; a < NACC; a += 2)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            if (LDS) fa = *reinterpret_cast<const bf16x8*>(base + ((r * 7 + a + q) & 31) * 512);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[a], 0, 0, 0);
            acc[a + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc[a + 1], 0, 0, 0);
          }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[a][i];
  out[blockIdx.x * blockDim.x + tid] = s;
}

int main() {
  float* out; CK(hipMalloc(&out, 1024 * 512 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  auto run = [&](const char* name, auto launch, int threads, int grid) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double mfma = (double)grid * (threads / 64) * iters * 36.0;
    printf("%-72s %8.1f us  %6.3f PFLOP/s  %5.1f cyc/MFMA/SIMD at 2.1 GHz\n", name, ms * 1e3, mfma * 32768 / (ms * 1e-3) / 1e15,
           (ms * 1e-3) * 2.1e9 / (mfma / (grid < 256 ? grid : 256) / 4));
  };
#define RUN(NACC, MODE, LDS, T, G, label) run(label, [&] { hipLaunchKernelGGL((probe<NACC, MODE, LDS>), dim3(G), dim3(T), 0, 0, out, iters); }, T, G);
  RUN(6, 0, 0, 256, 256, "4 waves/CU, 6 acc round-robin, registers only")
  RUN(6, 0, 0, 512, 256, "8 waves/CU, 6 acc round-robin, registers only")
  RUN(6, 1, 0, 512, 256, "8 waves/CU, (a0,a1)x3 pairs, registers only")
  RUN(2, 0, 0, 512, 256, "8 waves/CU, 2 acc round-robin, registers only")
  RUN(1, 0, 0, 512, 256, "8 waves/CU, 1 acc (fully dependent), registers only")
  RUN(6, 0, 1, 512, 256, "8 waves/CU, 6 acc round-robin, one ds_read_b128 per MFMA")
  RUN(6, 1, 1, 512, 256, "8 waves/CU, (a0,a1)x3 pairs, one ds_read_b128 per 2 MFMA")
  RUN(6, 0, 0, 512, 236, "8 waves/CU, 6 acc round-robin, registers only, 236 workgroups")
  RUN(6, 0, 0, 512, 512, "8 waves/CU x 2 rounds (512 workgroups), 6 acc round-robin")
  return 0;
}
