// Does a wide weight's W / m / v run at the good rate when its physical backing is SCATTERED by construction?  HIP virtual memory management:
// one contiguous virtual range per array, backed by 2 MB (or larger) physical chunks created one by one and mapped in a permuted order.
//   hipcc --offload-arch=gfx950 -O3 scripts/vmmprobe.hip -o /tmp/vmmprobe && /tmp/vmmprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void adam_runs(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V, int H, int F, long ld, int S) {
  __shared__ char pad[65536];
  if (threadIdx.x == 9999) pad[threadIdx.x] = 1;
  const int tiles_m = (H + 63) / 64, tiles_n = (F + 127) / 128;
  const int tm = blockIdx.x % tiles_m, c = blockIdx.x / tiles_m;
  for (int tn = c; tn < tiles_n; tn += S) {
    f4 p[4], m[4], v[4]; long off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = threadIdx.x + 512 * i, r = u >> 5, c4 = u & 31;
      const int row = tm * 64 + r, col = tn * 128 + 4 * c4;
      off[i] = (row < H && col < F) ? ((long)row * ld + col) >> 2 : -1;
      if (off[i] >= 0) { p[i] = __builtin_nontemporal_load((const f4*)W + off[i]); m[i] = __builtin_nontemporal_load((const f4*)M + off[i]); v[i] = __builtin_nontemporal_load((const f4*)V + off[i]); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (off[i] < 0) continue;
      __builtin_nontemporal_store(p[i] * 1.0001f, (f4*)W + off[i]); __builtin_nontemporal_store(m[i] * 0.9f, (f4*)M + off[i]); __builtin_nontemporal_store(v[i] * 0.999f, (f4*)V + off[i]);
    }
  }
}
__global__ void fill(float* p, long n, unsigned seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (float)(x >> 8) * (1.0f / 16777216.0f) * 0.02f + 1e-4f;
  }
}

static float* vmm_alloc(size_t bytes, size_t chunk, int mode, unsigned seed) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  if (chunk < gran) chunk = gran;
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t n = (bytes + chunk - 1) / chunk;
  void* va; CK(hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0));
  std::vector<hipMemGenericAllocationHandle_t> h(n);
  for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = i;
  if (mode == 1) { std::mt19937 rng(seed); std::shuffle(order.begin(), order.end(), rng); }
  if (mode == 2) std::reverse(order.begin(), order.end());
  for (size_t i = 0; i < n; ++i) CK(hipMemMap((char*)va + i * chunk, chunk, 0, h[order[i]], 0));
  hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, n * chunk, &acc, 1));
  return (float*)va;
}

int main() {
  const int H = 5000, F = 20000; const long ld = 20000; const size_t bytes = (size_t)H * ld * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, float* W, float* M, float* V) {
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, W, (long)H * ld, 1u);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, M, (long)H * ld, 2u);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, V, (long)H * ld, 3u);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(adam_runs, dim3(79 * 6), dim3(512), 0, 0, W, M, V, H, F, ld, 6);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(adam_runs, dim3(79 * 6), dim3(512), 0, 0, W, M, V, H, F, ld, 6);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-70s %8.1f us  %6.3f TB/s\n", name, ms / 10 * 1e3, 6.0 * H * F * 4 / (ms / 10 * 1e-3) / 1e12);
    fflush(stdout);
  };
  float *W, *M, *V;
  CK(hipMalloc(&W, bytes)); CK(hipMalloc(&M, bytes)); CK(hipMalloc(&V, bytes));
  run("hipMalloc x 3 (fresh process)", W, M, V);
  const size_t chunks[] = {2u << 20, 8u << 20, 32u << 20, 128u << 20};
  for (size_t ch : chunks)
    for (int mode = 0; mode < 3; ++mode) {
      char name[128];
      snprintf(name, sizeof name, "VMM chunks of %3zu MB, %s", ch >> 20, mode == 0 ? "mapped in creation order" : mode == 1 ? "mapped SHUFFLED" : "mapped reversed");
      float* w = vmm_alloc(bytes, ch, mode, 11); float* m = vmm_alloc(bytes, ch, mode, 22); float* v = vmm_alloc(bytes, ch, mode, 33);
      run(name, w, m, v);
      run(name, w, m, v);
    }
  return 0;
}
