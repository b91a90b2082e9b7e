// Address-translation reach of the box: a single lane chases pointers through N cache lines spaced `stride` bytes apart
// (random cyclic order).  N * 64 B of DATA always fits the XCD's 4 MB L2, so after the warm-up pass every access is an L2 data
// hit and what is left of the latency differences between strides is the translation path (UTCL1 / UTCL2 / page walk): with
// 2 MB fragments a 1 GB footprint needs 512 translations, with 64 KB fragments 16384.  DESIGN.md section 3.7 ("pattern
// sensitivity is a property of the box") is tested against this: the dW + Adam kernel's persistent schedule touches every
// row of three [5000, 20000] arrays at once = ~15000 distinct 64 KB windows.
//   hipcc --offload-arch=gfx950 -O3 scripts/tlbprobe.hip -o /tmp/tlbprobe && /tmp/tlbprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void chase(const unsigned long long* start, long steps, unsigned long long* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const unsigned long long* p = start;
  for (long i = 0; i < steps; ++i) p = (const unsigned long long*)*p;   // warm-up lap(s)
  const unsigned long long t0 = wall_clock64();
  for (long i = 0; i < steps; ++i) p = (const unsigned long long*)*p;
  const unsigned long long t1 = wall_clock64();
  out[0] = t1 - t0;
  out[1] = (unsigned long long)p;
}

int main() {
  const size_t cap = 6ull << 30;
  char* buf; CK(hipMalloc(&buf, cap));
  CK(hipMemset(buf, 0, cap));
  unsigned long long* out; CK(hipMalloc(&out, 16));
  int dev; CK(hipGetDevice(&dev));
  int wc_khz = 0; CK(hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, dev));
  printf("wall clock %d kHz; buffer %p (%zu MB)\n", wc_khz, (void*)buf, cap >> 20);
  std::mt19937_64 rng(1);
  printf("%10s", "N \\ stride");
  const size_t strides[] = {4096, 16384, 65536, 262144, 2097152};
  for (size_t s : strides) printf("%10zuK", s >> 10);
  printf("   (ns per dependent access, data L2-resident)\n");
  for (long N : {64L, 256L, 1024L, 4096L, 16384L, 65536L}) {
    printf("%10ld", N);
    for (size_t s : strides) {
      if ((size_t)N * s > cap) { printf("%11s", "-"); continue; }
      std::vector<long> order(N);
      for (long i = 0; i < N; ++i) order[i] = i;
      std::shuffle(order.begin() + 1, order.end(), rng);
      // write next pointers: line order[i] -> line order[i + 1]; line offset inside the stride window varies so that the
      // lines do not all fall into the same L2 set / channel
      auto addr = [&](long line) { return (unsigned long long)(buf + (size_t)line * s + ((line * 192) % (s < 4096 ? s : 4096))); };
      for (long i = 0; i < N; ++i) {
        const unsigned long long nxt = addr(order[(i + 1) % N]);
        CK(hipMemcpy((void*)addr(order[i]), &nxt, 8, hipMemcpyHostToDevice));
      }
      const long steps = N < 4096 ? 8192 : 2 * N;
      hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, (const unsigned long long*)addr(order[0]), steps, out);
      CK(hipDeviceSynchronize());
      unsigned long long h[2]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
      printf("%11.0f", (double)h[0] / steps * 1e6 / wc_khz);
    }
    printf("\n");
  }
  return 0;
}
