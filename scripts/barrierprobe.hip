// What does a phase boundary cost?  (a) a kernel boundary between two dependent small kernels inside a hipGraph, (b) a grid-wide
// barrier inside one persistent kernel (one atomic counter, agent-scope release / acquire around it so that every workgroup sees the
// others' global writes -- what a fused "narrow-kernel tail" would need between its phases), (c) the same barrier among the workgroups of
// ONE XCD only.  Each phase does a token amount of work: every thread reads a value another workgroup wrote in the previous phase.
//   hipcc --offload-arch=gfx950 -O3 scripts/barrierprobe.hip -o build_tmp/barrierprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void phase_kernel(float* buf, int phase, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int j = (i + 256 * 37) % n;                       // another workgroup's element
  buf[(phase & 1) * n + i] = buf[((phase + 1) & 1) * n + j] + 1.0f;
}

__device__ __forceinline__ void grid_barrier(unsigned* cnt, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                        // release: this workgroup's writes reach memory visible to the agent
    atomicAdd(cnt, 1u);
    long spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 50000000L) break;                       // bounded: never hang the box
    }
    __threadfence();
  }
  __syncthreads();
}

// xcd_only: workgroups not on XCC 0 exit at once; the barrier then counts the remaining ones
__global__ __launch_bounds__(256) void persistent_kernel(float* buf, unsigned* cnt, unsigned* members, int phases, int n, int xcd_only) {
  unsigned G = gridDim.x;
  if (xcd_only) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    if (xcc != 0) return;
    // census of the workgroups that stay: every one adds itself, then waits until all launched workgroups have either left or joined
    if (threadIdx.x == 0) atomicAdd(members, 1u);
    G = 0;                                                  // filled in below by the host-provided count (members is read after a delay)
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int j = (i + 256 * 37) % n;
  if (xcd_only) {
    // simple: the host launched exactly 8x the wanted count and knows gridDim/8 land on each XCD (round-robin dispatch)
    G = gridDim.x / 8;
  }
  for (int p = 0; p < phases; ++p) {
    buf[(p & 1) * n + i] = __builtin_nontemporal_load(&buf[((p + 1) & 1) * n + j]) + 1.0f;
    grid_barrier(cnt, G * (unsigned)(p + 1));
  }
}

int main() {
  const int G = 256, n = G * 256, phases = 200;
  float* buf; unsigned *cnt, *members;
  CK(hipMalloc(&buf, 2 * n * 4)); CK(hipMemset(buf, 0, 2 * n * 4)); CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&members, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  // (a) kernel boundaries: eager stream and hipGraph
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, st, buf, i, n);
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < phases; ++i) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, st, buf, i, n);
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%d dependent 256-workgroup kernels, eager stream:   %7.2f us per phase\n", phases, ms * 1e3 / phases);
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < phases; ++i) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, st, buf, i, n);
  CK(hipStreamEndCapture(st, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(exec, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%d dependent 256-workgroup kernels, one hipGraph:   %7.2f us per phase\n", phases, ms * 1e3 / phases);
  // (b) persistent kernel with grid barriers
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemsetAsync(cnt, 0, 4, st));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(persistent_kernel, dim3(G), dim3(256), 0, st, buf, cnt, members, phases, n, 0);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("one persistent kernel, %d grid barriers (256 WGs):  %7.2f us per phase\n", phases, ms * 1e3 / phases);
  // (c) barrier among the workgroups of one XCD (256 launched, 32 stay)
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemsetAsync(cnt, 0, 4, st)); CK(hipMemsetAsync(members, 0, 4, st));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(persistent_kernel, dim3(G), dim3(256), 0, st, buf, cnt, members, phases, n, 1);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  unsigned hm = 0; CK(hipMemcpy(&hm, members, 4, hipMemcpyDeviceToHost));
  printf("persistent kernel on ONE XCD (%u of 256 WGs stayed): %6.2f us per phase\n", hm, ms * 1e3 / phases);
  return 0;
}
