// Standalone probe for the hipGraphLaunch crash of DESIGN.md section 4.1 / 3.11 (a step graph with a THIRD forked branch segfaulted in
// hipGraphLaunch, only after level-1-like small graphs had been captured in the same process).  No torch: plain HIP runtime.
//   hipcc --offload-arch=gfx950 -O2 -o build_tmp/graphprobe scripts/graphprobe.hip && ./build_tmp/graphprobe [branches] [rounds] [nested]
// Shape of the step graph: origin stream: k0 -> fork -> `branches` side streams (each a chain of 3-6 kernels; with `nested` the first
// side branch forks a branch of its own and joins it) -> join -> two HBM-sized kernels.  Before it: three single-stream graphs of 5-20
// kernels each (the level-1 tapes), instantiated, launched, and either kept or destroyed (argument 4: 0 keep, 1 destroy before the step
// graph is captured, 2 destroy after it was instantiated, 3 destroy between its launches).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void small_k(float* p, int n, float a) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * a + 1.0f; }
__global__ void big_k(float* p, long n) { for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] += 1.0f; }

static hipGraphExec_t capture_chain(hipStream_t s, float* buf, int kernels) {
  hipGraph_t g; hipGraphExec_t e;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < kernels; ++i) hipLaunchKernelGGL(small_k, dim3(16), dim3(256), 0, s, buf, 4096, 1.0f);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
  CK(hipGraphDestroy(g));
  return e;
}

int main(int argc, char** argv) {
  const int branches = argc > 1 ? atoi(argv[1]) : 3, rounds = argc > 2 ? atoi(argv[2]) : 200, nested = argc > 3 ? atoi(argv[3]) : 1;
  const int destroy_mode = argc > 4 ? atoi(argv[4]) : 0;
  float *small, *big;
  const long nbig = 100L << 20;
  CK(hipMalloc(&small, 1 << 20)); CK(hipMalloc(&big, nbig * 4));
  CK(hipMemset(small, 0, 1 << 20)); CK(hipMemset(big, 0, nbig * 4));
  hipStream_t origin; CK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));
  std::vector<hipStream_t> side(branches + 1);
  for (auto& s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // level-1-like graphs
  std::vector<hipGraphExec_t> tapes;
  for (int k : {5, 12, 20}) tapes.push_back(capture_chain(origin, small, k));
  for (int r = 0; r < 3; ++r) for (auto e : tapes) CK(hipGraphLaunch(e, origin));
  CK(hipStreamSynchronize(origin));
  if (destroy_mode == 1) { for (auto e : tapes) CK(hipGraphExecDestroy(e)); tapes.clear(); }
  // the step graph
  hipGraph_t g; hipGraphExec_t step;
  std::vector<hipEvent_t> ev(2 * branches + 4);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  CK(hipStreamBeginCapture(origin, hipStreamCaptureModeGlobal));
  hipLaunchKernelGGL(small_k, dim3(16), dim3(256), 0, origin, small, 4096, 1.0f);
  CK(hipEventRecord(ev[0], origin));
  for (int b = 0; b < branches; ++b) {
    CK(hipStreamWaitEvent(side[b], ev[0], 0));
    for (int i = 0; i < 3 + b; ++i) hipLaunchKernelGGL(small_k, dim3(16 + 8 * b), dim3(256), 0, side[b], small + 8192 * (b + 1), 4096, 1.0f);
    if (nested && b == 0) {
      CK(hipEventRecord(ev[1], side[0]));
      CK(hipStreamWaitEvent(side[branches], ev[1], 0));
      for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(small_k, dim3(8), dim3(256), 0, side[branches], small + 8192 * 7, 4096, 1.0f);
      CK(hipEventRecord(ev[2], side[branches]));
      hipLaunchKernelGGL(big_k, dim3(512), dim3(512), 0, side[0], big, nbig / 4);
      CK(hipStreamWaitEvent(side[0], ev[2], 0));
    }
    CK(hipEventRecord(ev[4 + b], side[b]));
  }
  for (int b = 0; b < branches; ++b) CK(hipStreamWaitEvent(origin, ev[4 + b], 0));
  hipLaunchKernelGGL(big_k, dim3(512), dim3(512), 0, origin, big, nbig);
  hipLaunchKernelGGL(big_k, dim3(512), dim3(512), 0, origin, big, nbig);
  CK(hipStreamEndCapture(origin, &g));
  size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
  CK(hipGraphInstantiate(&step, g, nullptr, nullptr, 0));
  CK(hipGraphDestroy(g));
  if (destroy_mode == 2) { for (auto e : tapes) CK(hipGraphExecDestroy(e)); tapes.clear(); }
  for (int r = 0; r < rounds; ++r) {
    CK(hipGraphLaunch(step, origin));
    if (!tapes.empty() && r % 3 == 0) CK(hipGraphLaunch(tapes[r % tapes.size()], origin));
    if (destroy_mode == 3 && r == rounds / 2) { CK(hipStreamSynchronize(origin)); for (auto e : tapes) CK(hipGraphExecDestroy(e)); tapes.clear(); }
    if (r % 50 == 49) CK(hipStreamSynchronize(origin));
  }
  CK(hipStreamSynchronize(origin));
  float h[4]; CK(hipMemcpy(h, big, 16, hipMemcpyDeviceToHost));
  printf("branches %d nested %d destroy_mode %d: %zu nodes, %d rounds ok, big[0] = %.0f (expected %d)\n", branches, nested, destroy_mode, nn, rounds, h[0],
         rounds * (2 + (nested ? 1 : 0)));
  return 0;
}
