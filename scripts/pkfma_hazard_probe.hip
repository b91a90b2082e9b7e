// Packed-fp32 hazard probe (gfx950).  Victim: the engine's small-linear forward exactly as the compiler builds it with packed
// fp32 ops on -- broadcast ds_read2_b32 pairs, a v_mov_b32 that assembles the (row i, row i+1) operand pair, then
// v_pk_fma_f32 ... op_sel -- launched 20000 times on static inputs and compared on the device with its first result.
// Aggressor: another kernel looping on a second stream.  Measured on MI355X (ROCm 7.2, wrong outputs per 20000 launches):
//     plain / LDS-DMA staging + ds_read_b128, no MFMA ............ 0        MFMA from registers only ........... 0
//     ds_read_b128 -> 8 dependent MFMA, barrier each side ........ 0
//     ds_read_b128 -> 12 MFMA on two accumulators (the engine's K loops), with LDS-DMA staging 208-272, with register
//     staging 480-656, with no staging at all 1136
// Every wrong output is ONE dropped term x[r, k] * W[o, k] (k odd: the operand that went through the v_mov) in lanes 48..63
// of one wave: the low source of the v_pk_fma_f32 read as 0 in the last quarter of the wave.  It is not LDS-DMA, not LDS
// contents (a victim that only checks its LDS sees nothing) and not memory: the same victim built with
// -Xclang -target-feature -Xclang -packed-fp32-ops (v_fmac_f32, same ds_read2_b32) gives 0 in every mode.  The engine's
// library is therefore built without packed fp32 ops (csrc/build.py; tests/test_host_cpu.py guards it;
// tests/test_gpu_soak.py::test_small_fp32_kernels_beside_an_mfma_kernel_are_exact is the regression).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/pkfma_hazard_probe.hip -o build_tmp/pkfma_probe
//        (add -Xclang -target-feature -Xclang -packed-fp32-ops for the control)
// Aggressor modes 0-5 (simple loop): 0 plain loads + ds_write + ds_read_b128, 1 LDS-DMA + ds_read_b128, 2 ds_read_b128 + MFMA,
// 3 LDS-DMA + ds_read_b128 + MFMA, 4 LDS-DMA only, 5 MFMA only; modes 6-10: the double-buffered K loop (aggressor_db).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float floatx16;

template <int MODE>
__global__ __launch_bounds__(512) void aggressor(const float* src, long n_floats, int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (unsigned)(n_floats * 4), 0x00020000);
  floatx16 acc = {0};
  float facc = 0.f;
  constexpr bool STAGE_PLAIN = MODE == 0, STAGE_DMA = MODE == 1 || MODE == 3 || MODE == 4;
  constexpr bool READ = MODE <= 3, MFMA = MODE == 2 || MODE == 3 || MODE == 5;
  if (!STAGE_PLAIN && !STAGE_DMA && READ) { for (int i = tid; i < 8192; i += 512) lds[i] = 1.0f; }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    for (int c = 0; c < 4; ++c) {
      const unsigned off = ((unsigned)((((long)blockIdx.x * iters + it) * 8192 + c * 2048 + w * 256 + lane * 4) % (n_floats - 4)) * 4u) & ~15u;
      if (STAGE_PLAIN) {
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(src) + off);
        *reinterpret_cast<float4*>(&lds[c * 2048 + w * 256 + lane * 4]) = v;
      } else if (STAGE_DMA) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds + c * 2048 + w * 256), 16, off, 0, 0, 0);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      bf16x8_t a, b;
      if (READ) {
        a = *reinterpret_cast<const bf16x8_t*>(&lds[((q * 512 + tid) * 4) % 8192]);
        b = *reinterpret_cast<const bf16x8_t*>(&lds[((q * 512 + tid) * 4 + 4096) % 8192]);
      } else {
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(tid + q); b[e] = (__bf16)1.0f; }
      }
      if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      else facc += (float)a[0] + (float)b[3];
    }
    __syncthreads();
  }
  for (int e = 0; e < 16; ++e) facc += acc[e];
  if (facc == 123.456f) sink[0] = facc;
}

// double-buffered K loop like fx_gram_kb_group: the next stage's LDS-DMA (STAGING 1) or register-staged copy (STAGING 0) is in
// flight while the waves read the current stage with ds_read_b128 and run MFMAs
template <int STAGING>
__global__ __launch_bounds__(512) void aggressor_db(const float* src, long n_floats, int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) float lds[8192];              // 2 stages x 16 KB
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (unsigned)(n_floats * 4), 0x00020000);
  floatx16 acc0 = {0}, acc1 = {0};
  float4 held[2] = {};
  for (int i = tid; i < 8192; i += 512) lds[i] = 1.0f;
  auto request = [&](int stage, int it) {
    for (int c = 0; c < 2; ++c) {
      const unsigned off = ((unsigned)((((long)blockIdx.x * iters + it) * 4096 + c * 2048 + w * 256 + lane * 4) % (n_floats - 4)) * 4u) & ~15u;
      if (STAGING >= 2 && STAGING != 4) {}
      else if (STAGING == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds + stage * 4096 + c * 2048 + w * 256), 16, off, 0, 0, 0);
      else held[c] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(src) + off);
    }
  };
  auto land = [&](int stage) {
    if (STAGING == 0 || STAGING == 4) for (int c = 0; c < 2; ++c) *reinterpret_cast<float4*>(&lds[stage * 4096 + c * 2048 + w * 256 + lane * 4]) = held[c];
  };
  request(0, 0); land(0);
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (it + 1 < iters) request((it + 1) & 1, it + 1);
    const float* sb = lds + (it & 1) * 4096;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t ah, al, bh, bl;
      if (STAGING == 3) {
        for (int e = 0; e < 8; ++e) { ah[e] = (__bf16)(float)(tid + ks + it); al[e] = (__bf16)0.5f; bh[e] = (__bf16)(float)(lane); bl[e] = (__bf16)0.25f; }
      } else {
        ah = *reinterpret_cast<const bf16x8_t*>(sb + ((ks * 512 + tid) * 4) % 2048);
        al = *reinterpret_cast<const bf16x8_t*>(sb + 2048 + ((ks * 512 + tid) * 4) % 2048);
        bh = *reinterpret_cast<const bf16x8_t*>(sb + ((ks * 512 + tid * 3) * 4) % 2048);
        bl = *reinterpret_cast<const bf16x8_t*>(sb + 2048 + ((ks * 512 + tid * 3) * 4) % 2048);
      }
      if (STAGING == 4) { acc0[0] += (float)ah[0] + (float)bl[1]; acc1[1] += (float)al[2] + (float)bh[3]; continue; }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc1, 0, 0, 0);
    }
    if (it + 1 < iters) land((it + 1) & 1);
  }
  float facc = 0.f;
  for (int e = 0; e < 16; ++e) facc += acc0[e] + acc1[e];
  if (facc == 123.456f) sink[0] = facc;
}

#define SL_T 256
#define SL_TR 16
#define SL_TC 64
#define SL_KC 64
template <int ROWS>
__device__ __forceinline__ void sl_stage(float (*dst)[SL_KC + 1], const float* __restrict__ src, long ld, int r0, int nr, int c0, int nc) {
  constexpr int PER = ROWS * SL_KC / SL_T;
  float v[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int idx = threadIdx.x + SL_T * i, rr = idx / SL_KC, cc = idx % SL_KC;
    v[i] = src[(long)min(r0 + rr, nr - 1) * ld + min(c0 + cc, nc - 1)];
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int idx = threadIdx.x + SL_T * i, rr = idx / SL_KC, cc = idx % SL_KC;
    dst[rr][cc] = (r0 + rr < nr && c0 + cc < nc) ? v[i] : 0.f;
  }
}
__global__ __launch_bounds__(SL_T) void victim(float* y, const float* x, const float* W, const float* b, int R, int O, int K) {
  __shared__ float xs[SL_TR][SL_KC + 1];
  __shared__ float ws[SL_TC][SL_KC + 1];
  const int tiles_o = (O + SL_TC - 1) / SL_TC;
  const int r0 = (blockIdx.x / tiles_o) * SL_TR, o0 = (blockIdx.x % tiles_o) * SL_TC;
  const int ol = threadIdx.x & 63, rg = threadIdx.x >> 6;
  float acc[SL_TR / 4];
  const float bias = (o0 + ol < O) ? b[o0 + ol] : 0.f;
#pragma unroll
  for (int i = 0; i < SL_TR / 4; ++i) acc[i] = bias;
  for (int k0 = 0; k0 < K; k0 += SL_KC) {
    __syncthreads();
    sl_stage<SL_TR>(xs, x, K, r0, R, k0, K);
    sl_stage<SL_TC>(ws, W, K, o0, O, k0, K);
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < SL_KC; ++k) {
      const float wv = ws[ol][k];
#pragma unroll
      for (int i = 0; i < SL_TR / 4; ++i) acc[i] = fmaf(xs[rg + 4 * i][k], wv, acc[i]);
    }
  }
  if (o0 + ol < O)
#pragma unroll
    for (int i = 0; i < SL_TR / 4; ++i)
      if (r0 + rg + 4 * i < R) y[(long)(r0 + rg + 4 * i) * O + o0 + ol] = acc[i];
}
__global__ void compare(const float* y, const float* ref, int n, unsigned* bad, unsigned* bad_lanes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && y[i] != ref[i]) { atomicAdd(bad, 1u); atomicAdd(&bad_lanes[(i % 85) / 16], 1u); }
}

template <int MODE>
void run(const char* what, const float* src, long n, float* sink, float* y, float* ref, const float* x, const float* W, const float* b, unsigned* bad) {
  const int R = 128, O = 85, K = 170;
  hipStream_t sa, sv;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sv));
  CK(hipMemset(bad, 0, 64));
  const int wgs = ((R + 15) / 16) * 2;
  for (int rep = 0; rep < 20000; ++rep) {
    if (rep % 8 == 0) {
      if (MODE == 6) hipLaunchKernelGGL((aggressor_db<1>), dim3(1024), dim3(512), 0, sa, src, n, 48, sink);
      else if (MODE == 7) hipLaunchKernelGGL((aggressor_db<0>), dim3(1024), dim3(512), 0, sa, src, n, 48, sink);
      else if (MODE == 8) hipLaunchKernelGGL((aggressor_db<2>), dim3(1024), dim3(512), 0, sa, src, n, 48, sink);
      else if (MODE == 9) hipLaunchKernelGGL((aggressor_db<3>), dim3(1024), dim3(512), 0, sa, src, n, 48, sink);
      else if (MODE == 10) hipLaunchKernelGGL((aggressor_db<4>), dim3(1024), dim3(512), 0, sa, src, n, 48, sink);
      else hipLaunchKernelGGL((aggressor<MODE < 6 ? MODE : 0>), dim3(1024), dim3(512), 0, sa, src, n, 24, sink);
    }
    hipLaunchKernelGGL(victim, dim3(wgs), dim3(256), 0, sv, y, x, W, b, R, O, K);
    hipLaunchKernelGGL(compare, dim3((R * O + 255) / 256), dim3(256), 0, sv, y, ref, R * O, bad, bad + 1);
    if (rep % 512 == 511) CK(hipDeviceSynchronize());
  }
  CK(hipDeviceSynchronize());
  unsigned e[8];
  CK(hipMemcpy(e, bad, 32, hipMemcpyDeviceToHost));
  printf("%-52s wrong outputs %6u  by output column /16: %u %u %u %u %u %u\n", what, e[0], e[1], e[2], e[3], e[4], e[5], e[6]);
  CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sv));
}

int main() {
  const long n = 64L << 20;
  const int R = 128, O = 85, K = 170;
  float *src, *sink, *y, *ref, *x, *W, *b; unsigned* bad;
  CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&bad, 64));
  CK(hipMalloc(&y, R * O * 4)); CK(hipMalloc(&ref, R * O * 4)); CK(hipMalloc(&x, R * K * 4)); CK(hipMalloc(&W, O * K * 4)); CK(hipMalloc(&b, O * 4));
  CK(hipMemset(src, 0x3c, n * 4));
  std::vector<float> h(R * K + O * K + O);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
  CK(hipMemcpy(x, h.data(), R * K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data() + R * K, O * K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, h.data() + R * K + O * K, O * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(victim, dim3(16), dim3(256), 0, 0, ref, x, W, b, R, O, K);
  CK(hipDeviceSynchronize());
  run<0>("plain loads + ds_write + ds_read_b128", src, n, sink, y, ref, x, W, b, bad);
  run<1>("LDS-DMA + ds_read_b128", src, n, sink, y, ref, x, W, b, bad);
  run<2>("ds_read_b128 + MFMA", src, n, sink, y, ref, x, W, b, bad);
  run<3>("LDS-DMA + ds_read_b128 + MFMA", src, n, sink, y, ref, x, W, b, bad);
  run<4>("LDS-DMA only", src, n, sink, y, ref, x, W, b, bad);
  run<5>("MFMA only", src, n, sink, y, ref, x, W, b, bad);
  run<6>("double-buffered LDS-DMA under ds_read_b128 + MFMA", src, n, sink, y, ref, x, W, b, bad);
  run<7>("double-buffered register staging, same loop", src, n, sink, y, ref, x, W, b, bad);
  run<8>("same loop, no staging: ds_read_b128 + 12 MFMA / barrier", src, n, sink, y, ref, x, W, b, bad);
  run<9>("same loop, registers only: 12 MFMA / barrier", src, n, sink, y, ref, x, W, b, bad);
  run<10>("same loop, staging + ds_read_b128, no MFMA", src, n, sink, y, ref, x, W, b, bad);
  return 0;
}
