// Shared device/host helpers for libfxhip (gfx950 / CDNA4 only: wave = 64, MFMA f32-in forms).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FX_WAVE 64

// ---- error convention (include/fxhip.h): 0 = ok, <0 = -(hipError_t) or FX_EINVAL -------------
#define FX_EINVAL (-22)
void fx_set_error(const char* fmt, ...);
int fx_check_launch(const char* what);

#define FX_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      fx_set_error(__VA_ARGS__);         \
      return FX_EINVAL;                  \
    }                                    \
  } while (0)

// ---- step control block (device resident, 64 floats) ------------------------------------------
// Written by fx_step_begin / fx_clip_finalize, read by every optimiser kernel, so that a whole
// training step can be replayed from a hipGraph without host-side scalar arguments changing.
enum FxCtrl {
  FXC_STEP = 0,      // Adam step count t, low part: an exact fp32 integer in [0, 2^24); t = FXC_STEP_HI * 2^24 + FXC_STEP
  FXC_LR = 1,        // learning rate
  FXC_BC1 = 2,       // 1 - beta1^t
  FXC_BC2_SQRT = 3,  // sqrt(1 - beta2^t)
  FXC_CLIP_COEF = 4, // min(1, max_norm / (gnorm + 1e-6)); 1 when clipping is off
  FXC_GNORM = 5,     // global grad L2 norm of the step
  FXC_SUMSQ = 6,     // reserved
  FXC_LOSS_TOTAL = 7,
  FXC_BATCH_CURSOR = 8,  // float index of the current batch in the permutation buffer; fx_step_begin advances it by one
                         // row per step and wraps at n_batches
  FXC_RESERVED9 = 9,
  FXC_STEP_HI = 10,      // high part of the step count (number of 2^24 wraps)
  FXC_SIZE = 64
};

#define FX_BETA1 0.9f
#define FX_BETA2 0.999f
#define FX_ADAM_EPS 1e-8f
#define FX_BN_EPS 1e-5f
#define FX_BN_MOMENTUM 0.1f

// ---- wavefront reductions (64 lanes) -------------------------------------------------------------
__device__ __forceinline__ float fx_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double fx_wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float fx_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (<= 16 waves); result valid in every thread.
__device__ __forceinline__ float fx_block_sum(float v, float* smem16) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = fx_wave_sum(v);
  __syncthreads();
  if (lane == 0) smem16[wid] = v;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += smem16[i];
  return r;
}
__device__ __forceinline__ double fx_block_sum_d(double v, double* smem16) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = fx_wave_sum_d(v);
  __syncthreads();
  if (lane == 0) smem16[wid] = v;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  double r = 0.0;
  for (int i = 0; i < nw; ++i) r += smem16[i];
  return r;
}

// ---- Philox4x32-10 counter RNG (production dropout / eps / prior draws) ----------------------------
struct FxPhilox {
  uint32_t c[4];
  uint32_t k[2];
};
__device__ __forceinline__ void fx_philox_round(uint32_t* c, const uint32_t* k) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
// 4 x 32 random bits for (seed, counter).
__device__ __forceinline__ void fx_philox4(uint64_t seed, uint64_t counter, uint32_t out[4]) {
  uint32_t c[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    fx_philox_round(c, k);
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
__device__ __forceinline__ float fx_u01(uint32_t x) {  // (0,1]
  return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}
// uniform in (0,1] for element index i
__device__ __forceinline__ float fx_rand_uniform(uint64_t seed, uint64_t offset, uint64_t i) {
  uint32_t r[4];
  fx_philox4(seed, offset + (i >> 2), r);
  return fx_u01(r[i & 3]);
}
// standard normal for element index i (Box-Muller on a Philox pair)
__device__ __forceinline__ float fx_rand_normal(uint64_t seed, uint64_t offset, uint64_t i) {
  uint32_t r[4];
  fx_philox4(seed, offset + (i >> 1), r);
  const float u1 = fx_u01(r[(i & 1) * 2]), u2 = fx_u01(r[(i & 1) * 2 + 1]);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}
