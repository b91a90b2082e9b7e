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

// ---- Adam update (torch.optim.Adam defaults: lerp / addcmul / sqrt / div / add eps / addcdiv) --------------------------------
// One definition for every optimiser kernel (fx_adam_flat*, the dW + Adam epilogues, the fused dW + Adam + next-forward kernel),
// so that they stay bit-identical to each other.
//
// The wide kernels run at the package's POWER limit (1.36-1.40 kW while fx_linear_dw_adam_fwd_bf16x3 runs back to back, DESIGN.md
// section 3.9): their time is their energy, and every VALU instruction per element of a 100 M-element weight is worth ~1.4 us of a
// 450 us launch.  The textbook form with IEEE sqrt and two IEEE divisions is 54 instructions per element (32 of them the
// compiler's div_scale / div_fmas / div_fixup / refinement sequences); this one is 10:
//     m' = fma(g, (1 - b1) c, b1 m)          v' = fma(t, t, b2 v), t = g sqrt(1 - b2) c          (c = clip coefficient; t stays finite
//                                            for every finite clipped gradient: g g overflowed beyond |g| = 1.8e19 whatever c was)
//     d  = fma(v_sqrt_f32(v'), 1 / sqrt(1 - b2^t), eps)                  p' = fma(-lr / (1 - b1^t), m' v_rcp_f32(d), p)
// v_sqrt_f32 and v_rcp_f32 are accurate to 1 ulp, so the update differs from torch's by <= ~4 ulp (5e-7 relative) of a step of
// size ~lr -- two orders of magnitude below the 3e-5 relative error the split-bf16 products leave in the gradient itself, and far
// inside every parity tolerance (losses 1e-4).  m', v' differ from torch's lerp / addcmul sequence by 1 ulp.  Special values fall
// out right without fix-up code: v' = 0 gives d = eps; v' = inf (overflowed gradients) gives rcp(inf) = 0, a zero step like m / inf;
// NaN gradients propagate.  -DFX_ADAM_EXACT (scripts/build_variant.py) keeps the correctly rounded form of the first half of round 4
// (hardware approximations + one Markstein residual correction each, exact unless the true value lies within ~1e-7 ulp of a rounding
// boundary; 38 instructions) for A/B measurements.
struct FxAdamK {
  float c1, c2, nstep, rbc2s;
#ifdef FX_ADAM_EXACT
  float coef, step_size, bc2s;
#endif
};
__device__ __forceinline__ FxAdamK fx_adam_consts(float lr, float bc1, float bc2s, float coef) {
  FxAdamK k;
  k.c1 = (1.0f - FX_BETA1) * coef;
  k.c2 = 0.0316227766016837933f * coef;      // sqrt(1 - beta2) * coef: v' = fma(t, t, b2 v) with t = g * c2 stays finite for any finite clipped gradient (g * g did not)
  k.nstep = -(lr / bc1);
  k.rbc2s = 1.0f / bc2s;          // IEEE, once per thread and launch
#ifdef FX_ADAM_EXACT
  k.coef = coef; k.step_size = lr / bc1; k.bc2s = bc2s;
#endif
  return k;
}
#ifdef FX_ADAM_EXACT
__device__ __forceinline__ float fx_sqrt_rn(float x) {
  const float s0 = __builtin_amdgcn_sqrtf(x);
  const float r = __builtin_amdgcn_rcpf(s0);
  const float e = __builtin_fmaf(-s0, s0, x);
  const float s1 = __builtin_fmaf(e, 0.5f * r, s0);
  return __builtin_amdgcn_class(s0, 0x267) ? s0 : s1;      // 0, inf, NaN: as the hardware returned them
}
__device__ __forceinline__ float fx_div_rn(float a, float b) {       // b finite, normal, > 0
  const float r = __builtin_amdgcn_rcpf(b);
  const float q0 = a * r;
  const float e = __builtin_fmaf(-q0, b, a);
  return __builtin_fmaf(e, r, q0);
}
#endif
__device__ __forceinline__ void fx_adam_update(float& p, float& m, float& v, float g, const FxAdamK& k) {
#ifdef FX_ADAM_EXACT
  const float gr = g * k.coef;
  const float m2 = m + (gr - m) * (1.0f - FX_BETA1);
  const float v2 = v * FX_BETA2 + (1.0f - FX_BETA2) * gr * gr;
#ifdef FX_ADAM_IEEE       /* rounds 1-3: the compiler's IEEE sqrt / division expansions (54 instructions per element) */
  p = p - k.step_size * (m2 / (sqrtf(v2) / k.bc2s + FX_ADAM_EPS));
  m = m2;
  v = v2;
  return;
#endif
  const float s = fx_sqrt_rn(v2);
  const float q0 = s * k.rbc2s;
  const float d0 = __builtin_fmaf(__builtin_fmaf(-q0, k.bc2s, s), k.rbc2s, q0) + FX_ADAM_EPS;
  const float d = fminf(d0, 3.0e38f);
  p = p - k.step_size * fx_div_rn(m2, d);
#else
  const float m2 = __builtin_fmaf(g, k.c1, m * FX_BETA1);
  const float t = g * k.c2;
  const float v2 = __builtin_fmaf(t, t, v * FX_BETA2);
  const float d = __builtin_fmaf(__builtin_amdgcn_sqrtf(v2), k.rbc2s, FX_ADAM_EPS);
  p = __builtin_fmaf(k.nstep, m2 * __builtin_amdgcn_rcpf(d), p);
#endif
  m = m2;
  v = v2;
}

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
