// Optimiser-side kernels: step bookkeeping, global grad-norm (clip_grad_norm_(1.0), reference
// main.py:216-217), flat-arena Adam (torch.optim.Adam defaults, reference direct_pred.py:143),
// device-resident cohort gather (replaces MultiOmicDataset.__getitem__ + default_collate + the
// per-batch H2D copy, reference data.py:980-995 / main.py:289-298), and the library's error plumbing.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "fx_common.h"

// ---- error plumbing (thread-local so concurrent host threads on different streams stay re-entrant) --
static thread_local char g_err[512] = "";
void fx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int fx_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fx_set_error("%s: %s", what, hipGetErrorString(e));
    return -(int)e;
  }
  return 0;
}

// ---- step control ----------------------------------------------------------------------------------
__global__ void fx_step_begin_kernel(float* ctrl, float lr, int n_batches) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // The step count is kept as two exact fp32 integers, t = FXC_STEP_HI * 2^24 + FXC_STEP, so it keeps counting past
  // 2^24 steps (a single fp32 counter stops incrementing there).  Readers of FXC_STEP alone (the Philox offset uses its
  // low 20 bits) see a value that keeps changing every step.
  float t = ctrl[FXC_STEP] + 1.0f;
  float hi = ctrl[FXC_STEP_HI];
  if (t >= 16777216.0f) {
    t -= 16777216.0f;
    hi += 1.0f;
    ctrl[FXC_STEP_HI] = hi;
  }
  ctrl[FXC_STEP] = t;
  const double te = (double)hi * 16777216.0 + (double)t;
  ctrl[FXC_LR] = lr;
  ctrl[FXC_BC1] = (float)(1.0 - pow((double)FX_BETA1, te));
  ctrl[FXC_BC2_SQRT] = (float)sqrt(1.0 - pow((double)FX_BETA2, te));
  ctrl[FXC_CLIP_COEF] = 1.0f;
  ctrl[FXC_GNORM] = 0.0f;
  // The batch cursor advances by one table row per step and wraps at n_batches (start it at -1 for "row 0 first", at 0
  // when row 0 has already been assembled by a prefetch): exact for any number of steps.
  if (n_batches > 0) {
    int c = (int)ctrl[FXC_BATCH_CURSOR] + 1;
    if (c >= n_batches || c < 0) c = 0;
    ctrl[FXC_BATCH_CURSOR] = (float)c;
  }
}

// Plain streaming copy: one 16-byte non-temporal load + store per thread, one thread per element group, huge grid --
// the pattern that measured fastest on MI355X (scripts/copybench.hip: 6.5 TB/s read+write; grid-stride loops 4.1-5.0,
// per-block contiguous slices 5.3-5.5, hipMemcpyDtoD 5.1).  bench.py reports it as the box's practical HBM rate next to
// the dominant kernel's (boxes of the pool differ by +-10 %).
typedef float fx_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fx_stream_copy_kernel(fx_f32x4* __restrict__ dst, const fx_f32x4* __restrict__ src, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// x *= s[0] with the scalar read on the device: a no-op pass (every block returns after one load) when s[0] == 1, which
// is what loss.backward() hands down -- the drop-in autograd bridge applies the upstream gradient without a host sync
__global__ __launch_bounds__(256) void fx_scale_by_kernel(float* __restrict__ x, long n, const float* __restrict__ s) {
  const float k = s[0];
  if (k == 1.0f) return;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= k;
}

__global__ __launch_bounds__(256) void fx_fill_kernel(float* __restrict__ y, long n, float value) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = value;
}

// ---- sum of squares into double-precision slots ------------------------------------------------------
__global__ __launch_bounds__(256) void fx_sumsq_kernel(double* __restrict__ slots, const float* __restrict__ x, long n) {
  __shared__ double sm[16];
  double acc = 0.0;
  const long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) acc += (double)x[i] * x[i];
  acc = fx_block_sum_d(acc, sm);
  if (threadIdx.x == 0) slots[blockIdx.x] = acc;
}

// slot[0] = sum_ij G1[ij] * G2[ij]   (Gram identity: |dY^T X|_F^2 = <X X^T, dY dY^T>)
__global__ __launch_bounds__(256) void fx_hadamard_sum_kernel(double* __restrict__ slot, const float* __restrict__ g1,
                                                              const float* __restrict__ g2, long n) {
  __shared__ double sm[16];
  double acc = 0.0;
  for (long i = threadIdx.x; i < n; i += blockDim.x) acc += (double)g1[i] * (double)g2[i];
  acc = fx_block_sum_d(acc, sm);
  if (threadIdx.x == 0) slot[0] = acc;
}

__global__ __launch_bounds__(256) void fx_clip_finalize_kernel(float* ctrl, const double* __restrict__ slots, int n_slots,
                                                               float max_norm) {
  __shared__ double sm[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_slots; i += blockDim.x) acc += slots[i];
  acc = fx_block_sum_d(acc, sm);
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(acc);
    ctrl[FXC_GNORM] = total;
    float coef = 1.0f;
    if (max_norm > 0.f) {
      coef = max_norm / (total + 1e-6f);
      coef = coef > 1.0f ? 1.0f : coef;
    }
    ctrl[FXC_CLIP_COEF] = coef;
  }
}

// ---- Adam over a flat arena (all small parameters of a model live in one contiguous buffer) ----------
__global__ __launch_bounds__(256) void fx_adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v, long n,
                                                           const float* __restrict__ ctrl,
                                                           const float* __restrict__ trainable) {
  const FxAdamK ak = fx_adam_consts(ctrl[FXC_LR], ctrl[FXC_BC1], ctrl[FXC_BC2_SQRT], ctrl[FXC_CLIP_COEF]);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (trainable && trainable[i] == 0.f) continue;      // requires_grad=False: not in the optimiser (FineTuner, main.py:562-566)
    float pi = p[i], mi = m[i], vi = v[i];
    fx_adam_update(pi, mi, vi, g[i], ak);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// fx_clip_finalize + fx_adam_flat in one launch: EVERY workgroup adds up the norm slots itself (a few hundred doubles from
// L2, the same fixed order, so all workgroups get the same bits), derives the clip coefficient and applies Adam to its part
// of the arena; workgroup 0 also publishes the norm and the coefficient in the control block for the wide-weight kernels
// that follow.  One dependent launch less between the backward and the dW+Adam kernels.
__global__ __launch_bounds__(256) void fx_adam_flat_clip_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                float* __restrict__ m, float* __restrict__ v, long n,
                                                                float* __restrict__ ctrl, const float* __restrict__ trainable,
                                                                const double* __restrict__ slots, int n_slots, float max_norm) {
  __shared__ double sm[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_slots; i += blockDim.x) acc += slots[i];
  acc = fx_block_sum_d(acc, sm);
  const float total = (float)sqrt(acc);
  float coef = 1.0f;
  if (max_norm > 0.f) {
    coef = max_norm / (total + 1e-6f);
    coef = coef > 1.0f ? 1.0f : coef;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctrl[FXC_GNORM] = total;
    ctrl[FXC_CLIP_COEF] = coef;
  }
  const FxAdamK ak = fx_adam_consts(ctrl[FXC_LR], ctrl[FXC_BC1], ctrl[FXC_BC2_SQRT], coef);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (trainable && trainable[i] == 0.f) continue;
    float pi = p[i], mi = m[i], vi = v[i];
    fx_adam_update(pi, mi, vi, g[i], ak);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// ---- cohort gather: dst[r, :] = src[idx[r], :]  (16-byte coalesced row copies) -------------------------
__global__ __launch_bounds__(256) void fx_gather_rows_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                             const long* __restrict__ idx, int n_rows, int n_cols,
                                                             long ld_src, long ld_dst, const float* __restrict__ ctrl,
                                                             long cursor_stride, int vec) {
  if (ctrl) idx += (long)ctrl[FXC_BATCH_CURSOR] * cursor_stride;
  const int r = blockIdx.y;
  const long s = idx[r];
  const float* sp = src + s * ld_src;
  float* dp = dst + (long)r * ld_dst;
  if (vec) {
    const int n4 = n_cols >> 2;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n4; c += gridDim.x * blockDim.x)
      reinterpret_cast<float4*>(dp)[c] = reinterpret_cast<const float4*>(sp)[c];
  } else {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cols; c += gridDim.x * blockDim.x) dp[c] = sp[c];
  }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" {

const char* fx_last_error_string(void) { return g_err; }
int fx_version(void) { return 300; }  // 0.3.0
// the HIP RUNTIME the process runs on (hipRuntimeGetVersion: major * 10^7 + minor * 10^5 + patch), which on the GPU boxes differs
// from the hipcc that built this library; the host logs it against the versions the hipGraph workarounds were validated on
int fx_hip_runtime_version(void) {
  int v = 0;
  if (hipRuntimeGetVersion(&v) != hipSuccess) return -1;
  return v;
}

// SHA-256 of the sources this library was built from (csrc/build.py defines FX_SOURCE_HASH); build.py finds the literal
// in the binary behind the marker and rebuilds when the sources have changed.
#ifndef FX_SOURCE_HASH
#define FX_SOURCE_HASH "unknown"
#endif
const char* fx_source_hash(void) {
  static const char marked[] = "FXSRCHASH:" FX_SOURCE_HASH;
  return marked + 10;
}

int fx_step_begin(float* ctrl, float lr, int n_batches, hipStream_t stream) {
  FX_REQUIRE(ctrl != nullptr, "fx_step_begin: null ctrl");
  hipLaunchKernelGGL(fx_step_begin_kernel, dim3(1), dim3(64), 0, stream, ctrl, lr, n_batches);
  return fx_check_launch("fx_step_begin");
}

int fx_stream_copy(float* dst, const float* src, long n, hipStream_t stream) {
  FX_REQUIRE(dst && src && n > 0 && n % 4 == 0 && aligned16(dst) && aligned16(src), "fx_stream_copy: n %% 4 == 0 and 16-byte aligned buffers");
  const long n4 = n / 4;
  const long b = (n4 + 255) / 256;
  FX_REQUIRE(b < (1L << 31), "fx_stream_copy: too many elements for one launch");
  hipLaunchKernelGGL(fx_stream_copy_kernel, dim3((unsigned)b), dim3(256), 0, stream, (fx_f32x4*)dst, (const fx_f32x4*)src, n4);
  return fx_check_launch("fx_stream_copy");
}

int fx_scale_by(float* x, long n, const float* scale, hipStream_t stream) {
  FX_REQUIRE(x && scale && n > 0, "fx_scale_by: bad args");
  long b = (n + 256L * 8 - 1) / (256L * 8);
  hipLaunchKernelGGL(fx_scale_by_kernel, dim3((unsigned)(b > 2048 ? 2048 : b)), dim3(256), 0, stream, x, n, scale);
  return fx_check_launch("fx_scale_by");
}

int fx_fill(float* y, long n, float value, hipStream_t stream) {
  FX_REQUIRE(y && n > 0, "fx_fill: bad args");
  long b = (n + 255) / 256;
  hipLaunchKernelGGL(fx_fill_kernel, dim3((unsigned)(b > 1024 ? 1024 : b)), dim3(256), 0, stream, y, n, value);
  return fx_check_launch("fx_fill");
}

int fx_sumsq_blocks(long n) {
  long b = (n / 4 + 256L * 8 - 1) / (256L * 8);
  return (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
}

// writes fx_sumsq_blocks(n) partial sums into slots[0..)
int fx_sumsq(double* slots, const float* x, long n, hipStream_t stream) {
  FX_REQUIRE(slots && x && n > 0, "fx_sumsq: bad args");
  FX_REQUIRE(aligned16(x), "fx_sumsq: x must be 16-byte aligned");
  hipLaunchKernelGGL(fx_sumsq_kernel, dim3(fx_sumsq_blocks(n)), dim3(256), 0, stream, slots, x, n);
  return fx_check_launch("fx_sumsq");
}

int fx_hadamard_sum(double* slot, const float* g1, const float* g2, long n, hipStream_t stream) {
  FX_REQUIRE(slot && g1 && g2 && n > 0, "fx_hadamard_sum: bad args");
  hipLaunchKernelGGL(fx_hadamard_sum_kernel, dim3(1), dim3(256), 0, stream, slot, g1, g2, n);
  return fx_check_launch("fx_hadamard_sum");
}

int fx_clip_finalize(float* ctrl, const double* slots, int n_slots, float max_norm, hipStream_t stream) {
  FX_REQUIRE(ctrl && slots && n_slots > 0, "fx_clip_finalize: bad args");
  hipLaunchKernelGGL(fx_clip_finalize_kernel, dim3(1), dim3(256), 0, stream, ctrl, slots, n_slots, max_norm);
  return fx_check_launch("fx_clip_finalize");
}

int fx_adam_flat(float* p, const float* g, float* m, float* v, long n, const float* ctrl, const float* trainable,
                 hipStream_t stream) {
  FX_REQUIRE(p && g && m && v && ctrl && n > 0, "fx_adam_flat: bad args");
  long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  hipLaunchKernelGGL(fx_adam_flat_kernel, dim3((unsigned)b), dim3(256), 0, stream, p, g, m, v, n, ctrl, trainable);
  return fx_check_launch("fx_adam_flat");
}

int fx_adam_flat_clip(float* p, const float* g, float* m, float* v, long n, float* ctrl, const float* trainable,
                      const double* slots, int n_slots, float max_norm, hipStream_t stream) {
  FX_REQUIRE(p && g && m && v && ctrl && slots && n > 0 && n_slots > 0, "fx_adam_flat_clip: bad args");
  long b = (n + 255) / 256;
  if (b > 1024) b = 1024;
  hipLaunchKernelGGL(fx_adam_flat_clip_kernel, dim3((unsigned)b), dim3(256), 0, stream, p, g, m, v, n, ctrl, trainable, slots,
                     n_slots, max_norm);
  return fx_check_launch("fx_adam_flat_clip");
}

int fx_gather_rows(float* dst, const float* src, const long* idx, int n_rows, int n_cols, long ld_src, long ld_dst,
                   const float* ctrl_cursor, long cursor_stride, hipStream_t stream) {
  FX_REQUIRE(dst && src && idx && n_rows > 0 && n_cols > 0, "fx_gather_rows: bad args");
  const int vec = aligned16(dst) && aligned16(src) && (ld_src % 4 == 0) && (ld_dst % 4 == 0) && (n_cols % 4 == 0);
  int bx = ((vec ? n_cols / 4 : n_cols) + 255) / 256;
  if (bx > 64) bx = 64;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(fx_gather_rows_kernel, dim3(bx, n_rows), dim3(256), 0, stream, dst, src, idx, n_rows, n_cols, ld_src,
                     ld_dst, ctrl_cursor, cursor_stride, vec);
  return fx_check_launch("fx_gather_rows");
}

}  // extern "C"
