// BatchNorm1d + activation + dropout, forward and backward, for the two block orders of the
// reference (modules.py):
//   MLP      : Linear -> BatchNorm1d -> ReLU -> Dropout(0.1)        (pre_act = NONE,  post = RELU+dropout)
//   Enc/Dec  : Linear -> LeakyReLU(0.2) -> BatchNorm1d              (pre_act = LEAKY, post = NONE)
// Input x is [B, C] row-major (samples = rows).  One workgroup owns 32 columns x all B rows:
// 256 threads = 32 column lanes x 8 row groups, so every global access is a 128-byte row segment
// and the per-column batch statistics are wavefront/LDS reductions (no atomics, fixed order).
// Statistics are two-pass (mean, then sum (x-mean)^2), like the reference's CPU batch_norm.
#include "fx_common.h"
#include "fx_small.h"

enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_RELU = 2 };
#define LEAKY_SLOPE 0.2f
#define COLS 32
#define RGRP 8

__device__ __forceinline__ float pre_activate(float x, int pre_act) {
  return (pre_act == ACT_LEAKY) ? (x > 0.f ? x : x * LEAKY_SLOPE) : x;
}

// Column reduction helper: each thread holds a partial for column cx from its row group ry;
// returns the total over the 8 row groups (valid in all threads of that column).
__device__ __forceinline__ float col_reduce(float v, float (*red)[COLS], int cx, int ry) {
  __syncthreads();
  red[ry][cx] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < RGRP; ++i) s += red[i][cx];
  return s;
}

struct BnFwdArgs {
  float* out;          // [B,C] post-activation/dropout output
  const float* x;      // [B,C] pre-BN input (Linear output)
  const float* gamma;  // [C]
  const float* beta;   // [C]
  float* running_mean; // [C] (updated in train mode)
  float* running_var;  // [C]
  float* save_mean;    // [C] (train) for backward
  float* save_invstd;  // [C]
  const float* mask;   // [B,C] supplied 0/1 dropout mask (parity mode) or null
  float* mask_out;     // optional: where to record the Philox mask (or null)
  int B, C;
  long ldx, ldo;
  int pre_act, post_act;
  int train;
  float drop_p;        // 0 = no dropout
  unsigned long long seed, offset;
  const float* ctrl;   // optional step-control block: the Adam step count is folded into the RNG offset
};

// Philox counter layout: [63:44] training step | [43:0] caller offset + element/4 -- so a captured
// hipGraph replays with fresh randomness every step without any host-side argument changing.
__device__ __forceinline__ unsigned long long step_offset(const float* ctrl, unsigned long long offset) {
  return ctrl ? offset + (((unsigned long long)ctrl[FXC_STEP]) << 44) : offset;
}

__global__ __launch_bounds__(256) void fx_bn_fwd_kernel(BnFwdArgs a) {
  __shared__ float red[RGRP][COLS];
  const int cx = threadIdx.x & (COLS - 1), ry = threadIdx.x / COLS;
  const int c = blockIdx.x * COLS + cx;
  const bool cok = c < a.C;
  float mean, invstd;
  if (a.train) {
    float s = 0.f;
    if (cok)
      for (int r = ry; r < a.B; r += RGRP) s += pre_activate(a.x[(long)r * a.ldx + c], a.pre_act);
    mean = col_reduce(s, red, cx, ry) / (float)a.B;
    float q = 0.f;
    if (cok)
      for (int r = ry; r < a.B; r += RGRP) {
        const float d = pre_activate(a.x[(long)r * a.ldx + c], a.pre_act) - mean;
        q += d * d;
      }
    const float var_b = col_reduce(q, red, cx, ry) / (float)a.B;
    invstd = 1.0f / sqrtf(var_b + FX_BN_EPS);
    if (cok && ry == 0) {
      a.save_mean[c] = mean;
      a.save_invstd[c] = invstd;
      const float var_u = a.B > 1 ? var_b * ((float)a.B / (float)(a.B - 1)) : var_b;
      a.running_mean[c] = (1.0f - FX_BN_MOMENTUM) * a.running_mean[c] + FX_BN_MOMENTUM * mean;
      a.running_var[c] = (1.0f - FX_BN_MOMENTUM) * a.running_var[c] + FX_BN_MOMENTUM * var_u;
    }
  } else {
    mean = cok ? a.running_mean[c] : 0.f;
    invstd = cok ? 1.0f / sqrtf(a.running_var[c] + FX_BN_EPS) : 0.f;
  }
  if (!cok) return;
  const float gm = a.gamma[c], bt = a.beta[c];
  const bool drop = a.train && a.drop_p > 0.f;
  const float keep_scale = 1.0f / (1.0f - a.drop_p);
  const unsigned long long rng_off = step_offset(a.ctrl, a.offset);
  for (int r = ry; r < a.B; r += RGRP) {
    const float xv = pre_activate(a.x[(long)r * a.ldx + c], a.pre_act);
    float y = (xv - mean) * invstd * gm + bt;
    if (a.post_act == ACT_RELU) y = fmaxf(y, 0.f);
    if (drop) {
      float mk;
      if (a.mask) {
        mk = a.mask[(long)r * a.C + c];
      } else {
        mk = fx_rand_uniform(a.seed, rng_off, (unsigned long long)r * a.C + c) <= (1.0f - a.drop_p) ? 1.f : 0.f;
        if (a.mask_out) a.mask_out[(long)r * a.C + c] = mk;
      }
      y = y * (mk * keep_scale);  // mask scaled first, then multiplied (== F.dropout on CPU)
    }
    a.out[(long)r * a.ldo + c] = y;
  }
}

struct BnBwdArgs {
  float* dx;            // [B,C] grad wrt the Linear output (pre-BN, pre-pre_act)
  float* dgamma;        // [C]
  float* dbeta;         // [C]
  float* dbias;         // [C] = column sums of dx (grad of the Linear bias) or null
  const float* dout;    // [B,C] grad wrt the block output
  const float* x;       // [B,C] saved Linear output
  const float* out;     // [B,C] saved block output (needed for ReLU/dropout gating) or null
  const float* gamma;
  const float* save_mean;
  const float* save_invstd;
  int B, C;
  long ldx, ldo, lddo, lddx;
  int pre_act, post_act;
  float drop_p;         // dropout prob that was applied in forward (0 if none / eval)
  int accumulate;       // dgamma/dbeta/dbias += (second and later passes through a shared block: triplet)
};

__global__ __launch_bounds__(256) void fx_bn_bwd_kernel(BnBwdArgs a) {
  __shared__ float red[RGRP][COLS];
  const int cx = threadIdx.x & (COLS - 1), ry = threadIdx.x / COLS;
  const int c = blockIdx.x * COLS + cx;
  const bool cok = c < a.C;
  const float mean = cok ? a.save_mean[c] : 0.f, invstd = cok ? a.save_invstd[c] : 0.f;
  const float gm = cok ? a.gamma[c] : 0.f;
  const float gate_scale = 1.0f / (1.0f - a.drop_p);
  // dy = grad wrt the BN output: out = dropout(relu(bn)) -> dy = dout * (out > 0 ? 1/(1-p) : 0)
  auto dy_at = [&](int r) -> float {
    float d = a.dout[(long)r * a.lddo + c];
    if (a.post_act == ACT_RELU) d = (a.out[(long)r * a.ldo + c] > 0.f) ? d * gate_scale : 0.f;
    return d;
  };
  float s1 = 0.f, s2 = 0.f;
  if (cok)
    for (int r = ry; r < a.B; r += RGRP) {
      const float xh = (pre_activate(a.x[(long)r * a.ldx + c], a.pre_act) - mean) * invstd;
      const float dy = dy_at(r);
      s1 += dy;
      s2 += dy * xh;
    }
  const float sum_dy = col_reduce(s1, red, cx, ry);
  const float sum_dy_xh = col_reduce(s2, red, cx, ry);
  float sb = 0.f;
  if (cok) {
    const float invB = 1.0f / (float)a.B;
    for (int r = ry; r < a.B; r += RGRP) {
      const float xraw = a.x[(long)r * a.ldx + c];
      const float xh = (pre_activate(xraw, a.pre_act) - mean) * invstd;
      const float dy = dy_at(r);
      // dx_bn = gamma*invstd/B * (B*dy - sum(dy) - xhat*sum(dy*xhat))
      float d = gm * invstd * (dy - invB * sum_dy - xh * invB * sum_dy_xh);
      if (a.pre_act == ACT_LEAKY) d = xraw > 0.f ? d : d * LEAKY_SLOPE;
      a.dx[(long)r * a.lddx + c] = d;
      sb += d;
    }
  }
  const float sum_dx = col_reduce(sb, red, cx, ry);
  if (cok && ry == 0) {
    if (a.accumulate) {
      a.dgamma[c] += sum_dy_xh;
      a.dbeta[c] += sum_dy;
      if (a.dbias) a.dbias[c] += sum_dx;
    } else {
      a.dgamma[c] = sum_dy_xh;
      a.dbeta[c] = sum_dy;
      if (a.dbias) a.dbias[c] = sum_dx;
    }
  }
}

// out[c] = sum_r x[r,c]  (Linear bias gradient where no BN follows)
__global__ __launch_bounds__(256) void fx_colsum_kernel(float* __restrict__ out, const float* __restrict__ x, int B, int C,
                                                        long ldx) {
  __shared__ float red[RGRP][COLS];
  const int cx = threadIdx.x & (COLS - 1), ry = threadIdx.x / COLS;
  const int c = blockIdx.x * COLS + cx;
  float s = 0.f;
  if (c < C) {
    int r = ry;
    for (; r + 3 * RGRP < B; r += 4 * RGRP) {      // four rows in flight per thread (same summation order)
      const float v0 = x[(long)r * ldx + c], v1 = x[(long)(r + RGRP) * ldx + c];
      const float v2 = x[(long)(r + 2 * RGRP) * ldx + c], v3 = x[(long)(r + 3 * RGRP) * ldx + c];
      s += v0; s += v1; s += v2; s += v3;
    }
    for (; r < B; r += RGRP) s += x[(long)r * ldx + c];
  }
  s = col_reduce(s, red, cx, ry);
  if (c < C && ry == 0) out[c] = s;
}

// elementwise helpers ---------------------------------------------------------------------------------
// sigmoid forward in place + optional fused reconstruction term:  y = sigmoid(x)
__global__ void fx_sigmoid_kernel(float* __restrict__ y, const float* __restrict__ x, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = 1.0f / (1.0f + expf(-x[i]));
}
// dx = dy * y * (1 - y)   (backward of y = sigmoid(x) from its saved output)
__global__ void fx_sigmoid_bwd_kernel(float* __restrict__ dx, const float* __restrict__ dy, const float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = dy[i] * y[i] * (1.0f - y[i]);
}
// row-wise softmax of a small [B, C] logit matrix (predict(): class probabilities of a categorical head), one thread per row
__global__ void fx_softmax_rows_kernel(float* __restrict__ y, const float* __restrict__ x, int B, int C, long ldx, long ldy) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B) return;
  const float* xr = x + (long)r * ldx;
  float mx = xr[0];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, xr[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += expf(xr[c] - mx);
  for (int c = 0; c < C; ++c) y[(long)r * ldy + c] = expf(xr[c] - mx) / s;
}
// z = mean + log_var * eps (reference supervised_vae.py:198-199: log_var is used directly as the scale)
__global__ void fx_reparam_kernel(float* __restrict__ z, float* __restrict__ eps_out, const float* __restrict__ mean,
                                  const float* __restrict__ log_var, const float* __restrict__ eps, long n,
                                  unsigned long long seed, unsigned long long offset, const float* ctrl) {
  offset = step_offset(ctrl, offset);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float e = eps ? eps[i] : fx_rand_normal(seed, offset, (unsigned long long)i);
    if (eps_out) eps_out[i] = e;
    z[i] = mean[i] + log_var[i] * e;
  }
}
// y[i] = a[i] * b[i]   /  y[i] += a[i]
__global__ void fx_mul_kernel(float* __restrict__ y, const float* __restrict__ a, const float* __restrict__ b, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = a[i] * b[i];
}
__global__ void fx_fill_normal_kernel(float* __restrict__ y, long n, unsigned long long seed, unsigned long long offset,
                                      const float* ctrl) {
  offset = step_offset(ctrl, offset);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = fx_rand_normal(seed, offset, (unsigned long long)i);
}

static inline int grid_for(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" {

int fx_bn_act_fwd(float* out, const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                  float* save_mean, float* save_invstd, const float* mask, float* mask_out, int B, int C, long ldx, long ldo,
                  int pre_act, int post_act, int train, float drop_p, unsigned long long seed, unsigned long long offset,
                  const float* ctrl, hipStream_t stream) {
  FX_REQUIRE(out && x && gamma && beta && running_mean && running_var, "fx_bn_act_fwd: null pointer");
  FX_REQUIRE(B > 0 && C > 0, "fx_bn_act_fwd: bad shape B=%d C=%d", B, C);
  FX_REQUIRE(!train || (save_mean && save_invstd), "fx_bn_act_fwd: train mode needs save_mean/save_invstd");
  FX_REQUIRE(!(train && B < 2), "fx_bn_act_fwd: BatchNorm1d training needs more than 1 value per channel");
  FX_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "fx_bn_act_fwd: drop_p %f", (double)drop_p);
  if (B <= FX_BN_R16_MAX_B && mask_out == nullptr) {   // whole column in registers: one pass over x
    BnFwd16 r{out, nullptr, x, nullptr, 0, 0, nullptr, gamma, beta, running_mean, running_var, save_mean, save_invstd, mask,
              B, C, ldx, ldo, pre_act, post_act, train, drop_p, seed, offset, ctrl};
    return fx_launch_bn_fwd_r16(r, stream);
  }
  BnFwdArgs a{out, x, gamma, beta, running_mean, running_var, save_mean, save_invstd, mask, mask_out, B, C, ldx, ldo,
              pre_act, post_act, train, drop_p, seed, offset, ctrl};
  hipLaunchKernelGGL(fx_bn_fwd_kernel, dim3((C + COLS - 1) / COLS), dim3(256), 0, stream, a);
  return fx_check_launch("fx_bn_act_fwd");
}

int fx_bn_act_bwd(float* dx, float* dgamma, float* dbeta, float* dbias, const float* dout, const float* x, const float* out,
                  const float* gamma, const float* save_mean, const float* save_invstd, int B, int C, long ldx, long ldo,
                  long lddo, long lddx, int pre_act, int post_act, float drop_p, int accumulate, hipStream_t stream) {
  FX_REQUIRE(dx && dgamma && dbeta && dout && x && gamma && save_mean && save_invstd, "fx_bn_act_bwd: null pointer");
  FX_REQUIRE(post_act != ACT_RELU || out, "fx_bn_act_bwd: ReLU gating needs the saved block output");
  if (B <= FX_BN_R16_MAX_B) {
    BnBwd16 r{dx, dgamma, dbeta, dbias, dout, x, out, gamma, save_mean, save_invstd, B, C, ldx, ldo, lddo, lddx,
              pre_act, post_act, drop_p, accumulate};
    return fx_launch_bn_bwd_r16(r, stream);
  }
  BnBwdArgs a{dx, dgamma, dbeta, dbias, dout, x, out, gamma, save_mean, save_invstd, B, C, ldx, ldo, lddo, lddx,
              pre_act, post_act, drop_p, accumulate};
  hipLaunchKernelGGL(fx_bn_bwd_kernel, dim3((C + COLS - 1) / COLS), dim3(256), 0, stream, a);
  return fx_check_launch("fx_bn_act_bwd");
}

// EVAL-mode backward of the same blocks, with respect to the block input only (attribution: d output / d input through
// model.eval(); reference direct_pred.py:418-431 under Captum).  BatchNorm is the affine map of its running statistics:
//   MLP      out = relu(bn(x))           dx = dout * [out > 0] * gamma / sqrt(running_var + eps)
//   Enc/Dec  out = bn(leaky(x))          dx = dout * gamma / sqrt(running_var + eps) * (x > 0 ? 1 : 0.2)
__global__ __launch_bounds__(256) void fx_bn_eval_bwd_kernel(float* __restrict__ dx, const float* __restrict__ dout,
                                                             const float* __restrict__ x, const float* __restrict__ out,
                                                             const float* __restrict__ gamma, const float* __restrict__ rvar,
                                                             int B, int C, long ldx, long ldo, long lddo, long lddx,
                                                             int pre_act, int post_act) {
  const int c = blockIdx.x * COLS + (threadIdx.x & (COLS - 1)), ry = threadIdx.x / COLS;
  if (c >= C) return;
  const float k = gamma[c] / sqrtf(rvar[c] + FX_BN_EPS);
  for (int r = blockIdx.y * RGRP + ry; r < B; r += gridDim.y * RGRP) {
    float g = dout[(long)r * lddo + c] * k;
    if (post_act == ACT_RELU) g = out[(long)r * ldo + c] > 0.f ? g : 0.f;
    if (pre_act == ACT_LEAKY) g = x[(long)r * ldx + c] > 0.f ? g : g * LEAKY_SLOPE;
    dx[(long)r * lddx + c] = g;
  }
}

int fx_bn_eval_bwd(float* dx, const float* dout, const float* x, const float* out, const float* gamma, const float* running_var,
                   int B, int C, long ldx, long ldo, long lddo, long lddx, int pre_act, int post_act, hipStream_t stream) {
  FX_REQUIRE(dx && dout && gamma && running_var && B > 0 && C > 0, "fx_bn_eval_bwd: bad args");
  FX_REQUIRE(post_act != ACT_RELU || out, "fx_bn_eval_bwd: ReLU gating needs the block output");
  FX_REQUIRE(pre_act != ACT_LEAKY || x, "fx_bn_eval_bwd: the LeakyReLU slope needs the block input");
  int gy = (B + RGRP - 1) / RGRP;
  if (gy > 16) gy = 16;
  hipLaunchKernelGGL(fx_bn_eval_bwd_kernel, dim3((C + COLS - 1) / COLS, gy), dim3(256), 0, stream, dx, dout, x, out, gamma,
                     running_var, B, C, ldx, ldo, lddo, lddx, pre_act, post_act);
  return fx_check_launch("fx_bn_eval_bwd");
}

int fx_colsum(float* out, const float* x, int B, int C, long ldx, hipStream_t stream) {
  FX_REQUIRE(out && x && B > 0 && C > 0, "fx_colsum: bad args");
  hipLaunchKernelGGL(fx_colsum_kernel, dim3((C + COLS - 1) / COLS), dim3(256), 0, stream, out, x, B, C, ldx);
  return fx_check_launch("fx_colsum");
}

int fx_sigmoid(float* y, const float* x, long n, hipStream_t stream) {
  FX_REQUIRE(y && x && n > 0, "fx_sigmoid: bad args");
  hipLaunchKernelGGL(fx_sigmoid_kernel, dim3(grid_for(n)), dim3(256), 0, stream, y, x, n);
  return fx_check_launch("fx_sigmoid");
}

int fx_sigmoid_bwd(float* dx, const float* dy, const float* y, long n, hipStream_t stream) {
  FX_REQUIRE(dx && dy && y && n > 0, "fx_sigmoid_bwd: bad args");
  hipLaunchKernelGGL(fx_sigmoid_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dx, dy, y, n);
  return fx_check_launch("fx_sigmoid_bwd");
}

int fx_softmax_rows(float* y, const float* x, int B, int C, long ldx, long ldy, hipStream_t stream) {
  FX_REQUIRE(y && x && B > 0 && C > 0 && ldx >= C && ldy >= C, "fx_softmax_rows: bad args");
  hipLaunchKernelGGL(fx_softmax_rows_kernel, dim3((B + 127) / 128), dim3(128), 0, stream, y, x, B, C, ldx, ldy);
  return fx_check_launch("fx_softmax_rows");
}

int fx_reparam(float* z, float* eps_out, const float* mean, const float* log_var, const float* eps, long n,
               unsigned long long seed, unsigned long long offset, const float* ctrl, hipStream_t stream) {
  FX_REQUIRE(z && mean && log_var && n > 0, "fx_reparam: bad args");
  hipLaunchKernelGGL(fx_reparam_kernel, dim3(grid_for(n)), dim3(256), 0, stream, z, eps_out, mean, log_var, eps, n, seed,
                     offset, ctrl);
  return fx_check_launch("fx_reparam");
}

int fx_mul(float* y, const float* a, const float* b, long n, hipStream_t stream) {
  FX_REQUIRE(y && a && b && n > 0, "fx_mul: bad args");
  hipLaunchKernelGGL(fx_mul_kernel, dim3(grid_for(n)), dim3(256), 0, stream, y, a, b, n);
  return fx_check_launch("fx_mul");
}

int fx_fill_normal(float* y, long n, unsigned long long seed, unsigned long long offset, const float* ctrl,
                   hipStream_t stream) {
  FX_REQUIRE(y && n > 0, "fx_fill_normal: bad args");
  hipLaunchKernelGGL(fx_fill_normal_kernel, dim3(grid_for(n)), dim3(256), 0, stream, y, n, seed, offset, ctrl);
  return fx_check_launch("fx_fill_normal");
}

}  // extern "C"
