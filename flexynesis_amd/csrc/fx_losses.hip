// Loss heads of the flexynesis hot path: value AND gradient in one launch each (wavefront / LDS
// reductions, fixed summation order, no atomics).
//   masked MSE / masked softmax-CE   reference models/direct_pred.py:146-190
//   Cox partial likelihood           reference modules.py:265-305
//   triplet (squared L2, margin)     reference models/triplet_encoder.py:178-194
//   MMD (Gaussian kernel) + recon    reference models/supervised_vae.py:494-550
//   uncertainty-weighted total       reference models/direct_pred.py:192-223
// Every loss kernel writes the RAW loss value to loss_out[0] and the gradient of
//   (w * extra_scale * loss)  with  w = exp(-log_var) (or 1 when log_var == nullptr)
// into its gradient output, so the backward pass needs no second scaling pass.
#include "fx_common.h"
#include "fx_loss_dev.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LOSS_THREADS 256

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LOSS_THREADS) void fx_mse_kernel(float* loss_out, float* __restrict__ dyhat,
                                                              const float* __restrict__ yhat, const float* __restrict__ y,
                                                              int B, long ld, long ldd, const float* logvar,
                                                              float extra_scale) {
  __shared__ float sm[16];
  loss_mse_body(loss_out, dyhat, yhat, y, B, ld, ldd, logvar, extra_scale, sm);
}

__global__ __launch_bounds__(LOSS_THREADS) void fx_ce_kernel(float* loss_out, float* __restrict__ dlogits,
                                                             const float* __restrict__ logits, const float* __restrict__ y,
                                                             int B, int C, long ld, long ldd, const float* logvar,
                                                             float extra_scale) {
  __shared__ float sm[16];
  loss_ce_body(loss_out, dlogits, logits, y, B, C, ld, ldd, logvar, extra_scale, sm);
}

// Cox: one workgroup, B <= 1024 (the device body lives in fx_loss_dev.h)
#define COX_MAX 1024
__global__ __launch_bounds__(LOSS_THREADS) void fx_cox_kernel(float* loss_out, float* __restrict__ dout,
                                                              const float* __restrict__ out, const float* __restrict__ dur,
                                                              const float* __restrict__ ev, int B, long ld, long ldd,
                                                              const float* logvar, float extra_scale) {
  __shared__ float key[COX_MAX];
  __shared__ int idx[COX_MAX];
  __shared__ double scan[COX_MAX];
  __shared__ double dred[16];
  loss_cox_body<COX_MAX, COX_MAX / LOSS_THREADS>(loss_out, dout, out, dur, ev, B, ld, ldd, logvar, extra_scale, key, idx, scan, dred);
}

// ---------------------------------------------------------------------------------------------------
// triplet: one wave per row group; lanes stride the latent dimension.
__global__ __launch_bounds__(LOSS_THREADS) void fx_triplet_kernel(float* loss_out, float* __restrict__ da,
                                                                  float* __restrict__ dp, float* __restrict__ dn,
                                                                  const float* __restrict__ a, const float* __restrict__ p,
                                                                  const float* __restrict__ n, int B, int L, long ld,
                                                                  float margin, const float* logvar, float extra_scale) {
  __shared__ float sm[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float w = loss_weight(logvar) * extra_scale;
  float acc = 0.f;
  for (int b = wid; b < B; b += nw) {
    float sp = 0.f, sn = 0.f;
    for (int d = lane; d < L; d += 64) {
      const float av = a[(long)b * ld + d];
      const float x = av - p[(long)b * ld + d], y = av - n[(long)b * ld + d];
      sp += x * x;
      sn += y * y;
    }
    sp = fx_wave_sum(sp);
    sn = fx_wave_sum(sn);
    const float h = sp - sn + margin;
    const bool on = h > 0.f;
    if (lane == 0 && on) acc += h;
    const float g = on ? w * 2.0f / (float)B : 0.f;
    for (int d = lane; d < L; d += 64) {
      const float av = a[(long)b * ld + d], pv = p[(long)b * ld + d], nv = n[(long)b * ld + d];
      da[(long)b * ld + d] = g * (nv - pv);
      dp[(long)b * ld + d] = -g * (av - pv);
      dn[(long)b * ld + d] = g * (av - nv);
    }
  }
  const float s = fx_block_sum(acc, sm);
  if (threadIdx.x == 0) loss_out[0] = s / (float)B;
}

// ---------------------------------------------------------------------------------------------------
// MMD(prior, z):  k(x,y) = exp(-|x-y|^2 / L^2).  One workgroup per row of the stacked set
// [prior (P rows); z (B rows)]; row_sums[row] = {sum_j k(row, prior_j), sum_j k(row, z_j)}.
// For z rows the gradient wrt z is accumulated into dz (+= so that several modalities' priors add up).
#define MMD_MAXN 2048
// VEC: rows are read as float4 (L % 4 == 0, ldz % 4 == 0, 16-byte aligned bases) with a whole row's loads in flight -- the scalar
// loop is 64 dependent L2 round trips per pair (19 us per launch on an idle chip, 95-120 us beside an HBM-bound product).
// dz_overwrite: dz = term instead of dz += term (no zero-fill launch in front).
template <bool VEC>
__global__ __launch_bounds__(128) void fx_mmd_rows_kernel(float* __restrict__ row_sums, float* __restrict__ dz,
                                                          const float* __restrict__ prior, const float* __restrict__ z,
                                                          int P, int B, int L, long ldz, const float* logvar,
                                                          float extra_scale, int dz_overwrite) {
  __shared__ float kv[MMD_MAXN];
  __shared__ float self[256];
  __shared__ float sm[16];
  const int row = blockIdx.x;
  const bool is_z = row >= P;
  const float* me = is_z ? z + (long)(row - P) * ldz : prior + (long)row * L;
  for (int d = threadIdx.x; d < L; d += blockDim.x) self[d] = me[d];
  __syncthreads();
  const float invL2 = 1.0f / ((float)L * (float)L);
  float sp = 0.f, sz = 0.f;
  const int N = P + B;
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    const float* other = j < P ? prior + (long)j * L : z + (long)(j - P) * ldz;
    float d2 = 0.f;
    if (VEC) {
      for (int d0 = 0; d0 < L; d0 += 32) {           // 8 x 16 bytes in flight; the sum keeps the scalar loop's order (d ascending)
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(other + min(d0 + 4 * u, L - 4));
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (d0 + 4 * u < L)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = self[d0 + 4 * u + e] - v[u][e];
              d2 += t * t;
            }
      }
    } else {
      for (int d = 0; d < L; ++d) {
        const float t = self[d] - other[d];
        d2 += t * t;
      }
    }
    // reference: mean over dims then / dim  (supervised_vae.py:512)
    const float k = expf(-(d2 / (float)L) / (float)L);
    kv[j] = k;
    if (j < P) sp += k; else sz += k;
  }
  sp = fx_block_sum(sp, sm);
  sz = fx_block_sum(sz, sm);
  if (threadIdx.x == 0) { row_sums[2 * row] = sp; row_sums[2 * row + 1] = sz; }
  if (!is_z || dz == nullptr) return;
  __syncthreads();
  // d mmd / d z_a = (2/B^2) sum_b k_ab (-2/L^2)(z_a - z_b)  -  2 * (1/(P B)) sum_j k_ja (-2/L^2)(z_a - p_j)
  const float w = loss_weight(logvar) * extra_scale;
  const float czz = w * (2.0f / ((float)B * (float)B)) * (-2.0f * invL2);
  const float cpz = w * (-2.0f / ((float)P * (float)B)) * (-2.0f * invL2);
  for (int d = threadIdx.x; d < L; d += blockDim.x) {
    float gz = 0.f, gp = 0.f;
    const float sd = self[d];
#pragma unroll 8
    for (int j = 0; j < P; ++j) gp += kv[j] * (sd - prior[(long)j * L + d]);
#pragma unroll 8
    for (int j = 0; j < B; ++j) gz += kv[P + j] * (sd - z[(long)j * ldz + d]);
    float* dst = dz + (long)(row - P) * ldz + d;
    const float g = czz * gz + cpz * gp;
    *dst = dz_overwrite ? g : *dst + g;
  }
}

// recon term of MMD_loss: x_hat = sigmoid(logits); term = mean((x_hat - x)^2) over [B,F];
// writes dlogits = w*extra_scale * 2 (x_hat - x)/(B F) * x_hat (1 - x_hat) IN PLACE of logits when
// dlogits == logits, and per-block partial sums of (x_hat-x)^2.
__global__ __launch_bounds__(256) void fx_recon_sigmoid_kernel(float* __restrict__ partial, float* __restrict__ dlogits,
                                                               float* __restrict__ xhat_out, const float* __restrict__ logits,
                                                               const float* __restrict__ x, long n, const float* logvar,
                                                               float extra_scale) {
  __shared__ float sm[16];
  const float w = loss_weight(logvar) * extra_scale * 2.0f / (float)n;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float xh = 1.0f / (1.0f + expf(-logits[i]));
    const float d = xh - x[i];
    acc += d * d;
    if (xhat_out) xhat_out[i] = xh;
    if (dlogits) dlogits[i] = w * d * xh * (1.0f - xh);
  }
  acc = fx_block_sum(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// mmd_loss accumulation: loss_acc[0] (+)= extra_scale * ( mmd(prior,z) + recon_sum / n_recon )
__global__ __launch_bounds__(256) void fx_mmd_finalize_kernel(float* loss_acc, const float* __restrict__ row_sums, int P,
                                                              int B, const float* __restrict__ recon_partial,
                                                              int n_partial, float n_recon, float extra_scale,
                                                              int accumulate) {
  __shared__ double sm[16];
  double pp = 0.0, pz = 0.0, zz = 0.0, rs = 0.0;
  for (int r = threadIdx.x; r < P + B; r += blockDim.x) {
    if (r < P) { pp += row_sums[2 * r]; pz += row_sums[2 * r + 1]; }
    else zz += row_sums[2 * r + 1];
  }
  for (int i = threadIdx.x; i < n_partial; i += blockDim.x) rs += recon_partial[i];
  pp = fx_block_sum_d(pp, sm);
  pz = fx_block_sum_d(pz, sm);
  zz = fx_block_sum_d(zz, sm);
  rs = fx_block_sum_d(rs, sm);
  if (threadIdx.x == 0) {
    const double mmd = pp / ((double)P * P) + zz / ((double)B * B) - 2.0 * pz / ((double)P * B);
    const double v = (double)extra_scale * (mmd + (n_partial > 0 ? rs / (double)n_recon : 0.0));
    loss_acc[0] = (float)(accumulate ? (double)loss_acc[0] + v : v);
  }
}

// ---------------------------------------------------------------------------------------------------
// total = sum_i exp(-s_i) l_i + s_i   (weighted)   or   sum_i l_i ;  d s_i = 1 - exp(-s_i) l_i
struct TotalArgs {
  int n;
  int weighted;
  const float* loss[16];
  const float* logvar[16];
  float* dlogvar[16];
  float* epoch_acc;  // optional [n+2]: += each raw loss, += total, += 1 (per-epoch means without host syncs)
};
__global__ void fx_total_loss_kernel(float* total_out, TotalArgs t) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float tot = 0.f;
  for (int i = 0; i < t.n; ++i) {
    const float l = t.loss[i][0];
    if (t.weighted) {
      const float s = t.logvar[i][0];
      const float w = expf(-s);
      tot += w * l + s;
      if (t.dlogvar[i]) t.dlogvar[i][0] = 1.0f - w * l;
    } else {
      tot += l;
    }
    if (t.epoch_acc) t.epoch_acc[i] += l;
  }
  if (t.epoch_acc) {
    t.epoch_acc[t.n] += tot;
    t.epoch_acc[t.n + 1] += 1.0f;
  }
  total_out[0] = tot;
}

extern "C" {

int fx_mse_masked(float* loss_out, float* dyhat, const float* yhat, const float* y, int B, long ld, long ldd,
                  const float* logvar, float extra_scale, hipStream_t stream) {
  FX_REQUIRE(loss_out && dyhat && yhat && y && B > 0, "fx_mse_masked: bad args");
  hipLaunchKernelGGL(fx_mse_kernel, dim3(1), dim3(LOSS_THREADS), 0, stream, loss_out, dyhat, yhat, y, B, ld, ldd, logvar,
                     extra_scale);
  return fx_check_launch("fx_mse_masked");
}

int fx_ce_masked(float* loss_out, float* dlogits, const float* logits, const float* y, int B, int C, long ld, long ldd,
                 const float* logvar, float extra_scale, hipStream_t stream) {
  FX_REQUIRE(loss_out && dlogits && logits && y && B > 0 && C > 0, "fx_ce_masked: bad args");
  hipLaunchKernelGGL(fx_ce_kernel, dim3(1), dim3(LOSS_THREADS), 0, stream, loss_out, dlogits, logits, y, B, C, ld, ldd,
                     logvar, extra_scale);
  return fx_check_launch("fx_ce_masked");
}

int fx_cox_ph(float* loss_out, float* dout, const float* out, const float* durations, const float* events, int B, long ld,
              long ldd, const float* logvar, float extra_scale, hipStream_t stream) {
  FX_REQUIRE(loss_out && dout && out && durations && events, "fx_cox_ph: null pointer");
  FX_REQUIRE(B > 0 && B <= COX_MAX, "fx_cox_ph: batch %d outside 1..%d", B, COX_MAX);
  hipLaunchKernelGGL(fx_cox_kernel, dim3(1), dim3(LOSS_THREADS), 0, stream, loss_out, dout, out, durations, events, B, ld,
                     ldd, logvar, extra_scale);
  return fx_check_launch("fx_cox_ph");
}

int fx_triplet(float* loss_out, float* da, float* dp, float* dn, const float* a, const float* p, const float* n, int B,
               int L, long ld, float margin, const float* logvar, float extra_scale, hipStream_t stream) {
  FX_REQUIRE(loss_out && da && dp && dn && a && p && n && B > 0 && L > 0, "fx_triplet: bad args");
  hipLaunchKernelGGL(fx_triplet_kernel, dim3(1), dim3(LOSS_THREADS), 0, stream, loss_out, da, dp, dn, a, p, n, B, L, ld,
                     margin, logvar, extra_scale);
  return fx_check_launch("fx_triplet");
}

long fx_mmd_workspace_floats(int P, int B) { return 2L * (P + B); }

int fx_mmd_rows(float* row_sums, float* dz, const float* prior, const float* z, int P, int B, int L, long ldz,
                const float* logvar, float extra_scale, hipStream_t stream) {
  FX_REQUIRE(row_sums && prior && z && P > 0 && B > 0, "fx_mmd_rows: bad args");
  FX_REQUIRE(P + B <= MMD_MAXN && L <= 256, "fx_mmd_rows: P+B=%d (max %d), L=%d (max 256)", P + B, MMD_MAXN, L);
  hipLaunchKernelGGL(fx_mmd_rows_kernel<false>, dim3(P + B), dim3(128), 0, stream, row_sums, dz, prior, z, P, B, L, ldz, logvar,
                     extra_scale, 0);
  return fx_check_launch("fx_mmd_rows");
}

// fx_mmd_rows with 16-byte row loads where the layout allows (same sums in the same order: bit-identical results) and
// dz_overwrite: dz = term instead of dz += term.
int fx_mmd_rows_ex(float* row_sums, float* dz, const float* prior, const float* z, int P, int B, int L, long ldz,
                   const float* logvar, float extra_scale, int dz_overwrite, hipStream_t stream) {
  FX_REQUIRE(row_sums && prior && z && P > 0 && B > 0, "fx_mmd_rows_ex: bad args");
  FX_REQUIRE(P + B <= MMD_MAXN && L <= 256, "fx_mmd_rows_ex: P+B=%d (max %d), L=%d (max 256)", P + B, MMD_MAXN, L);
  const bool vec = L % 4 == 0 && L >= 4 && ldz % 4 == 0 && ((((uintptr_t)prior) | ((uintptr_t)z)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(fx_mmd_rows_kernel<true>, dim3(P + B), dim3(128), 0, stream, row_sums, dz, prior, z, P, B, L, ldz, logvar,
                       extra_scale, dz_overwrite);
  else
    hipLaunchKernelGGL(fx_mmd_rows_kernel<false>, dim3(P + B), dim3(128), 0, stream, row_sums, dz, prior, z, P, B, L, ldz, logvar,
                       extra_scale, dz_overwrite);
  return fx_check_launch("fx_mmd_rows_ex");
}

int fx_recon_blocks(long n) {
  long b = (n + 256L * 16 - 1) / (256L * 16);
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

int fx_recon_sigmoid(float* partial, float* dlogits, float* xhat_out, const float* logits, const float* x, long n,
                     const float* logvar, float extra_scale, hipStream_t stream) {
  FX_REQUIRE(partial && logits && x && n > 0, "fx_recon_sigmoid: bad args");
  hipLaunchKernelGGL(fx_recon_sigmoid_kernel, dim3(fx_recon_blocks(n)), dim3(256), 0, stream, partial, dlogits, xhat_out,
                     logits, x, n, logvar, extra_scale);
  return fx_check_launch("fx_recon_sigmoid");
}

int fx_mmd_finalize(float* loss_acc, const float* row_sums, int P, int B, const float* recon_partial, int n_partial,
                    float n_recon, float extra_scale, int accumulate, hipStream_t stream) {
  FX_REQUIRE(loss_acc && row_sums, "fx_mmd_finalize: bad args");
  hipLaunchKernelGGL(fx_mmd_finalize_kernel, dim3(1), dim3(256), 0, stream, loss_acc, row_sums, P, B, recon_partial,
                     n_partial, n_recon, extra_scale, accumulate);
  return fx_check_launch("fx_mmd_finalize");
}

int fx_total_loss(float* total_out, int n, int weighted, const float* const* losses, const float* const* logvars,
                  float* const* dlogvars, float* epoch_acc, hipStream_t stream) {
  FX_REQUIRE(total_out && n > 0 && n <= 16 && losses, "fx_total_loss: bad args (n=%d)", n);
  FX_REQUIRE(!weighted || logvars, "fx_total_loss: weighted total needs log_vars");
  TotalArgs t{};
  t.n = n;
  t.weighted = weighted;
  t.epoch_acc = epoch_acc;
  for (int i = 0; i < n; ++i) {
    t.loss[i] = losses[i];
    t.logvar[i] = weighted ? logvars[i] : nullptr;
    t.dlogvar[i] = (weighted && dlogvars) ? dlogvars[i] : nullptr;
  }
  hipLaunchKernelGGL(fx_total_loss_kernel, dim3(1), dim3(64), 0, stream, total_out, t);
  return fx_check_launch("fx_total_loss");
}

}  // extern "C"
