// Device bodies of the per-head loss functions (value AND output gradient), shared by the stand-alone loss kernels
// (fx_losses.hip) and the fused heads kernel (fx_heads.hip: forward -> loss -> backward of a supervisor head in one launch).
//   masked MSE / masked softmax-CE   reference models/direct_pred.py:146-190
//   Cox partial likelihood           reference modules.py:265-305
// Every body runs in ONE workgroup (any blockDim.x that is a multiple of 64, <= 1024), writes the RAW loss value to
// loss_out[0] and the gradient of (w * extra_scale * loss), w = exp(-log_var) (1 when log_var == nullptr).
#pragma once
#include "fx_common.h"

__device__ __forceinline__ float loss_weight(const float* logvar) { return logvar ? expf(-logvar[0]) : 1.0f; }

__device__ __forceinline__ void loss_mse_body(float* loss_out, float* __restrict__ dyhat, const float* __restrict__ yhat,
                                              const float* __restrict__ y, int B, long ld, long ldd, const float* logvar,
                                              float extra_scale, float* sm /* [16] */) {
  float cnt = 0.f, sq = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float t = y[b];
    if (!isnan(t)) {
      const float d = yhat[(long)b * ld] - t;
      cnt += 1.f;
      sq += d * d;
    }
  }
  const float n = fx_block_sum(cnt, sm);
  const float s = fx_block_sum(sq, sm);
  const float w = loss_weight(logvar) * extra_scale;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float t = y[b];
    dyhat[(long)b * ldd] = (!isnan(t) && n > 0.f) ? w * 2.0f * (yhat[(long)b * ld] - t) / n : 0.f;
  }
  if (threadIdx.x == 0) loss_out[0] = n > 0.f ? s / n : 0.f;
}

__device__ __forceinline__ void loss_ce_body(float* loss_out, float* __restrict__ dlogits, const float* __restrict__ logits,
                                             const float* __restrict__ y, int B, int C, long ld, long ldd, const float* logvar,
                                             float extra_scale, float* sm /* [16] */) {
  float cnt = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float t = y[b];
    const bool valid = !isnan(t) && t != -1.0f && (long)t >= 0 && (long)t < C;
    cnt += valid ? 1.f : 0.f;
  }
  const float n = fx_block_sum(cnt, sm);
  const float w = loss_weight(logvar) * extra_scale;
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float t = y[b];
    const bool valid = !isnan(t) && t != -1.0f && (long)t >= 0 && (long)t < C;
    const float* row = logits + (long)b * ld;
    float* drow = dlogits + (long)b * ldd;
    if (!valid || n <= 0.f) {
      for (int c = 0; c < C; ++c) drow[c] = 0.f;
      continue;
    }
    float mx = row[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(row[c] - mx);
    const float lse = mx + logf(se);
    const int tgt = (int)(long)t;
    acc += lse - row[tgt];
    const float g = w / n;
    for (int c = 0; c < C; ++c) drow[c] = g * (expf(row[c] - lse) - (c == tgt ? 1.f : 0.f));
  }
  const float s = fx_block_sum(acc, sm);
  if (threadIdx.x == 0) loss_out[0] = n > 0.f ? s / n : 0.f;
}

// Cox: B <= MAXN.  Bitonic sort by duration (descending) in LDS, double-precision inclusive scan of exp(o) (risk sets),
// reverse scan for the gradient.  key / idx / scan: [MAXN] LDS arrays, dred: [16] doubles; PER = ceil(MAXN / blockDim.x).
template <int MAXN, int PER>
__device__ __forceinline__ void loss_cox_body(float* loss_out, float* __restrict__ dout, const float* __restrict__ out,
                                              const float* __restrict__ dur, const float* __restrict__ ev, int B, long ld,
                                              long ldd, const float* logvar, float extra_scale, float* key, int* idx,
                                              double* scan, double* dred) {
  int P = 1;
  while (P < B) P <<= 1;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    bool valid = false;
    if (i < B) valid = !isnan(dur[i]) && !isnan(ev[i]);
    key[i] = valid ? dur[i] : -INFINITY;  // invalid rows sink to the end of the descending order
    idx[i] = valid ? i : -1;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const bool desc = ((i & k) == 0);  // overall descending
          const float a = key[i], b = key[ixj];
          const bool swap = desc ? (a < b) : (a > b);
          if (swap) {
            key[i] = b; key[ixj] = a;
            const int t = idx[i]; idx[i] = idx[ixj]; idx[ixj] = t;
          }
        }
      }
      __syncthreads();
    }
  }
  // hazards in sorted order
  for (int i = threadIdx.x; i < P; i += blockDim.x) scan[i] = (idx[i] >= 0) ? (double)expf(out[(long)idx[i] * ld]) : 0.0;
  __syncthreads();
  for (int off = 1; off < P; off <<= 1) {  // Hillis-Steele inclusive scan
    double v[PER];
    int c = 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x, ++c) v[c] = (i >= off) ? scan[i - off] : 0.0;
    __syncthreads();
    c = 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x, ++c) scan[i] += v[c];
    __syncthreads();
  }
  double num = 0.0, esum = 0.0, nvalid = 0.0;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    if (idx[i] >= 0) {
      const float e = ev[idx[i]];
      nvalid += 1.0;
      esum += (double)e;
      if (e == 1.0f) num += (double)out[(long)idx[i] * ld] - (double)logf((float)scan[i]);
    }
  }
  num = fx_block_sum_d(num, dred);
  esum = fx_block_sum_d(esum, dred);
  nvalid = fx_block_sum_d(nvalid, dred);
  const float total = (float)(-num / esum);
  const bool ok = nvalid > 0.0 && isfinite(total);
  // reverse scan of [e==1]/S_j  -> G_j = sum_{k >= j} [e_k==1]/S_k ; reuse key[] for hazards
  __syncthreads();
  for (int i = threadIdx.x; i < P; i += blockDim.x) {  // each i is owned by exactly one thread
    const double S = scan[i];
    key[i] = (idx[i] >= 0) ? expf(out[(long)idx[i] * ld]) : 0.f;
    scan[i] = (idx[i] >= 0 && ev[idx[i]] == 1.0f) ? 1.0 / S : 0.0;
  }
  __syncthreads();
  for (int off = 1; off < P; off <<= 1) {
    double v[PER];
    int c = 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x, ++c) v[c] = (i + off < P) ? scan[i + off] : 0.0;
    __syncthreads();
    c = 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x, ++c) scan[i] += v[c];
    __syncthreads();
  }
  const float w = loss_weight(logvar) * extra_scale;
  // Every row of dout is written exactly ONCE, with its final value: fx_heads_step runs this body in two workgroups per head (the
  // chain role and the weight-gradient role compute the same numbers and both store them), so a "zero everything, then scatter"
  // sequence lets one workgroup's zeros land between the other's final store and its read-back -- seen as run-to-run differences
  // of the Cox head's weight gradients once the two workgroups ran skewed (beside a wide product).
#ifdef FX_COX_TWO_PHASE          // (the round-3 form, kept for scripts/build_variant.py: it raced while two workgroups stored dout; one writer since round 5)
  for (int i = threadIdx.x; i < B; i += blockDim.x) dout[(long)i * ldd] = 0.f;
  __syncthreads();
  if (ok)
    for (int i = threadIdx.x; i < P; i += blockDim.x)
      if (idx[i] >= 0) {
        const float e1 = (ev[idx[i]] == 1.0f) ? 1.f : 0.f;
        const double gr = -((double)e1 - (double)key[i] * scan[i]) / esum;
        dout[(long)idx[i] * ldd] = w * (float)gr;
      }
  if (threadIdx.x == 0) loss_out[0] = ok ? total : 0.f;
  return;
#endif
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    if (idx[i] >= 0) {          // valid rows, by sorted position
      const float e1 = (ev[idx[i]] == 1.0f) ? 1.f : 0.f;
      const double gr = -((double)e1 - (double)key[i] * scan[i]) / esum;
      dout[(long)idx[i] * ldd] = ok ? w * (float)gr : 0.f;
    }
  }
  for (int i = threadIdx.x; i < B; i += blockDim.x)
    if (isnan(dur[i]) || isnan(ev[i])) dout[(long)i * ldd] = 0.f;      // rows without a label
  if (threadIdx.x == 0) loss_out[0] = ok ? total : 0.f;
}
