// fx_heads.hip -- all supervisor heads of a model in ONE launch each way.
//
// A supervisor head is the reference's MLP(latent -> supervisor_hidden -> n_out) (modules.py:106-150:
// Linear -> BatchNorm1d -> ReLU -> Dropout(0.1) -> Linear) applied to the [B, L] embedding
// (direct_pred.py:126-131, supervised_vae.py:190-196, triplet_encoder.py:160-164).  Every tensor involved is tiny
// (B <= 128 rows, L <= 128, hidden <= 32, outputs <= 32), so as separate GEMM / BatchNorm launches the heads were
// pure launch latency: 3 forward + 5 backward dependent launches per head at ~5-8 us each on the critical path
// between the wide forward and the dW+Adam kernels.  Here one workgroup per head does the whole forward through
// LDS, and one workgroup does the whole backward of all heads (sequentially, so the embedding gradient is summed in
// a fixed order: deterministic, no atomics).  Arithmetic is fp32 with the same expressions as fx_bn_act_fwd/bwd (same
// Philox stream for the dropout mask), so the fused and unfused paths agree to rounding; since round 3 the two
// [B, L]-sized products (layer_1 forward, the embedding gradient of fx_heads_step) run on the exact-fp32 matrix pipe
// (v_mfma_f32_16x16x4_f32) with their operands fetched straight from memory / the LDS tiles.
#include "fx_common.h"
#include "fx_chain_prof.h"
#include "fx_loss_dev.h"

#define FX_MAX_HEADS 8
constexpr int HB = 128;   // max rows
constexpr int HS = 32;    // max supervisor hidden width
constexpr int HC = 32;    // max head outputs
constexpr int HL = 128;   // max latent width
constexpr int HEADS_SHADOW = 4 * HB * HS + 2 * HS + 32;   // floats per head: the weight-gradient role's y1, a1, out, dout, statistics, loss (fx_heads_step)

struct FxHeadDesc {       // mirrors include/fxhip.h: fx_head_desc
  const float* W1; const float* b1; const float* gamma; const float* beta; float* rmean; float* rvar;
  const float* W2; const float* b2;
  float* y1; float* a1; float* save_mean; float* save_invstd; float* out;
  const float* mask;
  const float* dout;
  float* gW1; float* gb1; float* ggamma; float* gbeta; float* gW2; float* gb2;
  unsigned long long seed, offset;
  int S, C;
};

struct HeadsArgs {
  FxHeadDesc h[FX_MAX_HEADS];
  int n_heads;
  const float* x; long ldx;
  float* dx; long lddx; int dx_accumulate;
  int B, L, train;
  float drop_p;
  const float* ctrl;
  float* dx_part;      // optional scratch [n_heads][B][L] + one unsigned counter behind it: per-head dx workgroups
  unsigned* dx_count;
};

__device__ __forceinline__ unsigned long long heads_step_offset(const float* ctrl, unsigned long long offset) {
  return ctrl ? offset + (((unsigned long long)ctrl[FXC_STEP]) << 44) : offset;   // same layout as fx_norm_act.hip
}

// column sums over rows of a [HB][33] LDS tile: thread (col = t&31, rg = t>>5) adds rows rg, rg+8, ...; the 8 partials
// are then added in a fixed order by every thread of the column
__device__ __forceinline__ float heads_colsum(float partial, float (*part)[32], int col, int rg) {
  __syncthreads();
  part[rg][col] = partial;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += part[i][col];
  return s;
}

// Every global read inside a loop is staged through LDS first: a single workgroup has nothing to hide memory
// latency behind, and a dependent global load per loop iteration (~0.5 us each) made the first version of these
// kernels slower than the eight launches they replace.
typedef float Tile[HB][33];

// NOTE on every load below: addresses are CLAMPED into range and the load is unconditional; the select happens on the
// loaded value.  `cond ? p[i] : 0` makes hipcc wrap each load in its own exec-mask region with an s_waitcnt vmcnt(0)
// behind it, i.e. 16 dependent round trips to memory per staging call (measured: 45 us for the forward kernel).
// dst[rr][cc] = src[rr, col0 + cc] for rr < B, col0 + cc < ncols, else 0   (32-column chunk, coalesced, 16 loads in flight)
__device__ __forceinline__ void heads_stage(Tile dst, const float* __restrict__ src, long ld, int B, int col0, int ncols) {
  const int t = threadIdx.x;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = t + 256 * i, rr = idx >> 5, cc = idx & 31;
    v[i] = src[(long)min(rr, B - 1) * ld + min(col0 + cc, ncols - 1)];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = t + 256 * i, rr = idx >> 5, cc = idx & 31;
    dst[rr][cc] = (rr < B && col0 + cc < ncols) ? v[i] : 0.f;
  }
}
template <int MAXN>
__device__ __forceinline__ void heads_copy(float* dst, const float* __restrict__ src, int n) {
  float v[MAXN / 256];
#pragma unroll
  for (int i = 0; i < MAXN / 256; ++i) v[i] = src[min((int)threadIdx.x + 256 * i, n - 1)];
#pragma unroll
  for (int i = 0; i < MAXN / 256; ++i)
    if ((int)threadIdx.x + 256 * i < n) dst[threadIdx.x + 256 * i] = v[i];
}

// Dropout mask for 4 consecutive elements i0 .. i0+3 of the Philox stream used by fx_bn_act_fwd
// (element i -> word i&3 of block i>>2): two blocks cover any alignment; selection is branch-free.
__device__ __forceinline__ void heads_mask4(unsigned long long seed, unsigned long long off, unsigned long long i0, float keep,
                                            float mk[4]) {
  uint32_t b[8];
  fx_philox4(seed, off + (i0 >> 2), b);
  if ((i0 & 3) == 0) {                       // aligned (hidden width a multiple of 4): the four elements are one counter block
#pragma unroll
    for (int k = 0; k < 4; ++k) mk[k] = fx_u01(b[k]) <= keep ? 1.f : 0.f;
    return;
  }
  fx_philox4(seed, off + (i0 >> 2) + 1, b + 4);
  const int sh = (int)(i0 & 3);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint32_t w = b[k];
    w = sh == 1 ? b[k + 1] : w;
    w = sh == 2 ? b[k + 2] : w;
    w = sh == 3 ? b[k + 3] : w;
    mk[k] = fx_u01(w) <= keep ? 1.f : 0.f;
  }
}

// All per-column loops run over the PADDED width HS = 32 with zero-padded parameters, so they have compile-time trip
// counts and no predicates; only global stores are predicated on the true width.
struct HeadsFwdLds {
  Tile ys;                       // layer_1 output, then the block output (columns >= S stay zero)
  float W2s[HC * HS];            // [C][32], columns >= S zero
  float part[8][32];
  float stat[6][32];             // mean, invstd, gamma, beta, layer_1 bias, layer_out bias (zero padded)
};

__device__ __forceinline__ void heads_fwd_body(const HeadsArgs& a, const FxHeadDesc& h, HeadsFwdLds& F, bool update_running = true) {
  Tile& ys = F.ys;
  float* W2s = F.W2s;
  float (*part)[32] = F.part;
  float (*stat)[32] = F.stat;
  const int t = threadIdx.x, r = t & 127, hf = t >> 7, col = t & 31, rg = t >> 5;
  const int S = h.S, C = h.C, B = a.B, L = a.L;
  const int s0 = hf * (HS / 2);
  // ---- requested at entry, used several barriers later (round 6: each of these used to be a dependent round trip of the ONE workgroup
  // the whole step waits for -- the running statistics behind the batch statistics, the step counter in front of the dropout draws,
  // the layer_out bias in front of the last product; scripts/chain_profile.py: 14 of the launch's 30 us were this forward)
  const int scc = min(col, S - 1);
  const float rm0 = h.rmean[scc], rv0 = h.rvar[scc];
  const float stepf = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ctrl ? a.ctrl : h.W1), 0, a.ctrl ? FXC_SIZE * 4 : 0, 0x00020000), FXC_STEP * 4, 0, 0));
  // ---- layer_1 on the exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32): y1[r, s] = b1[s] + sum_l x[r, l] W1[s, l].
  // Operand A: lane (m = lane & 15, kq = lane >> 4) supplies x[row m][k], operand B: lane (n = lane & 15, kq) supplies
  // W1[s = n][k]; a lane fetches four consecutive k = 16 q + 4 kq + e straight from memory (x rows and W1 rows are both
  // contiguous in k) and instruction e uses element e on both sides.  Wave w owns rows 32 w .. 32 w + 31 (two row blocks)
  // x the 32 padded hidden columns (two column blocks); the result lands in the LDS tile ys for the BatchNorm phases.
  typedef float hf4 __attribute__((ext_vector_type(4)));
  const int lane = t & 63, wv_ = t >> 6, l15 = lane & 15, kq = lane >> 4;
  hf4 xa[2][HL / 16], wb[2][HL / 16];
  const int nq = (L + 15) >> 4;
  // one 16-byte load per operand fragment when the rows allow it (L % 16 == 0, 16-byte aligned rows: the engine's embeddings),
  // else four clamped scalar loads
  const bool vec = (L & 15) == 0 && (a.ldx & 3) == 0 && ((((uintptr_t)a.x) | ((uintptr_t)h.W1)) & 15) == 0;
  if (vec) {
#pragma unroll
    for (int q = 0; q < HL / 16; ++q) {
      if (q < nq) {
        const int k = 16 * q + 4 * kq;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int rr = 32 * wv_ + 16 * rb + l15;
          xa[rb][q] = *reinterpret_cast<const hf4*>(a.x + (long)min(rr, B - 1) * a.ldx + k);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) wb[cb][q] = *reinterpret_cast<const hf4*>(h.W1 + (long)min(16 * cb + l15, S - 1) * L + k);
      }
    }
#pragma unroll
    for (int q = 0; q < HL / 16; ++q) {
      if (q < nq) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          if (32 * wv_ + 16 * rb + l15 >= B) xa[rb][q] = hf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          if (16 * cb + l15 >= S) wb[cb][q] = hf4{0.f, 0.f, 0.f, 0.f};
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < HL / 16; ++q) {
      if (q < nq) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = 16 * q + 4 * kq + e, kc = min(k, L - 1);
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            const int rr = 32 * wv_ + 16 * rb + l15;
            const float v = a.x[(long)min(rr, B - 1) * a.ldx + kc];
            xa[rb][q][e] = (rr < B && k < L) ? v : 0.f;
          }
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            const int sc = 16 * cb + l15;
            const float v = h.W1[(long)min(sc, S - 1) * L + kc];
            wb[cb][q][e] = (sc < S && k < L) ? v : 0.f;
          }
        }
      }
    }
  }
  {
    float v[HC * HS / 256];
#pragma unroll
    for (int i = 0; i < HC * HS / 256; ++i) {
      const int idx = t + 256 * i, c = idx >> 5, s_ = idx & 31;
      v[i] = h.W2[min(c, C - 1) * S + min(s_, S - 1)];
    }
#pragma unroll
    for (int i = 0; i < HC * HS / 256; ++i) {
      const int idx = t + 256 * i, c = idx >> 5, s_ = idx & 31;
      W2s[idx] = (c < C && s_ < S) ? v[i] : 0.f;
    }
  }
  if (t < 32) {
    const int sc = min(t, S - 1);
    const float g_ = h.gamma[sc], b_ = h.beta[sc], b1_ = h.b1[sc];
    const float b2_ = (h.b2 ? h.b2 : h.b1)[h.b2 ? min(t, C - 1) : sc];
    stat[2][t] = t < S ? g_ : 0.f;
    stat[3][t] = t < S ? b_ : 0.f;
    stat[4][t] = t < S ? b1_ : 0.f;
    stat[5][t] = (h.b2 && t < C) ? b2_ : 0.f;
  }
  __syncthreads();
  {
    hf4 acc[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const float b1v = stat[4][16 * cb + l15];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) acc[rb][cb] = hf4{b1v, b1v, b1v, b1v};
    }
#pragma unroll
    for (int q = 0; q < HL / 16; ++q) {
      if (q < nq) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
              acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[rb][q][e], wb[cb][q][e], acc[rb][cb], 0, 0, 0);
      }
    }
    // result layout: lane holds rows 4 kq + i (i = 0..3) of the 16 x 16 block, column l15
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = 32 * wv_ + 16 * rb + 4 * kq + i, sc = 16 * cb + l15;
          const float v = (rr < B && sc < S) ? acc[rb][cb][i] : 0.f;
          ys[rr][sc] = v;
          if (rr < B && sc < S && h.y1) h.y1[(long)rr * S + sc] = v;
        }
  }
  // ---- BatchNorm statistics (two-pass, biased variance for normalisation, unbiased for running_var)
  float mean = 0.f, invstd = 0.f;
  __syncthreads();                                        // ys complete
  if (a.train) {
    float p = 0.f;
    for (int rr = rg; rr < B; rr += 8) p += ys[rr][col];
    mean = heads_colsum(p, part, col, rg) / (float)B;
    float q = 0.f;
    for (int rr = rg; rr < B; rr += 8) {
      const float d = ys[rr][col] - mean;
      q += d * d;
    }
    const float var_b = heads_colsum(q, part, col, rg) / (float)B;
    invstd = 1.0f / sqrtf(var_b + FX_BN_EPS);
    if (col < S && rg == 0) {
      h.save_mean[col] = mean;
      h.save_invstd[col] = invstd;
      const float var_u = B > 1 ? var_b * ((float)B / (float)(B - 1)) : var_b;
      if (update_running) {
        h.rmean[col] = (1.0f - FX_BN_MOMENTUM) * rm0 + FX_BN_MOMENTUM * mean;
        h.rvar[col] = (1.0f - FX_BN_MOMENTUM) * rv0 + FX_BN_MOMENTUM * var_u;
      }
    }
  } else {
    mean = rm0;
    invstd = 1.0f / sqrtf(rv0 + FX_BN_EPS);
  }
  if (rg == 0) { stat[0][col] = mean; stat[1][col] = invstd; }
  __syncthreads();
  // ---- normalise + ReLU + dropout (mask scaled first, then multiplied == F.dropout on CPU)
  const bool drop = a.train && a.drop_p > 0.f;
  const float keep_scale = 1.0f / (1.0f - a.drop_p);
  float mk[HS / 2];
  if (drop && h.mask) {
#pragma unroll
    for (int j = 0; j < HS / 2; ++j) mk[j] = h.mask[(long)min(r, B - 1) * S + min(s0 + j, S - 1)];
  } else if (drop) {
    const unsigned long long rng_off = a.ctrl ? h.offset + (((unsigned long long)stepf) << 44) : h.offset;    // = heads_step_offset
#pragma unroll
    for (int jb = 0; jb < HS / 8; ++jb) {
      if (s0 + 4 * jb < S) {                 // (padded columns hold zeros whatever the mask: no draw)
        heads_mask4(h.seed, rng_off, (unsigned long long)r * S + s0 + 4 * jb, 1.0f - a.drop_p, mk + 4 * jb);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) mk[4 * jb + k] = 1.f;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < HS / 2; ++j) mk[j] = 1.f;
  }
  float yv[HS / 2];
#pragma unroll
  for (int j = 0; j < HS / 2; ++j) {
    const int s = s0 + j;
    float y = (ys[r][s] - stat[0][s]) * stat[1][s] * stat[2][s] + stat[3][s];
    y = fmaxf(y, 0.f);
    if (drop) y = y * (mk[j] * keep_scale);
    yv[j] = (r < B) ? y : 0.f;
  }
#pragma unroll
  for (int j = 0; j < HS / 2; ++j) ys[r][s0 + j] = yv[j];
  if (r < B && h.a1) {
#pragma unroll
    for (int j = 0; j < HS / 2; ++j)
      if (s0 + j < S) h.a1[(long)r * S + s0 + j] = yv[j];
  }
  __syncthreads();
  // ---- layer_out: out[r, c] = b2[c] + sum_s a1[r, s] W2[c, s]   (padded columns of a1 / W2 are zero)
  const int Ch = (C + 1) >> 1;
  for (int c = hf * Ch; c < min(C, (hf + 1) * Ch); ++c) {
    float o = stat[5][c];
#pragma unroll
    for (int s = 0; s < HS; ++s) o += ys[r][s] * W2s[c * HS + s];
    if (r < B) h.out[(long)r * C + c] = o;
  }
}

__global__ __launch_bounds__(256) void fx_heads_fwd_kernel(HeadsArgs a) {
  __shared__ HeadsFwdLds F;
  heads_fwd_body(a, a.h[blockIdx.x], F);
}

// Backward.  grid = n_heads + 1 workgroups:
//   workgroups 0 .. n_heads-1 : parameter gradients of head i (layer_out, BatchNorm, layer_1)
//   workgroup  n_heads        : the embedding gradient dx (+)= sum over heads, heads in order (fixed summation order)
// Both kinds run the same PREFIX per head -- dout.W2 gated by the saved block output, then the BatchNorm backward --
// which leaves dy1 (the gradient at the layer_1 output) in the LDS tile R3; recomputing ~5 us of prefix is cheaper
// than a dependent launch.  Every global operand a phase needs is requested at the start of the head (registers), so
// a head costs about two memory round trips instead of six.
struct HeadsBwdLds {
  Tile R1;                 // dout (columns >= C zero), later 32-column chunks of the embedding
  Tile R2;                 // saved block output, then x-hat, then layer_1.weight as [S][L]
  Tile R3;                 // grad at the BatchNorm output, then dy1
  float W2s[HC * HS];      // [C][32], columns >= S zero
  float part[8][32];
  float stat[3][32];       // mean, invstd, gamma (zero padded)
};

// stages dout / a1 / W2 / statistics and leaves the gradient at the BatchNorm output in R3
__device__ __forceinline__ void heads_bwd_prefix(HeadsBwdLds& L, const FxHeadDesc& h, int B, float gate_scale) {
  const int t = threadIdx.x, r = t & 127, hf = t >> 7;
  const int S = h.S, C = h.C, s0 = hf * (HS / 2);
  // ---- request everything at once: dout, a1 (16 values each), W2 (4), statistics
  float vd[16], va[16], vw[HC * HS / 256];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = t + 256 * i, rr = min(idx >> 5, B - 1), cc = idx & 31;
    vd[i] = h.dout[(long)rr * C + min(cc, C - 1)];
    va[i] = h.a1[(long)rr * S + min(cc, S - 1)];
  }
#pragma unroll
  for (int i = 0; i < HC * HS / 256; ++i) {
    const int idx = t + 256 * i;
    vw[i] = h.W2[min(idx >> 5, C - 1) * S + min(idx & 31, S - 1)];
  }
  float m_ = 0.f, i_ = 0.f, g_ = 0.f;
  if (t < 32) {
    const int sc = min(t, S - 1);
    m_ = h.save_mean[sc]; i_ = h.save_invstd[sc]; g_ = h.gamma[sc];
  }
  __syncthreads();                                     // previous head's readers are done with the tiles
  if (t < 32) {
    L.stat[0][t] = t < S ? m_ : 0.f;
    L.stat[1][t] = t < S ? i_ : 0.f;
    L.stat[2][t] = t < S ? g_ : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = t + 256 * i, rr = idx >> 5, cc = idx & 31;
    L.R1[rr][cc] = (rr < B && cc < C) ? vd[i] : 0.f;
    L.R2[rr][cc] = (rr < B && cc < S) ? va[i] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < HC * HS / 256; ++i) {
    const int idx = t + 256 * i;
    L.W2s[idx] = ((idx >> 5) < C && (idx & 31) < S) ? vw[i] : 0.f;
  }
  __syncthreads();
  // ---- grad at the BatchNorm output: (dout . W2) gated by the saved block output (ReLU and dropout in one test)
  {
    float d[HS / 2];
#pragma unroll
    for (int j = 0; j < HS / 2; ++j) d[j] = 0.f;
    for (int c = 0; c < C; ++c) {
      const float dv = L.R1[r][c];
#pragma unroll
      for (int j = 0; j < HS / 2; ++j) d[j] = fmaf(dv, L.W2s[c * HS + s0 + j], d[j]);
    }
#pragma unroll
    for (int j = 0; j < HS / 2; ++j) L.R3[r][s0 + j] = (L.R2[r][s0 + j] > 0.f) ? d[j] * gate_scale : 0.f;
  }
  // (the caller's layer_out gradients read R1/R2 between these two barriers -- see fx_heads_bwd_kernel)
}

__device__ __forceinline__ void heads_bwd_bn(HeadsBwdLds& L, int B, const float vy[16], float& sum_dy, float& sum_dy_xh,
                                             float& sum_dx) {
  const int t = threadIdx.x, col = t & 31, rg = t >> 5;
  // x-hat of the saved layer_1 output replaces the block output in R2
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = t + 256 * i, rr = idx >> 5, cc = idx & 31;
    L.R2[rr][cc] = (rr < B) ? (vy[i] - L.stat[0][cc]) * L.stat[1][cc] : 0.f;       // padded columns: invstd = 0
  }
  __syncthreads();
  // BatchNorm backward (same expressions as fx_bn_bwd_kernel); padded columns have gamma = 0
  const float invstd = L.stat[1][col], gm = L.stat[2][col];
  float s1 = 0.f, s2 = 0.f;
  for (int rr = rg; rr < B; rr += 8) {
    const float dy = L.R3[rr][col];
    s1 += dy;
    s2 += dy * L.R2[rr][col];
  }
  sum_dy = heads_colsum(s1, L.part, col, rg);
  sum_dy_xh = heads_colsum(s2, L.part, col, rg);
  float sb = 0.f;
  const float invB = 1.0f / (float)B;
  for (int rr = rg; rr < B; rr += 8) {
    const float d = gm * invstd * (L.R3[rr][col] - invB * sum_dy - L.R2[rr][col] * invB * sum_dy_xh);
    L.R3[rr][col] = d;          // each (row, col) is owned by exactly one thread here
    sb += d;
  }
  sum_dx = heads_colsum(sb, L.part, col, rg);
  __syncthreads();
}

__global__ __launch_bounds__(256) void fx_heads_bwd_kernel(HeadsArgs a) {
  __shared__ HeadsBwdLds L;
  const int t = threadIdx.x, r = t & 127, hf = t >> 7, col = t & 31, rg = t >> 5;
  const int B = a.B, Ld = a.L;
  const float gate_scale = 1.0f / (1.0f - a.drop_p);
  const bool dx_role = (int)blockIdx.x >= a.n_heads;
  if (!dx_role) {
    // ================= parameter gradients of head blockIdx.x =================
    const FxHeadDesc& h = a.h[blockIdx.x];
    const int S = h.S, C = h.C;
    // y1 for the x-hat and both 32-column chunks... the embedding is staged chunk by chunk below
    float vy[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = t + 256 * i;
      vy[i] = h.y1[(long)min(idx >> 5, B - 1) * S + min(idx & 31, S - 1)];
    }
    float sum_dy, sum_dy_xh, sum_dx;
    heads_bwd_prefix(L, h, B, gate_scale);
    // layer_out.weight / bias gradients (over the padded [C][32] grid; only true columns are stored)
    for (int o = t; o < C * HS; o += 256) {
      const int c = o >> 5, s = o & 31;
      float g = 0.f;
#pragma unroll 8
      for (int rr = 0; rr < HB; ++rr) g = fmaf(L.R1[rr][c], L.R2[rr][s], g);      // rows >= B are zero
      if (s < S) h.gW2[c * S + s] = g;
    }
    if (h.gb2 && t < C) {
      float g = 0.f;
#pragma unroll 8
      for (int rr = 0; rr < HB; ++rr) g += L.R1[rr][t];
      h.gb2[t] = g;
    }
    __syncthreads();
    heads_bwd_bn(L, B, vy, sum_dy, sum_dy_xh, sum_dx);
    if (col < S && rg == 0) {
      h.ggamma[col] = sum_dy_xh;
      h.gbeta[col] = sum_dy;
      h.gb1[col] = sum_dx;
    }
    // layer_1.weight gradient: gW1[s, l] = sum_r dy1[r, s] x[r, l], 32 columns of x at a time through R1
    for (int c0 = 0; c0 < Ld; c0 += 32) {
      __syncthreads();
      heads_stage(L.R1, a.x, a.ldx, B, c0, Ld);
      __syncthreads();
      for (int o = t; o < S * 32; o += 256) {
        const int s = o >> 5, l = o & 31;
        float g = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < HB; ++rr) g = fmaf(L.R3[rr][s], L.R1[rr][l], g);
        if (c0 + l < Ld) h.gW1[(long)s * Ld + c0 + l] = g;
      }
    }
    return;
  }
  // ================= embedding gradient =================
  // One workgroup accumulates over all heads in registers (dx_part == NULL), or -- with scratch -- one workgroup per
  // head writes that head's contribution and the last one to finish adds the contributions in head order
  // (deterministic), so several heads cost one head's latency chain instead of their sum.
  if (!a.dx) return;
  const bool split = a.dx_part != nullptr && a.n_heads > 1;
  const int h_lo = split ? (int)blockIdx.x - a.n_heads : 0, h_hi = split ? h_lo + 1 : a.n_heads;
  const int Lh = (Ld + 1) >> 1, l0 = hf * Lh;
  float* W1s = &L.R2[0][0];             // HB*33 floats >= HS*HL
  float accx[HL / 2];
#pragma unroll
  for (int j = 0; j < HL / 2; ++j) accx[j] = 0.f;
  if (a.dx_accumulate && !split) {
#pragma unroll
    for (int j = 0; j < HL / 2; ++j) accx[j] = a.dx[(long)min(r, B - 1) * a.lddx + min(l0 + j, Ld - 1)];   // only in-range lanes are stored
  }
  for (int hi = h_lo; hi < h_hi; ++hi) {
    const FxHeadDesc& h = a.h[hi];
    const int S = h.S;
    float vy[16], vw1[HS * HL / 256];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int idx = t + 256 * i;
      vy[i] = h.y1[(long)min(idx >> 5, B - 1) * S + min(idx & 31, S - 1)];
    }
#pragma unroll
    for (int i = 0; i < HS * HL / 256; ++i) vw1[i] = h.W1[min(t + 256 * i, S * Ld - 1)];
    float sum_dy, sum_dy_xh, sum_dx;
    heads_bwd_prefix(L, h, B, gate_scale);
    __syncthreads();
    heads_bwd_bn(L, B, vy, sum_dy, sum_dy_xh, sum_dx);
    // layer_1.weight staged over the dead x-hat; branch-free FMA loop: lanes j with l0 + j >= L accumulate neighbouring
    // weights and are never stored
#pragma unroll
    for (int i = 0; i < HS * HL / 256; ++i)
      if (t + 256 * i < S * Ld) W1s[t + 256 * i] = vw1[i];
    __syncthreads();
    for (int s = 0; s < S; ++s) {
      const float d = L.R3[r][s];
      const float* w = W1s + s * Ld + l0;
#pragma unroll
      for (int j = 0; j < HL / 2; ++j) accx[j] = fmaf(d, w[j], accx[j]);
    }
  }
  if (!split) {
    if (r < B) {
#pragma unroll
      for (int j = 0; j < HL / 2; ++j)
        if (j < Lh && l0 + j < Ld) a.dx[(long)r * a.lddx + l0 + j] = accx[j];
    }
    return;
  }
  // per-head contribution -> scratch; the last workgroup to arrive sums them in head order
  if (r < B) {
    float* mine = a.dx_part + ((long)h_lo * B + r) * Ld;
#pragma unroll
    for (int j = 0; j < HL / 2; ++j)
      if (j < Lh && l0 + j < Ld) mine[l0 + j] = accx[j];
  }
  __threadfence();
  __syncthreads();
  __shared__ int s_last;
  if (t == 0) s_last = (atomicAdd(a.dx_count, 1u) == (unsigned)(a.n_heads - 1));
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (r < B) {
#pragma unroll
    for (int j = 0; j < HL / 2; ++j) {
      if (j < Lh && l0 + j < Ld) {
        float sacc = a.dx_accumulate ? a.dx[(long)r * a.lddx + l0 + j] : 0.f;
        for (int hi = 0; hi < a.n_heads; ++hi) sacc += __builtin_nontemporal_load(a.dx_part + ((long)hi * B + r) * Ld + l0 + j);
        a.dx[(long)r * a.lddx + l0 + j] = sacc;
      }
    }
  }
  if (t == 0) *a.dx_count = 0u;          // ready for the next launch / graph replay
}

// ---- one launch for the whole supervisor part of a training step ---------------------------------------------------------
// forward -> loss (value + output gradient) -> backward of every head, the summed embedding gradient and the model's total
// loss: what used to be fx_heads_fwd -> fx_mse_masked | fx_ce_masked | fx_cox_ph (one launch per head) -> fx_total_loss ->
// fx_heads_bwd, i.e. 4 + n_heads dependent launches (17 + 6 + 5 + 38 us plus ~25 us of graph-edge gaps at cfg2,
// profiles/r03_timeline_cfg2_a.txt).  One workgroup per head runs its head end to end: everything a phase needs from the
// previous one is already in this workgroup's registers, LDS or L2.  The workgroups meet once, at the end: each publishes its
// loss value and its share of the embedding gradient, and the last one to arrive adds the shares in head order
// (deterministic) and evaluates the uncertainty-weighted total (direct_pred.py:192-223) over ALL named loss terms of the
// model -- terms computed by earlier launches (triplet, MMD) are read from their slots.
// Two workgroups per head: workgroup 2 h runs the chain the NEXT launch waits for (forward, loss, BatchNorm backward, the
// embedding-gradient share, the meeting), workgroup 2 h + 1 the weight gradients of the head (layer_out, layer_1); it
// recomputes forward and loss for itself -- the same instructions on the same inputs, so what both write (saved
// activations, loss value, output gradient) is bit-identical -- and takes ~8 us of matrix products off the critical chain.
struct HeadsStepArgs {
  HeadsArgs ha;
  int kind[FX_MAX_HEADS];                  // 0 masked MSE, 1 masked softmax-CE, 2 Cox partial likelihood
  const float* y[FX_MAX_HEADS];            // labels [B] (Cox: event indicators)
  const float* dur[FX_MAX_HEADS];          // Cox: durations [B]
  const float* logvar[FX_MAX_HEADS];       // uncertainty weight of the head's loss, or NULL
  float* loss[FX_MAX_HEADS];               // raw loss value of the head
  int n_terms, weighted;
  const float* term_loss[16]; const float* term_logvar[16]; float* term_dlogvar[16];
  float* total_out; float* epoch_acc;
  float* shadow;                           // [n_heads][HEADS_SHADOW]: where the weight-gradient role keeps ITS copies of the saved tensors
};

union __attribute__((aligned(16))) HeadsStepLds {
  HeadsFwdLds f;
  HeadsBwdLds b;
};

__global__ __launch_bounds__(256) void fx_heads_step_kernel(HeadsStepArgs sa) {
  __shared__ HeadsStepLds U;
  __shared__ float ckey[HB];
  __shared__ int cidx[HB];
  __shared__ double cscan[HB];
  __shared__ double dred[16];
  __shared__ float sm[16];
  __shared__ int s_last;
  const HeadsArgs& a = sa.ha;
  const int hi = blockIdx.x >> 1;
  const bool chain_role = (blockIdx.x & 1) == 0;       // 0: the critical chain; 1: the head's weight gradients
  // Both roles run the head's forward and its loss (the weight-gradient role needs y1 / a1 / dout and cannot wait for another
  // workgroup).  Every SHARED output -- the saved tensors y1, a1, save_mean, save_invstd, out, the output gradient dout, the loss value,
  // the running statistics -- is stored by the chain role ONLY; the weight-gradient role stores and re-reads private copies in its
  // shadow block (two writers of one array were what the Cox race of round 4 needed).
  FxHeadDesc h = a.h[hi];
  float* loss_slot = sa.loss[hi];
  if (!chain_role) {
    float* sh = sa.shadow + (long)hi * HEADS_SHADOW;
    h.y1 = sh; h.a1 = sh + HB * HS; h.out = sh + 2 * HB * HS; h.dout = sh + 3 * HB * HS;
    h.save_mean = sh + 4 * HB * HS; h.save_invstd = sh + 4 * HB * HS + HS; loss_slot = sh + 4 * HB * HS + 2 * HS;
  }
  const int t = threadIdx.x, r = t & 127, hf = t >> 7, col = t & 31, rg = t >> 5;
  const int B = a.B, Ld = a.L, S = h.S, C = h.C;
  CP_TICK(blockIdx.x, 0);
  heads_fwd_body(a, h, U.f, chain_role);
  __syncthreads();                                   // out [B, C] is visible to the whole workgroup
  CP_TICK(blockIdx.x, 1);
  // ================= loss value + gradient at the head output =================
  float* dout = const_cast<float*>(h.dout);
  if (sa.kind[hi] == 0) loss_mse_body(loss_slot, dout, h.out, sa.y[hi], B, C, C, sa.logvar[hi], 1.0f, sm);
  else if (sa.kind[hi] == 1) loss_ce_body(loss_slot, dout, h.out, sa.y[hi], B, C, C, C, sa.logvar[hi], 1.0f, sm);
  else loss_cox_body<HB, 1>(loss_slot, dout, h.out, sa.dur[hi], sa.y[hi], B, C, C, sa.logvar[hi], 1.0f, ckey, cidx, cscan, dred);
  __syncthreads();
  CP_TICK(blockIdx.x, 2);
  // ================= backward: parameter gradients of this head and its share of the embedding gradient =================
  HeadsBwdLds& L = U.b;
  const float gate_scale = 1.0f / (1.0f - a.drop_p);
  float vy[16], vw1[HS * HL / 256];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = t + 256 * i;
    vy[i] = h.y1[(long)min(idx >> 5, B - 1) * S + min(idx & 31, S - 1)];
  }
#pragma unroll
  for (int i = 0; i < HS * HL / 256; ++i) vw1[i] = h.W1[min(t + 256 * i, S * Ld - 1)];
  // weight-gradient role: the B operand of gW1 = dy1^T . x (lane (l = lane & 15, kq): x[16 q + 4 kq + e][column block + l]) is
  // requested here, two dependent phases before its use; wave w owns column blocks w and w + 4 of the embedding
  typedef float hf4 __attribute__((ext_vector_type(4)));
  const int glane = t & 63, gw = t >> 6, gl15 = glane & 15, gkq = glane >> 4;
  hf4 gx[2][HB / 16];
  if (!chain_role) {
#pragma unroll
    for (int cbi = 0; cbi < 2; ++cbi) {
      const int cl = 16 * (gw + 4 * cbi) + gl15;
      if (16 * (gw + 4 * cbi) < Ld) {
#pragma unroll
        for (int q = 0; q < HB / 16; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int rr = 16 * q + 4 * gkq + e;
            const float v = a.x[(long)min(rr, B - 1) * a.ldx + min(cl, Ld - 1)];
            gx[cbi][q][e] = (rr < B && cl < Ld) ? v : 0.f;
          }
      }
    }
  }
  float sum_dy, sum_dy_xh, sum_dx;
  CP_TICK(blockIdx.x, 3);
  heads_bwd_prefix(L, h, B, gate_scale);
  CP_TICK(blockIdx.x, 4);
  if (!chain_role) {
    // layer_out.weight / bias gradients (over the padded [C][32] grid; only true columns are stored)
    for (int o = t; o < C * HS; o += 256) {
      const int c = o >> 5, s = o & 31;
      float g = 0.f;
#pragma unroll 8
      for (int rr = 0; rr < HB; ++rr) g = fmaf(L.R1[rr][c], L.R2[rr][s], g);      // rows >= B are zero
      if (s < S) h.gW2[c * S + s] = g;
    }
    if (h.gb2 && t < C) {
      float g = 0.f;
#pragma unroll 8
      for (int rr = 0; rr < HB; ++rr) g += L.R1[rr][t];
      h.gb2[t] = g;
    }
  }
  __syncthreads();
  heads_bwd_bn(L, B, vy, sum_dy, sum_dy_xh, sum_dx);
  CP_TICK(blockIdx.x, 5);
  if (!chain_role) {
    // ---- layer_1.weight gradient on the exact-fp32 matrix pipe: gW1[s, l] = sum_r dy1[r, s] x[r, l]
    // (A = dy1^T: lane (s = lane & 15, kq) reads R3[16 q + 4 kq + e][s block + s]; B = the x fragments requested above)
#pragma unroll
    for (int cbi = 0; cbi < 2; ++cbi) {
      const int cl = 16 * (gw + 4 * cbi) + gl15;
      if (16 * (gw + 4 * cbi) < Ld) {
#pragma unroll
        for (int sb = 0; sb < HS / 16; ++sb) {
          hf4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < HB / 16; ++q) {
            hf4 av;
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = L.R3[16 * q + 4 * gkq + e][16 * sb + gl15];
#pragma unroll
            for (int e = 0; e < 4; ++e) g = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], gx[cbi][q][e], g, 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int sr = 16 * sb + 4 * gkq + i;
            if (sr < S && cl < Ld) h.gW1[(long)sr * Ld + cl] = g[i];
          }
        }
      }
    }
    return;
  }
  if (col < S && rg == 0) {
    h.ggamma[col] = sum_dy_xh;
    h.gbeta[col] = sum_dy;
    h.gb1[col] = sum_dx;
  }
  // ---- embedding gradient share: dy1 . W1 (layer_1.weight staged over the dead x-hat)
  const bool want_dx = a.dx != nullptr;
  const bool split = a.n_heads > 1;
  if (want_dx) {
    // dx[r, l] = sum_s dy1[r, s] W1[s, l] on the exact-fp32 matrix pipe: A = dy1 (LDS tile R3, padded hidden columns are
    // zero), B = layer_1.weight staged as [S][Ld] over the dead x-hat; wave w owns rows 32 w .. 32 w + 31, all column blocks
    typedef float hf4 __attribute__((ext_vector_type(4)));
    float* W1s = &L.R2[0][0];
#pragma unroll
    for (int i = 0; i < HS * HL / 256; ++i)
      if (t + 256 * i < S * Ld) W1s[t + 256 * i] = vw1[i];
    __syncthreads();
    const int lane = t & 63, wv_ = t >> 6, l15 = lane & 15, kq = lane >> 4;
    hf4 av[2][HS / 16];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int q = 0; q < HS / 16; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) av[rb][q][e] = L.R3[32 * wv_ + 16 * rb + l15][16 * q + 4 * kq + e];
    float* dst = split ? a.dx_part + (long)hi * B * Ld : a.dx;
    const long ldd = split ? (long)Ld : a.lddx;
    const bool add_old = a.dx_accumulate && !split;
    const int ncb = (Ld + 15) >> 4;
#pragma unroll
    for (int cb = 0; cb < HL / 16; ++cb) {
      if (cb < ncb) {
        const int cl = 16 * cb + l15, clc = min(cl, Ld - 1);
        hf4 bv[HS / 16];
#pragma unroll
        for (int q = 0; q < HS / 16; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int sr = 16 * q + 4 * kq + e;
            const float v = W1s[min(sr, S - 1) * Ld + clc];
            bv[q][e] = (sr < S && cl < Ld) ? v : 0.f;
          }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          hf4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < HS / 16; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb][q][e], bv[q][e], acc, 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rr = 32 * wv_ + 16 * rb + 4 * kq + i;
            if (rr < B && cl < Ld) {
              float* d = dst + (long)rr * ldd + cl;
              *d = add_old ? *d + acc[i] : acc[i];
            }
          }
        }
      }
    }
  }
  CP_TICK(blockIdx.x, 6);
  const int Lh = (Ld + 1) >> 1, l0 = hf * Lh;      // (the meet below: a thread's rows / columns of the per-head shares)
  // ================= meet: the last workgroup adds the shares in head order and evaluates the total =================
  bool last = true;
  if (split) {
    __threadfence();
    __syncthreads();
    if (t == 0) s_last = (atomicAdd(a.dx_count, 1u) == (unsigned)(a.n_heads - 1));
    __syncthreads();
    last = s_last != 0;
    if (!last) return;
    __threadfence();
    if (want_dx && r < B) {
#pragma unroll
      for (int j = 0; j < HL / 2; ++j) {
        if (j < Lh && l0 + j < Ld) {
          float sacc = a.dx_accumulate ? a.dx[(long)r * a.lddx + l0 + j] : 0.f;
          for (int k = 0; k < a.n_heads; ++k) sacc += __builtin_nontemporal_load(a.dx_part + ((long)k * B + r) * Ld + l0 + j);
          a.dx[(long)r * a.lddx + l0 + j] = sacc;
        }
      }
    }
    if (t == 0) *a.dx_count = 0u;          // ready for the next launch / graph replay
  } else {
    __syncthreads();                       // thread 0's loss value (global) is ordered before the total below
  }
  if (t == 0 && sa.total_out) {
    float tot = 0.f;
    for (int i = 0; i < sa.n_terms; ++i) {
      const float l = __builtin_nontemporal_load(sa.term_loss[i]);
      if (sa.weighted) {
        const float sv = sa.term_logvar[i][0];
        const float w = expf(-sv);
        tot += w * l + sv;
        if (sa.term_dlogvar[i]) sa.term_dlogvar[i][0] = 1.0f - w * l;
      } else {
        tot += l;
      }
      if (sa.epoch_acc) sa.epoch_acc[i] += l;
    }
    if (sa.epoch_acc) {
      sa.epoch_acc[sa.n_terms] += tot;
      sa.epoch_acc[sa.n_terms + 1] += 1.0f;
    }
    sa.total_out[0] = tot;
  }
  CP_TICK(blockIdx.x, 7);
}

CP_READER(fx_debug_chain_stamps_heads)

extern "C" {

struct fx_head_desc;   // include/fxhip.h; layout identical to FxHeadDesc

static int heads_check(const FxHeadDesc* heads, int n_heads, const float* x, int B, int L, const char* who) {
  FX_REQUIRE(heads && x && n_heads > 0 && n_heads <= FX_MAX_HEADS, "%s: bad args (1..%d heads)", who, FX_MAX_HEADS);
  FX_REQUIRE(B > 0 && B <= HB && L > 0 && L <= HL, "%s: B=%d must be <= %d and latent=%d <= %d", who, B, HB, L, HL);
  for (int i = 0; i < n_heads; ++i)
    FX_REQUIRE(heads[i].S > 0 && heads[i].S <= HS && heads[i].C > 0 && heads[i].C <= HC && heads[i].W1 && heads[i].W2,
               "%s: head %d has hidden=%d (max %d), outputs=%d (max %d)", who, i, heads[i].S, HS, heads[i].C, HC);
  return 0;
}

int fx_heads_fwd(const void* heads_, int n_heads, const float* x, long ldx, int B, int L, int train, float drop_p,
                 const float* ctrl, hipStream_t stream) {
  const FxHeadDesc* heads = (const FxHeadDesc*)heads_;
  if (int rc = heads_check(heads, n_heads, x, B, L, "fx_heads_fwd")) return rc;
  FX_REQUIRE(!train || B > 1, "fx_heads_fwd: train mode needs B > 1");
  HeadsArgs a{};
  for (int i = 0; i < n_heads; ++i) {
    a.h[i] = heads[i];
    FX_REQUIRE(a.h[i].b1 && a.h[i].gamma && a.h[i].beta && a.h[i].rmean && a.h[i].rvar && a.h[i].out,
               "fx_heads_fwd: head %d has a null parameter/output pointer", i);
    FX_REQUIRE(!train || (a.h[i].save_mean && a.h[i].save_invstd && a.h[i].y1 && a.h[i].a1),
               "fx_heads_fwd: train mode needs y1/a1/save_mean/save_invstd (head %d)", i);
  }
  a.n_heads = n_heads; a.x = x; a.ldx = ldx; a.B = B; a.L = L; a.train = train; a.drop_p = drop_p; a.ctrl = ctrl;
  hipLaunchKernelGGL(fx_heads_fwd_kernel, dim3(n_heads), dim3(256), 0, stream, a);
  return fx_check_launch("fx_heads_fwd");
}

int fx_heads_bwd(const void* heads_, int n_heads, const float* x, long ldx, float* dx, long lddx, int dx_accumulate, int B,
                 int L, float drop_p, void* dx_scratch, hipStream_t stream) {
  const FxHeadDesc* heads = (const FxHeadDesc*)heads_;
  if (int rc = heads_check(heads, n_heads, x, B, L, "fx_heads_bwd")) return rc;
  HeadsArgs a{};
  for (int i = 0; i < n_heads; ++i) {
    a.h[i] = heads[i];
    FX_REQUIRE(a.h[i].dout && a.h[i].a1 && a.h[i].y1 && a.h[i].save_mean && a.h[i].save_invstd && a.h[i].gamma &&
                   a.h[i].gW1 && a.h[i].gb1 && a.h[i].ggamma && a.h[i].gbeta && a.h[i].gW2,
               "fx_heads_bwd: head %d has a null saved-tensor / gradient pointer", i);
  }
  a.n_heads = n_heads; a.x = x; a.ldx = ldx; a.dx = dx; a.lddx = lddx; a.dx_accumulate = dx_accumulate;
  a.B = B; a.L = L; a.train = 1; a.drop_p = drop_p;
  const bool split = dx_scratch != nullptr && dx != nullptr && n_heads > 1;
  if (split) {
    FX_REQUIRE((((uintptr_t)dx_scratch) & 3) == 0, "fx_heads_bwd: scratch must be 4-byte aligned");
    a.dx_part = (float*)dx_scratch;
    a.dx_count = (unsigned*)((float*)dx_scratch + (long)n_heads * B * L);    // zero on first use (caller zero-fills once)
  }
  hipLaunchKernelGGL(fx_heads_bwd_kernel, dim3(split ? 2 * n_heads : n_heads + 1), dim3(256), 0, stream, a);
  return fx_check_launch("fx_heads_bwd");
}

// floats of fx_heads_step's scratch: the per-head shares of dx + the arrival counter (as fx_heads_bwd), then one shadow block per head
long fx_heads_step_scratch_floats(int n_heads, int B, int L) {
  return (long)n_heads * B * L + 4 + (long)n_heads * HEADS_SHADOW;
}

// All heads of a training step in one launch: forward, loss, backward, summed embedding gradient, total loss (see
// fx_heads_step_kernel).  kinds[i]: 0 masked MSE, 1 masked softmax-CE, 2 Cox; labels[i] [B] (Cox: events), durations[i] (Cox
// only), logvars[i] / losses[i] the head's uncertainty weight (NULL: unweighted) and loss slot; heads[i].dout receives the
// output gradient.  terms: every named loss of the model in order (the heads' slots among them).
int fx_heads_step(const void* heads_, int n_heads, const int* kinds, const float* const* labels, const float* const* durations,
                  const float* const* logvars, float* const* losses, const float* x, long ldx, float* dx, long lddx,
                  int dx_accumulate, int B, int L, float drop_p, const float* ctrl, void* dx_scratch, int n_terms, int weighted,
                  const float* const* term_losses, const float* const* term_logvars, float* const* term_dlogvars, float* total_out,
                  float* epoch_acc, hipStream_t stream) {
  const FxHeadDesc* heads = (const FxHeadDesc*)heads_;
  if (int rc = heads_check(heads, n_heads, x, B, L, "fx_heads_step")) return rc;
  FX_REQUIRE(B > 1 && kinds && labels && losses, "fx_heads_step: train mode needs B > 1 and labels / loss slots");
  FX_REQUIRE(n_terms >= 0 && n_terms <= 16 && (n_terms == 0 || (term_losses && total_out)), "fx_heads_step: bad loss terms (n=%d)", n_terms);
  FX_REQUIRE(!weighted || n_terms == 0 || term_logvars, "fx_heads_step: weighted total needs log_vars");
  FX_REQUIRE(dx_scratch && (((uintptr_t)dx_scratch) & 3) == 0,
             "fx_heads_step: needs its scratch (fx_heads_step_scratch_floats(n_heads, B, L) floats, zero-filled once)");
  HeadsStepArgs s{};
  HeadsArgs& a = s.ha;
  for (int i = 0; i < n_heads; ++i) {
    a.h[i] = heads[i];
    FX_REQUIRE(a.h[i].b1 && a.h[i].gamma && a.h[i].beta && a.h[i].rmean && a.h[i].rvar && a.h[i].out && a.h[i].save_mean &&
                   a.h[i].save_invstd && a.h[i].y1 && a.h[i].a1 && a.h[i].dout && a.h[i].gW1 && a.h[i].gb1 && a.h[i].ggamma &&
                   a.h[i].gbeta && a.h[i].gW2,
               "fx_heads_step: head %d has a null parameter / saved-tensor / gradient pointer", i);
    FX_REQUIRE(kinds[i] >= 0 && kinds[i] <= 2 && labels[i] && losses[i] && (kinds[i] != 2 || (durations && durations[i])),
               "fx_heads_step: head %d: bad loss kind / labels", i);
    FX_REQUIRE(kinds[i] == 1 || a.h[i].C == 1, "fx_heads_step: head %d: MSE / Cox heads have one output", i);
    s.kind[i] = kinds[i]; s.y[i] = labels[i]; s.dur[i] = durations ? durations[i] : nullptr;
    s.logvar[i] = logvars ? logvars[i] : nullptr; s.loss[i] = losses[i];
  }
  a.n_heads = n_heads; a.x = x; a.ldx = ldx; a.dx = dx; a.lddx = lddx; a.dx_accumulate = dx_accumulate;
  a.B = B; a.L = L; a.train = 1; a.drop_p = drop_p; a.ctrl = ctrl;
  if (n_heads > 1) {
    a.dx_part = (float*)dx_scratch;
    a.dx_count = (unsigned*)((float*)dx_scratch + (long)n_heads * B * L);    // zero on first use (caller zero-fills once)
  }
  s.shadow = (float*)dx_scratch + (long)n_heads * B * L + 4;
  s.n_terms = n_terms; s.weighted = weighted; s.total_out = total_out; s.epoch_acc = epoch_acc;
  for (int i = 0; i < n_terms; ++i) {
    s.term_loss[i] = term_losses[i];
    s.term_logvar[i] = weighted ? term_logvars[i] : nullptr;
    s.term_dlogvar[i] = (weighted && term_dlogvars) ? term_dlogvars[i] : nullptr;
  }
  hipLaunchKernelGGL(fx_heads_step_kernel, dim3(2 * n_heads), dim3(256), 0, stream, s);
  return fx_check_launch("fx_heads_step");
}

}  // extern "C"
