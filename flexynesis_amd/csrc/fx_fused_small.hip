// Launch-count reduction for the narrow part of the step (cfg2: ~45 small launches of ~5-15 us each sit
// between the four HBM-bound wide-layer kernels; profiles/r01_b_*).  Every kernel here replaces two or
// three of the generic ones with a single pass:
//
//   fx_bn_act_fwd_r16 / _bwd_r16   BatchNorm(+act/dropout) for B <= 128 with the whole column held in
//                                  registers (16 rows per thread): one global read of x instead of three
//                                  dependent passes; the forward can take the split-K SLABS of the
//                                  preceding Linear directly (reduce + bias fused in).
//   fx_gram_hadamard               sum_ij (sum_z SX[z][ij]) * (sum_z SD[z][ij]) straight from the split-K
//                                  slabs of the two Gram GEMMs (replaces 2 slab reductions + the Hadamard sum).
//   fx_gather_split                cohort row gather + fp32 copy + bf16 hi/lo split + transposed hi/lo split
//                                  in one pass over the batch (replaces gather + 2 split kernels).
#include "fx_common.h"
#include "fx_small.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define OOB 0xFFFFFFF0u
#define ACT_NONE 0
#define ACT_LEAKY 1
#define ACT_RELU 2
#define LEAKY_SLOPE 0.2f
#define COLS 32
#define RGRP 8
#define RPT 16

__device__ __forceinline__ __amdgpu_buffer_rsrc_t s_rsrc(const void* p, long bytes) {
  const unsigned n = bytes > 0xFFFFFFF0L ? 0xFFFFFFF0u : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);
}
__device__ __forceinline__ float s_ld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void s_st1(float v, __amdgpu_buffer_rsrc_t r, unsigned off) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}
__device__ __forceinline__ float col_reduce16(float v, float (*red)[COLS], int cx, int ry) {
  __syncthreads();
  red[ry][cx] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < RGRP; ++i) s += red[i][cx];
  return s;
}
__device__ __forceinline__ unsigned long long step_off(const float* ctrl, unsigned long long offset) {
  return ctrl ? offset + (((unsigned long long)ctrl[FXC_STEP]) << 44) : offset;
}

__global__ __launch_bounds__(256) void fx_bn_fwd_r16_kernel(BnFwd16 a) {
  __shared__ float red[RGRP][COLS];
  const int cx = threadIdx.x & (COLS - 1), ry = threadIdx.x / COLS;
  const int c = blockIdx.x * COLS + cx;
  const bool cok = c < a.C;
  const unsigned cmask = cok ? 0u : OOB;
  float xv[RPT];
  if (a.slabs) {  // x = sum_z slab[z] + bias  (slabs are [nslabs][B][C] contiguous)
    const __amdgpu_buffer_rsrc_t rs = s_rsrc(a.slabs, (long)a.nslabs * a.slab_stride * 4);
    const float bv = (a.lin_bias && cok) ? a.lin_bias[c] : 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) xv[i] = bv;
    // z outermost: the 16 row loads of one slab are independent and in flight together (the other loop order
    // serialises nslabs x 16 dependent round trips: measured 45 us instead of ~12)
    for (int z = 0; z < a.nslabs; ++z) {
      float t[RPT];
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const int r = ry + RGRP * i;
        t[i] = s_ld1(rs, (unsigned)(((long)z * a.slab_stride + (long)r * a.C + c) * 4) | cmask | ((r < a.B) ? 0u : OOB));
      }
#pragma unroll
      for (int i = 0; i < RPT; ++i) xv[i] += t[i];
    }
    if (a.x_out) {
      const __amdgpu_buffer_rsrc_t rx = s_rsrc(a.x_out, (long)a.B * a.ldx * 4);
#pragma unroll
      for (int i = 0; i < RPT; ++i) s_st1(xv[i], rx, (unsigned)(((long)(ry + RGRP * i) * a.ldx + c) * 4) | cmask);
    }
  } else {
    const __amdgpu_buffer_rsrc_t rx = s_rsrc(a.x, (long)a.B * a.ldx * 4);
#pragma unroll
    for (int i = 0; i < RPT; ++i) xv[i] = s_ld1(rx, (unsigned)(((long)(ry + RGRP * i) * a.ldx + c) * 4) | cmask);
  }
  if (a.pre_act == ACT_LEAKY) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) xv[i] = xv[i] > 0.f ? xv[i] : xv[i] * LEAKY_SLOPE;
  }
  float mean, invstd;
  if (a.train) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) s += (ry + RGRP * i < a.B) ? xv[i] : 0.f;
    mean = col_reduce16(s, red, cx, ry) / (float)a.B;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const float d = xv[i] - mean;
      q += (ry + RGRP * i < a.B) ? d * d : 0.f;
    }
    const float var_b = col_reduce16(q, red, cx, ry) / (float)a.B;
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
  const float gm = cok ? a.gamma[c] : 0.f, bt = cok ? a.beta[c] : 0.f;
  const bool drop = a.train && a.drop_p > 0.f;
  const float keep_scale = 1.0f / (1.0f - a.drop_p);
  const unsigned long long rng_off = step_off(a.ctrl, a.offset);
  const __amdgpu_buffer_rsrc_t ro = s_rsrc(a.out, (long)a.B * a.ldo * 4);
  const __amdgpu_buffer_rsrc_t rmk = s_rsrc(a.mask, (long)a.B * a.C * 4);
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = ry + RGRP * i;
    float y = (xv[i] - mean) * invstd * gm + bt;
    if (a.post_act == ACT_RELU) y = fmaxf(y, 0.f);
    if (drop) {
      float mk;
      if (a.mask) mk = s_ld1(rmk, (unsigned)(((long)r * a.C + c) * 4) | cmask);
      else mk = fx_rand_uniform(a.seed, rng_off, (unsigned long long)r * a.C + c) <= (1.0f - a.drop_p) ? 1.f : 0.f;
      y = y * (mk * keep_scale);
    }
    s_st1(y, ro, (unsigned)(((long)r * a.ldo + c) * 4) | cmask);
  }
}

__global__ __launch_bounds__(256) void fx_bn_bwd_r16_kernel(BnBwd16 a) {
  __shared__ float red[RGRP][COLS];
  const int cx = threadIdx.x & (COLS - 1), ry = threadIdx.x / COLS;
  const int c = blockIdx.x * COLS + cx;
  const bool cok = c < a.C;
  const unsigned cmask = cok ? 0u : OOB;
  const float mean = cok ? a.save_mean[c] : 0.f, invstd = cok ? a.save_invstd[c] : 0.f;
  const float gm = cok ? a.gamma[c] : 0.f;
  const float gate_scale = 1.0f / (1.0f - a.drop_p);
  const __amdgpu_buffer_rsrc_t rx = s_rsrc(a.x, (long)a.B * a.ldx * 4);
  const __amdgpu_buffer_rsrc_t rd = s_rsrc(a.dout, (long)a.B * a.lddo * 4);
  const __amdgpu_buffer_rsrc_t rout = s_rsrc(a.out, (long)a.B * a.ldo * 4);
  float xr[RPT], dy[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const long r = ry + RGRP * i;
    xr[i] = s_ld1(rx, (unsigned)((r * a.ldx + c) * 4) | cmask);
    dy[i] = s_ld1(rd, (unsigned)((r * a.lddo + c) * 4) | cmask);     // rows >= B read as zero
  }
  if (a.post_act == ACT_RELU) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const float o = s_ld1(rout, (unsigned)(((long)(ry + RGRP * i) * a.ldo + c) * 4) | cmask);
      dy[i] = (o > 0.f) ? dy[i] * gate_scale : 0.f;
    }
  }
  float xh[RPT];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const float xa = (a.pre_act == ACT_LEAKY) ? (xr[i] > 0.f ? xr[i] : xr[i] * LEAKY_SLOPE) : xr[i];
    xh[i] = (xa - mean) * invstd;
    s1 += dy[i];                 // dy is exactly 0 for rows >= B
    s2 += dy[i] * xh[i];
  }
  const float sum_dy = col_reduce16(s1, red, cx, ry);
  const float sum_dy_xh = col_reduce16(s2, red, cx, ry);
  const float invB = 1.0f / (float)a.B;
  const __amdgpu_buffer_rsrc_t rdx = s_rsrc(a.dx, (long)a.B * a.lddx * 4);
  float sb = 0.f;
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const long r = ry + RGRP * i;
    float d = gm * invstd * (dy[i] - invB * sum_dy - xh[i] * invB * sum_dy_xh);
    if (a.pre_act == ACT_LEAKY) d = xr[i] > 0.f ? d : d * LEAKY_SLOPE;
    d = (r < a.B) ? d : 0.f;
    s_st1(d, rdx, (unsigned)((r * a.lddx + c) * 4) | cmask);
    sb += d;
  }
  const float sum_dx = col_reduce16(sb, red, cx, ry);
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

// slots[blockIdx.x] = sum over this block's 64 elements of (sum_z SX[z][i]) * (sum_z SD[z][i]).
// 256 threads = 64 elements x 4 slab groups, so the slab sums of one element are spread over 4 threads and every
// thread keeps several independent loads in flight.
__global__ __launch_bounds__(256) void fx_gram_hadamard_kernel(double* __restrict__ slots, const float* __restrict__ sx,
                                                               int nsx, const float* __restrict__ sd, int nsd, long n) {
  __shared__ float pa[4][64], pb[4][64];
  __shared__ double sm[16];
  const int e = threadIdx.x & 63, zg = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + e;
  float a = 0.f, b = 0.f;
  if (i < n) {
    for (int z = zg; z < nsx; z += 4) a += sx[(long)z * n + i];
    for (int z = zg; z < nsd; z += 4) b += sd[(long)z * n + i];
  }
  pa[zg][e] = a;
  pb[zg][e] = b;
  __syncthreads();
  double acc = 0.0;
  if (zg == 0) {
    const float ta = (pa[0][e] + pa[1][e]) + (pa[2][e] + pa[3][e]);
    const float tb = (pb[0][e] + pb[1][e]) + (pb[2][e] + pb[3][e]);
    acc = (double)ta * (double)tb;
  }
  acc = fx_block_sum_d(acc, sm);
  if (threadIdx.x == 0) slots[blockIdx.x] = acc;
}

// One 32 (batch rows) x 128 (features) tile per workgroup: gather the cohort rows, emit
//   x [R,F] fp32 (optional), hi/lo K-blocked [Fp/32][ldo rows][32] bf16, hiT/loT [F, ldt] bf16 (transposed through LDS).
// 16-byte loads of the source rows (when the row pitch allows), 16-byte stores of x, 8-byte stores of the K-blocked split,
// 16-byte stores of the transposed split (the first version moved 4 bytes in and 2 bytes out per lane: 156-278 us per
// modality at cfg4's 384 x 30000 batch, where three of them run beside the backward chain).
struct GatherSplit {
  float* x; __bf16* hi; __bf16* lo; __bf16* hiT; __bf16* loT;
  const float* src; const long* idx; const float* ctrl; long cursor_stride;
  int R, F; long ld_src, ldx, ldo, ldt;
  int vec;                 // 1: F % 4 == 0, ld_src % 4 == 0, ldx % 4 == 0 and 16-byte aligned bases -> float4 accesses
};
#define GS_COLS 128
typedef float gs_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 gs_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void gs_body(const GatherSplit& a) {
  __shared__ float tile[32][GS_COLS + 1];
  const long* idx = a.idx;
  if (a.ctrl) idx += (long)a.ctrl[FXC_BATCH_CURSOR] * a.cursor_stride;
  const int c0 = blockIdx.x * GS_COLS, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 lanes x 4 columns = 128 columns, 8 rows per pass
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = ty + 8 * i, r = r0 + rl, c = c0 + 4 * tx;
    const long s = idx[min(r, a.R - 1)];
    gs_f4 v;
    if (a.vec && c + 3 < a.F) {
      v = *reinterpret_cast<const gs_f4*>(a.src + s * a.ld_src + c);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (c + j < a.F) ? a.src[s * a.ld_src + c + j] : 0.f;
    }
    if (r >= a.R) v = gs_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[rl][4 * tx + j] = v[j];
    if (r < a.R) {
      if (a.x) {
        if (a.vec && c + 3 < a.F) *reinterpret_cast<gs_f4*>(a.x + (long)r * a.ldx + c) = v;
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (c + j < a.F) a.x[(long)r * a.ldx + c + j] = v[j];
        }
      }
      // K-blocked [Fp/32][ldo rows][32]: columns c .. c+3 lie in K-block c / 32 at offset c % 32; columns F..Fp-1 are zero
      if (c < ((a.F + 31) / 32) * 32) {
        gs_bf16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = (__bf16)v[j]; l[j] = (__bf16)(v[j] - (float)h[j]); }
        const long o = ((long)(c >> 5) * a.ldo + r) * 32 + (c & 31);
        *reinterpret_cast<gs_bf16x4*>(a.hi + o) = h;
        *reinterpret_cast<gs_bf16x4*>(a.lo + o) = l;
      }
    }
  }
  __syncthreads();
  // transposed: thread = (column t >> 1, 16 rows (t & 1) * 16 ..): two 16-byte stores per array
  {
    const int cl = threadIdx.x >> 1, rb = (threadIdx.x & 1) * 16, c = c0 + cl;
    if (c < a.F) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        gs_bf16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = tile[rb + 8 * q + j][cl];
          h[j] = (__bf16)v;
          l[j] = (__bf16)(v - (float)h[j]);
        }
        const long o = (long)c * a.ldt + r0 + rb + 8 * q;     // r0 + 32 <= Rp <= ldt (the grid covers exactly Rp rows)
        *reinterpret_cast<gs_bf16x8*>(a.hiT + o) = h;
        *reinterpret_cast<gs_bf16x8*>(a.loT + o) = l;
      }
    }
  }
}

__global__ __launch_bounds__(256) void fx_gather_split_kernel(GatherSplit a) { gs_body(a); }

// every modality of a batch in one launch: grid (column tiles of the widest, row blocks, modalities)
#define GS_MAX_GROUP 4
struct GatherSplitGroup {
  GatherSplit a[GS_MAX_GROUP];
};
__global__ __launch_bounds__(256) void fx_gather_split_group_kernel(GatherSplitGroup g) {
  // (a runtime index into the kernel-argument array would put the argument blocks in scratch: one call per constant index)
  switch (blockIdx.z) {
    case 0: if ((int)blockIdx.x * GS_COLS < ((g.a[0].F + 31) / 32) * 32) gs_body(g.a[0]); break;
    case 1: if ((int)blockIdx.x * GS_COLS < ((g.a[1].F + 31) / 32) * 32) gs_body(g.a[1]); break;
    case 2: if ((int)blockIdx.x * GS_COLS < ((g.a[2].F + 31) / 32) * 32) gs_body(g.a[2]); break;
    default: if ((int)blockIdx.x * GS_COLS < ((g.a[3].F + 31) / 32) * 32) gs_body(g.a[3]); break;
  }
}

// ---- host launchers used by the extern "C" entry points in fx_norm_act.hip ---------------------------
int fx_launch_bn_fwd_r16(const BnFwd16& a, hipStream_t stream) {
  hipLaunchKernelGGL(fx_bn_fwd_r16_kernel, dim3((a.C + COLS - 1) / COLS), dim3(256), 0, stream, a);
  return fx_check_launch("fx_bn_act_fwd(r16)");
}
int fx_launch_bn_bwd_r16(const BnBwd16& a, hipStream_t stream) {
  hipLaunchKernelGGL(fx_bn_bwd_r16_kernel, dim3((a.C + COLS - 1) / COLS), dim3(256), 0, stream, a);
  return fx_check_launch("fx_bn_act_bwd(r16)");
}

extern "C" {

// BatchNorm forward fed by the split-K slabs of the preceding Linear (x = sum_z slabs[z] + lin_bias is also
// written to x_out for the backward pass).  B <= 128.
int fx_bn_act_fwd_slabs(float* out, float* x_out, const float* slabs, int nslabs, long slab_stride, const float* lin_bias,
                        const float* gamma, const float* beta, float* running_mean, float* running_var, float* save_mean,
                        float* save_invstd, const float* mask, int B, int C, long ldx, long ldo, int pre_act, int post_act,
                        int train, float drop_p, unsigned long long seed, unsigned long long offset, const float* ctrl,
                        hipStream_t stream) {
  FX_REQUIRE(out && slabs && nslabs > 0 && gamma && beta && running_mean && running_var, "fx_bn_act_fwd_slabs: null pointer");
  FX_REQUIRE(B > 0 && B <= RGRP * RPT && C > 0, "fx_bn_act_fwd_slabs: B=%d must be in 1..%d", B, RGRP * RPT);
  FX_REQUIRE(!train || (save_mean && save_invstd && B > 1), "fx_bn_act_fwd_slabs: train mode needs save buffers and B > 1");
  BnFwd16 a{out, x_out, nullptr, slabs, nslabs, slab_stride, lin_bias, gamma, beta, running_mean, running_var, save_mean,
            save_invstd, mask, B, C, ldx, ldo, pre_act, post_act, train, drop_p, seed, offset, ctrl};
  return fx_launch_bn_fwd_r16(a, stream);
}

int fx_gram_hadamard_blocks(long n) { return (int)((n + 63) / 64); }

// writes fx_gram_hadamard_blocks(n) partial sums into slots[0..)
int fx_gram_hadamard(double* slots, const float* slabs_x, int nslabs_x, const float* slabs_d, int nslabs_d, long n,
                     hipStream_t stream) {
  FX_REQUIRE(slots && slabs_x && slabs_d && nslabs_x > 0 && nslabs_d > 0 && n > 0, "fx_gram_hadamard: bad args");
  hipLaunchKernelGGL(fx_gram_hadamard_kernel, dim3(fx_gram_hadamard_blocks(n)), dim3(256), 0, stream, slots, slabs_x,
                     nslabs_x, slabs_d, nslabs_d, n);
  return fx_check_launch("fx_gram_hadamard");
}

static int gs_fill(GatherSplit& a, float* x, void* hi, void* lo, void* hiT, void* loT, const float* src, const long* idx,
                   int n_rows, int n_cols, long ld_src, long ldx, long ldo, long ldt, const float* ctrl_cursor, long cursor_stride) {
  FX_REQUIRE(hi && lo && hiT && loT && src && idx && n_rows > 0 && n_cols > 0, "fx_gather_split: bad args");
  const int Rp = (n_rows + 31) / 32 * 32;
  FX_REQUIRE(ldo >= n_rows && ldo % 128 == 0 && ldt >= Rp,
             "fx_gather_split: hi/lo are K-blocked with rows padded to 128 (got %ld), hiT/loT need ld >= %d (got %ld)", ldo, Rp, ldt);
  FX_REQUIRE(ldt % 8 == 0 && (((uintptr_t)hiT | (uintptr_t)loT | (uintptr_t)hi | (uintptr_t)lo) & 15) == 0,
             "fx_gather_split: split outputs must be 16-byte aligned with ldt %% 8 == 0 (got %ld)", ldt);
  const int vec = (n_cols % 4 == 0) && (ld_src % 4 == 0) && (!x || ldx % 4 == 0) && ((((uintptr_t)src) | ((uintptr_t)x)) & 15) == 0;
  a = GatherSplit{x, (__bf16*)hi, (__bf16*)lo, (__bf16*)hiT, (__bf16*)loT, src, idx, ctrl_cursor, cursor_stride,
                  n_rows, n_cols, ld_src, ldx, ldo, ldt, vec};
  return 0;
}

int fx_gather_split(float* x, void* hi, void* lo, void* hiT, void* loT, const float* src, const long* idx, int n_rows,
                    int n_cols, long ld_src, long ldx, long ldo, long ldt, const float* ctrl_cursor, long cursor_stride,
                    hipStream_t stream) {
  GatherSplit a;
  if (int rc = gs_fill(a, x, hi, lo, hiT, loT, src, idx, n_rows, n_cols, ld_src, ldx, ldo, ldt, ctrl_cursor, cursor_stride)) return rc;
  const int Rp = (n_rows + 31) / 32 * 32, Fp = (n_cols + 31) / 32 * 32;
  hipLaunchKernelGGL(fx_gather_split_kernel, dim3((Fp + GS_COLS - 1) / GS_COLS, Rp / 32), dim3(256), 0, stream, a);
  return fx_check_launch("fx_gather_split");
}

struct fx_gather_split_desc_ {   // include/fxhip.h: fx_gather_split_desc
  float* x; void* hi; void* lo; void* hiT; void* loT; const float* src; int n_cols; long ld_src, ldx, ldo, ldt;
};

// fx_gather_split for up to 4 cohort layers in one launch: the same index table, cursor and row count for all of them.
int fx_gather_split_group(const void* descs_, int n, const long* idx, int n_rows, const float* ctrl_cursor, long cursor_stride,
                          hipStream_t stream) {
  const fx_gather_split_desc_* d = (const fx_gather_split_desc_*)descs_;
  FX_REQUIRE(d && n > 0 && n <= GS_MAX_GROUP, "fx_gather_split_group: 1..%d layers per launch", GS_MAX_GROUP);
  GatherSplitGroup g{};
  int max_tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (int rc = gs_fill(g.a[i], d[i].x, d[i].hi, d[i].lo, d[i].hiT, d[i].loT, d[i].src, idx, n_rows, d[i].n_cols, d[i].ld_src,
                         d[i].ldx, d[i].ldo, d[i].ldt, ctrl_cursor, cursor_stride))
      return rc;
    const int tiles = ((d[i].n_cols + 31) / 32 * 32 + GS_COLS - 1) / GS_COLS;
    max_tiles = tiles > max_tiles ? tiles : max_tiles;
  }
  for (int i = n; i < GS_MAX_GROUP; ++i) g.a[i] = g.a[0];
  const int Rp = (n_rows + 31) / 32 * 32;
  hipLaunchKernelGGL(fx_gather_split_group_kernel, dim3(max_tiles, Rp / 32, n), dim3(256), 0, stream, g);
  return fx_check_launch("fx_gather_split_group");
}

}  // extern "C"
