// Graph-convolution encoder kernels (reference flexGCN, modules.py:153-262; GNN model, models/gnn_early.py:103-158).
//
// Activations are [B, nodes, C] fp32, node-major inside a sample and contiguous, i.e. exactly the memory the
// reference's `x.view(-1, C)` (BatchNorm over batch*nodes rows) and `x.view(B, -1)` (flatten before fc) see; C is the
// node embedding width (4..32 in the reference's search space, config.py:45) or the node feature count (1..3).
// One graph is shared by every sample (gnn_early.py:93-96), so message passing is an SpMM with a tiny dense side:
//   fx_spmm_rows      out[b, i, :] = sum_{e in row i} w[e] * x[b, idx[e], :]        (CSR by target: forward;
//                                                                                    CSR by source: backward)
//   fx_rowlin2        out[r, :] = a[r, :] Wa^T (+ b[r, :] Wb^T) (+ bias)            (the convs' Linear layers; `trans`
//                                                                                    applies W instead of W^T: dX)
//   fx_rowlin_wgrad   dW = dY^T X, db = colsum(dY) over all rows                     (deterministic two-stage sum)
//   fx_bn_rows_*      BatchNorm1d over R = B*nodes rows + activation + Dropout, forward and backward
// Lane layout of every kernel: lane = row_sub * CP + channel with CP = the power of two >= C, so a wavefront reads
// 64/CP consecutive rows as one contiguous span and per-channel sums are shuffle reductions over the row_sub bits.
// The gathered operand of one sample ([nodes, C], <= 1.3 MB) is L2-resident: workgroups of one sample are placed on
// one XCD (blockIdx -> sample mapping below), so HBM traffic stays at one read of x and one write of out while the
// E*B*C gathered values come out of that XCD's L2.
#include "fx_common.h"

#define GN_T 256
enum { GACT_RELU = 0, GACT_SIGMOID = 1, GACT_LEAKY = 2, GACT_TANH = 3, GACT_GELU = 4 };  // flexGCN act_options
#define GN_LEAKY 0.01f  // nn.LeakyReLU() default slope (modules.py:213)

namespace {

__device__ __forceinline__ int cp_of(int C) {  // smallest power of two >= C (C <= 32)
  int p = 1;
  while (p < C) p <<= 1;
  return p;
}
inline int h_cp_of(int C) {
  int p = 1;
  while (p < C) p <<= 1;
  return p;
}

// sum over the row_sub bits of the lane id (lanes that differ only above log2(CP))
__device__ __forceinline__ float sum_over_rowsub(float v, int CP) {
  for (int off = 32; off >= CP; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- SpMM with a narrow dense side -------------------------------------------------------------------------------
// One wavefront walks a tile of `npw` nodes of one sample, NG nodes at a time.  A lane owns VEC consecutive channels
// (one 8/16-byte load when C % 2/4 == 0) of one edge slot; the 64 / CP edge slots (CP = power of two >= C / VEC) are
// split evenly over the NG nodes, two edges per slot are in flight, and a node's slots are combined with shuffles.
// Several nodes per wavefront matter because most nodes have few edges: with the whole wavefront on one node the
// kernel was a chain of per-node latencies (row pointers -> indices -> gather -> reduce -> store).
template <int VEC>
__global__ __launch_bounds__(GN_T) void fx_spmm_rows_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                            const int* __restrict__ rowptr, const int* __restrict__ idx,
                                                            const float* __restrict__ w, int B, int nodes, int C,
                                                            int tiles, int npw, int NG, int S) {
  // XCD-aware placement: consecutive block ids go round-robin over the 8 XCDs, so ids congruent mod 8 share an L2;
  // give each XCD whole samples.
  const int id = blockIdx.x;
  const int j = id >> 3, xcd = id & 7;
  // inside an XCD the samples are taken S at a time and interleaved tile by tile: a tile's slice of the edge list
  // (indices + weights, 8 B per edge) is then reused by S samples while their operands (S x nodes x C floats <= ~2 MB)
  // share the L2, instead of the whole edge list being re-streamed for every sample (PMC: 132 -> ~60 MB per launch)
  const int chunk = j / (tiles * S), rem = j - chunk * (tiles * S);
  const int tile = rem / S;
  const int b = xcd + 8 * (chunk * S + rem % S);
  if (b >= B) return;
  const int CQ = C / VEC, CP = cp_of(CQ), EG = 64 / CP, SPN = EG / NG;    // SPN edge slots per node
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cq = lane & (CP - 1), slot = lane / CP;
  const int ng = slot / SPN, es = slot - ng * SPN;
  const bool cok = cq < CQ;
  const int coff = (cok ? cq : 0) * VEC;
  const float* xb = x + (long)b * nodes * C + coff;
  float* ob = out + (long)b * nodes * C + coff;
  const int n0 = (tile * (GN_T / 64) + wave) * npw;
  for (int it = 0; it < npw; it += NG) {
    const int i = n0 + it + ng;
    const bool iok = i < nodes;
    const int ic = iok ? i : nodes - 1;
    const int e0 = rowptr[ic], e1 = iok ? rowptr[ic + 1] : e0;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int e = e0 + es; e < e1; e += 2 * SPN) {
      const int e2 = e + SPN;
      const bool two = e2 < e1;
      const int sa = idx[e], sb = idx[two ? e2 : e];
      const float wa = w[e], wb = two ? w[e2] : 0.f;
      float va[VEC], vb[VEC];
      if (VEC == 4) {
        const float4 ta = *(const float4*)(xb + (long)sa * C), tb = *(const float4*)(xb + (long)sb * C);
        va[0] = ta.x; va[1 % VEC] = ta.y; va[2 % VEC] = ta.z; va[3 % VEC] = ta.w;
        vb[0] = tb.x; vb[1 % VEC] = tb.y; vb[2 % VEC] = tb.z; vb[3 % VEC] = tb.w;
      } else if (VEC == 2) {
        const float2 ta = *(const float2*)(xb + (long)sa * C), tb = *(const float2*)(xb + (long)sb * C);
        va[0] = ta.x; va[1 % VEC] = ta.y;
        vb[0] = tb.x; vb[1 % VEC] = tb.y;
      } else {
        va[0] = xb[(long)sa * C];
        vb[0] = xb[(long)sb * C];
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] += wa * va[v] + wb * vb[v];
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float t = cok ? acc[v] : 0.f;
      for (int off = (CP * SPN) >> 1; off >= CP; off >>= 1) t += __shfl_xor(t, off, 64);
      acc[v] = t;
    }
    if (es == 0 && cok && iok) {
      if (VEC == 4) {
        *(float4*)(ob + (long)i * C) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
      } else if (VEC == 2) {
        *(float2*)(ob + (long)i * C) = make_float2(acc[0], acc[1 % VEC]);
      } else {
        ob[(long)i * C] = acc[0];
      }
    }
  }
}

// ---- two-input row-wise Linear ----------------------------------------------------------------------------------
// Thread per row; the zero padded weights sit in LDS and are read as broadcasts.  CI / CO = channel counts rounded up
// to 4, 8, 16 or 32, so the loops unroll and the accumulators stay in registers.
#define GN_CMAX 32
#define GN_WLD (GN_CMAX + 4)
template <int CI, int CO>
__global__ __launch_bounds__(GN_T) void fx_rowlin2_kernel(float* __restrict__ out, const float* __restrict__ a,
                                                          const float* __restrict__ Wa, const float* __restrict__ bsrc,
                                                          const float* __restrict__ Wb, const float* __restrict__ bias,
                                                          long R, int Ca, int Cb, int Cout, int trans, int accumulate) {
  __shared__ __attribute__((aligned(16))) float sWa[CO][GN_WLD], sWb[CO][GN_WLD];
  __shared__ float sbias[CO];
  for (int t = threadIdx.x; t < CO * CI; t += GN_T) {
    const int o = t / CI, c = t % CI;
    // trans == 0: W is [Cout, Cin] (nn.Linear layout), out = in W^T; trans == 1: W is [Cin, Cout], out = in W
    float va = 0.f, vb = 0.f;
    if (o < Cout && c < Ca) va = trans ? Wa[(long)c * Cout + o] : Wa[(long)o * Ca + c];
    if (bsrc && o < Cout && c < Cb) vb = trans ? Wb[(long)c * Cout + o] : Wb[(long)o * Cb + c];
    sWa[o][c] = va;
    sWb[o][c] = vb;
  }
  if (threadIdx.x < CO) sbias[threadIdx.x] = (bias && threadIdx.x < Cout) ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  const long r = (long)blockIdx.x * GN_T + threadIdx.x;
  if (r >= R) return;
  // padded channels re-read the last valid one (no conditional loads); their weights are zero
  float av[CI], bv[CI];
  if ((Ca & 3) == 0 && ((((uintptr_t)a) & 15) == 0)) {
#pragma unroll
    for (int c = 0; c < CI; c += 4) {
      const float4 t = *reinterpret_cast<const float4*>(a + r * Ca + (c < Ca ? c : Ca - 4));
      av[c] = t.x; av[c + 1] = t.y; av[c + 2] = t.z; av[c + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int c = 0; c < CI; ++c) av[c] = a[r * Ca + (c < Ca ? c : Ca - 1)];
  }
  if (bsrc && (Cb & 3) == 0 && ((((uintptr_t)bsrc) & 15) == 0)) {
#pragma unroll
    for (int c = 0; c < CI; c += 4) {
      const float4 t = *reinterpret_cast<const float4*>(bsrc + r * Cb + (c < Cb ? c : Cb - 4));
      bv[c] = t.x; bv[c + 1] = t.y; bv[c + 2] = t.z; bv[c + 3] = t.w;
    }
  } else if (bsrc) {
#pragma unroll
    for (int c = 0; c < CI; ++c) bv[c] = bsrc[r * Cb + (c < Cb ? c : Cb - 1)];
  } else {
#pragma unroll
    for (int c = 0; c < CI; ++c) bv[c] = 0.f;
  }
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    float s = sbias[o];
#pragma unroll
    for (int c = 0; c < CI; ++c) s += av[c] * sWa[o][c];
    if (bsrc) {
#pragma unroll
      for (int c = 0; c < CI; ++c) s += bv[c] * sWb[o][c];
    }
    acc[o] = s;
  }
  if ((Cout & 3) == 0 && ((((uintptr_t)out) & 15) == 0)) {     // 16-byte stores: a row is written as Cout / 4 full segments
#pragma unroll
    for (int o = 0; o < CO; o += 4) {
      if (o < Cout) {
        float4* dst = reinterpret_cast<float4*>(out + r * Cout + o);
        float4 v = make_float4(acc[o], acc[o + 1], acc[o + 2], acc[o + 3]);
        if (accumulate) {
          const float4 p = *dst;
          v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        *dst = v;
      }
    }
  } else {
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      if (o < Cout) {
        float v = acc[o];
        if (accumulate) v += out[r * Cout + o];
        out[r * Cout + o] = v;
      }
    }
  }
}

// ---- weight gradient of a row-wise Linear: dW[o][c] = sum_r dy[r][o] x[r][c], db[o] = sum_r dy[r][o] ---------------
// Stage 1: each workgroup reduces its row chunk into partial[blockIdx][Cout*(Cin+1)] (column Cin = the bias gradient);
// stage 2 adds the partials in a fixed order.  Tiles of 64 rows are staged in LDS.  A thread owns a 2 (out) x 4 (in)
// block of dW for one of RG row groups (RG = 256 / number of blocks, a power of two): per row it reads dy twice
// (broadcast) and x once (16 bytes) for 8 FMAs.  The row groups are summed in group order at the end.
#define GN_WG_ROWS 64
#define GN_XLD (GN_CMAX + 4)
// global [nr, C] (contiguous) -> LDS rows of stride ld, channels >= C and rows >= nr zeroed
__device__ __forceinline__ void stage_tile(float* __restrict__ dst, int ld, const float* __restrict__ src, int C, int nr,
                                           int tid) {
  const int total = GN_WG_ROWS * C, valid = nr * C;
  if ((C & 3) == 0 && ((((uintptr_t)src) & 15) == 0)) {
    float4 v[2];
    int e[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {          // 64 * 32 / 4 = 512 float4 at most: two per thread
      e[q] = 4 * (tid + q * GN_T);
      const int ec = e[q] < valid ? e[q] : 0;
      v[q] = *reinterpret_cast<const float4*>(src + ec);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (e[q] < total) {
        const bool ok = e[q] < valid;
        const int rr = e[q] / C, c = e[q] - rr * C;
        float* d = dst + rr * ld + c;
        d[0] = ok ? v[q].x : 0.f; d[1] = ok ? v[q].y : 0.f; d[2] = ok ? v[q].z : 0.f; d[3] = ok ? v[q].w : 0.f;
      }
    }
  } else {
    for (int e0 = tid; e0 < total; e0 += GN_T) {
      const int rr = e0 / C, c = e0 - rr * C;
      const float v = src[e0 < valid ? e0 : 0];
      dst[rr * ld + c] = e0 < valid ? v : 0.f;
    }
  }
}

__global__ __launch_bounds__(GN_T) void fx_rowlin_wgrad_kernel(float* __restrict__ partial, const float* __restrict__ dy,
                                                               const float* __restrict__ x, long R, int Cin, int Cout,
                                                               long rows_per_block) {
  __shared__ __attribute__((aligned(16))) float sx[GN_WG_ROWS][GN_XLD];
  __shared__ float sdy[GN_WG_ROWS][GN_CMAX + 2];
  __shared__ float comb[GN_T][10];                 // per thread: 8 dW entries + 2 bias entries
  const int tid = threadIdx.x;
  const int ob = (Cout + 1) >> 1, cb = (Cin + 3) >> 2, nb = ob * cb;    // <= 16 * 8 = 128 blocks
  int RG = 2;
  while (RG * 2 * nb <= GN_T && RG < GN_WG_ROWS) RG <<= 1;
  const int rg = tid / nb, blk = tid - rg * nb;
  const bool active = rg < RG;
  const int o0 = 2 * (blk / cb), c4 = blk % cb, c0 = 4 * c4;
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float accb[2] = {0.f, 0.f};
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  for (long rt = r0; rt < r1; rt += GN_WG_ROWS) {
    const int nr = (int)((r1 - rt) < GN_WG_ROWS ? (r1 - rt) : GN_WG_ROWS);
    __syncthreads();
    // a tile is one contiguous block of nr * C floats in each array: copy it with unconditional (clamped) 16-byte
    // loads issued together, then scatter into the padded LDS rows; rows past nr are stored as zeros
    stage_tile(&sdy[0][0], GN_CMAX + 2, dy + rt * Cout, Cout, nr, tid);
    stage_tile(&sx[0][0], GN_XLD, x + rt * Cin, Cin, nr, tid);
    __syncthreads();
    if (active) {
      for (int rr = rg; rr < GN_WG_ROWS; rr += RG) {
        const float d0 = sdy[rr][o0], d1 = sdy[rr][o0 + 1];
        const float4 xv = *reinterpret_cast<const float4*>(&sx[rr][c0]);
        acc[0][0] += d0 * xv.x; acc[0][1] += d0 * xv.y; acc[0][2] += d0 * xv.z; acc[0][3] += d0 * xv.w;
        acc[1][0] += d1 * xv.x; acc[1][1] += d1 * xv.y; acc[1][2] += d1 * xv.z; acc[1][3] += d1 * xv.w;
        accb[0] += d0;
        accb[1] += d1;
      }
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) comb[tid][i * 4 + j] = acc[i][j];
      comb[tid][8 + i] = accb[i];
    }
  }
  __syncthreads();
  if (tid < nb) {      // row group 0's threads add the groups in order and write this block's partial
    const int ld = Cin + 1;
    float* pp = partial + (long)blockIdx.x * Cout * ld;
    float t8[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) t8[q] = 0.f;
    for (int g2 = 0; g2 < RG; ++g2) {
#pragma unroll
      for (int q = 0; q < 10; ++q) t8[q] += comb[g2 * nb + tid][q];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (o0 + i < Cout) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c0 + j < Cin) pp[(o0 + i) * ld + c0 + j] = t8[i * 4 + j];
        if (c4 == 0) pp[(o0 + i) * ld + Cin] = t8[8 + i];
      }
    }
  }
}

// 256 threads = 32 pairs x 8 slices of the block range; the 8 slice sums are added in slice order.
__global__ __launch_bounds__(GN_T) void fx_rowlin_wgrad_merge_kernel(float* __restrict__ dW, float* __restrict__ db,
                                                                     const float* __restrict__ partial, int blocks,
                                                                     int Cin, int Cout, int accumulate) {
  __shared__ float part[8][32];
  const int npairs = Cout * (Cin + 1);
  const int pl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int p = blockIdx.x * 32 + pl;
  float s = 0.f;
  if (p < npairs) {
#pragma unroll 4
    for (int k = sl; k < blocks; k += 8) s += partial[(long)k * npairs + p];
  }
  part[sl][pl] = s;
  __syncthreads();
  if (sl != 0 || p >= npairs) return;
  float t = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) t += part[q][pl];
  const int o = p / (Cin + 1), c = p % (Cin + 1);
  if (c < Cin) {
    if (dW) dW[o * Cin + c] = accumulate ? dW[o * Cin + c] + t : t;
  } else if (db) {
    db[o] = accumulate ? db[o] + t : t;
  }
}

// ---- BatchNorm over R rows of C <= 32 channels -----------------------------------------------------------------------
__device__ __forceinline__ float gact(float z, int act) {
  switch (act) {
    case GACT_RELU: return fmaxf(z, 0.f);
    case GACT_SIGMOID: return 1.f / (1.f + expf(-z));
    case GACT_LEAKY: return z >= 0.f ? z : z * GN_LEAKY;
    case GACT_TANH: return tanhf(z);
    default: return 0.5f * z * (1.f + erff(z * 0.70710678118654752440f));
  }
}
__device__ __forceinline__ float gact_grad(float z, int act) {
  switch (act) {
    case GACT_RELU: return z > 0.f ? 1.f : 0.f;
    case GACT_SIGMOID: { const float s = 1.f / (1.f + expf(-z)); return s * (1.f - s); }
    case GACT_LEAKY: return z >= 0.f ? 1.f : GN_LEAKY;   // torch: grad 1 for z > 0, slope otherwise; z == 0 has measure 0
    case GACT_TANH: { const float t = tanhf(z); return 1.f - t * t; }
    default: return 0.5f * (1.f + erff(z * 0.70710678118654752440f)) + z * 0.39894228040143267794f * expf(-0.5f * z * z);
  }
}

// Two per-channel sums over the rows of a chunk, as doubles: partial[blockIdx][2][C].
//   mode 0 (forward statistics): s0 = sum (x - shift), s1 = sum (x - shift)^2   with shift = x[0, c]
//   mode 1 (backward):           g = dA * mask/keep * act'(z);  s0 = sum g, s1 = sum g * xhat;  g is written over dA
struct BnRows {
  const float* x;        // [R, C] conv output (BatchNorm input)
  float* da;             // mode 1: [R, C] upstream gradient in, g out
  const float* gamma; const float* beta; const float* save_mean; const float* save_invstd;
  const float* mask;     // optional explicit dropout mask [R, C]
  const float* ctrl;
  double* partial;
  long R; int C; int act; int mode; float drop_p;
  unsigned long long seed, offset;
  long rows_per_block;
};

__device__ __forceinline__ unsigned long long gn_step_offset(const float* ctrl, unsigned long long offset) {
  return ctrl ? offset + (((unsigned long long)ctrl[FXC_STEP]) << 44) : offset;
}
__device__ __forceinline__ float gn_mask(const BnRows& a, unsigned long long rng_off, long e) {
  if (a.drop_p <= 0.f) return 1.f;
  const float keep_scale = 1.0f / (1.0f - a.drop_p);
  const float mk = a.mask ? a.mask[e] : (fx_rand_uniform(a.seed, rng_off, (unsigned long long)e) <= (1.0f - a.drop_p) ? 1.f : 0.f);
  return mk * keep_scale;
}

__global__ __launch_bounds__(GN_T) void fx_bn_rows_reduce_kernel(BnRows a) {
  __shared__ double red[GN_T / 64][2][GN_CMAX];
  const int C = a.C, CP = cp_of(C), RS = 64 / CP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & (CP - 1), rs = lane / CP;
  const bool cok = c < C;
  const long r0 = (long)blockIdx.x * a.rows_per_block;
  long r1 = r0 + a.rows_per_block;
  if (r1 > a.R) r1 = a.R;
  double s0 = 0.0, s1 = 0.0;
  if (cok) {
    if (a.mode == 0) {
      const float shift = a.x[c];
      const long S = (GN_T / 64) * RS;
      long r = r0 + wave * RS + rs;
      for (; r + 3 * S < r1; r += 4 * S) {          // four independent rows in flight per lane
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = a.x[(r + u * S) * C + c];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float d = v[u] - shift;
          s0 += (double)d;
          s1 += (double)d * (double)d;
        }
      }
      for (; r < r1; r += S) {
        const float d = a.x[r * C + c] - shift;
        s0 += (double)d;
        s1 += (double)d * (double)d;
      }
    } else {
      const float mu = a.save_mean[c], is = a.save_invstd[c], gm = a.gamma[c], bt = a.beta[c];
      const unsigned long long rng_off = gn_step_offset(a.ctrl, a.offset);
      const long S = (GN_T / 64) * RS;
      long r = r0 + wave * RS + rs;
      for (; r + 3 * S < r1; r += 4 * S) {
        float xv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          xv[u] = a.x[(r + u * S) * C + c];
          dv[u] = a.da[(r + u * S) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long e = (r + u * S) * C + c;
          const float xhat = (xv[u] - mu) * is;
          const float g = dv[u] * gn_mask(a, rng_off, e) * gact_grad(xhat * gm + bt, a.act);
          a.da[e] = g;
          s0 += (double)g;
          s1 += (double)g * (double)xhat;
        }
      }
      for (; r < r1; r += S) {
        const long e = r * C + c;
        const float xhat = (a.x[e] - mu) * is;
        const float g = a.da[e] * gn_mask(a, rng_off, e) * gact_grad(xhat * gm + bt, a.act);
        a.da[e] = g;
        s0 += (double)g;
        s1 += (double)g * (double)xhat;
      }
    }
  }
  for (int off = 32; off >= CP; off >>= 1) {
    s0 += __shfl_xor(s0, off, 64);
    s1 += __shfl_xor(s1, off, 64);
  }
  if (rs == 0 && cok) {
    red[wave][0][c] = s0;
    red[wave][1][c] = s1;
  }
  __syncthreads();
  if (threadIdx.x < 2 * C) {
    const int k = threadIdx.x / C, cc = threadIdx.x % C;
    double t = 0.0;
    for (int wv = 0; wv < GN_T / 64; ++wv) t += red[wv][k][cc];
    a.partial[((long)blockIdx.x * 2 + k) * C + cc] = t;
  }
}

// Sum the per-block partials [blocks][2][C]: 256 threads = 32 channels x 8 slices of the block range, slice sums
// added in slice order (fixed summation order -> bit-reproducible).
__device__ __forceinline__ void bn_rows_sum_partials(const double* __restrict__ partial, int blocks, int C, double& s0,
                                                     double& s1) {
  __shared__ double part[2][8][32];
  const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
  double a0 = 0.0, a1 = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int k = sl; k < blocks; k += 8) {
      a0 += partial[((long)k * 2) * C + c];
      a1 += partial[((long)k * 2 + 1) * C + c];
    }
  }
  part[0][sl][c] = a0;
  part[1][sl][c] = a1;
  __syncthreads();
  s0 = 0.0;
  s1 = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    s0 += part[0][q][c];
    s1 += part[1][q][c];
  }
}

// forward finalize: batch statistics, running statistics (momentum 0.1, unbiased running_var)
__global__ __launch_bounds__(GN_T) void fx_bn_rows_stats_finalize_kernel(const double* __restrict__ partial, int blocks,
                                                                         const float* __restrict__ x, long R, int C,
                                                                         float* __restrict__ save_mean,
                                                                         float* __restrict__ save_invstd,
                                                                         float* __restrict__ running_mean,
                                                                         float* __restrict__ running_var) {
  double s0, s1;
  bn_rows_sum_partials(partial, blocks, C, s0, s1);
  const int c = threadIdx.x;
  if (c >= C) return;
  const double n = (double)R;
  const double dm = s0 / n;
  const double mean = (double)x[c] + dm;
  double var_b = s1 / n - dm * dm;
  if (var_b < 0.0) var_b = 0.0;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0 / sqrt(var_b + (double)FX_BN_EPS));
  const double var_u = R > 1 ? var_b * (n / (n - 1.0)) : var_b;
  running_mean[c] = (1.0f - FX_BN_MOMENTUM) * running_mean[c] + FX_BN_MOMENTUM * (float)mean;
  running_var[c] = (1.0f - FX_BN_MOMENTUM) * running_var[c] + FX_BN_MOMENTUM * (float)var_u;
}

// backward finalize: dgamma = sum g*xhat, dbeta = sum g
__global__ __launch_bounds__(GN_T) void fx_bn_rows_bwd_finalize_kernel(const double* __restrict__ partial, int blocks, int C,
                                                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                       float* __restrict__ sums /* [2][C] as floats */) {
  double s0, s1;
  bn_rows_sum_partials(partial, blocks, C, s0, s1);
  const int c = threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] = (float)s0;
  if (dgamma) dgamma[c] = (float)s1;
  sums[c] = (float)s0;
  sums[C + c] = (float)s1;
}

// forward apply: out = dropout(act(gamma * xhat + beta));  train == 0 uses the running statistics.  Same row chunks
// and lane layout as the reduction (no per-element modulo).
__global__ __launch_bounds__(GN_T) void fx_bn_rows_apply_kernel(float* __restrict__ out, BnRows a,
                                                                const float* __restrict__ running_mean,
                                                                const float* __restrict__ running_var, int train) {
  const int C = a.C, CP = cp_of(C), RS = 64 / CP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & (CP - 1), rs = lane / CP;
  if (c >= C) return;
  const long r0 = (long)blockIdx.x * a.rows_per_block;
  long r1 = r0 + a.rows_per_block;
  if (r1 > a.R) r1 = a.R;
  const float mu = train ? a.save_mean[c] : running_mean[c];
  const float is = train ? a.save_invstd[c] : 1.0f / sqrtf(running_var[c] + FX_BN_EPS);
  const float gm = a.gamma[c], bt = a.beta[c];
  const unsigned long long rng_off = gn_step_offset(a.ctrl, a.offset);
  const long S = (GN_T / 64) * RS;
  long r = r0 + wave * RS + rs;
  for (; r + 3 * S < r1; r += 4 * S) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = a.x[(r + u * S) * C + c];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long e = (r + u * S) * C + c;
      float y = gact((v[u] - mu) * is * gm + bt, a.act);
      if (train) y = y * gn_mask(a, rng_off, e);
      out[e] = y;
    }
  }
  for (; r < r1; r += S) {
    const long e = r * C + c;
    float y = gact((a.x[e] - mu) * is * gm + bt, a.act);
    if (train) y = y * gn_mask(a, rng_off, e);
    out[e] = y;
  }
}

// backward apply (in place on g): dx = gamma * invstd * (g - sum_g / R - xhat * sum_gxhat / R)
__global__ __launch_bounds__(GN_T) void fx_bn_rows_bwd_apply_kernel(float* __restrict__ g, const float* __restrict__ x,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ save_mean,
                                                                    const float* __restrict__ save_invstd,
                                                                    const float* __restrict__ sums, long R, int C,
                                                                    long rows_per_block) {
  const int CP = cp_of(C), RS = 64 / CP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & (CP - 1), rs = lane / CP;
  if (c >= C) return;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  const float inv_r = 1.0f / (float)R;
  const float is = save_invstd[c], mu = save_mean[c];
  const float k0 = gamma[c] * is, m0 = sums[c] * inv_r, m1 = sums[C + c] * inv_r;
  const long S = (GN_T / 64) * RS;
  long r = r0 + wave * RS + rs;
  for (; r + 3 * S < r1; r += 4 * S) {
    float xv[4], gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xv[u] = x[(r + u * S) * C + c];
      gv[u] = g[(r + u * S) * C + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) g[(r + u * S) * C + c] = k0 * (gv[u] - m0 - (xv[u] - mu) * is * m1);
  }
  for (; r < r1; r += S) {
    const long e = r * C + c;
    const float xhat = (x[e] - mu) * is;
    g[e] = k0 * (g[e] - m0 - xhat * m1);
  }
}

// ---- four channels per lane (C % 4 == 0): 16-byte accesses, and one Philox block serves the lane's four elements
// (element e takes output e & 3 of block e >> 2 in both layouts, so the masks are identical to the scalar kernels').
__device__ __forceinline__ float4 gn_mask4(const BnRows& a, unsigned long long rng_off, long e0) {
  if (a.drop_p <= 0.f) return make_float4(1.f, 1.f, 1.f, 1.f);
  const float ks = 1.0f / (1.0f - a.drop_p), keep = 1.0f - a.drop_p;
  if (a.mask) {
    const float4 m = *reinterpret_cast<const float4*>(a.mask + e0);
    return make_float4(m.x * ks, m.y * ks, m.z * ks, m.w * ks);
  }
  uint32_t r[4];
  fx_philox4(a.seed, rng_off + ((unsigned long long)e0 >> 2), r);
  return make_float4(fx_u01(r[0]) <= keep ? ks : 0.f, fx_u01(r[1]) <= keep ? ks : 0.f, fx_u01(r[2]) <= keep ? ks : 0.f,
                     fx_u01(r[3]) <= keep ? ks : 0.f);
}

__global__ __launch_bounds__(GN_T) void fx_bn_rows_reduce4_kernel(BnRows a) {
  __shared__ double red[GN_T / 64][2][GN_CMAX];
  const int C = a.C, CQ = C >> 2, CP = cp_of(CQ), RS = 64 / CP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cq = lane & (CP - 1), rs = lane / CP, c0 = 4 * cq;
  const bool cok = cq < CQ;
  const long r0 = (long)blockIdx.x * a.rows_per_block;
  long r1 = r0 + a.rows_per_block;
  if (r1 > a.R) r1 = a.R;
  double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
  if (cok) {
    const long S = (GN_T / 64) * RS;
    long r = r0 + wave * RS + rs;
    if (a.mode == 0) {
      const float4 sh = *reinterpret_cast<const float4*>(a.x + c0);
      const float shv[4] = {sh.x, sh.y, sh.z, sh.w};
      for (; r + S < r1; r += 2 * S) {
        const float4 va = *reinterpret_cast<const float4*>(a.x + r * C + c0);
        const float4 vb = *reinterpret_cast<const float4*>(a.x + (r + S) * C + c0);
        const float xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float da_ = xa[j] - shv[j], db_ = xb[j] - shv[j];
          s0[j] += (double)da_ + (double)db_;
          s1[j] += (double)da_ * (double)da_ + (double)db_ * (double)db_;
        }
      }
      for (; r < r1; r += S) {
        const float4 va = *reinterpret_cast<const float4*>(a.x + r * C + c0);
        const float xa[4] = {va.x, va.y, va.z, va.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = xa[j] - shv[j];
          s0[j] += (double)d;
          s1[j] += (double)d * (double)d;
        }
      }
    } else {
      const float4 mu4 = *reinterpret_cast<const float4*>(a.save_mean + c0), is4 = *reinterpret_cast<const float4*>(a.save_invstd + c0);
      const float4 gm4 = *reinterpret_cast<const float4*>(a.gamma + c0), bt4 = *reinterpret_cast<const float4*>(a.beta + c0);
      const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
      const float gm[4] = {gm4.x, gm4.y, gm4.z, gm4.w}, bt[4] = {bt4.x, bt4.y, bt4.z, bt4.w};
      const unsigned long long rng_off = gn_step_offset(a.ctrl, a.offset);
      for (; r < r1; r += S) {
        const long e0 = r * C + c0;
        const float4 xv4 = *reinterpret_cast<const float4*>(a.x + e0);
        const float4 dv4 = *reinterpret_cast<const float4*>(a.da + e0);
        const float4 mk4 = gn_mask4(a, rng_off, e0);
        const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w}, dv[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
        const float mk[4] = {mk4.x, mk4.y, mk4.z, mk4.w};
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xhat = (xv[j] - mu[j]) * is[j];
          g[j] = dv[j] * mk[j] * gact_grad(xhat * gm[j] + bt[j], a.act);
          s0[j] += (double)g[j];
          s1[j] += (double)g[j] * (double)xhat;
        }
        *reinterpret_cast<float4*>(a.da + e0) = make_float4(g[0], g[1], g[2], g[3]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    for (int off = 32; off >= CP; off >>= 1) {
      s0[j] += __shfl_xor(s0[j], off, 64);
      s1[j] += __shfl_xor(s1[j], off, 64);
    }
  }
  if (rs == 0 && cok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[wave][0][c0 + j] = s0[j];
      red[wave][1][c0 + j] = s1[j];
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * C) {
    const int k = threadIdx.x / C, cc = threadIdx.x % C;
    double t = 0.0;
    for (int wv = 0; wv < GN_T / 64; ++wv) t += red[wv][k][cc];
    a.partial[((long)blockIdx.x * 2 + k) * C + cc] = t;
  }
}

__global__ __launch_bounds__(GN_T) void fx_bn_rows_apply4_kernel(float* __restrict__ out, BnRows a,
                                                                 const float* __restrict__ running_mean,
                                                                 const float* __restrict__ running_var, int train) {
  const int C = a.C, CQ = C >> 2, CP = cp_of(CQ), RS = 64 / CP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cq = lane & (CP - 1), rs = lane / CP, c0 = 4 * cq;
  if (cq >= CQ) return;
  const long r0 = (long)blockIdx.x * a.rows_per_block;
  long r1 = r0 + a.rows_per_block;
  if (r1 > a.R) r1 = a.R;
  float mu[4], is[4], gm[4], bt[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = train ? a.save_mean[c0 + j] : running_mean[c0 + j];
    is[j] = train ? a.save_invstd[c0 + j] : 1.0f / sqrtf(running_var[c0 + j] + FX_BN_EPS);
    gm[j] = a.gamma[c0 + j];
    bt[j] = a.beta[c0 + j];
  }
  const unsigned long long rng_off = gn_step_offset(a.ctrl, a.offset);
  const long S = (GN_T / 64) * RS;
  long r = r0 + wave * RS + rs;
  for (; r + S < r1; r += 2 * S) {
    const long ea = r * C + c0, eb = (r + S) * C + c0;
    const float4 va = *reinterpret_cast<const float4*>(a.x + ea), vb = *reinterpret_cast<const float4*>(a.x + eb);
    float4 ma = make_float4(1.f, 1.f, 1.f, 1.f), mb = ma;
    if (train) { ma = gn_mask4(a, rng_off, ea); mb = gn_mask4(a, rng_off, eb); }
    const float xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w};
    const float ka[4] = {ma.x, ma.y, ma.z, ma.w}, kb[4] = {mb.x, mb.y, mb.z, mb.w};
    float ya[4], yb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ya[j] = gact((xa[j] - mu[j]) * is[j] * gm[j] + bt[j], a.act) * ka[j];
      yb[j] = gact((xb[j] - mu[j]) * is[j] * gm[j] + bt[j], a.act) * kb[j];
    }
    *reinterpret_cast<float4*>(out + ea) = make_float4(ya[0], ya[1], ya[2], ya[3]);
    *reinterpret_cast<float4*>(out + eb) = make_float4(yb[0], yb[1], yb[2], yb[3]);
  }
  for (; r < r1; r += S) {
    const long ea = r * C + c0;
    const float4 va = *reinterpret_cast<const float4*>(a.x + ea);
    float4 ma = make_float4(1.f, 1.f, 1.f, 1.f);
    if (train) ma = gn_mask4(a, rng_off, ea);
    const float xa[4] = {va.x, va.y, va.z, va.w}, ka[4] = {ma.x, ma.y, ma.z, ma.w};
    float ya[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ya[j] = gact((xa[j] - mu[j]) * is[j] * gm[j] + bt[j], a.act) * ka[j];
    *reinterpret_cast<float4*>(out + ea) = make_float4(ya[0], ya[1], ya[2], ya[3]);
  }
}

__global__ __launch_bounds__(GN_T) void fx_bn_rows_bwd_apply4_kernel(float* __restrict__ g, const float* __restrict__ x,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ save_mean,
                                                                     const float* __restrict__ save_invstd,
                                                                     const float* __restrict__ sums, long R, int C,
                                                                     long rows_per_block) {
  const int CQ = C >> 2, CP = cp_of(CQ), RS = 64 / CP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cq = lane & (CP - 1), rs = lane / CP, c0 = 4 * cq;
  if (cq >= CQ) return;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
  const float inv_r = 1.0f / (float)R;
  float mu[4], is[4], k0[4], m0[4], m1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    is[j] = save_invstd[c0 + j];
    mu[j] = save_mean[c0 + j];
    k0[j] = gamma[c0 + j] * is[j];
    m0[j] = sums[c0 + j] * inv_r;
    m1[j] = sums[C + c0 + j] * inv_r;
  }
  const long S = (GN_T / 64) * RS;
  long r = r0 + wave * RS + rs;
  for (; r + S < r1; r += 2 * S) {
    const long ea = r * C + c0, eb = (r + S) * C + c0;
    const float4 xa4 = *reinterpret_cast<const float4*>(x + ea), xb4 = *reinterpret_cast<const float4*>(x + eb);
    const float4 ga4 = *reinterpret_cast<const float4*>(g + ea), gb4 = *reinterpret_cast<const float4*>(g + eb);
    const float xa[4] = {xa4.x, xa4.y, xa4.z, xa4.w}, xb[4] = {xb4.x, xb4.y, xb4.z, xb4.w};
    const float ga[4] = {ga4.x, ga4.y, ga4.z, ga4.w}, gb[4] = {gb4.x, gb4.y, gb4.z, gb4.w};
    float oa[4], ob[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      oa[j] = k0[j] * (ga[j] - m0[j] - (xa[j] - mu[j]) * is[j] * m1[j]);
      ob[j] = k0[j] * (gb[j] - m0[j] - (xb[j] - mu[j]) * is[j] * m1[j]);
    }
    *reinterpret_cast<float4*>(g + ea) = make_float4(oa[0], oa[1], oa[2], oa[3]);
    *reinterpret_cast<float4*>(g + eb) = make_float4(ob[0], ob[1], ob[2], ob[3]);
  }
  for (; r < r1; r += S) {
    const long ea = r * C + c0;
    const float4 xa4 = *reinterpret_cast<const float4*>(x + ea), ga4 = *reinterpret_cast<const float4*>(g + ea);
    const float xa[4] = {xa4.x, xa4.y, xa4.z, xa4.w}, ga[4] = {ga4.x, ga4.y, ga4.z, ga4.w};
    float oa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) oa[j] = k0[j] * (ga[j] - m0[j] - (xa[j] - mu[j]) * is[j] * m1[j]);
    *reinterpret_cast<float4*>(g + ea) = make_float4(oa[0], oa[1], oa[2], oa[3]);
  }
}

inline int row_blocks(long R, long* rows_per_block) {
  long b = (R + 255) / 256;
  if (b > 512) b = 512;        // 2 workgroups per CU; the partial sums are merged by one small workgroup
  if (b < 1) b = 1;
  long rpb = (R + b - 1) / b;
  rpb = ((rpb + 63) / 64) * 64;  // whole 64-row tiles
  *rows_per_block = rpb;
  return (int)((R + rpb - 1) / rpb);
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }   // NULL counts as aligned
inline int row_blocks_n(long R) {
  long rpb;
  return row_blocks(R, &rpb);
}

}  // namespace

extern "C" {

int fx_gnn_row_blocks(long R) {
  long rpb;
  return row_blocks(R, &rpb);
}

int fx_spmm_rows(float* out, const float* x, const int* rowptr, const int* idx, const float* w, int B, int nodes, int C,
                 long n_edges, hipStream_t stream) {
  FX_REQUIRE(out && x && rowptr && idx && w, "fx_spmm_rows: null pointer");
  FX_REQUIRE(B > 0 && nodes > 0 && C > 0 && C <= GN_CMAX, "fx_spmm_rows: need 1 <= C <= 32 (B=%d nodes=%d C=%d)", B, nodes, C);
  FX_REQUIRE(out != x, "fx_spmm_rows: not in place");
  const int waves = GN_T / 64;
  int npw = 8;  // nodes per wavefront
  const int tiles = (nodes + waves * npw - 1) / (waves * npw);
  const int bgroups = (B + 7) / 8;
  const bool v4 = (C % 4 == 0) && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)out) & 15) == 0);
  const bool v2 = (C % 2 == 0) && ((((uintptr_t)x) & 7) == 0) && ((((uintptr_t)out) & 7) == 0);
  const int vec = v4 ? 4 : (v2 ? 2 : 1);
  const int eg = 64 / h_cp_of(C / vec);
  // Nodes in flight per wavefront: about 6 edges per edge slot and node keeps both the hubs' tail short and the
  // per-node latency chain amortised (measured at 8000 nodes: 50 edges/node -> 2 nodes of 8 slots, 10 -> 4 of 4).
  int ng = 2;
  if (n_edges > 0) {
    const double per_node = (double)n_edges / nodes;
    int spn = 1;
    while (spn * 2 <= eg && spn * 6 < per_node) spn *= 2;
    ng = eg / spn;
  }
  if (ng > eg) ng = eg;
  if (ng > npw) ng = npw;
  if (ng < 1) ng = 1;
  // samples of one XCD interleaved per tile, as many as keep their operands within ~2 MB of the 4 MB L2
  long S = (2L << 20) / ((long)nodes * C * 4);
  if (S < 1) S = 1;
  if (S > bgroups) S = bgroups;
  const int chunks = (bgroups + (int)S - 1) / (int)S;
  const dim3 grid(8 * chunks * (int)S * tiles);
  if (vec == 4)
    hipLaunchKernelGGL(fx_spmm_rows_kernel<4>, grid, dim3(GN_T), 0, stream, out, x, rowptr, idx, w, B, nodes, C, tiles, npw, ng,
                       (int)S);
  else if (vec == 2)
    hipLaunchKernelGGL(fx_spmm_rows_kernel<2>, grid, dim3(GN_T), 0, stream, out, x, rowptr, idx, w, B, nodes, C, tiles, npw, ng,
                       (int)S);
  else
    hipLaunchKernelGGL(fx_spmm_rows_kernel<1>, grid, dim3(GN_T), 0, stream, out, x, rowptr, idx, w, B, nodes, C, tiles, npw, ng,
                       (int)S);
  return fx_check_launch("fx_spmm_rows");
}

int fx_rowlin2(float* out, const float* a, const float* Wa, int Ca, const float* b, const float* Wb, int Cb,
               const float* bias, long R, int Cout, int trans, int accumulate, hipStream_t stream) {
  FX_REQUIRE(out && a && Wa, "fx_rowlin2: null pointer");
  FX_REQUIRE((b == nullptr) == (Wb == nullptr), "fx_rowlin2: second input and its weight go together");
  FX_REQUIRE(R > 0 && Ca > 0 && Ca <= GN_CMAX && Cout > 0 && Cout <= GN_CMAX && (!b || (Cb > 0 && Cb <= GN_CMAX)),
             "fx_rowlin2: channel counts must be in 1..32");
  const int cin = (b && Cb > Ca) ? Cb : Ca;
  const int ci = cin <= 4 ? 4 : (cin <= 8 ? 8 : (cin <= 16 ? 16 : 32));
  const int co = Cout <= 4 ? 4 : (Cout <= 8 ? 8 : (Cout <= 16 ? 16 : 32));
  const dim3 grid((unsigned)((R + GN_T - 1) / GN_T));
#define FX_RL(CI, CO)                                                                                             \
  hipLaunchKernelGGL((fx_rowlin2_kernel<CI, CO>), grid, dim3(GN_T), 0, stream, out, a, Wa, b, Wb, bias, R, Ca, Cb, Cout, \
                     trans, accumulate)
#define FX_RL_CO(CI)                                                                     \
  do {                                                                                   \
    if (co == 4) FX_RL(CI, 4); else if (co == 8) FX_RL(CI, 8); else if (co == 16) FX_RL(CI, 16); else FX_RL(CI, 32); \
  } while (0)
  if (ci == 4) FX_RL_CO(4); else if (ci == 8) FX_RL_CO(8); else if (ci == 16) FX_RL_CO(16); else FX_RL_CO(32);
#undef FX_RL_CO
#undef FX_RL
  return fx_check_launch("fx_rowlin2");
}

long fx_rowlin_wgrad_workspace_bytes(long R, int Cin, int Cout) {
  long rpb;
  const int blocks = row_blocks(R, &rpb);
  return (long)blocks * Cout * (Cin + 1) * sizeof(float) + 64;
}

int fx_rowlin_wgrad(float* dW, float* db, const float* dy, const float* x, long R, int Cin, int Cout, int accumulate,
                    void* ws, hipStream_t stream) {
  FX_REQUIRE(dy && x && ws && (dW || db), "fx_rowlin_wgrad: null pointer");
  FX_REQUIRE(R > 0 && Cin > 0 && Cin <= GN_CMAX && Cout > 0 && Cout <= GN_CMAX, "fx_rowlin_wgrad: channel counts must be in 1..32");
  long rpb;
  const int blocks = row_blocks(R, &rpb);
  float* partial = (float*)ws;
  hipLaunchKernelGGL(fx_rowlin_wgrad_kernel, dim3(blocks), dim3(GN_T), 0, stream, partial, dy, x, R, Cin, Cout, rpb);
  int rc = fx_check_launch("fx_rowlin_wgrad");
  if (rc) return rc;
  const int npairs = Cout * (Cin + 1);
  hipLaunchKernelGGL(fx_rowlin_wgrad_merge_kernel, dim3((npairs + 31) / 32), dim3(GN_T), 0, stream, dW, db, partial, blocks,
                     Cin, Cout, accumulate);
  return fx_check_launch("fx_rowlin_wgrad(merge)");
}

long fx_bn_rows_workspace_bytes(long R, int C) {
  long rpb;
  const int blocks = row_blocks(R, &rpb);
  return (long)blocks * 2 * C * sizeof(double) + 2 * C * sizeof(float) + 64;
}

// BatchNorm1d over the R rows of x [R, C] -> activation -> Dropout(drop_p), flexGCN.forward (modules.py:253-256).
int fx_bn_rows_fwd(float* out, const float* x, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float* save_mean, float* save_invstd, const float* mask, long R, int C, int act,
                   int train, float drop_p, unsigned long long seed, unsigned long long offset, const float* ctrl, void* ws,
                   hipStream_t stream) {
  FX_REQUIRE(out && x && gamma && beta && running_mean && running_var, "fx_bn_rows_fwd: null pointer");
  FX_REQUIRE(R > 0 && C > 0 && C <= GN_CMAX, "fx_bn_rows_fwd: need 1 <= C <= 32");
  FX_REQUIRE(act >= 0 && act <= GACT_GELU, "fx_bn_rows_fwd: unknown activation %d", act);
  FX_REQUIRE(!train || (save_mean && save_invstd && ws), "fx_bn_rows_fwd: training needs save_mean/save_invstd/workspace");
  const bool v4 = (C % 4 == 0) && al16(out) && al16(x) && al16(gamma) && al16(beta) && al16(save_mean) && al16(save_invstd) &&
                  al16(mask);
  BnRows a{};
  a.x = x; a.gamma = gamma; a.beta = beta; a.save_mean = save_mean; a.save_invstd = save_invstd; a.mask = mask; a.ctrl = ctrl;
  a.R = R; a.C = C; a.act = act; a.mode = 0; a.drop_p = train ? drop_p : 0.f; a.seed = seed; a.offset = offset;
  if (train) {
    long rpb;
    const int blocks = row_blocks(R, &rpb);
    a.partial = (double*)ws;
    a.rows_per_block = rpb;
    if (v4) hipLaunchKernelGGL(fx_bn_rows_reduce4_kernel, dim3(blocks), dim3(GN_T), 0, stream, a);
    else hipLaunchKernelGGL(fx_bn_rows_reduce_kernel, dim3(blocks), dim3(GN_T), 0, stream, a);
    int rc = fx_check_launch("fx_bn_rows_fwd(stats)");
    if (rc) return rc;
    hipLaunchKernelGGL(fx_bn_rows_stats_finalize_kernel, dim3(1), dim3(GN_T), 0, stream, a.partial, blocks, x, R, C, save_mean,
                       save_invstd, running_mean, running_var);
    rc = fx_check_launch("fx_bn_rows_fwd(finalize)");
    if (rc) return rc;
  }
  long rpb_a;
  const int blocks_a = row_blocks(R, &rpb_a);
  a.rows_per_block = rpb_a;
  if (v4) hipLaunchKernelGGL(fx_bn_rows_apply4_kernel, dim3(blocks_a), dim3(GN_T), 0, stream, out, a, running_mean, running_var, train);
  else hipLaunchKernelGGL(fx_bn_rows_apply_kernel, dim3(blocks_a), dim3(GN_T), 0, stream, out, a, running_mean, running_var, train);
  return fx_check_launch("fx_bn_rows_fwd(apply)");
}

// Backward of the same block, in place: da [R, C] holds dL/d(out) on entry and dL/dx on return.
int fx_bn_rows_bwd(float* da, float* dgamma, float* dbeta, const float* x, const float* gamma, const float* beta,
                   const float* save_mean, const float* save_invstd, const float* mask, long R, int C, int act, float drop_p,
                   unsigned long long seed, unsigned long long offset, const float* ctrl, void* ws, hipStream_t stream) {
  FX_REQUIRE(da && x && gamma && beta && save_mean && save_invstd && ws, "fx_bn_rows_bwd: null pointer");
  FX_REQUIRE(R > 0 && C > 0 && C <= GN_CMAX, "fx_bn_rows_bwd: need 1 <= C <= 32");
  FX_REQUIRE(act >= 0 && act <= GACT_GELU, "fx_bn_rows_bwd: unknown activation %d", act);
  const bool v4 = (C % 4 == 0) && al16(da) && al16(x) && al16(gamma) && al16(beta) && al16(save_mean) && al16(save_invstd) &&
                  al16(mask) && al16((const char*)ws + (long)row_blocks_n(R) * 2 * C * sizeof(double));
  BnRows a{};
  a.x = x; a.da = da; a.gamma = gamma; a.beta = beta; a.save_mean = save_mean; a.save_invstd = save_invstd; a.mask = mask;
  a.ctrl = ctrl; a.R = R; a.C = C; a.act = act; a.mode = 1; a.drop_p = drop_p; a.seed = seed; a.offset = offset;
  long rpb;
  const int blocks = row_blocks(R, &rpb);
  a.partial = (double*)ws;
  a.rows_per_block = rpb;
  float* sums = (float*)((char*)ws + (long)blocks * 2 * C * sizeof(double));
  if (v4) hipLaunchKernelGGL(fx_bn_rows_reduce4_kernel, dim3(blocks), dim3(GN_T), 0, stream, a);
  else hipLaunchKernelGGL(fx_bn_rows_reduce_kernel, dim3(blocks), dim3(GN_T), 0, stream, a);
  int rc = fx_check_launch("fx_bn_rows_bwd(reduce)");
  if (rc) return rc;
  hipLaunchKernelGGL(fx_bn_rows_bwd_finalize_kernel, dim3(1), dim3(GN_T), 0, stream, a.partial, blocks, C, dgamma, dbeta, sums);
  rc = fx_check_launch("fx_bn_rows_bwd(finalize)");
  if (rc) return rc;
  if (v4)
    hipLaunchKernelGGL(fx_bn_rows_bwd_apply4_kernel, dim3(blocks), dim3(GN_T), 0, stream, da, x, gamma, save_mean, save_invstd,
                       sums, R, C, rpb);
  else
    hipLaunchKernelGGL(fx_bn_rows_bwd_apply_kernel, dim3(blocks), dim3(GN_T), 0, stream, da, x, gamma, save_mean, save_invstd,
                       sums, R, C, rpb);
  return fx_check_launch("fx_bn_rows_bwd(apply)");
}

}  // extern "C"
