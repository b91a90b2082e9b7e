// fx_enc_tail.hip -- the forward of an encoder's tail, every modality in ONE launch, and the fusion layer behind it.
//
// Reference ops replaced, per modality (MLP encoder, modules.py:145-149: layer_1 -> BatchNorm1d -> ReLU -> Dropout ->
// layer_out; VAE encoder, modules.py:25-41,47-56: hidden Linear -> LeakyReLU -> BatchNorm1d -> FC_mean, FC_var) and for the
// concatenation + fusion Linear behind them (direct_pred.py:118-124).  On the engine path the wide Linear's output arrives
// as the partial-sum slabs that the previous step's dW + Adam launch left behind (fx_dw_adam_fwd.hip), so the forward
// chain of a modality used to be four dependent launches on its own hipGraph branch --
//     fx_reduce_slabs (+ bias) -> fx_bn_act_fwd -> fx_gemm_f32 (layer_out, split-K) -> fx_reduce_slabs
// -- then a join and fx_small_linear_fwd for the fusion layer: ~72 us of launch latencies, branch skew and join gaps at
// cfg2 (profiles/r02_timeline_cfg2.txt) for ~1 MB of traffic.  Here:
//
//   fx_enc_tail_fwd   grid (column blocks, modalities).  One workgroup owns 64 columns of the block and all B <= 128
//                     rows: ordered slab sum + bias -> x (saved for the backward), two-pass batch statistics, affine,
//                     activation, dropout (Philox: one counter block per 4 consecutive columns, the stream of
//                     fx_bn_act_fwd) -> out (saved), and the block's share of every following small Linear:
//                     part_k[blk][r, l] = sum_{c in block} out[r, c] W_k[l, c]  (plain fp32 FMA from LDS tiles).
//   fx_fusion_fwd     grid (row blocks).  One workgroup owns 4 rows: ecat[r, :] = for every modality the ordered sum of
//                     its column blocks' partial products + layer_out bias (four threads share the slab range of an
//                     output and combine in a fixed order), then emb[r, :] = ecat[r, :] W_f^T + b_f from an LDS copy of
//                     W_f.  Without a fusion layer (one modality) it only reduces.
//
// Deterministic: fixed summation orders, no atomics.
#include "fx_common.h"
#include "fx_chain_prof.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define ET_COLS 64
#define ET_T 512
#define ET_MAXB 128
#define ET_MAXL 128
#define ET_MAX_GROUP 4
#define ET_ACT_NONE 0
#define ET_ACT_LEAKY 1
#define ET_ACT_RELU 2
#define ET_LEAKY 0.2f
#define ET_OOB 0xFFFFFFF0u

struct EncTailDesc {      // mirrors include/fxhip.h: fx_enc_tail_desc
  const float* slabs; long slab_stride; const float* lin_bias;
  float* x; float* out;
  const float* gamma; const float* beta; float* running_mean; float* running_var; float* save_mean; float* save_invstd;
  const float* mask;
  const float* W[2]; float* part[2];
  unsigned long long seed, offset;
  int n_slabs, H, n_up, L[2];
};

struct EncTailArgs {
  EncTailDesc d[ET_MAX_GROUP];
  int n, B, pre_act, post_act, train;
  float drop_p;
  const float* ctrl;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t et_rsrc(const void* p, long bytes) {
  const unsigned n = bytes > 0xFFFFFFF0L ? 0xFFFFFFF0u : (bytes < 0 ? 0u : (unsigned)bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);
}
__device__ __forceinline__ f32x4 et_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
#ifndef ET_ST_AUX
#define ET_ST_AUX 0          // cache policy bits of the x / out / partial-product stores (A/B: 16 = sc1 write-through)
#endif
__device__ __forceinline__ void et_st4(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned off) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, ET_ST_AUX);
}

// Sum over the 32 row groups of a column: lanes of a wave hold (cq = lane & 15, row group bits 0-1 = lane >> 4), the 8 waves
// the row group bits 2-4.  Shuffles inside the wave, then 8 partials through LDS added in wave order by every thread.
__device__ __forceinline__ f32x4 et_colsum(f32x4 v, float (*red)[ET_COLS + 4], int w, int cq) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] += __shfl_xor(v[j], 16, 64);
    v[j] += __shfl_xor(v[j], 32, 64);
  }
  __syncthreads();
  if ((threadIdx.x & 63) < 16) *reinterpret_cast<f32x4*>(&red[w][4 * cq]) = v;
  __syncthreads();
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < ET_T / 64; ++i) {
    const f32x4 p = *reinterpret_cast<const f32x4*>(&red[i][4 * cq]);
    s += p;
  }
  return s;
}

__device__ __forceinline__ void et_body(const EncTailDesc& d, const EncTailArgs& a, int blk) {
  // LDS: As [32][132] one 32-column half of the block output, k-major (column, row); Ws [32][132] the matching 32 columns
  // of a following Linear's weight, k-major (column, output)
  __shared__ __attribute__((aligned(16))) float As[32][ET_MAXB + 4];
  __shared__ __attribute__((aligned(16))) float Ws[32][ET_MAXL + 4];
  __shared__ __attribute__((aligned(16))) float red[ET_T / 64][ET_COLS + 4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int cq = t & 15, rg = t >> 4;                 // columns 4 cq .. 4 cq + 3 of the block, rows rg + 32 i
  const int cpw = blockIdx.y * gridDim.x + blockIdx.x;
  CP_TICK(cpw, 0);
  const int B = a.B, H = d.H;
  const int c = blk * ET_COLS + 4 * cq;               // H % 4 == 0: a unit of 4 columns is inside or outside as a whole
  const unsigned cmask = (c < H) ? 0u : ET_OOB;
  const bool cok = c < H;
  const long xbytes = (long)B * H * 4;
  const __amdgpu_buffer_rsrc_t rx = et_rsrc(d.x, xbytes), ro = et_rsrc(d.out, xbytes);
  f32x4 xv[4];
  unsigned off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) off[i] = (unsigned)(((long)(rg + 32 * i) * H + c) * 4) | cmask;   // rows >= B lie beyond the range
  // ---- every read whose address is known at entry is REQUESTED at entry (round 6).  The kernel sits on the step's critical path with
  // one workgroup per CU and nothing else to hide a memory round trip behind; the first version waited for the Linear bias, then for
  // the slabs two at a time, then -- behind the statistics' barriers -- for the running statistics, the affine pair and the step
  // counter: eight dependent round trips of 1.5-2.5 us in a 32 us launch.  Same values, same summation order; only the issue order moved.
  // (range-checked buffer loads with an out-of-range offset instead of predicated loads: a predicated load gets an exec-mask region
  // of its own with an s_waitcnt behind it)
  const long hbytes = (long)H * 4;
  const unsigned coff = (unsigned)(c * 4) | cmask;
  const f32x4 bv = et_ld4(et_rsrc((d.slabs && d.lin_bias) ? d.lin_bias : d.x, (d.slabs && d.lin_bias) ? hbytes : 0), coff);
  const f32x4 gm = et_ld4(et_rsrc(d.gamma, hbytes), coff), bt = et_ld4(et_rsrc(d.beta, hbytes), coff);
  const f32x4 rm_in = et_ld4(et_rsrc(d.running_mean, hbytes), coff), rv_in = et_ld4(et_rsrc(d.running_var, hbytes), coff);
  // (the step counter as a VECTOR load: a scalar load would be waited for on the spot, in front of everything below)
  const float stepf = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(et_rsrc(a.ctrl ? (const void*)a.ctrl : (const void*)d.x,
                                                                                     a.ctrl ? FXC_SIZE * 4 : 0), FXC_STEP * 4, 0, 0));
  if (d.slabs) {
    if (d.n_slabs <= 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[i] = bv;
    }
    // slab order = summation order of fx_reduce_slabs; up to eight slabs (all four row loads of each) are in flight together
    for (int z0 = 0; z0 < d.n_slabs; z0 += 8) {
      f32x4 u[8][4];
#pragma unroll
      for (int zi = 0; zi < 8; ++zi) {
        const bool have = z0 + zi < d.n_slabs;               // (an absent slab reads nothing: zero-length range)
        const __amdgpu_buffer_rsrc_t rz = et_rsrc(d.slabs + (have ? (long)(z0 + zi) * d.slab_stride : 0), have ? xbytes : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) u[zi][i] = et_ld4(rz, off[i]);
      }
      if (z0 == 0) {                                         // (behind the requests: the bias is the first thing waited for)
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = bv;
      }
#pragma unroll
      for (int zi = 0; zi < 8; ++zi)
        if (z0 + zi < d.n_slabs) {
#pragma unroll
          for (int i = 0; i < 4; ++i) xv[i] += u[zi][i];
        }
    }
    // x (saved for the backward) is stored HERE only when the pre-activation is about to overwrite it; otherwise with the block
    // output below: the memory counter retires in order, so the statistics' first wait for anything younger (the weight columns
    // requested next) would sit out these four stores' acknowledgements
    if (a.pre_act != ET_ACT_NONE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) et_st4(xv[i], rx, off[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) xv[i] = et_ld4(rx, off[i]);
  }
  const bool x_late = d.slabs && a.pre_act == ET_ACT_NONE;
  CP_TICK(cpw, 1);
  // ---- the first following Linear's weight columns are requested now (consumed after the statistics)
  // thread -> (output l = t >> 3 (+ 64), columns 8 * (t & 7) .. + 7 of the block): 2 x 16 bytes per output row
  f32x4 wv[2][2][2];                                   // [up][l half][column quad]
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const bool have = k < d.n_up;                      // (an absent Linear reads nothing: zero-length range)
    const int Lk = have ? d.L[k] : 0;
    const __amdgpu_buffer_rsrc_t rw = et_rsrc(have ? d.W[k] : d.x, (long)Lk * H * 4);
#pragma unroll
    for (int lh = 0; lh < 2; ++lh)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int l = (t >> 3) + 64 * lh, cc = blk * ET_COLS + 8 * (t & 7) + 4 * q;
        wv[k][lh][q] = et_ld4(rw, (unsigned)(((long)l * H + cc) * 4) | ((cc < H && l < Lk) ? 0u : ET_OOB));
      }
  }
  if (a.pre_act == ET_ACT_LEAKY) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[i][j] = xv[i][j] > 0.f ? xv[i][j] : xv[i][j] * ET_LEAKY;
  }
  // ---- batch statistics (two-pass; rows >= B hold zeros from the range check and are masked out of the sums)
  f32x4 mean, invstd;
  if (a.train) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (rg + 32 * i < B) s += xv[i];
    mean = et_colsum(s, red, w, cq) / (float)B;
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (rg + 32 * i < B) {
        const f32x4 dlt = xv[i] - mean;
        q += dlt * dlt;
      }
    const f32x4 var_b = et_colsum(q, red, w, cq) / (float)B;
#pragma unroll
    for (int j = 0; j < 4; ++j) invstd[j] = 1.0f / sqrtf(var_b[j] + FX_BN_EPS);
    if (cok && rg == 0) {
      *reinterpret_cast<f32x4*>(d.save_mean + c) = mean;
      *reinterpret_cast<f32x4*>(d.save_invstd + c) = invstd;
      const float unb = B > 1 ? (float)B / (float)(B - 1) : 1.0f;
      f32x4 rm = rm_in, rv = rv_in;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        rm[j] = (1.0f - FX_BN_MOMENTUM) * rm[j] + FX_BN_MOMENTUM * mean[j];
        rv[j] = (1.0f - FX_BN_MOMENTUM) * rv[j] + FX_BN_MOMENTUM * (var_b[j] * unb);
      }
      *reinterpret_cast<f32x4*>(d.running_mean + c) = rm;
      *reinterpret_cast<f32x4*>(d.running_var + c) = rv;
    }
  } else {
    mean = f32x4{0.f, 0.f, 0.f, 0.f};
    invstd = f32x4{0.f, 0.f, 0.f, 0.f};
    if (cok) {
      mean = rm_in;
#pragma unroll
      for (int j = 0; j < 4; ++j) invstd[j] = 1.0f / sqrtf(rv_in[j] + FX_BN_EPS);
    }
  }
  CP_TICK(cpw, 2);
  // ---- affine, activation, dropout (mask scaled first, then multiplied == F.dropout on CPU)
  const bool drop = a.train && a.drop_p > 0.f;
  const float keep = 1.0f - a.drop_p, keep_scale = 1.0f / (1.0f - a.drop_p);
  const unsigned long long rng_off = a.ctrl ? d.offset + (((unsigned long long)stepf) << 44) : d.offset;
  const __amdgpu_buffer_rsrc_t rmk = et_rsrc(d.mask, xbytes);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = rg + 32 * i;
    if (x_late) et_st4(xv[i], rx, off[i]);
    f32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      y[j] = (xv[i][j] - mean[j]) * invstd[j] * gm[j] + bt[j];
      if (a.post_act == ET_ACT_RELU) y[j] = fmaxf(y[j], 0.f);
    }
    if (drop) {
      f32x4 mk;
      if (d.mask) {
        mk = et_ld4(rmk, off[i]);
      } else {                 // element (r, c + j) is word j of counter block (r H + c) / 4: fx_rand_uniform's stream
        uint32_t rnd[4];
        fx_philox4(d.seed, rng_off + (((unsigned long long)r * H + c) >> 2), rnd);
#pragma unroll
        for (int j = 0; j < 4; ++j) mk[j] = fx_u01(rnd[j]) <= keep ? 1.f : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = y[j] * (mk[j] * keep_scale);
    }
    if (r >= B || !cok) y = f32x4{0.f, 0.f, 0.f, 0.f};
    et_st4(y, ro, off[i]);
    xv[i] = y;                                          // the block output replaces x in registers
  }
  // ---- the block's share of the following Linears: part_k[blk][r, l] = sum_c out[r, c] W_k[l, c]
  // micro-tile (rows 4 rt .. + 3) x (outputs 4 lt .. + 3): for L <= 64 every thread has one (rt = t & 31, lt = t >> 5),
  // for L <= 128 two (lt and lt + 16)
  const int rt = t & 31, lt = t >> 5;
  for (int k = 0; k < d.n_up; ++k) {
    const int Lk = d.L[k];
    f32x4 acc[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();                                  // the previous users of As / Ws are done
      if ((cq >> 3) == half) {                          // this thread's 4 columns lie in this half
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) As[4 * (cq & 7) + j][rg + 32 * i] = xv[i][j];
      }
      // W_k columns of this half: the thread holds columns 8 (t & 7) .. + 7 -> half (t & 7) >> 2
      if (((t & 7) >> 2) == half) {
#pragma unroll
        for (int lh = 0; lh < 2; ++lh)
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) Ws[8 * (t & 3) + 4 * q + j][(t >> 3) + 64 * lh] = k == 0 ? wv[0][lh][q][j] : wv[1][lh][q][j];
      }
      __syncthreads();
      if (4 * lt < Lk) {
#pragma unroll 8
        for (int cc = 0; cc < 32; ++cc) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(&As[cc][4 * rt]);
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(&Ws[cc][4 * lt]);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[0][i] += av[i] * w0;
          if (Lk > 64) {
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(&Ws[cc][4 * lt + 64]);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[1][i] += av[i] * w1;
          }
        }
      }
    }
    float* pk = d.part[k] + (long)blk * B * Lk;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int l0 = 4 * lt + 64 * u;
      if (l0 < Lk)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * rt + i;
          if (r < B) {
            if (l0 + 3 < Lk && (Lk & 3) == 0) {
              *reinterpret_cast<f32x4*>(pk + (long)r * Lk + l0) = acc[u][i];
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (l0 + j < Lk) pk[(long)r * Lk + l0 + j] = acc[u][i][j];
            }
          }
        }
    }
  }
  CP_TICK(cpw, 5);
}

__global__ __launch_bounds__(ET_T) void fx_enc_tail_fwd_kernel(EncTailArgs a) {
  const int blk = blockIdx.x;
  // (a runtime index into the kernel-argument array would put the descriptors in scratch: one call per constant index)
  switch (blockIdx.y) {
    case 0: if (blk * ET_COLS < a.d[0].H) et_body(a.d[0], a, blk); break;
    case 1: if (blk * ET_COLS < a.d[1].H) et_body(a.d[1], a, blk); break;
    case 2: if (blk * ET_COLS < a.d[2].H) et_body(a.d[2], a, blk); break;
    default: if (blk * ET_COLS < a.d[3].H) et_body(a.d[3], a, blk); break;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
#define FU_ROWS 4
#define FU_SUB 8            // threads (neighbouring lanes) that split one unit's slab range
#define FU_T 1024
#define FU_MAXK 512
#define FU_FLIGHT 10        // slab loads in flight per thread
#define FU_MAX_LAYERS 8

struct FusionArgs {
  float* emb; long ldemb;
  float* ecat; long ldecat;
  const float* part[FU_MAX_LAYERS]; const float* bias[FU_MAX_LAYERS]; int n_part[FU_MAX_LAYERS]; int width[FU_MAX_LAYERS];
  int col0[FU_MAX_LAYERS];       // first ecat column of the layer (widths are ANY integers: a layer may start at an unaligned column)
  int grp0[FU_MAX_LAYERS];       // first 4-column group of the layer in the unit index space (a part's row pitch = its width rounded up to 4)
  int n_groups;
  int n_layers;
  const float* Wf; const float* bf;
  int B, L, Kf;
};

#define FU_KC 64
__device__ __forceinline__ void fx_fusion_body(const FusionArgs& a, float (*es)[FU_MAXK + 4], float (*wfs)[ET_MAXL + 1]) {
  const int t = threadIdx.x, r0 = blockIdx.x * FU_ROWS;
  const int B = a.B, Kf = a.Kf, L = a.L;
  const int cpw = 512 + blockIdx.y * gridDim.x + blockIdx.x;      // (chain profile: the fusion launch's rows of the table)
  CP_TICK(cpw, 0);
  // ---- first W_f chunk: requested before the slab sums, stored after them.  element idx -> (l = idx / 64, k = idx % 64)
  float wreg[ET_MAXL * FU_KC / FU_T];
  if (a.Wf) {
#pragma unroll
    for (int i = 0; i < ET_MAXL * FU_KC / FU_T; ++i) {
      const int idx = t + FU_T * i, l = idx >> 6, k = idx & 63;
      wreg[i] = a.Wf[(long)min(l, L - 1) * Kf + min(k, Kf - 1)];
    }
  }
  float bfv = 0.f;                       // the fusion bias of this thread's output (t % L < L: valid for every thread)
  if (a.Wf && a.bf) bfv = a.bf[t % L];
  // ---- ecat rows: unit = (row, 4 consecutive columns of one modality); its FU_SUB threads (neighbouring lanes) split the slab
  // range (4 rows x 8 sub-ranges: 32 workgroups at B = 128, ~10 slabs of 16 bytes per thread in 2-3 dependent rounds)
  const int sub = t & (FU_SUB - 1), unit0 = t / FU_SUB;
  const int units = FU_ROWS * a.n_groups;
  for (int u = unit0; u < units; u += FU_T / FU_SUB) {
    const int rr = u / a.n_groups, gg = u - rr * a.n_groups;
    const int r = r0 + rr;
    const float* part = a.part[0]; const float* bias = a.bias[0]; int np = a.n_part[0], wd = a.width[0], c0 = a.col0[0], g0 = 0;
#pragma unroll
    for (int i = 1; i < FU_MAX_LAYERS; ++i)
      if (i < a.n_layers && gg >= a.grp0[i]) { part = a.part[i]; bias = a.bias[i]; np = a.n_part[i]; wd = a.width[i]; c0 = a.col0[i]; g0 = a.grp0[i]; }
    const int pc = (gg - g0) << 2, pw = (wd + 3) & ~3;         // column of the part, its row pitch
    const int c = c0 + pc;                                     // first of (up to) four ecat columns
    const int per = (np + FU_SUB - 1) / FU_SUB, z0 = min(np, sub * per), z1 = min(np, z0 + per);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    // the layer_out bias of the unit's columns is requested with the slabs (it used to be a round trip of its own behind the sum)
    // (unconditional, from an address that is valid either way: a predicated load is waited for on the spot;
    // bias arrays are allocated up to a multiple of 4: zeros)
    const f32x4 bq = *reinterpret_cast<const f32x4*>(bias ? bias + pc : part + pc);
    if (r < B && z0 < z1) {
      const float* src = part + (long)r * pw + pc;
      const long stride = (long)B * pw;
      // up to FU_FLIGHT slabs of 16 bytes in flight per thread (79 column blocks at cfg2 = 10 per thread: one round instead of
      // three dependent ones); addresses clamped into the thread's range, the adds predicated: same order, same sum
      for (int z = z0; z < z1; z += FU_FLIGHT) {
        f32x4 v[FU_FLIGHT];
#pragma unroll
        for (int j = 0; j < FU_FLIGHT; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (long)min(z + j, z1 - 1) * stride);
#pragma unroll
        for (int j = 0; j < FU_FLIGHT; ++j)
          if (z + j < z1) s += v[j];
      }
    }
    // combine the sub-ranges in range order (fixed: deterministic)
    const int base = (t & 63) & ~(FU_SUB - 1);
    f32x4 tot = s;
#pragma unroll
    for (int q = 1; q < FU_SUB; ++q) {
      f32x4 sq;
#pragma unroll
      for (int j = 0; j < 4; ++j) sq[j] = __shfl(s[j], base + q, 64);
      tot += sq;
    }
    if (sub == 0) {
      if (bias) tot += bq;
      if (((c | wd) & 3) == 0) {
        *reinterpret_cast<f32x4*>(&es[rr][c]) = tot;
        if (r < B && a.ecat) *reinterpret_cast<f32x4*>(a.ecat + (long)r * a.ldecat + c) = tot;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (pc + j < wd) {
            es[rr][c + j] = tot[j];
            if (r < B && a.ecat) a.ecat[(long)r * a.ldecat + c + j] = tot[j];
          }
      }
    }
  }
  CP_TICK(cpw, 1);
  if (!a.Wf) return;
  // ---- emb[r, l] = b_f[l] + sum_k ecat[r, k] W_f[l, k]: one output per thread, W_f in 64-column chunks through LDS
  const int rr = t / L, l = t - rr * L;
  const bool own = t < FU_ROWS * L;
  float acc = bfv;
  for (int k0 = 0; k0 < Kf; k0 += FU_KC) {
    __syncthreads();                       // es complete (first pass) / the previous chunk has been consumed
#pragma unroll
    for (int i = 0; i < ET_MAXL * FU_KC / FU_T; ++i) {
      const int idx = t + FU_T * i;
      wfs[idx & 63][idx >> 6] = wreg[i];
    }
    if (k0 + FU_KC < Kf) {                 // next chunk in flight while this one is multiplied
#pragma unroll
      for (int i = 0; i < ET_MAXL * FU_KC / FU_T; ++i) {
        const int idx = t + FU_T * i, ll = idx >> 6, k = k0 + FU_KC + (idx & 63);
        wreg[i] = a.Wf[(long)min(ll, L - 1) * Kf + min(k, Kf - 1)];
      }
    }
    __syncthreads();
    if (own) {
      const int kn = min(FU_KC, Kf - k0);
#pragma unroll 8
      for (int k = 0; k < kn; ++k) acc = fmaf(es[rr][k0 + k], wfs[k][l], acc);
    }
  }
  if (own && r0 + rr < B) a.emb[(long)(r0 + rr) * a.ldemb + l] = acc;
  CP_TICK(cpw, 2);
}

__global__ __launch_bounds__(FU_T) void fx_fusion_fwd_kernel(FusionArgs a) {
  __shared__ __attribute__((aligned(16))) float es[FU_ROWS][FU_MAXK + 4];      // this workgroup's rows of ecat
  __shared__ float wfs[FU_KC][ET_MAXL + 1];                                    // one 64-column chunk of W_f, transposed [k][l]
  fx_fusion_body(a, es, wfs);
}

// two independent fusion layers over the same rows in one launch (blockIdx.y): the VAE's FC_mean over mcat and FC_log_var over vcat
struct FusionPair { FusionArgs a[2]; };
__global__ __launch_bounds__(FU_T) void fx_fusion_fwd_pair_kernel(FusionPair p) {
  __shared__ __attribute__((aligned(16))) float es[FU_ROWS][FU_MAXK + 4];
  __shared__ float wfs[FU_KC][ET_MAXL + 1];
  fx_fusion_body(p.a[blockIdx.y], es, wfs);
}

CP_READER(fx_debug_chain_stamps_tail)

extern "C" {

struct fx_enc_tail_desc;   // include/fxhip.h; layout identical to EncTailDesc

int fx_enc_tail_blocks(int H) { return (H + ET_COLS - 1) / ET_COLS; }

int fx_enc_tail_fwd(const void* descs_, int n, int B, int pre_act, int post_act, int train, float drop_p, const float* ctrl,
                    hipStream_t stream) {
  const EncTailDesc* descs = (const EncTailDesc*)descs_;
  FX_REQUIRE(descs && n > 0 && n <= ET_MAX_GROUP, "fx_enc_tail_fwd: 1..%d modalities per launch", ET_MAX_GROUP);
  FX_REQUIRE(!(train && B == 1), "fx_enc_tail_fwd: BatchNorm1d training needs more than 1 value per channel");    // (torch's wording, as fx_bn_act_fwd)
  FX_REQUIRE(B > 0 && B <= ET_MAXB, "fx_enc_tail_fwd: B=%d must be in 1..%d", B, ET_MAXB);
  EncTailArgs a{};
  int max_blocks = 0;
  for (int i = 0; i < n; ++i) {
    const EncTailDesc& d = descs[i];
    FX_REQUIRE(d.x && d.out && d.gamma && d.beta && d.running_mean && d.running_var, "fx_enc_tail_fwd: null pointer (modality %d)", i);
    FX_REQUIRE(!train || (d.save_mean && d.save_invstd), "fx_enc_tail_fwd: train mode needs save_mean / save_invstd");
    FX_REQUIRE(d.H > 0 && d.H % 4 == 0 && (long)B * d.H * 4 < 0xF0000000L, "fx_enc_tail_fwd: block width %d must be a multiple of 4", d.H);
    FX_REQUIRE(!d.slabs || (d.n_slabs > 0 && d.slab_stride >= (long)B * d.H && d.slab_stride % 4 == 0),
               "fx_enc_tail_fwd: bad slab layout (modality %d)", i);
    FX_REQUIRE(d.n_up >= 0 && d.n_up <= 2, "fx_enc_tail_fwd: at most two following Linears");
    for (int k = 0; k < d.n_up; ++k)
      FX_REQUIRE(d.W[k] && d.part[k] && d.L[k] > 0 && d.L[k] <= ET_MAXL, "fx_enc_tail_fwd: following Linear %d: width %d outside 1..%d",
                 k, d.L[k], ET_MAXL);
    const uintptr_t al = (uintptr_t)d.x | (uintptr_t)d.out | (uintptr_t)d.slabs | (uintptr_t)d.lin_bias | (uintptr_t)d.gamma |
                         (uintptr_t)d.beta | (uintptr_t)d.running_mean | (uintptr_t)d.running_var | (uintptr_t)d.save_mean |
                         (uintptr_t)d.save_invstd | (uintptr_t)d.mask | (uintptr_t)d.W[0] | (uintptr_t)d.W[1] |
                         (uintptr_t)d.part[0] | (uintptr_t)d.part[1];
    FX_REQUIRE((al & 15) == 0, "fx_enc_tail_fwd: every array must be 16-byte aligned (modality %d)", i);
    a.d[i] = d;
    const int nb = fx_enc_tail_blocks(d.H);
    max_blocks = nb > max_blocks ? nb : max_blocks;
  }
  for (int i = n; i < ET_MAX_GROUP; ++i) a.d[i] = descs[0];
  a.n = n; a.B = B; a.pre_act = pre_act; a.post_act = post_act; a.train = train; a.drop_p = drop_p; a.ctrl = ctrl;
  hipLaunchKernelGGL(fx_enc_tail_fwd_kernel, dim3(max_blocks, n), dim3(ET_T), 0, stream, a);
  return fx_check_launch("fx_enc_tail_fwd");
}

static int fusion_args(FusionArgs& a, float* emb, long ldemb, float* ecat, long ldecat, const float* const* parts, const int* n_parts,
                       const float* const* part_bias, const int* widths, int n_layers, const float* Wf, const float* bf, int B, int L) {
  FX_REQUIRE(parts && n_parts && widths && n_layers > 0 && n_layers <= FU_MAX_LAYERS, "fx_fusion_fwd: 1..%d layers", FU_MAX_LAYERS);
  FX_REQUIRE(B > 0 && (ecat || Wf), "fx_fusion_fwd: nothing to write");
  int Kf = 0, ng = 0;
  for (int i = 0; i < n_layers; ++i) {
    FX_REQUIRE(parts[i] && n_parts[i] > 0 && widths[i] > 0 && (((uintptr_t)parts[i]) & 15) == 0,
               "fx_fusion_fwd: layer %d: width %d must be positive and the partial sums 16-byte aligned", i, widths[i]);
    a.part[i] = parts[i]; a.bias[i] = part_bias ? part_bias[i] : nullptr; a.n_part[i] = n_parts[i]; a.width[i] = widths[i];
    FX_REQUIRE(!a.bias[i] || (((uintptr_t)a.bias[i]) & 15) == 0, "fx_fusion_fwd: layer %d: bias must be 16-byte aligned", i);
    a.col0[i] = Kf;
    a.grp0[i] = ng;
    Kf += widths[i];
    ng += (widths[i] + 3) / 4;
  }
  a.n_groups = ng;
  FX_REQUIRE(Kf <= FU_MAXK, "fx_fusion_fwd: concatenated width %d exceeds %d", Kf, FU_MAXK);
  FX_REQUIRE(!ecat || (ldecat >= Kf && (((Kf | ldecat) & 3) != 0 || (((uintptr_t)ecat) & 15) == 0)), "fx_fusion_fwd: ecat: ld >= %d, 16-byte aligned when the widths are multiples of 4", Kf);
  FX_REQUIRE(!Wf || (emb && L > 0 && ldemb >= L), "fx_fusion_fwd: the fusion layer needs emb");
  FX_REQUIRE(!Wf || L <= ET_MAXL, "fx_fusion_fwd: latent width %d exceeds %d", L, ET_MAXL);
  a.emb = emb; a.ldemb = ldemb; a.ecat = ecat; a.ldecat = ldecat; a.n_layers = n_layers; a.Wf = Wf; a.bf = bf;
  a.B = B; a.L = L; a.Kf = Kf;
  return 0;
}

int fx_fusion_fwd(float* emb, long ldemb, float* ecat, long ldecat, const float* const* parts, const int* n_parts,
                  const float* const* part_bias, const int* widths, int n_layers, const float* Wf, const float* bf, int B, int L,
                  hipStream_t stream) {
  FusionArgs a{};
  const int rc = fusion_args(a, emb, ldemb, ecat, ldecat, parts, n_parts, part_bias, widths, n_layers, Wf, bf, B, L);
  if (rc) return rc;
  hipLaunchKernelGGL(fx_fusion_fwd_kernel, dim3((B + FU_ROWS - 1) / FU_ROWS), dim3(FU_T), 0, stream, a);
  return fx_check_launch("fx_fusion_fwd");
}

// Two fusion layers over the same B rows in ONE launch: every per-layer argument of fx_fusion_fwd doubled (index 0 / 1; parts, n_parts
// and part_bias hold the first layer's n_layers entries, then the second's).  The VAE's mean = FC_mean(mcat) and log_var =
// FC_log_var(vcat) (reference supervised_vae.py:172-176).
int fx_fusion_fwd_pair(float* const* emb, const long* ldemb, float* const* ecat, const long* ldecat, const float* const* parts,
                       const int* n_parts, const float* const* part_bias, const int* widths, int n_layers, const float* const* Wf,
                       const float* const* bf, int B, int L, hipStream_t stream) {
  FX_REQUIRE(emb && ldemb && ecat && ldecat && Wf && bf, "fx_fusion_fwd_pair: null argument arrays");
  FusionPair p{};
  for (int j = 0; j < 2; ++j) {
    const int rc = fusion_args(p.a[j], emb[j], ldemb[j], ecat[j], ldecat[j], parts ? parts + j * n_layers : nullptr,
                               n_parts ? n_parts + j * n_layers : nullptr, part_bias ? part_bias + j * n_layers : nullptr, widths, n_layers,
                               Wf[j], bf[j], B, L);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(fx_fusion_fwd_pair_kernel, dim3((B + FU_ROWS - 1) / FU_ROWS, 2), dim3(FU_T), 0, stream, p);
  return fx_check_launch("fx_fusion_fwd_pair");
}

}  // extern "C"
