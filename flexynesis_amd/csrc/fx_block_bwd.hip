// fx_block_bwd.hip -- the whole backward of an encoder's "wide Linear -> BatchNorm block -> small Linear(s)" tail in
// ONE launch.
//
// Reference autograd ops replaced (per modality): mm (dX of the small Linears following the block), the BatchNorm /
// ReLU / Dropout (or LeakyReLU) backward, mm + sum (weight and bias gradients of the small Linears), and -- for the
// engine's fused optimiser -- the pieces the wide layer's dW+clip+Adam kernel needs: the transposed split-bf16 dY
// operand and this layer's contribution to the global gradient norm.  These were 7 dependent launches of 5-17 us per
// modality on the critical path between the loss and the clip coefficient (MLP encoder, modules.py:145-149:
// layer_out <- Dropout <- ReLU <- BatchNorm <- layer_1; VAE encoder, modules.py:25-41,47-56: FC_mean, FC_var <-
// BatchNorm <- LeakyReLU <- hidden Linear).
//
// One workgroup owns 32 columns of the block (all B <= 128 rows): everything is column-local except the norm, whose
// contribution ||dY_blk^T X||_F^2 = <dY_blk dY_blk^T, X X^T> is one double per workgroup (summed by
// fx_clip_finalize in a fixed order).  Thread (cx = column, ry = row group) keeps its 8 rows of x, the block output
// and the gradient in registers; the small operands (dE [B, L], 32-column tiles) go through LDS.  512 threads: the
// phases are LDS-latency bound, and with 157 workgroups on 256 CUs the only latency hiding is inside the workgroup
// (256 threads: 42 us, phase ablation in scripts/bench_block_bwd.py).  Round 3: the three products (da = dE . W,
// gW = dE^T . out, dY dY^T) moved from FMA loops over LDS operands to the exact-fp32 matrix pipe
// (v_mfma_f32_16x16x4_f32, one 16 x 16 block family per wave): 50 -> 27 us at cfg2 for two modalities, same arithmetic
// error against fp64 (1.6e-7 rms, scripts/block_bwd_error.py).
#include "fx_common.h"
#include "fx_chain_prof.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define BB_COLS 32
#define BB_RG 16
#define BB_RPT 8
#define BB_T 512
#define BB_MAXL 64
#define BB_ACT_NONE 0
#define BB_ACT_LEAKY 1
#define BB_ACT_RELU 2
#define BB_LEAKY 0.2f
#define BB_DLD 68            // LDS row stride of the staged dE chunk [128][64 + 4]
#define BB_WLD 68            // ... of the staged W chunk, transposed [32 columns][64 + 4]
// The three products of a workgroup -- da = dE . W, gW = dE^T . out, dY dY^T -- run on the exact-fp32 matrix pipe
// (v_mfma_f32_16x16x4_f32: fp32 operands, fp32 accumulate).  Operand A: lane (m = lane & 15, kq = lane >> 4) holds A[m][k]
// for one k per instruction, B: lane (n = lane & 15, kq) holds B[k][n]; result: lane holds D[4 (lane >> 4) + i][lane & 15],
// i = 0..3.  A lane reads four consecutive k (16 q + 4 kq + e) at once and instruction e uses element e on both sides: the
// contraction index is a dummy, any pairing of k between A and B that is the same on both sides gives the same sum.
#define BB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

struct BlockBwdArgs {
  // up to two small Linears fed by the block output: dE_k [B, L_k] (upstream gradient), W_k [L_k, C]
  const float* dE[2]; long ldE[2]; const float* W[2]; float* gW[2]; float* gb[2]; int L[2]; int n_up;
  const float* x;          // [B, C] saved wide-Linear output (block input)
  const float* out;        // [B, C] saved block output (input of the small Linears; ReLU/dropout gate)
  const float* gamma; const float* save_mean; const float* save_invstd;
  float* dgamma; float* dbeta; float* dbias;
  float* dy;               // optional [B, C] fp32 gradient at the wide Linear's output
  __bf16* dyT_hi; __bf16* dyT_lo; long ldt;     // optional transposed split [C, ldt] (ldt >= round32(B), zero padded)
  __bf16* dyK_hi; __bf16* dyK_lo; long ldk; int krow0;   // optional K-blocked split [ceil(C / 32)][ldk rows][32] (fx_split_bf16's layout), rows krow0 ..
  const float* gram_x;     // optional [B, B] = X X^T: then slots[blockIdx] = this block's share of ||dW_wide||_F^2
  double* slots;
  int B, C; long ldx, ldo;
  int pre_act, post_act; float drop_p;
  int accumulate;          // parameter gradients (gW, gb, dgamma, dbeta, dbias) are ADDED to: a further BatchNorm pass over the same weights
};

__device__ __forceinline__ float bb_colsum(float v, float (*red)[BB_COLS], int cx, int ry) {
  __syncthreads();
  red[ry][cx] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < BB_RG; ++i) s += red[i][cx];
  return s;
}

// One upstream Linear (dE [B, L], W [L, C]): dacc += dE . W for wave w's 16 rows x 32 columns (two 16 x 16 blocks in MFMA
// result layout), gW = dE^T . out, gb = colsum(dE).
// Plain pointer arguments: indexing the kernel-argument arrays with a runtime k put the whole struct in scratch.
__device__ __forceinline__ void bb_upstream(const float* __restrict__ dE, long ldE, const float* __restrict__ W,
                                            float* __restrict__ gW, float* __restrict__ gb, int L, int B, int C, int c0, int cc,
                                            float* dEs, float (*T)[132], float (*Wt)[BB_WLD], f32x4 dacc[2], int accumulate,
                                            int blk) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  // the upstream width L (latent size: any integer) is processed in chunks of 64 rows of W / columns of dE
  for (int lc0 = 0; lc0 < L; lc0 += BB_MAXL) {
    const int Lc = min(BB_MAXL, L - lc0);
    float wv[4];
    {
      const int wl = t >> 5;                       // 16 rows per pass, 4 passes
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = W[(long)(lc0 + min(wl + 16 * j, Lc - 1)) * C + cc];
    }
    __syncthreads();                               // previous users of dEs / Wt; T holds the block output
    {   // stage dE_k[:, lc0 : lc0+64] row-major [128][BB_DLD], zero padded: thread = (row t>>4 + 32 j, 4 columns)
      const int lq = 4 * (t & 15), rb = t >> 4;
      float v[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = min(rb + 32 * j, B - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = dE[(long)r * ldE + lc0 + min(lq + q, Lc - 1)];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = rb + 32 * j;
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (r < B && lq + q < Lc) ? v[j][q] : 0.f;
        *reinterpret_cast<f32x4*>(&dEs[r * BB_DLD + lq]) = o;
      }
    }
    {   // W chunk transposed: Wt[column][l]
      const int wl = t >> 5, cxx = t & 31;
#pragma unroll
      for (int j = 0; j < 4; ++j) Wt[cxx][wl + 16 * j] = (wl + 16 * j < Lc) ? wv[j] : 0.f;
    }
    __syncthreads();
    // ---- da[16 w + ., 0..31] += dE[16 w + ., :] . W[:, c0 ..]: A = dEs rows, B = Wt columns, 4 x 4 k-steps
#pragma unroll
    for (int q = 0; q < BB_MAXL / 16; ++q) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(&dEs[(16 * w + l15) * BB_DLD + 16 * q + 4 * kq]);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Wt[l15][16 * q + 4 * kq]);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Wt[16 + l15][16 * q + 4 * kq]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dacc[0] = BB_MFMA(av[e], b0[e], dacc[0]);
        dacc[1] = BB_MFMA(av[e], b1[e], dacc[1]);
      }
    }
    // ---- gW_k[lc0 + l][c0 + c'] = sum_r dE[r][l] out[r][c']: wave w owns the 16 x 16 block (l block w >> 1, column block w & 1)
    {
      const int lb = w >> 1, cb = w & 1;
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
      for (int q = 0; q < 128 / 16; ++q) {
        const int rk = 16 * q + 4 * kq;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(&T[16 * cb + l15][rk]);
        f32x4 av;
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = dEs[(rk + e) * BB_DLD + 16 * lb + l15];
#pragma unroll
        for (int e = 0; e < 4; ++e) g = BB_MFMA(av[e], bv[e], g);
      }
      const int col = c0 + 16 * cb + l15;
      if (col < C) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int l = 16 * lb + 4 * kq + i;
          if (l < Lc) {
            float* dst = gW + (long)(lc0 + l) * C + col;                 // every element has exactly one owner lane
            *dst = accumulate ? *dst + g[i] : g[i];
          }
        }
      }
      if (gb && blk == 0 && t < Lc) {   // bias gradient = column sums of dE_k
        float sg = 0.f;
#pragma unroll 8
        for (int r = 0; r < 128; ++r) sg += dEs[r * BB_DLD + t];
        gb[lc0 + t] = accumulate ? gb[lc0 + t] + sg : sg;
      }
    }
  }
}

__device__ __forceinline__ void bb_body(const BlockBwdArgs& a, int blk) {
  __shared__ __attribute__((aligned(16))) float dEs[128 * BB_DLD];       // [r][L] upstream gradient (rows >= B zero); later da [r][33]
  __shared__ __attribute__((aligned(16))) float T[BB_COLS][132];         // [c][r]: block output, later dy
  __shared__ __attribute__((aligned(16))) float Wt[BB_COLS][BB_WLD];     // one 64-row chunk of an upstream weight, transposed
  __shared__ float red[BB_RG][BB_COLS];
  __shared__ double dred[BB_T / 64];
  const int t = threadIdx.x, cx = t & 31, ry = t >> 5;
  const int c0 = blk * BB_COLS, c = c0 + cx, B = a.B;
  const int cc = min(c, a.C - 1);                    // clamped column: loads are unconditional, stores predicated
  const bool cok = c < a.C;
  const int r0 = ry * BB_RPT;
  const int cpw = blockIdx.y * gridDim.x + blockIdx.x;
  CP_TICK(cpw, 0);
  // ---- this thread's 16 rows of x and of the block output
  float xv[BB_RPT], ov[BB_RPT];
#pragma unroll
  for (int i = 0; i < BB_RPT; ++i) {
    const int r = min(r0 + i, B - 1);
    xv[i] = a.x[(long)r * a.ldx + cc];
    ov[i] = a.out[(long)r * a.ldo + cc];
  }
  const float mean = a.save_mean[cc], invstd = a.save_invstd[cc], gm = a.gamma[cc];
#pragma unroll
  for (int i = 0; i < BB_RPT; ++i) {
    const bool rok = r0 + i < B;
    ov[i] = (rok && cok) ? ov[i] : 0.f;
    T[cx][r0 + i] = ov[i];
  }
  CP_TICK(cpw, 1);
  // ---- da = sum_k dE_k . W_k (this thread: 16 rows x its column), and the small Linears' weight/bias gradients
  f32x4 dacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  bb_upstream(a.dE[0], a.ldE[0], a.W[0], a.gW[0], a.gb[0], a.L[0], B, a.C, c0, cc, dEs, T, Wt, dacc, a.accumulate, blk);
  if (a.n_up > 1) bb_upstream(a.dE[1], a.ldE[1], a.W[1], a.gW[1], a.gb[1], a.L[1], B, a.C, c0, cc, dEs, T, Wt, dacc, a.accumulate, blk);
  CP_TICK(cpw, 2);
  // da from the MFMA result layout to this thread's (column cx, rows r0 ..) through the dE stage (dead now)
  float da[BB_RPT];
  {
    const int lane = t & 63, w = t >> 6;
    __syncthreads();                                 // every wave is done reading dEs
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 4; ++i) dEs[(16 * w + 4 * (lane >> 4) + i) * 33 + 16 * nb + (lane & 15)] = dacc[nb][i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BB_RPT; ++i) da[i] = dEs[(r0 + i) * 33 + cx];
  }
  CP_TICK(cpw, 3);
  // ---- gate (ReLU + dropout in one test on the saved output) and BatchNorm backward (fx_bn_bwd_kernel's expressions)
  const float gate_scale = 1.0f / (1.0f - a.drop_p);
  float s1 = 0.f, s2 = 0.f;
  float xh[BB_RPT];
#pragma unroll
  for (int i = 0; i < BB_RPT; ++i) {
    const bool rok = r0 + i < B;
    float d = da[i];
    if (a.post_act == BB_ACT_RELU) d = (ov[i] > 0.f) ? d * gate_scale : 0.f;
    d = (rok && cok) ? d : 0.f;
    const float xr = xv[i];
    const float xa = (a.pre_act == BB_ACT_LEAKY) ? (xr > 0.f ? xr : xr * BB_LEAKY) : xr;
    xh[i] = rok ? (xa - mean) * invstd : 0.f;
    da[i] = d;
    s1 += d;
    s2 += d * xh[i];
  }
  const float sum_dy = bb_colsum(s1, red, cx, ry);
  const float sum_dy_xh = bb_colsum(s2, red, cx, ry);
  const float invB = 1.0f / (float)B;
  float sb = 0.f;
#pragma unroll
  for (int i = 0; i < BB_RPT; ++i) {
    const bool rok = r0 + i < B;
    float d = gm * invstd * (da[i] - invB * sum_dy - xh[i] * invB * sum_dy_xh);
    if (a.pre_act == BB_ACT_LEAKY) d = xv[i] > 0.f ? d : d * BB_LEAKY;
    d = (rok && cok) ? d : 0.f;
    da[i] = d;
    sb += d;
  }
  const float sum_dx = bb_colsum(sb, red, cx, ry);     // (its barriers also order the last reads of T above)
  CP_TICK(cpw, 4);
  // ---- the X X^T entries the norm share multiplies (rows 16 w + 4 kq + i, columns 16 bj + l15 of this lane's result blocks) are
  // REQUESTED here, in front of every store of the epilogue: read inside the product loop they were eight dependent L2 round trips
  // (8.5 of the launch's ~20 us, scripts/chain_profile.py); clamped addresses, the predicates stay on the use
  // (in two halves of 16 registers: all 32 at once cost the second workgroup of a CU its registers)
  float gxv[8][4];
#define BB_GRAM_REQ(b0)                                                                                                      \
  {                                                                                                                          \
    const int lane_ = t & 63, w_ = t >> 6, l15_ = lane_ & 15, kq_ = lane_ >> 4;                                              \
    _Pragma("unroll") for (int bj = (b0); bj < (b0) + 4; ++bj)                                                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                          \
        gxv[bj][i] = a.gram_x[(long)min(16 * w_ + 4 * kq_ + i, B - 1) * B + min(16 * bj + l15_, B - 1)];                     \
  }
  if (a.gram_x) BB_GRAM_REQ(0)
  if (cok && ry == 0) {
    a.dgamma[c] = a.accumulate ? a.dgamma[c] + sum_dy_xh : sum_dy_xh;
    a.dbeta[c] = a.accumulate ? a.dbeta[c] + sum_dy : sum_dy;
    if (a.dbias) a.dbias[c] = a.accumulate ? a.dbias[c] + sum_dx : sum_dx;
  }
  if (a.dy && cok) {
#pragma unroll
    for (int i = 0; i < BB_RPT; ++i)
      if (r0 + i < B) a.dy[(long)(r0 + i) * a.ldx + c] = da[i];
  }
  // ---- dy tile (transposed) for the operand split and the norm
#pragma unroll
  for (int i = 0; i < BB_RPT; ++i) T[cx][r0 + i] = da[i];
  __syncthreads();
  CP_TICK(cpw, 5);
  if (a.dyT_hi) {      // thread: column t >> 4, rows (t & 15) * 8 .. +8  -> one 16-byte store per array
    const int col = t >> 4, rb = (t & 15) * 8;
    if (c0 + col < a.C && rb < ((B + 31) & ~31)) {       // rows B .. round32(B) - 1 are written as zeros; nothing beyond (a pass of stacked rows owns only its columns)
      bf16x8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = T[col][rb + j];
        hi[j] = (__bf16)v;
        lo[j] = (__bf16)(v - (float)hi[j]);
      }
      *reinterpret_cast<bf16x8*>(a.dyT_hi + (long)(c0 + col) * a.ldt + rb) = hi;
      *reinterpret_cast<bf16x8*>(a.dyT_lo + (long)(c0 + col) * a.ldt + rb) = lo;
    }
  }
  if (a.dyK_hi) {      // the same values in the K-blocked layout: this workgroup's 32 columns ARE K-block blk; thread = (row t >> 2, columns 8 (t & 3) ..)
    const int row = t >> 2, cg = (t & 3) * 8;
    if (row < ((B + 31) & ~31)) {                      // rows B .. round32(B) - 1 as zeros, nothing beyond (as above)
      bf16x8 hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = T[cg + j][row];
        hi[j] = (__bf16)v;
        lo[j] = (__bf16)(v - (float)hi[j]);
      }
      const long o = ((long)blk * a.ldk + a.krow0 + row) * 32 + cg;
      *reinterpret_cast<bf16x8*>(a.dyK_hi + o) = hi;
      *reinterpret_cast<bf16x8*>(a.dyK_lo + o) = lo;
    }
  }
  CP_TICK(cpw, 6);
  if (a.gram_x) {      // <dY_blk dY_blk^T, X X^T>: wave w owns rows 16 w .. 16 w + 15 of the B x B product (8 blocks of 16 x 16)
    const int lane = t & 63, w = t >> 6, l15 = lane & 15, kq = lane >> 4;
    f32x4 av[BB_COLS / 16];
#pragma unroll
    for (int q = 0; q < BB_COLS / 16; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) av[q][e] = T[16 * q + 4 * kq + e][16 * w + l15];
    double acc = 0.0;
    BB_GRAM_REQ(4)
#pragma unroll
    for (int bj = 0; bj < 8; ++bj) {
      f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < BB_COLS / 16; ++q) {
        f32x4 bv;
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = T[16 * q + 4 * kq + e][16 * bj + l15];
#pragma unroll
        for (int e = 0; e < 4; ++e) p = BB_MFMA(av[q][e], bv[e], p);
      }
      const int j = 16 * bj + l15;
      if (j < B) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int ii = 16 * w + 4 * kq + i;
          if (ii < B) acc += (double)p[i] * (double)gxv[bj][i];
        }
      }
    }
    acc = fx_wave_sum_d(acc);
    if ((t & 63) == 0) dred[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
      double tot = 0.0;
#pragma unroll
      for (int i = 0; i < BB_T / 64; ++i) tot += dred[i];
      a.slots[blk] = tot;
    }
  }
  CP_TICK(cpw, 7);
}

__global__ __launch_bounds__(BB_T, 4) void fx_block_bwd_kernel(BlockBwdArgs a) { bb_body(a, blockIdx.x); }

// Several encoder tails (one per modality) in one launch: grid (column blocks of the widest, modalities).  The modalities'
// backward chains are independent until the optimiser; as two launches on two hipGraph branches they paid a fork and a join
// (~10 us each on the critical chain: cross-queue dependencies) for running side by side.
#define BB_MAX_GROUP 4
struct BlockBwdGroup {
  BlockBwdArgs a[BB_MAX_GROUP];
};
__global__ __launch_bounds__(BB_T, 4) void fx_block_bwd_group_kernel(BlockBwdGroup g) {
  const int blk = blockIdx.x;
  // (a runtime index into the kernel-argument array would put the argument blocks in scratch: one call per constant index)
  switch (blockIdx.y) {
    case 0: if (blk * BB_COLS < g.a[0].C) bb_body(g.a[0], blk); break;
    case 1: if (blk * BB_COLS < g.a[1].C) bb_body(g.a[1], blk); break;
    case 2: if (blk * BB_COLS < g.a[2].C) bb_body(g.a[2], blk); break;
    default: if (blk * BB_COLS < g.a[3].C) bb_body(g.a[3], blk); break;
  }
}

// two tails (the common case: two omics layers): half the inlined copies, no register spills
struct BlockBwdGroup2 {
  BlockBwdArgs a[2];
};
__global__ __launch_bounds__(BB_T, 4) void fx_block_bwd_group2_kernel(BlockBwdGroup2 g) {
  const int blk = blockIdx.x;
  if (blockIdx.y == 0) { if (blk * BB_COLS < g.a[0].C) bb_body(g.a[0], blk); }
  else { if (blk * BB_COLS < g.a[1].C) bb_body(g.a[1], blk); }
}

CP_READER(fx_debug_chain_stamps_bb)

extern "C" {

int fx_block_bwd_blocks(int C) { return (C + BB_COLS - 1) / BB_COLS; }

// see include/fxhip.h
static int bb_fill(BlockBwdArgs& a, const float* const* dE, const long* ldE, const float* const* W, float* const* gW,
                   float* const* gb, const int* L, int n_up, const float* x, const float* out, const float* gamma,
                   const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* dbias, float* dy,
                   void* dyT_hi, void* dyT_lo, long ldt, const float* gram_x, double* slots, int B, int C, long ldx, long ldo,
                   int pre_act, int post_act, float drop_p, int accumulate, void* dyK_hi = nullptr, void* dyK_lo = nullptr,
                   long ldk = 0, int krow0 = 0) {
  FX_REQUIRE(dE && ldE && W && gW && gb && L && n_up >= 1 && n_up <= 2, "fx_block_bwd: 1 or 2 upstream Linears");
  FX_REQUIRE(x && out && gamma && save_mean && save_invstd && dgamma && dbeta, "fx_block_bwd: null pointer");
  FX_REQUIRE(B > 1 && B <= 128 && C > 0, "fx_block_bwd: B=%d must be in 2..128", B);
  FX_REQUIRE(!dyT_hi || (dyT_lo && ldt % 8 == 0 && ldt >= (B + 31) / 32 * 32),
             "fx_block_bwd: dyT needs hi and lo, ld %% 8 == 0, ld >= round32(B) (got %ld)", ldt);
  FX_REQUIRE(!gram_x || slots, "fx_block_bwd: gram_x needs the norm slots");
  FX_REQUIRE(!dyK_hi || (dyK_lo && krow0 >= 0 && krow0 % 8 == 0 && ldk >= krow0 + (B + 31) / 32 * 32 && ((((uintptr_t)dyK_hi) | ((uintptr_t)dyK_lo)) & 15) == 0),
             "fx_block_bwd: the K-blocked split needs hi and lo (16-byte aligned), row0 %% 8 == 0, rows >= row0 + round32(B) (got %ld, %d)", ldk, krow0);
  a = BlockBwdArgs{};
  for (int k = 0; k < n_up; ++k) {
    FX_REQUIRE(dE[k] && W[k] && gW[k] && L[k] > 0 && ldE[k] >= L[k], "fx_block_bwd: upstream %d: bad L=%d / ld=%ld", k, L[k],
               ldE[k]);
    a.dE[k] = dE[k]; a.ldE[k] = ldE[k]; a.W[k] = W[k]; a.gW[k] = gW[k]; a.gb[k] = gb[k]; a.L[k] = L[k];
  }
  a.n_up = n_up;
  a.x = x; a.out = out; a.gamma = gamma; a.save_mean = save_mean; a.save_invstd = save_invstd;
  a.dgamma = dgamma; a.dbeta = dbeta; a.dbias = dbias; a.dy = dy;
  a.dyT_hi = (__bf16*)dyT_hi; a.dyT_lo = (__bf16*)dyT_lo; a.ldt = ldt;
  a.dyK_hi = (__bf16*)dyK_hi; a.dyK_lo = (__bf16*)dyK_lo; a.ldk = ldk; a.krow0 = krow0;
  a.gram_x = gram_x; a.slots = slots;
  a.B = B; a.C = C; a.ldx = ldx; a.ldo = ldo; a.pre_act = pre_act; a.post_act = post_act; a.drop_p = drop_p;
  a.accumulate = accumulate ? 1 : 0;
  return 0;
}

int fx_block_bwd(const float* const* dE, const long* ldE, const float* const* W, float* const* gW, float* const* gb,
                 const int* L, int n_up, const float* x, const float* out, const float* gamma, const float* save_mean,
                 const float* save_invstd, float* dgamma, float* dbeta, float* dbias, float* dy, void* dyT_hi, void* dyT_lo,
                 long ldt, const float* gram_x, double* slots, int B, int C, long ldx, long ldo, int pre_act, int post_act,
                 float drop_p, int accumulate, hipStream_t stream) {
  BlockBwdArgs a;
  if (int rc = bb_fill(a, dE, ldE, W, gW, gb, L, n_up, x, out, gamma, save_mean, save_invstd, dgamma, dbeta, dbias, dy, dyT_hi,
                       dyT_lo, ldt, gram_x, slots, B, C, ldx, ldo, pre_act, post_act, drop_p, accumulate))
    return rc;
  hipLaunchKernelGGL(fx_block_bwd_kernel, dim3(fx_block_bwd_blocks(C)), dim3(BB_T), 0, stream, a);
  return fx_check_launch("fx_block_bwd");
}

// fx_block_bwd + dY's K-blocked split (fx_split_bf16's layout [ceil(C / 32)][kb_rows][32], rows kb_row0 .. of it): see include/fxhip.h
int fx_block_bwd_ex(const float* const* dE, const long* ldE, const float* const* W, float* const* gW, float* const* gb,
                    const int* L, int n_up, const float* x, const float* out, const float* gamma, const float* save_mean,
                    const float* save_invstd, float* dgamma, float* dbeta, float* dbias, float* dy, void* dyT_hi, void* dyT_lo,
                    long ldt, const float* gram_x, double* slots, int B, int C, long ldx, long ldo, int pre_act, int post_act,
                    float drop_p, int accumulate, void* dy_kb_hi, void* dy_kb_lo, long kb_rows, int kb_row0, hipStream_t stream) {
  BlockBwdArgs a;
  if (int rc = bb_fill(a, dE, ldE, W, gW, gb, L, n_up, x, out, gamma, save_mean, save_invstd, dgamma, dbeta, dbias, dy, dyT_hi,
                       dyT_lo, ldt, gram_x, slots, B, C, ldx, ldo, pre_act, post_act, drop_p, accumulate, dy_kb_hi, dy_kb_lo, kb_rows, kb_row0))
    return rc;
  hipLaunchKernelGGL(fx_block_bwd_kernel, dim3(fx_block_bwd_blocks(C)), dim3(BB_T), 0, stream, a);
  return fx_check_launch("fx_block_bwd_ex");
}

struct fx_block_bwd_desc_ {   // include/fxhip.h: fx_block_bwd_desc
  const float* dE[2]; long ldE[2]; const float* W[2]; float* gW[2]; float* gb[2]; int L[2]; int n_up;
  const float* x; const float* out; const float* gamma; const float* save_mean; const float* save_invstd;
  float* dgamma; float* dbeta; float* dbias; float* dy; void* dyT_hi; void* dyT_lo; long ldt;
  const float* gram_x; double* slots; int C; long ldx, ldo; int accumulate;
};

// fx_block_bwd for n <= 4 independent encoder tails (one per modality) in one launch; every field as the argument of the
// same name of fx_block_bwd.
int fx_block_bwd_group(const void* descs_, int n, int B, int pre_act, int post_act, float drop_p, hipStream_t stream) {
  const fx_block_bwd_desc_* d = (const fx_block_bwd_desc_*)descs_;
  FX_REQUIRE(d && n > 0 && n <= BB_MAX_GROUP, "fx_block_bwd_group: 1..%d tails per launch", BB_MAX_GROUP);
  BlockBwdGroup g{};
  int max_blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (int rc = bb_fill(g.a[i], d[i].dE, d[i].ldE, d[i].W, d[i].gW, d[i].gb, d[i].L, d[i].n_up, d[i].x, d[i].out, d[i].gamma,
                         d[i].save_mean, d[i].save_invstd, d[i].dgamma, d[i].dbeta, d[i].dbias, d[i].dy, d[i].dyT_hi, d[i].dyT_lo,
                         d[i].ldt, d[i].gram_x, d[i].slots, B, d[i].C, d[i].ldx, d[i].ldo, pre_act, post_act, drop_p,
                         d[i].accumulate))
      return rc;
    const int nb = fx_block_bwd_blocks(d[i].C);
    max_blocks = nb > max_blocks ? nb : max_blocks;
  }
  for (int i = n; i < BB_MAX_GROUP; ++i) g.a[i] = g.a[0];
  if (n <= 2) {
    BlockBwdGroup2 g2{};
    g2.a[0] = g.a[0];
    g2.a[1] = g.a[n - 1];
    hipLaunchKernelGGL(fx_block_bwd_group2_kernel, dim3(max_blocks, n), dim3(BB_T), 0, stream, g2);
    return fx_check_launch("fx_block_bwd_group");
  }
  hipLaunchKernelGGL(fx_block_bwd_group_kernel, dim3(max_blocks, n), dim3(BB_T), 0, stream, g);
  return fx_check_launch("fx_block_bwd_group");
}

}  // extern "C"
