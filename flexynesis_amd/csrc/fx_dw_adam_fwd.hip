// Fused  dW + clip + Adam + NEXT-STEP FORWARD  for one wide weight (gfx950 / CDNA4, split-bf16 MFMA).
//
// The training step has to stream every wide weight twice: once for the optimiser (read + write W, m, v: 24 B/param)
// and once more for the following step's forward (4 B/param).  The forward of step t+1 only needs W_{t+1} and the
// batch of step t+1 -- and the engine assembles that batch one step ahead (PipelinedStep).  So this kernel, right after
// it has produced a tile of W_{t+1} in registers, multiplies it into the next batch and keeps the partial sums: the
// next step never re-reads W.  28 -> 24 B/param/step, one kernel launch and one split-K slab round trip less per wide
// weight.
//
//   W [M = n_out, N = k_in] (+ m, v)        fp32, row-major, rows padded (ldw)
//   dY^T [M, K], X^T [N, K]                  split bf16 (hi, lo), K = padded batch (fx_split_bf16_t / fx_gather_split)
//   Xn  K-blocked [ceil(N/32)][Rp][32]       split bf16 of the NEXT batch (fx_gather_split / fx_split_bf16); Rp = 128 MT rows,
//                                            MT = 1..3 M-tiles (the triplet network stacks anchor / positive / negative: 3 B rows)
//   Y slabs [S][Mn][M]                       partial sums of Xn . W_new^T; fx_reduce_slabs adds them (+ bias) in order
//
// Work decomposition.  A workgroup (512 threads) owns one 64-row block of W and a RUN of consecutive 128-column tiles;
// per tile:  (1) dW tile = dY^T . X on the MFMA (3 bf16 products, fp32 accumulate), operands streamed from L2 into LDS
// by LDS-DMA (buffer_load ... lds: no staging registers, the XOR swizzle is applied on the source address);
// (2) the tile is transposed through LDS so that every lane owns 4 consecutive columns; (3) W, m, v move as 16-byte
// non-temporal buffer accesses in 512-byte row segments, Adam in registers; W_new is split to bf16 (hi, lo) into LDS;
// (4) Y[b, h] += Xn[b, tile] . W_new[h, tile]^T on the MFMA, Xn K-steps by LDS-DMA, the accumulators stay in registers
// for the whole run; at the end of the run one [Mn x 64] block of one slab is stored.  S = floor(512 / row blocks) runs
// (run c = column tiles c, c + S, ...) cover the columns, so there is about one workgroup per resident slot (2 per CU)
// and every workgroup does the same work.
// All descriptors are rebased per row block, so W may exceed 4 GiB.  Deterministic: fixed run / slab order, no atomics.
#include "fx_common.h"
#include "fx_reduce.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define FT_M 64
#define FT_N 128
#define FT_K 32
#define FT_T 512
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// -DFT_PROFILE (scripts/build_variant.py, never the shipped library): wave 0 of every workgroup accumulates the 100 MHz wall-clock
// ticks it spends in each phase of a tile; fx_debug_dw_adam_fwd_profile() reads / resets the sums.
#ifdef FT_PROFILE
__device__ unsigned long long ft_prof[16];
__device__ unsigned long long ft_stamp[1024 * 8];      // [workgroup][k]: wall clock at the start of phase 3 of the workgroup's tile 3 k + 2
#define FT_STAMP(it) if (tid == 0 && ((it) % 3) == 2 && (it) / 3 < 8 && blockIdx.x < 1024) ft_stamp[blockIdx.x * 8 + (it) / 3] = wall_clock64();
#define FT_PROF_DECL unsigned long long pt_[16] = {0}, pl_ = wall_clock64()
#define FT_TICK(i) { const unsigned long long n_ = wall_clock64(); pt_[i] += n_ - pl_; pl_ = n_; }
#define FT_PROF_FLUSH if (tid == 0) { for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&ft_prof[i_], pt_[i_]); }
#define FT_PROF_WAITVM __builtin_amdgcn_s_waitcnt(0x0f70)   /* vmcnt(0) only (gfx9 encoding: vmcnt lo [3:0], hi [15:14]) */
#else
#define FT_PROF_DECL
#define FT_TICK(i)
#define FT_PROF_FLUSH
#define FT_PROF_WAITVM
#define FT_STAMP(it)
#endif

// -DFT_ABL=n (scripts/build_variant.py, never the shipped library; RESULTS ARE WRONG): energy / time ablations of the two GEMM phases --
// 3: no MFMAs (the fragment reads go with them; the operand LDS-DMA stays), 5: no operand LDS-DMA (MFMAs + fragment reads on stale LDS),
// 4: neither (transposes + Adam stream only), 6: MFMAs without their `lo` fragment reads (ah / bh used for all three products),
// 7: the three products in an order in which consecutive MFMAs share an operand (al bh, ah bh, ah bl) -- correct results, another summation order.
// scripts/fused_energy_ablation.py, profiles/r06_fused_energy_ablation.txt
#ifndef FT_ABL
#define FT_ABL 0
#endif
#define FT_DMA_ON (FT_ABL != 5 && FT_ABL != 4)
#define FT_MFMA_ON (FT_ABL != 3 && FT_ABL != 4)

struct DwAdamFwdArgs {
  float* W; float* m; float* v;
  float* Wd; float* md; float* vd;          // where the updated W / m / v go: == W / m / v (in place) or twins in another memory partition (out of place)
  const __bf16* Ah; const __bf16* Al;       // dY^T [M, lda]
  const __bf16* Bh; const __bf16* Bl;       // X^T  [N, ldb]
  const __bf16* Xh; const __bf16* Xl;       // next batch, K-blocked [kblocks][128][32]
  float* Y;                                 // slabs [S][Mn][M]
  const float* ctrl;
  int M, N, K;
  long lda, ldb, ldw;
  int Mn, kblocks;
  int tiles_m, tiles_n, S;                  // S = runs per row block of the "low" class (see ft_plan)
  long slab_stride;
  int xcd_group;                            // 0 plain, 1: the S runs of a row block share an XCD (block id -> (XCD, row block, run)), 2: XCD-contiguous row blocks
  int n_hi;                                 // plain mapping: the LAST n_hi row blocks are split into S + 1 runs (all 512 slots filled)
  int slots;                                // XCD-contiguous mapping: workgroup slots per XCD
  int S_total;                              // slabs the consumer adds up; a row block with fewer runs zero-fills the rest
  int prio;                                 // 0: the two workgroups of a CU take turns at issue priority (per tile); 1 off; 2, 3 variants
};

// ---- work decomposition -----------------------------------------------------------------------------
// A run = (64-row block, every S-th 128-column tile); one workgroup per run, one partial-sum slab per run index.  The chip holds
// G = 512 workgroups (2 per CU).  Round 2 used S = floor(G / row blocks) for every row block: 79 x 6 = 474 workgroups at the cfg2
// shape, i.e. 38 CUs ran ONE workgroup (12.4 us per tile against 8.8 us per tile for a CU that holds two) and sat idle for the
// last third of the launch.  Now the last n_hi = G - row blocks x S row blocks get S + 1 (shorter) runs, so that every slot is
// taken; because the hardware places workgroup b and b + 256 on the same CU and issues the OLDER workgroup's waves first (measured:
// 16.2 vs 19.3 us per tile), it is right that the higher-numbered workgroups are the ones with the shorter runs.
#define FT_G 512
struct FtPlan { int S_lo, n_hi, S_total; };
static FtPlan ft_plan(int tiles_m, int tiles_n, int slots) {
  FtPlan p;
  p.S_lo = slots / tiles_m;
  if (p.S_lo < 1) p.S_lo = 1;
  if (p.S_lo > tiles_n) p.S_lo = tiles_n;
  p.n_hi = 0;
  if (p.S_lo < tiles_n && tiles_m * p.S_lo < slots) {
    p.n_hi = slots - tiles_m * p.S_lo;
    if (p.n_hi > tiles_m) p.n_hi = tiles_m;
  }
  p.S_total = p.S_lo + (p.n_hi > 0 ? 1 : 0);
  return p;
}
__device__ __forceinline__ void ft_plan_dev(int tiles_m, int tiles_n, int slots, int& S_lo, int& n_hi) {
  S_lo = slots / tiles_m;
  S_lo = S_lo < 1 ? 1 : S_lo;
  S_lo = S_lo > tiles_n ? tiles_n : S_lo;
  n_hi = 0;
  if (S_lo < tiles_n && tiles_m * S_lo < slots) n_hi = min(slots - tiles_m * S_lo, tiles_m);
}

__device__ __forceinline__ int ft_swz(int row, int chunk) { return row * FT_K + ((chunk ^ ((row >> 2) & 3)) << 3); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ft_rsrc(const void* p, long bytes) {
  const unsigned n = bytes > 0xFFFFFFF0L ? 0xFFFFFFF0u : (bytes < 0 ? 0u : (unsigned)bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);
}

#define FT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// NB: takes the value as a SCALAR float.  __builtin_bit_cast(unsigned, vec[r]) applied directly to an element of an
// ext_vector is miscompiled by hipcc 7.2 (every r reads element 0).
__device__ __forceinline__ void ft_store32(float v, __amdgpu_buffer_rsrc_t r, unsigned off) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}

// MT = M-tiles of the next batch (rows / 128); UNITS = 16-byte W / m / v units per thread in flight in the Adam phase
// (4 for MT = 1; 2 for MT > 1, whose MT accumulator sets need the registers).
// NP = products per contraction: 3 = split bf16 (hi hi + hi lo + lo hi: fp32-grade, the parity mode), 1 = plain bf16 (hi hi only: the
// `lo` halves are neither loaded nor produced -- the throughput mode, torch's "medium" matmul precision of the reference, main.py:24).
template <int NT, int MT, int UNITS, int NP = 3>
__global__ __launch_bounds__(FT_T, 2) void fx_dw_adam_fwd_kernel(DwAdamFwdArgs g) {
  // LDS map (bf16 elements unless noted), 64 KB:
  //   phase 1 (dW K-loop)   stage s at s*12288: A hi [64][32] | A lo | B hi [128][32] | B lo          (2 x 24 KB)
  //   phase 2/3             ct fp32 [64][128] at 0 (32 KB) ; Wn hi [4][64][32] at 16384, Wn lo at 24576 (32 KB)
  //   phase 4 (forward)     X stage j at j*8192: hi [128][32] | lo                                     (2 x 16 KB)
  __shared__ __attribute__((aligned(16))) __bf16 smem[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave id, scalar
  int tm, c, S = g.S;                                              // S = this row block's number of runs
  if (g.xcd_group == 2) {
    // XCD-CONTIGUOUS row blocks: XCD x owns the row blocks [x * rpx, (x + 1) * rpx) with all their runs (the rows an XCD
    // touches -- and translates -- are 1/8 of the weight), and splits them over its own `slots` workgroups as ft_plan does
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, rpx = (g.tiles_m + 7) >> 3;
    const int r0 = xcd * rpx, nb = min(rpx, g.tiles_m - r0);
    if (nb <= 0) return;
    int S_lo, n_hi;
    ft_plan_dev(nb, g.tiles_n, g.slots, S_lo, n_hi);
    const int n_lo = nb - n_hi, low = n_lo * S_lo;
    if (j < low) { tm = r0 + j % n_lo; c = j / n_lo; S = S_lo; }
    else {
      const int j2 = j - low;
      if (j2 >= n_hi * (S_lo + 1)) return;
      tm = r0 + n_lo + j2 % n_hi; c = j2 / n_hi; S = S_lo + 1;
    }
  } else if (g.xcd_group) {
    // Workgroup ids go round-robin over the 8 XCDs.  Here every row block lives on ONE XCD with all its S runs, so its
    // dY^T tile (64 rows x K, re-read for every column tile) is shared in that XCD's L2: with K = 3 B = 384 the tiles of
    // the ~59 workgroups of an XCD would otherwise outgrow the 4 MB L2 (5.8 MB) and be re-fetched for every tile.
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tm = (j / g.S) * 8 + xcd;
    c = j % g.S;
    if (tm >= g.tiles_m) return;                       // padding of the index space (before any barrier)
  } else {
    const int n_lo = g.tiles_m - g.n_hi, low = n_lo * g.S;
    if ((int)blockIdx.x < low) { tm = blockIdx.x % n_lo; c = blockIdx.x / n_lo; }
    else { const int b2 = blockIdx.x - low; tm = n_lo + b2 % g.n_hi; c = b2 / g.n_hi; S = g.S + 1; }
  }
  const int m0 = tm * FT_M;
  const int rows_valid = min(FT_M, g.M - m0);
  const int nk = g.K / FT_K;

  // ---- descriptors, rebased to this row block (in-block offsets stay far below 4 GiB whatever the weight's size)
  const long wbytes = ((long)(rows_valid - 1) * g.ldw + g.N) * 4;
  const __amdgpu_buffer_rsrc_t rP = ft_rsrc(g.W + (long)m0 * g.ldw, wbytes), rM = ft_rsrc(g.m + (long)m0 * g.ldw, wbytes),
                               rV = ft_rsrc(g.v + (long)m0 * g.ldw, wbytes);
  const __amdgpu_buffer_rsrc_t rPd = ft_rsrc(g.Wd + (long)m0 * g.ldw, wbytes), rMd = ft_rsrc(g.md + (long)m0 * g.ldw, wbytes),
                               rVd = ft_rsrc(g.vd + (long)m0 * g.ldw, wbytes);
  const long abytes = (long)rows_valid * g.lda * 2;
  const __amdgpu_buffer_rsrc_t rAh = ft_rsrc(g.Ah + (long)m0 * g.lda, abytes), rAl = ft_rsrc(g.Al + (long)m0 * g.lda, abytes);
  const __amdgpu_buffer_rsrc_t rBh = ft_rsrc(g.Bh, (long)g.N * g.ldb * 2), rBl = ft_rsrc(g.Bl, (long)g.N * g.ldb * 2);
  const long xbytes = (long)g.kblocks * (128 * MT) * FT_K * 2;
  const __amdgpu_buffer_rsrc_t rXh = ft_rsrc(g.Xh, xbytes), rXl = ft_rsrc(g.Xl, xbytes);

  // ---- LDS-DMA source addressing: a wave instruction fills 16 rows x 64 B in lane order (row = lane / 4, 16-byte slot =
  // lane % 4); slot s of row r must hold chunk s ^ ((r >> 2) & 3) (the swizzle the fragment reads apply)
  const int dr = lane >> 2, ds = lane & 3;
  const int a_r = 16 * (w & 3) + dr;                                   // A rows 0..63: waves 0-3 fill hi, 4-7 fill lo
  const unsigned a_src = (unsigned)(((long)a_r * g.lda + 8 * (ds ^ ((a_r >> 2) & 3))) * 2);
  const int b_r = 16 * w + dr;                                         // B / X rows 0..127
  const int b_ch = ds ^ ((b_r >> 2) & 3);
  const unsigned x_src = (unsigned)((b_r * FT_K + 8 * b_ch) * 2);      // within one M-tile of a K-blocked [128 MT][32] block

  // ---- fragment coordinates
  const int l31 = lane & 31, kh = lane >> 5;
  const int rb = w & 1, cb = w >> 1;                                   // dW: 64 x 128 tile = 2 x 4 blocks of 32 x 32
  const int bq = w & 3, hq = w >> 2;                                   // forward: 128 (batch) x 64 (rows of W) = 4 x 2 blocks
  const int fa_d = 32 * rb + l31, fb_d = 32 * cb + l31;
  const int fa_f = 32 * bq + l31, fb_f = 32 * hq + l31;

  const FxAdamK ak = fx_adam_consts(g.ctrl[FXC_LR], g.ctrl[FXC_BC1], g.ctrl[FXC_BC2_SQRT], g.ctrl[FXC_CLIP_COEF]);

  f32x16 yacc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) yacc[m][i] = 0.f;

  FT_PROF_DECL;
  // Run c takes the column tiles c, c + S, c + 2 S, ...: the S workgroups of a row block sit on ADJACENT tiles at any
  // moment (a contiguous S x 512-byte window of every row of W / m / v moves along the rows), and all row blocks work on
  // the same few X tiles at the same time (L2 hits).  Contiguous chunks per run measured like the "slices" copy pattern
  // (scripts/copybench.hip: 5.3-5.5 TB/s against 6.5 for a moving contiguous window).
  for (int tn = c; tn < g.tiles_n; tn += S) {
    const int n0 = tn * FT_N;
    const unsigned b_src = (unsigned)(((long)(n0 + b_r) * g.ldb + 8 * b_ch) * 2);
    FT_STAMP((tn - c) / S);
    // Oldest-first issue arbitration favours the workgroup that arrived first on its CU (block b over block b + grid / 2): 14.5 vs
    // 20.7 us per tile, and the two halves of the grid drift apart by several column tiles, which costs L2 hits on the operands.
    // prio 0 (default): both halves alternate 1 / 0 per tile (in anti-phase at the start): 457 -> 446 us per launch, drift 1.9 -> 0.85
    // tiles; prio 1: off; prio 2: the first half alternates 2 / 0, the second half stays at 1 (each side wins half of the time whatever
    // their relative phase: drift 0.47 tiles, same time); prio 3: static, second half above first (mirrors the problem)
    if (g.prio == 0) {
      if ((((tn - c) / S) + (blockIdx.x >= (gridDim.x >> 1) ? 1 : 0)) & 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    } else if (g.prio == 2) {
      if (blockIdx.x >= (gridDim.x >> 1)) __builtin_amdgcn_s_setprio(1);
      else if (((tn - c) / S) & 1) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(0);
    } else if (g.prio == 3) {
      if (blockIdx.x >= (gridDim.x >> 1)) __builtin_amdgcn_s_setprio(1);
    }
    // ================= phase 1: dW tile [64 x 128] = dY^T[m0.., :] . X^T[n0.., :]^T over the batch =================
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#define FT_GLDS_K(stage, kt)                                                                              \
  {                                                                                                       \
    __bf16* sb = smem + (stage) * 12288;                                                                  \
    const unsigned ko = (unsigned)(kt) * (FT_K * 2u);                                                     \
    if (!FT_DMA_ON) {} else                                                                               \
    if (w < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rAh, LDS_PTR(sb + (w & 3) * 512), 16, a_src + ko, 0, 0, 0);        \
    else if (NP == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rAl, LDS_PTR(sb + 2048 + (w & 3) * 512), 16, a_src + ko, 0, 0, 0);       \
    if (FT_DMA_ON) __builtin_amdgcn_raw_ptr_buffer_load_lds(rBh, LDS_PTR(sb + 4096 + w * 512), 16, b_src + ko, 0, 0, 0);   \
    if (NP == 3 && FT_DMA_ON) __builtin_amdgcn_raw_ptr_buffer_load_lds(rBl, LDS_PTR(sb + 8192 + w * 512), 16, b_src + ko, 0, 0, 0);     \
  }
    FT_TICK(0);
    FT_GLDS_K(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      FT_PROF_WAITVM; FT_TICK(13);
      __syncthreads();                       // stage kt & 1 has landed (the barrier drains the DMA); the other one is free
      FT_TICK(1);
      if (kt + 1 < nk) FT_GLDS_K((kt + 1) & 1, kt + 1);
      const __bf16* sb = smem + (kt & 1) * 12288;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (!FT_MFMA_ON) continue;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sb + ft_swz(fa_d, 2 * ks + kh));
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sb + 4096 + ft_swz(fb_d, 2 * ks + kh));
        if (NP == 3) {
          const bf16x8 al = FT_ABL == 6 ? ah : *reinterpret_cast<const bf16x8*>(sb + 2048 + ft_swz(fa_d, 2 * ks + kh));
          const bf16x8 bl = FT_ABL == 6 ? bh : *reinterpret_cast<const bf16x8*>(sb + 8192 + ft_swz(fb_d, 2 * ks + kh));
          acc = FT_MFMA(al, bh, acc);
          if (FT_ABL == 7) { acc = FT_MFMA(ah, bh, acc); acc = FT_MFMA(ah, bl, acc); continue; }   // operand-sharing order (experiment)
          acc = FT_MFMA(ah, bl, acc);
        }
        acc = FT_MFMA(ah, bh, acc);
      }
      FT_TICK(2);
    }
    __syncthreads();                         // every wave is done with the operand stages
    FT_TICK(3);
    // ================= phase 2: transpose through LDS (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
    float* ct = reinterpret_cast<float*>(smem);                        // [64][128] fp32
#pragma unroll
    for (int r = 0; r < 16; ++r) ct[(32 * rb + (r & 3) + 8 * (r >> 2) + 4 * kh) * FT_N + 32 * cb + l31] = acc[r];
    __syncthreads();
    FT_TICK(4);
    // ================= phase 3: Adam on 512-byte row segments; W_new -> bf16 (hi, lo) in LDS =================
    __bf16* wn_hi = smem + 16384;
    __bf16* wn_lo = smem + 24576;
#pragma unroll
    for (int pass = 0; pass < 4 / UNITS; ++pass) {
      u32x4 p4[UNITS], m4[UNITS], v4[UNITS];
      unsigned off[UNITS];
#pragma unroll
      for (int i = 0; i < UNITS; ++i) {
        const int u = tid + FT_T * (pass * UNITS + i), row = u >> 5, c4 = u & 31;
        const int gn = n0 + 4 * c4;
        // rows >= rows_valid fall beyond num_records; columns >= N are pushed out of range explicitly
        off[i] = (unsigned)(((long)row * g.ldw + gn) * 4) | ((gn < g.N) ? 0u : 0xFFFFFFF0u);
        p4[i] = __builtin_amdgcn_raw_buffer_load_b128(rP, off[i], 0, NT);
        m4[i] = __builtin_amdgcn_raw_buffer_load_b128(rM, off[i], 0, NT);
        v4[i] = __builtin_amdgcn_raw_buffer_load_b128(rV, off[i], 0, NT);
      }
      FT_TICK(5);
      FT_PROF_WAITVM;
      FT_TICK(6);
#pragma unroll
      for (int i = 0; i < UNITS; ++i) {
        const int u = tid + FT_T * (pass * UNITS + i), row = u >> 5, c4 = u & 31;
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(ct + row * FT_N + 4 * c4);
        const f32x4 pf = __builtin_bit_cast(f32x4, p4[i]), mf = __builtin_bit_cast(f32x4, m4[i]);
        const f32x4 vf = __builtin_bit_cast(f32x4, v4[i]);
        f32x4 po, mo, vo;
        bf16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float pj = pf[j], mj = mf[j], vj = vf[j];
          fx_adam_update(pj, mj, vj, g4[j], ak);
          po[j] = pj;
          mo[j] = mj;
          vo[j] = vj;
          h[j] = (__bf16)po[j];
          if (NP == 3) l[j] = (__bf16)(po[j] - (float)h[j]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, po), rPd, off[i], 0, NT);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mo), rMd, off[i], 0, NT);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vo), rVd, off[i], 0, NT);
        // columns 4 c4 .. 4 c4 + 3 of the tile: K-step block c4 >> 3, 16-byte chunk (c4 & 7) >> 1, half (c4 & 1)
        const int wo = (c4 >> 3) * 2048 + ft_swz(row, (c4 & 7) >> 1) + ((c4 & 1) << 2);
        *reinterpret_cast<bf16x4*>(wn_hi + wo) = h;
        if (NP == 3) *reinterpret_cast<bf16x4*>(wn_lo + wo) = l;
      }
      FT_TICK(7);
    }
    FT_PROF_WAITVM; FT_TICK(14);
    __syncthreads();                         // W_new (hi, lo) visible; ct is dead -> its space stages the next batch
    FT_TICK(8);
    // ================= phase 4: Y[b, m0 + h] += Xn[b, n0 .. n0 + 127] . W_new[h, :]^T =================
    // sub-step q = m * 4 + kb: M-tile m of the next batch against K-step block kb of the tile
#define FT_GLDS_X(stage, q)                                                                               \
  {                                                                                                       \
    __bf16* xb = smem + (stage) * 8192;                                                                   \
    const unsigned xo = (unsigned)(n0 / FT_K + ((q) & 3)) * (128u * MT * FT_K * 2u) + (unsigned)((q) >> 2) * (128u * FT_K * 2u) + x_src; \
    if (FT_DMA_ON) __builtin_amdgcn_raw_ptr_buffer_load_lds(rXh, LDS_PTR(xb + w * 512), 16, xo, 0, 0, 0); \
    if (NP == 3 && FT_DMA_ON) __builtin_amdgcn_raw_ptr_buffer_load_lds(rXl, LDS_PTR(xb + 4096 + w * 512), 16, xo, 0, 0, 0);         \
  }
    FT_GLDS_X(0, 0);
#pragma unroll
    for (int q = 0; q < 4 * MT; ++q) {
      FT_PROF_WAITVM; FT_TICK(15);
      __syncthreads();
      FT_TICK(9);
      if (q + 1 < 4 * MT) FT_GLDS_X((q + 1) & 1, q + 1);
      const int kb = q & 3, m = q >> 2;
      const __bf16* xb = smem + (q & 1) * 8192;
      const __bf16* wh = smem + 16384 + kb * 2048;
      const __bf16* wl = smem + 24576 + kb * 2048;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (!FT_MFMA_ON) continue;
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(xb + ft_swz(fa_f, 2 * ks + kh));
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wh + ft_swz(fb_f, 2 * ks + kh));
        if (NP == 3) {
          const bf16x8 al = FT_ABL == 6 ? ah : *reinterpret_cast<const bf16x8*>(xb + 4096 + ft_swz(fa_f, 2 * ks + kh));
          const bf16x8 bl = FT_ABL == 6 ? bh : *reinterpret_cast<const bf16x8*>(wl + ft_swz(fb_f, 2 * ks + kh));
          yacc[m] = FT_MFMA(al, bh, yacc[m]);
          if (FT_ABL == 7) { yacc[m] = FT_MFMA(ah, bh, yacc[m]); yacc[m] = FT_MFMA(ah, bl, yacc[m]); continue; }
          yacc[m] = FT_MFMA(ah, bl, yacc[m]);
        }
        yacc[m] = FT_MFMA(ah, bh, yacc[m]);
      }
      FT_TICK(10);
    }
    __syncthreads();                         // the next tile's first DMA overwrites the X stages / W_new
    FT_TICK(11);
  }

  // ---- the run's partial sums: rows = batch (dropped beyond Mn by the range check), columns = rows of W
  const __amdgpu_buffer_rsrc_t rY = ft_rsrc(g.Y + (long)c * g.slab_stride, (long)g.Mn * g.M * 4);
  const int hcol = m0 + 32 * hq + l31;
  const unsigned oob = (hcol < g.M) ? 0u : 0xFFFFFFF0u;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int b = 128 * m + 32 * bq + (r & 3) + 8 * (r >> 2) + 4 * kh;
      ft_store32(yacc[m][r], rY, (unsigned)(((long)b * g.M + hcol) * 4) | oob);
    }
  if (c == 0)                                      // slabs this row block has no run for: zeros (the consumer adds S_total slabs)
    for (int z = S; z < g.S_total; ++z) {
      const __amdgpu_buffer_rsrc_t rZ = ft_rsrc(g.Y + (long)z * g.slab_stride, (long)g.Mn * g.M * 4);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int b = 128 * m + 32 * bq + (r & 3) + 8 * (r >> 2) + 4 * kh;
          ft_store32(0.f, rZ, (unsigned)(((long)b * g.M + hcol) * 4) | oob);
        }
    }
  FT_TICK(12);
  FT_PROF_FLUSH;
}


static inline bool ft_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// mapping actually used for a launch: 1 = XCD-grouped interleaved row blocks (K = 3 B: the dY^T tiles of an XCD's workgroups would
// outgrow its L2), 2 = XCD-contiguous (large weights), 0 = plain
// The XCD-contiguous grid is fixed at 8 x (FT_G / 8) workgroups, one run each: it reaches every row block only while an XCD's share of
// row blocks fits its slots (n_out <= 32768); taller weights take the plain mapping, whose grid grows with the row blocks.
static bool ft_contig_ok(int tiles_m) { return tiles_m >= 16 && (tiles_m + 7) / 8 <= FT_G / 8; }
static int ft_mapping(int map_flag, int tiles_m, int batch_padded) {
  if (map_flag == 1) return 0;
  if (map_flag == 2) return 1;
  if (map_flag == 3) return ft_contig_ok(tiles_m) ? 2 : 0;
  if ((long)FT_M * batch_padded * 4 * 60 > (3L << 20)) return 1;
  return ft_contig_ok(tiles_m) ? 2 : 0;
}
static int ft_slabs_for(int mapping, int tiles_m, int tiles_n) {
  if (mapping == 1) return ft_plan(tiles_m, tiles_n, FT_G).S_lo;
  if (mapping == 0) return ft_plan(tiles_m, tiles_n, FT_G).S_total;
  int mx = 1;
  const int rpx = (tiles_m + 7) / 8;
  for (int x = 0; x < 8; ++x) {
    const int nb = tiles_m - x * rpx < rpx ? tiles_m - x * rpx : rpx;
    if (nb <= 0) break;
    const int st = ft_plan(nb, tiles_n, FT_G / 8).S_total;
    mx = st > mx ? st : mx;
  }
  return mx;
}

extern "C" {

#ifdef FT_PROFILE
// out[16] = tick sums (100 MHz) per phase over all workgroups since the last reset; reset != 0 clears them afterwards
int fx_debug_dw_adam_fwd_profile(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ft_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ft_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
// out[1024 * 8]: the wall-clock stamps (100 MHz) of the last launch
int fx_debug_dw_adam_fwd_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ft_stamp), sizeof(unsigned long long) * 1024 * 8) == hipSuccess ? 0 : -1;
}
#endif

// Number of partial-sum slabs fx_linear_dw_adam_fwd_bf16x3 writes for a weight [n_out, k_in].
// Partial-sum slabs a launch with these flags / this padded batch writes (the consumer adds that many; a slab buffer may be larger:
// the launch zero-fills every slab it is given).
int fx_linear_dw_adam_fwd_bf16x3_slabs_ex(int n_out, int k_in, int batch_padded, int flags) {
  if (n_out <= 0 || k_in <= 0) return 0;
  const int tiles_m = (n_out + FT_M - 1) / FT_M, tiles_n = (k_in + FT_N - 1) / FT_N;
  const int forced = (flags >> 8) & 0xFF;
  if (forced) return forced > tiles_n ? tiles_n : forced;
  return ft_slabs_for(ft_mapping((flags >> 1) & 3, tiles_m, batch_padded), tiles_m, tiles_n);
}
// ... with the default flags and a batch of <= 128 rows
int fx_linear_dw_adam_fwd_bf16x3_slabs(int n_out, int k_in) { return fx_linear_dw_adam_fwd_bf16x3_slabs_ex(n_out, k_in, 128, 0); }

static int dw_adam_fwd_impl(float* W, float* adam_m, float* adam_v, float* W_dst, float* m_dst, float* v_dst, const void* dyT_hi,
                            const void* dyT_lo, const void* xT_hi, const void* xT_lo, int batch_padded, int n_out, int k_in, long lddy,
                            long ldx, long ldw, const float* ctrl, const void* xn_hi, const void* xn_lo,
                            long xn_rows_padded, int next_rows, float* y_slabs, long y_slabs_bytes, int flags,
                            hipStream_t stream) {
  FX_REQUIRE(W && adam_m && adam_v && W_dst && m_dst && v_dst && dyT_hi && xT_hi && ctrl && xn_hi && y_slabs,
             "fx_linear_dw_adam_fwd_bf16x3: null pointer");
  // the three `lo` operands come together or not at all: without them the contractions are plain bf16 (one product, fp32 accumulate)
  const bool plain = dyT_lo == nullptr;
  FX_REQUIRE((xT_lo == nullptr) == plain && (xn_lo == nullptr) == plain,
             "fx_linear_dw_adam_fwd_bf16x3: dyT_lo, xT_lo and xn_lo must be all given (split bf16) or all NULL (plain bf16)");
  FX_REQUIRE(((W_dst == W) == (m_dst == adam_m)) && ((W_dst == W) == (v_dst == adam_v)) && ft_aligned16(W_dst) && ft_aligned16(m_dst) && ft_aligned16(v_dst),
             "fx_linear_dw_adam_fwd_bf16x3: the destinations are W / m / v themselves (in place) or three other 16-byte aligned arrays");
  FX_REQUIRE(batch_padded > 0 && batch_padded % FT_K == 0, "fx_linear_dw_adam_fwd_bf16x3: padded batch %d must be a multiple of %d",
             batch_padded, FT_K);
  FX_REQUIRE(n_out > 0 && k_in > 0 && ldw >= k_in && ldw % 4 == 0 && k_in % 4 == 0 && ft_aligned16(W) && ft_aligned16(adam_m) &&
                 ft_aligned16(adam_v),
             "fx_linear_dw_adam_fwd_bf16x3: W / m / v must be 16-byte aligned with k_in and ldw multiples of 4");
  FX_REQUIRE(lddy % 8 == 0 && ldx % 8 == 0 && lddy >= batch_padded && ldx >= batch_padded && ft_aligned16(dyT_hi) &&
                 ft_aligned16(dyT_lo) && ft_aligned16(xT_hi) && ft_aligned16(xT_lo),
             "fx_linear_dw_adam_fwd_bf16x3: transposed operands must be 16-byte aligned with ld %% 8 == 0");
  FX_REQUIRE((xn_rows_padded == 128 || xn_rows_padded == 256 || xn_rows_padded == 384) && next_rows > 0 &&
                 next_rows <= xn_rows_padded && ft_aligned16(xn_hi) && ft_aligned16(xn_lo),
             "fx_linear_dw_adam_fwd_bf16x3: the next batch is a K-blocked split of at most 384 rows (rows padded to 128 / 256 / 384)");
  FX_REQUIRE((long)((k_in + FT_K - 1) / FT_K) * xn_rows_padded * FT_K * 2 < 0xF0000000L, "fx_linear_dw_adam_fwd_bf16x3: next batch exceeds 4 GiB");
  FX_REQUIRE((long)k_in * ldx * 2 < 0xF0000000L && (long)FT_M * ldw * 4 < 0xF0000000L && (long)FT_M * lddy * 2 < 0xF0000000L,
             "fx_linear_dw_adam_fwd_bf16x3: operand row block exceeds 4 GiB");
  DwAdamFwdArgs g{};
  g.W = W; g.m = adam_m; g.v = adam_v;
  g.Wd = W_dst; g.md = m_dst; g.vd = v_dst;
  g.Ah = (const __bf16*)dyT_hi; g.Al = (const __bf16*)dyT_lo;
  g.Bh = (const __bf16*)xT_hi; g.Bl = (const __bf16*)xT_lo;
  g.Xh = (const __bf16*)xn_hi; g.Xl = (const __bf16*)xn_lo;
  g.Y = y_slabs; g.ctrl = ctrl;
  g.M = n_out; g.N = k_in; g.K = batch_padded;
  g.lda = lddy; g.ldb = ldx; g.ldw = ldw;
  g.Mn = next_rows; g.kblocks = (k_in + FT_K - 1) / FT_K;
  g.tiles_m = (n_out + FT_M - 1) / FT_M; g.tiles_n = (k_in + FT_N - 1) / FT_N;
  // flags: bit 0 = non-temporal W / m / v accesses; bits 1-2 = block mapping: 0 auto (XCD-grouped row blocks when the
  // dY^T tiles of an XCD's workgroups would outgrow its L2), 1 = plain, 2 = XCD-grouped, 3 = XCD-contiguous row blocks;
  // bits 8-15: explicit number of runs per row block, uniform (experiments)
  const int nt = flags & 1;
  const int mapping = ft_mapping((flags >> 1) & 3, g.tiles_m, batch_padded);
  g.xcd_group = mapping;
  g.slots = FT_G / 8;
  const FtPlan pl = ft_plan(g.tiles_m, g.tiles_n, FT_G);
  g.S = pl.S_lo;
  g.n_hi = mapping == 0 ? pl.n_hi : 0;
  g.prio = (flags >> 17) & 3;
  const int forced = (flags >> 8) & 0xFF;
  if (forced) {
    g.S = forced > g.tiles_n ? g.tiles_n : forced;
    g.n_hi = 0;
    if (g.xcd_group == 2) g.xcd_group = 0;
  }
  g.slab_stride = (long)next_rows * n_out;
  const int need = fx_linear_dw_adam_fwd_bf16x3_slabs_ex(n_out, k_in, batch_padded, flags);
  FX_REQUIRE(y_slabs_bytes >= (long)need * g.slab_stride * 4, "fx_linear_dw_adam_fwd_bf16x3: slab buffer too small (%ld bytes for %d slabs)",
             y_slabs_bytes, need);
  const long given = y_slabs_bytes / (g.slab_stride * 4);          // every slab of the buffer is written (zeros where a row block has no run)
  g.S_total = given > 255 ? 255 : (int)given;
  long nblk;
  if (g.xcd_group == 2) nblk = 8L * g.slots;
  else if (g.xcd_group == 1) nblk = 8L * ((g.tiles_m + 7) / 8) * g.S;
  else nblk = (long)(g.tiles_m - g.n_hi) * g.S + (long)g.n_hi * (g.S + 1);
  FX_REQUIRE(nblk < (1L << 31), "fx_linear_dw_adam_fwd_bf16x3: grid too large");
  const dim3 grid((unsigned)nblk), blk(FT_T);
  const int mt = (int)(xn_rows_padded / 128);
#define FT_LAUNCH(NTV, MTV, UV) { if (plain) hipLaunchKernelGGL((fx_dw_adam_fwd_kernel<NTV, MTV, UV, 1>), grid, blk, 0, stream, g); \
                                 else hipLaunchKernelGGL((fx_dw_adam_fwd_kernel<NTV, MTV, UV, 3>), grid, blk, 0, stream, g); }
  if (mt == 1) { if (nt) FT_LAUNCH(2, 1, 4) else FT_LAUNCH(0, 1, 4) }
  else if (mt == 2) { if (nt) FT_LAUNCH(2, 2, 2) else FT_LAUNCH(0, 2, 2) }
  else { if (nt) FT_LAUNCH(2, 3, 2) else FT_LAUNCH(0, 3, 2) }
#undef FT_LAUNCH
  return fx_check_launch("fx_linear_dw_adam_fwd_bf16x3");
}

int fx_linear_dw_adam_fwd_bf16x3(float* W, float* adam_m, float* adam_v, const void* dyT_hi, const void* dyT_lo,
                                 const void* xT_hi, const void* xT_lo, int batch_padded, int n_out, int k_in, long lddy,
                                 long ldx, long ldw, const float* ctrl, const void* xn_hi, const void* xn_lo,
                                 long xn_rows_padded, int next_rows, float* y_slabs, long y_slabs_bytes, int flags,
                                 hipStream_t stream) {
  return dw_adam_fwd_impl(W, adam_m, adam_v, W, adam_m, adam_v, dyT_hi, dyT_lo, xT_hi, xT_lo, batch_padded, n_out, k_in, lddy, ldx, ldw,
                          ctrl, xn_hi, xn_lo, xn_rows_padded, next_rows, y_slabs, y_slabs_bytes, flags, stream);
}

// The same step OUT OF PLACE: W / m / v are only read, the updated values go to W_dst / m_dst / v_dst (same shape and pitch), which
// the caller has placed in OTHER memory partitions than the sources -- every partition then sees only reads or only writes during the
// launch -- and swaps with the sources for the next step (profiles/r06_outofplace.txt).
int fx_linear_dw_adam_fwd_bf16x3_oop(float* W_dst, float* m_dst, float* v_dst, const float* W, const float* adam_m, const float* adam_v,
                                     const void* dyT_hi, const void* dyT_lo, const void* xT_hi, const void* xT_lo, int batch_padded,
                                     int n_out, int k_in, long lddy, long ldx, long ldw, const float* ctrl, const void* xn_hi,
                                     const void* xn_lo, long xn_rows_padded, int next_rows, float* y_slabs, long y_slabs_bytes,
                                     int flags, hipStream_t stream) {
  FX_REQUIRE(W_dst != W && m_dst != adam_m && v_dst != adam_v, "fx_linear_dw_adam_fwd_bf16x3_oop: destinations must differ from the sources");
  return dw_adam_fwd_impl(const_cast<float*>(W), const_cast<float*>(adam_m), const_cast<float*>(adam_v), W_dst, m_dst, v_dst, dyT_hi,
                          dyT_lo, xT_hi, xT_lo, batch_padded, n_out, k_in, lddy, ldx, ldw, ctrl, xn_hi, xn_lo, xn_rows_padded,
                          next_rows, y_slabs, y_slabs_bytes, flags, stream);
}

// ---- placement probe -----------------------------------------------------------------------------------------------------------
// The GEMM-free twin of fx_dw_adam_fwd_kernel's memory schedule: the same runs of 64 x 128 tiles, two workgroups per CU, W / m / v read
// with 16-byte non-temporal loads and written back UNCHANGED (idempotent: safe on live parameters).  Its time depends on WHERE the three
// arrays landed in physical memory: 400 to 494 us for the [5000, 20000] weight on one and the same MI355X, by nothing but the
// allocation history of the process (scripts/adamprobe.hip `r placement`, profiles/r04_placement.txt) -- what rounds 2-4 had
// taken for two kinds of box.  The host allocates a few candidates, probes each and keeps the fastest (engine.ParamStore).
__global__ __launch_bounds__(512, 2) void fx_placement_probe_kernel(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V,
                                                                    float* __restrict__ Wd, float* __restrict__ Md, float* __restrict__ Vd,
                                                                    int H, int F, long ld, int S, int n_hi) {
  __shared__ char pad[65536];                                      // the fused kernel's LDS footprint: two workgroups per CU
  if (threadIdx.x == 9999) pad[threadIdx.x] = 1;
  const int tiles_m = (H + 63) / 64, tiles_n = (F + 127) / 128;
  int tm, c, runs = S;
  const int n_lo = tiles_m - n_hi, low = n_lo * S;
  if ((int)blockIdx.x < low) { tm = blockIdx.x % n_lo; c = blockIdx.x / n_lo; }
  else { const int b2 = blockIdx.x - low; tm = n_lo + b2 % n_hi; c = b2 / n_hi; runs = S + 1; }
  for (int tn = c; tn < tiles_n; tn += runs) {
    f32x4 p[4], m[4], v[4];
    long off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = threadIdx.x + 512 * i, r = u >> 5, c4 = u & 31;
      const int row = tm * 64 + r, col = tn * 128 + 4 * c4;
      off[i] = (row < H && col < F) ? ((long)row * ld + col) >> 2 : -1;
      if (off[i] >= 0) {
        p[i] = __builtin_nontemporal_load((const f32x4*)W + off[i]);
        if (M) m[i] = __builtin_nontemporal_load((const f32x4*)M + off[i]);
        if (V) v[i] = __builtin_nontemporal_load((const f32x4*)V + off[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (off[i] < 0) continue;
      __builtin_nontemporal_store(p[i], (f32x4*)Wd + off[i]);
      if (M) __builtin_nontemporal_store(m[i], (f32x4*)Md + off[i]);
      if (V) __builtin_nontemporal_store(v[i], (f32x4*)Vd + off[i]);
    }
  }
}

// Many slabs, small output (the VAE's dz: 3 shares + 2 x 40 split-K partial sums of [128, 64]): with one thread per four outputs the
// ordered sum is ~20 dependent rounds of loads on 8 workgroups (23 us).  Here RP_SUB neighbouring lanes share four outputs, each sums a
// contiguous range of the slabs, and the ranges are combined in range order -- a fixed order (deterministic), but not the serial one.
#define RP_SUB 8
__global__ __launch_bounds__(256) void fx_reduce_slabs_par_kernel(float* __restrict__ Y, const float* __restrict__ slabs,
                                                                  const float* __restrict__ bias, int N, long ldy, int n_slabs,
                                                                  long slab_stride, unsigned total4) {
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  const unsigned unit = gid / RP_SUB, sub = gid % RP_SUB;
  const unsigned u = unit < total4 ? unit : total4 - 1;           // every lane takes part in the shuffles
  const int per = (n_slabs + RP_SUB - 1) / RP_SUB, z0 = min(n_slabs, (int)sub * per), z1 = min(n_slabs, z0 + per);
  const float* src = slabs + ((long)u << 2);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int z = z0;
  for (; z + 4 <= z1; z += 4) {
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(src + (long)(z + j) * slab_stride);
#pragma unroll
    for (int j = 0; j < 4; ++j) s += v[j];
  }
  for (; z < z1; ++z) s += *reinterpret_cast<const f32x4*>(src + (long)z * slab_stride);
  const int base = (threadIdx.x & 63) & ~(RP_SUB - 1);
  f32x4 tot = s;
#pragma unroll
  for (int q = 1; q < RP_SUB; ++q) {
    f32x4 sq;
#pragma unroll
    for (int j = 0; j < 4; ++j) sq[j] = __shfl(s[j], base + q, 64);
    tot += sq;
  }
  if (sub == 0 && unit < total4) {
    const unsigned n4 = (unsigned)N >> 2, m = u / n4, n = (u - m * n4) << 2;
    if (bias) tot += *reinterpret_cast<const f32x4*>(bias + n);
    *reinterpret_cast<f32x4*>(Y + (long)m * ldy + n) = tot;
  }
}

// Y[M, N] = sum_z slabs[z][M][N] (+ bias[N]) in slab order: consumes the partial sums of fx_linear_dw_adam_fwd_bf16x3
// (and of any other split-K producer that leaves its slabs unreduced).
int fx_reduce_slabs(float* Y, const float* slabs, const float* bias, int M, int N, long ldy, int n_slabs, long slab_stride,
                    hipStream_t stream) {
  FX_REQUIRE(Y && slabs && M > 0 && N > 0 && n_slabs > 0 && ldy >= N && slab_stride >= (long)M * N, "fx_reduce_slabs: bad args");
  fx_launch_reduce_slabs(Y, slabs, bias, M, N, ldy, n_slabs, slab_stride, 0, stream);
  return fx_check_launch("fx_reduce_slabs");
}

// One pass of the fused kernel's W / m / v traffic pattern without the GEMMs; contents unchanged.  W, m, v [n_out, k_in] with row pitch
// ldw (k_in % 4 == 0, ldw % 4 == 0, 16-byte aligned bases).  The caller times it (HIP events) to rate a placement of the three arrays.
int fx_placement_probe(float* W, float* m, float* v, int n_out, int k_in, long ldw, hipStream_t stream) {
  FX_REQUIRE(W && n_out > 0 && k_in > 0 && ldw >= k_in, "fx_placement_probe: bad args");       // m / v may be NULL: one array on its own
  FX_REQUIRE(k_in % 4 == 0 && ldw % 4 == 0 && ((((uintptr_t)W) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0,
             "fx_placement_probe: k_in and ldw must be multiples of 4 and the bases 16-byte aligned");
  const int tiles_m = (n_out + 63) / 64, tiles_n = (k_in + 127) / 128;
  const FtPlan pl = ft_plan(tiles_m, tiles_n, FT_G);
  const int grid = (tiles_m - pl.n_hi) * pl.S_lo + pl.n_hi * (pl.S_lo + 1);
  hipLaunchKernelGGL(fx_placement_probe_kernel, dim3(grid), dim3(512), 0, stream, W, m, v, W, m, v, n_out, k_in, ldw, pl.S_lo, pl.n_hi);
  return fx_check_launch("fx_placement_probe");
}

// The out-of-place twin: W / m / v are read, the same values are written to W_dst / m_dst / v_dst (same shape and pitch; m, v and their
// destinations may be NULL together) -- rates a (source partitions, destination partitions) layout for fx_linear_dw_adam_fwd_bf16x3_oop.
int fx_placement_probe_oop(float* W_dst, float* m_dst, float* v_dst, const float* W, const float* m, const float* v, int n_out, int k_in,
                           long ldw, hipStream_t stream) {
  FX_REQUIRE(W && W_dst && n_out > 0 && k_in > 0 && ldw >= k_in && (m == nullptr) == (m_dst == nullptr) && (v == nullptr) == (v_dst == nullptr),
             "fx_placement_probe_oop: bad args");
  FX_REQUIRE(k_in % 4 == 0 && ldw % 4 == 0 &&
                 ((((uintptr_t)W) | ((uintptr_t)m) | ((uintptr_t)v) | ((uintptr_t)W_dst) | ((uintptr_t)m_dst) | ((uintptr_t)v_dst)) & 15) == 0,
             "fx_placement_probe_oop: k_in and ldw must be multiples of 4 and the bases 16-byte aligned");
  const int tiles_m = (n_out + 63) / 64, tiles_n = (k_in + 127) / 128;
  const FtPlan pl = ft_plan(tiles_m, tiles_n, FT_G);
  const int grid = (tiles_m - pl.n_hi) * pl.S_lo + pl.n_hi * (pl.S_lo + 1);
  hipLaunchKernelGGL(fx_placement_probe_kernel, dim3(grid), dim3(512), 0, stream, const_cast<float*>(W), const_cast<float*>(m),
                     const_cast<float*>(v), W_dst, m_dst, v_dst, n_out, k_in, ldw, pl.S_lo, pl.n_hi);
  return fx_check_launch("fx_placement_probe_oop");
}

// The same sum for MANY slabs of a SMALL output (N, ldy, slab_stride multiples of 4, 16-byte aligned bases): 8 lanes per four outputs, each
// over a contiguous range of slabs, combined in range order (deterministic; the order differs from fx_reduce_slabs').
int fx_reduce_slabs_par(float* Y, const float* slabs, const float* bias, int M, int N, long ldy, int n_slabs, long slab_stride,
                        hipStream_t stream) {
  FX_REQUIRE(Y && slabs && M > 0 && N > 0 && n_slabs > 0 && ldy >= N && slab_stride >= (long)M * N, "fx_reduce_slabs_par: bad args");
  FX_REQUIRE(N % 4 == 0 && ldy % 4 == 0 && slab_stride % 4 == 0 && (long)M * N < (1L << 28) &&
             ((((uintptr_t)Y) | ((uintptr_t)slabs) | ((uintptr_t)bias)) & 15) == 0,
             "fx_reduce_slabs_par: N, ldy and the slab stride must be multiples of 4 and the bases 16-byte aligned");
  const unsigned total4 = (unsigned)(((long)M * N) >> 2);
  const unsigned threads = total4 * RP_SUB;
  hipLaunchKernelGGL(fx_reduce_slabs_par_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, Y, slabs, bias, N, ldy, n_slabs,
                     slab_stride, total4);
  return fx_check_launch("fx_reduce_slabs_par");
}

}  // extern "C"
