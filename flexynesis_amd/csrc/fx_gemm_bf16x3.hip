// Split-bf16 ("bf16x3") MFMA GEMMs for the WIDE layers of the flexynesis hot path (gfx950 / CDNA4).
//
// Why: the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate, which makes the
// wide forward GEMM (B x 20000 x 5000) compute-bound and leaves the fused dW+Adam kernel MFMA-limited
// (profiles/r01_a_*).  Here every fp32 operand x is split into two bf16 terms, x ~= hi + lo
// (hi = rne_bf16(x), lo = rne_bf16(x - hi), 16 significand bits together), and each product is
// evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 3/16 of the
// fp32-MFMA cost, per-product relative error <= ~2^-16 (random sign); measured per-step loss error vs the
// fp32 CPU oracle <= 5e-7 (scripts/precision_study.py).  Master weights, Adam moments, accumulation and
// everything narrow stay fp32.
//
//   fx_split_bf16 / fx_split_bf16_t   fp32 [R,C] -> (hi, lo) bf16 [R,Cp] or transposed [C,Rp], zero padded
//   fx_linear_fwd_bf16x3              Y[M,N] = X[M,K] . W[N,K]^T  X pre-split (re-read by every column
//                                     tile, from L2), W fp32 streamed ONCE from HBM and split in registers
//                                     (more than 128 or at most 64 batch rows: fx_fwd_bf16x3_reg_kernel below)
//   fx_linear_dw_adam_bf16x3          W[N,K] -= Adam(clip * dY^T X): both operands pre-split + transposed
//                                     (dYT [N,Bp], XT [K,Bp]); W/m/v streamed once (24 B/param)
//
// Structure: tile 128 x (32*WN) x 32 with 2*WN waves (each 64x32 = two 32x32 MFMA blocks); WN = 4
// (128x128, 512 threads) halves the L2->LDS re-reads of the shared activation operand, which an ablation
// showed to cost 36 % of the 128x64 forward kernel.  LDS double buffer, global loads prefetched TWO tiles
// ahead in named registers, one barrier per K-step.  ALL global accesses are raw buffer loads/stores
// through SRSRC descriptors: the hardware range check zero-fills out-of-range rows and drops out-of-range
// stores, so there is no per-lane branch around a memory instruction (a "load or zero" select makes hipcc
// branch around every load and drain vmcnt(0) each time -- measured 4x slower; cdna guide section 5 trap
// (c)).  LDS rows are 64 B (32 bf16); the 16-byte chunk index is XOR-swizzled with (row>>2)&3 so the
// 16-lane groups of ds_read_b128 hit 16 distinct 4-bank slots (conflict-free) without padding.
#include "fx_common.h"
#include "fx_reduce.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define TM 128
#define TK 32

enum { XEPI_STORE = 0, XEPI_ADAM = 1 };

struct XGemmArgs {
  const __bf16* Ahi;  // a_rp == 0: [M, K] k-contiguous, ld = lda (bf16 elements, multiple of 8), K multiple of 32
  const __bf16* Alo;  // a_rp  > 0: K-BLOCKED [K/32][a_rp rows][32] -- one 128-row K-step tile is 8 KB contiguous
  long a_rp;
  const float* Bf;    // B as fp32 [N, Ktrue] (split in-kernel) -- or null when pre-split
  const __bf16* Bhi;  // B pre-split [N, K]
  const __bf16* Blo;
  float* C;
  int M, N, K, Ktrue;
  long lda, ldb, ldc;
  int splitk, kchunk;
  int n_fast;         // block index walks N tiles first (set when M > N)
  int xcd_blk;        // > 0: XCD-partitioned, L2-blocked tile order with groups of xcd_blk tiles (see the kernel)
  long slab_stride;
  float* adam_m;
  float* adam_v;
  const float* ctrl;
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * TK + ((chunk ^ ((row >> 2) & 3)) << 3); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t fx_rsrc(const void* p, long bytes) {
  const unsigned n = bytes > 0xFFFFFFF0L ? 0xFFFFFFF0u : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);
}
// AUX = cache-policy bits of the buffer instruction: 0 = default, 2 = nt (data streamed exactly once should
// not displace the L2-resident activation operand).
// the same for a descriptor the compiler cannot prove wave-uniform on its own (it would wrap every load in a
// readfirstlane "waterfall" loop): base and size are made scalar explicitly
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fx_rsrc_uniform(const void* p, long bytes) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const unsigned n = __builtin_amdgcn_readfirstlane(bytes > 0xFFFFFFF0L ? 0xFFFFFFF0u : (unsigned)bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}
template <int AUX = 0>
__device__ __forceinline__ u32x4 bld128(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX);
}
__device__ __forceinline__ unsigned bld32u(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
}
template <int AUX = 0>
__device__ __forceinline__ float bld32f(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void bst32f(float v, __amdgpu_buffer_rsrc_t r, unsigned off) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, AUX);
}

// split 4 fp32 (as raw u32x4) into 4 hi + 4 lo bf16 and store them 8 bytes each
template <bool LO = true>
__device__ __forceinline__ void split_store4(const u32x4 raw, __bf16* hi_dst, __bf16* lo_dst) {
  // NB: bit_cast the WHOLE vector; __builtin_bit_cast(float, raw[j]) on a vector element is miscompiled by
  // hipcc 7.2 (every j reads element 0 and the load is narrowed to one dword).
  const f32x4 f = __builtin_bit_cast(f32x4, raw);
  bf16x4 h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v = f[j];
    h[j] = (__bf16)v;
    if (LO) l[j] = (__bf16)(v - (float)h[j]);
  }
  *reinterpret_cast<bf16x4*>(hi_dst) = h;
  if (LO) *reinterpret_cast<bf16x4*>(lo_dst) = l;
}

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// WN = number of 32-column wave slices (2 -> 128x64 tile / 256 threads, 4 -> 128x128 tile / 512 threads)
// B_KN: the fp32 B operand is stored [K, N] (N contiguous) instead of [N, K] -- the data-gradient contraction
// dX = dY . W through a wide layer (W [out=K, in=N]).  Each thread then loads the 8 consecutive k of ONE column n as
// eight coalesced dword loads (64 lanes = 256 contiguous bytes of a W row) and owns exactly one 16-byte k-chunk of
// the [n][k] LDS tile, so the MFMA side is unchanged and no LDS transpose is needed.
// NP = products per contraction: 3 = split bf16 (hi hi + hi lo + lo hi), 1 = plain bf16 (hi hi only; the `lo` operands are neither
// loaded nor produced: the throughput mode, selected by the host when the caller passes NULL for them).
template <bool B_F32, int EPI, int NT, int WN, bool B_KN = false, int NP = 3>
__global__ __launch_bounds__(128 * WN) void fx_gemm_bf16x3_kernel(XGemmArgs g) {
  static_assert(!B_KN || (B_F32 && WN == 4), "B_KN needs the fp32 B operand and the 128x128 tile");
  constexpr int TN = 32 * WN, T = 128 * WN;
  constexpr int STAGE_ELEMS = 2 * TM * TK + 2 * TN * TK;
  constexpr bool A2 = (T == 256);  // two A chunks per array per thread (else one)
  // WN = 8 (128 x 256 tile, 1024 threads, one workgroup per CU: the activation tile is re-read from the L2 once per 32 KB of W instead of
  // once per 16 KB): only the first 512 threads fetch A -- the others issue the same instructions with an out-of-range offset (no
  // memory traffic: the descriptor's range check answers) and stash into a scratch slot of their own (no branch around a load or a
  // store: see the note above the K loop)
  constexpr bool AHALF = (T == 1024);
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * STAGE_ELEMS + (AHALF ? 1024 : 0)];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid & 1, wc = wid >> 1;
  const int tiles_m = (g.M + TM - 1) / TM;
  int lin = blockIdx.x;
  const int z = lin % g.splitk;
  lin /= g.splitk;
  // Consecutive workgroups walk the SMALLER operand's tiles: its whole split (2.5 MB at cfg2/cfg3 shapes) stays in L2
  // while each tile of the larger one is fetched once and reused by the following workgroups.  With the fixed
  // M-fastest order the decoders' weights (out = 20000 rows > in = 5000) re-streamed their 10 MB dY^T operand 40 times
  // and their dW+Adam launches took 508-632 us instead of ~430.
  int tm, tn;
  if (g.xcd_blk > 0) {
    // XCD-partitioned, L2-blocked order (dW+Adam).  Workgroup ids go round-robin over the 8 XCDs and every XCD has its
    // own 4 MB L2, so with a plain linear order each operand tile is fetched once per XCD, and when the smaller operand
    // no longer fits one L2 (the triplet network's 3B-row operands: 11.5 MB) it is re-fetched for every tile of the
    // other one -- PMC: 9.0 GB of HBM traffic per launch at cfg4 for 5.4 GB of W / m / v.  Here XCD x owns a contiguous
    // 1/8 of the tiles along the longer dimension; inside it walks groups of xcd_blk tiles of the shorter dimension
    // (the group's operand rows, <= 1.5 MB, stay L2-resident) against each of its own tiles, which is thereby reused
    // xcd_blk times.
    const int tiles_n = (g.N + TN - 1) / TN;
    const int xcd = lin & 7, j = lin >> 3, G = g.xcd_blk;
    const int ig = j % G, t2 = j / G;
    if (tiles_n >= tiles_m) {          // split N over the XCDs, group M
      const int per = (tiles_n + 7) / 8;
      tn = xcd * per + t2 % per;
      tm = (t2 / per) * G + ig;
    } else {                           // split M over the XCDs, group N
      const int per = (tiles_m + 7) / 8;
      tm = xcd * per + t2 % per;
      tn = (t2 / per) * G + ig;
    }
    if (tm >= tiles_m || tn >= tiles_n) return;       // padding of the index space (before any barrier)
  } else if (g.n_fast) {
    const int tiles_n = (g.N + TN - 1) / TN;
    tn = lin % tiles_n;
    tm = lin / tiles_n;
  } else {
    tm = lin % tiles_m;
    tn = lin / tiles_m;
  }
  const int m0 = tm * TM, n0 = tn * TN;
  const int k_begin = z * g.kchunk;
  const int k_end = min(g.K, k_begin + g.kchunk);
  const int nk = (k_end > k_begin) ? (k_end - k_begin) / TK : 0;  // K, kchunk are multiples of TK

  // ---- buffer descriptors: rows >= M (or >= N) fall beyond num_records and read as zero
  const long a_bytes = g.a_rp ? (long)g.K * g.a_rp * 2 : (long)g.M * g.lda * 2;
  const __amdgpu_buffer_rsrc_t rAh = fx_rsrc(g.Ahi, a_bytes), rAl = fx_rsrc(g.Alo, a_bytes);
  // The fp32 weight operand is addressed through a descriptor REBASED to this workgroup's rows (row-major [N, K]: the
  // N tile; [K, N]: the K slice), so the 32-bit buffer offsets only span one tile / slice and the weight itself may
  // exceed 4 GiB (config.py:7-15 allows hidden_dim_factor 0.5: 50000 features -> a 5 GB layer_1.weight).
  const long brow0 = B_KN ? (long)k_begin : (long)n0;
  const long brows = B_KN ? (long)max(min(g.Ktrue, k_end) - k_begin, 0) : (long)max(min(TN, g.N - n0), 0);
  const __amdgpu_buffer_rsrc_t rB0 = B_F32 ? fx_rsrc(g.Bf + brow0 * g.ldb, brows * g.ldb * 4) : fx_rsrc(g.Bhi, (long)g.N * g.ldb * 2);
  const __amdgpu_buffer_rsrc_t rB1 = B_F32 ? rB0 : fx_rsrc(g.Blo, (long)g.N * g.ldb * 2);

  // ---- per-thread constant addressing (bytes) and LDS destinations (elements)
  const int a_row0 = tid >> 2, a_c = tid & 3, a_row1 = a_row0 + 64;  // a_row1 only used when A2
  // K-blocked A (the wide forward's activations): rows of a K-step tile are 64 B apart, K-steps a_rp*64 B apart.
  // Row-major, a 128-row x 32-k tile is 128 half cache lines 2*lda bytes apart: every line is pulled from L2 twice
  // and the forward ran at 3.3 TB/s; blocked it is 64 full lines and reaches 4.9 (scripts/fwdprobe.hip).
  const long a_ld = g.a_rp ? TK : g.lda;
  const unsigned a_off0 = (unsigned)(((long)(m0 + a_row0) * a_ld + 8 * a_c) * 2);
  const unsigned a_off1 = (unsigned)(((long)(m0 + a_row1) * a_ld + 8 * a_c) * 2);
  const bool a_on = !AHALF || tid < 512;
  const unsigned a_oob = a_on ? 0u : 0xFFFFFFF0u;
  const int a_lds0 = swz(a_row0, a_c), a_lds1 = swz(a_row1, a_c);
  // (stash destinations relative to a stage's base; the idle half's slots lie behind both stages, 16 bytes per lane, hi and lo planes)
  const int a_dump = 2 * STAGE_ELEMS + (tid & 63) * 8;
  unsigned b_off0, b_off1;
  int b_lds0, b_lds1;
  if (B_KN) {   // thread (n = tid & 127, kq = tid >> 7): W[k0 + 8 kq + j][n0 + n], j = 0..7  -> LDS chunk kq of row n
    const int n = tid & 127, kq = tid >> 7;
    b_off0 = (unsigned)(((long)(8 * kq) * g.ldb + n0 + n) * 4);
    b_off1 = b_off0 + (unsigned)(4 * g.ldb * 4);
    b_lds0 = swz(n, kq);
    b_lds1 = b_lds0 + 4;
  } else if (B_F32) {  // TN rows x 8 float4 per row; thread handles (n = tid>>3, k4 = tid&7) and n + TN/2
    const int n = tid >> 3, k4 = tid & 7;
    b_off0 = (unsigned)(((long)n * g.ldb + 4 * k4) * 4);                 // rows relative to the rebased descriptor
    b_off1 = (unsigned)(((long)(n + TN / 2) * g.ldb + 4 * k4) * 4);
    b_lds0 = swz(n, k4 >> 1) + ((k4 & 1) << 2);
    b_lds1 = swz(n + TN / 2, k4 >> 1) + ((k4 & 1) << 2);
  } else {      // TN rows x 4 chunks: one 16-byte chunk of hi and of lo per thread
    const int row = tid >> 2, c = tid & 3;
    b_off0 = b_off1 = (unsigned)(((long)(n0 + row) * g.ldb + 8 * c) * 2);
    b_lds0 = b_lds1 = swz(row, c);
  }
  const unsigned a_step = g.a_rp ? (unsigned)g.a_rp * (TK * 2u) : TK * 2u;
  const unsigned b_step = B_KN ? (unsigned)(TK * g.ldb * 4) : TK * (B_F32 ? 4u : 2u);
  const unsigned a_kb = (unsigned)(k_begin / TK) * a_step;
  const unsigned b_kb = B_KN ? 0u : (unsigned)k_begin * (B_F32 ? 4u : 2u);        // [K, N]: the descriptor starts at k_begin
  const unsigned b_row = (unsigned)(g.ldb * 4);      // B_KN: bytes between consecutive k

  // ---- two register stages (named: no arrays, no references -> nothing can land in scratch)
  u32x4 s0_ah0, s0_ah1, s0_al0, s0_al1, s0_b0, s0_b1;
  u32x4 s1_ah0, s1_ah1, s1_al0, s1_al1, s1_b0, s1_b1;

#define LOAD_STAGE(P, kt)                                              \
  {                                                                    \
    const unsigned ka = a_kb + (unsigned)(kt) * a_step;                \
    const unsigned kb = b_kb + (unsigned)(kt) * b_step;                \
    P##_ah0 = bld128(rAh, (a_off0 + ka) | a_oob);                      \
    if (NP == 3) P##_al0 = bld128(rAl, (a_off0 + ka) | a_oob);         \
    if (A2) {                                                          \
      P##_ah1 = bld128(rAh, a_off1 + ka);                              \
      if (NP == 3) P##_al1 = bld128(rAl, a_off1 + ka);                 \
    }                                                                  \
    if (B_KN) {                                                        \
      P##_b0 = u32x4{bld32u(rB0, b_off0 + kb), bld32u(rB0, b_off0 + kb + b_row), bld32u(rB0, b_off0 + kb + 2 * b_row),      \
                     bld32u(rB0, b_off0 + kb + 3 * b_row)};            \
      P##_b1 = u32x4{bld32u(rB0, b_off1 + kb), bld32u(rB0, b_off1 + kb + b_row), bld32u(rB0, b_off1 + kb + 2 * b_row),      \
                     bld32u(rB0, b_off1 + kb + 3 * b_row)};            \
    } else {                                                           \
      P##_b0 = bld128<B_F32 ? NT : 0>(rB0, b_off0 + kb);               \
      if (B_F32 || NP == 3) P##_b1 = bld128<B_F32 ? NT : 0>(B_F32 ? rB0 : rB1, b_off1 + kb); \
    }                                                                  \
  }
#define STASH_STAGE(P, buf)                                            \
  {                                                                    \
    __bf16* base = smem + (buf) * STAGE_ELEMS;                         \
    *reinterpret_cast<u32x4*>(a_on ? base + a_lds0 : smem + a_dump) = P##_ah0;                \
    if (NP == 3) *reinterpret_cast<u32x4*>(a_on ? base + TM * TK + a_lds0 : smem + a_dump + 512) = P##_al0;      \
    if (A2) {                                                          \
      *reinterpret_cast<u32x4*>(base + a_lds1) = P##_ah1;              \
      if (NP == 3) *reinterpret_cast<u32x4*>(base + TM * TK + a_lds1) = P##_al1;    \
    }                                                                  \
    __bf16* bh = base + 2 * TM * TK;                                   \
    __bf16* bl = bh + TN * TK;                                         \
    if (B_F32) {                                                       \
      split_store4<NP == 3>(P##_b0, bh + b_lds0, bl + b_lds0);         \
      split_store4<NP == 3>(P##_b1, bh + b_lds1, bl + b_lds1);         \
    } else {                                                           \
      *reinterpret_cast<u32x4*>(bh + b_lds0) = P##_b0;                 \
      if (NP == 3) *reinterpret_cast<u32x4*>(bl + b_lds1) = P##_b1;    \
    }                                                                  \
  }

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

  const int arow = wr * 64 + (lane & 31), brow = wc * 32 + (lane & 31), kh = lane >> 5;
  const int fa0 = swz(arow, kh), fa1 = swz(arow, 2 + kh);          // k16-step 0 / 1 chunk of the A rows
  const int fa0b = swz(arow + 32, kh), fa1b = swz(arow + 32, 2 + kh);
  const int fb0 = swz(brow, kh), fb1 = swz(brow, 2 + kh);

#define COMPUTE(buf)                                                                         \
  {                                                                                          \
    const __bf16* Ah = smem + (buf) * STAGE_ELEMS;                                           \
    const __bf16* Al = Ah + TM * TK;                                                         \
    const __bf16* Bh = Ah + 2 * TM * TK;                                                     \
    const __bf16* Bl = Bh + TN * TK;                                                         \
    {                                                                                        \
      const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(Ah + fa0);                         \
      const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(Ah + fa0b);                        \
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Bh + fb0);                          \
      if (NP == 3) {                                                                         \
        const bf16x8 al0 = *reinterpret_cast<const bf16x8*>(Al + fa0);                       \
        const bf16x8 al1 = *reinterpret_cast<const bf16x8*>(Al + fa0b);                      \
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Bl + fb0);                        \
        acc0 = MFMA_BF16(al0, bh, acc0); acc1 = MFMA_BF16(al1, bh, acc1);                    \
        acc0 = MFMA_BF16(ah0, bl, acc0); acc1 = MFMA_BF16(ah1, bl, acc1);                    \
      }                                                                                      \
      acc0 = MFMA_BF16(ah0, bh, acc0); acc1 = MFMA_BF16(ah1, bh, acc1);                      \
    }                                                                                        \
    {                                                                                        \
      const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(Ah + fa1);                         \
      const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(Ah + fa1b);                        \
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Bh + fb1);                          \
      if (NP == 3) {                                                                         \
        const bf16x8 al0 = *reinterpret_cast<const bf16x8*>(Al + fa1);                       \
        const bf16x8 al1 = *reinterpret_cast<const bf16x8*>(Al + fa1b);                      \
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Bl + fb1);                        \
        acc0 = MFMA_BF16(al0, bh, acc0); acc1 = MFMA_BF16(al1, bh, acc1);                    \
        acc0 = MFMA_BF16(ah0, bl, acc0); acc1 = MFMA_BF16(ah1, bl, acc1);                    \
      }                                                                                      \
      acc0 = MFMA_BF16(ah0, bh, acc0); acc1 = MFMA_BF16(ah1, bh, acc1);                      \
    }                                                                                        \
  }

  // The loads and LDS stashes inside the loop are UNCONDITIONAL: a branch around LOAD_STAGE makes hipcc's
  // waitcnt insertion assume the not-taken path at the join, i.e. it emits vmcnt(0/1) before the stash and
  // drains the tile that was just requested -- prefetch depth 2 silently becomes depth 1 (seen in the .s).
  // Tiles past this workgroup's K-slice are therefore fetched (at most 3, harmless: the buffer range check
  // returns zeros beyond the tensor) and stashed into a buffer that is never computed on.
  LOAD_STAGE(s0, 0);
  LOAD_STAGE(s1, 1);
  STASH_STAGE(s0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    LOAD_STAGE(s0, kt + 2);
    COMPUTE(0);
    STASH_STAGE(s1, 1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    LOAD_STAGE(s1, kt + 3);
    COMPUTE(1);
    STASH_STAGE(s0, 0);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // Buffer stores drop rows >= M by the range check; columns >= N are pushed out of range explicitly.
  const int n = n0 + wc * 32 + (lane & 31);
  const unsigned oob = (n < g.N) ? 0u : 0xFFFFFFF0u;     // OR-ed into the byte offset: always >= num_records
  if (EPI == XEPI_STORE) {
    const __amdgpu_buffer_rsrc_t rC = fx_rsrc(g.C + (long)z * g.slab_stride, (long)g.M * g.ldc * 4);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int mbase = m0 + wr * 64 + blk * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        const unsigned off = (unsigned)(((long)m * g.ldc + n) * 4) | oob;
        bst32f(blk == 0 ? acc0[r] : acc1[r], rC, off);
      }
    }
  } else {
#ifdef FX_EPI_PRIO      /* experiment (scripts/build_variant.py): the workgroup that is in its memory phase outranks the one in its K-loop */
    __builtin_amdgcn_s_setprio(FX_EPI_PRIO);
#endif
    // descriptors rebased to the tile's first row: offsets stay below TM * ldc * 4 bytes whatever the weight's size
    const long bytes = (long)max(min(TM, g.M - m0), 0) * g.ldc * 4;
    const long rbase = (long)m0 * g.ldc;
    const __amdgpu_buffer_rsrc_t rP = fx_rsrc(g.C + rbase, bytes), rM = fx_rsrc(g.adam_m + rbase, bytes), rV = fx_rsrc(g.adam_v + rbase, bytes);
    const FxAdamK ak = fx_adam_consts(g.ctrl[FXC_LR], g.ctrl[FXC_BC1], g.ctrl[FXC_BC2_SQRT], g.ctrl[FXC_CLIP_COEF]);
    if ((g.N & 3) == 0 && (g.ldc & 3) == 0) {
      // Row-contiguous 16-byte streaming of W/m/v: the dW tile is transposed through LDS (the operand
      // buffers are free after the K loop) so that consecutive lanes own consecutive float4 of a weight row --
      // each wave touches 2 rows x 512 B instead of 32 scattered 128-byte segments (the HBM probe in
      // scripts/membench.hip: 5.9 vs 4.9 TB/s) with 4x fewer memory instructions.
      float* ct = reinterpret_cast<float*>(smem);                       // [TM][TN] fp32
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int rbase = wr * 64 + blk * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ct[(rbase + (r & 3) + 8 * (r >> 2)) * TN + wc * 32 + (lane & 31)] = (blk == 0 ? acc0[r] : acc1[r]);
      }
      __syncthreads();
      constexpr int UPR = TN / 4;                                       // float4 units per tile row
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        u32x4 p4[4], m4[4], v4[4];
        unsigned off[4];
        int lidx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int u = tid + T * (half * 4 + i), row = u / UPR, c4 = u % UPR;
          const int gn = n0 + 4 * c4;
          lidx[i] = row * TN + 4 * c4;
          off[i] = (unsigned)(((long)row * g.ldc + gn) * 4) | ((gn < g.N) ? 0u : 0xFFFFFFF0u);
          p4[i] = bld128<NT>(rP, off[i]);
          m4[i] = bld128<NT>(rM, off[i]);
          v4[i] = bld128<NT>(rV, off[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(ct + lidx[i]);
          const f32x4 pf = __builtin_bit_cast(f32x4, p4[i]), mf = __builtin_bit_cast(f32x4, m4[i]);
          const f32x4 vf = __builtin_bit_cast(f32x4, v4[i]);
          f32x4 po, mo, vo;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float pj = pf[j], mj = mf[j], vj = vf[j];
            fx_adam_update(pj, mj, vj, g4[j], ak);
            po[j] = pj;
            mo[j] = mj;
            vo[j] = vj;
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, po), rP, off[i], 0, NT);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mo), rM, off[i], 0, NT);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vo), rV, off[i], 0, NT);
        }
      }
    } else {
      // generic path (weight width not a multiple of 4): dword accesses in the MFMA C/D layout
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int mbase = wr * 64 + blk * 32 + 4 * (lane >> 5);      // relative to the rebased descriptors
        float pv[16], mv[16], vv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          const unsigned off = (unsigned)(((long)m * g.ldc + n) * 4) | oob;
          pv[r] = bld32f<NT>(rP, off);
          mv[r] = bld32f<NT>(rM, off);
          vv[r] = bld32f<NT>(rV, off);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          const unsigned off = (unsigned)(((long)m * g.ldc + n) * 4) | oob;
          float pj = pv[r], mj = mv[r], vj = vv[r];
          fx_adam_update(pj, mj, vj, blk == 0 ? acc0[r] : acc1[r], ak);
          bst32f<NT>(pj, rP, off);
          bst32f<NT>(mj, rM, off);
          bst32f<NT>(vj, rV, off);
        }
      }
    }
  }
}

// ---- wide forward with several M tiles per workgroup ------------------------------------------------------
// M > 128 (the triplet network stacks anchor / positive / negative: M = 3B) used to launch one workgroup per 128-row
// M tile, each streaming the same W tile from HBM (3x the traffic: 650 us instead of ~250 per modality at cfg4;
// co-scheduling the tiles on one XCD did not produce L2 hits).  Here one workgroup owns MT M-tiles of one N tile:
// W is loaded and split once per K-step and multiplied into MT accumulator sets.  LDS holds two stages of
// [MT x (X hi, X lo)] + [W hi, W lo] (2 x 64 KB at MT = 3); X is prefetched one K-step ahead in registers and W, the
// HBM stream, two (its latency is ~2 us; X comes from L2).  Loads and stashes are unconditional; K-steps past the
// slice are requested out of range (zeros).
template <int MT>
__global__ __launch_bounds__(512) void fx_fwd_bf16x3_mt_kernel(XGemmArgs g) {
  constexpr int TN = 128, ARR = TM * TK;                        // one 128 x 32 bf16 array
  constexpr int STAGE = (2 * MT + 2) * ARR;                     // MT x (X hi, X lo) + (W hi, W lo)
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * STAGE];   // two stages: 96 KB (MT = 2) / 128 KB (MT = 3)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid & 1, wc = wid >> 1;
  const int groups_m = (g.M + MT * TM - 1) / (MT * TM);
  int lin = blockIdx.x;
  const int z = lin % g.splitk;
  lin /= g.splitk;
  const int gm = lin % groups_m, tn = lin / groups_m;
  const int m0 = gm * MT * TM, n0 = tn * TN;
  const int k_begin = z * g.kchunk;
  const int k_end = min(g.K, k_begin + g.kchunk);
  const int nk = (k_end > k_begin) ? (k_end - k_begin) / TK : 0;
  const __amdgpu_buffer_rsrc_t rAh = fx_rsrc(g.Ahi, (long)g.K * g.a_rp * 2), rAl = fx_rsrc(g.Alo, (long)g.K * g.a_rp * 2);
  const __amdgpu_buffer_rsrc_t rB = fx_rsrc(g.Bf + (long)n0 * g.ldb, (long)max(min(TN, g.N - n0), 0) * g.ldb * 4);   // rebased: W may exceed 4 GiB
  const int a_row = tid >> 2, a_c = tid & 3;
  const unsigned a_off = (unsigned)(((long)(m0 + a_row) * TK + 8 * a_c) * 2);
  const unsigned a_tile = TM * TK * 2u;                         // bytes between M tiles of one K-step (K-blocked X)
  const unsigned a_step = (unsigned)g.a_rp * (TK * 2u), a_kb = (unsigned)(k_begin / TK) * a_step;
  const int bn = tid >> 3, k4 = tid & 7;
  const unsigned b_off0 = (unsigned)(((long)bn * g.ldb + 4 * k4) * 4);
  const unsigned b_off1 = (unsigned)(((long)(bn + 64) * g.ldb + 4 * k4) * 4);
  const unsigned b_kb = (unsigned)k_begin * 4u, b_step = TK * 4u;
  const int a_lds = swz(a_row, a_c);
  const int b_lds0 = swz(bn, k4 >> 1) + ((k4 & 1) << 2), b_lds1 = swz(bn + 64, k4 >> 1) + ((k4 & 1) << 2);
  u32x4 pah[MT], pal[MT], s0_b0, s0_b1, s1_b0, s1_b1;
#define MT_LOAD_A(kt)                                                            \
  {                                                                              \
    const unsigned past = ((kt) < nk) ? 0u : 0xFFFFFFF0u;                        \
    const unsigned ka = a_kb + (unsigned)(kt) * a_step;                          \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                             \
      pah[m] = bld128(rAh, (a_off + m * a_tile + ka) | past);                    \
      pal[m] = bld128(rAl, (a_off + m * a_tile + ka) | past);                    \
    }                                                                            \
  }
#define MT_LOAD_B(P, kt)                                                         \
  {                                                                              \
    const unsigned past = ((kt) < nk) ? 0u : 0xFFFFFFF0u;                        \
    const unsigned kb = b_kb + (unsigned)(kt) * b_step;                          \
    P##_b0 = bld128(rB, (b_off0 + kb) | past);                                   \
    P##_b1 = bld128(rB, (b_off1 + kb) | past);                                   \
  }
#define MT_STASH(P, buf)                                                         \
  {                                                                              \
    __bf16* base = smem + (buf) * STAGE;                                         \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                             \
      *reinterpret_cast<u32x4*>(base + (2 * m) * ARR + a_lds) = pah[m];          \
      *reinterpret_cast<u32x4*>(base + (2 * m + 1) * ARR + a_lds) = pal[m];      \
    }                                                                            \
    split_store4(P##_b0, base + 2 * MT * ARR + b_lds0, base + (2 * MT + 1) * ARR + b_lds0); \
    split_store4(P##_b1, base + 2 * MT * ARR + b_lds1, base + (2 * MT + 1) * ARR + b_lds1); \
  }
  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[m][0][i] = 0.f; acc[m][1][i] = 0.f; }
  const int arow = wr * 64 + (lane & 31), brow = wc * 32 + (lane & 31), kh = lane >> 5;
#define MT_COMPUTE(buf)                                                                           \
  {                                                                                               \
    const __bf16* sb = smem + (buf) * STAGE;                                                      \
    const __bf16* Bh = sb + 2 * MT * ARR;                                                         \
    const __bf16* Bl = Bh + ARR;                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                            \
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Bh + swz(brow, 2 * ks + kh));            \
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Bl + swz(brow, 2 * ks + kh));            \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                            \
        const __bf16* Ah = sb + (2 * m) * ARR;                                                    \
        const __bf16* Al = Ah + ARR;                                                              \
        const bf16x8 ah0 = *reinterpret_cast<const bf16x8*>(Ah + swz(arow, 2 * ks + kh));         \
        const bf16x8 al0 = *reinterpret_cast<const bf16x8*>(Al + swz(arow, 2 * ks + kh));         \
        const bf16x8 ah1 = *reinterpret_cast<const bf16x8*>(Ah + swz(arow + 32, 2 * ks + kh));    \
        const bf16x8 al1 = *reinterpret_cast<const bf16x8*>(Al + swz(arow + 32, 2 * ks + kh));    \
        acc[m][0] = MFMA_BF16(al0, bh, acc[m][0]); acc[m][1] = MFMA_BF16(al1, bh, acc[m][1]);     \
        acc[m][0] = MFMA_BF16(ah0, bl, acc[m][0]); acc[m][1] = MFMA_BF16(ah1, bl, acc[m][1]);     \
        acc[m][0] = MFMA_BF16(ah0, bh, acc[m][0]); acc[m][1] = MFMA_BF16(ah1, bh, acc[m][1]);     \
      }                                                                                           \
    }                                                                                             \
  }
  // Two LDS stages, one barrier per K-step: while stage j is multiplied, the next K-step's operands (already in
  // registers) are stashed into the other stage.  (The first version had a single stage and two barriers per step --
  // stash and MFMA never overlapped: 580 us per modality at cfg4.)
  MT_LOAD_B(s0, 0);
  MT_LOAD_B(s1, 1);
  MT_LOAD_A(0);
  MT_STASH(s0, 0);
  __syncthreads();
  MT_LOAD_A(1);
  MT_LOAD_B(s0, 2);
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 0; kt < nk; kt += 2) {
    MT_COMPUTE(0);                      // K-step kt
    MT_STASH(s1, 1);                    // K-step kt + 1 (zeros past the slice)
    __syncthreads();
    MT_LOAD_A(kt + 2);
    MT_LOAD_B(s1, kt + 3);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 >= nk) break;
    MT_COMPUTE(1);                      // K-step kt + 1
    MT_STASH(s0, 0);                    // K-step kt + 2
    __syncthreads();
    MT_LOAD_A(kt + 3);
    MT_LOAD_B(s0, kt + 4);
    __builtin_amdgcn_sched_barrier(0);
  }
  const int n = n0 + wc * 32 + (lane & 31);
  const unsigned oob = (n < g.N) ? 0u : 0xFFFFFFF0u;
  const __amdgpu_buffer_rsrc_t rC = fx_rsrc(g.C + (long)z * g.slab_stride, (long)g.M * g.ldc * 4);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int mbase = m0 + m * TM + wr * 64 + blk * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mm = mbase + (r & 3) + 8 * (r >> 2);
        const unsigned off = (unsigned)(((long)mm * g.ldc + n) * 4) | oob;       // rows >= M fall outside the slab
        bst32f(acc[m][blk][r], rC, off);
      }
    }
}

// ---- multi-M-tile forward, second design: X by LDS-DMA, loads dealt out between the MFMAs -------------------
// The same contraction as fx_fwd_bf16x3_mt_kernel.  What that kernel's 650 us at the cfg4 shape (384 x 30000 -> 7500) are made
// of was measured by switching parts of THIS kernel off (1 x MI355X, fwd + slab reduce, us):
//     fragment reads + barriers + loop only 254 | + MFMA 320 | + MFMA + X 341 | + MFMA + W 427 | + X + W (no MFMA) 366 | all 570
//     MFMA alone (no LDS reads, loads, barriers) 282; scripts/mfmaprobe.hip: the same MFMA mix sustains 2.0-2.1 PFLOP/s = 260 us
// MFMA, LDS fragment reads and the X stream (2.7 GB from the L2: X is re-read by all 59 column tiles) overlap well; what does
// not is the 0.9 GB HBM stream of W on top of them (+ 230 us), whatever its form -- straight into MFMA operand layout or
// coalesced, two or four K-steps ahead, temporal or non-temporal all measure 565-590 us.  The kernel is ~12 % faster than the
// first design (cfg4 step 6.31 -> 6.0 ms) because
//  * X goes global -> LDS by DMA, two K-steps ahead in three stages: no register staging, no ds_write;
//  * W is loaded coalesced (8 lanes x 16 B = the 128 bytes a row contributes to a K-step) four K-steps ahead into two
//    registers per thread, split into (hi, lo) at the top of its K-step and written to a single 16 KB tile between two barriers;
//  * the K-step's load instructions are dealt out between its MFMAs instead of issued in one burst after the barrier (an
//    in-order wave cannot issue MFMAs behind a load the memory pipeline has not accepted yet), the fragment reads run one
//    product ahead, and an accumulator comes round every 2 MT MFMAs;
//  * one raw s_barrier pair per K-step with an explicit vmcnt (a __syncthreads would drain the K-steps in flight).
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
__device__ __forceinline__ void split8(const u32x4 a, const u32x4 b, bf16x8& h, bf16x8& l) {
  const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);   // whole-vector casts (see split_store4)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float va = fa[j], vb = fb[j];
    h[j] = (__bf16)va;
    l[j] = (__bf16)(va - (float)h[j]);
    h[4 + j] = (__bf16)vb;
    l[4 + j] = (__bf16)(vb - (float)h[4 + j]);
  }
}

template <int MT, int NT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void fx_fwd_bf16x3_dma_kernel(XGemmArgs g) {
  constexpr int ROWS = MT * TM;                 // batch rows per workgroup
  constexpr int HALF = ROWS * TK;               // bf16 elements of the hi (or lo) array of one stage
  constexpr int STAGE = 2 * HALF;
  __shared__ __attribute__((aligned(16))) __bf16 smem[3 * STAGE + 8192];   // 3 x 48 KB at MT = 3 + 16 KB of W fragments = all 160 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w & 1, wc = w >> 1;
  const int groups_m = (g.M + ROWS - 1) / ROWS;
  int lin = blockIdx.x;
  const int z = lin % g.splitk;
  lin /= g.splitk;
  const int gm = lin % groups_m, tn = lin / groups_m;
  const int m0 = gm * ROWS, n0 = tn * 128;
  const int k_begin = z * g.kchunk;
  const int k_end = min(g.K, k_begin + g.kchunk);
  const int nk = (k_end > k_begin) ? (k_end - k_begin) / TK : 0;
  const __amdgpu_buffer_rsrc_t rAh = fx_rsrc(g.Ahi, (long)g.K * g.a_rp * 2), rAl = fx_rsrc(g.Alo, (long)g.K * g.a_rp * 2);
  const __amdgpu_buffer_rsrc_t rB = fx_rsrc_uniform(g.Bf + (long)n0 * g.ldb, (long)max(min(128, g.N - n0), 0) * g.ldb * 4);   // rebased: W may exceed 4 GiB
  // LDS-DMA: one wave instruction fills 16 rows x 64 B in lane order (row = lane / 4, slot = lane % 4); slot s of row r
  // must hold chunk s ^ ((r >> 2) & 3).  Wave w fills rows 16 w .. 16 w + 15 of every M tile, hi and lo.
  const int dr = lane >> 2, dsl = lane & 3;
  const unsigned a_src = (unsigned)(((long)(m0 + 16 * w + dr) * TK + 8 * (dsl ^ ((dr >> 2) & 3))) * 2);
  const unsigned a_tile = TM * TK * 2u;
  const unsigned a_step = (unsigned)g.a_rp * (TK * 2u), a_kb = (unsigned)(k_begin / TK) * a_step;
  const int l31 = lane & 31, kh = lane >> 5;
  // W: coalesced -- 8 lanes x 16 B cover the 128 bytes one row contributes to a K-step; thread t loads rows t / 8 and t / 8 + 64
  const int bn = tid >> 3, k4 = tid & 7;
  const unsigned b_off = (unsigned)(((long)bn * g.ldb + 4 * k4) * 4), b_row64 = (unsigned)((long)64 * g.ldb * 4);
  const int b_lds0 = swz(bn, k4 >> 1) + ((k4 & 1) << 2), b_lds1 = swz(bn + 64, k4 >> 1) + ((k4 & 1) << 2);
  const unsigned b_kb = (unsigned)k_begin * 4u, b_step = TK * 4u;
#define WR_ISSUE_X(kt_, stage_)                                                                                    \
  {                                                                                                                \
    const unsigned past = ((kt_) < nk) ? 0u : 0xFFFFFFF0u;                                                         \
    const unsigned ka = a_kb + (unsigned)(kt_) * a_step;                                                           \
    __bf16* sb = smem + (stage_) * STAGE + 16 * w * TK;                                                            \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rAh, LDS_PTR(sb + m * TM * TK), 16, (a_src + m * a_tile + ka) | past, 0, 0, 0);        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rAl, LDS_PTR(sb + HALF + m * TM * TK), 16, (a_src + m * a_tile + ka) | past, 0, 0, 0); \
    }                                                                                                              \
  }
#define WR_ISSUE_W(P, kt_)                                                                                         \
  {                                                                                                                \
    const unsigned past = ((kt_) < nk) ? 0u : 0xFFFFFFF0u;                                                         \
    const unsigned kb = b_off + b_kb + (unsigned)(kt_) * b_step;                                                   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) P[j] = bld128<NT>(rB, (kb + j * b_row64) | past);                \
  }
  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[m][0][i] = 0.f; acc[m][1][i] = 0.f; }
  // fragment offsets (elements) of sub-step ks within a stage: row = wr * 64 + l31 (+ 32, + 128 m), chunk 2 ks + kh
  const int arow = wr * 64 + l31;
  const int f_off0 = swz(arow, kh), f_off1 = swz(arow, 2 + kh);
  const int fb_off0 = swz(wc * 32 + l31, kh), fb_off1 = swz(wc * 32 + l31, 2 + kh);
// fragments of sub-step ks: F[2 m + blk] = rows wr * 64 + 32 blk + l31 of M tile m, the hi (half = 0) or lo (half = 1) array
#define WR_RD(F, xs, ks, half)                                                                          \
  {                                                                                                     \
    const __bf16* fp = (xs) + ((ks) ? f_off1 : f_off0) + (half) * HALF;                                 \
    _Pragma("unroll") for (int i = 0; i < 2 * MT; ++i)                                                  \
      F[i] = *reinterpret_cast<const bf16x8*>(fp + (i >> 1) * TM * TK + (i & 1) * 32 * TK);             \
  }
// one of the three products against all 2 MT accumulators
#define WR_P(F, B)                                                                                      \
  {                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 2 * MT; ++i) acc[i >> 1][i & 1] = MFMA_BF16(F[i], B, acc[i >> 1][i & 1]); \
  }
// schedule of one K-step: 0x008 = MFMA, 0x100 = LDS read, 0x020 = VMEM read.  Six products of 2 MT MFMAs (lo x hi, hi x lo,
// hi x hi of sub-step 0, then of sub-step 1); the next product's fragment reads go one per MFMA, the K-step's 2 MT + 4 load
// instructions one every third MFMA.
#define WR_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
#define WR_STEP(P, kt_)                                                                                 \
  {                                                                                                     \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MT + 4) : "memory"); /* X of K-step kt_ (this wave's pieces) has landed: all but the W loads issued with it and the whole next group */ \
    asm volatile("s_barrier" ::: "memory");            /* everyone's has; stage st - 1 and the W fragments are free again */ \
    split_store4(P[0], wfr + b_lds0, wfr + 4096 + b_lds0);     /* W of this K-step: (hi, lo) [128][32], swizzled like X */ \
    split_store4(P[1], wfr + b_lds1, wfr + 4096 + b_lds1);                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
    asm volatile("s_barrier" ::: "memory");            /* the W tile is complete */                     \
    bf16x8 bh[2], bl[2];                                                                                \
    bh[0] = *reinterpret_cast<const bf16x8*>(wfr + fb_off0);                                            \
    bl[0] = *reinterpret_cast<const bf16x8*>(wfr + 4096 + fb_off0);                                     \
    bh[1] = *reinterpret_cast<const bf16x8*>(wfr + fb_off1);                                            \
    bl[1] = *reinterpret_cast<const bf16x8*>(wfr + 4096 + fb_off1);                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    const __bf16* xs = smem + st * STAGE;                                                               \
    bf16x8 al0[2 * MT], ah0[2 * MT], al1[2 * MT], ah1[2 * MT];                                          \
    WR_RD(al0, xs, 0, 1);                                                                               \
    WR_RD(ah0, xs, 0, 0);                                                                               \
    WR_ISSUE_X((kt_) + 2, st >= 1 ? st - 1 : 2);               /* X two K-steps ahead (L2) ... */        \
    WR_ISSUE_W(P, (kt_) + 4);                                  /* ... W four (HBM), issued AFTER it: vmcnt retires in order */ \
    WR_P(al0, bh[0]);                                                                                   \
    WR_RD(al1, xs, 1, 1);                                                                               \
    WR_P(ah0, bl[0]);                                                                                   \
    WR_P(ah0, bh[0]);                                                                                   \
    WR_RD(ah1, xs, 1, 0);                                                                               \
    WR_P(al1, bh[1]);                                                                                   \
    WR_P(ah1, bl[1]);                                                                                   \
    WR_P(ah1, bh[1]);                                                                                   \
    WR_SGB(0x100, 2 * MT)                                                                  /* lo fragments of sub-step 0 */ \
    _Pragma("unroll") for (int u = 0; u < 2 * MT; ++u) { WR_SGB(0x100, 1) WR_SGB(0x008, 1) }  /* hi fragments of 0 | lo x hi */ \
    _Pragma("unroll") for (int u = 0; u < 2 * MT; ++u) { WR_SGB(0x008, 1) WR_SGB(0x020, 1) }  /* hi x lo | the X pieces of K-step kt + 2 */ \
    _Pragma("unroll") for (int u = 0; u < 2 * MT; ++u) { WR_SGB(0x008, 1) WR_SGB(0x100, 1) }  /* hi x hi | lo fragments of sub-step 1 */ \
    _Pragma("unroll") for (int u = 0; u < 2 * MT; ++u) { WR_SGB(0x008, 1) WR_SGB(0x100, 1) }  /* lo x hi of 1 | hi fragments of 1 */ \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) { WR_SGB(0x008, 1) WR_SGB(0x020, 1) }       /* hi x lo | the W loads of K-step kt + 2 */ \
    WR_SGB(0x008, 4 * MT - 2)                                                              /* hi x hi */ \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    st = st == 2 ? 0 : st + 1;                                                                          \
  }
  // the stages start as zeros: a K-step past the slice must find finite numbers whatever an out-of-range DMA leaves behind
  {
    const u32x4 zz = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < (3 * STAGE * 2) / (512 * 16); ++i) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(smem) + (i * 512 + tid) * 16) = zz;
    __syncthreads();
  }
  __bf16* wfr = smem + 3 * STAGE;      // W of the current K-step: hi [128][32] | lo [128][32]
  u32x4 q0[2], q1[2], q2[2], q3[2];
  int st = 0;
  // issue order = the vmcnt bookkeeping: per K-step the X pieces, then the W loads that are issued with them
  WR_ISSUE_W(q0, 0);
  WR_ISSUE_W(q1, 1);
  __builtin_amdgcn_sched_barrier(0);
  WR_ISSUE_X(0, 0);
  WR_ISSUE_W(q2, 2);
  __builtin_amdgcn_sched_barrier(0);
  WR_ISSUE_X(1, 1);
  WR_ISSUE_W(q3, 3);
  __builtin_amdgcn_sched_barrier(0);
  // K-steps in fours without an exit in between (one basic block, so that the compiler's own vmcnt bookkeeping for the W
  // registers stays exact): K-steps past the slice multiply zeros (their loads are out of range)
  for (int kt = 0; kt < nk; kt += 4) {
    WR_STEP(q0, kt);
    WR_STEP(q1, kt + 1);
    WR_STEP(q2, kt + 2);
    WR_STEP(q3, kt + 3);
  }
  const int n = n0 + wc * 32 + l31;
  const unsigned oob = (n < g.N) ? 0u : 0xFFFFFFF0u;
  const __amdgpu_buffer_rsrc_t rC = fx_rsrc(g.C + (long)z * g.slab_stride, (long)g.M * g.ldc * 4);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int mbase = m0 + m * TM + wr * 64 + blk * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mm = mbase + (r & 3) + 8 * (r >> 2);
        const unsigned off = (unsigned)(((long)mm * g.ldc + n) * 4) | oob;       // rows >= M fall outside the slab
        bst32f(acc[m][blk][r], rC, off);
      }
    }
}

// ---- stacked-rows forward, fourth design (round 5): X never enters the LDS ----------------------------------------------------------
// profiles/r05_fwd_mt_power.txt: the DMA kernel above is LDS-bound on zeros (8 ds_read_b128 per 6 MFMA = 50 LDS cycles per 32-cycle MFMA,
// measured 56) and POWER-bound on real operands (1400 W, 1.73 GHz).  Both say the same thing: move fewer bytes per product.  Here wave w
// of eight owns rows 16 R w .. 16 R (w + 1) of the tile and ALL 128 columns, on v_mfma_f32_16x16x32_bf16 (one MFMA contracts a whole
// 32-k block; same rate as 32x32x16):
//  * its X fragments are shared with no other wave, so they go L2 -> registers directly, one block ahead: with 16-row MFMAs a fragment load is
//    16 rows x 64 B of the K-blocked split = 1 KB contiguous.  No X in the LDS at all (it was 96 of the 128 KB written per 64 k);
//  * W (the only operand the eight waves share) is loaded coalesced four blocks ahead, split once two blocks ahead, written as (hi, lo)
//    fragments to one of three 16.5 KB buffers: one barrier per 32 k, and nothing right behind it waits for the LDS (the fragments of the
//    next block's first half are read during this block's second half);
//  * per block a wave reads 16 fragments of W for 24 R MFMAs (R = 3: 4.5 MFMA per ds_read_b128, at half the flops each; the DMA kernel: 0.75).
// (A first version with FOUR waves, one per SIMD on 32x32x16 MFMAs and 96 x 128 per wave, ran 428 us on zeros / 509 on random operands at
// the triplet shape against 396 / 565 shipped: with one wave per SIMD every load instruction's issue stalls the MFMA stream behind it --
// timing ablations: no X loads 293 us, no W loads 335, no barrier 354, MFMAs alone 271 on zeros and 335 on random operands.)
#define MFMA16_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
template <int R, int WV, int NT, int NP = 3>
__global__ __launch_bounds__(64 * WV) __attribute__((amdgpu_waves_per_eu(2, 2))) void fx_fwd_bf16x3_reg_kernel(XGemmArgs g) {
  constexpr int ROWS = 16 * R * WV;             // batch rows per workgroup: WV waves x 16 R (WV = 4: two workgroups per CU)
  constexpr int LW = 16 / WV;                   // 16-byte pieces of W per thread and block
  constexpr int PLANE = 2048 + 64;              // one 8-k chunk of all 128 columns (16 B each) + a 16-bank shift per chunk: the split's 8-byte
                                                // stores of 4 columns x 4 chunks (32 lanes) land on 64 distinct banks
  constexpr int HALFB = 4 * PLANE, BUF = 2 * HALFB;
  __shared__ __attribute__((aligned(16))) char smem[3 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int groups_m = (g.M + ROWS - 1) / ROWS;
  int lin = blockIdx.x;
  const int z = lin % g.splitk;
  lin /= g.splitk;
  const int gm = lin % groups_m, tn = lin / groups_m;
  const int m0 = gm * ROWS + w * 16 * R, n0 = tn * 128;
  const int k_begin = z * g.kchunk;
  const int k_end = min(g.K, k_begin + g.kchunk);
  const int nb = (k_end > k_begin) ? (k_end - k_begin) / TK : 0;
  const __amdgpu_buffer_rsrc_t rAh = fx_rsrc(g.Ahi, (long)g.K * g.a_rp * 2), rAl = fx_rsrc(g.Alo, (long)g.K * g.a_rp * 2);
  const __amdgpu_buffer_rsrc_t rB = fx_rsrc_uniform(g.Bf + (long)n0 * g.ldb, (long)max(min(128, g.N - n0), 0) * g.ldb * 4);   // rebased: W may exceed 4 GiB
  // X fragment of row block r of block j: row m0 + 16 r + l15, 16-byte chunk kq of the row's 64 bytes
  const unsigned a_src = (unsigned)(((long)(m0 + l15) * TK + 8 * kq) * 2);
  const unsigned a_step = (unsigned)g.a_rp * (TK * 2u), a_kb = (unsigned)(k_begin / TK) * a_step;
  // W: 8 lanes x 16 B = the 128 bytes a row contributes to a block; thread t loads rows t / 8 + 8 WV i
  const int bn = tid >> 3, k4 = tid & 7;
  const unsigned b_off = (unsigned)(((long)bn * g.ldb + 4 * k4) * 4), b_rows = (unsigned)((long)(8 * WV) * g.ldb * 4);
  const unsigned b_kb = (unsigned)k_begin * 4u, b_step = TK * 4u;
  const int b_lds = (k4 >> 1) * PLANE + bn * 16 + (k4 & 1) * 8;      // + 128 WV per 8 WV rows, + HALFB for lo
  const int fb = kq * PLANE + l15 * 16;                              // + 256 per 16 columns
  f32x4 acc[R][8];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[r][c][i] = 0.f;
#define RG_LOAD_A(AH, AL, j_)                                                                                      \
  {                                                                                                                \
    const unsigned past = ((j_) < nb) ? 0u : 0xFFFFFFF0u;                                                          \
    const unsigned ka = a_src + a_kb + (unsigned)(j_) * a_step;                                                    \
    _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                                \
      AH[r] = __builtin_bit_cast(bf16x8, bld128<0>(rAh, (ka + r * (16 * TK * 2)) | past));                         \
      if (NP == 3) AL[r] = __builtin_bit_cast(bf16x8, bld128<0>(rAl, (ka + r * (16 * TK * 2)) | past));            \
    }                                                                                                              \
  }
#define RG_LOAD_W(Q, j_)                                                                                           \
  {                                                                                                                \
    const unsigned past = ((j_) < nb) ? 0u : 0xFFFFFFF0u;                                                          \
    const unsigned kb = b_off + b_kb + (unsigned)(j_) * b_step;                                                    \
    _Pragma("unroll") for (int i = 0; i < LW; ++i) Q[i] = bld128<NT>(rB, (kb + i * b_rows) | past);                \
  }
#define RG_STAGE_W(Q, buf_)                                                                                        \
  {                                                                                                                \
    char* d = smem + (buf_) * BUF + b_lds;                                                                         \
    _Pragma("unroll") for (int i = 0; i < LW; ++i)                                                                 \
      split_store4<NP == 3>(Q[i], reinterpret_cast<__bf16*>(d + i * 128 * WV), reinterpret_cast<__bf16*>(d + HALFB + i * 128 * WV)); \
  }
#define RG_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0);
// timing ablations (scripts/build_variant.py -DRG_NO_*; results wrong): which part of a block the MFMAs wait for
#ifdef RG_NO_A
#define RG_A_ON 0
#else
#define RG_A_ON 1
#endif
#ifdef RG_NO_W
#define RG_W_ON 0
#else
#define RG_W_ON 1
#endif
#ifdef RG_NO_STAGE
#define RG_STAGE_ON 0
#else
#define RG_STAGE_ON 1
#endif
#ifdef RG_NO_BAR
#define RG_SYNC_L()
#else
#define RG_SYNC_L() __syncthreads()
#endif
// Block j (position JB of three in the unrolled loop) in eight steps of 16 columns.  Step c: the W fragments of step c + 2 are read into a
// ring of three (those of the next block's first two steps from buffer BN, complete since the barrier that ended block j - 1), ONE of the
// block's loads is issued -- the 2 R fragments of X for block j + 2, the two pieces of W for block j + 4 -- and the step's 3 R MFMAs run
// (small terms first).  W of block j + 2 (requested during block j - 2) is split into buffer BS in step 4.  X is requested TWO blocks ahead
// because register loads retire in order: a fragment of X (an L2 hit) cannot be used before every older piece of W (HBM) has arrived.
// 0x008 MFMA, 0x020 VMEM read, 0x100 LDS read, 0x200 LDS write, 0x002 VALU
#define RG_BLOCK(JB, AHC, ALC, AHN, ALN, QS, QL, j_, BC, BN, BS)                                                   \
  {                                                                                                                \
    const unsigned pastA = ((j_) + 2 < nb) ? 0u : 0xFFFFFFF0u, pastW = ((j_) + 4 < nb) ? 0u : 0xFFFFFFF0u;          \
    const unsigned ka = a_src + a_kb + (unsigned)((j_) + 2) * a_step;                                              \
    const unsigned kb = b_off + b_kb + (unsigned)((j_) + 4) * b_step;                                              \
    _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                                                \
      {                                                                                                            \
        const int c2 = (c + 2) & 7, s2 = (8 * (JB) + c + 2) % 3;                                                   \
        const char* sp = smem + ((c + 2 < 8) ? (BC) : (BN)) * BUF + fb + c2 * 256;                                 \
        fh[s2] = *reinterpret_cast<const bf16x8*>(sp);                                                             \
        if (NP == 3) fl[s2] = *reinterpret_cast<const bf16x8*>(sp + HALFB);                                        \
      }                                                                                                            \
      if (c < 2 * R) {                                                                                             \
        if (RG_A_ON) {                                                                                             \
          if (c & 1) { if (NP == 3) ALN[c >> 1] = __builtin_bit_cast(bf16x8, bld128<0>(rAl, (ka + (c >> 1) * (16 * TK * 2)) | pastA)); } \
          else AHN[c >> 1] = __builtin_bit_cast(bf16x8, bld128<0>(rAh, (ka + (c >> 1) * (16 * TK * 2)) | pastA));  \
        }                                                                                                          \
      } else if (RG_W_ON && c - 2 * R < LW) {                                                                      \
        QL[c - 2 * R] = bld128<NT>(rB, (kb + (c - 2 * R) * b_rows) | pastW);                                       \
      }                                                                                                            \
      if (c == 4 && RG_STAGE_ON) RG_STAGE_W(QS, BS);                                                               \
      const int s0 = (8 * (JB) + c) % 3;                                                                           \
      if (NP == 3) {                                                                                               \
        _Pragma("unroll") for (int r = 0; r < R; ++r) acc[r][c] = MFMA16_BF16(ALC[r], fh[s0], acc[r][c]);          \
        _Pragma("unroll") for (int r = 0; r < R; ++r) acc[r][c] = MFMA16_BF16(AHC[r], fl[s0], acc[r][c]);          \
      }                                                                                                            \
      _Pragma("unroll") for (int r = 0; r < R; ++r) acc[r][c] = MFMA16_BF16(AHC[r], fh[s0], acc[r][c]);            \
      if (NP == 3) {                                                                                               \
        RG_SGB(0x008, 1) RG_SGB(0x100, 1) RG_SGB(0x008, 1) RG_SGB(0x100, 1) RG_SGB(0x008, 1) RG_SGB(0x020, 1)      \
        if (c == 4) { RG_SGB(0x008, 1) RG_SGB(0x002, 6 * LW) RG_SGB(0x008, 1) RG_SGB(0x002, 6 * LW) RG_SGB(0x008, 1) RG_SGB(0x002, 6 * LW) \
                      RG_SGB(0x008, 1) RG_SGB(0x200, LW) RG_SGB(0x008, 1) RG_SGB(0x200, LW) }                      \
        RG_SGB(0x008, 3 * R)                                                                                       \
      }                                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
    }                                                                                                              \
    RG_SYNC_L();                                                                                                   \
  }
  static_assert(2 * R + LW <= 8, "step c issues load c of the block's 2 R + LW");
  bf16x8 ah0[R], al0[R], ah1[R], al1[R], ah2[R], al2[R], fh[3], fl[3];
  u32x4 q0[LW], q1[LW], q2[LW];
  RG_LOAD_W(q0, 0);
  RG_LOAD_W(q1, 1);
  RG_LOAD_A(ah0, al0, 0);
  RG_LOAD_A(ah1, al1, 1);
#if defined(RG_NO_A)
  RG_LOAD_A(ah2, al2, 2);
#endif
  RG_LOAD_W(q2, 2);
  RG_STAGE_W(q0, 0);
  RG_LOAD_W(q0, 3);
  RG_STAGE_W(q1, 1);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    fh[c] = *reinterpret_cast<const bf16x8*>(smem + fb + c * 256);
    if (NP == 3) fl[c] = *reinterpret_cast<const bf16x8*>(smem + fb + c * 256 + HALFB);
  }
  for (int j = 0; j < nb; j += 3) {            // a block past the slice multiplies zeros (its loads are out of range)
    RG_BLOCK(0, ah0, al0, ah2, al2, q2, q1, j, 0, 1, 2);
    RG_BLOCK(1, ah1, al1, ah0, al0, q0, q2, j + 1, 1, 2, 0);
    RG_BLOCK(2, ah2, al2, ah1, al1, q1, q0, j + 2, 2, 0, 1);
  }
  // D of a 16x16 MFMA: column lane & 15, rows 4 (lane >> 4) + i
  const unsigned rowsz = (unsigned)g.ldc * 4u;
  const __amdgpu_buffer_rsrc_t rC = fx_rsrc(g.C + (long)z * g.slab_stride, (long)g.M * g.ldc * 4);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int n = n0 + 16 * c + l15;
    const unsigned oob = (n < g.N) ? 0u : 0xFFFFFFF0u;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mm = m0 + 16 * r + 4 * kq + i;
        bst32f(acc[r][c][i], rC, ((unsigned)mm * rowsz + (unsigned)n * 4u) | oob);       // rows >= M fall outside the slab
      }
  }
}

// ---- operand splitting ----------------------------------------------------------------------------------
// hi/lo K-BLOCKED [Cp/32][Rp][32] from x [R, C]: element (r, c) at ((c/32)*Rp + r)*32 + c%32; columns C..Cp-1 are
// written as zeros (Cp = C rounded up to 32); rows R..Rp-1 are never written (the caller allocates them as zeros)
__global__ __launch_bounds__(256) void fx_split_bf16_kernel(__bf16* __restrict__ hi, __bf16* __restrict__ lo,
                                                            const float* __restrict__ x, int R, int C, int Cp, long ldx,
                                                            long Rp) {
  const int r = blockIdx.y;
  const __amdgpu_buffer_rsrc_t rx = fx_rsrc(x + (long)r * ldx, (long)C * 4);
  for (int c4 = blockIdx.x * blockDim.x + threadIdx.x; c4 < Cp / 4; c4 += gridDim.x * blockDim.x) {
    const u32x4 raw = bld128(rx, (unsigned)c4 * 16u);      // out-of-range columns read as 0
    const long o = ((long)(c4 >> 3) * Rp + r) * 32 + 4 * (c4 & 7);
    split_store4(raw, hi + o, lo + o);
  }
}

// transposed: hiT/loT [C, ldo] with hiT[c][r] = hi(x[r][c]); rows r in R..Rp-1 zero (Rp = R rounded up to 32)
__global__ __launch_bounds__(256) void fx_split_bf16_t_kernel(__bf16* __restrict__ hiT, __bf16* __restrict__ loT,
                                                              const float* __restrict__ x, int R, int C, int Rp, long ldx,
                                                              long ldo) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    const bool ok = r < R && c < C;
    const float v = x[(long)min(r, R - 1) * ldx + min(c, C - 1)];
    tile[ty + 8 * i][tx] = ok ? v : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C && r < Rp) {
      const float v = tile[tx][ty + 8 * i];
      const __bf16 h = (__bf16)v;
      hiT[(long)c * ldo + r] = h;
      loT[(long)c * ldo + r] = (__bf16)(v - (float)h);
    }
  }
}

// Reconstruction epilogue of a decoder's FC_output forward (reference supervised_vae.py:301-313, modules.py:99-103): the ordered
// sum of the forward's split-K slabs + bias = logits (never stored), x_hat = sigmoid(logits), per-block sums of (x_hat - x)^2,
// dlogits = w (x_hat - x) x_hat (1 - x_hat) in fp32 (the weight-gradient operands are made from it) and dlogits' K-blocked bf16
// split (fx_split_bf16's layout: the A operand of the data-gradient product through FC_output).  One pass instead of
// fx_reduce_slabs4 -> fx_recon_sigmoid -> fx_split_bf16.  Same summation order and formulas as those kernels.
__global__ __launch_bounds__(256) void fx_recon_sigmoid_slabs_kernel(float* __restrict__ partial, float* __restrict__ dlogits,
                                                                     __bf16* __restrict__ hi, __bf16* __restrict__ lo,
                                                                     const float* __restrict__ slabs, int nslabs, long slab_stride,
                                                                     const float* __restrict__ bias, const float* __restrict__ x,
                                                                     int F, int Cp, long Rp, const float* logvar, float extra_scale, float n) {
  __shared__ float sm[16];
  const int r = blockIdx.y;
  const float w = (logvar ? expf(-logvar[0]) : 1.0f) * extra_scale * 2.0f / n;      // (fx_recon_sigmoid's expression, term for term)
  float acc = 0.f;
  for (int c4 = blockIdx.x * blockDim.x + threadIdx.x; c4 < Cp / 4; c4 += gridDim.x * blockDim.x) {
    f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
    if (4 * c4 < F) {
      const long e = (long)r * F + 4 * c4;
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      int z = 0;
      for (; z + 4 <= nslabs; z += 4) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(slabs + (long)(z + j) * slab_stride + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += v[j];
      }
      for (; z < nslabs; ++z) s += *reinterpret_cast<const f32x4*>(slabs + (long)z * slab_stride + e);
      if (bias) s += *reinterpret_cast<const f32x4*>(bias + 4 * c4);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + e);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = 1.0f / (1.0f + expf(-s[j]));
        const float d = xh - xv[j];
        acc += d * d;
        d4[j] = w * d * xh * (1.0f - xh);
      }
      if (dlogits) *reinterpret_cast<f32x4*>(dlogits + e) = d4;
    }
    if (hi) {          // columns F..Cp-1 are written as zeros, like fx_split_bf16
      const long o = ((long)(c4 >> 3) * Rp + r) * 32 + 4 * (c4 & 7);
      split_store4(__builtin_bit_cast(u32x4, d4), hi + o, lo + o);
    }
  }
  acc = fx_block_sum(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = acc;
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Tuning knobs of the wide kernels.  They are ARGUMENTS of the *_ex entry points (0 = the shipped choice); the library
// itself reads no environment variable and keeps no mutable state (include/fxhip.h).
struct FwdTune { int splitk, wn, mt, nt; };      // forced split-K (0 auto) | 2 or 4 wave columns (0 -> 4) | 1 = no multi-M-tile kernel | non-temporal W loads
static inline int fwd_wn(const FwdTune& t) { return t.wn == 2 ? 2 : (t.wn == 8 ? 8 : 4); }

// split-K so that the grid fills the chip about once: 256 CUs x (3 workgroups of 256 threads | 2 of 512)
// M > 128 takes fx_fwd_bf16x3_mt_kernel: MT M-tiles per workgroup, one workgroup (64 KB LDS, ~190 VGPRs) per CU
static int fwd_mt(int M) { return M > 2 * TM ? 3 : (M > TM ? 2 : 1); }

static int pick_splitk_x(int M, int N, int K, int wn, const FwdTune& t) {
  if (t.splitk > 0) return t.splitk;
  if (wn == 4 && M > TM && t.mt != 1) {
    const int mt = fwd_mt(M);
    const long tiles = (long)((M + mt * TM - 1) / (mt * TM)) * ((N + 127) / 128);
    if (tiles >= 128 || K <= 8 * TK) return 1;
    int s = (int)(256 / tiles);
    const int maxs = (K + 8 * TK - 1) / (8 * TK);
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
  }
  const int tn = 32 * wn;
  const long tiles = (long)((M + TM - 1) / TM) * ((N + tn - 1) / tn);
  const long slots = wn == 2 ? 768 : (wn == 8 ? 256 : 512);
  if (tiles >= slots / 2 || K <= 8 * TK) return 1;
  int s = (int)(slots / tiles);
  const int maxs = (K + 8 * TK - 1) / (8 * TK);  // keep >= 8 K-steps per slice
  if (s > maxs) s = maxs;
  if (s > 64) s = 64;
  if (wn == 2 && s >= 8) s = (s / 8) * 8;  // K-slice <-> XCD affinity (workgroup id mod 8)
  return s < 1 ? 1 : s;
}

extern "C" {

int fx_split_bf16(void* hi, void* lo, const float* x, int R, int C, long ldx, long rows_padded, hipStream_t stream) {
  FX_REQUIRE(hi && lo && x && R > 0 && C > 0, "fx_split_bf16: bad args");
  const int Cp = (C + 31) / 32 * 32;
  FX_REQUIRE(rows_padded >= R && rows_padded % 128 == 0 && aligned16(hi) && aligned16(lo),
             "fx_split_bf16: rows_padded %ld must be a multiple of 128 and >= %d", rows_padded, R);
  int bx = (Cp / 4 + 255) / 256;
  if (bx > 32) bx = 32;
  hipLaunchKernelGGL(fx_split_bf16_kernel, dim3(bx, R), dim3(256), 0, stream, (__bf16*)hi, (__bf16*)lo, x, R, C, Cp, ldx,
                     rows_padded);
  return fx_check_launch("fx_split_bf16");
}

static int recon_slabs_bx(int F) {
  const int Cp = (F + 31) / 32 * 32;
  int bx = (Cp / 4 + 255) / 256;
  return bx > 32 ? 32 : bx;
}

// number of per-block partial sums fx_recon_sigmoid_slabs writes (the n_partial of fx_mmd_finalize)
int fx_recon_sigmoid_slabs_blocks(int B, int F) { return B * recon_slabs_bx(F); }

// slabs [nslabs][B * F] (the output of fx_linear_fwd_bf16x3_slabs), bias [F], x [B, F] contiguous; dlogits [B, F] and / or the
// K-blocked split (hi, lo: fx_split_bf16 layout with rows_padded rows) may be null.  w = exp(-logvar) * extra_scale * 2 / (B F).
int fx_recon_sigmoid_slabs(float* partial, float* dlogits, void* hi, void* lo, const float* slabs, int nslabs, long slab_stride,
                           const float* bias, const float* x, int B, int F, long rows_padded, const float* logvar, float extra_scale,
                           hipStream_t stream) {
  FX_REQUIRE(partial && slabs && x && B > 0 && F > 0 && nslabs > 0, "fx_recon_sigmoid_slabs: bad args");
  FX_REQUIRE(F % 4 == 0 && slab_stride % 4 == 0 && aligned16(slabs) && aligned16(x) && (!bias || aligned16(bias)) &&
             (!dlogits || aligned16(dlogits)), "fx_recon_sigmoid_slabs: F (%d) and the slab stride must be multiples of 4, bases 16-byte aligned", F);
  FX_REQUIRE((hi == nullptr) == (lo == nullptr), "fx_recon_sigmoid_slabs: hi and lo come together");
  FX_REQUIRE(!hi || (rows_padded >= B && rows_padded % 128 == 0 && aligned16(hi) && aligned16(lo)),
             "fx_recon_sigmoid_slabs: rows_padded %ld must be a multiple of 128 and >= %d", rows_padded, B);
  const int Cp = (F + 31) / 32 * 32;
  hipLaunchKernelGGL(fx_recon_sigmoid_slabs_kernel, dim3(recon_slabs_bx(F), B), dim3(256), 0, stream, partial, dlogits, (__bf16*)hi,
                     (__bf16*)lo, slabs, nslabs, slab_stride, bias, x, F, Cp, rows_padded, logvar, extra_scale, (float)((long)B * F));
  return fx_check_launch("fx_recon_sigmoid_slabs");
}

int fx_split_bf16_t(void* hiT, void* loT, const float* x, int R, int C, long ldx, long ldo, hipStream_t stream) {
  FX_REQUIRE(hiT && loT && x && R > 0 && C > 0, "fx_split_bf16_t: bad args");
  const int Rp = (R + 31) / 32 * 32;
  FX_REQUIRE(ldo >= Rp && ldo % 8 == 0 && aligned16(hiT) && aligned16(loT), "fx_split_bf16_t: ldo %ld must be >= %d, %%8", ldo, Rp);
  hipLaunchKernelGGL(fx_split_bf16_t_kernel, dim3((C + 31) / 32, Rp / 32), dim3(256), 0, stream, (__bf16*)hiT, (__bf16*)loT,
                     x, R, C, Rp, ldx, ldo);
  return fx_check_launch("fx_split_bf16_t");
}

long fx_linear_fwd_bf16x3_workspace_bytes(int M, int N, int K) {
  const int Kp = (K + TK - 1) / TK * TK;
  const FwdTune t{};
  int s = pick_splitk_x(M, N, Kp, 2, t), s4 = pick_splitk_x(M, N, Kp, 4, t);
  if (s4 > s) s = s4;
  return (long)s * M * N * (long)sizeof(float);  // always goes through slabs (bias is added by the reduce)
}

int fx_linear_fwd_bf16x3_splitk(int M, int N, int K) {
  const FwdTune t{};
  return pick_splitk_x(M, N, (K + TK - 1) / TK * TK, fwd_wn(t), t);
}

static int fwd_bf16x3_impl(float* Y, const void* xhi, const void* xlo, const float* W, const float* bias, int M, int N,
                           int K, long ldx, long ldw, long ldy, void* workspace, long workspace_bytes, bool reduce,
                           hipStream_t stream, bool kn = false, FwdTune tune = FwdTune{});

// Y[M,N] = X[M,K] . W[N,K]^T + bias ; X given as K-BLOCKED split bf16 (fx_split_bf16 / fx_gather_split layout:
// [ceil(K/32)][ldx rows][32], ldx = rows padded to a multiple of 128, padding zero)
int fx_linear_fwd_bf16x3(float* Y, const void* xhi, const void* xlo, const float* W, const float* bias, int M, int N, int K,
                         long ldx, long ldw, long ldy, void* workspace, long workspace_bytes, hipStream_t stream) {
  FX_REQUIRE(Y != nullptr, "fx_linear_fwd_bf16x3: null output");
  return fwd_bf16x3_impl(Y, xhi, xlo, W, bias, M, N, K, ldx, ldw, ldy, workspace, workspace_bytes, true, stream);
}

// The same with explicit tuning (A/B experiments; every variant computes the same contraction): splitk 0 = auto
// (workspace must hold splitk slabs), wave_cols 0|4 = 128x128 tile, 2 = 128x64; no_mt 1 = one workgroup per M tile (2 = the first multi-M-tile kernel) even
// for M > 128; nt 1 = non-temporal W loads.
int fx_linear_fwd_bf16x3_ex(float* Y, const void* xhi, const void* xlo, const float* W, const float* bias, int M, int N, int K,
                            long ldx, long ldw, long ldy, void* workspace, long workspace_bytes, int splitk, int wave_cols,
                            int no_mt, int nt, hipStream_t stream) {
  FX_REQUIRE(Y != nullptr, "fx_linear_fwd_bf16x3_ex: null output");
  return fwd_bf16x3_impl(Y, xhi, xlo, W, bias, M, N, K, ldx, ldw, ldy, workspace, workspace_bytes, true, stream, false,
                         FwdTune{splitk, wave_cols, no_mt, nt});
}

// Same, but the fx_linear_fwd_bf16x3_splitk(M,N,K) partial-sum slabs ([s][M][N]) are left in `slabs` (no bias) for
// a consumer that reduces them in its own pass (fx_bn_act_fwd_slabs).
int fx_linear_fwd_bf16x3_slabs(float* slabs, long slabs_bytes, const void* xhi, const void* xlo, const float* W, int M,
                               int N, int K, long ldx, long ldw, hipStream_t stream) {
  return fwd_bf16x3_impl(nullptr, xhi, xlo, W, nullptr, M, N, K, ldx, ldw, N, slabs, slabs_bytes, false, stream);
}

// ... with the output tile chosen by the caller: wave_cols 0 | 4 = 128 x 128, 8 = 128 x 256 (M <= 128: the activation tile crosses L2 -> LDS
// once per 32 KB of W instead of once per 16 KB; same K slices and summation order, i.e. the same slabs bit for bit at equal split-K)
int fx_linear_fwd_bf16x3_splitk_ex(int M, int N, int K, int wave_cols) {
  FwdTune t{};
  t.wn = wave_cols;
  const int wn = (fwd_wn(t) == 8 && M <= TM) ? 8 : 4;
  return pick_splitk_x(M, N, (K + TK - 1) / TK * TK, wn, t);
}
int fx_linear_fwd_bf16x3_slabs_ex(float* slabs, long slabs_bytes, const void* xhi, const void* xlo, const float* W, int M,
                                  int N, int K, long ldx, long ldw, int wave_cols, hipStream_t stream) {
  FwdTune t{};
  t.wn = wave_cols;
  return fwd_bf16x3_impl(nullptr, xhi, xlo, W, nullptr, M, N, K, ldx, ldw, N, slabs, slabs_bytes, false, stream, false, t);
}

// dX[M,N] = dY[M,K] . W[K,N]   (W = the layer's weight [out=K, in=N] as stored): the data gradient through a WIDE
// layer (autograd's mm in Linear backward, e.g. the supervised_vae / CrossModalPred decoders' FC_output).  dY given
// as a K-blocked split (fx_split_bf16 with rows_padded = lddy_rows).  Workspace: fx_linear_fwd_bf16x3_workspace_bytes.
int fx_linear_bwd_x_bf16x3(float* dX, const void* dyhi, const void* dylo, const float* W, int M, int N, int K, long dy_rows_padded,
                           long ldw, long lddx, void* workspace, long workspace_bytes, hipStream_t stream) {
  FX_REQUIRE(dX != nullptr, "fx_linear_bwd_x_bf16x3: null output");
  FX_REQUIRE(ldw >= N, "fx_linear_bwd_x_bf16x3: ldw %ld < N %d", ldw, N);
  return fwd_bf16x3_impl(dX, dyhi, dylo, W, nullptr, M, N, K, dy_rows_padded, ldw, lddx, workspace, workspace_bytes, true, stream, true);
}

}  // extern "C"

static int fwd_bf16x3_impl(float* Y, const void* xhi, const void* xlo, const float* W, const float* bias, int M, int N,
                           int K, long ldx, long ldw, long ldy, void* workspace, long workspace_bytes, bool reduce,
                           hipStream_t stream, bool kn, FwdTune tune) {
  FX_REQUIRE(xhi && W && M > 0 && N > 0 && K > 0, "fx_linear_fwd_bf16x3: bad args");
  const bool plain = xlo == nullptr;        // no `lo` operand: plain bf16 (one product); W is rounded to bf16 in-kernel
  const int Kp = (K + TK - 1) / TK * TK;
  FX_REQUIRE(ldx >= M && ldx % 128 == 0 && aligned16(xhi) && aligned16(xlo),
             "fx_linear_fwd_bf16x3: X must be a K-blocked split with rows padded to a multiple of 128 (got %ld for M=%d)", ldx, M);
  FX_REQUIRE(!(plain && (tune.mt == 2 || tune.mt == 3)), "fx_linear_fwd_bf16x3: the A/B kernels (no_mt 2 / 3) need the lo operand");
  // wave_cols 8: the 128 x 256 tile (at most 128 rows, row-major W); plain bf16 otherwise always takes the 128 x 128 tile
  const int wn_t = fwd_wn(tune);
  const int wn = (wn_t == 8 && !kn && M <= TM) ? 8 : ((kn || plain) ? 4 : (wn_t == 8 ? 4 : wn_t)), tn = 32 * wn;
  const int s = pick_splitk_x(M, N, Kp, wn, tune);
  // W is addressed per N tile (or per K slice when stored [K, N]) through a rebased descriptor: only that block must
  // stay below 4 GiB, not the weight
  FX_REQUIRE((kn ? (long)(((Kp / TK + s - 1) / s) * TK + 4 * TK) : (long)tn) * ldw * 4 < 0xF0000000L && (long)Kp * ldx * 2 < 0xF0000000L,
             "fx_linear_fwd_bf16x3: operand block exceeds 4 GiB");
  FX_REQUIRE(workspace && workspace_bytes >= (long)s * M * N * (long)sizeof(float), "fx_linear_fwd_bf16x3: workspace too small");
  XGemmArgs g{};
  g.Ahi = (const __bf16*)xhi; g.Alo = (const __bf16*)xlo;
  g.Bf = W;
  g.C = (float*)workspace;
  g.M = M; g.N = N; g.K = Kp; g.Ktrue = K;
  g.lda = 0; g.a_rp = ldx; g.ldb = ldw; g.ldc = N;
  g.splitk = s;
  g.kchunk = ((Kp / TK + s - 1) / s) * TK;
  g.slab_stride = (long)M * N;
  const long nblk = (long)((M + TM - 1) / TM) * ((N + tn - 1) / tn) * s;
  FX_REQUIRE(nblk < (1L << 31), "fx_linear_fwd_bf16x3: grid too large");
  const int nt = tune.nt;
  if (!kn && wn == 4 && M > TM && tune.mt != 1) {
    const int mt = fwd_mt(M);
    const long nb = (long)((M + mt * TM - 1) / (mt * TM)) * ((N + 127) / 128) * s;
    FX_REQUIRE(nb < (1L << 31), "fx_linear_fwd_bf16x3: grid too large");
    if (tune.mt == 2) {                                   // A/B: the first design, both operands staged through registers
      if (mt == 3) hipLaunchKernelGGL((fx_fwd_bf16x3_mt_kernel<3>), dim3((unsigned)nb), dim3(512), 0, stream, g);
      else hipLaunchKernelGGL((fx_fwd_bf16x3_mt_kernel<2>), dim3((unsigned)nb), dim3(512), 0, stream, g);
    } else if (tune.mt == 3) {                            // A/B: the second design, X by LDS-DMA
      if (mt == 3) hipLaunchKernelGGL((fx_fwd_bf16x3_dma_kernel<3, 0>), dim3((unsigned)nb), dim3(512), 0, stream, g);
      else hipLaunchKernelGGL((fx_fwd_bf16x3_dma_kernel<2, 0>), dim3((unsigned)nb), dim3(512), 0, stream, g);
    } else if (mt == 3) {                                 // X fragments straight into registers
      if (plain) hipLaunchKernelGGL((fx_fwd_bf16x3_reg_kernel<3, 8, 0, 1>), dim3((unsigned)nb), dim3(512), 0, stream, g);
      else if (nt) hipLaunchKernelGGL((fx_fwd_bf16x3_reg_kernel<3, 8, 2>), dim3((unsigned)nb), dim3(512), 0, stream, g);
      else hipLaunchKernelGGL((fx_fwd_bf16x3_reg_kernel<3, 8, 0>), dim3((unsigned)nb), dim3(512), 0, stream, g);
    } else {
      if (plain) hipLaunchKernelGGL((fx_fwd_bf16x3_reg_kernel<2, 8, 0, 1>), dim3((unsigned)nb), dim3(512), 0, stream, g);
      else if (nt) hipLaunchKernelGGL((fx_fwd_bf16x3_reg_kernel<2, 8, 2>), dim3((unsigned)nb), dim3(512), 0, stream, g);
      else hipLaunchKernelGGL((fx_fwd_bf16x3_reg_kernel<2, 8, 0>), dim3((unsigned)nb), dim3(512), 0, stream, g);
    }
  } else if (!kn && wn == 4 && tune.mt != 1 && (M <= 64 || tune.mt == 4)) {
    // at most 64 rows: the register-fragment kernel with four waves of 16 rows, two workgroups per CU (4.5-5.1 TB/s of W where the
    // 128-row tile below streams 4.2-4.5); 65..128 rows: the two measure the same (both sit on the package power limit), FX_FWD_MT=4 is the A/B
    if (M > 64 && plain) hipLaunchKernelGGL((fx_fwd_bf16x3_reg_kernel<2, 4, 0, 1>), dim3((unsigned)nblk), dim3(256), 0, stream, g);
    else if (M > 64) hipLaunchKernelGGL((fx_fwd_bf16x3_reg_kernel<2, 4, 0>), dim3((unsigned)nblk), dim3(256), 0, stream, g);
    else if (plain) hipLaunchKernelGGL((fx_fwd_bf16x3_reg_kernel<1, 4, 0, 1>), dim3((unsigned)nblk), dim3(256), 0, stream, g);
    else if (nt) hipLaunchKernelGGL((fx_fwd_bf16x3_reg_kernel<1, 4, 2>), dim3((unsigned)nblk), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((fx_fwd_bf16x3_reg_kernel<1, 4, 0>), dim3((unsigned)nblk), dim3(256), 0, stream, g);
  } else if (wn == 8) {
    if (plain) hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<true, XEPI_STORE, 0, 8, false, 1>), dim3((unsigned)nblk), dim3(1024), 0, stream, g);
    else hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<true, XEPI_STORE, 0, 8, false, 3>), dim3((unsigned)nblk), dim3(1024), 0, stream, g);
  } else if (kn) {
    if (plain) hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<true, XEPI_STORE, 0, 4, true, 1>), dim3((unsigned)nblk), dim3(512), 0, stream, g);
    else hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<true, XEPI_STORE, 0, 4, true>), dim3((unsigned)nblk), dim3(512), 0, stream, g);
  } else if (wn == 4) {
    if (plain) hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<true, XEPI_STORE, 0, 4, false, 1>), dim3((unsigned)nblk), dim3(512), 0, stream, g);
    else if (nt) hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<true, XEPI_STORE, 2, 4>), dim3((unsigned)nblk), dim3(512), 0, stream, g);
    else hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<true, XEPI_STORE, 0, 4>), dim3((unsigned)nblk), dim3(512), 0, stream, g);
  } else {
    if (nt) hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<true, XEPI_STORE, 2, 2>), dim3((unsigned)nblk), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<true, XEPI_STORE, 0, 2>), dim3((unsigned)nblk), dim3(256), 0, stream, g);
  }
  int rc = fx_check_launch("fx_linear_fwd_bf16x3");
  if (rc || !reduce) return rc;
  fx_launch_reduce_slabs(Y, (const float*)workspace, bias, M, N, ldy, s, g.slab_stride, 0, stream);
  return fx_check_launch("fx_reduce_slabs");
}

extern "C" {

static int dw_adam_bf16x3_impl(float* W, float* adam_m, float* adam_v, const void* dyT_hi, const void* dyT_lo,
                               const void* xT_hi, const void* xT_lo, int batch_padded, int n_out, int k_in, long lddy,
                               long ldx, long ldw, const float* ctrl, int tile_order, int wave_cols, int plain_loads,
                               hipStream_t stream) {
  FX_REQUIRE(W && adam_m && adam_v && dyT_hi && xT_hi && ctrl, "fx_linear_dw_adam_bf16x3: null pointer");
  const bool plain = dyT_lo == nullptr;       // no `lo` operands: plain bf16 (one product)
  FX_REQUIRE((xT_lo == nullptr) == plain, "fx_linear_dw_adam_bf16x3: dyT_lo and xT_lo must be both given (split bf16) or both NULL (plain bf16)");
  FX_REQUIRE(batch_padded > 0 && batch_padded % TK == 0, "fx_linear_dw_adam_bf16x3: padded batch %d must be a multiple of %d",
             batch_padded, TK);
  FX_REQUIRE(lddy % 8 == 0 && ldx % 8 == 0 && aligned16(dyT_hi) && aligned16(dyT_lo) && aligned16(xT_hi) && aligned16(xT_lo),
             "fx_linear_dw_adam_bf16x3: operands must be 16-byte aligned with ld %% 8 == 0");
  // W / m / v are addressed per 128-row tile through rebased descriptors: the weight itself may exceed 4 GiB
  FX_REQUIRE((long)TM * ldw * 4 < 0xF0000000L && (long)n_out * lddy * 2 < 0xF0000000L && (long)k_in * ldx * 2 < 0xF0000000L,
             "fx_linear_dw_adam_bf16x3: operand block exceeds 4 GiB");
  FX_REQUIRE(tile_order >= 0 && tile_order <= 2, "fx_linear_dw_adam_bf16x3: tile_order is 0 (auto), 1 (linear) or 2 (XCD-partitioned)");
  XGemmArgs g{};
  g.Ahi = (const __bf16*)dyT_hi; g.Alo = (const __bf16*)dyT_lo;
  g.Bhi = (const __bf16*)xT_hi; g.Blo = (const __bf16*)xT_lo;
  g.C = W;
  g.M = n_out; g.N = k_in; g.K = batch_padded; g.Ktrue = batch_padded;
  g.lda = lddy; g.ldb = ldx; g.ldc = ldw;
  g.splitk = 1; g.kchunk = batch_padded;
  g.n_fast = n_out > k_in;
  g.adam_m = adam_m; g.adam_v = adam_v; g.ctrl = ctrl;
  const int wn = (wave_cols == 2 && !plain) ? 2 : 4, tn = 32 * wn;
  long nblk = (long)((n_out + TM - 1) / TM) * ((k_in + tn - 1) / tn);
  if (tile_order != 1) {
    const int tiles_m = (n_out + TM - 1) / TM, tiles_n = (k_in + tn - 1) / tn;
    const int longer = tiles_n >= tiles_m ? tiles_n : tiles_m, shorter = tiles_n >= tiles_m ? tiles_m : tiles_n;
    // group size: the group's operand rows (tile rows x K x (hi + lo)) should fill about 1.5 MB of the 4 MB L2
    const long tile_bytes = (long)(tiles_n >= tiles_m ? TM : tn) * batch_padded * 4;
    // Only when the shorter dimension's whole operand does not fit one L2 anyway (measured: cfg2 / cfg3 shapes, 2.5 MB,
    // are as fast or faster in the linear order; cfg4's 11.5 MB operand costs 9.0 instead of 5.9 GB of HBM traffic)
    if (longer >= 16 && (tile_order == 2 || (long)shorter * tile_bytes > (3L << 20))) {
      int G = (int)((3L << 19) / tile_bytes);
      if (G < 1) G = 1;
      if (G > shorter) G = shorter;
      g.xcd_blk = G;
      const int per = (longer + 7) / 8, groups = (shorter + G - 1) / G;
      nblk = 8L * per * groups * G;
    }
  }
  FX_REQUIRE(nblk < (1L << 31), "fx_linear_dw_adam_bf16x3: grid too large");
  if (plain) {
    if (!plain_loads) hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<false, XEPI_ADAM, 2, 4, false, 1>), dim3((unsigned)nblk), dim3(512), 0, stream, g);
    else hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<false, XEPI_ADAM, 0, 4, false, 1>), dim3((unsigned)nblk), dim3(512), 0, stream, g);
  } else if (wn == 4) {
    if (!plain_loads) hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<false, XEPI_ADAM, 2, 4>), dim3((unsigned)nblk), dim3(512), 0, stream, g);
    else hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<false, XEPI_ADAM, 0, 4>), dim3((unsigned)nblk), dim3(512), 0, stream, g);
  } else {
    if (!plain_loads) hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<false, XEPI_ADAM, 2, 2>), dim3((unsigned)nblk), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((fx_gemm_bf16x3_kernel<false, XEPI_ADAM, 0, 2>), dim3((unsigned)nblk), dim3(256), 0, stream, g);
  }
  return fx_check_launch("fx_linear_dw_adam_bf16x3");
}

int fx_linear_dw_adam_bf16x3(float* W, float* adam_m, float* adam_v, const void* dyT_hi, const void* dyT_lo,
                             const void* xT_hi, const void* xT_lo, int batch_padded, int n_out, int k_in, long lddy,
                             long ldx, long ldw, const float* ctrl, hipStream_t stream) {
  return dw_adam_bf16x3_impl(W, adam_m, adam_v, dyT_hi, dyT_lo, xT_hi, xT_lo, batch_padded, n_out, k_in, lddy, ldx, ldw, ctrl, 0, 0,
                             0, stream);
}

// The same with explicit tuning: tile_order 0 = auto, 1 = linear, 2 = XCD-partitioned / L2-blocked (bit-identical results);
// wave_cols 0|4 = 128x128 tile, 2 = 128x64; plain_loads 1 = default cache policy instead of non-temporal W / m / v accesses.
int fx_linear_dw_adam_bf16x3_ex(float* W, float* adam_m, float* adam_v, const void* dyT_hi, const void* dyT_lo,
                                const void* xT_hi, const void* xT_lo, int batch_padded, int n_out, int k_in, long lddy,
                                long ldx, long ldw, const float* ctrl, int tile_order, int wave_cols, int plain_loads,
                                hipStream_t stream) {
  return dw_adam_bf16x3_impl(W, adam_m, adam_v, dyT_hi, dyT_lo, xT_hi, xT_lo, batch_padded, n_out, k_in, lddy, ldx, ldw, ctrl,
                             tile_order, wave_cols, plain_loads, stream);
}

}  // extern "C"
