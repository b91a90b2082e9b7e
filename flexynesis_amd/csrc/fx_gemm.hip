// fp32 MFMA GEMM family for the Linear layers of the flexynesis hot path (gfx950 / CDNA4).
//
//   C[m,n] (op)= sum_k Aop(m,k) * Bop(k,n)
//
// All three Linear contractions are one kernel template, selected by which operand dimension is
// contiguous in memory (reference shapes: SURVEY.md section 2.3):
//   NT  (A_KC=1,B_KC=1)  forward    Y[B,H]   = X[B,F]   . W[H,F]^T      (K1/K3/K5/K7/K8)
//   NN  (A_KC=1,B_KC=0)  bwd-data   dX[B,F]  = dY[B,H]  . W[H,F]        (autograd mm)
//   TN  (A_KC=0,B_KC=0)  bwd-weight dW[H,F]  = dY[B,H]^T. X[B,F]        (K18), optionally with the
//        clip+Adam update applied in the epilogue so dW is never written to HBM (K16+K17+K18 fused).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- f32 in / f32 accumulate, bit-for-bit an fmaf chain, i.e. the
// same numerics class as the reference's fp32 CPU matmul.  This is the kernel behind every NARROW layer
// (latent / fusion / head Linear layers, Gram matrices) in both precision modes and behind the wide
// layers in precision="f32"; the wide layers' default path is fx_gemm_bf16x3.hip.
//
// Tiling: 128(M) x 64(N) x 32(K) per 256-thread workgroup (4 waves; each wave owns a 64x32 slice =
// two 32x32 MFMA blocks).  Operands are staged through LDS with one register-prefetched tile in flight.
// The LDS image keeps the operand's memory-contiguous dimension innermost; for K-contiguous operands
// rows are padded to 33 floats so the 32-lane ds_read_b32 fragment reads hit 32 distinct banks.
//
// Addressing: every global access is a raw buffer load/store through an SRSRC descriptor.  Out-of-range
// rows fall beyond num_records (hardware returns 0 / drops the store); out-of-range positions inside a
// row are pushed out of range by OR-ing 0xFFFFFFF0 into the byte offset.  No memory instruction sits
// behind a per-lane branch (that pattern makes hipcc drain vmcnt(0) around every load).  VEC = both
// operands 16-byte aligned with extents divisible by 4 -> dwordx4 loads; otherwise dword loads with
// per-element masks.
//
// Split-K: grid = tiles_m * tiles_n * splitk, K-slice index fastest (K-slice <-> XCD affinity for the
// shared activation operand).  Partial sums go to fp32 slabs combined in a fixed order by
// fx_reduce_slabs (deterministic; no float atomics).
#include "fx_common.h"
#include "fx_reduce.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 64
#define BK 32
#define PADK (BK + 1)
#define OOB 0xFFFFFFF0u

enum { EPI_STORE = 0, EPI_ADAM = 1 };

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;  // [N] or null (only applied when splitk == 1)
  int M, N, K;
  long lda, ldb, ldc;
  int splitk, kchunk;
  long slab_stride;  // elements between split-K slabs in C (splitk > 1)
  int accumulate;    // C += (splitk == 1 only)
  // EPI_ADAM: C is the parameter tensor; m/v its Adam moments; ctrl the step control block
  float* adam_m;
  float* adam_v;
  const float* ctrl;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t g_rsrc(const void* p, long bytes) {
  const unsigned n = bytes > 0xFFFFFFF0L ? 0xFFFFFFF0u : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, n, 0x00020000);
}
__device__ __forceinline__ f32x4 g_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float g_ld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void g_st1(float v, __amdgpu_buffer_rsrc_t r, unsigned off) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}

// One operand element group: 4 consecutive elements along the contiguous dimension starting at `c`,
// in row `row` of a [rows, ld] array.  `c_end` bounds the contiguous dimension (K-slice end or M/N).
template <bool VEC>
__device__ __forceinline__ f32x4 ld_group(__amdgpu_buffer_rsrc_t rs, long row, long ld, int c, int c_end) {
  const unsigned base = (unsigned)((row * ld + c) * 4);
  if (VEC) return g_ld4(rs, base | (c < c_end ? 0u : OOB));
  f32x4 v;
  v[0] = g_ld1(rs, base | (c < c_end ? 0u : OOB));
  v[1] = g_ld1(rs, (base + 4) | (c + 1 < c_end ? 0u : OOB));
  v[2] = g_ld1(rs, (base + 8) | (c + 2 < c_end ? 0u : OOB));
  v[3] = g_ld1(rs, (base + 12) | (c + 3 < c_end ? 0u : OOB));
  return v;
}

template <bool A_KC, bool B_KC, int EPI, bool VEC>
__global__ __launch_bounds__(256) void fx_gemm_f32_kernel(GemmArgs g) {
  // one LDS object (cdna guide: a second __shared__ object de-pipelines waits)
  __shared__ __attribute__((aligned(16))) float smem[BM * PADK + BN * PADK];
  float* As = smem;
  float* Bs = smem + BM * PADK;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wr = wid & 1, wc = wid >> 1;

  const int tiles_m = (g.M + BM - 1) / BM;
  int lin = blockIdx.x;
  const int z = lin % g.splitk;
  lin /= g.splitk;
  const int tm = lin % tiles_m, tn = lin / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int k_begin = z * g.kchunk;
  const int k_end = min(g.K, k_begin + g.kchunk);
  const int nk = (k_end > k_begin) ? (k_end - k_begin + BK - 1) / BK : 0;

  // rows of a K-contiguous operand are M (or N); rows of the other kind are K
  const __amdgpu_buffer_rsrc_t rA = g_rsrc(g.A, (long)(A_KC ? g.M : g.K) * g.lda * 4);
  const __amdgpu_buffer_rsrc_t rB = g_rsrc(g.B, (long)(B_KC ? g.N : g.K) * g.ldb * 4);

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

  f32x4 ra0, ra1, ra2, ra3, rb0, rb1;

  // A tile = 128x32 floats = 1024 groups of 4 -> 4 per thread; B tile = 64x32 = 512 groups -> 2 per thread.
  // Loads are unconditional (tile index may run one past the slice: masked to zero / never stored).
#define LOAD_TILE(kt)                                                                               \
  {                                                                                                 \
    const int k0 = k_begin + (kt) * BK;                                                             \
    if (A_KC) { /* A stored [M,K]: groups along k; row = m */                                       \
      ra0 = ld_group<VEC>(rA, m0 + ((tid + 0) >> 3), g.lda, k0 + 4 * (tid & 7), k_end);             \
      ra1 = ld_group<VEC>(rA, m0 + ((tid + 256) >> 3), g.lda, k0 + 4 * (tid & 7), k_end);           \
      ra2 = ld_group<VEC>(rA, m0 + ((tid + 512) >> 3), g.lda, k0 + 4 * (tid & 7), k_end);           \
      ra3 = ld_group<VEC>(rA, m0 + ((tid + 768) >> 3), g.lda, k0 + 4 * (tid & 7), k_end);           \
    } else { /* A stored [K,M]: groups along m; row = k (rows >= K read as zero by the range check) */ \
      ra0 = ld_group<VEC>(rA, k0 + (tid >> 5), g.lda, m0 + 4 * (tid & 31), g.M);                    \
      ra1 = ld_group<VEC>(rA, k0 + 8 + (tid >> 5), g.lda, m0 + 4 * (tid & 31), g.M);                \
      ra2 = ld_group<VEC>(rA, k0 + 16 + (tid >> 5), g.lda, m0 + 4 * (tid & 31), g.M);               \
      ra3 = ld_group<VEC>(rA, k0 + 24 + (tid >> 5), g.lda, m0 + 4 * (tid & 31), g.M);               \
    }                                                                                               \
    if (B_KC) { /* B stored [N,K] */                                                                \
      rb0 = ld_group<VEC>(rB, n0 + (tid >> 3), g.ldb, k0 + 4 * (tid & 7), k_end);                   \
      rb1 = ld_group<VEC>(rB, n0 + 32 + (tid >> 3), g.ldb, k0 + 4 * (tid & 7), k_end);              \
    } else { /* B stored [K,N] */                                                                   \
      rb0 = ld_group<VEC>(rB, k0 + (tid >> 4), g.ldb, n0 + 4 * (tid & 15), g.N);                    \
      rb1 = ld_group<VEC>(rB, k0 + 16 + (tid >> 4), g.ldb, n0 + 4 * (tid & 15), g.N);               \
    }                                                                                               \
  }
  // NB (!KC operands): a K row index >= K lands beyond num_records; a row index between k_end and K can only
  // occur in the over-fetched tile after the last K-step, whose registers are never stored to LDS.

  auto st_kc = [&](float* S, int row, int k4, const f32x4 v) {
    float* d = S + row * PADK + 4 * k4;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  };
#define STORE_TILE()                                                                  \
  {                                                                                   \
    if (A_KC) {                                                                       \
      st_kc(As, (tid + 0) >> 3, tid & 7, ra0);                                        \
      st_kc(As, (tid + 256) >> 3, tid & 7, ra1);                                      \
      st_kc(As, (tid + 512) >> 3, tid & 7, ra2);                                      \
      st_kc(As, (tid + 768) >> 3, tid & 7, ra3);                                      \
    } else {                                                                          \
      *reinterpret_cast<f32x4*>(As + ((tid >> 5)) * BM + 4 * (tid & 31)) = ra0;       \
      *reinterpret_cast<f32x4*>(As + (8 + (tid >> 5)) * BM + 4 * (tid & 31)) = ra1;   \
      *reinterpret_cast<f32x4*>(As + (16 + (tid >> 5)) * BM + 4 * (tid & 31)) = ra2;  \
      *reinterpret_cast<f32x4*>(As + (24 + (tid >> 5)) * BM + 4 * (tid & 31)) = ra3;  \
    }                                                                                 \
    if (B_KC) {                                                                       \
      st_kc(Bs, tid >> 3, tid & 7, rb0);                                              \
      st_kc(Bs, 32 + (tid >> 3), tid & 7, rb1);                                       \
    } else {                                                                          \
      *reinterpret_cast<f32x4*>(Bs + (tid >> 4) * BN + 4 * (tid & 15)) = rb0;         \
      *reinterpret_cast<f32x4*>(Bs + (16 + (tid >> 4)) * BN + 4 * (tid & 15)) = rb1;  \
    }                                                                                 \
  }

  const int fm = wr * 64 + (lane & 31);  // fragment row within the tile (second block: +32)
  const int fn = wc * 32 + (lane & 31);
  const int fk = lane >> 5;

  LOAD_TILE(0);
  for (int kt = 0; kt < nk; ++kt) {
    STORE_TILE();
    __syncthreads();
    LOAD_TILE(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int k = 2 * kk + fk;
      const float a0 = A_KC ? As[fm * PADK + k] : As[k * BM + fm];
      const float a1 = A_KC ? As[(fm + 32) * PADK + k] : As[k * BM + fm + 32];
      const float b = B_KC ? Bs[fn * PADK + k] : Bs[k * BN + fn];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int n = n0 + wc * 32 + (lane & 31);
  const unsigned oob = (n < g.N) ? 0u : OOB;
  const long cbytes = (long)g.M * g.ldc * 4;
  if (EPI == EPI_STORE) {
    const __amdgpu_buffer_rsrc_t rC = g_rsrc(g.C + (long)z * g.slab_stride, cbytes);
    const float bv = (g.bias != nullptr && g.splitk == 1 && n < g.N) ? g.bias[min(n, g.N - 1)] : 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int mbase = m0 + wr * 64 + blk * 32 + 4 * (lane >> 5);
      float old[16];
      if (g.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          old[r] = g_ld1(rC, (unsigned)(((long)(mbase + (r & 3) + 8 * (r >> 2)) * g.ldc + n) * 4) | oob);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned off = (unsigned)(((long)(mbase + (r & 3) + 8 * (r >> 2)) * g.ldc + n) * 4) | oob;
        float val = (blk == 0 ? acc0[r] : acc1[r]) + bv;
        if (g.accumulate) val += old[r];
        g_st1(val, rC, off);
      }
    }
  } else {  // EPI_ADAM: fused clip + Adam on the parameter tile (torch.optim.Adam defaults)
    const __amdgpu_buffer_rsrc_t rP = g_rsrc(g.C, cbytes), rM = g_rsrc(g.adam_m, cbytes), rV = g_rsrc(g.adam_v, cbytes);
    const FxAdamK ak = fx_adam_consts(g.ctrl[FXC_LR], g.ctrl[FXC_BC1], g.ctrl[FXC_BC2_SQRT], g.ctrl[FXC_CLIP_COEF]);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int mbase = m0 + wr * 64 + blk * 32 + 4 * (lane >> 5);
      float pv[16], mv[16], vv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned off = (unsigned)(((long)(mbase + (r & 3) + 8 * (r >> 2)) * g.ldc + n) * 4) | oob;
        pv[r] = g_ld1(rP, off);
        mv[r] = g_ld1(rM, off);
        vv[r] = g_ld1(rV, off);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned off = (unsigned)(((long)(mbase + (r & 3) + 8 * (r >> 2)) * g.ldc + n) * 4) | oob;
        float pj = pv[r], mj = mv[r], vj = vv[r];
        fx_adam_update(pj, mj, vj, blk == 0 ? acc0[r] : acc1[r], ak);
        g_st1(pj, rP, off);
        g_st1(mj, rM, off);
        g_st1(vj, rV, off);
      }
    }
  }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <bool A_KC, bool B_KC>
static int launch_gemm(GemmArgs& g, int epi, bool vec, hipStream_t stream) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const long nblk = (long)tiles_m * tiles_n * g.splitk;
  FX_REQUIRE(nblk > 0 && nblk < (1L << 31), "fx_gemm: bad grid (%ld blocks)", nblk);
  const dim3 grid((unsigned)nblk), block(256);
  if (epi == EPI_STORE) {
    if (vec) hipLaunchKernelGGL((fx_gemm_f32_kernel<A_KC, B_KC, EPI_STORE, true>), grid, block, 0, stream, g);
    else hipLaunchKernelGGL((fx_gemm_f32_kernel<A_KC, B_KC, EPI_STORE, false>), grid, block, 0, stream, g);
  } else {
    if (vec) hipLaunchKernelGGL((fx_gemm_f32_kernel<A_KC, B_KC, EPI_ADAM, true>), grid, block, 0, stream, g);
    else hipLaunchKernelGGL((fx_gemm_f32_kernel<A_KC, B_KC, EPI_ADAM, false>), grid, block, 0, stream, g);
  }
  return fx_check_launch("fx_gemm");
}

// dwordx4 operand loads need 16-byte aligned rows and whole groups of 4 along the contiguous dimension
static bool vec_ok(const GemmArgs& g, int a_kc, int b_kc) {
  const bool a = aligned16(g.A) && (g.lda % 4 == 0) && ((a_kc ? g.K : g.M) % 4 == 0);
  const bool b = aligned16(g.B) && (g.ldb % 4 == 0) && ((b_kc ? g.K : g.N) % 4 == 0);
  return a && b;
}

static int gemm_dispatch(int a_kc, int b_kc, GemmArgs& g, int epi, hipStream_t stream) {
  const bool vec = vec_ok(g, a_kc, b_kc);
  // 3.75 GiB: leaves room for the (<= 128 rows) a tile may reach past the tensor before the 32-bit offset wraps
  FX_REQUIRE((long)(a_kc ? g.M : g.K) * g.lda * 4 < 0xF0000000L && (long)(b_kc ? g.N : g.K) * g.ldb * 4 < 0xF0000000L &&
                 (long)g.M * g.ldc * 4 < 0xF0000000L,
             "fx_gemm: operand exceeds the 4 GiB buffer-descriptor range");
  if (a_kc && b_kc) return launch_gemm<true, true>(g, epi, vec, stream);
  if (a_kc && !b_kc) return launch_gemm<true, false>(g, epi, vec, stream);
  if (!a_kc && !b_kc) return launch_gemm<false, false>(g, epi, vec, stream);
  return launch_gemm<false, true>(g, epi, vec, stream);
}

// Heuristic: enough workgroups to fill 256 CUs a few times over, K-slices a multiple of BK.
static int pick_splitk(int M, int N, int K) {
  const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (tiles >= 512 || K <= 4 * BK) return 1;
  int s = (int)((1024 + tiles - 1) / tiles);
  const int maxs = (K + 4 * BK - 1) / (4 * BK);  // keep >= 4 K-steps per slice
  if (s > maxs) s = maxs;
  if (s > 64) s = 64;
  if (s >= 8) s = (s / 8) * 8;  // keep K-slice <-> XCD affinity
  return s < 1 ? 1 : s;
}

extern "C" {

long fx_gemm_workspace_bytes(int M, int N, int K) {
  const int s = pick_splitk(M, N, K);
  return s > 1 ? (long)s * M * N * (long)sizeof(float) : 0;
}

// layout: 0 = NT (A[M,K], B[N,K]), 1 = NN (A[M,K], B[K,N]), 2 = TN (A[K,M], B[K,N])
int fx_gemm_f32(int layout, float* C, const float* A, const float* B, const float* bias, int M, int N, int K, long lda,
                long ldb, long ldc, int accumulate, void* workspace, long workspace_bytes, hipStream_t stream) {
  FX_REQUIRE(layout >= 0 && layout <= 2, "fx_gemm_f32: layout %d", layout);
  FX_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "fx_gemm_f32: bad shape/pointer M=%d N=%d K=%d", M, N, K);
  const int a_kc = layout != 2, b_kc = layout == 0;
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  int s = pick_splitk(M, N, K);
  if (s > 1 && (workspace == nullptr || workspace_bytes < (long)s * M * N * (long)sizeof(float))) s = 1;
  g.splitk = s;
  g.kchunk = ((K + s - 1) / s + BK - 1) / BK * BK;
  if (s == 1) {
    g.accumulate = accumulate;
    return gemm_dispatch(a_kc, b_kc, g, EPI_STORE, stream);
  }
  g.C = (float*)workspace;
  g.ldc = N;
  g.slab_stride = (long)M * N;
  g.accumulate = 0;
  int rc = gemm_dispatch(a_kc, b_kc, g, EPI_STORE, stream);
  if (rc) return rc;
  fx_launch_reduce_slabs(C, (const float*)workspace, bias, M, N, ldc, s, g.slab_stride, accumulate, stream);
  return fx_check_launch("fx_reduce_slabs");
}

// Fused weight-gradient + global-norm-clip + Adam for one Linear weight W[N_out, K_in] (row-major):
//   dW = dY[B,N_out]^T . X[B,K_in] computed tile by tile on the MFMA and consumed in registers;
//   W, m, v are streamed exactly once (24 B/param); dW never touches HBM.
// ctrl supplies lr, bias corrections and the clip coefficient (fx_step_begin / fx_clip_finalize).
int fx_linear_dw_adam_f32(float* W, float* adam_m, float* adam_v, const float* dY, const float* X, int batch, int n_out,
                          int k_in, long lddy, long ldx, long ldw, const float* ctrl, hipStream_t stream) {
  FX_REQUIRE(W && adam_m && adam_v && dY && X && ctrl, "fx_linear_dw_adam_f32: null pointer");
  FX_REQUIRE(batch > 0 && n_out > 0 && k_in > 0, "fx_linear_dw_adam_f32: bad shape");
  GemmArgs g{};
  g.A = dY; g.B = X; g.C = W;
  g.M = n_out; g.N = k_in; g.K = batch;
  g.lda = lddy; g.ldb = ldx; g.ldc = ldw;
  g.splitk = 1;
  g.kchunk = (batch + BK - 1) / BK * BK;
  g.adam_m = adam_m; g.adam_v = adam_v; g.ctrl = ctrl;
  return gemm_dispatch(0, 0, g, EPI_ADAM, stream);
}

int fx_gemm_splitk(int M, int N, int K) { return pick_splitk(M, N, K); }

// Same contraction as fx_gemm_f32 but the split-K partial sums are LEFT in `slabs` ([fx_gemm_splitk(M,N,K)][M][N],
// contiguous) for a consumer that folds the reduction into its own pass (fx_gram_hadamard, fx_bn_act_fwd_slabs).
int fx_gemm_f32_slabs(int layout, float* slabs, const float* A, const float* B, int M, int N, int K, long lda, long ldb,
                      hipStream_t stream) {
  FX_REQUIRE(layout >= 0 && layout <= 2, "fx_gemm_f32_slabs: layout %d", layout);
  FX_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && slabs, "fx_gemm_f32_slabs: bad shape/pointer");
  const int a_kc = layout != 2, b_kc = layout == 0;
  GemmArgs g{};
  g.A = A; g.B = B; g.C = slabs;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = N;
  const int s = pick_splitk(M, N, K);
  g.splitk = s;
  g.kchunk = ((K + s - 1) / s + BK - 1) / BK * BK;
  g.slab_stride = (long)M * N;
  return gemm_dispatch(a_kc, b_kc, g, EPI_STORE, stream);
}

}  // extern "C"
