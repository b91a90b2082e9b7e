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
// same numerics class as the reference's fp32 CPU matmul (needed for the 1e-4 loss-parity gate).
//
// Tiling: 128(M) x 64(N) x 32(K) per 256-thread workgroup (4 waves; each wave owns a 64x32 slice =
// two 32x32 MFMA blocks -> 32 accumulator VGPRs).  Operands are staged through LDS with one
// register-prefetched tile in flight (global_load_dwordx4 issued before the MFMA phase of the
// previous tile).  The LDS image keeps the operand's memory-contiguous dimension innermost; for
// K-contiguous operands rows are padded to 33 floats so the 32-lane ds_read_b32 fragment reads
// (lane -> row) hit 32 distinct banks.
//
// Split-K: grid = tiles_m * tiles_n * splitk, decoded so that consecutive workgroup ids differ in the
// K-slice first: with splitk == 8 every XCD (workgroup id mod 8) streams ONE K-slice of the shared
// [B,K] activation operand, which then stays in that XCD's 4 MiB L2.  Partial sums go to fp32 slabs
// and are combined in a fixed order by fx_reduce_slabs (deterministic; no float atomics).
#include "fx_common.h"
#include "fx_reduce.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 64
#define BK 32
#define PADK (BK + 1)

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
  int vecA, vecB;
  int accumulate;  // C += (splitk == 1 only)
  // EPI_ADAM: C is the parameter tensor; m/v its Adam moments; ctrl the step control block
  float* adam_m;
  float* adam_v;
  const float* ctrl;
};

__device__ __forceinline__ float4 fx_ld4(const float* __restrict__ base, long row_off, bool row_ok, int c0,
                                         int c_end, bool vec) {
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!row_ok || c0 >= c_end) return r;
  const float* p = base + row_off + c0;
  if (vec && c0 + 3 < c_end) return *reinterpret_cast<const float4*>(p);
  r.x = p[0];
  if (c0 + 1 < c_end) r.y = p[1];
  if (c0 + 2 < c_end) r.z = p[2];
  if (c0 + 3 < c_end) r.w = p[3];
  return r;
}

template <bool A_KC, bool B_KC, int EPI>
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

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

  float4 ra[4], rb[2];
  const bool vA = g.vecA != 0, vB = g.vecB != 0;

  auto load_tile = [&](int kt) {
    const int k0 = k_begin + kt * BK;
    if (A_KC) {  // A stored [M,K]: float4 along k
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256, m = f >> 3, k4 = f & 7;
        ra[i] = fx_ld4(g.A, (long)(m0 + m) * g.lda, (m0 + m) < g.M, k0 + 4 * k4, k_end, vA);
      }
    } else {  // A stored [K,M]: float4 along m
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256, k = f >> 5, m4 = f & 31;
        ra[i] = fx_ld4(g.A, (long)(k0 + k) * g.lda, (k0 + k) < k_end, m0 + 4 * m4, g.M, vA);
      }
    }
    if (B_KC) {  // B stored [N,K]
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int f = tid + i * 256, n = f >> 3, k4 = f & 7;
        rb[i] = fx_ld4(g.B, (long)(n0 + n) * g.ldb, (n0 + n) < g.N, k0 + 4 * k4, k_end, vB);
      }
    } else {  // B stored [K,N]
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int f = tid + i * 256, k = f >> 4, n4 = f & 15;
        rb[i] = fx_ld4(g.B, (long)(k0 + k) * g.ldb, (k0 + k) < k_end, n0 + 4 * n4, g.N, vB);
      }
    }
  };
  auto store_tile = [&]() {
    if (A_KC) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256, m = f >> 3, k4 = f & 7;
        float* d = As + m * PADK + 4 * k4;
        d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256, k = f >> 5, m4 = f & 31;
        *reinterpret_cast<float4*>(As + k * BM + 4 * m4) = ra[i];
      }
    }
    if (B_KC) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int f = tid + i * 256, n = f >> 3, k4 = f & 7;
        float* d = Bs + n * PADK + 4 * k4;
        d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int f = tid + i * 256, k = f >> 4, n4 = f & 15;
        *reinterpret_cast<float4*>(Bs + k * BN + 4 * n4) = rb[i];
      }
    }
  };

  const int fm = wr * 64 + (lane & 31);  // fragment row within the tile (second block: +32)
  const int fn = wc * 32 + (lane & 31);
  const int fk = lane >> 5;

  if (nk > 0) load_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    store_tile();
    __syncthreads();
    if (kt + 1 < nk) load_tile(kt + 1);
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
  if (n >= g.N) return;
  if (EPI == EPI_STORE) {
    float* Cz = g.C + (long)z * g.slab_stride;
    const float bv = (g.bias != nullptr && g.splitk == 1) ? g.bias[n] : 0.f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 64 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < g.M) {
          float val = (blk == 0 ? acc0[r] : acc1[r]) + bv;
          float* dst = Cz + (long)m * g.ldc + n;
          if (g.accumulate) val += *dst;
          *dst = val;
        }
      }
    }
  } else {  // EPI_ADAM: fused clip + Adam on the parameter tile (torch.optim.Adam defaults)
    const float lr = g.ctrl[FXC_LR], bc1 = g.ctrl[FXC_BC1], bc2s = g.ctrl[FXC_BC2_SQRT];
    const float coef = g.ctrl[FXC_CLIP_COEF];
    const float step_size = lr / bc1;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      float pv[16], mv[16], vv[16];
      long off[16];
      bool ok[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 64 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        ok[r] = m < g.M;
        off[r] = (long)m * g.ldc + n;
        if (ok[r]) { pv[r] = g.C[off[r]]; mv[r] = g.adam_m[off[r]]; vv[r] = g.adam_v[off[r]]; }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (ok[r]) {
          const float gr = (blk == 0 ? acc0[r] : acc1[r]) * coef;
          const float m2 = mv[r] + (gr - mv[r]) * (1.0f - FX_BETA1);
          const float v2 = vv[r] * FX_BETA2 + (1.0f - FX_BETA2) * gr * gr;
          const float denom = sqrtf(v2) / bc2s + FX_ADAM_EPS;
          g.C[off[r]] = pv[r] - step_size * (m2 / denom);
          g.adam_m[off[r]] = m2;
          g.adam_v[off[r]] = v2;
        }
      }
    }
  }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <bool A_KC, bool B_KC>
static int launch_gemm(GemmArgs& g, int epi, hipStream_t stream) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const long nblk = (long)tiles_m * tiles_n * g.splitk;
  FX_REQUIRE(nblk > 0 && nblk < (1L << 31), "fx_gemm: bad grid (%ld blocks)", nblk);
  if (epi == EPI_STORE)
    hipLaunchKernelGGL((fx_gemm_f32_kernel<A_KC, B_KC, EPI_STORE>), dim3((unsigned)nblk), dim3(256), 0, stream, g);
  else
    hipLaunchKernelGGL((fx_gemm_f32_kernel<A_KC, B_KC, EPI_ADAM>), dim3((unsigned)nblk), dim3(256), 0, stream, g);
  return fx_check_launch("fx_gemm");
}

static int gemm_dispatch(int a_kc, int b_kc, GemmArgs& g, int epi, hipStream_t stream) {
  if (a_kc && b_kc) return launch_gemm<true, true>(g, epi, stream);
  if (a_kc && !b_kc) return launch_gemm<true, false>(g, epi, stream);
  if (!a_kc && !b_kc) return launch_gemm<false, false>(g, epi, stream);
  return launch_gemm<false, true>(g, epi, stream);
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
  g.vecA = aligned16(A) && (lda % 4 == 0);
  g.vecB = aligned16(B) && (ldb % 4 == 0);
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
  const long total = (long)M * N;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(fx_reduce_slabs_kernel, dim3(blocks), dim3(256), 0, stream, C, (const float*)workspace, bias, M, N,
                     ldc, s, g.slab_stride, accumulate);
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
  g.vecA = aligned16(dY) && (lddy % 4 == 0);
  g.vecB = aligned16(X) && (ldx % 4 == 0);
  g.splitk = 1;
  g.kchunk = (batch + BK - 1) / BK * BK;
  g.adam_m = adam_m; g.adam_v = adam_v; g.ctrl = ctrl;
  return gemm_dispatch(0, 0, g, EPI_ADAM, stream);
}

}  // extern "C"
