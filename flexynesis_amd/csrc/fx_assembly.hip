// fx_assembly.hip -- the batch-only part of a step for ALL modalities in a few launches.
//
// While step t runs its latency-bound head / backward chain, the batch of step t + 1 is assembled on a side stream
// (PipelinedStep): row gather + bf16 splits per modality (fx_gather_split) and X X^T [B, B], the batch-only factor of the
// Gram identity |dY^T X|_F^2 = <X X^T, dY dY^T> that gives the wide layers' gradient norm without forming dW
// (clip_grad_norm_(1.0), reference main.py:216).  Per modality that was gather_split -> exact-fp32 split-K GEMM ->
// slab reduce: 3 n launches, 94 us at cfg2 -- longer than the narrow chain of the main stream it hides under, so its tail
// ran beside (and slowed) the encoder-tail backward.  Here:
//
//   fx_gram_kb_group     X X^T partial sums of every modality in one launch, on the bf16 MFMA from the K-blocked split
//                        (hi, lo) that fx_gather_split has just written: three products per term, fp32 accumulation
//                        (16 mantissa bits per operand: relative error ~2^-16 of a norm that is only compared with 1).
//                        grid (K slices, modalities); a workgroup owns one K slice and the whole [128 x 128] output;
//                        the operand K-steps arrive by LDS-DMA; X is both MFMA operands.
//   fx_reduce_group      ordered sums of the partial-sum slabs of up to 4 jobs in one launch (deterministic).
//
// fx_gather_split_group lives with its single-modality kernel in fx_fused_small.hip.
#include "fx_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GK_T 512
#define GK_MAX_GROUP 4
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

struct GramJob {
  const __bf16* hi; const __bf16* lo;      // K-blocked [kblocks][128][32]
  float* slabs;                            // [n_slices][R][R]
  int kblocks, R;
};
struct GramArgs {
  GramJob j[GK_MAX_GROUP];
  int ksteps;                              // K-steps (of 32 columns) per slice
};

__device__ __forceinline__ int gk_swz(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3); }

__device__ __forceinline__ void gk_body(const GramJob& j, int ksteps, int z) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * 8192];          // stage s: hi [128][32] | lo   (2 x 16 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k0 = z * ksteps, k1 = min(j.kblocks, k0 + ksteps);
  const long bytes = (long)j.kblocks * 128 * 32 * 2;
  const unsigned nrec = bytes > 0xFFFFFFF0L ? 0xFFFFFFF0u : (unsigned)bytes;
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(j.hi), 0, nrec, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(j.lo), 0, nrec, 0x00020000);
  // DMA source: a wave instruction fills 16 rows x 64 B in lane order; slot s of row r holds chunk s ^ ((r >> 2) & 3)
  const int dr = lane >> 2, ds = lane & 3;
  const int x_r = 16 * w + dr;
  const unsigned x_src = (unsigned)((x_r * 32 + 8 * (ds ^ ((x_r >> 2) & 3))) * 2);
  const int l31 = lane & 31, kh = lane >> 5;
  const int bi = w & 3, bj0 = 2 * (w >> 2);                               // output blocks (bi, bj0), (bi, bj0 + 1) of 32 x 32
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
#define GK_DMA(stage, kb)                                                                                     \
  {                                                                                                           \
    const unsigned so = (unsigned)(kb) * (128u * 32u * 2u) + x_src;                                           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, LDS_PTR(smem + (stage) * 8192 + w * 512), 16, so, 0, 0, 0);  \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, LDS_PTR(smem + (stage) * 8192 + 4096 + w * 512), 16, so, 0, 0, 0); \
  }
  if (k0 < k1) GK_DMA(0, k0);
  for (int kb = k0; kb < k1; ++kb) {
    __syncthreads();                       // stage (kb - k0) & 1 has landed (the barrier drains the DMA); the other one is free
    if (kb + 1 < k1) GK_DMA((kb + 1 - k0) & 1, kb + 1);
    const __bf16* sb = smem + ((kb - k0) & 1) * 8192;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(sb + gk_swz(32 * bi + l31, 2 * ks + kh));
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(sb + 4096 + gk_swz(32 * bi + l31, 2 * ks + kh));
      const bf16x8 b0h = *reinterpret_cast<const bf16x8*>(sb + gk_swz(32 * bj0 + l31, 2 * ks + kh));
      const bf16x8 b0l = *reinterpret_cast<const bf16x8*>(sb + 4096 + gk_swz(32 * bj0 + l31, 2 * ks + kh));
      const bf16x8 b1h = *reinterpret_cast<const bf16x8*>(sb + gk_swz(32 * bj0 + 32 + l31, 2 * ks + kh));
      const bf16x8 b1l = *reinterpret_cast<const bf16x8*>(sb + 4096 + gk_swz(32 * bj0 + 32 + l31, 2 * ks + kh));
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0h, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0l, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0h, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b1h, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1l, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1h, acc1, 0, 0, 0);
    }
  }
#undef GK_DMA
  // C/D layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* out = j.slabs + (long)z * j.R * j.R;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * kh;
    if (row < j.R) {
      const int c0 = 32 * bj0 + l31, c1 = c0 + 32;
      if (c0 < j.R) out[(long)row * j.R + c0] = acc0[r];
      if (c1 < j.R) out[(long)row * j.R + c1] = acc1[r];
    }
  }
}

__global__ __launch_bounds__(GK_T) void fx_gram_kb_group_kernel(GramArgs a) {
  const int z = blockIdx.x;
  // (a runtime index into the kernel-argument array would put the jobs in scratch: one call per constant index)
  switch (blockIdx.y) {
    case 0: if (z * a.ksteps < a.j[0].kblocks) gk_body(a.j[0], a.ksteps, z); break;
    case 1: if (z * a.ksteps < a.j[1].kblocks) gk_body(a.j[1], a.ksteps, z); break;
    case 2: if (z * a.ksteps < a.j[2].kblocks) gk_body(a.j[2], a.ksteps, z); break;
    default: if (z * a.ksteps < a.j[3].kblocks) gk_body(a.j[3], a.ksteps, z); break;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
struct ReduceJob {
  float* y; const float* slabs; const float* bias; long n; int n_slabs, bias_n;      // y[i] = sum_z slabs[z][i] (+ bias[i % bias_n])
};
struct ReduceArgs {
  ReduceJob j[GK_MAX_GROUP];
};

__device__ __forceinline__ void rg_body(const ReduceJob& j) {
  const long n4 = j.n >> 2;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
    const float* src = j.slabs + (q << 2);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 8 <= j.n_slabs; z += 8) {           // eight slab loads in flight, added in slab order
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + (long)(z + k) * j.n);
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + (long)min(z + k, j.n_slabs - 1) * j.n);   // clamped: loads stay unconditional
#pragma unroll
      for (int k = 0; k < 8; ++k) s += (z + k < j.n_slabs) ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (j.bias) {
      const long i = (q << 2) % j.bias_n;
      s += *reinterpret_cast<const f32x4*>(j.bias + i);
    }
    *reinterpret_cast<f32x4*>(j.y + (q << 2)) = s;
  }
}

__global__ __launch_bounds__(256) void fx_reduce_group_kernel(ReduceArgs a) {
  switch (blockIdx.y) {
    case 0: rg_body(a.j[0]); break;
    case 1: rg_body(a.j[1]); break;
    case 2: rg_body(a.j[2]); break;
    default: rg_body(a.j[3]); break;
  }
}

extern "C" {

// K slices fx_gram_kb_group uses for a cohort layer of k_in features (= partial-sum slabs per modality).
int fx_gram_kb_slices(int k_in) {
  const int kblocks = (k_in + 31) / 32;
  int ks = 16;                                   // K-steps per workgroup: 40 workgroups per modality at 20000 features
  int n = (kblocks + ks - 1) / ks;
  return n < 1 ? 1 : n;
}

// For every modality i < n: slabs[i][z] [R, R] = X_i[:, K slice z] X_i[:, K slice z]^T from the K-blocked split (hi, lo) of
// X_i [R <= 128 rows, k_in[i] columns] (fx_gather_split / fx_split_bf16 layout, rows padded to 128), z < fx_gram_kb_slices(k_in[i]).
int fx_gram_kb_group(const void* const* hi, const void* const* lo, float* const* slabs, const int* k_in, int n, int R,
                     hipStream_t stream) {
  FX_REQUIRE(hi && lo && slabs && k_in && n > 0 && n <= GK_MAX_GROUP, "fx_gram_kb_group: 1..%d modalities per launch", GK_MAX_GROUP);
  FX_REQUIRE(R > 0 && R <= 128, "fx_gram_kb_group: R=%d must be in 1..128 (one M-tile of the K-blocked split)", R);
  GramArgs a{};
  a.ksteps = 16;
  int max_slices = 0;
  for (int i = 0; i < n; ++i) {
    FX_REQUIRE(hi[i] && lo[i] && slabs[i] && k_in[i] > 0 && ((((uintptr_t)hi[i]) | ((uintptr_t)lo[i])) & 15) == 0,
               "fx_gram_kb_group: modality %d: null / misaligned operand", i);
    a.j[i].hi = (const __bf16*)hi[i]; a.j[i].lo = (const __bf16*)lo[i]; a.j[i].slabs = slabs[i];
    a.j[i].kblocks = (k_in[i] + 31) / 32; a.j[i].R = R;
    const int ns = fx_gram_kb_slices(k_in[i]);
    max_slices = ns > max_slices ? ns : max_slices;
  }
  for (int i = n; i < GK_MAX_GROUP; ++i) a.j[i] = a.j[0];
  hipLaunchKernelGGL(fx_gram_kb_group_kernel, dim3(max_slices, n), dim3(GK_T), 0, stream, a);
  return fx_check_launch("fx_gram_kb_group");
}

// y[i][k] = sum_z slabs[i][z][k] (+ bias[i][k % bias_n[i]]), k < len[i], slabs in order (deterministic), for up to 4 jobs in
// one launch.  len % 4 == 0, bias_n % 4 == 0, every array 16-byte aligned.
int fx_reduce_group(float* const* y, const float* const* slabs, const float* const* bias, const long* len, const int* n_slabs,
                    const int* bias_n, int n, hipStream_t stream) {
  FX_REQUIRE(y && slabs && len && n_slabs && n > 0 && n <= GK_MAX_GROUP, "fx_reduce_group: 1..%d jobs per launch", GK_MAX_GROUP);
  ReduceArgs a{};
  long max_n = 0;
  for (int i = 0; i < n; ++i) {
    const float* b = bias ? bias[i] : nullptr;
    FX_REQUIRE(y[i] && slabs[i] && len[i] > 0 && len[i] % 4 == 0 && n_slabs[i] > 0 &&
                   ((((uintptr_t)y[i]) | ((uintptr_t)slabs[i]) | ((uintptr_t)b)) & 15) == 0,
               "fx_reduce_group: job %d: length %ld must be a multiple of 4 and the arrays 16-byte aligned", i, len[i]);
    FX_REQUIRE(!b || (bias_n && bias_n[i] > 0 && bias_n[i] % 4 == 0 && len[i] % bias_n[i] == 0), "fx_reduce_group: job %d: bad bias period", i);
    a.j[i].y = y[i]; a.j[i].slabs = slabs[i]; a.j[i].bias = b; a.j[i].n = len[i]; a.j[i].n_slabs = n_slabs[i];
    a.j[i].bias_n = b ? bias_n[i] : 4;
    max_n = len[i] > max_n ? len[i] : max_n;
  }
  for (int i = n; i < GK_MAX_GROUP; ++i) a.j[i] = a.j[0];
  const long blocks = ((max_n >> 2) + 255) / 256;
  hipLaunchKernelGGL(fx_reduce_group_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks), n), dim3(256), 0, stream, a);
  return fx_check_launch("fx_reduce_group");
}

}  // extern "C"
