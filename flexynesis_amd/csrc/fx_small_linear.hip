// Small dense layers on the critical chain between the wide forward and the dW+Adam launches: the fusion layer
// (reference direct_pred.py:87-93,121-124: Linear(n * latent -> latent) on the concatenated embeddings) and the VAE's
// FC_mean / FC_log_var (supervised_vae.py:104-107,172-176).  Shapes are tiny (rows <= 3 * 128, <= 128 outputs, <= 3 * 128
// inputs), so the tiled MFMA GEMM with its split-K machinery spent 10-15 us per call on fixed latency, and the backward
// was three dependent launches (weight gradient, bias gradient, data gradient).  Here: one forward kernel and ONE backward
// kernel for all three gradients, plain fp32 FMA through LDS tiles, a fixed summation order (deterministic).
#include "fx_common.h"

#define SL_T 256
#define SL_TR 16          // rows (or outputs, for the weight gradient) per tile
#define SL_TC 64          // columns per tile: one per lane of a wavefront
#define SL_KC 64          // contraction chunk staged in LDS

struct SmallLinArgs {
  // forward: y[r, o] = b[o] + sum_k x[r, k] W[o, k]
  // backward: dx[r, k] = sum_o dy[r, o] W[o, k] ; gW[o, k] = sum_r dy[r, o] x[r, k] ; gb[o] = sum_r dy[r, o]
  float* y; const float* x; const float* W; const float* b;
  float* dx; float* gW; float* gb; const float* dy;
  int R, O, K;
  long ldx, ldy, lddx, lddy;       // leading dimensions of x, y, dx, dy (W / gW are contiguous [O, K])
  int tiles_dx;                    // backward: workgroups [0, tiles_dx) compute dx tiles, the rest gW (+ gb) tiles
  int dx_accumulate;
  const float* dy_mul; long ldmul; // backward: the upstream gradient is dy[r, o] * dy_mul[r, o] (NULL: dy as it is)
  int grid;                        // grouped launch: workgroups of this job
};

// stage a [rows x cols] block of src (row stride ld) into dst[rows][SL_KC + 1]; out-of-range elements become 0.  Addresses
// are clamped and the loads unconditional (a predicated load costs a dependent round trip each, fx_heads.hip)
template <int ROWS>
__device__ __forceinline__ void sl_stage_load(float (&v)[ROWS * SL_KC / SL_T], const float* __restrict__ src, long ld, int r0, int nr,
                                              int c0, int nc) {
#pragma unroll
  for (int i = 0; i < ROWS * SL_KC / SL_T; ++i) {
    const int idx = threadIdx.x + SL_T * i, rr = idx / SL_KC, cc = idx % SL_KC;
    v[i] = src[(long)min(r0 + rr, nr - 1) * ld + min(c0 + cc, nc - 1)];
  }
}
template <int ROWS>
__device__ __forceinline__ void sl_stage_store(float (*dst)[SL_KC + 1], const float (&v)[ROWS * SL_KC / SL_T], int r0, int nr, int c0, int nc) {
#pragma unroll
  for (int i = 0; i < ROWS * SL_KC / SL_T; ++i) {
    const int idx = threadIdx.x + SL_T * i, rr = idx / SL_KC, cc = idx % SL_KC;
    dst[rr][cc] = (r0 + rr < nr && c0 + cc < nc) ? v[i] : 0.f;
  }
}
template <int ROWS>
__device__ __forceinline__ void sl_stage(float (*dst)[SL_KC + 1], const float* __restrict__ src, long ld, int r0, int nr,
                                         int c0, int nc) {
  float v[ROWS * SL_KC / SL_T];
  sl_stage_load<ROWS>(v, src, ld, r0, nr, c0, nc);
  sl_stage_store<ROWS>(dst, v, r0, nr, c0, nc);
}

__global__ __launch_bounds__(SL_T) void fx_small_linear_fwd_kernel(SmallLinArgs a) {
  __shared__ float xs[SL_TR][SL_KC + 1];
  __shared__ float ws[SL_TC][SL_KC + 1];
  const int tiles_o = (a.O + SL_TC - 1) / SL_TC;
  const int r0 = (blockIdx.x / tiles_o) * SL_TR, o0 = (blockIdx.x % tiles_o) * SL_TC;
  const int ol = threadIdx.x & 63, rg = threadIdx.x >> 6;                 // thread: output o0 + ol, rows rg, rg + 4, ...
  float acc[SL_TR / 4];
  const float bias = (a.b && o0 + ol < a.O) ? a.b[o0 + ol] : 0.f;
#pragma unroll
  for (int i = 0; i < SL_TR / 4; ++i) acc[i] = bias;
  float vx[SL_TR * SL_KC / SL_T], vw[SL_TC * SL_KC / SL_T];           // (the next K chunk's loads in flight while this one is multiplied)
  sl_stage_load<SL_TR>(vx, a.x, a.ldx, r0, a.R, 0, a.K);
  sl_stage_load<SL_TC>(vw, a.W, a.K, o0, a.O, 0, a.K);
  for (int k0 = 0; k0 < a.K; k0 += SL_KC) {
    __syncthreads();
    sl_stage_store<SL_TR>(xs, vx, r0, a.R, k0, a.K);
    sl_stage_store<SL_TC>(ws, vw, o0, a.O, k0, a.K);
    if (k0 + SL_KC < a.K) {
      sl_stage_load<SL_TR>(vx, a.x, a.ldx, r0, a.R, k0 + SL_KC, a.K);
      sl_stage_load<SL_TC>(vw, a.W, a.K, o0, a.O, k0 + SL_KC, a.K);
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < SL_KC; ++k) {
      const float wv = ws[ol][k];
#pragma unroll
      for (int i = 0; i < SL_TR / 4; ++i) acc[i] = fmaf(xs[rg + 4 * i][k], wv, acc[i]);
    }
  }
  if (o0 + ol < a.O)
#pragma unroll
    for (int i = 0; i < SL_TR / 4; ++i)
      if (r0 + rg + 4 * i < a.R) a.y[(long)(r0 + rg + 4 * i) * a.ldy + o0 + ol] = acc[i];
}

// upstream gradient element (clamped address, unconditional loads)
__device__ __forceinline__ float sl_dy(const SmallLinArgs& a, int r, int o) {
  const int rc = min(r, a.R - 1), oc = min(o, a.O - 1);
  const float v = a.dy[(long)rc * a.lddy + oc];
  return a.dy_mul ? v * a.dy_mul[(long)rc * a.ldmul + oc] : v;
}

__device__ __forceinline__ void sl_bwd_body(const SmallLinArgs& a, const int bid, float (*s1)[SL_KC + 1], float (*s2)[SL_TR + 1]) {
  const int tiles_k = (a.K + SL_TC - 1) / SL_TC;
  const int kl = threadIdx.x & 63, g4 = threadIdx.x >> 6;
  float acc[SL_TR / 4];
#pragma unroll
  for (int i = 0; i < SL_TR / 4; ++i) acc[i] = 0.f;
  if (bid < a.tiles_dx) {
    // ---- dx tile: rows r0.., columns k0..: sum over outputs o
    const int r0 = (bid / tiles_k) * SL_TR, k0 = (bid % tiles_k) * SL_TC;
    for (int o0 = 0; o0 < a.O; o0 += SL_KC) {
      __syncthreads();
      sl_stage<SL_KC>(s1, a.W, a.K, o0, a.O, k0, a.K);                 // s1[o][k]
      {                                                                 // s2[o][r] = dy[r0 + r, o0 + o]
        constexpr int PER = SL_TR * SL_KC / SL_T;
        float v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int idx = threadIdx.x + SL_T * i, rr = idx / SL_KC, oo = idx % SL_KC;
          v[i] = sl_dy(a, r0 + rr, o0 + oo);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int idx = threadIdx.x + SL_T * i, rr = idx / SL_KC, oo = idx % SL_KC;
          s2[oo][rr] = (r0 + rr < a.R && o0 + oo < a.O) ? v[i] : 0.f;
        }
      }
      __syncthreads();
#pragma unroll 8
      for (int o = 0; o < SL_KC; ++o) {
        const float wv = s1[o][kl];
#pragma unroll
        for (int i = 0; i < SL_TR / 4; ++i) acc[i] = fmaf(s2[o][g4 + 4 * i], wv, acc[i]);
      }
    }
    if (k0 + kl < a.K)
#pragma unroll
      for (int i = 0; i < SL_TR / 4; ++i) {
        const int r = r0 + g4 + 4 * i;
        if (r < a.R) {
          float* d = a.dx + (long)r * a.lddx + k0 + kl;
          *d = a.dx_accumulate ? *d + acc[i] : acc[i];
        }
      }
    return;
  }
  // ---- gW tile: outputs o0.., columns k0..: sum over rows r; the k-tile 0 workgroups also produce gb
  const int t = bid - a.tiles_dx;
  const int o0 = (t / tiles_k) * SL_TR, k0 = (t % tiles_k) * SL_TC;
  float accb = 0.f;
  // The row chunks are a dependent chain of (load -> LDS -> multiply) in a handful of workgroups that the whole launch waits for (the
  // stacked-rows plans have six chunks: 56 us at R = 384): the NEXT chunk's loads are in flight while this one is multiplied (round 6).
  // Same chunks in the same order: the same sums.
  constexpr int PERX = SL_KC * SL_KC / SL_T, PERD = SL_KC * SL_TR / SL_T;
  float vx[PERX], vd[PERD];
#define SL_GW_REQ(rr0)                                                                       \
  {                                                                                          \
    sl_stage_load<SL_KC>(vx, a.x, a.ldx, (rr0), a.R, k0, a.K);                               \
    _Pragma("unroll") for (int i = 0; i < PERD; ++i) {                                       \
      const int idx = threadIdx.x + SL_T * i, rr = idx / SL_TR, oo = idx % SL_TR;            \
      vd[i] = sl_dy(a, (rr0) + rr, o0 + oo);                                                 \
    }                                                                                        \
  }
  SL_GW_REQ(0)
  for (int r0 = 0; r0 < a.R; r0 += SL_KC) {
    __syncthreads();
    sl_stage_store<SL_KC>(s1, vx, r0, a.R, k0, a.K);                     // s1[r][k]
#pragma unroll
    for (int i = 0; i < PERD; ++i) {                                      // s2[r][o] = dy[r0 + r, o0 + o]
      const int idx = threadIdx.x + SL_T * i, rr = idx / SL_TR, oo = idx % SL_TR;
      s2[rr][oo] = (r0 + rr < a.R && o0 + oo < a.O) ? vd[i] : 0.f;
    }
    if (r0 + SL_KC < a.R) SL_GW_REQ(r0 + SL_KC)
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < SL_KC; ++r) {
      const float xv = s1[r][kl];
#pragma unroll
      for (int i = 0; i < SL_TR / 4; ++i) acc[i] = fmaf(s2[r][g4 + 4 * i], xv, acc[i]);
    }
    if (k0 == 0 && a.gb && threadIdx.x < SL_TR)
      for (int r = 0; r < SL_KC; ++r) accb += s2[r][threadIdx.x];
  }
  if (k0 + kl < a.K)
#pragma unroll
    for (int i = 0; i < SL_TR / 4; ++i) {
      const int o = o0 + g4 + 4 * i;
      if (o < a.O) a.gW[(long)o * a.K + k0 + kl] = acc[i];
    }
  if (k0 == 0 && a.gb && threadIdx.x < SL_TR && o0 + (int)threadIdx.x < a.O) a.gb[o0 + threadIdx.x] = accb;
}

__global__ __launch_bounds__(SL_T) void fx_small_linear_bwd_kernel(SmallLinArgs a) {
  __shared__ float s1[SL_KC][SL_KC + 1];      // dx: W chunk [o][k]        gW: x chunk [r][k]
  __shared__ float s2[SL_KC][SL_TR + 1];      // dx: dy tile, transposed [o][r]   gW: dy chunk [r][o]
  sl_bwd_body(a, blockIdx.x, s1, s2);
}

// several layers' backward in one launch (blockIdx.y = layer): the VAE's FC_mean and FC_log_var
#define SL_MAX_GROUP 4
struct SmallLinGroup { SmallLinArgs job[SL_MAX_GROUP]; };
__global__ __launch_bounds__(SL_T) void fx_small_linear_bwd_group_kernel(SmallLinGroup g) {
  __shared__ float s1[SL_KC][SL_KC + 1];
  __shared__ float s2[SL_KC][SL_TR + 1];
  const SmallLinArgs& a = g.job[blockIdx.y];
  if ((int)blockIdx.x >= a.grid) return;
  sl_bwd_body(a, blockIdx.x, s1, s2);
}

extern "C" {

// y[R, O] = x[R, K] W[O, K]^T + b   (W contiguous; b may be NULL).  For the small layers only: R, O, K <= 4096.
int fx_small_linear_fwd(float* y, const float* x, const float* W, const float* b, int R, int O, int K, long ldx, long ldy,
                        hipStream_t stream) {
  FX_REQUIRE(y && x && W && R > 0 && O > 0 && K > 0 && ldx >= K && ldy >= O, "fx_small_linear_fwd: bad args");
  FX_REQUIRE(R <= 4096 && O <= 4096 && K <= 4096, "fx_small_linear_fwd: meant for small layers (got %d x %d x %d)", R, O, K);
  SmallLinArgs a{};
  a.y = y; a.x = x; a.W = W; a.b = b; a.R = R; a.O = O; a.K = K; a.ldx = ldx; a.ldy = ldy;
  const int grid = ((R + SL_TR - 1) / SL_TR) * ((O + SL_TC - 1) / SL_TC);
  hipLaunchKernelGGL(fx_small_linear_fwd_kernel, dim3(grid), dim3(SL_T), 0, stream, a);
  return fx_check_launch("fx_small_linear_fwd");
}

// All three gradients of y = x W^T + b in one launch: dx[R, K] (+)= dy W (NULL: not needed), gW[O, K] = dy^T x,
// gb[O] = column sums of dy (NULL: the layer has no bias).
int fx_small_linear_bwd(float* dx, float* gW, float* gb, const float* dy, const float* x, const float* W, int R, int O, int K,
                        long ldx, long lddy, long lddx, int dx_accumulate, hipStream_t stream) {
  FX_REQUIRE(gW && dy && x && W && R > 0 && O > 0 && K > 0 && ldx >= K && lddy >= O && (!dx || lddx >= K),
             "fx_small_linear_bwd: bad args");
  FX_REQUIRE(R <= 4096 && O <= 4096 && K <= 4096, "fx_small_linear_bwd: meant for small layers (got %d x %d x %d)", R, O, K);
  SmallLinArgs a{};
  a.dx = dx; a.gW = gW; a.gb = gb; a.dy = dy; a.x = x; a.W = W; a.R = R; a.O = O; a.K = K;
  a.ldx = ldx; a.lddy = lddy; a.lddx = lddx; a.dx_accumulate = dx_accumulate;
  const int tiles_k = (K + SL_TC - 1) / SL_TC;
  a.tiles_dx = dx ? ((R + SL_TR - 1) / SL_TR) * tiles_k : 0;
  const int grid = a.tiles_dx + ((O + SL_TR - 1) / SL_TR) * tiles_k;
  hipLaunchKernelGGL(fx_small_linear_bwd_kernel, dim3(grid), dim3(SL_T), 0, stream, a);
  return fx_check_launch("fx_small_linear_bwd");
}

// The same for up to 4 layers in ONE launch.  jobs: fx_small_linear_job[n] (include/fxhip.h; host memory, read during the call);
// a job's upstream gradient is dy[r, o] * dy_mul[r, o] when dy_mul is given (the reparameterisation's d log_var = dz * eps,
// reference supervised_vae.py:190-200, without a launch of its own).
struct SmallLinJob {
  float* dx; float* gW; float* gb; const float* dy; const float* dy_mul; const float* x; const float* W;
  int R, O, K, dx_accumulate;
  long ldx, lddy, ldmul, lddx;
};
int fx_small_linear_bwd_group(const void* jobs_, int n, hipStream_t stream) {
  const SmallLinJob* jobs = (const SmallLinJob*)jobs_;
  FX_REQUIRE(jobs && n > 0 && n <= SL_MAX_GROUP, "fx_small_linear_bwd_group: 1..%d jobs", SL_MAX_GROUP);
  SmallLinGroup g{};
  int grid = 0;
  for (int i = 0; i < n; ++i) {
    const SmallLinJob& j = jobs[i];
    FX_REQUIRE(j.gW && j.dy && j.x && j.W && j.R > 0 && j.O > 0 && j.K > 0 && j.ldx >= j.K && j.lddy >= j.O && (!j.dx || j.lddx >= j.K) &&
               (!j.dy_mul || j.ldmul >= j.O), "fx_small_linear_bwd_group: bad args in job %d", i);
    FX_REQUIRE(j.R <= 4096 && j.O <= 4096 && j.K <= 4096, "fx_small_linear_bwd_group: meant for small layers (got %d x %d x %d)", j.R, j.O, j.K);
    SmallLinArgs& a = g.job[i];
    a.dx = j.dx; a.gW = j.gW; a.gb = j.gb; a.dy = j.dy; a.dy_mul = j.dy_mul; a.x = j.x; a.W = j.W; a.R = j.R; a.O = j.O; a.K = j.K;
    a.ldx = j.ldx; a.lddy = j.lddy; a.ldmul = j.ldmul; a.lddx = j.lddx; a.dx_accumulate = j.dx_accumulate;
    const int tiles_k = (j.K + SL_TC - 1) / SL_TC;
    a.tiles_dx = j.dx ? ((j.R + SL_TR - 1) / SL_TR) * tiles_k : 0;
    a.grid = a.tiles_dx + ((j.O + SL_TR - 1) / SL_TR) * tiles_k;
    if (a.grid > grid) grid = a.grid;
  }
  hipLaunchKernelGGL(fx_small_linear_bwd_group_kernel, dim3(grid, n), dim3(SL_T), 0, stream, g);
  return fx_check_launch("fx_small_linear_bwd_group");
}

}  // extern "C"
