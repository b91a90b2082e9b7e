// Device-side ingest of a raw omics matrix.
//
// Input is the layout the reference's HDF5 importer hands over: a contiguous [n_samples, n_features] matrix,
// samples as rows (reference h5_dataloader.py:88-116, csv_to_h5.py:13-21), fp32 (HDF5 path) or fp64 (what
// pd.read_csv yields, reference data.py:313-330).  The kernels are the per-matrix stages of
// DataImporter.cleanup_data / transform_data / normalize_data (reference data.py:360-452, 519-545):
//   fx_col_moments       per-feature count / mean / sum of squared deviations   (df.var(axis=1), StandardScaler.fit)
//   fx_col_median        per-feature median of the non-NaN entries               (df_T.median(axis=0))
//   fx_row_moments       per-sample variance over the kept, imputed features     (df.std(axis=0))
//   fx_ingest_transform  gather kept rows/columns, impute, log1p, standardise -> fp32 [rows, cols]
// All of them are HBM-bound streaming passes: one thread per feature column so that a wavefront reads 256 B of a
// sample row at a time, rows split over blockIdx.y for occupancy, fp64 accumulation (the reference path is
// pandas/sklearn fp64), fixed-order merges so that results are bit-reproducible.
#include "fx_common.h"

#define ING_T 256        // threads per workgroup = columns per workgroup

namespace {

template <typename T>
struct InMode;
template <>
struct InMode<float> { static constexpr bool f32 = true; };
template <>
struct InMode<double> { static constexpr bool f32 = false; };

// The value the reference's pipeline holds after NaN imputation and the optional log1p.  In fp32 mode the
// DataFrame is float32 throughout (imputed median and np.log1p results are float32), in fp64 mode float64.
template <typename T, bool LOG>
__device__ __forceinline__ double ingest_value(T raw, double medv) {
  // medv = the column's imputation value, NaN when there is none (a NaN entry then stays NaN): branch-free select
  if (InMode<T>::f32) {
    float v = (float)raw;
    v = (v != v) ? (float)medv : v;
    if (LOG) v = (float)log1p((double)v);
    return (double)v;
  } else {
    double v = (double)raw;
    v = (v != v) ? medv : v;
    if (LOG) v = log1p(v);
    return v;
  }
}
__device__ __forceinline__ double ingest_med(const double* __restrict__ med, int c) {
  return med ? med[c] : __builtin_nan("");
}

// ---- per-column moments ----------------------------------------------------------------------------------------
// Shifted one-pass sums: d = value - K with K the column's first non-NaN value among the first rows (a tiny
// pre-pass), so that q - s*s/n does not cancel (K is within a few standard deviations of the mean).  NaNs are
// skipped (pandas skipna, sklearn nansum).  Each (column block, row chunk) workgroup writes its partial n/s/q; the
// merge adds the chunks in index order, so the result is bit-reproducible.
// Two adjacent columns per thread are read with one 8/16-byte load when F, ldx and the base address allow it.
#define ING_PIVOT_ROWS 8
template <typename T, bool LOG>
__global__ __launch_bounds__(ING_T) void fx_col_pivot_kernel(const T* __restrict__ x, long ldx, int n_rows, int F,
                                                             const int* __restrict__ rows,
                                                             const double* __restrict__ med,
                                                             double* __restrict__ pivot) {
  const int c = blockIdx.x * ING_T + threadIdx.x;
  if (c >= F) return;
  const double mv = ingest_med(med, c);
  const int np = n_rows < ING_PIVOT_ROWS ? n_rows : ING_PIVOT_ROWS;
  double K = 0.0;
  bool have = false;
  for (int i = 0; i < np; ++i) {
    const long r = rows ? rows[i] : i;
    const double v = ingest_value<T, LOG>(x[r * ldx + c], mv);
    if (!have && v == v) { K = v; have = true; }
  }
  pivot[c] = K;
}

template <typename T, int VEC>
struct InVec { T v[VEC]; };

template <typename T, int VEC, bool LOG>
__global__ __launch_bounds__(ING_T) void fx_col_moments_kernel(const T* __restrict__ x, long ldx, int n_rows, int F,
                                                               const int* __restrict__ rows,
                                                               const double* __restrict__ med,
                                                               int rows_per_chunk, const double* __restrict__ pivot,
                                                               int* __restrict__ pn, double* __restrict__ ps,
                                                               double* __restrict__ pq) {
  typedef InVec<T, VEC> __attribute__((aligned(sizeof(T) * VEC))) VT;
  const int c = (blockIdx.x * ING_T + threadIdx.x) * VEC;
  if (c >= F) return;
  double K[VEC], mv[VEC], s[VEC], q[VEC];
  int n[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    mv[e] = ingest_med(med, c + e);
    K[e] = pivot[c + e];
    s[e] = 0.0; q[e] = 0.0; n[e] = 0;
  }
  const int r0 = blockIdx.y * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  if (r1 > n_rows) r1 = n_rows;
  constexpr int UN = 8;
  int i = r0;
  for (; i + UN <= r1; i += UN) {
    VT raw[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long r = rows ? rows[i + u] : (i + u);
      raw[u] = *(const VT*)(x + r * ldx + c);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const double v = ingest_value<T, LOG>(raw[u].v[e], mv[e]);
        const bool ok = (v == v);
        const double d = ok ? v - K[e] : 0.0;
        n[e] += ok ? 1 : 0;
        s[e] += d;
        q[e] += d * d;
      }
    }
  }
  for (; i < r1; ++i) {
    const long r = rows ? rows[i] : i;
    const VT raw = *(const VT*)(x + r * ldx + c);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const double v = ingest_value<T, LOG>(raw.v[e], mv[e]);
      const bool ok = (v == v);
      const double d = ok ? v - K[e] : 0.0;
      n[e] += ok ? 1 : 0;
      s[e] += d;
      q[e] += d * d;
    }
  }
  const long o = (long)blockIdx.y * F + c;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    pn[o + e] = n[e];
    ps[o + e] = s[e];
    pq[o + e] = q[e];
  }
}

__global__ __launch_bounds__(ING_T) void fx_col_moments_merge_kernel(const int* __restrict__ pn,
                                                                     const double* __restrict__ ps,
                                                                     const double* __restrict__ pq,
                                                                     const double* __restrict__ pivot, int chunks, int F,
                                                                     int* __restrict__ count, double* __restrict__ mean,
                                                                     double* __restrict__ m2) {
  const int c = blockIdx.x * ING_T + threadIdx.x;
  if (c >= F) return;
  int n = 0;
  double s = 0.0, q = 0.0;
#pragma unroll 8
  for (int k = 0; k < chunks; ++k) {
    const long o = (long)k * F + c;
    n += pn[o];
    s += ps[o];
    q += pq[o];
  }
  count[c] = n;
  if (n == 0) {
    mean[c] = __builtin_nan("");
    m2[c] = __builtin_nan("");
  } else {
    const double sm = s / (double)n;
    const double r = q - s * sm;
    mean[c] = pivot[c] + sm;
    m2[c] = r > 0.0 ? r : 0.0;
  }
}

// ---- per-column median -----------------------------------------------------------------------------------------
// One workgroup per listed column: MSB-first radix select (8 x 8 bits) on the order-preserving 64-bit image of the
// value (fp32 input is widened exactly).  Even counts average the two middle order statistics, as numpy does.
__device__ __forceinline__ unsigned long long ing_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double ing_unkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

#define ING_KEYS 16  // keys cached in registers per thread: columns of up to 4096 rows are read once
template <typename T>
__global__ __launch_bounds__(ING_T) void fx_col_median_kernel(const T* __restrict__ x, long ldx, int N,
                                                              const int* __restrict__ cols,
                                                              double* __restrict__ med_out) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned s_k;
  __shared__ int s_wave[ING_T / 64];
  const int c = cols[blockIdx.x];
  const int tid = threadIdx.x;
  // key 0 is not the image of any non-NaN value (-inf maps to 0x000F...F), so it marks "NaN / past the end"
  unsigned long long key[ING_KEYS];
  int n_loc = 0;
#pragma unroll
  for (int u = 0; u < ING_KEYS; ++u) {
    int r = tid + u * ING_T;
    const bool in = r < N;
    if (!in) r = N - 1;
    const double v = (double)x[(long)r * ldx + c];
    const bool ok = in && (v == v);
    key[u] = ok ? ing_key(v) : 0ull;
    n_loc += ok ? 1 : 0;
  }
  for (int r = tid + ING_KEYS * ING_T; r < N; r += ING_T) {
    const double v = (double)x[(long)r * ldx + c];
    n_loc += (v == v) ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) n_loc += __shfl_xor(n_loc, off, 64);
  if ((tid & 63) == 0) s_wave[tid >> 6] = n_loc;
  __syncthreads();
  int n = 0;
  for (int w = 0; w < ING_T / 64; ++w) n += s_wave[w];
  if (n == 0) {
    if (tid == 0) med_out[c] = __builtin_nan("");
    return;
  }
  double val[2];
  const unsigned rank[2] = {(unsigned)((n - 1) / 2), (unsigned)(n / 2)};
  for (int which = 0; which < 2; ++which) {
    if (which == 1 && rank[1] == rank[0]) { val[1] = val[0]; break; }
    if (tid == 0) { s_prefix = 0ull; s_k = rank[which]; }
    for (int pass = 7; pass >= 0; --pass) {
      const int shift = pass * 8;
      hist[tid] = 0u;  // ING_T == 256 bins
      __syncthreads();
      const unsigned long long prefix = s_prefix;
      const unsigned long long want = (pass == 7) ? 0ull : (prefix >> (shift + 8));
#pragma unroll
      for (int u = 0; u < ING_KEYS; ++u) {
        const unsigned long long k = key[u];
        const bool match = (k != 0ull) && ((pass == 7) || ((k >> (shift + 8)) == want));
        if (match) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
      }
      for (int r = tid + ING_KEYS * ING_T; r < N; r += ING_T) {
        const double v = (double)x[(long)r * ldx + c];
        if (v == v) {
          const unsigned long long k = ing_key(v);
          const bool match = (pass == 7) || ((k >> (shift + 8)) == want);
          if (match) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
        }
      }
      __syncthreads();
      if (tid < 64) {  // one wavefront scans the 256 bins: 4 bins per lane + a shuffle prefix sum
        const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
        const unsigned tot = h0 + h1 + h2 + h3;
        unsigned inc = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned t = __shfl_up(inc, off, 64);
          if (tid >= off) inc += t;
        }
        const unsigned exc = inc - tot, k = s_k;
        if (exc <= k && k < inc) {  // exactly one lane holds the rank
          unsigned cum = exc;
          int b = 4 * tid;
          if (cum + h0 <= k) { cum += h0; ++b;
            if (cum + h1 <= k) { cum += h1; ++b;
              if (cum + h2 <= k) { cum += h2; ++b; } } }
          s_k = k - cum;
          s_prefix = prefix | ((unsigned long long)b << shift);
        }
      }
      __syncthreads();
    }
    val[which] = ing_unkey(s_prefix);
    __syncthreads();
  }
  if (tid == 0) {
    const double m = (val[0] + val[1]) * 0.5;
    med_out[c] = InMode<T>::f32 ? (double)(float)m : m;  // a float32 frame holds a float32 median
  }
}

// ---- per-row variance over the kept, imputed columns (ddof = 1) ---------------------------------------------------
// One 1024-thread workgroup per 8 samples: a thread owns a kept column per step and reads it for the 8 rows, so the
// column index and its imputation value are fetched once per 8 elements and 8 loads are in flight per thread; the
// 24 partial sums are reduced with wavefront shuffles and one LDS exchange, in a fixed order.
#define ING_RT 1024
#define ING_RR 8
template <typename T>
__global__ __launch_bounds__(ING_RT) void fx_row_moments_kernel(const T* __restrict__ x, long ldx, int N,
                                                                const int* __restrict__ cols, int n_cols,
                                                                const double* __restrict__ med,
                                                                double* __restrict__ var_out) {
  __shared__ double red[ING_RT / 64][3 * ING_RR];
  const int r0 = blockIdx.x * ING_RR;
  const int c0 = cols[0];
  const double m0 = ingest_med(med, c0);
  const T* xr[ING_RR];
  double K[ING_RR], s[ING_RR], q[ING_RR], n[ING_RR];
#pragma unroll
  for (int u = 0; u < ING_RR; ++u) {
    int r = r0 + u;
    if (r >= N) r = N - 1;
    xr[u] = x + (long)r * ldx;
    const double k0 = ingest_value<T, false>(xr[u][c0], m0);
    K[u] = (k0 == k0) ? k0 : 0.0;
    s[u] = 0.0; q[u] = 0.0; n[u] = 0.0;
  }
  for (int j = threadIdx.x; j < n_cols; j += ING_RT) {
    const int c = cols[j];
    const double mv = ingest_med(med, c);
    T raw[ING_RR];
#pragma unroll
    for (int u = 0; u < ING_RR; ++u) raw[u] = xr[u][c];
#pragma unroll
    for (int u = 0; u < ING_RR; ++u) {
      const double v = ingest_value<T, false>(raw[u], mv);
      const bool ok = (v == v);
      const double d = ok ? v - K[u] : 0.0;
      n[u] += ok ? 1.0 : 0.0;
      s[u] += d;
      q[u] += d * d;
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int u = 0; u < ING_RR; ++u) {
    const double a = fx_wave_sum_d(n[u]), b = fx_wave_sum_d(s[u]), c = fx_wave_sum_d(q[u]);
    if (lane == 0) {
      red[wid][3 * u] = a;
      red[wid][3 * u + 1] = b;
      red[wid][3 * u + 2] = c;
    }
  }
  __syncthreads();
  if (threadIdx.x < ING_RR && r0 + (int)threadIdx.x < N) {
    const int u = threadIdx.x;
    double nt = 0.0, st = 0.0, qt = 0.0;
    for (int w = 0; w < ING_RT / 64; ++w) {
      nt += red[w][3 * u];
      st += red[w][3 * u + 1];
      qt += red[w][3 * u + 2];
    }
    if (nt < 2.0) {
      var_out[r0 + u] = __builtin_nan("");
    } else {
      const double m2 = qt - st * (st / nt);
      var_out[r0 + u] = (m2 > 0.0 ? m2 : 0.0) / (nt - 1.0);
    }
  }
}

// ---- gather + impute + log1p + standardise ------------------------------------------------------------------------
// fp64 input: the reference scales the float64 frame and casts the result once (data.py:533-540, 549).  fp32 input:
// sklearn's in-place `X -= mean_; X /= scale_` on a float32 array computes each op in fp64 and rounds to fp32.
#define ING_ROWS 16
template <typename T, bool LOG>
__global__ __launch_bounds__(ING_T) void fx_ingest_transform_kernel(const T* __restrict__ x, long ldx,
                                                                    const int* __restrict__ rows, int n_rows,
                                                                    const int* __restrict__ cols, int n_cols,
                                                                    const double* __restrict__ med,
                                                                    const double* __restrict__ mean,
                                                                    const double* __restrict__ scale,
                                                                    float* __restrict__ out, long ldo) {
  const int j = blockIdx.x * ING_T + threadIdx.x;
  if (j >= n_cols) return;
  const int c = cols ? cols[j] : j;
  const double mu = mean ? mean[j] : 0.0;
  const double sc = scale ? scale[j] : 1.0;
  const double mv = ingest_med(med, c);
  const int i0 = blockIdx.y * ING_ROWS;
  T raw[ING_ROWS];
#pragma unroll
  for (int u = 0; u < ING_ROWS; ++u) {
    int i = i0 + u;
    if (i >= n_rows) i = n_rows - 1;
    const long r = rows ? rows[i] : i;
    raw[u] = x[r * ldx + c];
  }
#pragma unroll
  for (int u = 0; u < ING_ROWS; ++u) {
    const int i = i0 + u;
    if (i < n_rows) {
      const double v = ingest_value<T, LOG>(raw[u], mv);
      float o;
      if (mean == nullptr && scale == nullptr) {
        o = (float)v;
      } else if (InMode<T>::f32) {
        const float d = (float)(v - mu);
        o = (float)((double)d / sc);
      } else {
        o = (float)((v - mu) / sc);
      }
      out[(long)i * ldo + j] = o;
    }
  }
}

inline int col_chunks(int n_rows, int F) {
  const int cb = (F + 2 * ING_T - 1) / (2 * ING_T);  // column blocks of the 2-wide path
  int want = (512 + cb - 1) / cb;                    // ~512 workgroups: 2 per CU, each streaming >= 16 rows
  const int maxc = (n_rows + 15) / 16;
  if (want > maxc) want = maxc;
  if (want > 256) want = 256;
  if (want < 1) want = 1;
  return want;
}

}  // namespace

extern "C" {

int fx_col_moments_chunks(int n_rows, int F) { return col_chunks(n_rows, F); }

long fx_col_moments_workspace_bytes(int n_rows, int F) {
  const long ch = col_chunks(n_rows, F);
  return ch * F * (long)(sizeof(int) + 2 * sizeof(double)) + (long)F * sizeof(double) + 64;
}

int fx_col_moments(const void* x, int dtype, long ldx, int n_rows, int F, const int* rows, const double* med,
                   int log1p_flag, int* count, double* mean, double* m2, void* ws, hipStream_t stream) {
  FX_REQUIRE(x && count && mean && m2 && ws, "fx_col_moments: null pointer");
  FX_REQUIRE(n_rows > 0 && F > 0 && ldx >= F, "fx_col_moments: bad shape (n_rows=%d F=%d ldx=%ld)", n_rows, F, ldx);
  FX_REQUIRE(dtype == 0 || dtype == 1, "fx_col_moments: dtype must be 0 (f32) or 1 (f64)");
  FX_REQUIRE((((uintptr_t)ws) & 7) == 0, "fx_col_moments: workspace must be 8-byte aligned");
  const int ch = col_chunks(n_rows, F);
  const int rpc = (n_rows + ch - 1) / ch;
  double* ps = (double*)ws;
  double* pq = ps + (long)ch * F;
  double* pivot = pq + (long)ch * F;
  int* pn = (int*)(pivot + F);
  {
    const dim3 g1((F + ING_T - 1) / ING_T);
#define FX_PV_LAUNCH(T, LG) \
  hipLaunchKernelGGL((fx_col_pivot_kernel<T, LG>), g1, dim3(ING_T), 0, stream, (const T*)x, ldx, n_rows, F, rows, med, pivot)
    if (dtype == 0) { if (log1p_flag) FX_PV_LAUNCH(float, true); else FX_PV_LAUNCH(float, false); }
    else { if (log1p_flag) FX_PV_LAUNCH(double, true); else FX_PV_LAUNCH(double, false); }
#undef FX_PV_LAUNCH
  }
  // two adjacent columns per thread (8/16-byte loads) when alignment allows; measured 33-40 us for 2048 x 20000
  // fp32 across 1/2/4 columns per thread, fewer and fatter workgroups being what matters
  const bool can2 = (F % 2 == 0) && (ldx % 2 == 0) && ((((uintptr_t)x) & 15) == 0);
  const int per = can2 ? 2 : 1;
  const dim3 grid((F + ING_T * per - 1) / (ING_T * per), ch);
#define FX_CM_LAUNCH2(T, V, LG)                                                                                          \
  hipLaunchKernelGGL((fx_col_moments_kernel<T, V, LG>), grid, dim3(ING_T), 0, stream, (const T*)x, ldx, n_rows, F, rows, \
                     med, rpc, pivot, pn, ps, pq)
#define FX_CM_LAUNCH(T, V) do { if (log1p_flag) FX_CM_LAUNCH2(T, V, true); else FX_CM_LAUNCH2(T, V, false); } while (0)
  if (dtype == 0) {
    if (per == 2) FX_CM_LAUNCH(float, 2); else FX_CM_LAUNCH(float, 1);
  } else {
    if (per == 2) FX_CM_LAUNCH(double, 2); else FX_CM_LAUNCH(double, 1);
  }
#undef FX_CM_LAUNCH2
#undef FX_CM_LAUNCH
  int rc = fx_check_launch("fx_col_moments");
  if (rc) return rc;
  hipLaunchKernelGGL(fx_col_moments_merge_kernel, dim3((F + ING_T - 1) / ING_T), dim3(ING_T), 0, stream, pn, ps, pq, pivot, ch, F, count,
                     mean, m2);
  return fx_check_launch("fx_col_moments(merge)");
}

int fx_col_median(const void* x, int dtype, long ldx, int N, const int* cols, int n_cols, double* med_out,
                  hipStream_t stream) {
  FX_REQUIRE(x && cols && med_out, "fx_col_median: null pointer");
  FX_REQUIRE(N > 0 && n_cols > 0 && ldx > 0, "fx_col_median: bad shape");
  FX_REQUIRE(dtype == 0 || dtype == 1, "fx_col_median: dtype must be 0 (f32) or 1 (f64)");
  if (dtype == 0)
    hipLaunchKernelGGL(fx_col_median_kernel<float>, dim3(n_cols), dim3(ING_T), 0, stream, (const float*)x, ldx, N, cols,
                       med_out);
  else
    hipLaunchKernelGGL(fx_col_median_kernel<double>, dim3(n_cols), dim3(ING_T), 0, stream, (const double*)x, ldx, N,
                       cols, med_out);
  return fx_check_launch("fx_col_median");
}

int fx_row_moments(const void* x, int dtype, long ldx, int N, const int* cols, int n_cols, const double* med,
                   double* var_out, hipStream_t stream) {
  FX_REQUIRE(x && cols && var_out, "fx_row_moments: null pointer");
  FX_REQUIRE(N > 0 && n_cols > 0 && ldx > 0, "fx_row_moments: bad shape");
  FX_REQUIRE(dtype == 0 || dtype == 1, "fx_row_moments: dtype must be 0 (f32) or 1 (f64)");
  const dim3 grid((N + ING_RR - 1) / ING_RR);
  if (dtype == 0)
    hipLaunchKernelGGL(fx_row_moments_kernel<float>, grid, dim3(ING_RT), 0, stream, (const float*)x, ldx, N, cols, n_cols,
                       med, var_out);
  else
    hipLaunchKernelGGL(fx_row_moments_kernel<double>, grid, dim3(ING_RT), 0, stream, (const double*)x, ldx, N, cols,
                       n_cols, med, var_out);
  return fx_check_launch("fx_row_moments");
}

int fx_ingest_transform(const void* x, int dtype, long ldx, const int* rows, int n_rows, const int* cols, int n_cols,
                        const double* med, int log1p_flag, const double* mean, const double* scale, float* out, long ldo,
                        hipStream_t stream) {
  FX_REQUIRE(x && out, "fx_ingest_transform: null pointer");
  FX_REQUIRE(n_rows > 0 && n_cols > 0 && ldx > 0 && ldo >= n_cols, "fx_ingest_transform: bad shape");
  FX_REQUIRE(dtype == 0 || dtype == 1, "fx_ingest_transform: dtype must be 0 (f32) or 1 (f64)");
  FX_REQUIRE((mean == nullptr) == (scale == nullptr), "fx_ingest_transform: mean and scale go together");
  const dim3 grid((n_cols + ING_T - 1) / ING_T, (n_rows + ING_ROWS - 1) / ING_ROWS);
#define FX_TR_LAUNCH(T, LG)                                                                                       \
  hipLaunchKernelGGL((fx_ingest_transform_kernel<T, LG>), grid, dim3(ING_T), 0, stream, (const T*)x, ldx, rows, n_rows, \
                     cols, n_cols, med, mean, scale, out, ldo)
  if (dtype == 0) { if (log1p_flag) FX_TR_LAUNCH(float, true); else FX_TR_LAUNCH(float, false); }
  else { if (log1p_flag) FX_TR_LAUNCH(double, true); else FX_TR_LAUNCH(double, false); }
#undef FX_TR_LAUNCH
  return fx_check_launch("fx_ingest_transform");
}

}  // extern "C"
