// What does the memory system give a kernel with the dW+Adam kernel's traffic -- read + write of three fp32 arrays
// [H, F] in 16-byte non-temporal accesses -- as a function of the TILE SHAPE a workgroup streams and of how many
// workgroups a CU holds?  No GEMM: the "gradient" is a constant.   hipcc --offload-arch=gfx950 -O3 scripts/adamprobe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// one workgroup (512 threads) per tile of R rows x C columns (R * C = 8192 or 16384 elements), tiles walked row-block-fastest
// (ORDER 0) or column-fastest (ORDER 1); LDSB bytes of dummy LDS limit the workgroups per CU.
template <int R, int C, int ORDER, int LDSB>
__global__ __launch_bounds__(512) void adam_tiles(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V, int H, int F, long ld) {
  __shared__ char pad[LDSB];
  if (threadIdx.x == 9999) pad[threadIdx.x % LDSB] = 1;
  const int tiles_m = (H + R - 1) / R, tiles_n = (F + C - 1) / C;
  int tm, tn;
  if (ORDER == 0) { tm = blockIdx.x % tiles_m; tn = blockIdx.x / tiles_m; } else { tn = blockIdx.x % tiles_n; tm = blockIdx.x / tiles_n; }
  constexpr int UPR = C / 4, UNITS = R * UPR, PER = UNITS / 512;
  f4 p[PER], m[PER], v[PER];
  long off[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int u = threadIdx.x + 512 * i, r = u / UPR, c4 = u % UPR;
    const int row = tm * R + r, col = tn * C + 4 * c4;
    off[i] = (row < H && col < F) ? ((long)row * ld + col) / 4 : -1;
    if (off[i] >= 0) {
      p[i] = __builtin_nontemporal_load((const f4*)W + off[i]);
      m[i] = __builtin_nontemporal_load((const f4*)M + off[i]);
      v[i] = __builtin_nontemporal_load((const f4*)V + off[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if (off[i] < 0) continue;
    f4 po, mo, vo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = 1e-3f;
      const float m2 = m[i][j] + (g - m[i][j]) * 0.1f;
      const float v2 = v[i][j] * 0.999f + 0.001f * g * g;
      po[j] = p[i][j] - 1e-3f * (m2 / (sqrtf(v2) + 1e-8f));
      mo[j] = m2; vo[j] = v2;
    }
    __builtin_nontemporal_store(po, (f4*)W + off[i]);
    __builtin_nontemporal_store(mo, (f4*)M + off[i]);
    __builtin_nontemporal_store(vo, (f4*)V + off[i]);
  }
}

// persistent: workgroup (tm, c) walks the column tiles c, c + S, ... of its 64-row block (the fused kernel's schedule)
template <int LDSB>
__global__ __launch_bounds__(512) void adam_runs(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V, int H, int F, long ld, int S, int contiguous) {
  __shared__ char pad[LDSB];
  if (threadIdx.x == 9999) pad[threadIdx.x % LDSB] = 1;
  constexpr int R = 64, C = 128, UPR = 32, PER = 4;
  const int tiles_m = (H + R - 1) / R, tiles_n = (F + C - 1) / C;
  int tm = blockIdx.x % tiles_m, c = blockIdx.x / tiles_m;
  if (contiguous == 2) {   // XCD-contiguous row blocks: XCD x (= blockIdx & 7) owns row blocks [x * rpx, (x + 1) * rpx), interleaved tiles
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, rpx = (tiles_m + 7) >> 3;
    tm = xcd * rpx + j % rpx; c = j / rpx;
    if (tm >= tiles_m || c >= S) return;
  }
  int t0 = c, t1 = tiles_n, ts = S;
  if (contiguous == 1) { t0 = (int)((long)c * tiles_n / S); t1 = (int)((long)(c + 1) * tiles_n / S); ts = 1; }
  for (int tn = t0; tn < t1; tn += ts) {
    f4 p[PER], m[PER], v[PER];
    long off[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = threadIdx.x + 512 * i, r = u / UPR, c4 = u % UPR;
      const int row = tm * R + r, col = tn * C + 4 * c4;
      off[i] = (row < H && col < F) ? ((long)row * ld + col) / 4 : -1;
      if (off[i] >= 0) {
        p[i] = __builtin_nontemporal_load((const f4*)W + off[i]);
        m[i] = __builtin_nontemporal_load((const f4*)M + off[i]);
        v[i] = __builtin_nontemporal_load((const f4*)V + off[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (off[i] < 0) continue;
      f4 po, mo, vo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = 1e-3f;
        const float m2 = m[i][j] + (g - m[i][j]) * 0.1f;
        const float v2 = v[i][j] * 0.999f + 0.001f * g * g;
        po[j] = p[i][j] - 1e-3f * (m2 / (sqrtf(v2) + 1e-8f));
        mo[j] = m2; vo[j] = v2;
      }
      __builtin_nontemporal_store(po, (f4*)W + off[i]);
      __builtin_nontemporal_store(mo, (f4*)M + off[i]);
      __builtin_nontemporal_store(vo, (f4*)V + off[i]);
    }
  }
}

// persistent runs, XCD-contiguous row blocks (the shipped mapping), with a ROTATED start tile per run: do the ~500 concurrent streams have to
// march through the same column position?  rot 0: all runs start at their first tile (shipped); 1: XCD x starts x / 8 of the way in;
// 2: every row block at a pseudo-random position; 3: the row blocks of an XCD spread evenly over the run
template <int LDSB>
__global__ __launch_bounds__(512) void adam_runs_rot(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V, int H, int F, long ld, int S, int rot) {
  __shared__ char pad[LDSB];
  if (threadIdx.x == 9999) pad[threadIdx.x % LDSB] = 1;
  constexpr int R = 64, C = 128, UPR = 32, PER = 4;
  const int tiles_m = (H + R - 1) / R, tiles_n = (F + C - 1) / C;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, rpx = (tiles_m + 7) >> 3;
  const int tm = xcd * rpx + j % rpx, c = j / rpx;
  if (tm >= tiles_m || c >= S) return;
  const int T = (tiles_n - c + S - 1) / S;
  int off = 0;
  if (rot == 1) off = (xcd * T) / 8;
  else if (rot == 2) off = (int)(((unsigned)tm * 2654435761u >> 8) % (unsigned)T);
  else if (rot == 3) off = ((j % rpx) * T) / rpx;
  for (int k = 0; k < T; ++k) {
    int kk = k + off; if (kk >= T) kk -= T;
    const int tn = c + kk * S;
    f4 p[PER], m[PER], v[PER];
    long o[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = threadIdx.x + 512 * i, r = u / UPR, c4 = u % UPR;
      const int row = tm * R + r, col = tn * C + 4 * c4;
      o[i] = (row < H && col < F) ? ((long)row * ld + col) / 4 : -1;
      if (o[i] >= 0) {
        p[i] = __builtin_nontemporal_load((const f4*)W + o[i]);
        m[i] = __builtin_nontemporal_load((const f4*)M + o[i]);
        v[i] = __builtin_nontemporal_load((const f4*)V + o[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (o[i] < 0) continue;
      __builtin_nontemporal_store(p[i], (f4*)W + o[i]);
      __builtin_nontemporal_store(m[i], (f4*)M + o[i]);
      __builtin_nontemporal_store(v[i], (f4*)V + o[i]);
    }
  }
}

// persistent runs (the fused kernel's schedule) over INTERLEAVED storage: mode 1 = W separate [H][ld], m / v interleaved per 128-float
// segment ([H][F/128][2][128]: a tile row reads 1 KB contiguous of m | v); mode 2 = all three interleaved ([H][F/128][3][128]: 1.5 KB
// contiguous per tile row, ONE read and ONE write stream).  Is it the number of streams that the pattern-sensitive boxes dislike?
template <int LDSB>
__global__ __launch_bounds__(512) void adam_runs_il(float* __restrict__ W, float* __restrict__ MV, int H, int F, long ld, int S, int mode) {
  __shared__ char pad[LDSB];
  if (threadIdx.x == 9999) pad[threadIdx.x % LDSB] = 1;
  constexpr int R = 64, C = 128, UPR = 32, PER = 4;
  const int tiles_m = (H + R - 1) / R, tiles_n = (F + C - 1) / C;
  const int tm = blockIdx.x % tiles_m, c = blockIdx.x / tiles_m;
  for (int tn = c; tn < tiles_n; tn += S) {
    f4 p[PER], m[PER], v[PER];
    long op[PER], om[PER], ov[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = threadIdx.x + 512 * i, r = u / UPR, c4 = u % UPR;
      const int row = tm * R + r, col = tn * C + 4 * c4;
      const bool ok = row < H && col < F;
      if (mode == 1) {
        op[i] = ok ? ((long)row * ld + col) / 4 : -1;
        om[i] = (((long)row * tiles_n + tn) * 2 + 0) * 32 + c4;        // f4 units: 128 floats = 32 f4
        ov[i] = (((long)row * tiles_n + tn) * 2 + 1) * 32 + c4;
      } else {
        op[i] = ok ? (((long)row * tiles_n + tn) * 3 + 0) * 32 + c4 : -1;
        om[i] = (((long)row * tiles_n + tn) * 3 + 1) * 32 + c4;
        ov[i] = (((long)row * tiles_n + tn) * 3 + 2) * 32 + c4;
      }
      if (op[i] >= 0) {
        p[i] = __builtin_nontemporal_load((const f4*)(mode == 1 ? W : MV) + op[i]);
        m[i] = __builtin_nontemporal_load((const f4*)MV + om[i]);
        v[i] = __builtin_nontemporal_load((const f4*)MV + ov[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (op[i] < 0) continue;
      f4 po, mo, vo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = 1e-3f;
        const float m2 = m[i][j] + (g - m[i][j]) * 0.1f;
        const float v2 = v[i][j] * 0.999f + 0.001f * g * g;
        po[j] = p[i][j] - 1e-3f * (m2 / (sqrtf(v2) + 1e-8f));
        mo[j] = m2; vo[j] = v2;
      }
      __builtin_nontemporal_store(po, (f4*)(mode == 1 ? W : MV) + op[i]);
      __builtin_nontemporal_store(mo, (f4*)MV + om[i]);
      __builtin_nontemporal_store(vo, (f4*)MV + ov[i]);
    }
  }
}

// non-persistent short runs: workgroup = (row block tm, run q) handles column tiles q*r .. q*r + r - 1; consecutive workgroups =
// consecutive runs of the SAME row block (column-fastest), so the resident set covers a few row blocks at full width
template <int LDSB>
__global__ __launch_bounds__(512) void adam_short_runs(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V, int H, int F, long ld, int r, int rot) {
  __shared__ char pad[LDSB];
  if (threadIdx.x == 9999) pad[threadIdx.x % LDSB] = 1;
  constexpr int R = 64, C = 128, UPR = 32, PER = 4;
  const int tiles_n = (F + C - 1) / C, runs = (tiles_n + r - 1) / r;
  const int q = blockIdx.x % runs, tm = blockIdx.x / runs;
  for (int i0 = 0; i0 < r; ++i0) {
    const int tn = q * r + (rot ? (i0 + tm) % r : i0);
    if (tn >= tiles_n) continue;
    f4 p[PER], m[PER], v[PER];
    long off[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int u = threadIdx.x + 512 * i, rr = u / UPR, c4 = u % UPR;
      const int row = tm * R + rr, col = tn * C + 4 * c4;
      off[i] = (row < H && col < F) ? ((long)row * ld + col) / 4 : -1;
      if (off[i] >= 0) {
        p[i] = __builtin_nontemporal_load((const f4*)W + off[i]);
        m[i] = __builtin_nontemporal_load((const f4*)M + off[i]);
        v[i] = __builtin_nontemporal_load((const f4*)V + off[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (off[i] < 0) continue;
      f4 po, mo, vo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = 1e-3f;
        const float m2 = m[i][j] + (g - m[i][j]) * 0.1f;
        const float v2 = v[i][j] * 0.999f + 0.001f * g * g;
        po[j] = p[i][j] - 1e-3f * (m2 / (sqrtf(v2) + 1e-8f));
        mo[j] = m2; vo[j] = v2;
      }
      __builtin_nontemporal_store(po, (f4*)W + off[i]);
      __builtin_nontemporal_store(mo, (f4*)M + off[i]);
      __builtin_nontemporal_store(vo, (f4*)V + off[i]);
    }
  }
}

// TILE-MAJOR storage: every 64 x 128 tile is one contiguous 32 KB chunk per array.  mode 0: non-persistent, workgroup b = chunk b
// (the flat pattern); mode 1: persistent, chunks stored row-block-major [tm][tn], workgroup (tm, c) walks tn = c, c + S, ...;
// mode 2: persistent, chunks stored step-major [k][tm][c] (tile (tm, tn) at ((tn / S) * tiles_m + tm) * S + tn % S), so that at
// any instant the whole grid reads ONE contiguous window of each array.
template <int LDSB>
__global__ __launch_bounds__(512) void adam_chunks(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V, long n_chunks, int tiles_m, int tiles_n, int S, int mode, unsigned long long* cnt) {
  __shared__ char pad[LDSB];
  if (threadIdx.x == 9999) pad[threadIdx.x % LDSB] = 1;
  constexpr int PER = 4;
  long c0 = blockIdx.x, cstep = S, cend = n_chunks;
  if (mode == 0) { c0 = blockIdx.x; cstep = 1; cend = c0 + 1; }
  else if (mode == 1) { const int tm = blockIdx.x % tiles_m, c = blockIdx.x / tiles_m; c0 = (long)tm * tiles_n + c; cstep = S; cend = (long)(tm + 1) * tiles_n; }
  for (long ch = c0; ch < cend; ch += cstep) {
    if (cnt && threadIdx.x == 0) atomicAdd(cnt, 1ull);
    f4 p[PER], m[PER], v[PER];
    long off[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      off[i] = ch * 2048 + threadIdx.x + 512 * i;
      p[i] = __builtin_nontemporal_load((const f4*)W + off[i]);
      m[i] = __builtin_nontemporal_load((const f4*)M + off[i]);
      v[i] = __builtin_nontemporal_load((const f4*)V + off[i]);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      f4 po, mo, vo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = 1e-3f;
        const float m2 = m[i][j] + (g - m[i][j]) * 0.1f;
        const float v2 = v[i][j] * 0.999f + 0.001f * g * g;
        po[j] = p[i][j] - 1e-3f * (m2 / (sqrtf(v2) + 1e-8f));
        mo[j] = m2; vo[j] = v2;
      }
      __builtin_nontemporal_store(po, (f4*)W + off[i]);
      __builtin_nontemporal_store(mo, (f4*)M + off[i]);
      __builtin_nontemporal_store(vo, (f4*)V + off[i]);
    }
  }
}

__global__ void fill_random(float* p, long n, unsigned seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = (float)(x >> 8) * (1.0f / 16777216.0f) * 0.02f + 1e-4f;
  }
}

int main(int argc, char** argv) {
  const int H = 5000, F = 20000; long ld = 20000;
  const bool random_data = argc > 1;
  const bool pitch_sweep = argc > 2 && !strcmp(argv[2], "pitch");      // `adamprobe r pitch`: only the persistent / rows-fastest / cols-fastest patterns, by row pitch
  float *W, *M, *V;
  const size_t bytes = (size_t)H * (pitch_sweep ? 65600 : ld) * 4;
  setvbuf(stdout, NULL, _IOLBF, 0);
  CK(hipMalloc(&W, bytes)); CK(hipMalloc(&M, bytes)); CK(hipMalloc(&V, bytes));
  CK(hipMemset(W, 0, bytes)); CK(hipMemset(M, 0, bytes)); CK(hipMemset(V, 0, bytes));
  if (random_data) {   // the arrays hold noise instead of zeros (data-dependent power: zeros are the easy case)
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, W, (long)H * ld, 1u);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, M, (long)H * ld, 2u);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, V, (long)H * ld, 3u);
    CK(hipDeviceSynchronize());
  }
  printf("data: %s\n", random_data ? "random" : "zeros");
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int it = 10;
    for (int i = 0; i < it; ++i) launch();
    CK(hipGetLastError());
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %8.1f us  %6.3f TB/s\n", name, ms / it * 1e3, 6.0 * H * F * 4 / (ms / it * 1e-3) / 1e12);
  };
#define TILES(R, C, ORDER, LDSB, label) run(label, [&] { const int g = ((H + R - 1) / R) * ((F + C - 1) / C); \
    hipLaunchKernelGGL((adam_tiles<R, C, ORDER, LDSB>), dim3(g), dim3(512), 0, 0, W, M, V, H, F, ld); });
  if (argc > 3 && !strcmp(argv[2], "loop")) {
    // `adamprobe r loop cols|rows|pers <seconds>`: one pattern back to back for a while (scripts/clocks_under_load.sh samples power meanwhile)
    const double secs = argc > 4 ? atof(argv[4]) : 8.0;
    const int g64 = ((H + 63) / 64) * ((F + 127) / 128);
    long n = 0;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    float ms = 0.f;
    while (ms < secs * 1e3) {
      for (int i = 0; i < 200; ++i) {
        if (!strcmp(argv[3], "cols")) hipLaunchKernelGGL((adam_tiles<64, 128, 1, 65536>), dim3(g64), dim3(512), 0, 0, W, M, V, H, F, ld);
        else if (!strcmp(argv[3], "rows")) hipLaunchKernelGGL((adam_tiles<64, 128, 0, 65536>), dim3(g64), dim3(512), 0, 0, W, M, V, H, F, ld);
        else hipLaunchKernelGGL((adam_runs<65536>), dim3(79 * 6), dim3(512), 0, 0, W, M, V, H, F, ld, 6, 0);
      }
      n += 200;
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    printf("%s: %ld launches, %.1f us per launch\n", argv[3], n, ms * 1e3 / n);
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "offsets")) {
    // `adamprobe r offsets`: do the RELATIVE placements of W, m and v matter?  The three arrays of a tile are read (and written) together;
    // separate allocations are 2 MB aligned, i.e. equal in their low 21 address bits.  One arena, m and v shifted by d and 2 d bytes.
    const size_t slot = ((bytes + (2u << 20) - 1) >> 21 << 21) + (8u << 20);
    char* arena; CK(hipMalloc(&arena, 3 * slot + (64u << 20)));
    const long deltas[] = {0, 256, 1024, 4096, 16384, 65536, 262144, 1048576, 1048576 + 4096 + 256, 3 * 1048576 + 7 * 4096 + 512};
    for (int rep = 0; rep < 2; ++rep)
      for (long d : deltas) {
        float* w = (float*)arena; float* m = (float*)(arena + slot + d); float* v = (float*)(arena + 2 * slot + 2 * d);
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, w, (long)H * ld, 1u);
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, m, (long)H * ld, 2u);
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, v, (long)H * ld, 3u);
        CK(hipDeviceSynchronize());
        char name[128];
        snprintf(name, sizeof name, "persistent S=6, one arena, m +%ld B, v +%ld B", d, 2 * d);
        run(name, [&] { hipLaunchKernelGGL((adam_runs<65536>), dim3(79 * 6), dim3(512), 0, 0, w, m, v, H, F, ld, 6, 0); });
        snprintf(name, sizeof name, "cols fastest,   one arena, m +%ld B, v +%ld B", d, 2 * d);
        const int g64 = ((H + 63) / 64) * ((F + 127) / 128);
        run(name, [&] { hipLaunchKernelGGL((adam_tiles<64, 128, 1, 65536>), dim3(g64), dim3(512), 0, 0, w, m, v, H, F, ld); });
      }
    run("persistent S=6, three hipMalloc allocations", [&] { hipLaunchKernelGGL((adam_runs<65536>), dim3(79 * 6), dim3(512), 0, 0, W, M, V, H, F, ld, 6, 0); });
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "arena")) {
    // `adamprobe r arena`: ONE allocation, W at 0, m at `gap`, v at 2 * gap: which spacings of the three arrays run fast?
    CK(hipFree(W)); CK(hipFree(M)); CK(hipFree(V));
    char* arena; CK(hipMalloc(&arena, (size_t)3 << 30));
    const size_t b2 = (bytes + (2u << 20) - 1) >> 21 << 21;       // 384 MB
    const size_t gaps[] = {b2, bytes, b2 + 4096, b2 + 65536, b2 + (1u << 20), b2 + (2u << 20), b2 + (6u << 20), b2 + (16u << 20) + 4096,
                           b2 + (50u << 20), b2 + (64u << 20), b2 + (100u << 20) + 12288, b2 + (127u << 20), b2 + (128u << 20),
                           b2 + (200u << 20) + 8192, (size_t)512 << 20, ((size_t)512 << 20) + (2u << 20), (size_t)640 << 20, (size_t)700 << 20,
                           (size_t)777 << 20, (size_t)1000 << 20};
    for (size_t gap : gaps) {
      float* w = (float*)arena; float* m = (float*)(arena + gap); float* v = (float*)(arena + 2 * gap);
      hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, w, (long)H * ld, 1u);
      hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, m, (long)H * ld, 2u);
      hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, v, (long)H * ld, 3u);
      CK(hipDeviceSynchronize());
      char name[128];
      snprintf(name, sizeof name, "persistent S=6, one arena, gap %zu B (= 384 MB + %ld)", gap, (long)gap - (long)b2);
      run(name, [&] { hipLaunchKernelGGL((adam_runs<65536>), dim3(79 * 6), dim3(512), 0, 0, w, m, v, H, F, ld, 6, 0); });
    }
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "placement")) {
    // `adamprobe r placement`: does WHERE the three arrays land matter?  A dummy allocation of varying size in front of them shifts their
    // virtual (and physical) placement; everything else is identical.
    CK(hipFree(W)); CK(hipFree(M)); CK(hipFree(V));
    const size_t shifts[] = {0, 2u << 20, 6u << 20, 14u << 20, 30u << 20, 62u << 20, 126u << 20, 254u << 20, 510u << 20, (size_t)1022 << 20,
                             (size_t)2046 << 20, 3u << 20, 5u << 20, 100u << 20, 333u << 20, 777u << 20};
    for (size_t sh : shifts) {
      char* dummy = nullptr;
      if (sh) CK(hipMalloc(&dummy, sh));
      float *w, *m, *v;
      CK(hipMalloc(&w, bytes)); CK(hipMalloc(&m, bytes)); CK(hipMalloc(&v, bytes));
      hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, w, (long)H * ld, 1u);
      hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, m, (long)H * ld, 2u);
      hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, v, (long)H * ld, 3u);
      CK(hipDeviceSynchronize());
      char name[128];
      snprintf(name, sizeof name, "persistent S=6, %4zu MB allocated in front (W at %p)", sh >> 20, (void*)w);
      run(name, [&] { hipLaunchKernelGGL((adam_runs<65536>), dim3(79 * 6), dim3(512), 0, 0, w, m, v, H, F, ld, 6, 0); });
      CK(hipFree(w)); CK(hipFree(m)); CK(hipFree(v));
      if (dummy) CK(hipFree(dummy));
    }
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "rotate")) {
    // `adamprobe r rotate`: placements as in `placement`; per placement the persistent pattern with the start tile of the runs rotated
    CK(hipFree(W)); CK(hipFree(M)); CK(hipFree(V));
    const size_t shifts[] = {0, 2u << 20, 6u << 20, 30u << 20, 126u << 20, 254u << 20, 510u << 20, (size_t)1022 << 20, (size_t)2046 << 20, 3u << 20, 100u << 20, 777u << 20};
    for (size_t sh : shifts) {
      char* dummy = nullptr;
      if (sh) CK(hipMalloc(&dummy, sh));
      float *w, *m, *v;
      CK(hipMalloc(&w, bytes)); CK(hipMalloc(&m, bytes)); CK(hipMalloc(&v, bytes));
      hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, w, (long)H * ld, 1u);
      hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, m, (long)H * ld, 2u);
      hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, v, (long)H * ld, 3u);
      CK(hipDeviceSynchronize());
      for (int rot = 0; rot < 4; ++rot) {
        char name[128];
        snprintf(name, sizeof name, "%4zu MB in front, XCD-contiguous S=6, rotation %d", sh >> 20, rot);
        run(name, [&] { hipLaunchKernelGGL((adam_runs_rot<65536>), dim3(8 * 10 * 6), dim3(512), 0, 0, w, m, v, H, F, ld, 6, rot); });
      }
      CK(hipFree(w)); CK(hipFree(m)); CK(hipFree(v));
      if (dummy) CK(hipFree(dummy));
    }
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "streams")) {
    // `adamprobe r streams`: the persistent pattern with 3 + 3, 2 + 2 and 1 + 1 read + write streams (interleaved storage), and the plain one
    float* MV; CK(hipMalloc(&MV, (size_t)H * 157 * 3 * 128 * 4 + 4096));
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, MV, (long)H * 157 * 3 * 128, 5u);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
      run("persistent S=6, W / m / v separate arrays (3 + 3 streams)", [&] { hipLaunchKernelGGL((adam_runs<65536>), dim3(79 * 6), dim3(512), 0, 0, W, M, V, H, F, ld, 6, 0); });
      run("persistent S=6, W separate, m | v interleaved per 512 B (2 + 2 streams)", [&] { hipLaunchKernelGGL((adam_runs_il<65536>), dim3(79 * 6), dim3(512), 0, 0, W, MV, H, F, ld, 6, 1); });
      run("persistent S=6, W | m | v interleaved per 512 B (1 + 1 streams)", [&] { hipLaunchKernelGGL((adam_runs_il<65536>), dim3(79 * 6), dim3(512), 0, 0, W, MV, H, F, ld, 6, 2); });
      TILES(64, 128, 1, 65536, "tile 64x128 cols fastest, separate arrays")
    }
    return 0;
  }
  if (pitch_sweep) {
    // Is the pool's "pattern-sensitive" kind of box sensitive to the ROW PITCH?  (A vertical band of tiles -- what persistent runs
    // and rows-fastest grids touch at any instant -- is a stride-`pitch` address pattern; an uneven channel / bank hash of that
    // stride would show as a pitch dependence.)
    for (long p : {20000L, 20032L, 20480L, 24576L, 28672L, 32768L, 32800L, 40960L, 49152L, 65536L, 65568L}) {
      ld = p;
      char nm[160];
      snprintf(nm, 160, "pitch %5ld floats (%6ld B): persistent S=6 interleaved", p, p * 4);
      run(nm, [&] { hipLaunchKernelGGL((adam_runs<65536>), dim3(79 * 6), dim3(512), 0, 0, W, M, V, H, F, ld, 6, 0); });
      snprintf(nm, 160, "pitch %5ld floats (%6ld B): 64x128 rows fastest", p, p * 4);
      TILES(64, 128, 0, 65536, nm)
      snprintf(nm, 160, "pitch %5ld floats (%6ld B): 64x128 cols fastest", p, p * 4);
      TILES(64, 128, 1, 65536, nm)
    }
    return 0;
  }
  TILES(128, 128, 0, 65536, "tile 128x128 (512 B segs), rows fastest, 2 WG/CU")
  TILES(128, 128, 1, 65536, "tile 128x128 (512 B segs), cols fastest, 2 WG/CU")
  TILES(64, 128, 0, 65536, "tile  64x128 (512 B segs), rows fastest, 2 WG/CU")
  TILES(64, 128, 1, 65536, "tile  64x128 (512 B segs), cols fastest, 2 WG/CU")
  TILES(64, 128, 1, 32768, "tile  64x128 (512 B segs), cols fastest, 4 WG/CU")
  TILES(64, 128, 0, 32768, "tile  64x128 (512 B segs), rows fastest, 4 WG/CU")
  TILES(32, 256, 1, 65536, "tile  32x256 (1 KB segs),  cols fastest, 2 WG/CU")
  TILES(32, 256, 0, 65536, "tile  32x256 (1 KB segs),  rows fastest, 2 WG/CU")
  TILES(16, 512, 1, 65536, "tile  16x512 (2 KB segs),  cols fastest, 2 WG/CU")
  TILES(16, 512, 0, 65536, "tile  16x512 (2 KB segs),  rows fastest, 2 WG/CU")
  TILES(8, 1024, 1, 65536, "tile   8x1024 (4 KB segs), cols fastest, 2 WG/CU")
  TILES(4, 2048, 1, 65536, "tile   4x2048 (8 KB segs), cols fastest, 2 WG/CU")
  TILES(4, 2048, 1, 16384, "tile   4x2048 (8 KB segs), cols fastest, 4+ WG/CU")
  TILES(64, 128, 1, 16384, "tile  64x128 (512 B segs), cols fastest, 4+ WG/CU")
  for (int S : {6, 12}) for (int contiguous : {0, 1}) {
    char nm[128];
    snprintf(nm, 128, "persistent runs 64x128, S=%d, %s, 2 WG/CU", S, contiguous ? "contiguous chunks" : "interleaved tiles");
    run(nm, [&] { hipLaunchKernelGGL((adam_runs<65536>), dim3(79 * S), dim3(512), 0, 0, W, M, V, H, F, ld, S, contiguous); });
  }
  for (int S : {6, 12}) {
    char nm[128];
    snprintf(nm, 128, "persistent runs 64x128, S=%d, interleaved, XCD-contiguous row blocks, 2 WG/CU", S);
    run(nm, [&] { hipLaunchKernelGGL((adam_runs<65536>), dim3(8 * 10 * S), dim3(512), 0, 0, W, M, V, H, F, ld, S, 2); });
  }
  for (int S : {12, 24}) {
    char nm[128];
    snprintf(nm, 128, "persistent runs 64x128, S=%d, interleaved, 4 WG/CU", S);
    run(nm, [&] { hipLaunchKernelGGL((adam_runs<32768>), dim3(79 * S), dim3(512), 0, 0, W, M, V, H, F, ld, S, 0); });
  }
  for (int r : {1, 2, 4, 8, 13, 26}) {
    char nm[128];
    snprintf(nm, 128, "short runs of %d tiles 64x128, column-fastest, 2 WG/CU", r);
    const int runs = (157 + r - 1) / r;
    run(nm, [&] { hipLaunchKernelGGL((adam_short_runs<65536>), dim3(79 * runs), dim3(512), 0, 0, W, M, V, H, F, ld, r, 0); });
  }
  {
    const int tiles_m = 78, tiles_n = 156;                      // 78 * 156 * 8192 = 99.68 M elements <= H * F
    const long n_chunks = (long)tiles_m * tiles_n;
    unsigned long long* cnt; CK(hipMalloc(&cnt, 8)); CK(hipMemset(cnt, 0, 8));
    auto chunks = [&](const char* nm, int grid, int S, int mode) {
      run(nm, [&] { hipLaunchKernelGGL((adam_chunks<65536>), dim3(grid), dim3(512), 0, 0, W, M, V, n_chunks, tiles_m, tiles_n, S, mode, cnt); });
      unsigned long long h = 0; CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost)); printf("   chunks visited so far: %llu (12 launches of %ld expected per line)\n", h, n_chunks);
    };
    // note: TB/s printed assumes H*F elements; these touch 79*156*8192 (0.9908 of it) -- scale by 0.997
    chunks("tile-major chunks, non-persistent (flat), 2 WG/CU  [x0.997]", (int)n_chunks, 0, 0);
    chunks("tile-major [tm][tn], persistent S=6, 2 WG/CU       [x0.997]", 78 * 6, 6, 1);
    chunks("tile-major [tm][tn], persistent S=12, 2 WG/CU      [x0.997]", 78 * 12, 12, 1);
    chunks("tile-major [k][tm][c], persistent grid 468, 2 WG/CU [x0.997]", 468, 468, 2);
    chunks("tile-major [k][tm][c], persistent grid 512, 2 WG/CU [x0.997]", 512, 512, 2);
    chunks("tile-major [k][tm][c], persistent grid 948, 2 WG/CU [x0.997]", 948, 948, 2);
  }
  return 0;
}
