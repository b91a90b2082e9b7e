// Index sampling on the device (gfx950): the epoch's shuffle and the triplet network's positive / negative draws.
//
// Replaces, on the training path, what the reference does on the host: DataLoader(shuffle=True)'s torch.randperm per epoch
// (main.py:289-298) and TripletMultiOmicDataset.__getitem__'s np.random.choice draws (data.py:1106-1131).  Round 3 used torch's
// device randperm / rand / bucketize here, i.e. rocprim sort and at::native kernels inside the timed region.
//
//   fx_randperm        a uniformly random permutation: element i gets the 32-bit Philox4x32-10 word (seed, offset, i) as its key, the
//                      pairs (key << 32 | i) are sorted -- n <= 4096: one workgroup, bitonic network in LDS (32 KB); larger n: the same
//                      network with the strides >= 4096 as global passes -- and the low words are the permutation (ties between
//                      keys, probability n^2 / 2^33, fall back to index order).  Optionally composed with an index map (out = src[perm]).
//   fx_triplet_sample  per anchor: positive = uniform among the OTHER members of the anchor's label group, negative = uniform member
//                      of a uniformly chosen other group (the reference's two-stage draw), from three Philox words.
#include "fx_common.h"

typedef unsigned long long u64;

__device__ __forceinline__ u64 rp_key(u64 seed, u64 offset, unsigned i) {
  uint32_t r[4];
  fx_philox4(seed, offset + (i >> 2), r);
  return ((u64)r[i & 3] << 32) | (u64)i;
}

// ascending bitonic compare-exchange of elements i and i ^ j inside a sequence whose direction is given by (i & k)
__device__ __forceinline__ void rp_cx(u64& a, u64& b, bool up) {
  const bool sw = (a > b) == up;
  if (sw) { const u64 t = a; a = b; b = t; }
}

#define RP_CHUNK 4096
__global__ __launch_bounds__(1024) void fx_randperm_small_kernel(long* __restrict__ out, const long* __restrict__ src, int n, int P,
                                                                 u64 seed, u64 offset) {
  __shared__ u64 s[RP_CHUNK];
  for (int i = threadIdx.x; i < P; i += 1024) s[i] = i < n ? rp_key(seed, offset, (unsigned)i) : ~0ull;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (P >> 1); t += 1024) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));          // the lower index of the pair (bit j clear)
        rp_cx(s[i], s[i | j], (i & k) == 0);
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < n; i += 1024) {
    const long idx = (long)(s[i] & 0xFFFFFFFFull);
    out[i] = src ? src[idx] : idx;
  }
}

__global__ __launch_bounds__(256) void fx_randperm_keys_kernel(u64* __restrict__ keys, long n, long P, u64 seed, u64 offset) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long)gridDim.x * blockDim.x)
    keys[i] = i < n ? rp_key(seed, offset, (unsigned)i) : ~0ull;
}
// strides j < RP_CHUNK of stage k (k_lo = 2 for the first launch: all stages up to RP_CHUNK; otherwise only stage k) inside LDS
__global__ __launch_bounds__(1024) void fx_bitonic_local_kernel(u64* __restrict__ keys, long k_first, long k_last) {
  __shared__ u64 s[RP_CHUNK];
  const long base = (long)blockIdx.x * RP_CHUNK;
  for (int i = threadIdx.x; i < RP_CHUNK; i += 1024) s[i] = keys[base + i];
  __syncthreads();
  for (long k = k_first; k <= k_last; k <<= 1)
    for (int j = (int)((k >> 1) < RP_CHUNK ? (k >> 1) : (RP_CHUNK >> 1)); j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (RP_CHUNK >> 1); t += 1024) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        rp_cx(s[i], s[i | j], ((base + i) & k) == 0);
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < RP_CHUNK; i += 1024) keys[base + i] = s[i];
}
__global__ __launch_bounds__(256) void fx_bitonic_global_kernel(u64* __restrict__ keys, long P, long j, long k) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (P >> 1); t += (long)gridDim.x * blockDim.x) {
    const long i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    u64 a = keys[i], b = keys[i | j];
    const u64 a0 = a;
    rp_cx(a, b, (i & k) == 0);
    if (a != a0) { keys[i] = a; keys[i | j] = b; }
  }
}
__global__ __launch_bounds__(256) void fx_randperm_emit_kernel(long* __restrict__ out, const long* __restrict__ src, const u64* __restrict__ keys, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long idx = (long)(keys[i] & 0xFFFFFFFFull);
    out[i] = src ? src[idx] : idx;
  }
}

__global__ __launch_bounds__(256) void fx_triplet_sample_kernel(long* __restrict__ pos, long* __restrict__ neg, const long* __restrict__ anchors,
                                                                long n, const long* __restrict__ gid, const long* __restrict__ order,
                                                                const long* __restrict__ starts, const long* __restrict__ counts,
                                                                const long* __restrict__ rank_in_group, int n_groups, u64 seed, u64 offset,
                                                                int* __restrict__ err) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long a = anchors[i], g = gid[a], cnt = counts[g];
    uint32_t r[4];
    fx_philox4(seed, offset + (u64)i, r);
    const float u1 = (float)(r[0] >> 8) * (1.0f / 16777216.0f), u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f),
                u3 = (float)(r[2] >> 8) * (1.0f / 16777216.0f);                      // [0, 1)
    if (cnt < 2) {                       // no other member: the reference raises (np.random.choice of an empty list)
      atomicOr(err, 1);
      pos[i] = a;
    } else {
      long rr = (long)(u1 * (float)(cnt - 1));
      rr = rr > cnt - 2 ? cnt - 2 : rr;
      rr += (rr >= rank_in_group[a]) ? 1 : 0;                                       // skip the anchor itself
      pos[i] = order[starts[g] + rr];
    }
    long og = (long)(u2 * (float)(n_groups - 1));
    og = og > n_groups - 2 ? n_groups - 2 : og;
    og += (og >= g) ? 1 : 0;                                                        // skip the anchor's group
    const long ocnt = counts[og];
    long rn = (long)(u3 * (float)ocnt);
    rn = rn > ocnt - 1 ? ocnt - 1 : rn;
    neg[i] = order[starts[og] + rn];
  }
}

static long rp_pow2(long n) {
  long p = 1;
  while (p < n) p <<= 1;
  return p;
}

extern "C" {

// bytes of scratch fx_randperm needs for n elements (0 for n <= 4096)
long fx_randperm_scratch_bytes(long n) { return n <= RP_CHUNK ? 0 : rp_pow2(n) * (long)sizeof(u64); }

int fx_randperm(long* out, const long* src, long n, unsigned long long seed, unsigned long long offset, void* scratch,
                long scratch_bytes, hipStream_t stream) {
  FX_REQUIRE(out && n > 0 && n < (1L << 31), "fx_randperm: bad args (n = %ld)", n);
  if (n <= RP_CHUNK) {
    const long P = rp_pow2(n < 2 ? 2 : n);
    hipLaunchKernelGGL(fx_randperm_small_kernel, dim3(1), dim3(1024), 0, stream, out, src, (int)n, (int)P, seed, offset);
    return fx_check_launch("fx_randperm");
  }
  const long P = rp_pow2(n);
  FX_REQUIRE(scratch && scratch_bytes >= P * (long)sizeof(u64), "fx_randperm: %ld bytes of scratch needed for n = %ld", P * (long)sizeof(u64), n);
  u64* keys = (u64*)scratch;
  const int gb = (int)((P / 2 + 255) / 256 > 4096 ? 4096 : (P / 2 + 255) / 256);
  hipLaunchKernelGGL(fx_randperm_keys_kernel, dim3(gb), dim3(256), 0, stream, keys, n, P, seed, offset);
  hipLaunchKernelGGL(fx_bitonic_local_kernel, dim3((unsigned)(P / RP_CHUNK)), dim3(1024), 0, stream, keys, 2L, (long)RP_CHUNK);
  for (long k = 2L * RP_CHUNK; k <= P; k <<= 1) {
    for (long j = k >> 1; j >= RP_CHUNK; j >>= 1) hipLaunchKernelGGL(fx_bitonic_global_kernel, dim3(gb), dim3(256), 0, stream, keys, P, j, k);
    hipLaunchKernelGGL(fx_bitonic_local_kernel, dim3((unsigned)(P / RP_CHUNK)), dim3(1024), 0, stream, keys, k, k);
  }
  hipLaunchKernelGGL(fx_randperm_emit_kernel, dim3(gb), dim3(256), 0, stream, out, src, keys, n);
  return fx_check_launch("fx_randperm");
}

int fx_triplet_sample(long* pos, long* neg, const long* anchors, long n, const long* gid, const long* order, const long* starts,
                      const long* counts, const long* rank_in_group, int n_groups, unsigned long long seed, unsigned long long offset,
                      int* err_flag, hipStream_t stream) {
  FX_REQUIRE(pos && neg && anchors && gid && order && starts && counts && rank_in_group && err_flag && n > 0 && n_groups >= 2,
             "fx_triplet_sample: bad args (n = %ld, %d label groups; two groups at least)", n, n_groups);
  const int gb = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(fx_triplet_sample_kernel, dim3(gb), dim3(256), 0, stream, pos, neg, anchors, n, gid, order, starts, counts,
                     rank_in_group, n_groups, seed, offset, err_flag);
  return fx_check_launch("fx_triplet_sample");
}

}  // extern "C"
