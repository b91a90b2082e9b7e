// fx_chain_prof.h -- phase stamps of the narrow-chain kernels (-DFX_CHAIN_PROFILE: scripts/build_variant.py, never the shipped library).
// Thread 0 of a workgroup writes the 100 MHz wall clock at phase boundaries into a per-translation-unit table; the reader the unit
// declares with CP_READER copies it out (scripts/chain_profile.py turns it into a per-phase table).  Without the flag the macros vanish.
#pragma once
#ifdef FX_CHAIN_PROFILE
#define CP_SLOTS 16
#define CP_WGS 1024
static __device__ unsigned long long cp_stamp[CP_WGS * CP_SLOTS];
#define CP_TICK(wg, k) do { if (threadIdx.x == 0 && (int)(wg) < CP_WGS) cp_stamp[(int)(wg) * CP_SLOTS + (k)] = wall_clock64(); } while (0)
#define CP_READER(name)                                                                                                    \
  extern "C" int name(unsigned long long* out, int reset) {                                                               \
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(cp_stamp), sizeof(cp_stamp)) != hipSuccess) return -1;                         \
    if (reset) { static unsigned long long z[CP_WGS * CP_SLOTS]; if (hipMemcpyToSymbol(HIP_SYMBOL(cp_stamp), z, sizeof(z)) != hipSuccess) return -1; } \
    return 0;                                                                                                              \
  }
#else
#define CP_TICK(wg, k) do { } while (0)
#define CP_READER(name)
#endif
