// Argument blocks + host launchers of the register-resident BatchNorm kernels (fx_fused_small.hip), shared with
// the extern "C" entry points in fx_norm_act.hip.
#pragma once
#include "fx_common.h"

struct BnFwd16 {
  float* out; float* x_out; const float* x;
  const float* slabs; int nslabs; long slab_stride; const float* lin_bias;   // optional fused split-K reduce
  const float* gamma; const float* beta; float* running_mean; float* running_var;
  float* save_mean; float* save_invstd; const float* mask;
  int B, C; long ldx, ldo;
  int pre_act, post_act, train; float drop_p;
  unsigned long long seed, offset; const float* ctrl;
};

struct BnBwd16 {
  float* dx; float* dgamma; float* dbeta; float* dbias;
  const float* dout; const float* x; const float* out; const float* gamma;
  const float* save_mean; const float* save_invstd;
  int B, C; long ldx, ldo, lddo, lddx;
  int pre_act, post_act; float drop_p; int accumulate;
};

int fx_launch_bn_fwd_r16(const BnFwd16& a, hipStream_t stream);
int fx_launch_bn_bwd_r16(const BnBwd16& a, hipStream_t stream);
#define FX_BN_R16_MAX_B 128
