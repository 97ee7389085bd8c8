// Parameters and launchers of the vector-ALU kernels for grouped strided convs with 4 -> 4 channels per group
// (conv_small_group.hip). Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

namespace tdvc {

struct SmallGroupP {
  const float* x; long x_bs;            // [B][G*4][Tin]
  const float* w;                       // [G*4][4][K]
  const float* bias;                    // fwd: [G*4] or null
  const float* mask; long mask_bs;      // dgrad / wgrad: stored post-activation y (dy is multiplied by lrelu'(y)), or null
  float* y; long y_bs;                  // fwd: y [B][G*4][Tout]; dgrad: dx [B][G*4][Tin]
  const float* dy; long dy_bs;          // dgrad / wgrad
  float* slab; long slab_stride;        // wgrad
  int B, G, Tin, Tout, K, s, pad, J;    // J = ceil(K / s)
  int act_in; float slope_in;           // fwd / wgrad: LeakyReLU on x
  int post; float post_slope; float in_scale, out_scale, dy_scale; float m_slope;
  int CS;                               // LDS column stride of the time-to-depth tile
  int nsplit;                           // wgrad: sample groups per group of channels
};

hipError_t launch_small_group_fwd(SmallGroupP p, hipStream_t st);
hipError_t launch_small_group_dgrad(SmallGroupP p, hipStream_t st);
hipError_t launch_small_group_wgrad(SmallGroupP p, float* dw, float* dbias, void* workspace, size_t workspace_bytes, hipStream_t st);
size_t small_group_wgrad_workspace(int B, int G, int K);

}  // namespace tdvc
