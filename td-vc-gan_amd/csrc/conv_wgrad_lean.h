// Parameter block of the lean weight-gradient kernels (conv_wgrad_lean.hip, conv_wgrad_pipe.hip), shared with conv_api.hip.
#pragma once
#include "conv_common.h"

namespace tdvc {

struct WgLeanP {
  Opnd a;            // dy-like rows [B][R][N]
  Opnd x;            // x-like rows  [B][Cin][T]
  int R, Cin, N, pad, K, reflect, B;
  int lo, span, i0;
  int ntiles;        // 256-step chunks per sample
  int tpb, ngroups;  // chunks per block, chunk groups per sample (slabs = B * ngroups)
  float* slab; long slab_stride;
  int vec;
  int xrp, xnp;      // pipelined tile kernel: x-tile rows per row-walk pass, passes
  long bias_off;     // >= 0: per-slab bias partial sums (row sums of the staged dy' tile) at slab[bias_off + row]
};

hipError_t launch_conv_wgrad_lean(WgLeanP p, int B, int J, int D, hipStream_t st);
bool wgrad_lean_supported(int J, int D);
int wgrad_lean_nslab(int R, int Cin, int N, int K, int B);
// 136 input channels cost 3 tiles of 64 (192, 29 % padding) but 5 tiles of 32 (160, 15 %): take the narrower tile when it
// saves more than 10 % of the MFMA work
static inline bool wgrad_prefers_ct32(int Cin) { return ((Cin + 31) / 32 * 32) * 10 < ((Cin + 63) / 64 * 64) * 9; }
// pipelined register-tile kernel (conv_wgrad_pipe.hip); returns hipErrorNotSupported outside its contract
hipError_t launch_conv_wgrad_pipe(WgLeanP& p, int J, int D, hipStream_t st);

}  // namespace tdvc
