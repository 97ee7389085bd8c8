// Parameter block of the fused FiLM-block forward kernel (film_block.hip), shared with its C entry point.
#pragma once
#include <hip/hip_runtime.h>

namespace tdvc {

struct FilmBlockP {
  const float* x; const float* w1; const float* b1; float* h;
  const float* gb; const float* w2; const float* b2; const float* add; float* y;
  int x_bs, h_bs, gb_bs, add_bs, y_bs;
  int T, K, d;
  int span, lo, i0, XS, WS, xrp, xnp, wrp, wnp, xs_floats;
  float slope, scale;
};

hipError_t launch_film_block_fwd(FilmBlockP p, int B, hipStream_t st);

}  // namespace tdvc
