// Parameter block of the lean conv kernel (conv_lean.hip), shared with the host mapping in conv_api.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace tdvc {

enum { LXF_ACT = 0, LXF_FILM = 1, LXF_MASK_LRELU = 2, LXF_MASK_TANH = 3, LXF_COND = 4 };   // prologue kinds

struct LeanP {
  const float* x; const float* w; float* y;
  const float* bias; const float* bias3; const float* res; const float* add;
  const float* aux;                    // prologue second tensor: FiLM gamma/beta or activation output
  const float* mx; const float* gb; float* dgb;
  int x_bs, y_bs, res_bs, add_bs, aux_bs, mx_bs, gb_bs, dgb_bs;
  int T, Cin, Cout, Cw, K, d, pad, flip, reflect, mirror;
  int Cc, span, lo, i0, XS, WS;
  int xrp, xnp, wrp, wnp;              // row-walk staging: rows per pass / passes for the input tile and the weight tile
  int post;
  const float* cw; const float* k3; float* cv0;   // LXF_COND: cond_var.0 excitation-window weights, edge bias, cv0 output
  int cw_stride, Cv, cv0_bs, ES;
  unsigned* sbits; const unsigned* mbits;   // sign-bit output of the forward epilogue / sign-bit mask source of EPI_MASK ([B][C][T/32] words)
  int sb_bs, mb_bs;                    // their batch strides in words
  int swz;                             // XCD-aware block order (see conv_lean_kernel)
  int fold, seg, Bn;                   // FOLD: samples per 64-column tile, LDS columns per sample segment, batch size
  int vec;                             // host-checked: T % 4 == 0, every pointer 16-byte aligned, batch strides % 4 == 0
  float slope, in_scale, out_scale, add_scale, m_slope;
};

hipError_t launch_conv_lean(LeanP p, int B, int xfk, int epi, hipStream_t st);
hipError_t launch_conv_lean_cond(LeanP p, int B, hipStream_t st);

}  // namespace tdvc
