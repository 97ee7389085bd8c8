// Internal (device + host) definitions shared by the conv kernels. Not part of the C ABI.
//
// Every Conv1d / ConvTranspose1d forward, input-grad and weight-grad on the path is mapped by the
// host onto ONE "reduced" stride-1 problem per (group, sample):
//     Y'[r][n] = sum_{c < Cred} sum_{j < J}  A[r][c][j] * X'[c][n + j*d + lo]
// with three staging modes that define X' (and how rows/cols map back to tensors):
//   DIRECT : X'[c][q]        = xf(x[c][q])                 (stride-1 conv, also its dgrad with flipped taps)
//   DOWN   : X'[(c,phi)][t]  = xf(x[c][t*s + phi - pad])   (strided conv as a time-to-depth + J=ceil(K/s) tap conv)
//   UP     : rows r=(m,phi); out position u = n*s + phi - pad (transposed conv as J-tap conv + depth-to-time)
// so dilated, strided, grouped and transposed convolutions share one MFMA main loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tdvc {

enum { MODE_DIRECT = 0, MODE_DOWN = 1, MODE_UP = 2 };
enum { XF_NONE = 0, XF_LRELU = 1, XF_FILM_LRELU = 2, XF_MASK_LRELU = 3, XF_MASK_TANH = 4 };
enum { EPI_FWD = 0, EPI_MASK = 1, EPI_FILM = 2, EPI_PLAIN = 3 };
enum { POST_NONE = 0, POST_LRELU = 1, POST_TANH = 2 };

struct Xf {
  int kind; float slope; float scale;
  const float* aux; long aux_bs;
};

// Tensor operand [B][groups*Cg][T], batch stride bs (elements), channel stride T.
struct Opnd {
  const float* p; long bs; int T; int Cg; Xf xf;
};

struct GemmConvP {
  Opnd x;                       // B-matrix source
  const float* w; long w_sg, w_sm, w_sc; int K; int tap_flip;   // A[r][c][k] = w[g*w_sg + m*w_sm + c*w_sc + k]
  int mode, s, d, pad, reflect, J;
  int R, Cred, Cc, N, groups, lo, span, XS, WS;
  int mirror_pad;               // > 0: fold reflect-pad halo (dgrad of a reflect conv)
  int stage_rows;               // DOWN with a large stride: stage with lanes along the reduced rows
  float* y; long y_bs; int Ty; int Cy_g;
  int epi;
  const float* bias;
  const float* bias3;           // [B][Ctot][3] edge/interior/edge bias
  const float* res; long res_bs;
  int post; float post_slope; float out_scale;
  const float* add; long add_bs; float add_scale;
  const float* mx; long mx_bs; float m_slope;
  const float* gb; long gb_bs; float* dgb; long dgb_bs;
};

struct WgradP {
  Opnd a;                       // rows operand (dy-like), N columns
  Opnd x;                       // column operand (x-like)
  int mode, s, d, pad, reflect, J, K;
  int R, Cred, N, groups, lo, span, NTc, XS, AS;
  int ntiles;                   // time chunks per sample
  float* slab; long slab_stride; // slab[(b*ntiles+tile)][groups*R*Cx_g*K (+ bias rows)]
  long w_sg, w_sm, w_sc;        // slab element index = g*w_sg + m*w_sm + c*w_sc + k
};

__device__ __forceinline__ float lrelu_f(float v, float s) { return v > 0.f ? v : v * s; }

// Value of the transformed operand element (b, ch, t) given its raw value v; ch is the absolute channel.
__device__ __forceinline__ float apply_xf(const Xf& xf, float v, int b, int ch, int t, int T, int Ctot) {
  switch (xf.kind) {
    case XF_LRELU: v = lrelu_f(v, xf.slope); break;
    case XF_FILM_LRELU: {
      const float* g = xf.aux + (long)b * xf.aux_bs + (long)ch * T + t;
      v = lrelu_f(v * (1.f + g[0]) + g[(long)Ctot * T], xf.slope);
    } break;
    case XF_MASK_LRELU: {
      float a = xf.aux[(long)b * xf.aux_bs + (long)ch * T + t];
      v = a > 0.f ? v : v * xf.slope;
    } break;
    case XF_MASK_TANH: {
      float a = xf.aux[(long)b * xf.aux_bs + (long)ch * T + t];
      v = v * (1.f - a * a);
    } break;
    default: break;
  }
  return v * xf.scale;
}

// Fetch transformed operand at absolute channel ch, time q with zero / reflect padding.
__device__ __forceinline__ float fetch_opnd(const Opnd& o, int b, int ch, int q, int reflect, int Ctot) {
  if (reflect) {
    if (q < 0) q = -q;
    else if (q >= o.T) q = 2 * (o.T - 1) - q;
  }
  if (q < 0 || q >= o.T) return 0.f;
  float v = o.p[(long)b * o.bs + (long)ch * o.T + q];
  return apply_xf(o.xf, v, b, ch, q, o.T, Ctot);
}

}  // namespace tdvc
