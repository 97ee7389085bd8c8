// Forward of a 3-tap stride-1 'same' conv with 65..160 input channels (FiLM's cond_var.2, model/generator.py:86-92: 136 -> 2C,
// ~9 % of the train step) on the bf16 matrix pipe at fp32 accuracy -- the split-bf16 x6 scheme of conv_wgrad_x6.hip (three
// exact bf16 pieces per operand, six piece products on v_mfma_f32_16x16x32_bf16, fp32 accumulation).
//     y[co][t] = bias[co] + sum_{ci,j} W[co][ci][j] * x'[ci][t + j - 1],      x' = LeakyReLU(x) (or x)
// K of the product is (tap, input channel): a lane's 8 consecutive k values are 8 consecutive CHANNELS at one time step, so the
// activation tile is held TRANSPOSED in LDS ([time][channel], channel fastest; the tap shift is a row shift -> every fragment
// read is an aligned ds_read_b128). The transposition happens in registers on the way in: a staging thread loads 8 channel
// rows x 4 consecutive steps (8 coalesced float4 loads), splits the 32 values and writes, per step and piece, the 8 channels
// as one 16-byte LDS store. The weights arrive pre-split ([piece][co][tap][160 channels] bf16, tdvc_conv_x6_weight_planes:
// once per optimizer step) and are copied as they are.
// Block = 128 steps x (32 | 64) output channels, reduction in chunks of 32 input channels (x all 3 taps); wave w owns steps
// 32w .. 32w + 31 (two 16-row tiles) x all output-channel tiles. Loads of chunk c + 1 are in flight during the MFMAs of chunk c.
#include "conv_common.h"
#include "api_util.h"

PROF_DEFINE(tdvc_debug_fwdx6_prof)

namespace tdvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct FwdX6P {
  const float* x; long x_bs;            // [B][Cin][T]
  const unsigned short* wp;             // [3 pieces][Cout][3 taps][FX_CP] bf16
  const float* bias;                    // [Cout] or null
  float* y; long y_bs;
  int T, Cin, Cout;
  float slope;                          // LeakyReLU slope of the prologue (1 = none)
};

constexpr int FX_NT = 128;              // steps per block
constexpr int FX_CP = 160;              // padded channel count of the weight planes (5 chunks of 32)
constexpr int FX_RS = 40;               // bf16 row stride of both LDS images: 80 B -> 16 rows start on 16 distinct 4-bank groups
constexpr int FX_XROWS = FX_NT + 8;     // window [n0 - 4, n0 + 132): row i <-> position n0 - 4 + i
constexpr int FX_XPL = FX_XROWS * FX_RS;
constexpr int FX_NV = FX_XROWS / 4;     // 34 float4 columns per channel row

__device__ __forceinline__ void split1(const float f, unsigned& h, unsigned& m, unsigned& l) {     // exact: f = h + m + l (upper halves)
  h = __builtin_bit_cast(unsigned, f) & 0xffff0000u;
  const float r1 = f - __builtin_bit_cast(float, h);
  m = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
  const float r2 = r1 - __builtin_bit_cast(float, m);
  l = __builtin_bit_cast(unsigned, r2) & 0xffff0000u;
}

template <int CO_TILES>
__global__ __launch_bounds__(256, 2) void conv_fwd_x6_kernel(const FwdX6P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  constexpr int MT = 16 * CO_TILES;
  constexpr int WPL = MT * 3 * FX_RS;                 // elements of one weight piece: [co][tap][FX_RS]
  constexpr int WNV = 3 * MT * 3 * 4;                 // 16-byte vectors of a weight chunk: (piece, co, tap) x 4
  constexpr int WPT = (WNV + 255) / 256;
  unsigned short* xt = smem16;                        // [3 pieces][FX_XROWS][FX_RS]
  unsigned short* ws = smem16 + 3 * FX_XPL;           // [3 pieces][MT][3][FX_RS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, g = lane >> 4;
  const int T = p.T, n0 = blockIdx.x * FX_NT, r0 = blockIdx.y * MT, b = blockIdx.z;
  const int nchunk = (p.Cin + 31) / 32;

  // staging roles. x: thread (channel group cg of 8, float4 column v) for tid < 4 * 34; weights: vector e = tid + i * 256
  const int cg = tid / FX_NV, xv = tid - cg * FX_NV;
  const bool xact = tid < 4 * FX_NV;
  const int q = n0 - 4 + 4 * xv;
  const bool xin = xact && q >= 0 && q < T;            // T % 4 == 0: a float4 lies wholly inside or outside
  const srd_t xrs = make_srd(p.x + (long)b * p.x_bs, p.Cin * T * 4);         // channels >= Cin read as zero
  const srd_t wrs = make_srd(reinterpret_cast<const float*>(p.wp), 3 * p.Cout * 3 * FX_CP * 2);

  f32x4 acc[2][CO_TILES];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int ct = 0; ct < CO_TILES; ++ct) acc[tt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 xr[8];
  u32x4 wr[WPT];
  auto issue = [&](int c) {
    const int c0 = c * 32;
#pragma unroll
    for (int e = 0; e < 8; ++e) xr[e] = buf_load4(xrs, xin ? ((c0 + 8 * cg + e) * T + q) * 4 : 0x7f000000);
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = tid + i * 256;                    // (piece, co, tap, 16-byte quarter)
      const int row = e >> 2, qu = e & 3;             // row = (piece * MT + co) * 3 + tap
      const int pc = row / (MT * 3), rem = row - pc * MT * 3;
      const int co = rem / 3, j = rem - co * 3;
      wr[i] = __builtin_bit_cast(u32x4, buf_load4(wrs, e < WNV ? (((pc * p.Cout + r0 + co) * 3 + j) * FX_CP + c0 + 8 * qu) * 2 : 0x7f000000));
    }
  };
  PROF_DECL
  issue(0);
  PROF(0)

  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                                  // the previous chunk's fragments are consumed
    PROF(1)
    PROF_WAITV()
    PROF(2)
    if (xact) {
      // 8 channels x 4 steps in registers -> per step and piece one 16-byte store of the 8 channels
      unsigned hh[4][8], mm[4][8], ll[4][8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float f0 = xr[e][k];
          const float f = fmaxf(f0, f0 * p.slope);    // slope in (0, 1]: LeakyReLU; 1: identity
          split1(f, hh[k][e], mm[k][e], ll[k][e]);
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int off = (4 * xv + k) * FX_RS + 8 * cg;
        *reinterpret_cast<u32x4*>(xt + 0 * FX_XPL + off) = (u32x4){(hh[k][0] >> 16) | hh[k][1], (hh[k][2] >> 16) | hh[k][3], (hh[k][4] >> 16) | hh[k][5], (hh[k][6] >> 16) | hh[k][7]};
        *reinterpret_cast<u32x4*>(xt + 1 * FX_XPL + off) = (u32x4){(mm[k][0] >> 16) | mm[k][1], (mm[k][2] >> 16) | mm[k][3], (mm[k][4] >> 16) | mm[k][5], (mm[k][6] >> 16) | mm[k][7]};
        *reinterpret_cast<u32x4*>(xt + 2 * FX_XPL + off) = (u32x4){(ll[k][0] >> 16) | ll[k][1], (ll[k][2] >> 16) | ll[k][3], (ll[k][4] >> 16) | ll[k][5], (ll[k][6] >> 16) | ll[k][7]};
      }
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = tid + i * 256;
      if (e < WNV) *reinterpret_cast<u32x4*>(ws + (e >> 2) * FX_RS + 8 * (e & 3)) = wr[i];
    }
    PROF(3)
    __syncthreads();
    PROF(4)
    if (c + 1 < nchunk) issue(c + 1);
    PROF(5)

    // ---- one k-block (32 channels) per tap: A = x'^T rows (steps), B = weights (output channels); lane (ln, g): k = 8g .. 8g + 7
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      bf16x8 af[2][3];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)                 // step t = n0 + 32w + 16tt + ln needs position t + j - 1 = window row 32w + 16tt + ln + j + 3
          af[tt][pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xt + pc * FX_XPL + (32 * wave + 16 * tt + ln + j + 3) * FX_RS + 8 * g));
#pragma unroll
      for (int ct = 0; ct < CO_TILES; ++ct) {
        bf16x8 bf[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          bf[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ws + pc * WPL + ((16 * ct + ln) * 3 + j) * FX_RS + 8 * g));
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {              // smallest products first
          f32x4 cc = acc[tt][ct];
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt][2], bf[0], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt][0], bf[2], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt][1], bf[1], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt][1], bf[0], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt][0], bf[1], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tt][0], bf[0], cc, 0, 0, 0);
          acc[tt][ct] = cc;
        }
      }
    }
    PROF(6)
  }

  // ---- epilogue: D[t = 4g + r][co = ln] -> 4 consecutive steps of one output channel per lane
#pragma unroll
  for (int ct = 0; ct < CO_TILES; ++ct) {
    const int co = r0 + 16 * ct + ln;
    const float bias = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t0 = n0 + 32 * wave + 16 * tt + 4 * g;
      if (co < p.Cout && t0 < T) {
        f32x4 v = acc[tt][ct];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += bias;
        *reinterpret_cast<f32x4*>(p.y + (long)b * p.y_bs + (long)co * T + t0) = v;
      }
    }
  }
  PROF(8)
  PROF_END
}

// fp32 weights [Cout][Cin][3] -> [piece][Cout][tap][FX_CP] bf16 pieces (zero for channels >= Cin)
__global__ __launch_bounds__(256) void conv_x6_weight_planes_kernel(const float* w, int Cout, int Cin, unsigned short* planes) {
  const int n = Cout * 3 * FX_CP;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int ci = i % FX_CP, rest = i / FX_CP, j = rest % 3, co = rest / 3;
    unsigned h = 0, m = 0, l = 0;
    if (ci < Cin) split1(w[((long)co * Cin + ci) * 3 + j], h, m, l);
    planes[i] = (unsigned short)(h >> 16); planes[n + i] = (unsigned short)(m >> 16); planes[2 * n + i] = (unsigned short)(l >> 16);
  }
}

}  // namespace tdvc

using namespace tdvc;

extern "C" size_t tdvc_conv_x6_weight_planes_bytes(int32_t Cout, int32_t Cin, int32_t K) {
  if (Cout <= 0 || Cin <= 0 || Cin > FX_CP || K != 3) return 0;
  return (size_t)3 * Cout * 3 * FX_CP * sizeof(unsigned short);
}

extern "C" int tdvc_conv_x6_weight_planes(const float* w, int32_t Cout, int32_t Cin, int32_t K, void* planes, void* stream) {
  if (!w || !planes || tdvc_conv_x6_weight_planes_bytes(Cout, Cin, K) == 0) return tdvc_fail(TDVC_EINVAL, "conv_x6_weight_planes: bad arguments");
  hipLaunchKernelGGL(conv_x6_weight_planes_kernel, dim3(tdvc_grid((long)Cout * 3 * FX_CP, 256, 1024)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin,
                     (unsigned short*)planes);
  TDVC_CHECK_LAUNCH();
  return TDVC_OK;
}

extern "C" int tdvc_conv_fwd_x6(const tdvc_conv_desc* d, const tdvc_conv_fwd_args* a, const void* weight_planes, void* stream) {
  if (!d || !a || !a->x || !a->y || !weight_planes) return tdvc_fail(TDVC_EINVAL, "conv_fwd_x6: null pointer");
  const bool shape = d->kind == TDVC_CONV && d->stride == 1 && d->groups == 1 && d->K == 3 && d->dilation == 1 && d->pad == 1 && !d->reflect &&
                     d->Tin == d->Tout && d->w_cin == 0 && d->Cin > 64 && d->Cin <= FX_CP && d->Cout >= 32 && d->Cout % 32 == 0 && d->Tin >= FX_NT &&
                     (d->Tin & 3) == 0 && (long)d->Cin * d->Tin < (1L << 29) && d->B > 0 && d->B < 65536;
  const bool plain = a->x_xf.kind <= TDVC_XF_LRELU && (a->x_xf.scale == 0.f || a->x_xf.scale == 1.f) && !a->res && !a->add && !a->bias3 && !a->sign_bits &&
                     a->post_act == TDVC_POST_NONE && (a->out_scale == 0.f || a->out_scale == 1.f) &&
                     (a->x_xf.kind == TDVC_XF_NONE || (a->x_xf.slope > 0.f && a->x_xf.slope <= 1.f));
  const bool al = ((((uintptr_t)a->x) | ((uintptr_t)a->y) | ((uintptr_t)weight_planes)) & 15) == 0 && (a->x_bs & 3) == 0 && (a->y_bs & 3) == 0;
  if (g_knob[6] || g_force_tile >= 0 || g_force_generic || !shape || !plain || !al) return tdvc_fail(TDVC_EUNSUPPORTED, "conv_fwd_x6: outside the split-bf16 forward kernel's contract");
  FwdX6P p = {};
  p.x = a->x; p.x_bs = a->x_bs; p.wp = (const unsigned short*)weight_planes; p.bias = a->bias; p.y = a->y; p.y_bs = a->y_bs;
  p.T = d->Tin; p.Cin = d->Cin; p.Cout = d->Cout; p.slope = a->x_xf.kind == TDVC_XF_LRELU ? a->x_xf.slope : 1.f;
  hipStream_t st = (hipStream_t)stream;
  const int nt = (d->Tin + FX_NT - 1) / FX_NT;
  if (d->Cout % 64 == 0) {
    auto k = conv_fwd_x6_kernel<4>;
    TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
    hipLaunchKernelGGL(k, dim3(nt, d->Cout / 64, d->B), dim3(256), (size_t)(3 * FX_XPL + 3 * 64 * 3 * FX_RS) * 2, st, p);
  } else {
    auto k = conv_fwd_x6_kernel<2>;
    TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
    hipLaunchKernelGGL(k, dim3(nt, d->Cout / 32, d->B), dim3(256), (size_t)(3 * FX_XPL + 3 * 32 * 3 * FX_RS) * 2, st, p);
  }
  TDVC_CHECK_LAUNCH();
  return TDVC_OK;
}
