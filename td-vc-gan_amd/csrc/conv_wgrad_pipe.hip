// Pipelined register-tile weight-gradient kernel for wide stride-1 layers (R >= 32 and Cin >= 32, aligned rows):
//   dW[co][ci][j] = sum_{b,t} dy'[co][t] * x'[ci][t + j*D - pad]
// The 4 waves form a 2x2 grid over a (32*M_REP) x (32*C_REP) tile of (co, ci), each wave keeps its sub-tile for all
// J taps in registers, and the block walks `tpb` consecutive (sample, 64-step) chunks.
//
// Software pipeline, one barrier per chunk (cdna_hip_programming.md T14 + LDS double buffering):
//   * LDS holds two stages of the operand tiles; while the MFMAs of chunk q read stage q&1, the same wave writes
//     chunk q+1 (already in registers) into the other stage and then issues the loads of chunk q+2 into those
//     registers. The stores and loads are spread over the 8 unrolled MFMA sub-steps of the chunk, so they issue in
//     the shadow of the wave's own matrix instructions instead of in a separate staging phase.
//   * staging is a row walk through raw buffer descriptors (conv_common.h): thread = (row of the pass, float4
//     column); rows past the tensor and columns outside [0, T) carry an out-of-range offset and load as zero, so the
//     chunks at the sequence ends take the same path. Reflect padding is loaded element-wise into the same
//     registers by the few lanes that own halo columns of an end chunk.
//   * bias-gradient row sums are accumulated from the registers on their way into LDS.
// x' prologues: none / LeakyReLU, or FiLM (h*(1+gamma)+beta -> LeakyReLU, three streams) for the 1x1 posconv.
#include "conv_common.h"
#include "conv_wgrad_lean.h"

PROF_DEFINE(tdvc_debug_wgrad_prof)

namespace tdvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WP_NTC = 64;
constexpr int WP_AS = 66;      // 2 (mod 4): conflict-free 16-lane fragment reads, 8-byte aligned rows
template <int J, int D> struct WpGeom {
  // row stride of the x' stage: >= span + 2 for every padding the host can ask for, 2 (mod 4)
  static constexpr int XSW = (J == 1) ? 66 : ((WP_NTC + (J - 1) * D + 8 + 3) / 4) * 4 + 2;
};

template <int M_REP, int C_REP, int J, int D, bool XFILM>
__global__ __launch_bounds__(256, 2) void conv_wgrad_pipe_kernel(const WgLeanP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MT = 32 * M_REP, CT = 32 * C_REP, XSW = WpGeom<J, D>::XSW;
  constexpr int AP = MT / 16;                  // dy' passes: 16 rows x 16 float4 each
  constexpr int XPM = XFILM ? 4 : 4 * C_REP;   // max x' passes (rows per pass >= 8 since span <= 128; 16 for the 1x1 FiLM case)
  constexpr int SF = MT * WP_AS + CT * XSW;    // floats per stage
  constexpr int NE = AP + XPM;                 // staged elements per thread and chunk

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, kq = lane >> 4;
  const int wm = wave >> 1, wc = wave & 1;
  const int ctiles = (p.Cin + CT - 1) / CT;
  // integer divisions go through the vector ALU: pin the results back to SGPRs so the buffer descriptors built from
  // them are provably wave-uniform (no waterfall loops around the loads -- cdna_hip_programming.md T20)
  const int mt = __builtin_amdgcn_readfirstlane(blockIdx.y / ctiles);
  const int ct = blockIdx.y - mt * ctiles;
  const int r0 = mt * MT, c0 = ct * CT;
  const int T = p.x.T;
  PROF_DECL

  f32x4 acc[M_REP][C_REP][J];
#pragma unroll
  for (int m = 0; m < M_REP; ++m)
#pragma unroll
    for (int c = 0; c < C_REP; ++c)
#pragma unroll
      for (int j = 0; j < J; ++j) acc[m][c][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bias_acc[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) bias_acc[i] = 0.f;
  const bool want_bias = p.bias_off >= 0 && ct == 0;

  const int nchunks = p.ntiles * p.B;
  const int q_begin = blockIdx.x * p.tpb, q_end = min(nchunks, q_begin + p.tpb);
  if (q_begin >= q_end) return;                // uniform per block

  // ---- thread roles of the two row walks
  const int arow = tid >> 4, avv = tid & 15;
  const int a_lds = arow * WP_AS + 4 * avv;
  const int xnv = p.span >> 2;
  const int xrow = (int)(((float)tid + 0.5f) * (1.0f / (float)xnv));
  const int xvv = tid - xrow * xnv;
  const bool xact = xrow < p.xrp;
  const int x_lds = xrow * XSW + 4 * xvv;
  const int x_lstep = p.xrp * XSW;
  const int x_gstep = p.xrp * T * 4;
  const float a_slope = p.a.xf.kind == XF_LRELU ? p.a.xf.slope : 1.f, a_scale = p.a.xf.scale;
  const float x_slope = p.x.xf.kind == XF_NONE ? 1.f : p.x.xf.slope, x_scale = p.x.xf.scale;

  f32x4 ar[AP], xr[XPM], gr[XFILM ? XPM : 1], br[XFILM ? XPM : 1];

  // chunk being issued next: (sample ib, tile it)
  // (made provably wave-uniform: the descriptors below must live in SGPRs, not behind a waterfall loop -- T20)
  int ib = __builtin_amdgcn_readfirstlane(q_begin / p.ntiles);
  int it = __builtin_amdgcn_readfirstlane(q_begin - ib * p.ntiles);
  srd_t a_rs, x_rs, g_rs, b_rs;
  int a_vo = 0, x_vo = 0;
  // Per chunk: descriptors and the thread's first offsets. Called once per chunk before the per-element loads.
  auto issue_setup = [&]() {
    const int nc0 = it * WP_NTC;
    a_rs = make_srd(p.a.p + (long)ib * p.a.bs + (long)r0 * p.N, (p.R - r0) * p.N * 4);
    x_rs = make_srd(p.x.p + (long)ib * p.x.bs + (long)c0 * T, (p.Cin - c0) * T * 4);
    if (XFILM) {
      const float* gb = p.x.xf.aux + (long)ib * p.x.xf.aux_bs;
      g_rs = make_srd(gb + (long)c0 * T, (p.Cin - c0) * T * 4);
      b_rs = make_srd(gb + (long)(p.Cin + c0) * T, (p.Cin - c0) * T * 4);
    }
    const int na = nc0 + 4 * avv;
    a_vo = na < p.N ? (arow * p.N + na) * 4 : 0x7f000000;
    const int qx = nc0 + p.lo + 4 * xvv;
    x_vo = (xact && qx >= 0 && qx < T) ? (xrow * T + qx) * 4 : 0x7f000000;
  };
  auto issue_elem = [&](int e) {               // e: compile-time constant after unrolling
    if (e < AP) {
      ar[e] = buf_load4(a_rs, a_vo + e * 16 * p.N * 4);
    } else {
      const int i = e - AP;
      if (i < walk_opaque(p.xnp)) {
        const int vo = x_vo + i * x_gstep;
        xr[i] = buf_load4(x_rs, vo);
        if (XFILM) { gr[i] = buf_load4(g_rs, vo); br[i] = buf_load4(b_rs, vo); }
      }
    }
  };
  // reflect padding: the lanes that own halo float4 columns of an end chunk fetch them element-wise
  auto issue_reflect = [&]() {
    if (XFILM || !p.reflect) return;
    const int q0 = it * WP_NTC + p.lo;
    if (q0 >= 0 && q0 + p.span <= T) return;   // uniform: interior chunk
    const int qx = q0 + 4 * xvv;
    if (!xact || (qx >= 0 && qx < T)) return;
    const float* xb = p.x.p + (long)ib * p.x.bs + (long)c0 * T;
#pragma unroll
    for (int i = 0; i < XPM; ++i) {
      if (i < walk_opaque(p.xnp)) {
        const int r = i * p.xrp + xrow;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c0 + r < p.Cin) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            int qq = qx + k;
            qq = qq < 0 ? -qq : 2 * (T - 1) - qq;
            if (qq >= 0 && qq < T) v[k] = xb[(long)r * T + qq];
          }
        }
        xr[i] = v;
      }
    }
  };
  auto issue_advance = [&]() { if (++it == p.ntiles) { it = 0; ++ib; } };

  float bias_w = 1.f;                          // 0 for the repeated commit past the block's last chunk
  auto commit_elem = [&](int e, float* stage) {
    if (e < AP) {
      f32x4 v = ar[e];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (v[k] > 0.f ? v[k] : v[k] * a_slope) * a_scale;
      if (want_bias) bias_acc[e] += bias_w * ((v[0] + v[1]) + (v[2] + v[3]));
      f32x2* d = reinterpret_cast<f32x2*>(stage + a_lds + e * 16 * WP_AS);
      d[0] = (f32x2){v[0], v[1]}; d[1] = (f32x2){v[2], v[3]};
    } else {
      const int i = e - AP;
      if (i < walk_opaque(p.xnp)) {
        f32x4 v = xr[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float h = v[k];
          if (XFILM) h = h * (1.f + gr[i][k]) + br[i][k];
          v[k] = (h > 0.f ? h : h * x_slope) * x_scale;
        }
        if (xact && i * p.xrp + xrow < CT) {     // the last pass may reach past the tile
          f32x2* d = reinterpret_cast<f32x2*>(stage + MT * WP_AS + x_lds + i * x_lstep);
          d[0] = (f32x2){v[0], v[1]}; d[1] = (f32x2){v[2], v[3]};
        }
      }
    }
  };

  // ---- prologue: chunk q_begin -> stage 0, chunk q_begin + 1 -> registers. qi = chunk the issue state points at.
  int qi = q_begin;
  issue_setup();
#pragma unroll
  for (int e = 0; e < NE; ++e) issue_elem(e);
  issue_reflect();
#pragma unroll
  for (int e = 0; e < NE; ++e) commit_elem(e, smem);
  if (qi + 1 < q_end) { issue_advance(); ++qi; }
  issue_setup();
#pragma unroll
  for (int e = 0; e < NE; ++e) issue_elem(e);
  issue_reflect();
  if (qi + 1 < q_end) { issue_advance(); ++qi; }
  __syncthreads();
  PROF(0)
  for (int q = q_begin; q < q_end; ++q) {
    const int cur = (q - q_begin) & 1;
    float* st_cur = smem + cur * SF;
    float* st_nxt = smem + (cur ^ 1) * SF;
    bias_w = (q + 1 < q_end) ? 1.f : 0.f;

    const float* a_lane = st_cur + (wm * 16 * M_REP + ln) * WP_AS + kq;
    const float* x_lane = st_cur + MT * WP_AS + (wc * 16 * C_REP + ln) * XSW + kq + p.i0;
    float av[2][M_REP], bv[2][C_REP][J];
    auto load_frag = [&](int buf, int nn) {
#pragma unroll
      for (int m = 0; m < M_REP; ++m) av[buf][m] = a_lane[nn + m * 16 * WP_AS];
#pragma unroll
      for (int c = 0; c < C_REP; ++c)
#pragma unroll
        for (int j = 0; j < J; ++j) bv[buf][c][j] = x_lane[nn + c * 16 * XSW + j * D];
    };
    auto mma = [&](int buf) {
#pragma unroll
      for (int m = 0; m < M_REP; ++m)
#pragma unroll
        for (int c = 0; c < C_REP; ++c)
#pragma unroll
          for (int j = 0; j < J; ++j)
            acc[m][c][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][m], bv[buf][c][j], acc[m][c][j], 0, 0, 0);
    };
    load_frag(0, 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int nn = u * 8;
      load_frag(1, nn + 4);
      mma(0);
      // Sub-steps 2..4: the registers (chunk q+1) go to the other stage; sub-steps 5..7: the same registers take the
      // loads of chunk q+2. Every store precedes every load and neither sits under a data-dependent branch, so the
      // compiler needs one vmcnt wait per chunk (before the first store) and the loads issue back to back. Past the
      // end of the block's range both are harmless repeats of its last chunk.
      if (u >= 2 && u <= 4) {
#pragma unroll
        for (int e = ((u - 2) * NE) / 3; e < ((u - 1) * NE) / 3; ++e) commit_elem(e, st_nxt);
      }
      if (u >= 5) {
        if (u == 5) issue_setup();
#pragma unroll
        for (int e = ((u - 5) * NE) / 3; e < ((u - 4) * NE) / 3; ++e) issue_elem(e);
        if (u == 7) { issue_reflect(); if (qi + 1 < q_end) { issue_advance(); ++qi; } }
      }
      if (u < 7) load_frag(0, nn + 8);
      mma(1);
    }
    PROF(5)
    __syncthreads();
    PROF(1)
  }

  // ---- the block's partial tile -> its slab (module weight layout [R][Cin][K])
  float* slab = p.slab + (long)blockIdx.x * p.slab_stride;
  const long rowlen = (long)p.Cin * p.K;
#pragma unroll
  for (int m = 0; m < M_REP; ++m)
#pragma unroll
    for (int c = 0; c < C_REP; ++c) {
      const int ci = c0 + (wc * C_REP + c) * 16 + ln;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + (wm * M_REP + m) * 16 + kq * 4 + r;
        if (row < p.R && ci < p.Cin) {
          float* dst = slab + row * rowlen + (long)ci * p.K;
#pragma unroll
          for (int j = 0; j < J; ++j) dst[j] = acc[m][c][j][r];
        }
      }
    }
  if (want_bias) {       // a row's 16 float4 columns sit in 16 consecutive lanes
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      float s = bias_acc[i];
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
      const int row = r0 + i * 16 + arow;
      if (avv == 0 && row < p.R) slab[p.bias_off + row] = s;
    }
  }
  PROF(6)
  PROF_END
}

template <int M_REP, int C_REP, int J, int D, bool XFILM>
static hipError_t wp_launch2(const WgLeanP& p, hipStream_t st) {
  constexpr int MT = 32 * M_REP, CT = 32 * C_REP, XSW = WpGeom<J, D>::XSW;
  auto k = conv_wgrad_pipe_kernel<M_REP, C_REP, J, D, XFILM>;
  TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
  if (p.span + 2 > XSW || p.xnp > (XFILM ? 4 : 4 * C_REP)) return hipErrorNotSupported;
  dim3 grid(p.ngroups, ((p.R + MT - 1) / MT) * ((p.Cin + CT - 1) / CT), 1);
  const size_t lds = (size_t)2 * (MT * WP_AS + CT * XSW) * sizeof(float);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  return hipGetLastError();
}

template <int M_REP, int C_REP, int J, int D>
static hipError_t wp_launch(WgLeanP& p, hipStream_t st) {
  constexpr int CT = 32 * C_REP;
  const int nvec = p.span >> 2;
  p.xrp = 256 / nvec; if (p.xrp > CT) p.xrp = CT;
  p.xnp = (CT + p.xrp - 1) / p.xrp;
  if (p.x.xf.kind == XF_FILM_LRELU) {
    if (J != 1) return hipErrorNotSupported;
    return wp_launch2<M_REP, C_REP, 1, 1, true>(p, st);
  }
  return wp_launch2<M_REP, C_REP, J, D, false>(p, st);
}

template <int J, int D>
static hipError_t wp_launch_jd(WgLeanP& p, hipStream_t st) {
  p.span = ((WP_NTC + (J - 1) * D - p.pad - p.lo) + 3) / 4 * 4;
  const bool ct64 = J <= 3 && !wgrad_prefers_ct32(p.Cin);   // 64 input channels per block tile unless that pads too much (136)
  if (p.R <= 32 || J >= 11) {                             // (11 taps: the 64-row tile would spill its accumulators)
    if (ct64) return wp_launch<1, 2, J, D>(p, st);       // 32 x 64 block tile, 16 x 32 per wave
    return wp_launch<1, 1, J, D>(p, st);                  // 32 x 32 block tile
  }
  if (ct64) return wp_launch<2, 2, J, D>(p, st);         // 64 x 64 block tile, 32 x 32 per wave
  return wp_launch<2, 1, J, D>(p, st);                    // 64 x 32 block tile, 32 x 16 per wave
}

// Contract: p.vec (16-byte aligned rows, N % 4 == 0), dy' prologue none / LeakyReLU, x' prologue none / LeakyReLU /
// FiLM (1x1 only). p.lo, p.i0, p.ntiles, p.tpb, p.ngroups, p.B are set by the caller (conv_wgrad_lean.hip).
hipError_t launch_conv_wgrad_pipe(WgLeanP& p, int J, int D, hipStream_t st) {
  if (!p.vec || p.a.xf.kind > XF_LRELU) return hipErrorNotSupported;
  if (p.x.xf.kind > XF_LRELU && p.x.xf.kind != XF_FILM_LRELU) return hipErrorNotSupported;
  if (p.x.xf.kind == XF_FILM_LRELU && (p.x.xf.aux == nullptr || (((uintptr_t)p.x.xf.aux) & 15) || (p.x.xf.aux_bs & 3))) return hipErrorNotSupported;
#define WP_CASE(JJ, DD) if (J == JJ && D == DD) return wp_launch_jd<JJ, DD>(p, st);
  WP_CASE(1, 1) WP_CASE(5, 1)
  WP_CASE(3, 1) WP_CASE(3, 3) WP_CASE(3, 5)
  WP_CASE(7, 1) WP_CASE(7, 3) WP_CASE(7, 5)
  WP_CASE(11, 1) WP_CASE(11, 3) WP_CASE(11, 5)
#undef WP_CASE
  return hipErrorNotSupported;
}

}  // namespace tdvc
