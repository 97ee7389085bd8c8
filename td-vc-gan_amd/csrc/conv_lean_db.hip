// Software-pipelined main loop of the lean stride-1 conv kernel (plain / LeakyReLU prologue), for its MFMA-bound tiles.
//
// conv_lean.hip walks the channel chunks as  barrier -> commit(registers -> LDS) -> barrier -> issue next loads -> MFMA:
// while a wave commits or issues, the matrix pipe of its SIMD only has the co-resident blocks' waves to feed it, and the
// phase stamps (profiles/r02_a_phase_cycles.txt) show commit + issue + barriers at 25-35 % of a block's cycles with the
// MFMA pipe busy 37-54 % of the time (profiles/r02_a_pmc.txt). Here the staging rides in the shadow of the wave's OWN
// matrix instructions instead (an MFMA issues in 4 cycles and executes for 32: the wave's vector / LDS / memory
// instructions fit in between):
//   * LDS holds TWO stages of (input tile, weight tile). While the MFMAs of chunk c read stage c&1, the same wave stores
//     chunk c+1 (already in registers) into the other stage and then issues the global loads of chunk c+2 into those
//     registers, both right behind the first MFMA steps of the chunk;
//   * ONE barrier per chunk (end of the MFMA loop: everybody is done reading stage c&1 and done writing the other one).
// Everything else — row-walk staging through raw buffer descriptors, fragment double buffering, reflect halo patch of
// the end tiles, mirror fold, fused epilogues — is the code of conv_lean.hip (shared pieces: conv_lean_parts.h).
#include "conv_lean_parts.h"

namespace tdvc {

template <int M_REP, int N_REP, int WM>
constexpr int lean_db_min_blocks() { return M_REP * N_REP >= 12 ? 2 : 3; }

template <int M_REP, int N_REP, int WM, int WN, int EPI>
__global__ __launch_bounds__(256, (lean_db_min_blocks<M_REP, N_REP, WM>())) void conv_lean_db_kernel(const LeanP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MT = 16 * M_REP * WM, NT = 16 * N_REP * WN;
  constexpr int XVP = 12;                                 // max row-walk passes of the prefetched input tile
  constexpr int WVP = MT >= 48 ? 10 : 6;                  // ... and of the weight tile
  const int SFX = p.xnp * p.xrp * p.XS;                   // floats of one stage: input tile, then weight tile
  const int SF = (SFX + p.wnp * p.wrp * p.WS + 3) & ~3;   // stage stride, 16-byte aligned

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ln = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * NT, r0 = blockIdx.y * MT, b = blockIdx.z;
  const int wcol0 = wn * 16 * N_REP, wrow0 = wm * 16 * M_REP;

  f32x4 acc[M_REP][N_REP];
#pragma unroll
  for (int m = 0; m < M_REP; ++m)
#pragma unroll
    for (int n = 0; n < N_REP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int q0 = n0 + p.lo;
  const bool tail = (p.T & 3) != 0;
  const bool end_reflect = p.reflect && (q0 < 0 || q0 + p.span > p.T);     // block-uniform
  const int jc = p.K * p.Cc;
  const float* xrow0 = p.x + (long)b * p.x_bs;
  const float* wgrow = p.w + (long)r0 * p.Cw;
  const RowWalk xw = make_walk(tid, p.span >> 2, p.xrp, p.T, p.XS, q0, p.T);
  const RowWalk ww = make_walk(tid, jc >> 2, p.wrp, p.Cw, p.WS, 0, 1 << 30);
  RegTile<XVP> xr;
  RegTile<WVP> wr;
  srd_t x_rs, w_rs;                                       // descriptors of the chunk whose loads are being issued
  auto set_srd = [&](int c0) __attribute__((always_inline)) {
    x_rs = make_srd(xrow0 + (long)c0 * p.T, (p.Cin - c0) * p.T * 4);
    w_rs = make_srd(wgrow + (long)c0 * p.K, ((p.Cout - r0) * p.Cw - c0 * p.K) * 4);
  };
  // one row-walk pass each; `i` is a compile-time constant at every call site
  auto x_issue1 = [&](int i) __attribute__((always_inline)) { xr.v[i] = buf_load4(x_rs, xw.voff + i * xw.gstep); };
  auto w_issue1 = [&](int i) __attribute__((always_inline)) { wr.v[i] = buf_load4(w_rs, ww.voff + i * ww.gstep); };
  auto x_commit1 = [&](int i, float* xs) __attribute__((always_inline)) {
    if (!xw.active) return;
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float v = xr.v[i][q]; o[q] = fmaxf(v, v * p.slope); }
    if (p.in_scale != 1.f) o *= p.in_scale;
    if (tail) {
#pragma unroll
      for (int q = 1; q < 4; ++q) o[q] = q < xw.nk ? o[q] : 0.f;
    }
    *reinterpret_cast<f32x4*>(xs + xw.loff + i * xw.lstep) = o;
  };
  auto w_commit1 = [&](int i, float* ws) __attribute__((always_inline)) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    if (!ww.active) return;
    f32x2_t* d2 = reinterpret_cast<f32x2_t*>(ws + ww.loff + i * ww.lstep);
    d2[0] = (f32x2_t){wr.v[i][0], wr.v[i][1]};
    d2[1] = (f32x2_t){wr.v[i][2], wr.v[i][3]};
  };
  // reflect halo of the first / last time tile (forward trunk convs): columns with q < 0 or q >= T, element-wise
  auto reflect_patch = [&](float* xs, int c0) {
    const int cvalid = min(p.Cc, p.Cin - c0);
    const int nl = q0 < 0 ? -q0 : 0;
    const int rfirst = p.T - q0;
    const int nr = rfirst < p.span ? p.span - rfirst : 0;
    const int nh = nl + nr;
    const float* xc = xrow0 + (long)c0 * p.T;
    const float inv = 1.0f / (float)nh;
    for (int e = tid; e < cvalid * nh; e += 256) {
      const int r = (int)(((float)e + 0.5f) * inv);
      const int h = e - r * nh;
      const int i = h < nl ? h : rfirst + (h - nl);
      int q = q0 + i;
      q = q < 0 ? -q : 2 * (p.T - 1) - q;
      float v = (q >= 0 && q < p.T) ? xc[(long)r * p.T + q] : 0.f;
      v = (v > 0.f ? v : v * p.slope) * p.in_scale;
      xs[r * p.XS + i] = v;
    }
  };

  // ---- prologue: chunk 0 -> stage 0, chunk 1 -> registers
  set_srd(0);
#pragma unroll
  for (int i = 0; i < XVP; ++i) if (i < walk_opaque(p.xnp)) x_issue1(i);
#pragma unroll
  for (int i = 0; i < WVP; ++i) if (i < walk_opaque(p.wnp)) w_issue1(i);
#pragma unroll
  for (int i = 0; i < XVP; ++i) if (i < walk_opaque(p.xnp)) x_commit1(i, smem);
#pragma unroll
  for (int i = 0; i < WVP; ++i) if (i < walk_opaque(p.wnp)) w_commit1(i, smem + SFX);
  if (p.Cc < p.Cin) {
    set_srd(p.Cc);
#pragma unroll
    for (int i = 0; i < XVP; ++i) if (i < walk_opaque(p.xnp)) x_issue1(i);
#pragma unroll
    for (int i = 0; i < WVP; ++i) if (i < walk_opaque(p.wnp)) w_issue1(i);
  }
  if (end_reflect) { __syncthreads(); reflect_patch(smem, 0); }
  __syncthreads();

  bool needL = false, needR = false;
  if (p.mirror > 0) {
    const int c_lo = n0 + wcol0, c_hi = c_lo + 16 * N_REP - 1;
    needL = (c_lo <= p.mirror) && (c_hi >= 1);
    needR = (c_lo <= p.T - 2) && (c_hi >= p.T - 1 - p.mirror);
  }

  int cur = 0;
  for (int c0 = 0; c0 < p.Cin; c0 += p.Cc, cur ^= 1) {
    const int cvalid = min(p.Cc, p.Cin - c0);
    float* xs = smem + cur * SF;
    float* ws = xs + SFX;
    float* xs_n = smem + (cur ^ 1) * SF;
    float* ws_n = xs_n + SFX;
    const bool have1 = c0 + p.Cc < p.Cin, have2 = c0 + 2 * p.Cc < p.Cin;
    if (have2) set_srd(c0 + 2 * p.Cc);
    // Side work of this chunk: commit chunk c+1 right behind the first pair of MFMA steps (its vector / LDS-store
    // instructions issue while those 2 x M_REP*N_REP MFMAs execute), issue the loads of chunk c+2 behind the second pair.
    // Both are straight-line code with compile-time register indices: finer-grained variants (one pass per MFMA pair
    // through a switch, five batches keyed on a counter) made the compiler move the register tiles to scratch memory.
    auto side_commit = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < XVP; ++i) if (i < walk_opaque(p.xnp)) x_commit1(i, xs_n);
#pragma unroll
      for (int i = 0; i < WVP; ++i) if (i < walk_opaque(p.wnp)) w_commit1(i, ws_n);
    };
    auto side_issue = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < XVP; ++i) if (i < walk_opaque(p.xnp)) x_issue1(i);
#pragma unroll
      for (int i = 0; i < WVP; ++i) if (i < walk_opaque(p.wnp)) w_issue1(i);
    };

    // MFMA: D[t][co] += X'[t][k] * W[k][co]; one step = 4 channels of one tap (see conv_lean.hip)
    const int csteps = cvalid >> 2;
    const int nsteps = p.K * csteps;
    const float* w_lane = ws + (wrow0 + ln) * p.WS + kq * p.K;
    const float* x_lane = xs + kq * p.XS + wcol0 + ln + p.i0;
    const int wrep = 16 * p.WS;
    int sj = 0, scs = 0;
    int woff = p.flip ? p.K - 1 : 0, xoff = 0;
    auto advance = [&]() __attribute__((always_inline)) {
      if (++scs == csteps) { scs = 0; ++sj; woff = p.flip ? p.K - 1 - sj : sj; xoff = sj * p.d; }
      else { woff += 4 * p.K; xoff += 4 * p.XS; }
    };
    float wv[2][M_REP], xv[2][N_REP];
    auto load_frag = [&](int buf) __attribute__((always_inline)) {
      const float* wp = w_lane + woff;
      const float* xp = x_lane + xoff;
#pragma unroll
      for (int m = 0; m < M_REP; ++m) wv[buf][m] = wp[m * wrep];
#pragma unroll
      for (int n = 0; n < N_REP; ++n) xv[buf][n] = xp[n * 16];
    };
    auto mma = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < M_REP; ++m)
#pragma unroll
        for (int n = 0; n < N_REP; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[buf][n], wv[buf][m], acc[m][n], 0, 0, 0);
    };
    load_frag(0);
    int s = 0;
    for (; s + 2 <= nsteps; s += 2) {
      advance(); load_frag(1);
      mma(0);
      if (s + 2 < nsteps) { advance(); load_frag(0); }
      mma(1);
      if (s == 0 && have1) side_commit();
      if (s == 2 && have2) side_issue();
    }
    if (s < nsteps) mma(0);
    if (nsteps < 2 && have1) side_commit();       // chunks too short for the in-loop slots
    if (nsteps < 4 && have2) side_issue();

    lean_mirror_fold<M_REP, N_REP>(p, acc, xs, w_lane, wrep, csteps, needL, needR, n0, wcol0, ln, kq);
    if (end_reflect && have1) { __syncthreads(); reflect_patch(xs_n, c0 + p.Cc); }
    __syncthreads();
  }

  lean_epilogue<M_REP, N_REP, EPI>(p, acc, b, n0, r0, wcol0, wrow0, ln, kq, p.vec != 0);
}

// ------------------------------------------------------------------------------------------ host
template <int M_REP, int N_REP, int WM, int WN, int EPI>
static hipError_t lean_db_launch3(const LeanP& p, int B, hipStream_t st) {
  constexpr int MT = 16 * M_REP * WM, NT = 16 * N_REP * WN;
  auto k = conv_lean_db_kernel<M_REP, N_REP, WM, WN, EPI>;
  TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
  dim3 grid((p.T + NT - 1) / NT, (p.Cout + MT - 1) / MT, B);
  const size_t lds = (size_t)2 * ((p.xnp * p.xrp * p.XS + p.wnp * p.wrp * p.WS + 3) & ~3) * sizeof(float);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  return hipGetLastError();
}

template <int M_REP, int N_REP, int WM, int WN>
static hipError_t lean_db_launch2(const LeanP& p, int B, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_FWD: return lean_db_launch3<M_REP, N_REP, WM, WN, EPI_FWD>(p, B, st);
    case EPI_MASK: return lean_db_launch3<M_REP, N_REP, WM, WN, EPI_MASK>(p, B, st);
    case EPI_FILM: return lean_db_launch3<M_REP, N_REP, WM, WN, EPI_FILM>(p, B, st);
    default: return lean_db_launch3<M_REP, N_REP, WM, WN, EPI_PLAIN>(p, B, st);
  }
}

// cfg: tile configuration of launch_conv_lean (1: 32x256, 2: 64x256, 4: 64x64, 5: 48x256); plain prologue only
hipError_t launch_conv_lean_db(const LeanP& p, int B, int cfg, int epi, hipStream_t st) {
  switch (cfg) {
    case 1: return lean_db_launch2<2, 4, 1, 4>(p, B, epi, st);
    case 2: return lean_db_launch2<4, 4, 1, 4>(p, B, epi, st);
    case 4: return lean_db_launch2<1, 4, 4, 1>(p, B, epi, st);
    case 5: return lean_db_launch2<3, 4, 1, 4>(p, B, epi, st);
    default: return hipErrorNotSupported;
  }
}

}  // namespace tdvc
