// Pieces of the lean conv kernel shared by its two main-loop variants (conv_lean.hip: single LDS stage, two barriers
// per channel chunk; conv_lean_db.hip: two LDS stages, one barrier, staging interleaved with the MFMAs):
// the reflect-pad mirror fold of the input-gradient and the fused epilogue.
#pragma once
#include "conv_common.h"
#include "conv_lean.h"

namespace tdvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Reflect-pad fold (input-grad of a reflect-padded conv): the columns next to the sequence ends also collect the
// contributions of the mirrored (padded) positions. Edge waves only; reads the staged chunk in `xs` / `w_lane`.
template <int M_REP, int N_REP>
__device__ __forceinline__ void lean_mirror_fold(const LeanP& p, f32x4 (&acc)[M_REP][N_REP], const float* xs, const float* w_lane, int wrep,
                                                 int csteps, bool needL, bool needR, int n0, int wcol0, int ln, int kq) {
    if (M_REP >= 2 && (needL || needR)) {   // reflect-pad fold (input-grad of a reflect conv): edge waves only
      // Per side, only the 16-column sub-tiles that hold mirrored columns take part (pad <= 25: one or two of them). Each
      // runs a software-pipelined (tap, channel-group) loop into a temporary accumulator that is then added to the
      // sub-tile's own accumulators, so the extra work of the blocks at the sequence ends stays a fraction of a chunk.
      for (int side = 0; side < 2; ++side) {
        if (side == 0 ? !needL : !needR) continue;
#pragma unroll
        for (int n = 0; n < N_REP; ++n) {
          const int ca = n0 + wcol0 + n * 16, cb = ca + 15;
          const int use = __builtin_amdgcn_readfirstlane(side == 0 ? (ca <= p.mirror && cb >= 1) : (ca <= p.T - 2 && cb >= p.T - 1 - p.mirror));
          if (!use) continue;
          const int u = ca + ln;
          bool mv; int mb;
          if (side == 0) { mv = (u >= 1 && u <= p.mirror && u < p.T); mb = -u - n0 + p.i0; }
          else { mv = (u >= p.T - 1 - p.mirror && u <= p.T - 2 && u >= 0); mb = 2 * (p.T - 1) - u - n0 + p.i0; }
          f32x4 tacc[M_REP];
#pragma unroll
          for (int m = 0; m < M_REP; ++m) tacc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
          const int nit = p.K * csteps;
          float wq[2][M_REP], xq[2];
          int mj = 0, mcs = 0;
          auto mload = [&](int buf) {
            const float* wj = w_lane + (p.flip ? p.K - 1 - mj : mj) + mcs * 4 * p.K;
#pragma unroll
            for (int m = 0; m < M_REP; ++m) wq[buf][m] = wj[m * wrep];
            const int idx = mb + mj * p.d;
            const bool ok = mv && idx >= 0 && idx < p.span;
            const float t = xs[(mcs * 4 + kq) * p.XS + (ok ? idx : 0)];
            xq[buf] = ok ? t : 0.f;
            if (++mcs == csteps) { mcs = 0; ++mj; }
          };
          mload(0);
          for (int it = 0; it < nit; it += 2) {
            if (it + 1 < nit) mload(1);
#pragma unroll
            for (int m = 0; m < M_REP; ++m) tacc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[0], wq[0][m], tacc[m], 0, 0, 0);
            if (it + 1 < nit) {
              if (it + 2 < nit) mload(0);
#pragma unroll
              for (int m = 0; m < M_REP; ++m) tacc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[1], wq[1][m], tacc[m], 0, 0, 0);
            }
          }
#pragma unroll
          for (int m = 0; m < M_REP; ++m) acc[m][n] += tacc[m];
        }
      }
    }
    if (M_REP == 1 && (needL || needR)) {   // reflect-pad fold, 16-row tiles: plain loop (one MFMA per step, nothing to pipeline)
      for (int side = 0; side < 2; ++side) {
        if (side == 0 ? !needL : !needR) continue;
        int mb[N_REP]; bool mv[N_REP];
#pragma unroll
        for (int n = 0; n < N_REP; ++n) {
          const int u = n0 + wcol0 + n * 16 + ln;
          if (side == 0) { mv[n] = (u >= 1 && u <= p.mirror && u < p.T); mb[n] = -u - n0 + p.i0; }
          else { mv[n] = (u >= p.T - 1 - p.mirror && u <= p.T - 2 && u >= 0); mb[n] = 2 * (p.T - 1) - u - n0 + p.i0; }
        }
        for (int j = 0; j < p.K; ++j) {
          const float* wj = w_lane + (p.flip ? p.K - 1 - j : j);
          for (int cs = 0; cs < csteps; ++cs) {
            float wm_[M_REP], xm_[N_REP];
#pragma unroll
            for (int m = 0; m < M_REP; ++m) wm_[m] = wj[m * 16 * p.WS + cs * 4 * p.K];
#pragma unroll
            for (int n = 0; n < N_REP; ++n) {
              const int idx = mb[n] + j * p.d;
              const bool ok = mv[n] && idx >= 0 && idx < p.span;
              const float t = xs[(cs * 4 + kq) * p.XS + (ok ? idx : 0)];
              xm_[n] = ok ? t : 0.f;
            }
#pragma unroll
            for (int m = 0; m < M_REP; ++m)
#pragma unroll
              for (int n = 0; n < N_REP; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xm_[n], wm_[m], acc[m][n], 0, 0, 0);
          }
        }
      }
    }
}

// Epilogue: lane owns channel co = .. + ln and time steps t0 .. t0+3 (vector path) or single steps (short rows).
template <int M_REP, int N_REP, int EPI>
__device__ __forceinline__ void lean_epilogue(const LeanP& p, f32x4 (&acc)[M_REP][N_REP], int b, int n0, int r0, int wcol0, int wrow0, int ln,
                                              int kq, bool vec_ok) {
  // ---- epilogue: lane owns channel co = .. + ln and time steps t0 .. t0+3 (t0 % 4 == 0, T % 4 == 0)
  if (!vec_ok) {   // short, unaligned sequences (T = 50, 63): scalar epilogue
#pragma unroll
    for (int m = 0; m < M_REP; ++m) {
      const int co = r0 + wrow0 + m * 16 + ln;
      if (co >= p.Cout) continue;
      const long ro = (long)co * p.T;
      const float bias = (EPI == EPI_FWD && p.bias) ? p.bias[co] : 0.f;
#pragma unroll
      for (int n = 0; n < N_REP; ++n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = n0 + wcol0 + n * 16 + kq * 4 + q;
          if (t >= p.T) continue;
          const long oi = ro + t;
          float v = acc[m][n][q];
          if (EPI == EPI_FWD) {
            v += bias;
            if (p.bias3) v += p.bias3[((long)b * p.Cout + co) * 3 + (t == 0 ? 0 : (t == p.T - 1 ? 2 : 1))];
            if (p.res) v += p.res[(long)b * p.res_bs + oi];
            if (p.post == POST_LRELU) v = fmaxf(v, v * p.m_slope);
            else if (p.post == POST_TANH) v = tanhf(v);
            v *= p.out_scale;
          } else if (EPI == EPI_MASK) {
            v = p.mx[(long)b * p.mx_bs + oi] > 0.f ? v : v * p.m_slope;
          } else if (EPI == EPI_FILM) {
            const float h = p.mx[(long)b * p.mx_bs + oi];
            const float* gp = p.gb + (long)b * p.gb_bs + oi;
            const float ga = gp[0], be = gp[(long)p.Cout * p.T];
            const float dh2 = (h * (1.f + ga) + be) > 0.f ? v : v * p.m_slope;
            float* dg = p.dgb + (long)b * p.dgb_bs + oi;
            dg[0] = dh2 * h; dg[(long)p.Cout * p.T] = dh2;
            v = dh2 * (1.f + ga);
          }
          if (p.add) v += p.add_scale * p.add[(long)b * p.add_bs + oi];
          p.y[(long)b * p.y_bs + oi] = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < M_REP; ++m) {
    const int co = r0 + wrow0 + m * 16 + ln;
    if (co >= p.Cout) continue;
    const long ro = (long)co * p.T;
    float bias = 0.f;
    if (EPI == EPI_FWD && p.bias) bias = p.bias[co];
#pragma unroll
    for (int n = 0; n < N_REP; ++n) {
      const int t0 = n0 + wcol0 + n * 16 + kq * 4;
      if (t0 >= p.T) continue;
      const long oi = ro + t0;
      f32x4 v = acc[m][n];
      if (EPI == EPI_FWD) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += bias;
        if (p.bias3) {
          const float* k3 = p.bias3 + ((long)b * p.Cout + co) * 3;
          const float mid = k3[1];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += mid;
          if (t0 == 0) v[0] += k3[0] - mid;
          if (t0 + 4 == p.T) v[3] += k3[2] - mid;
        }
        if (p.res) { const f32x4 r = *reinterpret_cast<const f32x4*>(p.res + (long)b * p.res_bs + oi); v += r; }
        if (p.post == POST_LRELU) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], v[q] * p.m_slope);
        } else if (p.post == POST_TANH) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = tanhf(v[q]);
        }
        v *= p.out_scale;
      } else if (EPI == EPI_MASK) {
        const f32x4 mm = *reinterpret_cast<const f32x4*>(p.mx + (long)b * p.mx_bs + oi);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = mm[q] > 0.f ? v[q] : v[q] * p.m_slope;
      } else if (EPI == EPI_FILM) {
        const f32x4 h = *reinterpret_cast<const f32x4*>(p.mx + (long)b * p.mx_bs + oi);
        const float* gp = p.gb + (long)b * p.gb_bs + oi;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gp);
        const f32x4 be = *reinterpret_cast<const f32x4*>(gp + (long)p.Cout * p.T);
        f32x4 dga, dbe;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float h2 = h[q] * (1.f + ga[q]) + be[q];
          const float dh2 = h2 > 0.f ? v[q] : v[q] * p.m_slope;
          dga[q] = dh2 * h[q]; dbe[q] = dh2; v[q] = dh2 * (1.f + ga[q]);
        }
        float* dg = p.dgb + (long)b * p.dgb_bs + oi;
        *reinterpret_cast<f32x4*>(dg) = dga;
        *reinterpret_cast<f32x4*>(dg + (long)p.Cout * p.T) = dbe;
      }
      if (p.add) { const f32x4 a4 = *reinterpret_cast<const f32x4*>(p.add + (long)b * p.add_bs + oi); v += a4 * p.add_scale; }
      *reinterpret_cast<f32x4*>(p.y + (long)b * p.y_bs + oi) = v;
    }
  }
}

}  // namespace tdvc
