// Lean weight-gradient kernel for stride-1, groups == 1 convolutions with long sequences (N > 128):
//   dW[co][ci][j] = sum_{b,t} dy'[co][t] * x'[ci][t + j*D - pad]
// Per block: a (16*M_REP) x (16*C_REP) tile of (co, ci) for all J taps over one 256-step time chunk of one
// sample; the 4 waves split the chunk (K-split) and are summed through LDS; partial tiles go to a slab that
// slab_reduce_kernel folds into dW (deterministic, no atomics on the weights).
// J (taps) and D (dilation) are template parameters and the LDS strides are constants, so every fragment
// address in the MFMA loop is base + immediate: one vector add per step. Fragments of step s+1 are fetched
// before the MFMAs of step s issue.
#include "conv_common.h"
#include "conv_wgrad_lean.h"

namespace tdvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));


constexpr int WG_NTC = 256;
constexpr int WG_AS = 258;     // 2 (mod 32)
constexpr int WG_XS = 322;     // 2 (mod 32), >= 256 + (11-1)*5 + alignment slack

template <int M_REP, int C_REP, int J, int D>
__global__ __launch_bounds__(256, (M_REP * C_REP * J >= 24 ? 2 : 3)) void conv_wgrad_lean_kernel(const WgLeanP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MT = 16 * M_REP, CT = 16 * C_REP;
  constexpr int E = M_REP * C_REP * J * 4;
  float* as = smem;                     // [MT][WG_AS]
  float* xs = smem + MT * WG_AS;        // [CT][WG_XS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, kq = lane >> 4;
  const int ctiles = (p.Cin + CT - 1) / CT;
  const int ct = blockIdx.y % ctiles, mt = blockIdx.y / ctiles;
  const int b = blockIdx.x / p.ngroups, grp = blockIdx.x % p.ngroups;
  const int r0 = mt * MT, c0 = ct * CT;

  f32x4 acc[M_REP][C_REP][J];
#pragma unroll
  for (int m = 0; m < M_REP; ++m)
#pragma unroll
    for (int c = 0; c < C_REP; ++c)
#pragma unroll
      for (int j = 0; j < J; ++j) acc[m][c][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float bias_part[MT / 16];   // thread (tid>>4) owns rows (tid>>4) + 16*i
#pragma unroll
  for (int i = 0; i < MT / 16; ++i) bias_part[i] = 0.f;
  const int tile_end = min(p.ntiles, (grp + 1) * p.tpb);
  for (int tile = grp * p.tpb; tile < tile_end; ++tile) {
  const int nc0 = tile * WG_NTC;
  __syncthreads();      // previous chunk's fragments fully consumed
  // ---- stage both operand tiles (vector path for interior chunks, padding logic at the sequence ends)
  const bool a_fast = p.vec && nc0 + WG_NTC <= p.N;
  const bool x_fast = p.vec && nc0 + p.lo >= 0 && nc0 + p.lo + p.span <= p.x.T;
  if (a_fast) stage_rows_batched<8>(p.a, as, WG_AS, b, r0, min(MT, p.R - r0), MT, nc0, WG_NTC, p.R, tid);
  else {
    for (int m = wave; m < MT; m += 4) {
      const int row = r0 + m;
      float* dst = as + m * WG_AS;
      for (int i = lane; i < WG_NTC; i += 64) {
        const int n = nc0 + i;
        dst[i] = (row < p.R && n < p.N) ? fetch_opnd(p.a, b, row, n, 0, p.R) : 0.f;
      }
    }
  }
  if (x_fast) stage_rows_batched<8>(p.x, xs, WG_XS, b, c0, min(CT, p.Cin - c0), CT, nc0 + p.lo, p.span, p.Cin, tid);
  else {
    for (int r = wave; r < CT; r += 4) {
      const int c = c0 + r;
      float* dst = xs + r * WG_XS;
      for (int i = lane; i < p.span; i += 64) dst[i] = (c < p.Cin) ? fetch_opnd(p.x, b, c, nc0 + p.lo + i, p.reflect, p.Cin) : 0.f;
    }
  }
  __syncthreads();

  if (p.bias_off >= 0 && ct == 0) {   // bias partial: sum_t dy'[co][t], 16 lanes per row, shuffle-reduced, kept per thread
    for (int rr = tid >> 4; rr < MT; rr += 16) {
      float sacc = 0.f;
      for (int i = tid & 15; i < WG_NTC; i += 16) sacc += as[rr * WG_AS + i];
      sacc += __shfl_xor(sacc, 1); sacc += __shfl_xor(sacc, 2); sacc += __shfl_xor(sacc, 4); sacc += __shfl_xor(sacc, 8);
      if ((tid & 15) == 0) bias_part[rr >> 4] += sacc;
    }
  }

  // ---- MFMA: D[co][ci] += dy'[co][t] * x'[t + j*D][ci]; A = dy (row co = ln, k = kq), B = x (k = kq, col ci = ln)
  constexpr int PER_WAVE = WG_NTC / 4;
  const float* a_lane = as + ln * WG_AS + kq + wave * PER_WAVE;
  const float* x_lane = xs + ln * WG_XS + kq + p.i0 + wave * PER_WAVE;
  float av[2][M_REP], bv[2][C_REP][J];
  auto load_frag = [&](int buf, int nn) {
#pragma unroll
    for (int m = 0; m < M_REP; ++m) av[buf][m] = a_lane[nn + m * 16 * WG_AS];
#pragma unroll
    for (int c = 0; c < C_REP; ++c)
#pragma unroll
      for (int j = 0; j < J; ++j) bv[buf][c][j] = x_lane[nn + c * 16 * WG_XS + j * D];
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
#pragma unroll 1
  for (int nn = 0; nn < PER_WAVE; nn += 8) {
    load_frag(1, nn + 4);
    mma(0);
    if (nn + 8 < PER_WAVE) load_frag(0, nn + 8);
    mma(1);
  }
  }   // chunk loop

  // ---- cross-wave sum through LDS, then the block's partial tile goes to its slab
  float* red = smem;
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < M_REP; ++m)
#pragma unroll
        for (int c = 0; c < C_REP; ++c)
#pragma unroll
          for (int j = 0; j < J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int e = ((m * C_REP + c) * J + j) * 4 + r;
              if (w == 0) red[e * 64 + lane] = acc[m][c][j][r];
              else red[e * 64 + lane] += acc[m][c][j][r];
            }
    }
  }
  __syncthreads();
  float* slab = p.slab + (long)blockIdx.x * p.slab_stride;
  const long rowlen = (long)p.Cin * p.K;
  for (int idx = tid; idx < E * 64; idx += 256) {
    const int e = idx >> 6, l = idx & 63;
    const int r = e & 3, mcj = e >> 2;
    const int j = mcj % J, mc = mcj / J;
    const int c = mc % C_REP, m = mc / C_REP;
    const int row = r0 + m * 16 + (l >> 4) * 4 + r;
    const int ci = c0 + c * 16 + (l & 15);
    if (row < p.R && ci < p.Cin) slab[row * rowlen + (long)ci * p.K + j] = red[idx];
  }
  if (p.bias_off >= 0 && ct == 0 && (tid & 15) == 0) {
#pragma unroll
    for (int i = 0; i < MT / 16; ++i) {
      const int row = r0 + (tid >> 4) + 16 * i;
      if (row < p.R) slab[p.bias_off + row] = bias_part[i];
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Register-tile variant for wide layers (R >= 32 and Cin >= 32): the 4 waves form a 2x2 grid over a
// (32*M_REP) x (32*C_REP) tile of (co, ci) and each keeps its sub-tile (all J taps) in registers while the
// block walks `tpb` consecutive 64-step time chunks; the operand rows of chunk i+1 are prefetched into
// registers while the MFMAs of chunk i run (same issue-early / commit-late split as the forward kernel).
// No cross-wave reduction; one slab per (sample, chunk group).
constexpr int WT_NTC = 64;
constexpr int WT_AS = 66;      // 2 (mod 32)
constexpr int WT_XS = 130;     // 2 (mod 32), >= 64 + 50 + alignment slack

// SCAL = 0: aligned float4 prefetch (long sequences; the chunks at the sequence ends fall back to element-wise
// staging); SCAL = 1: scalar prefetch with padding logic for every chunk (short / unaligned sequences, masked dy);
// SCAL: 0 = aligned float4 staging, 1 = scalar staging with padding logic, 2 = contiguous-tile staging of short unaligned rows
template <int M_REP, int C_REP, int J, int D, int SCAL>
__global__ __launch_bounds__(256, 2) void conv_wgrad_tile_kernel(const WgLeanP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MT = 32 * M_REP, CT = 32 * C_REP;
  constexpr int AV = (MT * (WT_NTC / 4) + 255) / 256;       // float4 per thread: dy rows
  constexpr int XVN = (CT * 32 + 255) / 256;                // float4 per thread: x rows (span <= 128)
  constexpr int AS_N = SCAL == 1 ? (MT * WT_NTC + 255) / 256 : 1;   // scalars per thread (unaligned / edge chunks)
  constexpr int XS_N = SCAL == 1 ? (CT * 128 + 255) / 256 : 1;
  constexpr int AVV = SCAL ? 1 : AV, XVV = SCAL ? 1 : XVN;
  float* as = smem;                     // [MT][WT_AS]
  float* xs = smem + MT * WT_AS;        // [CT][WT_XS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, kq = lane >> 4;
  const int wm = wave >> 1, wc = wave & 1;
  const int ctiles = (p.Cin + CT - 1) / CT;
  const int ct = blockIdx.y % ctiles, mt = blockIdx.y / ctiles;
  const int r0 = mt * MT, c0 = ct * CT;

  f32x4 acc[M_REP][C_REP][J];
#pragma unroll
  for (int m = 0; m < M_REP; ++m)
#pragma unroll
    for (int c = 0; c < C_REP; ++c)
#pragma unroll
      for (int j = 0; j < J; ++j) acc[m][c][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bias_part[MT / 16];
#pragma unroll
  for (int i = 0; i < MT / 16; ++i) bias_part[i] = 0.f;

  // chunks are (sample, 64-step tile) pairs: q -> (b = q / ntiles, tile = q % ntiles); a block owns tpb of them
  const int nchunks = p.ntiles * p.B;
  const int q_begin = blockIdx.x * p.tpb, q_end = min(nchunks, q_begin + p.tpb);
  const int avalid = min(MT, p.R - r0), xvalid = min(CT, p.Cin - c0);
  const int akind = p.a.xf.kind, xkind = p.x.xf.kind;
  // mode per chunk: 2 = aligned float4 prefetch, 1 = scalar prefetch with padding logic, 0 = element-wise fallback
  const bool vec_kinds = akind <= XF_LRELU && xkind <= XF_LRELU;
  const bool sc_kinds = (akind <= XF_LRELU || akind == XF_MASK_LRELU) && xkind <= XF_LRELU && p.span <= 128;
  constexpr bool contig = SCAL == 2;         // mode 3 below (host-checked contract)
  auto chunk_mode = [&](int q) {
    const int tile = q % p.ntiles, nc0 = tile * WT_NTC;
    if (!SCAL) return (p.vec && vec_kinds && nc0 + WT_NTC <= p.N && nc0 + p.lo >= 0 && nc0 + p.lo + p.span <= p.x.T) ? 2 : 0;
    if (contig) return 3;
    return sc_kinds ? 1 : 0;
  };
  RegTile<AVV> ar; RegTile<XVV> xr;
  RegS<AS_N> as_r, as_a; RegS<XS_N> xs_r;
  // Mode 3 (SCAL == 2): one whole short sequence per chunk (N = T <= 64, e.g. D layer 5: T = 63) whose rows are not 16-byte
  // aligned. The MT (CT) rows of a tile are still ONE contiguous, 16-byte aligned run of MT*N (CT*T) floats, so they are
  // fetched as float4 of that run (4 / 2 loads per thread instead of 16 / 9 scalar ones with per-element index arithmetic)
  // and scattered to their (row, column) in LDS at the commit. Columns >= N of the dy tile and the halo of the x tile are
  // zeroed once per block and never written again.
  constexpr int AF4 = SCAL == 2 ? (MT * WT_NTC / 4 + 255) / 256 : 1, XF4 = SCAL == 2 ? (CT * WT_NTC / 4 + 255) / 256 : 1;
  RegTile<AF4> ca, cm; RegTile<XF4> cx;
  const int n4a = (MT * p.N) >> 2, n4x = (CT * p.x.T) >> 2;
  auto issue_c = [&](int q) {
    const float* abase = p.a.p + (long)q * p.a.bs + (long)r0 * p.N;
    const float* xbase = p.x.p + (long)q * p.x.bs + (long)c0 * p.x.T;
    const srd_t ars = make_srd(abase, n4a * 16), xrs = make_srd(xbase, n4x * 16);
#pragma unroll
    for (int i = 0; i < AF4; ++i) ca.v[i] = buf_load4(ars, (tid + 256 * i) * 16);
    if (akind == XF_MASK_LRELU) {
      const srd_t mrs = make_srd(p.a.xf.aux + (long)q * p.a.xf.aux_bs + (long)r0 * p.N, n4a * 16);
#pragma unroll
      for (int i = 0; i < AF4; ++i) cm.v[i] = buf_load4(mrs, (tid + 256 * i) * 16);
    }
#pragma unroll
    for (int i = 0; i < XF4; ++i) cx.v[i] = buf_load4(xrs, (tid + 256 * i) * 16);
  };
  auto commit_c = [&]() {
    const float invn = 1.0f / (float)p.N, invt = 1.0f / (float)p.x.T;
    const float asl = p.a.xf.slope, asc = p.a.xf.scale, xsl = p.x.xf.slope, xsc = p.x.xf.scale;
#pragma unroll
    for (int i = 0; i < AF4; ++i) {
      const int f = tid + 256 * i;
      if (f < n4a) {
        int row = (int)(((float)(4 * f) + 0.5f) * invn), col = 4 * f - row * p.N;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v = ca.v[i][k];
          if (akind == XF_LRELU) v = fmaxf(v, v * asl);
          else if (akind == XF_MASK_LRELU) v = cm.v[i][k] > 0.f ? v : v * asl;
          as[row * WT_AS + col] = v * asc;
          if (++col == p.N) { col = 0; ++row; }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < XF4; ++i) {
      const int f = tid + 256 * i;
      if (f < n4x) {
        int row = (int)(((float)(4 * f) + 0.5f) * invt), col = 4 * f - row * p.x.T;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float v = cx.v[i][k];
          if (xkind == XF_LRELU) v = fmaxf(v, v * xsl);
          xs[row * WT_XS + col - p.lo] = v * xsc;
          if (++col == p.x.T) { col = 0; ++row; }
        }
      }
    }
  };
  if (contig) {
    for (int i = tid; i < MT * WT_AS + CT * WT_XS; i += 256) smem[i] = 0.f;
  }
  auto issue = [&](int q, int mode) {
    if (mode == 3) { issue_c(q); return; }
    const int b = q / p.ntiles, nc0 = (q % p.ntiles) * WT_NTC;
    const float* abase = p.a.p + (long)b * p.a.bs + (long)r0 * p.a.T;
    const float* xbase = p.x.p + (long)b * p.x.bs + (long)c0 * p.x.T;
    if (!SCAL) {
      tile_issue<AVV>(ar, abase + nc0, p.a.T, avalid, MT, WT_NTC, WT_NTC, 0, tid);
      tile_issue<XVV>(xr, xbase + nc0 + p.lo, p.x.T, xvalid, CT, p.span, p.span, 0, tid);
    } else if (SCAL == 1) {
      tile_issue_s<AS_N>(as_r, abase, p.N, avalid, MT, WT_NTC, nc0, 0, tid);
      if (akind == XF_MASK_LRELU)
        tile_issue_s<AS_N>(as_a, p.a.xf.aux + (long)b * p.a.xf.aux_bs + (long)r0 * p.a.T, p.N, avalid, MT, WT_NTC, nc0, 0, tid);
      tile_issue_s<XS_N>(xs_r, xbase, p.x.T, xvalid, CT, p.span, nc0 + p.lo, p.reflect, tid);
    }
  };
  int mode = (q_begin < q_end) ? chunk_mode(q_begin) : 0;
  if (mode) issue(q_begin, mode);

  for (int q = q_begin; q < q_end; ++q) {
    const int b = q / p.ntiles, nc0 = (q % p.ntiles) * WT_NTC;
    __syncthreads();
    if (!SCAL && mode == 2) {
      tile_commit<AVV>(ar, nullptr, nullptr, p.a.xf, as, WT_AS, avalid, MT, WT_NTC, 0, tid);
      tile_commit<XVV>(xr, nullptr, nullptr, p.x.xf, xs, WT_XS, xvalid, CT, p.span, 0, tid);
    } else if (SCAL == 2 && mode == 3) {
      commit_c();
    } else if (SCAL == 1 && mode == 1) {
      tile_commit_s<AS_N>(as_r, &as_a, p.a.xf, as, WT_AS, MT, WT_NTC, tid);
      tile_commit_s<XS_N>(xs_r, nullptr, p.x.xf, xs, WT_XS, CT, p.span, tid);
    } else {
      for (int m = wave; m < MT; m += 4) {
        float* dst = as + m * WT_AS;
        for (int i = lane; i < WT_NTC; i += 64) {
          const int n = nc0 + i;
          dst[i] = (r0 + m < p.R && n < p.N) ? fetch_opnd(p.a, b, r0 + m, n, 0, p.R) : 0.f;
        }
      }
      for (int r = wave; r < CT; r += 4) {
        float* dst = xs + r * WT_XS;
        for (int i = lane; i < p.span; i += 64) dst[i] = (c0 + r < p.Cin) ? fetch_opnd(p.x, b, c0 + r, nc0 + p.lo + i, p.reflect, p.Cin) : 0.f;
      }
    }
    __syncthreads();
    mode = (q + 1 < q_end) ? chunk_mode(q + 1) : 0;
    if (mode) issue(q + 1, mode);

    if (p.bias_off >= 0 && ct == 0) {
      for (int rr = tid >> 4; rr < MT; rr += 16) {
        float sacc = 0.f;
        for (int i = tid & 15; i < WT_NTC; i += 16) sacc += as[rr * WT_AS + i];
        sacc += __shfl_xor(sacc, 1); sacc += __shfl_xor(sacc, 2); sacc += __shfl_xor(sacc, 4); sacc += __shfl_xor(sacc, 8);
        if ((tid & 15) == 0) bias_part[rr >> 4] += sacc;
      }
    }

    const float* a_lane = as + (wm * 16 * M_REP + ln) * WT_AS + kq;
    const float* x_lane = xs + (wc * 16 * C_REP + ln) * WT_XS + kq + p.i0;
    float av[2][M_REP], bv[2][C_REP][J];
    auto load_frag = [&](int buf, int nn) {
#pragma unroll
      for (int m = 0; m < M_REP; ++m) av[buf][m] = a_lane[nn + m * 16 * WT_AS];
#pragma unroll
      for (int c = 0; c < C_REP; ++c)
#pragma unroll
        for (int j = 0; j < J; ++j) bv[buf][c][j] = x_lane[nn + c * 16 * WT_XS + j * D];
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
    const int ncols = min(WT_NTC, ((p.N - nc0) + 7) & ~7);      // skip the all-zero tail of a short last tile
    load_frag(0, 0);
#pragma unroll 1
    for (int nn = 0; nn < ncols; nn += 8) {
      load_frag(1, nn + 4);
      mma(0);
      if (nn + 8 < ncols) load_frag(0, nn + 8);
      mma(1);
    }
  }

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
  if (p.bias_off >= 0 && ct == 0 && (tid & 15) == 0) {
#pragma unroll
    for (int i = 0; i < MT / 16; ++i) {
      const int row = r0 + (tid >> 4) + 16 * i;
      if (row < p.R) slab[p.bias_off + row] = bias_part[i];
    }
  }
}

template <int M_REP, int C_REP, int J, int D, int SCAL>
static hipError_t wt_launch2(const WgLeanP& p, int B, hipStream_t st) {
  constexpr int MT = 32 * M_REP, CT = 32 * C_REP;
  auto k = conv_wgrad_tile_kernel<M_REP, C_REP, J, D, SCAL>;
  TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
  dim3 grid(p.ngroups, ((p.R + MT - 1) / MT) * ((p.Cin + CT - 1) / CT), 1);      // wide: ngroups counts (sample, tile) chunk groups
  const size_t lds = (size_t)(MT * WT_AS + CT * WT_XS) * sizeof(float);
  const WgLeanP& q = p;
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, q);
  return hipGetLastError();
}

template <int M_REP, int C_REP, int J, int D>
static hipError_t wg_launch(const WgLeanP& p, int B, hipStream_t st) {
  constexpr int MT = 16 * M_REP, CT = 16 * C_REP;
  auto k = conv_wgrad_lean_kernel<M_REP, C_REP, J, D>;
  TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
  dim3 grid(B * p.ngroups, ((p.R + MT - 1) / MT) * ((p.Cin + CT - 1) / CT), 1);
  size_t lds = (size_t)(MT * WG_AS + CT * WG_XS) * sizeof(float);
  const size_t red = (size_t)M_REP * C_REP * J * 4 * 64 * sizeof(float);
  if (red > lds) lds = red;
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  return hipGetLastError();
}

void wgrad_lean_plan(int R, int Cin, int N, int K, int B, int* ntiles, int* tpb, int* ngroups);

template <int M_REP, int C_REP, int J, int D>
static hipError_t wt_launch(const WgLeanP& p, int B, hipStream_t st) {
  // aligned long sequences with plain prologues -> float4 prefetch; everything else -> scalar prefetch
  const bool vec = p.vec && p.a.xf.kind <= XF_LRELU && p.x.xf.kind <= XF_LRELU && p.N >= 2 * WT_NTC;
  if (vec) return wt_launch2<M_REP, C_REP, J, D, 0>(p, B, st);
  // contiguous-tile staging (kernel, mode 3): one whole short sequence per chunk, full tiles, rows back to back, zero padding, tiles
  // that start on 16-byte boundaries (D layer 5: 1024 -> 1024, T = 63)
  constexpr int MT = 32 * M_REP, CT = 32 * C_REP;
  auto al16 = [](const void* ptr) { return (((uintptr_t)ptr) & 15) == 0; };
  const bool contig = g_knob[4] == 0 && p.ntiles == 1 && p.N <= WT_NTC && p.x.T == p.N && p.a.T == p.N && !p.reflect &&
                      p.R % MT == 0 && p.Cin % CT == 0 && (MT * p.N) % 4 == 0 && (CT * p.N) % 4 == 0 && (p.a.bs & 3) == 0 && (p.x.bs & 3) == 0 &&
                      al16(p.a.p) && al16(p.x.p) && p.lo <= 0 && p.N - p.lo <= WT_XS &&
                      (p.a.xf.kind <= XF_LRELU || (p.a.xf.kind == XF_MASK_LRELU && al16(p.a.xf.aux) && (p.a.xf.aux_bs & 3) == 0)) &&
                      p.x.xf.kind <= XF_LRELU;
  return contig ? wt_launch2<M_REP, C_REP, J, D, 2>(p, B, st) : wt_launch2<M_REP, C_REP, J, D, 1>(p, B, st);
}

template <int J, int D>
static hipError_t wg_launch_jd(WgLeanP& p, int B, hipStream_t st) {
  wgrad_lean_plan(p.R, p.Cin, p.N, p.K, B, &p.ntiles, &p.tpb, &p.ngroups);
  p.B = B;
  if (p.R <= 16 || p.Cin <= 16) return wg_launch<1, 1, J, D>(p, B, st);
  if (J == 15) return hipErrorNotSupported;      // 15-tap layers on the path are narrow (D layer 0: 1 -> 16)
  {   // aligned rows with plain / FiLM prologues: the pipelined kernel (conv_wgrad_pipe.hip)
    const hipError_t e = launch_conv_wgrad_pipe(p, J, D, st);
    if (e != hipErrorNotSupported) return e;
  }
  // wide layers: 64-step chunks, register tile per wave
  p.span = ((WT_NTC + (J - 1) * D - p.pad - p.lo) + 3) / 4 * 4;
  if (p.R <= 32) {
    if (J <= 3 && !wgrad_prefers_ct32(p.Cin)) return wt_launch<1, 2, J, D>(p, B, st);    // 32 x 64 block tile, 16 x 32 per wave
    return wt_launch<1, 1, J, D>(p, B, st);                 // 32 x 32 block tile
  }
  if (J <= 3 && !wgrad_prefers_ct32(p.Cin)) return wt_launch<2, 2, J, D>(p, B, st);      // 64 x 64 block tile, 32 x 32 per wave
  return wt_launch<2, 1, J, D>(p, B, st);                   // 64 x 32 block tile, 32 x 16 per wave
}

// Grid plan shared by the workspace query and the launch: chunk length, chunks per block, chunk groups per sample.
void wgrad_lean_plan(int R, int Cin, int N, int K, int B, int* ntiles, int* tpb, int* ngroups) {
  const bool narrow = (R <= 16 || Cin <= 16);
  const int ntc = narrow ? WG_NTC : WT_NTC;
  const int mt = narrow ? 16 : ((R <= 32 || K >= 11) ? 32 : 64), ctw = narrow ? 16 : ((K <= 3 && !wgrad_prefers_ct32(Cin)) ? 64 : 32);
  *ntiles = (N + ntc - 1) / ntc;
  const long tiles = (long)((R + mt - 1) / mt) * ((Cin + ctw - 1) / ctw);
  const long blocks = (long)B * (*ntiles) * tiles;
  // narrow: ~4 blocks per CU (latency-bound, no intra-block pipeline); wide: the register-tile kernels run 2 blocks
  // per CU and walk their chunks in a software pipeline, so the grid is sized to ONE resident wave of blocks
  // (<= 512 on 256 CUs): a 513th block would run alone after the others and double the kernel time
  int t = (int)(blocks / 1024);
  if (!narrow) {
    const long groups = tiles >= 512 ? 1 : 512 / tiles;
    t = (int)(((long)B * (*ntiles) + groups - 1) / groups);
  }
  if (t < 1) t = 1;
  if (narrow) {                                  // groups per sample
    if (t > *ntiles) t = *ntiles;
    *tpb = t;
    *ngroups = (*ntiles + t - 1) / t;
  } else {                                       // groups over all B * ntiles (sample, tile) chunks
    const long nchunks = (long)B * (*ntiles);
    if (t > nchunks) t = (int)nchunks;
    *tpb = t;
    *ngroups = (int)((nchunks + t - 1) / t);
  }
}

// slabs written by one launch
int wgrad_lean_nslab(int R, int Cin, int N, int K, int B) {
  int ntiles, tpb, ngroups;
  wgrad_lean_plan(R, Cin, N, K, B, &ntiles, &tpb, &ngroups);
  return (R <= 16 || Cin <= 16) ? B * ngroups : ngroups;
}

bool wgrad_lean_supported(int J, int D) {
  if (J == 1) return D == 1;
  if (J == 5 || J == 15) return D == 1;
  return (J == 3 || J == 7 || J == 11) && (D == 1 || D == 3 || D == 5);
}

// number of slabs = B * ntiles; slab element layout = module weight layout [R][Cin][K]
hipError_t launch_conv_wgrad_lean(WgLeanP p, int B, int J, int D, hipStream_t st) {
  const int first = -p.pad;
  p.lo = -(((-first) + 3) / 4 * 4);
  p.i0 = first - p.lo;
  p.span = ((WG_NTC + (J - 1) * D - p.pad - p.lo) + 3) / 4 * 4;
  if (p.span > WG_XS - 2) return hipErrorNotSupported;
#define WG_CASE(JJ, DD) if (J == JJ && D == DD) return wg_launch_jd<JJ, DD>(p, B, st);
  WG_CASE(1, 1) WG_CASE(5, 1) WG_CASE(15, 1)
  WG_CASE(3, 1) WG_CASE(3, 3) WG_CASE(3, 5)
  WG_CASE(7, 1) WG_CASE(7, 3) WG_CASE(7, 5)
  WG_CASE(11, 1) WG_CASE(11, 3) WG_CASE(11, 5)
#undef WG_CASE
  return hipErrorNotSupported;
}

}  // namespace tdvc
