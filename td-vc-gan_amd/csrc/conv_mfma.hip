// MFMA implicit-GEMM kernels for the reduced stride-1 conv problem (see conv_common.h).
// gfx950 only: v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD), 64-wide waves,
// LDS-staged input tile + weight tile, fused prologue (activation / FiLM) on load and fused
// epilogue (bias, residual, activation, MRF running mean, activation-grad masks, FiLM grads).
#include "conv_common.h"

PROF_DEFINE(tdvc_debug_gemm_prof)
namespace tdvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------------------------
// element-wise epilogue shared by the MFMA and the scalar kernels
__device__ __forceinline__ void conv_epilogue(const GemmConvP& p, float acc, int b, int ch, int u, int Ctot) {
  const long oi = (long)ch * p.Ty + u;
  float v = acc;
  if (p.epi == EPI_FWD) {
    if (p.bias) v += p.bias[ch];
    if (p.bias3) v += p.bias3[((long)b * Ctot + ch) * 3 + (u == 0 ? 0 : (u == p.Ty - 1 ? 2 : 1))];
    if (p.res) v += p.res[(long)b * p.res_bs + oi];
    if (p.post == POST_LRELU) v = lrelu_f(v, p.post_slope);
    else if (p.post == POST_TANH) v = tanhf(v);
    v *= p.out_scale;
  } else if (p.epi == EPI_MASK) {
    float m = p.mx[(long)b * p.mx_bs + oi];
    v = m > 0.f ? v : v * p.m_slope;
  } else if (p.epi == EPI_FILM) {
    const float h = p.mx[(long)b * p.mx_bs + oi];
    const float* gp = p.gb + (long)b * p.gb_bs + oi;
    const float ga = gp[0], be = gp[(long)Ctot * p.Ty];
    const float h2 = h * (1.f + ga) + be;
    const float dh2 = h2 > 0.f ? v : v * p.m_slope;
    float* dg = p.dgb + (long)b * p.dgb_bs + oi;
    dg[0] = dh2 * h;
    dg[(long)Ctot * p.Ty] = dh2;
    v = dh2 * (1.f + ga);
  }
  if (p.add) v += p.add_scale * p.add[(long)b * p.add_bs + oi];
  p.y[(long)b * p.y_bs + oi] = v;
}

// A-matrix element for reduced row `row`, reduced channel `cr`, reduced tap `j` of group g.
template <int MODE>
__device__ __forceinline__ float weight_elem(const GemmConvP& p, int g, int row, int cr, int j) {
  if (row >= p.R || cr >= p.Cred) return 0.f;
  int m = row, c = cr, k;
  if (MODE == MODE_DIRECT) {
    k = j;                                 // a flipped tap order (dgrad) is applied where the MFMA loop reads LDS
  } else if (MODE == MODE_DOWN) {
    c = cr / p.s; int phi = cr - c * p.s;
    k = phi + j * p.s;
  } else {
    m = row / p.s; int phi = row - m * p.s;
    k = phi + (p.J - 1 - j) * p.s;
  }
  if (k >= p.K) return 0.f;
  return p.w[(long)g * p.w_sg + (long)m * p.w_sm + (long)c * p.w_sc + k];
}

// ----------------------------------------------------------------------------------------------
// Block tile: MT = 16*M_REP*WM rows x NT = 16*N_REP*WN columns, 4 waves arranged WM x WN.
// PIPE: the input prologue is NONE/LeakyReLU, so interior tiles prefetch their rows into registers one chunk
// ahead; !PIPE: prologues that read a second/third tensor (FiLM, activation-grad masks) stage in batches.
template <int MODE, int M_REP, int N_REP, int WM, int WN, bool PIPE>
// (strided / transposed instances hold a prefetched weight tile and a prefetched input tile: 2 blocks = 256 VGPRs (3 = 168 for
//  the 16-row tiles), no spills;
//  at 4 blocks they spilled 36-95 registers and every prefetch load was followed by a scratch store that waited for it)
__global__ __launch_bounds__(256, (MODE != MODE_DIRECT ? (M_REP * WM == 1 ? (N_REP == 1 ? 4 : 3) : (M_REP == 1 && WM == 4 ? 3 : 2)) : (M_REP * N_REP >= 16 ? 3 : 4))) void conv_gemm_kernel(const GemmConvP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MT = 16 * M_REP * WM;
  constexpr int NT = 16 * N_REP * WN;
  static_assert(WM * WN == 4, "4 waves per block");
  float* xs = smem;                    // [Cc][XS]
  float* ws = smem + p.Cc * p.XS;      // [MT][WS], ws[m][j*Cc + c]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ln = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * NT;
  const int mtiles = (p.R + MT - 1) / MT;
  const int mt = blockIdx.y % mtiles, g = blockIdx.y / mtiles;
  const int b = blockIdx.z;
  const int r0 = mt * MT;
  const int Cx_tot = p.groups * p.x.Cg;
  const int i0 = p.i0;
  PROF_DECL

  f32x4 acc[M_REP][N_REP];
#pragma unroll
  for (int m = 0; m < M_REP; ++m)
#pragma unroll
    for (int n = 0; n < N_REP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wcol0 = wn * 16 * N_REP;   // wave's first column inside the block tile
  const int wrow0_l = wm * 16 * M_REP;

  // reflect-fold bookkeeping (dgrad of a reflect-padded stride-1 conv): a column u also collects
  // the padded positions -u (1<=u<=mp) and 2(T-1)-u (T-1-mp<=u<=T-2).
  bool needL = false, needR = false;
  if (MODE == MODE_DIRECT && p.mirror_pad > 0) {
    const int c_lo = n0 + wcol0, c_hi = c_lo + 16 * N_REP - 1;
    needL = (c_lo <= p.mirror_pad) && (c_hi >= 1);
    needR = (c_lo <= p.N - 2) && (c_hi >= p.N - 1 - p.mirror_pad);
  }

  // Software pipeline over channel chunks (cdna_hip_programming.md T14): the global loads of chunk c+1
  // (input rows and weight rows, both as aligned float4) are issued before the MFMA loop of chunk c and
  // committed to LDS after it, so HBM/L2 latency hides under the matrix work.
  constexpr int XV = NT <= 64 ? 4 : 8;     // float4 per thread: input rows (64-column tiles: the 16 x 64 one keeps 4 blocks per CU)
  constexpr int WVV = MT >= 64 ? 10 : 6;   // float4 per thread: weight rows
  RegTile<PIPE ? XV : 1> xr;
  RegTile<WVV> wr;
  const int jc = p.J * p.Cc;               // LDS weight row: ws[m][c*J + j]
  const float inv_j = 1.0f / (float)p.J;
  // float4 row tiles: interior tiles, and -- without reflect padding -- the tiles at the sequence ends too (q0 % 4 == 0 and
  // T % 4 == 0: a float4 column lies wholly inside or wholly outside [0, T) and the outside ones load as zero). Short
  // sequences (T = 500 on 256-column tiles) consist of end tiles only.
  const int q0x = n0 + p.lo;
  const bool fast_tile = (MODE != MODE_DOWN) && rows_align_ok(p.x, q0x, p.span) && ((q0x >= 0 && q0x + p.span <= p.x.T) || !p.reflect);
  const int vlo = q0x < 0 ? (-q0x) >> 2 : 0, vhi = (p.x.T - q0x) >> 2;
  const bool pipelined = PIPE && fast_tile && p.Cc * (p.span >> 2) <= XV * 256;
  // weights: natural layout (rows contiguous over (c, k)) and 16-byte alignable -> float4 row copies
  const bool wfast = (p.w_packed || (MODE == MODE_DIRECT && p.w_nat)) && MT * (jc >> 2) <= WVV * 256;
  const float* xrow0 = p.x.p + (long)b * p.x.bs + (long)(g * p.x.Cg) * p.x.T + (n0 + p.lo);
  const float* wrow0 = p.w + (long)g * p.w_sg + (long)r0 * p.w_sm;
  const Xf wxf = {XF_NONE, 0.f, 1.f, nullptr, 0};

  auto x_issue = [&](int c0) {
    if (PIPE) tile_issue<PIPE ? XV : 1>(xr, xrow0 + (long)c0 * p.x.T, p.x.T, min(p.Cc, p.Cred - c0), p.Cc, p.span, p.span, 0, tid, vlo, vhi);
  };
  auto w_issue = [&](int c0) {
    tile_issue<WVV>(wr, wrow0 + (long)c0 * p.J, (int)p.w_sm, min(MT, p.R - r0), MT, jc, min(p.Cc, ((p.Cred + 3) & ~3) - c0) * p.J, 0, tid);   // (DIRECT: J == K)
  };
  // MODE_DOWN, whole channels per chunk (p.chan_stage): coalesced time-to-depth. The s phase rows (c, 0..s-1) of one channel
  // are the SAME contiguous run of span*s samples x[c][(n0+lo)*s - pad ...], so consecutive lanes read consecutive float4 of
  // that run (one wide coalesced stream per channel instead of s stride-s gathers) and the de-interleave happens on the way
  // into LDS: sample e of the run belongs to phase e % s, column e / s. Issue and commit are split like the row tiles above:
  // a chunk that fits the 8 float4 per thread is prefetched under the previous chunk's MFMAs.
  const int xd_nvr = (p.span * p.s) >> 2;                         // float4 per channel run
  const int xd_qbase = (n0 + p.lo) * p.s - p.pad;
  const bool down_chan = PIPE && MODE == MODE_DOWN && p.chan_stage;
  // Interior blocks whose chunk fits the 8 float4 per thread take the prefetched form with everything that does not depend
  // on the chunk worked out ONCE per thread: xo[i] = byte offset of slot i's float4 inside the chunk's channels (or the
  // out-of-range sentinel), xm[i] = LDS destination | valid components << 16 | first phase << 20. Per chunk that leaves one
  // buffer load per slot and four LDS stores, no index arithmetic and no lane masks (hoisted out of the chunk loop those
  // cost 340 spilled SGPRs). First tiles whose run starts inside a float4 (pad % 4 != 0) and ragged chunk counts keep the general form.
  const bool down_pipe = down_chan && (p.Cc / p.s) * xd_nvr <= XV * 256 && (xd_qbase >= 0 || (xd_qbase & 3) == 0) && p.Cred % p.Cc == 0;
  int xo[XV], xm[XV];
  const int xd_dump = p.Cc * p.XS - 1;                              // a pad column nobody reads (XS >= span + 12)
  if (down_pipe) {
    const int tot = (p.Cc / p.s) * xd_nvr;
    const float inv_nvr = 1.0f / (float)xd_nvr, inv_s = 1.0f / (float)p.s;
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      xo[i] = 0x7f000000; xm[i] = xd_dump | (1 << 25);
      if (walk_opaque(i * 256) >= tot) continue;             // (uniform: small chunks fill one or two slots)
      const int e = tid + i * 256;
      const int ch = (int)(((float)e + 0.5f) * inv_nvr);
      const int ee = 4 * (e - ch * xd_nvr);
      const int q = xd_qbase + ee;
      const bool ok = e < tot && q >= 0 && q < p.x.T;      // (runs that start before sample 0 do so by whole float4 here)
      const int col = (int)(((float)ee + 0.5f) * inv_s);
      const int phi = ee - col * p.s;
      const int nk = ok ? min(4, p.x.T - q) : 0;
      xo[i] = ok ? (ch * p.x.T + q) * 4 : 0x7f000000;
      xm[i] = (e < tot ? ch * p.s * p.XS + phi * p.XS + col : xd_dump) | (nk << 16) | ((e < tot ? phi : 0) << 20) | ((e < tot ? 0 : 1) << 25);
    }
  }
  auto xf_issue = [&](int c0) {
    const srd_t rs = make_srd(p.x.p + (long)b * p.x.bs + (long)(g * p.x.Cg + c0 / p.s) * p.x.T, (p.Cc / p.s) * p.x.T * 4);
#pragma unroll
    for (int i = 0; i < XV; ++i)
      if (walk_opaque(i * 256) < (p.Cc / p.s) * xd_nvr) xr.v[i] = buf_load4(rs, xo[i]);
  };
  auto xf_commit = [&]() {
    const bool act = p.x.xf.kind == XF_LRELU;
    const float sl = p.x.xf.slope, sc = p.x.xf.scale;
    const int wrap = 1 - (p.s - 1) * p.XS;                         // phase s-1 -> phase 0 of the next column
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      if (walk_opaque(i * 256) >= (p.Cc / p.s) * xd_nvr) break;
      int dst = xm[i] & 0xffff;
      const int nk = (xm[i] >> 16) & 15;
      int phi = (xm[i] >> 20) & 31;
      const bool dead = (xm[i] >> 25) & 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float val = xr.v[i][k];
        if (act) val = fmaxf(val, val * sl);
        xs[dst] = k < nk ? val * sc : 0.f;
        const bool w = phi + 1 == p.s;
        dst = dead ? xd_dump : dst + (w ? wrap : p.XS);          // slots past the chunk keep writing the one pad word
        phi = w ? 0 : phi + 1;
      }
    }
  };
  if (pipelined) x_issue(0);
  if (down_pipe) xf_issue(0);
  if (wfast) w_issue(0);

  PROF(0)
  for (int c0 = 0; c0 < p.Cred; c0 += p.Cc) {
    __syncthreads();                       // every wave is done reading the previous chunk
    PROF(1)
    // ---- X' chunk -> LDS
    if (pipelined) {
      tile_commit<PIPE ? XV : 1>(xr, nullptr, nullptr, p.x.xf, xs, p.XS, min(p.Cc, p.Cred - c0), p.Cc, p.span, 0, tid);
    } else if (MODE == MODE_DOWN && p.stage_rows) {
      // large stride (STFT framing): consecutive reduced rows (c,phi) are consecutive samples in HBM,
      // so lanes run along rows; XS is odd-ish (17 mod 32) so the LDS writes stay conflict-free
      const int tot = p.Cc * p.span;
      for (int e = tid; e < tot; e += 256) {
        const int i = (int)(((float)e + 0.5f) * (1.0f / (float)p.Cc));
        const int r = e - i * p.Cc;
        const int cr = c0 + r;
        float v = 0.f;
        if (cr < p.Cred) {
          const int c = cr / p.s, phi = cr - c * p.s;
          v = fetch_opnd(p.x, b, g * p.x.Cg + c, (n0 + p.lo + i) * p.s + phi - p.pad, p.reflect, Cx_tot);
        }
        xs[r * p.XS + i] = v;
      }
    } else if (down_chan) {
      if (down_pipe) {
        xf_commit();
      } else {
        stage_down_runs<2>(p.x, xs, p.XS, b, g * p.x.Cg + c0 / p.s, p.Cc / p.s, (min(p.Cred, c0 + p.Cc) - c0) / p.s, p.s, xd_qbase, p.span,
                           Cx_tot, tid);
      }
    } else if (MODE == MODE_DOWN && p.chan_stage) {
      // prologues with a second tensor (activation-grad masks): element-wise
      const int nch = p.Cc / p.s, len = p.span * p.s;
      const int nchv = (min(p.Cred, c0 + p.Cc) - c0) / p.s;
      const int chan0 = g * p.x.Cg + c0 / p.s;
      const float inv_s = 1.0f / (float)p.s;
      for (int ch = 0; ch < nch; ++ch) {
        const bool rv = ch < nchv;
        float* rows = xs + ch * p.s * p.XS;
        for (int e = tid; e < len; e += 256) {
          const int i = (int)(((float)e + 0.5f) * inv_s);
          const int phi = e - i * p.s;
          rows[phi * p.XS + i] = rv ? fetch_opnd(p.x, b, chan0 + ch, xd_qbase + e, 0, Cx_tot) : 0.f;
        }
      }
    } else if (!PIPE && fast_tile) {
      stage_rows_batched<4>(p.x, xs, p.XS, b, g * p.x.Cg + c0, min(p.Cc, p.Cred - c0), p.Cc, n0 + p.lo, p.span, Cx_tot, tid, vlo, vhi);
    } else {
      // any mode, any position (sequence ends, short sequences): element-wise with the full padding logic, 8 elements per
      // thread in flight
      constexpr int SB = MODE == MODE_DIRECT ? 4 : 8;       // (the DIRECT instances already sit at their register cap)
      const int tot = p.Cc * p.span;
      const float inv_span = 1.0f / (float)p.span;
      for (int eb = 0; eb < tot; eb += SB * 256) {
        float v[SB];
#pragma unroll
        for (int i = 0; i < SB; ++i) {
          const int e = eb + tid + i * 256;
          const int r = (int)(((float)e + 0.5f) * inv_span);
          const int ii = e - r * p.span;
          const int cr = c0 + r;
          int ch, phi = 0;
          if (MODE == MODE_DOWN) { int c = cr / p.s; phi = cr - c * p.s; ch = g * p.x.Cg + c; }
          else ch = g * p.x.Cg + cr;
          const int xi = n0 + p.lo + ii;
          const int q = (MODE == MODE_DOWN) ? xi * p.s + phi - p.pad : xi;
          v[i] = (e < tot && cr < p.Cred) ? fetch_opnd(p.x, b, ch, q, p.reflect, Cx_tot) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
          const int e = eb + tid + i * 256;
          if (e >= tot) continue;
          const int r = (int)(((float)e + 0.5f) * inv_span);
          xs[r * p.XS + (e - r * p.span)] = v[i];
        }
      }
    }
    PROF(2)
    // ---- A chunk -> LDS, ws[m][c*J + j]
    if (wfast) {
      tile_commit<WVV>(wr, nullptr, nullptr, wxf, ws, p.WS, min(MT, p.R - r0), MT, jc, 0, tid);
    } else {
      // reduced-layout gather (strided / transposed / grouped weights): 8 elements per thread in flight
      constexpr int WB = MODE == MODE_DIRECT ? 4 : 8;
      const int wtot = MT * jc;
      const float inv_jc = 1.0f / (float)jc;
      for (int eb = 0; eb < wtot; eb += WB * 256) {
        float wv[WB];
#pragma unroll
        for (int i = 0; i < WB; ++i) {
          const int e = eb + tid + i * 256;
          const int m = (int)(((float)e + 0.5f) * inv_jc);
          const int idx = e - m * jc;
          const int c = (int)(((float)idx + 0.5f) * inv_j);
          wv[i] = e < wtot ? weight_elem<MODE>(p, g, r0 + m, c0 + c, idx - c * p.J) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < WB; ++i) {
          const int e = eb + tid + i * 256;
          if (e >= wtot) continue;
          const int m = (int)(((float)e + 0.5f) * inv_jc);
          ws[m * p.WS + (e - m * jc)] = wv[i];
        }
      }
    }
    PROF(8)
    __syncthreads();
    PROF(3)
    if (c0 + p.Cc < p.Cred) {              // next chunk's loads fly while this chunk's MFMAs run
      if (pipelined) x_issue(c0 + p.Cc);
      if (down_pipe) xf_issue(c0 + p.Cc);
      if (wfast) w_issue(c0 + p.Cc);
    }

    PROF(4)
    // ---- MFMA main loop: K dimension = (tap j, channel c); each 16x16x4 step takes 4 channels of one tap
    const int csteps = p.Cc >> 2;
    const float* a_base = ws + (wrow0_l + ln) * p.WS + kq * p.J;
    const float* b_base = xs + kq * p.XS + wcol0 + ln + i0;
    for (int j = 0; j < p.J; ++j) {
      const float* aj = a_base + (p.tap_flip ? p.J - 1 - j : j);
      const float* bj = b_base + j * p.d;
      for (int cs = 0; cs < csteps; ++cs) {
        float a[M_REP], bv[N_REP];
#pragma unroll
        for (int m = 0; m < M_REP; ++m) a[m] = aj[m * 16 * p.WS + cs * 4 * p.J];
#pragma unroll
        for (int n = 0; n < N_REP; ++n) bv[n] = bj[cs * 4 * p.XS + n * 16];
#pragma unroll
        for (int m = 0; m < M_REP; ++m)
#pragma unroll
          for (int n = 0; n < N_REP; ++n) {
            // MODE_DOWN computes the transposed tile (operands swapped: same registers): a lane then holds four CONSECUTIVE
            // columns of one row, which the epilogue stores as one float4 along time
            if (MODE == MODE_DOWN) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[n], a[m], acc[m][n], 0, 0, 0);
            else acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bv[n], acc[m][n], 0, 0, 0);
          }
      }
    }

    if (MODE == MODE_DIRECT && (needL || needR)) {
      for (int side = 0; side < 2; ++side) {
        if (side == 0 ? !needL : !needR) continue;
        int mb[N_REP]; bool mv[N_REP];
#pragma unroll
        for (int n = 0; n < N_REP; ++n) {
          const int u = n0 + wcol0 + n * 16 + ln;
          if (side == 0) { mv[n] = (u >= 1 && u <= p.mirror_pad && u < p.N); mb[n] = -u - n0 + i0; }
          else { mv[n] = (u >= p.N - 1 - p.mirror_pad && u <= p.N - 2 && u >= 0); mb[n] = 2 * (p.N - 1) - u - n0 + i0; }
        }
        for (int j = 0; j < p.J; ++j) {
          const float* aj = a_base + (p.tap_flip ? p.J - 1 - j : j);
          for (int cs = 0; cs < csteps; ++cs) {
            float a[M_REP], bv[N_REP];
#pragma unroll
            for (int m = 0; m < M_REP; ++m) a[m] = aj[m * 16 * p.WS + cs * 4 * p.J];
#pragma unroll
            for (int n = 0; n < N_REP; ++n) {
              const int idx = mb[n] + j * p.d;
              const bool ok = mv[n] && idx >= 0 && idx < p.span;
              const float t = xs[(cs * 4 + kq) * p.XS + (ok ? idx : 0)];
              bv[n] = ok ? t : 0.f;
            }
#pragma unroll
            for (int m = 0; m < M_REP; ++m)
#pragma unroll
              for (int n = 0; n < N_REP; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bv[n], acc[m][n], 0, 0, 0);
          }
        }
      }
    }
    PROF(5)
  }

  // ---- epilogue: C/D layout col = lane&15, row = (lane>>4)*4 + reg
  const int Cy_tot = p.groups * p.Cy_g;
  if (MODE == MODE_DOWN) {
    // transposed tile: lane (ln, kq) holds row ln, columns kq*4 .. kq*4+3
    const bool vec = (p.Ty & 3) == 0 && (p.y_bs & 3) == 0 && (((uintptr_t)p.y) & 15) == 0 && !p.bias3 && p.epi != EPI_FILM &&
                     (!p.res || ((p.res_bs & 3) == 0 && (((uintptr_t)p.res) & 15) == 0)) &&
                     (!p.add || ((p.add_bs & 3) == 0 && (((uintptr_t)p.add) & 15) == 0)) &&
                     (p.epi != EPI_MASK || ((p.mx_bs & 3) == 0 && (((uintptr_t)p.mx) & 15) == 0));
#pragma unroll
    for (int m = 0; m < M_REP; ++m) {
      const int row = r0 + wrow0_l + m * 16 + ln;
      if (row >= p.R) continue;
      const int ch = g * p.Cy_g + row;
      const float bias = (p.epi == EPI_FWD && p.bias) ? p.bias[ch] : 0.f;
#pragma unroll
      for (int n = 0; n < N_REP; ++n) {
        const int u0 = n0 + wcol0 + n * 16 + kq * 4;
        if (u0 >= p.N) continue;
        f32x4 v = acc[m][n];
        if (vec && u0 + 3 < p.N) {
          const long oi = (long)ch * p.Ty + u0;
          if (p.epi == EPI_FWD) {
            v += bias;
            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (long)b * p.res_bs + oi);
            if (p.post == POST_LRELU) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = lrelu_f(v[q], p.post_slope);
            } else if (p.post == POST_TANH) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = tanhf(v[q]);
            }
            v *= p.out_scale;
          } else if (p.epi == EPI_MASK) {
            const f32x4 mm = *reinterpret_cast<const f32x4*>(p.mx + (long)b * p.mx_bs + oi);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = mm[q] > 0.f ? v[q] : v[q] * p.m_slope;
          }
          if (p.add) v += p.add_scale * *reinterpret_cast<const f32x4*>(p.add + (long)b * p.add_bs + oi);
          *reinterpret_cast<f32x4*>(p.y + (long)b * p.y_bs + oi) = v;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (u0 + q < p.N) conv_epilogue(p, v[q], b, ch, u0 + q, Cy_tot);
        }
      }
    }
    PROF(6)
    PROF_END
    return;
  }
  const bool up_epi_ok = (p.epi == EPI_PLAIN || p.epi == EPI_MASK || (p.epi == EPI_FWD && !p.res && !p.bias3)) && !p.add;
  if (MODE == MODE_UP && (p.s == 4 || p.s == 8) && up_epi_ok && (p.pad & 3) == 0 && (p.Ty & 3) == 0 &&
      (p.y_bs & 3) == 0 && (((uintptr_t)p.y) & 15) == 0 && (p.epi != EPI_MASK || ((p.mx_bs & 3) == 0 && (((uintptr_t)p.mx) & 15) == 0))) {
    // Depth-to-time with stride 4 / 8: the lane's four accumulator rows are four CONSECUTIVE phases of one output channel,
    // i.e. four consecutive output samples u0 .. u0+3 with u0 % 4 == 0 -> one dwordx4 store per 16x16 tile and lane
    // (consecutive lanes = consecutive columns = consecutive float4) instead of four stride-s scalar stores.
#pragma unroll
    for (int m = 0; m < M_REP; ++m) {
      const int row0 = r0 + wrow0_l + m * 16 + kq * 4;
      if (row0 >= p.R) continue;
      const int mm = row0 / p.s, phi0 = row0 - mm * p.s;
      const long rbase = (long)(g * p.Cy_g + mm) * p.Ty;
      const float bias = (p.epi == EPI_FWD && p.bias) ? p.bias[g * p.Cy_g + mm] : 0.f;
#pragma unroll
      for (int n = 0; n < N_REP; ++n) {
        const int col = n0 + wcol0 + n * 16 + ln;
        if (col >= p.N) continue;
        const int u0 = col * p.s + phi0 - p.pad;
        if (u0 + 3 < 0 || u0 >= p.Ty) continue;
        f32x4 v = acc[m][n];
        if (u0 >= 0 && u0 + 3 < p.Ty) {
          if (p.epi == EPI_MASK) {
            const f32x4 mm4 = *reinterpret_cast<const f32x4*>(p.mx + (long)b * p.mx_bs + rbase + u0);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = mm4[q] > 0.f ? v[q] : v[q] * p.m_slope;
          } else if (p.epi == EPI_FWD) {
            v += bias;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = p.post == POST_LRELU ? lrelu_f(v[q], p.post_slope) : (p.post == POST_TANH ? tanhf(v[q]) : v[q]);
            v *= p.out_scale;
          }
          *reinterpret_cast<f32x4*>(p.y + (long)b * p.y_bs + rbase + u0) = v;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (u0 + q >= 0 && u0 + q < p.Ty) conv_epilogue(p, v[q], b, g * p.Cy_g + mm, u0 + q, Cy_tot);
        }
      }
    }
    PROF(6)
    PROF_END
    return;
  }
  if (MODE == MODE_UP && p.s == 2 && up_epi_ok && (p.R & 3) == 0) {
    // stride 2: the lane's four rows are (channel mm, phases 0 1) and (channel mm + 1, phases 0 1): two float2 stores
    // (8 bytes at 4-byte alignment when the padding is odd); 16 lanes = 16 consecutive float2 of one channel
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int m = 0; m < M_REP; ++m) {
      const int row0 = r0 + wrow0_l + m * 16 + kq * 4;
      if (row0 >= p.R) continue;
      const int mm0 = row0 >> 1;
#pragma unroll
      for (int n = 0; n < N_REP; ++n) {
        const int col = n0 + wcol0 + n * 16 + ln;
        if (col >= p.N) continue;
        const int u0 = col * 2 - p.pad;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ch = g * p.Cy_g + mm0 + h;
          f32x2 v = {acc[m][n][2 * h], acc[m][n][2 * h + 1]};
          if (u0 >= 0 && u0 + 1 < p.Ty) {
            const long oi = (long)ch * p.Ty + u0;
            if (p.epi == EPI_MASK) {
              const f32x2 m2 = *reinterpret_cast<const f32x2*>(p.mx + (long)b * p.mx_bs + oi);
              v[0] = m2[0] > 0.f ? v[0] : v[0] * p.m_slope; v[1] = m2[1] > 0.f ? v[1] : v[1] * p.m_slope;
            } else if (p.epi == EPI_FWD) {
              const float bias = p.bias ? p.bias[ch] : 0.f;
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                float t = v[q] + bias;
                t = p.post == POST_LRELU ? lrelu_f(t, p.post_slope) : (p.post == POST_TANH ? tanhf(t) : t);
                v[q] = t * p.out_scale;
              }
            }
            *reinterpret_cast<f32x2*>(p.y + (long)b * p.y_bs + oi) = v;
          } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
              if (u0 + q >= 0 && u0 + q < p.Ty) conv_epilogue(p, v[q], b, ch, u0 + q, Cy_tot);
          }
        }
      }
    }
    PROF(6)
    PROF_END
    return;
  }
#pragma unroll
  for (int m = 0; m < M_REP; ++m) {
#pragma unroll
    for (int n = 0; n < N_REP; ++n) {
      const int col = n0 + wcol0 + n * 16 + ln;
      if (col >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + wrow0_l + m * 16 + kq * 4 + r;
        if (row >= p.R) continue;
        int ch, u;
        if (MODE == MODE_UP) {
          const int mm = row / p.s, phi = row - mm * p.s;
          ch = g * p.Cy_g + mm; u = col * p.s + phi - p.pad;
          if (u < 0 || u >= p.Ty) continue;
        } else { ch = g * p.Cy_g + row; u = col; }
        conv_epilogue(p, acc[m][n][r], b, ch, u, Cy_tot);
      }
    }
  }
  PROF(6)
  PROF_END
}

// ----------------------------------------------------------------------------------------------
// Scalar kernel for the same reduced problem: one thread per output element. Used for shapes the
// MFMA tiling does not cover (Cred % 4 != 0, i.e. 1-channel inputs and depthwise FIR filters) and
// as an independent cross-check of the MFMA path in tests.
template <int MODE>
__global__ __launch_bounds__(256) void conv_scalar_kernel(const GemmConvP p) {
  const long total = (long)p.N * p.R;
  const int g = blockIdx.y, b = blockIdx.z;
  const int Cx_tot = p.groups * p.x.Cg, Cy_tot = p.groups * p.Cy_g;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int row = (int)(e / p.N), col = (int)(e - (long)row * p.N);
    float acc = 0.f;
    for (int cr = 0; cr < p.Cred; ++cr) {
      int ch, phi = 0;
      if (MODE == MODE_DOWN) { int c = cr / p.s; phi = cr - c * p.s; ch = g * p.x.Cg + c; }
      else ch = g * p.x.Cg + cr;
      for (int j = 0; j < p.J; ++j) {
        const float wv = weight_elem<MODE>(p, g, row, cr, (MODE == MODE_DIRECT && p.tap_flip) ? p.J - 1 - j : j);
        int xi;
        if (MODE == MODE_DIRECT) xi = col + j * p.d - p.pad;
        else if (MODE == MODE_DOWN) xi = col + j;
        else xi = col - (p.J - 1) + j;
        const int q = (MODE == MODE_DOWN) ? xi * p.s + phi - p.pad : xi;
        float xv = fetch_opnd(p.x, b, ch, q, p.reflect, Cx_tot);
        if (MODE == MODE_DIRECT && p.mirror_pad > 0) {
          // fold: padded positions -col and 2(T-1)-col also map onto this column
          if (col >= 1 && col <= p.mirror_pad) xv += fetch_opnd(p.x, b, ch, -col + j * p.d - p.pad, 0, Cx_tot);
          if (col >= p.N - 1 - p.mirror_pad && col <= p.N - 2)
            xv += fetch_opnd(p.x, b, ch, 2 * (p.N - 1) - col + j * p.d - p.pad, 0, Cx_tot);
        }
        acc += wv * xv;
      }
    }
    int ch, u;
    if (MODE == MODE_UP) {
      const int mm = row / p.s, phi = row - mm * p.s;
      ch = g * p.Cy_g + mm; u = col * p.s + phi - p.pad;
      if (u < 0 || u >= p.Ty) continue;
    } else { ch = g * p.Cy_g + row; u = col; }
    conv_epilogue(p, acc, b, ch, u, Cy_tot);
  }
}

// ----------------------------------------------------------------------------------------------
// Weight-gradient: dW[r][c][j] = sum_{b,n} A'[r][n] * X'[c][n + j*d + lo]. One block = one
// (row tile, 16-channel tile) x (time chunk, batch group); the 4 waves split the time chunk and
// are summed through LDS; the block writes its partial into a slab (no atomics: deterministic).
template <int MODE, int M_REP, int J>
#ifndef WGRAD_OCC
#define WGRAD_OCC 4      // resident blocks per CU asked of the generic weight-grad kernel (A/B: make ab EXTRA=-DWGRAD_OCC=3)
#endif
__global__ __launch_bounds__(256, (M_REP * J >= 14 ? 3 : WGRAD_OCC)) void conv_wgrad_kernel(const WgradP p, int bpb, int B) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MT = 16 * M_REP;
  constexpr int E = M_REP * J * 4;
  float* as = smem;                 // [MT][AS]
  float* xs = smem + MT * p.AS;     // [16][XS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, kq = lane >> 4;
  const int mtiles = (p.R + MT - 1) / MT, ctiles = (p.Cred + 15) / 16;
  int yy = blockIdx.y;
  const int ct = yy % ctiles; yy /= ctiles;
  const int mt = yy % mtiles; const int g = yy / mtiles;
  const int bg = blockIdx.x / p.ntiles, tile = blockIdx.x % p.ntiles;
  const int nc0 = tile * p.NTc;
  const int r0 = mt * MT, c0 = ct * 16;
  const int Ca_tot = p.groups * p.a.Cg, Cx_tot = p.groups * p.x.Cg;
  const int i0 = p.i0;

  f32x4 acc[M_REP][J];
#pragma unroll
  for (int m = 0; m < M_REP; ++m)
#pragma unroll
    for (int j = 0; j < J; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int b_lo = bg * bpb, b_hi = min(B, b_lo + bpb);
  for (int b = b_lo; b < b_hi; ++b) {
    __syncthreads();
    if (rows_align_ok(p.a, nc0, p.NTc) && p.N == p.a.T) {   // float4 rows; the chunk past the sequence end loads zeros
      stage_rows_batched<8>(p.a, as, p.AS, b, g * p.a.Cg + r0, min(MT, p.R - r0), MT, nc0, p.NTc, Ca_tot, tid, 0, (p.a.T - nc0) >> 2);
    } else {
      for (int m = wave; m < MT; m += 4) {
        const int row = r0 + m;
        const int ch = g * p.a.Cg + row;
        float* dst = as + m * p.AS;
        for (int i = lane; i < p.NTc; i += 64) {
          const int n = nc0 + i;
          dst[i] = (row < p.R && n < p.N) ? fetch_opnd(p.a, b, ch, n, 0, Ca_tot) : 0.f;
        }
      }
    }
    const int qx = nc0 + p.lo;
    if (MODE != MODE_DOWN && rows_align_ok(p.x, qx, p.span) && ((qx >= 0 && qx + p.span <= p.x.T) || !p.reflect)) {
      stage_rows_batched<8>(p.x, xs, p.XS, b, g * p.x.Cg + c0, min(16, p.Cred - c0), 16, qx, p.span, Cx_tot, tid, qx < 0 ? (-qx) >> 2 : 0,
                            (p.x.T - qx) >> 2);
    } else if (MODE == MODE_DOWN && (16 % p.s) == 0 && p.x.xf.kind <= XF_LRELU) {
      // coalesced time-to-depth (see conv_gemm_kernel): the 16 reduced rows are 16/s whole channels
      stage_down_runs<4>(p.x, xs, p.XS, b, g * p.x.Cg + c0 / p.s, 16 / p.s, (min(p.Cred, c0 + 16) - c0) / p.s, p.s, qx * p.s - p.pad, p.span,
                         Cx_tot, tid);
    } else if (MODE == MODE_DOWN && (16 % p.s) == 0) {
      // coalesced time-to-depth (see conv_gemm_kernel): the 16 reduced rows are 16/s whole channels
      const int nch = 16 / p.s, len = p.span * p.s;
      const long qbase = (long)(nc0 + p.lo) * p.s - p.pad;
      const float inv_s = 1.0f / (float)p.s;
      for (int chi = 0; chi < nch; ++chi) {
        const int cr = c0 + chi * p.s;
        const bool rv = cr < p.Cred;
        const int chan = g * p.x.Cg + cr / p.s;
        float* rows = xs + chi * p.s * p.XS;
        for (int e = tid; e < len; e += 256) {
          const int i = (int)(((float)e + 0.5f) * inv_s);
          const int phi = e - i * p.s;
          rows[phi * p.XS + i] = rv ? fetch_opnd(p.x, b, chan, (int)(qbase + e), 0, Cx_tot) : 0.f;
        }
      }
    } else {
      for (int r = wave; r < 16; r += 4) {
        const int cr = c0 + r;
        const bool rv = cr < p.Cred;
        int ch, phi = 0;
        if (MODE == MODE_DOWN) { int c = cr / p.s; phi = cr - c * p.s; ch = g * p.x.Cg + c; }
        else ch = g * p.x.Cg + cr;
        float* dst = xs + r * p.XS;
        for (int i = lane; i < p.span; i += 64) {
          const int xi = nc0 + p.lo + i;
          const int q = (MODE == MODE_DOWN) ? xi * p.s + phi - p.pad : xi;
          dst[i] = rv ? fetch_opnd(p.x, b, ch, q, p.reflect, Cx_tot) : 0.f;
        }
      }
    }
    __syncthreads();

    const int per_wave = p.NTc >> 2;
    const float* a_base = as + ln * p.AS + kq;
    const float* b_base = xs + ln * p.XS + kq + i0;
    for (int nn = wave * per_wave; nn < (wave + 1) * per_wave; nn += 4) {
      float a[M_REP];
#pragma unroll
      for (int m = 0; m < M_REP; ++m) a[m] = a_base[m * 16 * p.AS + nn];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const float bv = b_base[nn + j * p.d];
#pragma unroll
        for (int m = 0; m < M_REP; ++m)
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bv, acc[m][j], 0, 0, 0);
      }
    }
  }

  // ---- cross-wave sum through LDS, one wave at a time (bounds the buffer to E*64 floats)
  float* red = smem;
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < M_REP; ++m)
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = (m * J + j) * 4 + r;
            if (w == 0) red[e * 64 + lane] = acc[m][j][r];
            else red[e * 64 + lane] += acc[m][j][r];
          }
    }
  }
  __syncthreads();
  float* slab = p.slab + (long)blockIdx.x * p.slab_stride;
  for (int idx = tid; idx < E * 64; idx += 256) {
    const int e = idx >> 6, l = idx & 63;
    const int r = e & 3, mj = e >> 2;
    const int j = mj % J, m = mj / J;
    const int row = r0 + m * 16 + (l >> 4) * 4 + r;
    const int cr = c0 + (l & 15);
    if (row >= p.R || cr >= p.Cred) continue;
    int c = cr, k = j;
    if (MODE == MODE_DOWN) { c = cr / p.s; k = (cr - c * p.s) + j * p.s; if (k >= p.K) continue; }
    slab[(long)g * p.w_sg + (long)row * p.w_sm + (long)c * p.w_sc + k] = red[idx];
  }
}

// Row sums of the transformed dy operand -> bias gradient. One block per (channel, sample).
__global__ __launch_bounds__(256) void conv_bias_grad_kernel(const Opnd a, int N, int Ctot, float* dbias) {
  const int ch = blockIdx.x, b = blockIdx.y;
  float s = 0.f;
  const float* row = a.p + (long)b * a.bs + (long)ch * a.T;
  const float* mrow = a.xf.aux ? a.xf.aux + (long)b * a.xf.aux_bs + (long)ch * a.T : nullptr;
  const int kind = a.xf.kind;
  if ((N & 3) == 0 && N == a.T && (((uintptr_t)row) & 15) == 0 && (kind <= XF_LRELU || (kind == XF_MASK_LRELU && (((uintptr_t)mrow) & 15) == 0))) {
    // aligned rows, plain / LeakyReLU / LeakyReLU-mask prologue: float4 loads (block-uniform choice)
    const float sl = a.xf.slope, sc = a.xf.scale;
    for (int n = threadIdx.x * 4; n < N; n += 1024) {
      f32x4_t v = *reinterpret_cast<const f32x4_t*>(row + n);
      if (kind == XF_LRELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = lrelu_f(v[q], sl);
      } else if (kind == XF_MASK_LRELU) {
        const f32x4_t m = *reinterpret_cast<const f32x4_t*>(mrow + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = m[q] > 0.f ? v[q] : v[q] * sl;
      }
      s += ((v[0] + v[1]) + (v[2] + v[3])) * sc;
    }
  } else {
    for (int n = threadIdx.x; n < N; n += 256) s += fetch_opnd(a, b, ch, n, 0, Ctot);
  }
  __shared__ float sh[4];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&dbias[ch], sh[0] + sh[1] + sh[2] + sh[3]);
}

// dw[map(i)] += sum_s slab[s][i]; grid.y splits the slabs so small weights still fill the chip.
// map: compact row-major [rows][rowlen] -> dst row stride (channel-window weights), identity otherwise.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* slab, int nslab, long stride, long n, float* dw,
                                                          int rowlen, long dst_row_stride, int per_y, long n_w, float* dbias) {
  const int s0 = blockIdx.y * per_y, s1 = min(nslab, s0 + per_y);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int k = s0; k < s1; ++k) s += slab[(long)k * stride + i];
    float* dstp;
    if (i >= n_w) dstp = dbias + (i - n_w);          // trailing bias partials
    else {
      long dst = i;
      if (dst_row_stride != rowlen) { const long r = i / rowlen; dst = r * dst_row_stride + (i - r * rowlen); }
      dstp = dw + dst;
    }
    if (gridDim.y == 1) *dstp += s; else atomicAdd(dstp, s);
  }
}

// Several folds in one launch (deferred folds, tdvc_fold_defer): the folds of a backward pass are ~270 launches of ~4-16 us
// each, mostly launch latency; batched, their blocks run side by side. A block finds its fold by its position in the
// concatenated grids.
struct FoldDesc {
  const float* slab; float* dw; float* dbias;
  long stride, n, dst_row_stride, n_w;
  int nslab, rowlen, gx, gy, per_y, blk0;
};
constexpr int FOLD_MAX = 24;
struct FoldBatch { FoldDesc d[FOLD_MAX]; int count, nblocks; };

__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(const FoldBatch fb) {
  int k = 0;
  for (int i = 1; i < fb.count; ++i)
    if ((int)blockIdx.x >= fb.d[i].blk0) k = i;
  const FoldDesc& f = fb.d[k];
  const int lb = blockIdx.x - f.blk0, bx = lb % f.gx, by = lb / f.gx;
  const int s0 = by * f.per_y, s1 = min(f.nslab, s0 + f.per_y);
  for (long i = (long)bx * 256 + threadIdx.x; i < f.n; i += (long)f.gx * 256) {
    float s = 0.f;
    for (int q = s0; q < s1; ++q) s += f.slab[(long)q * f.stride + i];
    float* dstp;
    if (i >= f.n_w) dstp = f.dbias + (i - f.n_w);
    else {
      long dst = i;
      if (f.dst_row_stride != f.rowlen) { const long r = i / f.rowlen; dst = r * f.dst_row_stride + (i - r * f.rowlen); }
      dstp = f.dw + dst;
    }
    if (f.gy == 1) *dstp += s; else atomicAdd(dstp, s);
  }
}

// Scalar weight-gradient: one block per weight element, reduction over (b, n).
template <int MODE>
__global__ __launch_bounds__(256) void conv_wgrad_scalar_kernel(const WgradP p, int B, float* dw) {
  // blockIdx.x enumerates (g, row, c, k) of the module weight
  const int Cw = p.x.Cg;            // channels per group in the module weight's 2nd dim
  long e = blockIdx.x;
  const int k = (int)(e % p.K); e /= p.K;
  const int c = (int)(e % Cw); e /= Cw;
  const int row = (int)(e % p.R); const int g = (int)(e / p.R);
  const int Ca_tot = p.groups * p.a.Cg, Cx_tot = p.groups * p.x.Cg;
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    for (int n = threadIdx.x; n < p.N; n += 256) {
      const float av = fetch_opnd(p.a, b, g * p.a.Cg + row, n, 0, Ca_tot);
      const int q = (MODE == MODE_DOWN) ? n * p.s + k - p.pad : n + k * p.d - p.pad;
      s += av * fetch_opnd(p.x, b, g * p.x.Cg + c, q, p.reflect, Cx_tot);
    }
  }
  __shared__ float sh[4];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) dw[(long)g * p.w_sg + (long)row * p.w_sm + (long)c * p.w_sc + k] += sh[0] + sh[1] + sh[2] + sh[3];
}

// ----------------------------------------------------------------------------------------------
// Reduced-layout weight copy for the strided / transposed / grouped problems: dst[g][r][cr][j] = A[r][cr][j] of group g,
// reduced channels padded to a multiple of 4 with zeros. The conv kernel then copies weight rows as float4, prefetched one
// chunk ahead, instead of gathering them element by element (index arithmetic + a dependent L2 round trip per batch) in
// every block and chunk: that gather was 50 % of the block time of the 64 -> 128 stride-8 conv (profiles/r02_f_*).
template <int MODE>
__global__ __launch_bounds__(256) void weight_repack_kernel(const GemmConvP p, float* dst, int Cred4, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int j = (int)(i % p.J);
    long t = i / p.J;
    const int cr = (int)(t % Cred4);
    t /= Cred4;
    const int r = (int)(t % p.R), g = (int)(t / p.R);
    dst[i] = weight_elem<MODE>(p, g, r, cr, j);
  }
}

}  // namespace tdvc

// ================================================================================================
// host-side launchers (called from conv_api.cpp)
#include <map>
namespace tdvc {

// Scratch for the repacked weights of ONE launch, per stream: repack -> conv -> next repack are ordered by the stream, so
// one buffer per stream is enough. Buffers are allocated outside graph capture only (the eager warm-up step sizes them) and
// never freed or moved, because a captured graph holds their address. A stream first seen DURING capture (torch captures on
// a side stream of its own) borrows the buffer the warm-up made on another stream: the capture stream and the warm-up
// stream never run convs at the same time. nullptr = no buffer can be had: the caller keeps the in-kernel gather.
static float* weight_scratch(hipStream_t st, size_t bytes) {
  static std::mutex mu;
  static std::map<hipStream_t, std::pair<float*, size_t>> pool;
  std::lock_guard<std::mutex> lk(mu);
  auto it = pool.find(st);
  if (it != pool.end() && it->second.second >= bytes) return it->second.first;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
  (void)hipGetLastError();
  if (capturing) {
    for (auto& kv : pool)
      if (kv.second.second >= bytes) { pool[st] = kv.second; return kv.second.first; }
    return nullptr;
  }
  size_t nb = bytes < ((size_t)8 << 20) ? ((size_t)8 << 20) : (bytes + ((size_t)1 << 20) - 1) / ((size_t)1 << 20) * ((size_t)1 << 20);
  float* nbuf = nullptr;
  if (hipMalloc(&nbuf, nb) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  pool[st] = std::make_pair(nbuf, nb);          // an outgrown buffer stays allocated: a captured graph may still point at it
  return nbuf;
}

template <int MODE, int M_REP, int N_REP, int WM, int WN, bool PIPE>
static hipError_t launch_gemm_cfg2(const GemmConvP& p, int B, hipStream_t st) {
  constexpr int MT = 16 * M_REP * WM, NT = 16 * N_REP * WN;
  auto k = conv_gemm_kernel<MODE, M_REP, N_REP, WM, WN, PIPE>;
  TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
  dim3 grid((p.N + NT - 1) / NT, p.groups * ((p.R + MT - 1) / MT), B);
  size_t lds = (size_t)(p.Cc * p.XS + MT * p.WS) * sizeof(float);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  return hipGetLastError();
}

template <int MODE, int M_REP, int N_REP, int WM, int WN>
static hipError_t launch_gemm_cfg(const GemmConvP& p, int B, hipStream_t st) {
  if (p.x.xf.kind <= XF_LRELU) return launch_gemm_cfg2<MODE, M_REP, N_REP, WM, WN, true>(p, B, st);
  return launch_gemm_cfg2<MODE, M_REP, N_REP, WM, WN, false>(p, B, st);
}

// Tile shape selection; fills the LDS geometry (Cc, span, XS, WS) for the chosen NT/MT.
template <int MODE>
hipError_t launch_conv_gemm(GemmConvP p, int B, hipStream_t st) {
  // Tile choice: the largest tile whose grid still gives every CU about two blocks (256 CUs); small problems (short
  // sequences, few samples x few row tiles) fall through to the smallest tile = the most blocks.
  struct Cand { int MT, NT, cfg; };
  static const Cand cands[] = {{64, 256, 2}, {32, 256, 1}, {64, 64, 4}, {16, 256, 0}, {16, 64, 3}};
  int MT = 16, NT = 64, cfg = 3;
  for (const Cand& c : cands) {
    if (c.MT > 16 && p.R <= 16) continue;
    if (c.MT > 32 && p.R <= 32) continue;
    if (c.NT == 256 && p.N <= 80) continue;
    MT = c.MT; NT = c.NT; cfg = c.cfg;
    const long blocks = (long)((p.R + c.MT - 1) / c.MT) * ((p.N + c.NT - 1) / c.NT) * p.groups * B;
    if (blocks >= 512) break;
  }
  const int hi = (MODE == MODE_DIRECT) ? (p.J - 1) * p.d - p.pad + p.mirror_pad : (MODE == MODE_DOWN ? p.J - 1 : 0);
  const int first = (MODE == MODE_DIRECT) ? -p.pad : (MODE == MODE_DOWN ? 0 : -(p.J - 1));   // position read by column 0, tap 0
  int lo = (MODE == MODE_DIRECT) ? -p.pad - p.mirror_pad : first;
  lo = -(((-lo) + 3) / 4 * 4);              // tile origins are multiples of 4: keeps interior tiles float4-aligned
  p.lo = lo;
  p.i0 = first - lo;
  p.span = ((NT + hi - lo) + 3) / 4 * 4;
  p.XS = ((p.span + 31) / 32) * 32 + 16;   // == 16 (mod 32): the two k-rows of a 32-lane group hit disjoint banks
  // channel chunk: multiple of 4, sized so that LDS stays <= ~60 KB (>= 2 blocks per CU)
  const int unit = 4;
  p.stage_rows = (MODE == MODE_DOWN && p.s >= 16) ? 1 : 0;
  if (p.stage_rows) p.XS += 1;             // 17 (mod 32): row-major lane order writes conflict-free, reads 1 extra cycle
  int Cc = unit;
  while (true) {
    int next = Cc + unit;
    // (16-row tiles do 1-4 MFMAs per reduction step and wave: a longer chunk keeps the barrier count of long reductions down)
    if (next > (MT == 16 ? 64 : 32) || next > ((p.Cred + unit - 1) / unit) * unit) break;
    size_t lds = (size_t)(next * p.XS + MT * (p.J * next + 2)) * 4;
    if (lds > 60 * 1024) break;
    // keep the chunk inside the kernels' register-prefetch budgets (4 / 8 float4 of input, 24/36 weights per thread)
    if (Cc >= 8 && ((long)next * (p.span >> 2) > (NT <= 64 ? 4 : 8) * 256 || (long)MT * p.J * next > (MT >= 64 ? 10 : 6) * 1024)) break;
    Cc = next;
  }
  p.chan_stage = 0;
  if (MODE == MODE_DOWN && !p.stage_rows) {   // whole channels per chunk -> coalesced time-to-depth staging (kernel)
    int unit_s = p.s;                         // lcm(4, s)
    while (unit_s % 4) unit_s += p.s;
    if (Cc >= unit_s) {
      Cc = Cc / unit_s * unit_s; p.chan_stage = 1;
      while (Cc > unit_s && p.Cred % Cc != 0) Cc -= unit_s;   // equal chunks: the prefetched staging needs Cred % Cc == 0
    }
    else if (((p.Cred + 3) / 4) * 4 == Cc && Cc % p.s == 0) p.chan_stage = 1;   // the whole reduction in one chunk
  }
  p.Cc = Cc;
  p.WS = p.J * Cc + 2;                      // WS/2 odd: 16 rows x 2 k-lanes hit 32 distinct banks (odd J)
  p.w_nat = (MODE == MODE_DIRECT && p.w_sc == p.K && (p.w_sm & 3) == 0 && (p.w_sg & 3) == 0 && (p.Cred & 3) == 0 &&
             (((uintptr_t)p.w) & 15) == 0) ? 1 : 0;
  p.w_packed = 0;
  if (!p.w_nat && g_knob[1] == 0) {
    const int Cred4 = (p.Cred + 3) / 4 * 4;
    const long total = (long)p.groups * p.R * Cred4 * p.J;
    float* scr = weight_scratch(st, (size_t)total * sizeof(float));
    if (scr) {
      auto rk = weight_repack_kernel<MODE>;
      TDVC_TRACE(rk);
      long gx = (total + 255) / 256; if (gx > 2048) gx = 2048;
      hipLaunchKernelGGL(rk, dim3((unsigned)gx), dim3(256), 0, st, p, scr, Cred4, total);
      p.w = scr; p.w_sg = (long)p.R * Cred4 * p.J; p.w_sm = (long)Cred4 * p.J; p.w_sc = p.J; p.w_packed = 1;
    }
  }
  switch (cfg) {
    case 0: return launch_gemm_cfg<MODE, 1, 4, 1, 4>(p, B, st);
    case 1: return launch_gemm_cfg<MODE, 2, 4, 1, 4>(p, B, st);
    case 2: return launch_gemm_cfg<MODE, 4, 4, 1, 4>(p, B, st);
    case 3: return launch_gemm_cfg<MODE, 1, 1, 1, 4>(p, B, st);
    default: return launch_gemm_cfg<MODE, 1, 4, 4, 1>(p, B, st);
  }
}

template <int MODE>
hipError_t launch_conv_scalar(GemmConvP p, int B, hipStream_t st) {
  p.lo = 0; p.span = 0;
  long total = (long)p.N * p.R;
  int gx = (int)((total + 255) / 256); if (gx > 4096) gx = 4096; if (gx < 1) gx = 1;
  TDVC_TRACE(conv_scalar_kernel<MODE>);
  hipLaunchKernelGGL(conv_scalar_kernel<MODE>, dim3(gx, p.groups, B), dim3(256), 0, st, p);
  return hipGetLastError();
}

template hipError_t launch_conv_gemm<MODE_DIRECT>(GemmConvP, int, hipStream_t);
template hipError_t launch_conv_gemm<MODE_DOWN>(GemmConvP, int, hipStream_t);
template hipError_t launch_conv_gemm<MODE_UP>(GemmConvP, int, hipStream_t);
template hipError_t launch_conv_scalar<MODE_DIRECT>(GemmConvP, int, hipStream_t);
template hipError_t launch_conv_scalar<MODE_DOWN>(GemmConvP, int, hipStream_t);
template hipError_t launch_conv_scalar<MODE_UP>(GemmConvP, int, hipStream_t);

template <int MODE, int M_REP, int J>
static hipError_t launch_wgrad_cfg(WgradP& p, int B, int bpb, hipStream_t st) {
  constexpr int MT = 16 * M_REP;
  auto k = conv_wgrad_kernel<MODE, M_REP, J>;
  TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
  const int mtiles = (p.R + MT - 1) / MT, ctiles = (p.Cred + 15) / 16;
  const int nbg = (B + bpb - 1) / bpb;
  dim3 grid(nbg * p.ntiles, p.groups * mtiles * ctiles, 1);
  size_t lds = (size_t)(MT * p.AS + 16 * p.XS) * sizeof(float);
  size_t red = (size_t)M_REP * J * 4 * 64 * sizeof(float);
  if (red > lds) lds = red;
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p, bpb, B);
  return hipGetLastError();
}

template <int MODE, int M_REP>
static hipError_t launch_wgrad_j(WgradP& p, int B, int bpb, hipStream_t st) {
  switch (p.J) {
    case 1: return launch_wgrad_cfg<MODE, M_REP, 1>(p, B, bpb, st);
    case 2: return launch_wgrad_cfg<MODE, M_REP, 2>(p, B, bpb, st);
    case 3: return launch_wgrad_cfg<MODE, M_REP, 3>(p, B, bpb, st);
    case 5: return launch_wgrad_cfg<MODE, M_REP, 5>(p, B, bpb, st);
    case 7: return launch_wgrad_cfg<MODE, M_REP, 7>(p, B, bpb, st);
    case 11: return launch_wgrad_cfg<MODE, M_REP, 11>(p, B, bpb, st);
    default: return hipErrorInvalidValue;
  }
}

bool wgrad_mfma_supported(int J) { return J == 1 || J == 2 || J == 3 || J == 5 || J == 7 || J == 11; }

// Geometry shared by the workspace query and the launch. Returns number of slabs.
int wgrad_geometry(WgradP& p, int B, int* bpb_out) {
  const bool direct = p.mode == MODE_DIRECT;
  int NTc = p.N >= 512 ? 512 : ((p.N + 15) / 16) * 16;
  if (NTc < 16) NTc = 16;
  const int M_REP = p.R > 16 ? 2 : 1;
  if (M_REP == 2 && NTc > 256) NTc = 256;
  p.NTc = NTc;
  p.ntiles = (p.N + NTc - 1) / NTc;
  const int hi = direct ? (p.J - 1) * p.d - p.pad : p.J - 1;
  const int first = direct ? -p.pad : 0;
  p.lo = -(((-first) + 3) / 4 * 4);
  p.i0 = first - p.lo;
  p.span = ((NTc + hi - p.lo) + 3) / 4 * 4;
  p.XS = ((p.span + 31) / 32) * 32 + 2;     // 2 (mod 32): 16 rows x 2 k-lanes hit 32 distinct banks
  p.AS = ((NTc + 31) / 32) * 32 + 2;
  // short sequences: loop several samples inside one block (fewer slabs to write and fold), but keep about two blocks
  // per CU: a 64-sample batch walked by 64 blocks left three quarters of the chip idle
  int bpb = 1;
  if (p.N <= 128) {
    const int MTg = 16 * M_REP;
    const long gy = (long)p.groups * ((p.R + MTg - 1) / MTg) * ((p.Cred + 15) / 16) * p.ntiles;
    long nbg = (512 + gy - 1) / gy;               // batch groups wanted
    if (nbg > B) nbg = B;
    if (nbg < 1) nbg = 1;
    bpb = (int)((B + nbg - 1) / nbg);
  }
  *bpb_out = bpb;
  return ((B + bpb - 1) / bpb) * p.ntiles;
}

template <int MODE>
hipError_t launch_conv_wgrad(WgradP p, int B, int bpb, hipStream_t st) {
  if (p.R > 16) return launch_wgrad_j<MODE, 2>(p, B, bpb, st);
  return launch_wgrad_j<MODE, 1>(p, B, bpb, st);
}
template hipError_t launch_conv_wgrad<MODE_DIRECT>(WgradP, int, int, hipStream_t);
template hipError_t launch_conv_wgrad<MODE_DOWN>(WgradP, int, int, hipStream_t);

template <int MODE>
hipError_t launch_conv_wgrad_scalar(WgradP p, int B, long nweights, float* dw, hipStream_t st) {
  TDVC_TRACE(conv_wgrad_scalar_kernel<MODE>);
  hipLaunchKernelGGL(conv_wgrad_scalar_kernel<MODE>, dim3((unsigned)nweights), dim3(256), 0, st, p, B, dw);
  return hipGetLastError();
}
template hipError_t launch_conv_wgrad_scalar<MODE_DIRECT>(WgradP, int, long, float*, hipStream_t);
template hipError_t launch_conv_wgrad_scalar<MODE_DOWN>(WgradP, int, long, float*, hipStream_t);

// Deferred folds (tdvc_fold_defer(1)): launch_slab_reduce queues the fold on its stream instead of launching it; the queue
// is flushed as ONE launch when it is full, when a queued fold already targets the same gradient (keeps the accumulation
// order of a layer that is used twice in a backward pass), or by tdvc_fold_flush. The caller keeps the slabs intact until
// the flush (td-vc-gan_amd/ops.py hands every weight-grad its own region of a ring and flushes on wrap) and flushes before
// anything reads the gradients.
static int g_fold_defer = 0;
static std::mutex g_fold_mu;
static std::map<hipStream_t, FoldBatch>& fold_pending() { static std::map<hipStream_t, FoldBatch> m; return m; }

static hipError_t fold_flush_locked(hipStream_t st) {
  auto it = fold_pending().find(st);
  if (it == fold_pending().end() || it->second.count == 0) return hipSuccess;
  FoldBatch& fb = it->second;
  TDVC_TRACE(slab_reduce_multi_kernel);
  hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(fb.nblocks), dim3(256), 0, st, fb);
  fb.count = 0; fb.nblocks = 0;
  return hipGetLastError();
}
hipError_t fold_flush(hipStream_t st) { std::lock_guard<std::mutex> lk(g_fold_mu); return fold_flush_locked(st); }
void fold_set_defer(int on) { g_fold_defer = on ? 1 : 0; }
// Drop the queued folds of a stream without running them (a backward pass that raised, an abandoned graph capture): their
// slab pointers refer to ring regions that later weight-grad calls overwrite.
void fold_reset(hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_fold_mu);
  auto it = fold_pending().find(st);
  if (it != fold_pending().end()) { it->second.count = 0; it->second.nblocks = 0; }
}

hipError_t launch_slab_reduce(const float* slab, int nslab, long stride, long n, float* dw, int rowlen, long dst_row_stride,
                              hipStream_t st, long n_w = -1, float* dbias = nullptr) {
  if (n_w < 0) n_w = n;
  int gx = (int)((n + 255) / 256); if (gx > 2048) gx = 2048; if (gx < 1) gx = 1;
  int gy = 1;
  if (nslab > 8) { gy = 1024 / gx; if (gy > (nslab + 7) / 8) gy = (nslab + 7) / 8; if (gy < 1) gy = 1; }
  const int per_y = (nslab + gy - 1) / gy;
  gy = (nslab + per_y - 1) / per_y;
  if (g_fold_defer) {
    std::lock_guard<std::mutex> lk(g_fold_mu);
    FoldBatch& fb = fold_pending()[st];
    bool clash = fb.count == FOLD_MAX;
    for (int i = 0; i < fb.count && !clash; ++i)
      clash = fb.d[i].dw == dw || (dbias && fb.d[i].dbias == dbias);
    if (clash) { const hipError_t e = fold_flush_locked(st); if (e != hipSuccess) return e; }
    FoldDesc& f = fb.d[fb.count++];
    f.slab = slab; f.dw = dw; f.dbias = dbias; f.stride = stride; f.n = n; f.dst_row_stride = dst_row_stride; f.n_w = n_w;
    f.nslab = nslab; f.rowlen = rowlen; f.gx = gx; f.gy = gy; f.per_y = per_y; f.blk0 = fb.nblocks;
    fb.nblocks += gx * gy;
    return hipSuccess;
  }
  TDVC_TRACE(slab_reduce_kernel);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(gx, gy), dim3(256), 0, st, slab, nslab, stride, n, dw, rowlen, dst_row_stride, per_y, n_w, dbias);
  return hipGetLastError();
}

hipError_t launch_bias_grad(const Opnd& a, int N, int Ctot, int B, float* dbias, hipStream_t st) {
  TDVC_TRACE(conv_bias_grad_kernel);
  hipLaunchKernelGGL(conv_bias_grad_kernel, dim3(Ctot, B), dim3(256), 0, st, a, N, Ctot, dbias);
  return hipGetLastError();
}

}  // namespace tdvc
