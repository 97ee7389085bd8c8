// Lean MFMA kernel for the hot stride-1 convolutions (dilated trunk convs, 1x1 posconvs, FiLM conditioning
// convs, D layer 5) and their input-gradients: groups == 1, Tout == Tin, T % 4 == 0, Cin % 4 == 0, weights
// with contiguous (c,k) rows (module layout for forward, the pre-transposed copy for input-grad).
//
// Differences from the generic kernel in conv_mfma.hip (which stays the fallback for strided / transposed /
// grouped / odd shapes and for edge cases):
//   * prologue kind and epilogue kind are template parameters -> no per-element switches, ~5x fewer VALU ops;
//   * MFMA operand roles are swapped (A = input tile, B = weights) so that each lane owns 4 CONSECUTIVE time
//     steps of one output channel: the epilogue reads/writes float4 (dwordx4) instead of scalars;
//   * a compact parameter block (fits in SGPRs without spilling);
//   * interior tiles stage through registers one channel chunk ahead of the MFMA loop (T14 split).
#include "conv_common.h"
#include "conv_lean.h"

PROF_DEFINE(tdvc_debug_lean_prof)

namespace tdvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// prologue kinds LXF_* are declared in conv_lean.h (ACT: none or LeakyReLU by slope)
// LXF_COND: the input tile is not loaded but COMPUTED in the block: rows = channels of FiLM's cond_var.0 output
//   cv0[c][t] = sum_{ce,j} W0x[c][ce][j] * exc[ce][t+j-1] + k3[b][c][edge(t)]   (8 excitation channels, 3 taps, K = 24)
// as a small MFMA pre-pass per channel chunk, LeakyReLU'd straight into the LDS tile that cond_var.2 then convolves;
// the 136-channel intermediate is written once (for the backward pass) and never read back by the forward.



template <int XFK>
__device__ __forceinline__ float lean_xform(const LeanP& p, float v, int b, int c, int q) {
  if (XFK == LXF_ACT) { v = fmaxf(v, v * p.slope); }
  else {
    const float* a = p.aux + (long)b * p.aux_bs + (long)c * p.T + q;
    if (XFK == LXF_FILM) { const float h = v * (1.f + a[0]) + a[(long)p.Cin * p.T]; v = fmaxf(h, h * p.slope); }
    else if (XFK == LXF_MASK_LRELU) { v = a[0] > 0.f ? v : v * p.slope; }
    else { v = v * (1.f - a[0] * a[0]); }
  }
  return v * p.in_scale;
}

template <int XFK>
__device__ __forceinline__ float lean_fetch(const LeanP& p, int b, int c, int q) {
  if (p.reflect) { if (q < 0) q = -q; else if (q >= p.T) q = 2 * (p.T - 1) - q; }
  if (q < 0 || q >= p.T) return 0.f;
  return lean_xform<XFK>(p, p.x[(long)b * p.x_bs + (long)c * p.T + q], b, c, q);
}

#ifndef LEAN_OCC_SMALL
#define LEAN_OCC_SMALL 4      // resident blocks per CU asked for the 16-row tiles (A/B knob: make ab EXTRA=-DLEAN_OCC_SMALL=5)
#endif
// Resident blocks per CU the register budget must allow: 2 for the 48/64/144-row x 256/64-column tiles (>= 36 accumulator
// registers), 3 for 32 x 256, 4 for the rest -- including the 64 x 64 tile (16 accumulator registers), whose 1024-block
// grids (D layer 5) then run as ONE round instead of 1.33 rounds of 768 slots.
constexpr int lean_min_blocks(int m_rep, int n_rep, int wm, int xfk) {
  if (m_rep == 1 && n_rep == 4 && wm == 1 && xfk == LXF_ACT) return 5;   // 16 x 256 tile, plain prologue: 5 resident blocks (<= 96 VGPRs) is what the HBM-bound convs run at
  return m_rep * n_rep >= 9 ? 2 : ((m_rep * n_rep >= 8 || (wm == 4 && xfk == LXF_FILM)) ? 3 : LEAN_OCC_SMALL);   // (FiLM prologue on 64 x 64: 12 spills at 4)
}
// FOLD: short sequences (T = 16 / 32: discriminator layer 5 of the two sub-sampled discriminators) put p.fold = 64 / T SAMPLES
// side by side in one 64-column tile -- segment s of the LDS tile holds sample b0 + s with its own zero-padded halo, sub-tile
// n of the MFMA loop reads from its sample's segment, the epilogue scatters the sub-tiles back to their samples. Without it
// three quarters (T = 16) or half (T = 32) of every tile's matrix work is padding.
template <int M_REP, int N_REP, int WM, int WN, int XFK, int EPI, bool FOLD = false>
__global__ __launch_bounds__(256, lean_min_blocks(M_REP, N_REP, WM, XFK)) void conv_lean_kernel(const LeanP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MT = 16 * M_REP * WM, NT = 16 * N_REP * WN;
  constexpr int XVP = (M_REP >= 9 || NT <= 64) ? 4 : (MT >= 32 ? 12 : 6);   // max row-walk passes of the prefetched input tile
  constexpr int WVP = MT >= 48 ? 10 : (MT >= 32 ? 6 : 4);   // ... and of the weight tile
  float* xs = smem;
  float* ws = smem + p.xnp * p.xrp * p.XS;
  constexpr int WX = 25;                                  // odd row stride of the staged W0x chunk
  float* es = ws + p.wnp * p.wrp * p.WS;                  // LXF_COND: excitation tile [Cv][ES]
  float* wx = es + p.Cv * p.ES;                           // LXF_COND: W0x chunk [Cc][WX]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ln = lane & 15, kq = lane >> 4;
  // Block order. The dispatcher deals blocks round-robin over the 8 XCDs (private L2 each), so with the plain
  // (time tile, row tile, sample) order the row tiles that read one input tile, and the time neighbours that share
  // its halo, sit behind eight different L2s. Remap (bijective for any grid size, cdna_hip_programming.md T1): the
  // blocks that share an XCD take one contiguous run of work items, row tile fastest, then time, then sample.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.swz) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int nwg = gx * gy * gridDim.z;
    const int orig = bx + gx * (by + gy * bz);
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    by = wg % gy;
    const int t = wg / gy;
    bx = t % gx; bz = t / gx;
  }
  const int n0 = FOLD ? 0 : bx * NT, r0 = by * MT, b = FOLD ? bz * p.fold : bz;     // FOLD: b = first sample of the tile
  const int wcol0 = wn * 16 * N_REP, wrow0 = wm * 16 * M_REP;
  PROF_DECL

  f32x4 acc[M_REP][N_REP];
#pragma unroll
  for (int m = 0; m < M_REP; ++m)
#pragma unroll
    for (int n = 0; n < N_REP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int q0 = n0 + p.lo;
  const bool vec_ok = p.vec != 0;                      // rows 16-byte aligned -> float4 staging / epilogue
  // Every 16-byte-aligned tile is staged by the row walk, sequence ends included: a float4 column lies either wholly
  // inside [0, T) or wholly outside (q0 % 4 == 0, T % 4 == 0), and outside columns carry an out-of-range offset, so
  // the buffer loads return the zero padding. Reflect padding (forward trunk convs, ACT prologue only) is patched
  // into the halo columns of the two end tiles after the commit.
  // (the row walk only needs dword-aligned rows: sequences with T % 4 != 0 take it too, with the components past the row
  //  end masked at the commit; only the dwordx4 epilogue needs 16-byte aligned rows)
  const bool interior = (XFK != LXF_COND) && (XFK == LXF_ACT || !p.reflect);
  const bool tail = (p.T & 3) != 0;
  const bool end_tile = q0 < 0 || q0 + p.span > p.T;
  const int jc = p.K * p.Cc;
  const float* xrow0 = p.x + (long)b * p.x_bs;
  const float* arow0 = (XFK != LXF_ACT && XFK != LXF_COND) ? p.aux + (long)b * p.aux_bs : nullptr;
  const float* wgrow = p.w + (long)r0 * p.Cw;

  // Row-walk staging (conv_common.h): thread = (row of the pass, float4 column); a pass covers xrp (wrp) whole rows,
  // so the per-element cost is one buffer load, one offset add and one LDS store -- no index arithmetic in the loop.
  // Rows past the tensor end fall outside the descriptor's range and load as zero.
  RowWalk xw = make_walk(tid, p.span >> 2, p.xrp, p.T, p.XS, q0, p.T);
  if (FOLD) {   // float4 column vv of the staged row = column vv % nvs of segment vv / nvs: same LDS offset, another sample's row
    const int nvec = p.span >> 2, nvs = p.seg >> 2;
    const int rsub = (int)(((float)tid + 0.5f) * (1.0f / (float)nvec));
    const int vv = tid - rsub * nvec;
    const int sg = vv / nvs, q = q0 + 4 * (vv - sg * nvs);
    xw.voff = (xw.active && q >= 0 && q < p.T && b + sg < p.Bn) ? (sg * p.x_bs + rsub * p.T + q) * 4 : 0x7f000000;
  }
  const int fold_bytes = FOLD ? (p.fold - 1) * p.x_bs * 4 : 0;      // the descriptors span the tile's samples
  const RowWalk ww = make_walk(tid, jc >> 2, p.wrp, p.Cw, p.WS, 0, 1 << 30);
  RegTile<XVP> xr;
  RegTile<WVP> wr;
  const bool xpipe = (XFK == LXF_ACT) && interior;
  if (XFK == LXF_COND) {   // excitation tile: positions q0-1 .. q0+span, zero outside the sequence ('same' zero padding)
    const float* eb = p.x + (long)b * p.x_bs;
    for (int r = wave; r < p.Cv; r += 4)
      for (int i = lane; i < p.span + 2; i += 64) {
        const int pos = q0 - 1 + i;
        es[r * p.ES + i] = (pos >= 0 && pos < p.T) ? eb[(long)r * p.T + pos] : 0.f;
      }
  }
  auto x_issue = [&](int c0) {
    const srd_t rs = make_srd(xrow0 + (long)c0 * p.T, (p.Cin - c0) * p.T * 4 + fold_bytes);
    walk_issue<XVP>(xr, rs, xw, p.xnp);
  };
  auto w_issue = [&](int c0) {
    const srd_t rs = make_srd(wgrow + (long)c0 * p.K, ((p.Cout - r0) * p.Cw - c0 * p.K) * 4);
    walk_issue<WVP>(wr, rs, ww, p.wnp);
  };
  if (xpipe) x_issue(0);
  w_issue(0);

  bool needL = false, needR = false;
  if (p.mirror > 0) {
    const int c_lo = n0 + wcol0, c_hi = c_lo + 16 * N_REP - 1;
    needL = (c_lo <= p.mirror) && (c_hi >= 1);
    needR = (c_lo <= p.T - 2) && (c_hi >= p.T - 1 - p.mirror);
  }

  for (int c0 = 0; c0 < p.Cin; c0 += p.Cc) {
    const int cvalid = min(p.Cc, p.Cin - c0);
    PROF(0)
    __syncthreads();
    PROF(1)
    PROF_WAITV()
    PROF(8)
    if (XFK == LXF_COND) {
      // stage this chunk's W0x rows, then compute lrelu(cv0) for the chunk with MFMA: D[t][c] = E[t][k] * W0x[k][c]
      const int kv = p.Cv * 3;
      for (int idx = tid; idx < p.Cc * kv; idx += 256) {
        const int c = idx / kv, k = idx - c * kv;
        wx[c * WX + k] = (c < cvalid) ? p.cw[(long)(c0 + c) * p.cw_stride + k] : 0.f;
      }
      __syncthreads();
      const int ntl = p.Cc >> 4;                          // channel tiles of 16 (Cc is 16 or 32)
      float k3v[2][3];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int c = c0 + nt * 16 + ln;
        const bool cv = nt < ntl && (nt * 16 + ln) < cvalid;
#pragma unroll
        for (int e = 0; e < 3; ++e) k3v[nt][e] = cv ? p.k3[((long)b * p.Cin + c) * 3 + e] : 0.f;
      }
      const int mtl = (p.span + 15) >> 4;
      // per-lane operands of the pre-pass that do not depend on the time tile: the (channel, tap) walk of the A operand
      // and the whole B operand (W0x chunk) live in registers for all the time tiles of this chunk
      int eoff[6]; float bwr[2][6];
#pragma unroll
      for (int st = 0; st < 6; ++st) {
        const int k = st * 4 + kq;
        const int ce = (k * 11) >> 5;                       // k / 3 for k < 32
        eoff[st] = (k < kv) ? ce * p.ES + (k - ce * 3) : -1;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bwr[nt][st] = (nt < ntl && k < kv) ? wx[(nt * 16 + ln) * WX + k] : 0.f;
      }
      for (int mtile = wave; mtile < mtl; mtile += 4) {
        f32x4 pa[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        const float* ep = es + mtile * 16 + ln;
#pragma unroll
        for (int st = 0; st < 6; ++st) {
          if (st * 4 < kv) {
            const float a = eoff[st] >= 0 ? ep[eoff[st]] : 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              if (nt >= ntl) continue;
              pa[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bwr[nt][st], pa[nt], 0, 0, 0);
            }
          }
        }
        const int t0 = mtile * 16 + kq * 4;                 // lane owns 4 consecutive positions of channel (nt*16 + ln)
        if (t0 < p.span) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            if (nt >= ntl) continue;
            const int cl = nt * 16 + ln;
            f32x4 raw, act;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int pos = q0 + t0 + r;
              const bool inside = pos >= 0 && pos < p.T && cl < cvalid;
              const float v = pa[nt][r] + (pos == 0 ? k3v[nt][0] : (pos == p.T - 1 ? k3v[nt][2] : k3v[nt][1]));
              raw[r] = inside ? v : 0.f;
              act[r] = inside ? fmaxf(v, v * p.slope) * p.in_scale : 0.f;
            }
            *reinterpret_cast<f32x4*>(xs + cl * p.XS + t0) = act;
            // the block row 0 of each time tile also materialises cv0 (own columns only) for the backward pass
            const int pos0 = q0 + t0;
            if (p.cv0 && by == 0 && cl < cvalid && pos0 >= n0 && pos0 < n0 + NT && pos0 < p.T)
              *reinterpret_cast<f32x4*>(p.cv0 + (long)b * p.cv0_bs + (long)(c0 + cl) * p.T + pos0) = raw;
          }
        }
      }
    } else if (xpipe) {
      walk_commit_act<XVP>(xr, xw, p.xnp, xs, p.slope, p.in_scale, tail);
      if (end_tile && p.reflect) {   // reflect halo of the first / last tile: columns with q < 0 or q >= T
        const int nl = q0 < 0 ? -q0 : 0;
        const int rfirst = p.T - q0;                       // first column index with q >= T
        const int nr = rfirst < p.span ? p.span - rfirst : 0;
        const int nh = nl + nr;
        __syncthreads();                                   // the zero-filled halo columns were written by other threads
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
      }
    } else if (interior) {   // FiLM / activation-mask prologues: batches of 4 passes, up to three tensors in flight
      const int xbytes = (p.Cin - c0) * p.T * 4 + fold_bytes;
      const srd_t rx = make_srd(xrow0 + (long)c0 * p.T, xbytes);
      const srd_t ra = make_srd(arow0 + (long)c0 * p.T, xbytes);
      const srd_t rb = make_srd(arow0 + (long)(p.Cin + c0) * p.T, XFK == LXF_FILM ? xbytes : 0);
      int vo = xw.voff;
      float* dst = xs + xw.loff;
      for (int pb = 0; pb < p.xnp; pb += 4) {
        f32x4 t[4], a[4], c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (pb + i < p.xnp) {
            t[i] = buf_load4(rx, vo); a[i] = buf_load4(ra, vo);
            if (XFK == LXF_FILM) c[i] = buf_load4(rb, vo);
            vo += xw.gstep;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (pb + i < p.xnp) {
            f32x4 v = t[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (XFK == LXF_FILM) { const float h = v[q] * (1.f + a[i][q]) + c[i][q]; v[q] = fmaxf(h, h * p.slope); }
              else if (XFK == LXF_MASK_LRELU) v[q] = a[i][q] > 0.f ? v[q] : v[q] * p.slope;
              else v[q] = v[q] * (1.f - a[i][q] * a[i][q]);
              v[q] *= p.in_scale;
              if (tail && q >= xw.nk) v[q] = 0.f;
            }
            if (xw.active) *reinterpret_cast<f32x4*>(dst) = v;
            dst += xw.lstep;
          }
        }
      }
    } else {   // edge tile: per-element padding logic
      for (int r = wave; r < p.Cc; r += 4) {
        float* row = xs + r * p.XS;
        const bool rv = r < cvalid;
        for (int i = lane; i < p.span; i += 64) row[i] = rv ? lean_fetch<(XFK == LXF_COND ? LXF_ACT : XFK)>(p, b, c0 + r, q0 + i) : 0.f;
      }
    }
    walk_commit_w<WVP>(wr, ww, p.wnp, ws);
    PROF(2)
    __syncthreads();
    PROF(3)
    if (c0 + p.Cc < p.Cin) {
      if (xpipe) x_issue(c0 + p.Cc);
      w_issue(c0 + p.Cc);
    }
    PROF(4)

    // MFMA: D[t][co] += X'[t][k] * W[k][co]; one step = 4 channels of one tap. Fragments of step s+1 are
    // read from LDS into the other register set before the MFMAs of step s issue (software pipelining);
    // the (tap, channel-group) walk is kept in scalar registers so a step costs 1 + M_REP vector adds.
    const int csteps = cvalid >> 2;                        // rows past cvalid are never read (Cin % 4 == 0)
    const int nsteps = p.K * csteps;
    const float* w_lane = ws + (wrow0 + ln) * p.WS + kq * p.K;
    const float* x_lane = xs + kq * p.XS + wcol0 + ln + p.i0;
    const int wrep = 16 * p.WS;
    int sj = 0, scs = 0;                                   // scalar walk state
    int woff = p.flip ? p.K - 1 : 0, xoff = 0;
    auto advance = [&]() {
      if (++scs == csteps) { scs = 0; ++sj; woff = p.flip ? p.K - 1 - sj : sj; xoff = sj * p.d; }
      else { woff += 4 * p.K; xoff += 4 * p.XS; }
    };
    float wv[2][M_REP], xv[2][N_REP];
    int noff[N_REP];                                       // LDS column of sub-tile n relative to the wave's first column
#pragma unroll
    for (int n = 0; n < N_REP; ++n) {
      const int col = wcol0 + n * 16;
      noff[n] = FOLD ? (col / p.T) * p.seg + (col % p.T) - wcol0 : n * 16;
    }
    auto load_frag = [&](int buf) {
      const float* wp = w_lane + woff;
      const float* xp = x_lane + xoff;
#pragma unroll
      for (int m = 0; m < M_REP; ++m) wv[buf][m] = wp[m * wrep];
#pragma unroll
      for (int n = 0; n < N_REP; ++n) xv[buf][n] = xp[FOLD ? noff[n] : n * 16];
    };
    auto mma = [&](int buf) {
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
    }
    if (s < nsteps) mma(0);

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
    PROF(5)
  }

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
  // One float4 of the output: channel co, time steps t0 .. t0+3 (all inside the tensor). Returns the stored values.
  auto epi_store = [&](f32x4 v, const int co, const int t0, const float bias, const int b) -> f32x4 {     // b: the sample (shadows the block's)
    const long oi = (long)co * p.T + t0;
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
      if (N_REP >= 2 && p.mbits) {   // sign bits instead of the fp32 mask source: one word per 32 time steps (host: T % 32 == 0)
        const unsigned wbits = p.mbits[(long)b * p.mb_bs + (long)co * (p.T >> 5) + (t0 >> 5)] >> (t0 & 31);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ((wbits >> q) & 1u) ? v[q] : v[q] * p.m_slope;
      } else {
        const f32x4 mm = *reinterpret_cast<const f32x4*>(p.mx + (long)b * p.mx_bs + oi);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = mm[q] > 0.f ? v[q] : v[q] * p.m_slope;
      }
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
    return v;
  };
  auto sign_nibble = [](const f32x4& v, const int t0) -> unsigned {   // this float4's sign bits at their place in the 32-step word
    return ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u)) << (t0 & 31);
  };

#pragma unroll
  for (int m = 0; m < M_REP; ++m) {
    const int co = r0 + wrow0 + m * 16 + ln;
    if (co >= p.Cout) continue;
    float bias = 0.f;
    if (EPI == EPI_FWD && p.bias) bias = p.bias[co];
    unsigned sb_word = 0;
#pragma unroll
    for (int n = 0; n < N_REP; ++n) {
      int t0 = n0 + wcol0 + n * 16 + kq * 4, bb = b;
      if (FOLD) { const int sg = t0 / p.T; t0 -= sg * p.T; bb = b + sg; if (bb >= p.Bn) continue; }
      if (t0 >= p.T) continue;
      const f32x4 v = epi_store(acc[m][n], co, t0, bias, bb);
      if (EPI == EPI_FWD && N_REP >= 2 && p.sbits) {
        // sign bits of the stored values: the 4 lanes (kq = 0..3) of a channel are OR-ed with two wave shuffles, and sub-tiles
        // (n, n+1) make one 32-bit word (the wave's first column is a multiple of 32: 16 * N_REP columns per wave, N_REP even)
        const unsigned h = sign_nibble(v, t0);
        sb_word = (n & 1) ? (sb_word | h) : h;
        if (n & 1) {
          unsigned w = sb_word;
          w |= (unsigned)__shfl_xor((int)w, 16);
          w |= (unsigned)__shfl_xor((int)w, 32);
          if (kq == 0) p.sbits[(long)b * p.sb_bs + (long)co * (p.T >> 5) + (t0 >> 5)] = w;
        }
      }
    }
  }
  PROF(6)
  PROF_END
}

// ------------------------------------------------------------------------------------------ host
// Row-walk geometry of a tile of `rows` rows x `nvec` float4: rows per pass and number of passes (conv_common.h RowWalk).
static inline void walk_geometry(int rows, int nvec, int* rp, int* np) {
  *rp = nvec <= 256 ? 256 / nvec : 0;
  if (*rp > rows) *rp = rows;
  *np = *rp ? (rows + *rp - 1) / *rp : 1 << 20;
}

template <int M_REP, int N_REP, int WM, int WN, int XFK, int EPI, bool FOLD = false>
static hipError_t lean_launch3(const LeanP& p, int B, hipStream_t st) {
  constexpr int MT = 16 * M_REP * WM, NT = 16 * N_REP * WN;
  auto k = conv_lean_kernel<M_REP, N_REP, WM, WN, XFK, EPI, FOLD>;
  TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
  dim3 grid(FOLD ? 1 : (p.T + NT - 1) / NT, (p.Cout + MT - 1) / MT, FOLD ? (B + p.fold - 1) / p.fold : B);
  LeanP q = p; q.swz = g_knob[0] && (long)grid.x * grid.y * grid.z >= 16;
  const size_t lds = (size_t)(p.xnp * p.xrp * p.XS + p.wnp * p.wrp * p.WS) * sizeof(float);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, q);
  return hipGetLastError();
}

// folded short-sequence launches (64-column tiles only): the pairs discriminator layer 5 uses
template <int M_REP, int N_REP, int WM, int WN>
static hipError_t lean_launch_fold(const LeanP& p, int B, int xfk, int epi, hipStream_t st) {
  if (xfk == LXF_ACT && epi == EPI_FWD) return lean_launch3<M_REP, N_REP, WM, WN, LXF_ACT, EPI_FWD, true>(p, B, st);
  if (xfk == LXF_ACT && epi == EPI_PLAIN) return lean_launch3<M_REP, N_REP, WM, WN, LXF_ACT, EPI_PLAIN, true>(p, B, st);
  if (xfk == LXF_MASK_LRELU && epi == EPI_PLAIN) return lean_launch3<M_REP, N_REP, WM, WN, LXF_MASK_LRELU, EPI_PLAIN, true>(p, B, st);
  return hipErrorNotSupported;
}

template <int M_REP, int N_REP, int WM, int WN>
static hipError_t lean_launch2(const LeanP& p, int B, int xfk, int epi, hipStream_t st) {
  // the (prologue, epilogue) pairs the train step actually uses; anything else goes to the generic kernel
  if (xfk == LXF_ACT && epi == EPI_FWD) return lean_launch3<M_REP, N_REP, WM, WN, LXF_ACT, EPI_FWD>(p, B, st);
  if (xfk == LXF_FILM && epi == EPI_FWD) return lean_launch3<M_REP, N_REP, WM, WN, LXF_FILM, EPI_FWD>(p, B, st);
  if (xfk == LXF_ACT && epi == EPI_MASK) return lean_launch3<M_REP, N_REP, WM, WN, LXF_ACT, EPI_MASK>(p, B, st);
  if (xfk == LXF_ACT && epi == EPI_FILM) return lean_launch3<M_REP, N_REP, WM, WN, LXF_ACT, EPI_FILM>(p, B, st);
  if (xfk == LXF_ACT && epi == EPI_PLAIN) return lean_launch3<M_REP, N_REP, WM, WN, LXF_ACT, EPI_PLAIN>(p, B, st);
  if (xfk == LXF_MASK_LRELU && epi == EPI_PLAIN) return lean_launch3<M_REP, N_REP, WM, WN, LXF_MASK_LRELU, EPI_PLAIN>(p, B, st);
  return hipErrorNotSupported;
}

// FiLM conditioning forward: gb = cond_var.2(lrelu(cond_var.0(c))) with the cond_var.0 output computed per tile in
// LDS (LXF_COND). p.x = excitation [B][Cv][T]; p.Cin = channels of cv0 (= Cin of cond_var.2); p.w = cond_var.2 weight.
hipError_t launch_conv_lean_cond(LeanP p, int B, hipStream_t st) {
  if ((p.T & 3) || !p.vec || p.K != 3 || p.d != 1 || p.pad != 1 || p.Cv * 3 > 24) return hipErrorNotSupported;
  const int MT = p.Cout <= 32 ? 32 : 64, NT = 256;
  p.mirror = 0; p.flip = 0; p.reflect = 0;
  const int first = -p.pad;
  const int lo = -4;                                   // aligned origin
  p.lo = lo; p.i0 = first - lo;
  p.span = ((NT + (p.K - 1) * p.d - p.pad - lo) + 3) / 4 * 4;
  p.XS = ((p.span + 15) / 32) * 32 + 16;           // smallest stride >= span that is 16 (mod 32): the 4 rows of a fragment read hit disjoint banks
  p.ES = ((p.span + 15) / 16) * 16 + 4;
  int Cc = 32;
  auto geom = [&](int cc) {
    p.Cc = cc; p.WS = p.K * cc + 2;
    p.xrp = cc; p.xnp = 1;                               // the input tile is computed, not loaded: Cc rows
    walk_geometry(MT, p.K * cc / 4, &p.wrp, &p.wnp);
    return (size_t)(cc * p.XS + p.wnp * p.wrp * p.WS + p.Cv * p.ES + cc * 25) * sizeof(float);
  };
  size_t lds = geom(Cc);
  if (lds > (size_t)(g_lds_cap > 0 ? g_lds_cap : 80 * 1024) || p.wnp > (MT >= 48 ? 10 : 6)) lds = geom(Cc = 16);
  dim3 grid((p.T + NT - 1) / NT, (p.Cout + MT - 1) / MT, B);
  p.swz = g_knob[0] && (long)grid.x * grid.y * grid.z >= 16;
  if (MT == 32) {
    auto k = conv_lean_kernel<2, 4, 1, 4, LXF_COND, EPI_FWD>;
    TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  } else {
    auto k = conv_lean_kernel<4, 4, 1, 4, LXF_COND, EPI_FWD>;
    TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  }
  return hipGetLastError();
}

// Returns hipErrorNotSupported when the shape/variant is outside the lean kernel's contract.
hipError_t launch_conv_lean(LeanP p, int B, int xfk, int epi, hipStream_t st) {
  const bool combo = (xfk == LXF_ACT) || (xfk == LXF_FILM && epi == EPI_FWD) || (xfk == LXF_MASK_LRELU && epi == EPI_PLAIN);
  if (!combo) return hipErrorNotSupported;
  // Tile choice: the largest tile whose grid still covers the chip ~2x (256 CUs), else the smallest one.
  struct Cand { int MT, NT, cfg; };
  static const Cand cands[] = {{64, 256, 2}, {48, 256, 5}, {32, 256, 1}, {64, 64, 4}, {32, 64, 6}, {16, 256, 0}, {16, 64, 3}};
  static const Cand tall = {144, 64, 7};     // all 136 rows of cond_var.2's input-grad / cond_var.0 in one block: the input tile is read once
  const int R = p.Cout;
  int MT = 16, NT = 64, cfg = 3;
  for (const Cand& c : cands) {
    if (c.MT > 16 && R <= 16) continue;
    if (c.MT > 32 && R <= 32) continue;
    if (c.MT == 48 && !(R % 64 != 0 && (R + 47) / 48 * 48 < (R + 63) / 64 * 64)) continue;
    if (c.MT == 64 && c.NT == 256 && (R % 64 != 0 && (R + 47) / 48 * 48 < (R + 63) / 64 * 64)) continue;
    if (c.NT == 256 && p.T <= 80) continue;
    const long blocks = (long)((R + c.MT - 1) / c.MT) * ((p.T + c.NT - 1) / c.NT) * B;
    MT = c.MT; NT = c.NT; cfg = c.cfg;
    // 64 x 64 blocks sit 4 to a CU: below one full round of them (D layer 5: exactly 1024) the 32 x 64 tile, with twice
    // the blocks, fills the chip better (128 -> 128 k7 T = 500: 47.7 vs 51.0 us; 136 -> 256 input-grad: 55 vs 61)
    if (blocks >= (c.cfg == 4 ? 1024 : 512)) break;
  }
  // a tiny reduction (the 8-channel excitation window of cond_var.0: Cin*K = 24) makes the conv a pure HBM stream of its
  // output: the 16-row tile keeps the most blocks resident (94 vs 111 us for 8 -> 136, T = 16000, 32 samples)
  if ((long)p.Cin * p.K <= 48 && p.T > 80 && p.mirror == 0) { MT = 16; NT = 256; cfg = 0; }
  // short reductions with many (R = 136: the input-grad of cond_var.2 at the 16/32-channel stages) or few (R = 32)
  // output rows are bound by their epilogue traffic, not by the matrix pipe: 16-row tiles (5 blocks per CU) measured
  // 4-10 % faster than the 48/32-row ones there (profiles/r02_b_tile_sweep.txt)
  if ((long)p.Cin * p.K <= 224 && p.T > 80 && (R >= 128 || R == 32)) { MT = 16; NT = 256; cfg = 0; }
  if (g_force_tile >= 0) {   // test-only (tdvc_debug_force_tile): pin the tile so that small shapes reach every instance
    for (const Cand& c : cands)
      if (c.cfg == g_force_tile) { MT = c.MT; NT = c.NT; cfg = c.cfg; }
    if (g_force_tile == tall.cfg) { MT = tall.MT; NT = tall.NT; cfg = tall.cfg; }
  }
  if (p.sbits || p.mbits) {   // sign-bit words span two 16-column sub-tiles of one wave: not on the tiles with one sub-tile per wave
    if (!p.vec || (p.T & 31) || p.T <= 80) return hipErrorNotSupported;
    if (cfg == 3 || cfg == tall.cfg) { MT = 16; NT = 256; cfg = 0; }
  }
  const int first = -p.pad;
  int lo = -p.pad - p.mirror;
  lo = -(((-lo) + 3) / 4 * 4);
  const int hi = (p.K - 1) * p.d - p.pad + p.mirror;
  p.lo = lo; p.i0 = first - lo;
  p.span = ((NT + hi - lo) + 3) / 4 * 4;
  // short sequences: several samples per 64-column tile (kernel comment at FOLD)
  p.fold = 1; p.seg = 0; p.Bn = B;
  const bool fold_combo = (xfk == LXF_ACT && (epi == EPI_FWD || epi == EPI_PLAIN)) || (xfk == LXF_MASK_LRELU && epi == EPI_PLAIN);
  if (g_knob[4] == 0 && g_force_tile < 0 && NT == 64 && R >= 32 && (p.T == 16 || p.T == 32) && p.Cin % 16 == 0 && B >= 64 / p.T && p.vec && !p.reflect && p.mirror == 0 && fold_combo &&
      !p.sbits && !p.mbits && (!p.aux || p.aux_bs == p.x_bs) && (long)(64 / p.T) * p.x_bs < (1L << 28)) {
    p.fold = 64 / p.T;
    p.seg = ((p.T + hi - lo) + 3) / 4 * 4;
    p.span = p.fold * p.seg;
    // a quarter / half as many blocks as unfolded: take the 32-row tile unless the 64-row one still fills a round of 1024
    const long nb64 = (long)((R + 63) / 64) * ((B + p.fold - 1) / p.fold);
    if (nb64 >= 1024) { MT = 64; cfg = 4; } else { MT = 32; cfg = 6; }
  }
  p.XS = ((p.span + 15) / 32) * 32 + 16;           // smallest stride >= span that is 16 (mod 32): the 4 rows of a fragment read hit disjoint banks
  const int xvp = (MT >= 144 || NT <= 64) ? 4 : (MT >= 32 ? 12 : 6), wvp = MT >= 48 ? 10 : (MT >= 32 ? 6 : 4);   // = the kernel's XVP / WVP
  // LDS budget per block = what lets the blocks the register budget allows (launch_bounds of the instance) actually be
  // resident on a CU with 160 KB: 3 blocks -> 52 KB. With 64 KB only two 32-row x 256-column blocks fit and that kernel
  // ran 22 % slower; the other tiles gain 0-6 % (tools/tile_sweep.py, profiles/r02_b_tile_sweep.txt).
  const size_t lds_cap = g_lds_cap > 0 ? (size_t)g_lds_cap : (size_t)((cfg == 4 ? 40 : 52) * 1024);   // 4 resident 64 x 64 blocks
  int Cc = 0;
  // 16-row tiles over a long reduction (the discriminators' 1024 -> 16 heads: 32 serial chunks of 32 channels on 32-64 blocks, each
  // chunk one exposed load latency: 77 us for 0.2 GFLOP) take chunks of up to 64 channels when the prefetch registers hold them
  const int cc_cap = (MT == 16 && p.Cin >= 256) ? 64 : 32;
  for (int cc = 4; cc <= cc_cap && cc <= ((p.Cin + 3) / 4) * 4; cc += 4) {
    int xrp, xnp, wrp, wnp;
    walk_geometry(cc, p.span / 4, &xrp, &xnp);
    walk_geometry(MT, p.K * cc / 4, &wrp, &wnp);
    const size_t lds = (size_t)(xnp * xrp * p.XS + wnp * wrp * (p.K * cc + 2)) * 4;
    if (lds > lds_cap || wnp > wvp || (xfk == LXF_ACT && xnp > xvp)) continue;
    if (p.fold > 1 && p.Cin % cc) continue;          // folded tiles: whole channel chunks only (a partial one would run into the next sample)
    Cc = cc; p.xrp = xrp; p.xnp = xnp; p.wrp = wrp; p.wnp = wnp;
  }
  if (!Cc) return hipErrorNotSupported;                  // weight tile would not fit the register prefetch
  p.Cc = Cc;
  p.WS = p.K * Cc + 2;
  if (p.fold > 1) return cfg == 4 ? lean_launch_fold<1, 4, 4, 1>(p, B, xfk, epi, st) : lean_launch_fold<1, 2, 2, 2>(p, B, xfk, epi, st);
  switch (cfg) {
    case 0: return lean_launch2<1, 4, 1, 4>(p, B, xfk, epi, st);
    case 1: return lean_launch2<2, 4, 1, 4>(p, B, xfk, epi, st);
    case 2: return lean_launch2<4, 4, 1, 4>(p, B, xfk, epi, st);
    case 3: return lean_launch2<1, 1, 1, 4>(p, B, xfk, epi, st);
    case 5: return lean_launch2<3, 4, 1, 4>(p, B, xfk, epi, st);
    case 6: return lean_launch2<1, 2, 2, 2>(p, B, xfk, epi, st);
    case 7: return lean_launch2<9, 1, 1, 4>(p, B, xfk, epi, st);
    default: return lean_launch2<1, 4, 4, 1>(p, B, xfk, epi, st);
  }
}

}  // namespace tdvc
