// Forward of a 3-tap stride-1 'same' conv with 65..160 input channels (FiLM's cond_var.2, model/generator.py:86-92: 136 -> 2C,
// ~9 % of the train step) on the bf16 matrix pipe at fp32 accuracy -- the split-bf16 x6 scheme of conv_wgrad_x6.hip (three
// exact bf16 pieces per operand, six piece products on v_mfma_f32_16x16x32_bf16, fp32 accumulation).
//     y[co][t] = bias[co] + sum_{ci,j} W[co][ci][j] * x'[ci][t + j - 1],      x' = LeakyReLU(x) (or x)
// K of the product is (tap, input channel): a lane's 8 consecutive k values are 8 consecutive CHANNELS at one time step, so the
// activation tile is held TRANSPOSED in LDS ([time][channel], channel fastest). The transposition happens in registers on the
// way in: every thread loads 4 channel rows x 4 consecutive steps (coalesced float4 loads), splits the 16 values and writes,
// per step and piece, its 4 channels as one 8-byte LDS store. The weights arrive pre-split and already in the order of the
// LDS image (tdvc_conv_x6_weight_planes: once per optimizer step), so a chunk is copied with linear 16-byte loads.
// Roles in the MFMA: A = weights (rows = output channels), B = activations (columns = time). The columns of sub-tile n are
// the steps 4l + n (l = 0..15), so that a lane's four accumulators of one output channel are four CONSECUTIVE steps and a
// store instruction covers 4 channel rows x 256 B (16 rows x 64 B with the natural column order). For conflict-free fragment
// reads under that permutation the activation tile is kept as four phase planes (step mod 4).
// Block = 128 steps x (32 | 64) output channels, reduction in chunks of 32 input channels (x all 3 taps); wave w owns steps
// 64 (w & 1) .. + 63 x half of the output-channel tiles. Loads of chunk c + 1 are in flight during the MFMAs of chunk c.
#include "conv_common.h"
#include "api_util.h"

PROF_DEFINE(tdvc_debug_fwdx6_prof)

namespace tdvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct FwdX6P {
  const float* x; long x_bs;            // [B][Cin][T]
  const unsigned short* wp;             // [Cout / MT][5 chunks][MT / 32 records][3 pieces][32 co][3 taps][32 ci] bf16, MT = 64 | 32 (x6_mt)
  const float* bias;                    // [Cout] or null
  float* y; long y_bs;
  int T, Cin, Cout;
  float slope;                          // LeakyReLU slope of the prologue (1 = none)
};

constexpr int FX_NT = 128;              // steps per block
constexpr int FX_CP = 160;              // padded channel count of the weight planes (5 chunks of 32)
constexpr int FX_RS = 32;               // bf16 row length of both LDS images: 64 B, no padding; the 16-byte slot q of row r lives at slot q ^ ((r >> 1) & 3)
constexpr int FX_PR = 34;               // rows of one phase plane: window index s = position - (n0 - 4) = 4 row + phase, s in [3, 132]
constexpr int FX_XPL = 4 * FX_PR * FX_RS;   // one piece of the activation tile: [phase][row][FX_RS]
constexpr int FX_WREC = 3 * 32 * 3 * 4;     // 16-byte vectors of one weight record (32 output channels x one chunk): [piece][co][tap][4]

// exact: f = h + m + l, each piece the UPPER half of its word (l's lower half is zero by construction: 24 significant bits in all;
// it is left unmasked and dropped by the pack)
__device__ __forceinline__ void split1(const float f, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, f) & 0xffff0000u;
  const float r1 = f - __builtin_bit_cast(float, h);
  m = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
  const float r2 = r1 - __builtin_bit_cast(float, m);
  l = __builtin_bit_cast(unsigned, r2);
}
// LDS element offset of slot q (8 bf16) of row r. ds_read_b128 serves a wave in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, ...:
// MI355X_MICROARCH.md, LDS), so a group mixes rows of two k-quarters: padded 80-byte rows made every group 2-way conflicting
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50 in profiles/r03_pmc.txt). With this XOR any 16 consecutive rows read by lanes
// (row = base + (lane & 15), q = lane >> 4) touch 16 distinct 4-bank slots in every group, for every base (tools/lds_swizzle_check.py).
__device__ __forceinline__ int swz(int r, int q) { return r * FX_RS + 8 * (q ^ ((r >> 1) & 3)); }
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) {      // {upper half of a, upper half of b}: a in the low 16 bits
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
}

template <int CO_TILES>
__global__ __launch_bounds__(256, CO_TILES == 2 ? 3 : 2) void conv_fwd_x6_kernel(const FwdX6P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  constexpr int MT = 16 * CO_TILES;
  constexpr int CB = MT / 32;                         // weight records per chunk
  constexpr int CT2 = CO_TILES / 2;                   // output-channel tiles of one wave
  constexpr int WPL = MT * 3 * FX_RS;                 // elements of one weight piece in LDS: [tap][co][FX_RS] (swizzled rows)
  constexpr int WNV = CB * FX_WREC;                   // 16-byte vectors of a weight chunk
  constexpr int WPT = (WNV + 255) / 256;
  unsigned short* xt = smem16;                        // [3 pieces][4 phases][FX_PR][FX_RS]
  unsigned short* ws = smem16 + 3 * FX_XPL;           // [3 pieces][3 taps][MT][FX_RS]; both images start on multiples of 8 rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wt = wave & 1, wc = wave >> 1;
  const int ln = lane & 15, g = lane >> 4;
  const int T = p.T, n0 = blockIdx.x * FX_NT, r0 = blockIdx.y * MT, b = blockIdx.z;
  const int nchunk = (p.Cin + 31) >> 5;

  // staging roles. x: thread (channel group c4 of 4, float4 column xv) -> steps n0 + 4 xv .. + 3; threads 0..63 also carry one halo
  // element each (channel tid & 31 at position n0 - 1 | n0 + 128); weights: vector e = tid + 256 i of the chunk's records
  const int c4 = tid >> 5, xv = tid & 31;
  const bool xin = n0 + 4 * xv < T;                    // T % 4 == 0: a float4 lies wholly inside or outside
  const int hpos = (tid & 32) ? n0 + FX_NT : n0 - 1;
  const bool hin = tid < 64 && hpos >= 0 && hpos < T;
  const srd_t xrs = make_srd(p.x + (long)b * p.x_bs, p.Cin * T * 4);         // (channels >= Cin are masked per load: the scalar chunk offset may not take part in the range check)
  const srd_t wrs = make_srd(reinterpret_cast<const float*>(p.wp), (p.Cout >> 5) * 5 * FX_WREC * 16);

  f32x4 acc[CT2][4];
#pragma unroll
  for (int ct = 0; ct < CT2; ++ct)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[ct][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float bias[CT2][4];                                 // loaded up front: a dependent global load in the epilogue costs its full latency
#pragma unroll
  for (int ct = 0; ct < CT2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = r0 + 16 * (wc * CT2 + ct) + 4 * g + r;
      bias[ct][r] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    }

  const int xo = (4 * c4 * T + n0 + 4 * xv) * 4, ho = ((tid & 31) * T + hpos) * 4;      // byte offsets inside a chunk's 32 channel rows
  int wl[WPT];                                        // LDS element offset of the thread's i-th weight vector (chunk invariant)
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int e = tid + i * 256;
    const int cb = e / FX_WREC, er = e - cb * FX_WREC;
    const int row = er >> 2, pc = row / 96, rr = row - pc * 96;            // rr = co * 3 + tap
    const int co = rr / 3, j = rr - co * 3;
    wl[i] = pc * WPL + swz(j * MT + cb * 32 + co, er & 3);
  }

  f32x4 xr[4];
  float hr;
  u32x4 wr[WPT];
  auto issue = [&](int c) {
    const int c0 = c * 32;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      xr[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (xin && c0 + 4 * c4 + e < p.Cin) ? xo + e * T * 4 : 0x7f000000, c0 * T * 4, 0));
    hr = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (hin && c0 + (tid & 31) < p.Cin) ? ho : 0x7f000000, c0 * T * 4, 0));
#pragma unroll
    for (int i = 0; i < WPT; ++i)                     // the block's chunk is one linear run of WNV vectors
      wr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (tid + i * 256 < WNV) ? (tid + i * 256) * 16 : 0x7f000000,
                                                                              (blockIdx.y * 5 + c) * (WNV * 16), 0));
  };
  PROF_DECL
  issue(0);
  PROF(0)

  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                                  // the previous chunk's fragments are consumed
    PROF(1)
    PROF_WAITV()
    PROF(2)
    // 4 channels x 4 steps in registers -> per step and piece one 8-byte store of the 4 channels. Step n0 + 4 xv + k is window index
    // 4 (xv + 1) + k: phase k, row xv + 1
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float f0 = xr[e][k];
        const float f = fmaxf(f0, f0 * p.slope);      // slope in (0, 1]: LeakyReLU; 1: identity
        split1(f, hh[e], mm[e], ll[e]);
      }
      const int off = swz(k * FX_PR + xv + 1, c4 >> 1) + 4 * (c4 & 1);
      *reinterpret_cast<u32x2*>(xt + 0 * FX_XPL + off) = (u32x2){pack_hi(hh[0], hh[1]), pack_hi(hh[2], hh[3])};
      *reinterpret_cast<u32x2*>(xt + 1 * FX_XPL + off) = (u32x2){pack_hi(mm[0], mm[1]), pack_hi(mm[2], mm[3])};
      *reinterpret_cast<u32x2*>(xt + 2 * FX_XPL + off) = (u32x2){pack_hi(ll[0], ll[1]), pack_hi(ll[2], ll[3])};
    }
    if (tid < 64) {                                   // halo: window index 3 (phase 3, row 0) | 132 (phase 0, row 33)
      unsigned h, m, l;
      split1(fmaxf(hr, hr * p.slope), h, m, l);
      const int off = swz((tid & 32) ? (FX_PR - 1) : 3 * FX_PR, (tid & 31) >> 3) + (tid & 7);
      xt[0 * FX_XPL + off] = (unsigned short)(h >> 16);
      xt[1 * FX_XPL + off] = (unsigned short)(m >> 16);
      xt[2 * FX_XPL + off] = (unsigned short)(l >> 16);
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i)
      if (tid + i * 256 < WNV) *reinterpret_cast<u32x4*>(ws + wl[i]) = wr[i];
    PROF(3)
    __syncthreads();
    PROF(4)
    if (c + 1 < nchunk) issue(c + 1);
    PROF(5)

    // ---- one k-block (32 channels) per tap; lane (ln, g): k = 8g .. 8g + 7. Column ln of sub-tile n is step 64 wt + 4 ln + n, which
    // at tap j reads window index 64 wt + 4 ln + (n + j + 3): phase (n + j + 3) & 3, row 16 wt + ln + ((n + j + 3) >> 2)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      bf16x8 af[CT2][3];
#pragma unroll
      for (int ct = 0; ct < CT2; ++ct)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          af[ct][pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ws + pc * WPL + swz(j * MT + 16 * (wc * CT2 + ct) + ln, g)));
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int m = n + j + 3;
        bf16x8 bf[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          bf[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xt + pc * FX_XPL + swz((m & 3) * FX_PR + 16 * wt + ln + (m >> 2), g)));
#pragma unroll
        for (int ct = 0; ct < CT2; ++ct) {            // smallest products first
          f32x4 cc = acc[ct][n];
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct][2], bf[0], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct][0], bf[2], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct][1], bf[1], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct][1], bf[0], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct][0], bf[1], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct][0], bf[0], cc, 0, 0, 0);
          acc[ct][n] = cc;
        }
      }
    }
    PROF(6)
  }

  // ---- epilogue: D[co = 4g + r][column ln] of sub-tiles n = 0..3 -> 4 consecutive steps of one output channel per lane and register
  const int t0 = n0 + 64 * wt + 4 * ln;
#pragma unroll
  for (int ct = 0; ct < CT2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = r0 + 16 * (wc * CT2 + ct) + 4 * g + r;
      if (co < p.Cout && t0 < T) {
        const float bs = bias[ct][r];
        const float v0 = acc[ct][0][r], v1 = acc[ct][1][r], v2 = acc[ct][2][r], v3 = acc[ct][3][r];
        *reinterpret_cast<f32x4*>(p.y + (long)b * p.y_bs + (long)co * T + t0) = (f32x4){v0 + bs, v1 + bs, v2 + bs, v3 + bs};
      }
    }
  PROF(8)
  PROF_END
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same forward with cond_var.0's excitation window computed on the fly (FiLM conditioning, model/generator.py:86-92):
//     cv0[c][t] = k3[c][t == 0 | interior | t == T - 1] + sum_{ce < 8, j} W0x[c][ce][j] * exc[ce][t + j - 1]
//     gb = cond_var.2(LeakyReLU(cv0))
// Per 32-channel chunk of the reduction the block computes its own cv0 tile (fp32 MFMA, K = 24) straight into the accumulator layout
// of the permuted sub-tiles, so that (a) the fp32 intermediate goes to HBM once, as 4 rows x 256 B stores, for the backward pass,
// (b) its sign bits are packed in registers, (c) LeakyReLU + the bf16 split feed the x6 product's LDS tile: the 136-channel tensor is
// never read back by the forward (two launches: written by one, read by the next). The B operand of the small product (the
// excitation window) does not depend on the chunk and stays in registers.
struct CondX6P {
  const float* exc; long exc_bs;        // [B][8][T]
  const float* w0; int w0_rs, w0_off;   // W0x[c][k] = w0[c * w0_rs + w0_off + k], k = ce * 3 + j: 24 contiguous floats per row
  const float* k3;                      // [B][nc][3], cond_var.0's bias included
  const unsigned short* wp;             // cond_var.2 weight pieces (tdvc_conv_x6_weight_planes)
  const float* bias;                    // cond_var.2 bias or null
  float* cv0; long cv0_bs;              // [B][nc][T] or null
  unsigned* bits; long bits_bs;         // [B][nc][T / 32] or null (T % 32 == 0)
  float* y; long y_bs;                  // gb [B][Cout][T]
  int T, Cin, Cout;                     // Cin = nc
  float slope;
};
constexpr int FX_ES = 140;               // row stride of the excitation tile: window [n0 - 4, n0 + 132) + 4 (rows 16-byte aligned)

template <int CO_TILES>
__global__ __launch_bounds__(256, CO_TILES == 2 ? 3 : 2) void film_cond_fwd_x6_kernel(const CondX6P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  constexpr int MT = 16 * CO_TILES;
  constexpr int CB = MT / 32;
  constexpr int CT2 = CO_TILES / 2;
  constexpr int WPL = MT * 3 * FX_RS;
  constexpr int WNV = CB * FX_WREC;
  constexpr int WPT = (WNV + 255) / 256;
  unsigned short* xt = smem16;                        // [3 pieces][4 phases][FX_PR][FX_RS]
  unsigned short* ws = smem16 + 3 * FX_XPL;           // [3 pieces][3 taps][MT][FX_RS]
  float* es = reinterpret_cast<float*>(ws + 3 * WPL); // [8][FX_ES] excitation window [n0 - 4, n0 + 132), fp32
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wt = wave & 1, wc = wave >> 1;            // x6 product: step half, output-channel half
  const int mt = wave & 1, grp = wave >> 1;           // cv0 product: 16-channel tile of the chunk, 64-step group
  const int ln = lane & 15, g = lane >> 4;
  const int T = p.T, n0 = blockIdx.x * FX_NT, r0 = blockIdx.y * MT, b = blockIdx.z;
  const int nc = p.Cin, nchunk = (nc + 31) >> 5;
  const bool writer = blockIdx.y == 0;                // every output-channel block computes cv0; one stores it
  const srd_t wrs = make_srd(reinterpret_cast<const float*>(p.wp), (p.Cout >> 5) * 5 * FX_WREC * 16);

  // ---- excitation tile -> LDS, phase planes
  {
    const srd_t ers = make_srd(p.exc + (long)b * p.exc_bs, 8 * T * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * 256;
      if (e < 8 * 34) {
        const int row = e / 34, v = e - row * 34, q = n0 - 4 + 4 * v;
        const f32x4 x4 = buf_load4(ers, (q >= 0 && q < T) ? (row * T + q) * 4 : 0x7f000000);
        *reinterpret_cast<f32x4*>(es + row * FX_ES + 4 * v) = x4;
      }
    }
  }

  f32x4 acc[CT2][4];
#pragma unroll
  for (int ct = 0; ct < CT2; ++ct)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[ct][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bias[CT2][4];
#pragma unroll
  for (int ct = 0; ct < CT2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = r0 + 16 * (wc * CT2 + ct) + 4 * g + r;
      bias[ct][r] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    }
  int wl[WPT];
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int e = tid + i * 256;
    const int cb = e / FX_WREC, er = e - cb * FX_WREC;
    const int row = er >> 2, pc = row / 96, rr = row - pc * 96;
    const int co = rr / 3, j = rr - co * 3;
    wl[i] = pc * WPL + swz(j * MT + cb * 32 + co, er & 3);
  }

  __syncthreads();                                    // the excitation tile is in LDS
  // B operand of the cv0 product, chunk invariant: lane (column ln, k-lane g) of k-step s holds E'[k = 4s + g][step], k = ce * 3 + j,
  // E'[(ce, j)][t] = exc[ce][t + j - 1]. Column ln of sub-tile n is step n0 + 64 grp + 4 ln + n = window index 64 grp + 4 ln + (n + j + 3)
  // (re-read from LDS every chunk: 24 registers less let a third block of the 32-channel variant be resident; the stride-4 lane
  // pattern is 2..4-way conflicting, 24 dword reads per chunk)
  int eb[6];
  float eh[6];
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int k = 4 * s + g, ce = k / 3, j = k - 3 * ce;
    eb[s] = ce * FX_ES + 64 * grp + 4 * ln + j + 3;
    // halo columns (waves of group 0): column 0 = step n0 - 1 (window index j + 2), column 1 = step n0 + 128 (window index j + 131)
    eh[s] = es[ce * FX_ES + ((ln == 1) ? j + 131 : j + 2)];
  }

  const srd_t w0rs = make_srd(p.w0, (nc - 1) * p.w0_rs * 4 + (p.w0_off + 24) * 4);
  const srd_t k3rs = make_srd(p.k3 + (long)b * nc * 3, nc * 3 * 4);
  float aw[6];
  f32x4 k3v[3];
  u32x4 wr[WPT];
  auto issue = [&](int c) {
    const int ch = c * 32 + 16 * mt + ln;             // A operand row of this lane
#pragma unroll
    for (int s = 0; s < 6; ++s)
      aw[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(w0rs, ch < nc ? (ch * p.w0_rs + p.w0_off + 4 * s + g) * 4 : 0x7f000000, 0, 0));
    const int c4 = c * 32 + 16 * mt + 4 * g;          // this lane's 4 output channels: 12 contiguous floats of k3
#pragma unroll
    for (int i = 0; i < 3; ++i) k3v[i] = buf_load4(k3rs, c4 < nc ? (c4 * 3 + 4 * i) * 4 : 0x7f000000);
#pragma unroll
    for (int i = 0; i < WPT; ++i)
      wr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (tid + i * 256 < WNV) ? (tid + i * 256) * 16 : 0x7f000000,
                                                                              (blockIdx.y * 5 + c) * (WNV * 16), 0));
  };
  issue(0);

  const int pos0 = n0 + 64 * grp + 4 * ln;            // first of this lane's 4 consecutive steps
  const int hpos = (ln == 1) ? n0 + FX_NT : n0 - 1;
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                                  // the previous chunk's fragments are consumed
    // ---- cv0 chunk: D[channel 4g + r][step column ln] per sub-tile n
    f32x4 cv[4], ch4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 4; ++n) cv[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 6; ++s) {
#pragma unroll
      for (int n = 0; n < 4; ++n) cv[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[s], es[eb[s] + n], cv[n], 0, 0, 0);
      if (grp == 0) ch4 = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[s], eh[s], ch4, 0, 0, 0);
    }
    const int cbase = c * 32 + 16 * mt + 4 * g;       // channels cbase + r
    const float kf[12] = {k3v[0][0], k3v[0][1], k3v[0][2], k3v[0][3], k3v[1][0], k3v[1][1], k3v[1][2], k3v[1][3],
                          k3v[2][0], k3v[2][1], k3v[2][2], k3v[2][3]};
    float v[4][4];                                    // [n][r]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float c0v = cv[0][r], c1v = cv[1][r], c2v = cv[2][r], c3v = cv[3][r];
      const float cc[4] = {c0v, c1v, c2v, c3v};
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int pos = pos0 + n;
        const float bs = pos == 0 ? kf[3 * r] : (pos == T - 1 ? kf[3 * r + 2] : kf[3 * r + 1]);
        v[n][r] = cc[n] + bs;
      }
    }
    const bool own = pos0 < T;                        // T % 4 == 0: the lane's 4 steps are inside or outside together
    if (writer && own && cbase < nc) {
      if (p.cv0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<f32x4*>(p.cv0 + (long)b * p.cv0_bs + (long)(cbase + r) * T + pos0) = (f32x4){v[0][r], v[1][r], v[2][r], v[3][r]};
      }
      if (p.bits) {                                   // word = 32 steps of one channel = the nibbles of 8 consecutive lanes
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          unsigned w = (v[0][r] > 0.f ? 1u : 0u) | (v[1][r] > 0.f ? 2u : 0u) | (v[2][r] > 0.f ? 4u : 0u) | (v[3][r] > 0.f ? 8u : 0u);
          w <<= 4 * (ln & 7);
          w |= __shfl_xor(w, 1); w |= __shfl_xor(w, 2); w |= __shfl_xor(w, 4);
          if ((ln & 7) == 0) p.bits[(long)b * p.bits_bs + (long)(cbase + r) * (T >> 5) + ((n0 + 64 * grp) >> 5) + (ln >> 3)] = w;
        }
      }
    }
    // LeakyReLU, zero outside the sequence (cond_var.2's zero padding), split, 8-byte store of the lane's 4 channels per step
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float f0 = own ? v[n][r] : 0.f;
        split1(fmaxf(f0, f0 * p.slope), hh[r], mm[r], ll[r]);
      }
      const int off = swz(n * FX_PR + 16 * grp + ln + 1, 2 * mt + (g >> 1)) + 4 * (g & 1);
      *reinterpret_cast<u32x2*>(xt + 0 * FX_XPL + off) = (u32x2){pack_hi(hh[0], hh[1]), pack_hi(hh[2], hh[3])};
      *reinterpret_cast<u32x2*>(xt + 1 * FX_XPL + off) = (u32x2){pack_hi(mm[0], mm[1]), pack_hi(mm[2], mm[3])};
      *reinterpret_cast<u32x2*>(xt + 2 * FX_XPL + off) = (u32x2){pack_hi(ll[0], ll[1]), pack_hi(ll[2], ll[3])};
    }
    if (grp == 0 && ln < 2) {                         // halo steps: window index 3 (phase 3, row 0) | 132 (phase 0, row 33)
      const bool hin = hpos >= 0 && hpos < T;
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float bs = hpos == 0 ? kf[3 * r] : (hpos == T - 1 ? kf[3 * r + 2] : kf[3 * r + 1]);
        const float hv = ch4[r];
        const float f0 = hin ? hv + bs : 0.f;
        split1(fmaxf(f0, f0 * p.slope), hh[r], mm[r], ll[r]);
      }
      const int off = swz(ln == 1 ? FX_PR - 1 : 3 * FX_PR, 2 * mt + (g >> 1)) + 4 * (g & 1);
      *reinterpret_cast<u32x2*>(xt + 0 * FX_XPL + off) = (u32x2){pack_hi(hh[0], hh[1]), pack_hi(hh[2], hh[3])};
      *reinterpret_cast<u32x2*>(xt + 1 * FX_XPL + off) = (u32x2){pack_hi(mm[0], mm[1]), pack_hi(mm[2], mm[3])};
      *reinterpret_cast<u32x2*>(xt + 2 * FX_XPL + off) = (u32x2){pack_hi(ll[0], ll[1]), pack_hi(ll[2], ll[3])};
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i)
      if (tid + i * 256 < WNV) *reinterpret_cast<u32x4*>(ws + wl[i]) = wr[i];
    __syncthreads();
    if (c + 1 < nchunk) issue(c + 1);

#pragma unroll
    for (int j = 0; j < 3; ++j) {
      bf16x8 af[CT2][3];
#pragma unroll
      for (int ct = 0; ct < CT2; ++ct)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          af[ct][pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ws + pc * WPL + swz(j * MT + 16 * (wc * CT2 + ct) + ln, g)));
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int m = n + j + 3;
        bf16x8 bf[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          bf[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xt + pc * FX_XPL + swz((m & 3) * FX_PR + 16 * wt + ln + (m >> 2), g)));
#pragma unroll
        for (int ct = 0; ct < CT2; ++ct) {
          f32x4 cc = acc[ct][n];
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct][2], bf[0], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct][0], bf[2], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct][1], bf[1], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct][1], bf[0], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct][0], bf[1], cc, 0, 0, 0);
          cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct][0], bf[0], cc, 0, 0, 0);
          acc[ct][n] = cc;
        }
      }
    }
  }

  const int t0 = n0 + 64 * wt + 4 * ln;
#pragma unroll
  for (int ct = 0; ct < CT2; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = r0 + 16 * (wc * CT2 + ct) + 4 * g + r;
      if (co < p.Cout && t0 < T) {
        const float bs = bias[ct][r];
        const float v0 = acc[ct][0][r], v1 = acc[ct][1][r], v2 = acc[ct][2][r], v3 = acc[ct][3][r];
        *reinterpret_cast<f32x4*>(p.y + (long)b * p.y_bs + (long)co * T + t0) = (f32x4){v0 + bs, v1 + bs, v2 + bs, v3 + bs};
      }
    }
}

__host__ __device__ __forceinline__ int x6_mt(int Cout) { return (Cout % 64 == 0) ? 64 : 32; }      // output channels per block

// fp32 weights [Cout][Cin][3] -> [Cout / MT][5 chunks][MT / 32 records][piece][32 co][tap][32 ci] bf16 pieces (zero for channels >= Cin):
// the 32-channel chunk of one block's MT output channels is one linear run, in the order of the kernel's LDS image
__global__ __launch_bounds__(256) void conv_x6_weight_planes_kernel(const float* w, int Cout, int Cin, unsigned short* planes) {
  const int n = Cout * 3 * FX_CP;                     // elements of one piece over all records
  const int cbn = x6_mt(Cout) >> 5;                   // records per (block, chunk)
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int ci32 = i & 31, j = (i >> 5) % 3, rest = (i >> 5) / 3;         // rest = ((blk * 5 + chunk) * cbn + cb) * 32 + co32
    const int co32 = rest & 31, rec = rest >> 5, cb = rec % cbn, chunk = (rec / cbn) % 5, blk = rec / (cbn * 5);
    const int co = (blk * cbn + cb) * 32 + co32, ci = chunk * 32 + ci32;
    unsigned h = 0, m = 0, l = 0;
    if (ci < Cin) split1(w[((long)co * Cin + ci) * 3 + j], h, m, l);
    unsigned short* o = planes + (long)rec * (3 * 32 * 3 * 32) + (co32 * 3 + j) * 32 + ci32;
    o[0] = (unsigned short)(h >> 16); o[32 * 3 * 32] = (unsigned short)(m >> 16); o[2 * 32 * 3 * 32] = (unsigned short)(l >> 16);
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
  if (x6_mt(d->Cout) == 64) {
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

extern "C" int tdvc_film_cond_fwd_x6(const tdvc_film_cond_args* a, const void* w2_planes, uint32_t* cv0_sign_bits, int64_t bits_bs, void* stream) {
  if (!a || !a->exc || !a->w0 || !a->k3 || !a->gb || !w2_planes) return tdvc_fail(TDVC_EINVAL, "film_cond_fwd_x6: null pointer");
  const bool shape = a->n_var == 8 && (a->n_cond & 3) == 0 && a->n_cond > 64 && a->n_cond <= FX_CP && a->C2 >= 32 && a->C2 % 32 == 0 && a->T >= FX_NT &&
                     (a->T & 3) == 0 && (!cv0_sign_bits || (a->T & 31) == 0) && (long)a->n_cond * a->T < (1L << 29) && a->B > 0 && a->B < 65536 &&
                     a->slope > 0.f && a->slope <= 1.f;
  const bool al = ((((uintptr_t)a->exc) | ((uintptr_t)a->gb) | ((uintptr_t)a->cv0) | ((uintptr_t)a->k3) | ((uintptr_t)w2_planes)) & 15) == 0 &&
                  (a->exc_bs & 3) == 0 && (a->gb_bs & 3) == 0 && (!a->cv0 || (a->cv0_bs & 3) == 0);
  if (g_knob[6] || g_force_tile >= 0 || g_force_generic || !shape || !al)
    return tdvc_fail(TDVC_EUNSUPPORTED, "film_cond_fwd_x6: outside the fused split-bf16 conditioning forward's contract");
  CondX6P p = {};
  p.exc = a->exc; p.exc_bs = a->exc_bs; p.w0 = a->w0; p.w0_rs = a->n_cond * 3; p.w0_off = (a->n_cond - a->n_var) * 3; p.k3 = a->k3;
  p.wp = (const unsigned short*)w2_planes; p.bias = a->b2; p.cv0 = a->cv0; p.cv0_bs = a->cv0_bs; p.bits = cv0_sign_bits; p.bits_bs = bits_bs;
  p.y = a->gb; p.y_bs = a->gb_bs; p.T = a->T; p.Cin = a->n_cond; p.Cout = a->C2; p.slope = a->slope;
  hipStream_t st = (hipStream_t)stream;
  const int nt = (a->T + FX_NT - 1) / FX_NT;
  const size_t es_bytes = (size_t)8 * FX_ES * sizeof(float);
  if (x6_mt(a->C2) == 64) {
    auto k = film_cond_fwd_x6_kernel<4>;
    TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
    hipLaunchKernelGGL(k, dim3(nt, a->C2 / 64, a->B), dim3(256), (size_t)(3 * FX_XPL + 3 * 64 * 3 * FX_RS) * 2 + es_bytes, st, p);
  } else {
    auto k = film_cond_fwd_x6_kernel<2>;
    TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
    hipLaunchKernelGGL(k, dim3(nt, a->C2 / 32, a->B), dim3(256), (size_t)(3 * FX_XPL + 3 * 32 * 3 * FX_RS) * 2 + es_bytes, st, p);
  }
  TDVC_CHECK_LAUNCH();
  return TDVC_OK;
}
