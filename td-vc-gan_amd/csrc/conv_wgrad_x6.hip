// Weight gradient of a 3-tap stride-1 'same' conv with up to 144 input channels (FiLM's cond_var.2, model/generator.py:86-92:
// 136 -> 2C, the widest-input layer of the generator and ~10 % of the train step) on the bf16 matrix pipe at fp32 accuracy:
//     dW[co][ci][j] = sum_{b,t} dy[co][t] * x'[ci][t + j - 1],       x' = LeakyReLU(x) (or x)
// "Split-bf16 x6": every fp32 operand is cut EXACTLY into three bf16 pieces (x = hi + mid + lo: 3 x 8 significant bits, by
// truncation, no rounding anywhere) and the product is the six piece products whose weight is >= 2^-16 of the leading one,
//     a*b ~= lo.hi + hi.lo + mid.mid + mid.hi + hi.mid + hi.hi      (dropped: mid.lo, lo.mid, lo.lo <= 2^-24 relative)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_16x16x32_bf16. Six 16-cycle instructions do the work of
// eight 32-cycle v_mfma_f32_16x16x4_f32: 2.7x the matrix rate of the exact-fp32 path at its accuracy (rel-L2 vs float64
// 4.5e-7 on the D-layer-5 GEMM shape, profiles/r02_l_split_bf16_probe.txt; the op tests hold 2e-5).
//
// K of the product is TIME, which is contiguous in both operands ([B][C][T] rows): a lane's 8 consecutive k values are 16
// contiguous bytes of a bf16 row -- no transposition. The tap shift is applied to the dy side, which has few rows: the dy tile is
// stored as three copies shifted by +1 / 0 / -1 steps (dW[j] = sum_u dy[u + 1 - j] * x'[u]), so every fragment read is a
// 16-byte aligned ds_read_b128.
// A block owns 32 output rows x all (<= 144) input channels x 3 taps and walks `tpb` consecutive (sample, 32-step) chunks with its
// 54 accumulator tiles spread over the 4 waves (wave = one 16-row half x every second 16-channel tile): dy AND x are read from
// HBM once per row block (the fp32 kernel's 32 x 32 tiles re-read dy five times). Per chunk: global loads of chunk q + 1 are in
// flight during the MFMAs of chunk q; split + LDS store; one k-block of 32 steps = 90 MFMAs per wave.
#include "conv_common.h"
#include "conv_wgrad_lean.h"
#include "api_util.h"

namespace tdvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct WgX6P {
  const float* dy; long dy_bs;         // [B][R][T]
  const float* x; long x_bs;           // [B][Cin][T]
  float* slab; long slab_stride;       // per chunk group: module layout [R][Cin][3], bias partials at bias_off
  long bias_off;                       // < 0: no bias gradient
  int R, Cin, T, ntiles, tpb, nchunks;
  float x_slope;                       // LeakyReLU slope of the x prologue (1 = none)
};

constexpr int X6_NT = 32;              // steps per chunk = one k-block
constexpr int X6_RS = 32;              // bf16 row length: 64 B, no padding; the 16-byte slot q of row r lives at slot q ^ ((r >> 1) & 3) (x6_swz)
constexpr int X6_XROWS = 144, X6_MT = 32;
// ds_read_b128 serves a wave in four NON-contiguous 16-lane groups (MI355X_MICROARCH.md, LDS), so a group mixes rows of two k-quarters and
// the padded 80-byte rows of the first version made every fragment read 2-way conflicting (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.49,
// profiles/r03_pmc.txt). This XOR is conflict-free for any 16 consecutive rows (tools/lds_swizzle_check.py); planes are multiples of 8 rows.
__device__ __forceinline__ int x6_swz(int r, int q) { return r * X6_RS + 8 * (q ^ ((r >> 1) & 3)); }
constexpr int X6_XPL = X6_XROWS * X6_RS;            // elements of one x plane
constexpr int X6_APL = X6_MT * X6_RS;               // elements of one dy plane of one shift
constexpr int X6_LDS_BYTES = (3 * X6_XPL + 9 * X6_APL) * 2;
constexpr int X6_XP = 5;               // float4 per thread of the x tile (144 rows x 8 float4 = 1152 = 4.5 x 256)
constexpr int X6_NU = 5;               // (row half, channel tile) units per wave: channel tiles (wave >> 1) + 2k

// exact 3-way split of 4 fp32 values into bf16 pieces, packed 4 x 16 bit per piece (element q in bits 16q .. 16q + 15)
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& mid, u32x2& lo) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float f = v[q];                // (hipcc 7.2: __builtin_bit_cast applied DIRECTLY to a vector element expression reads element 0 for every q)
    const unsigned u = __builtin_bit_cast(unsigned, f);
    h[q] = u & 0xffff0000u;
    const float r1 = f - __builtin_bit_cast(float, h[q]);                  // exact
    const unsigned u1 = __builtin_bit_cast(unsigned, r1);
    m[q] = u1 & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, m[q]);                    // exact; <= 8 significant bits left
    l[q] = __builtin_bit_cast(unsigned, r2) & 0xffff0000u;
  }
  hi = (u32x2){(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
  mid = (u32x2){(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
  lo = (u32x2){(l[0] >> 16) | l[1], (l[2] >> 16) | l[3]};
}

__global__ __launch_bounds__(256, 3) void conv_wgrad_x6_kernel(const WgX6P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  unsigned short* xs = smem16;                        // [3 pieces][144][X6_RS]
  unsigned short* as = smem16 + 3 * X6_XPL;           // [3 shifts][3 pieces][32][X6_RS]: shift 0: dy[u + 1], 1: dy[u], 2: dy[u - 1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, g = lane >> 4;
  const int T = p.T;
  const int r0 = blockIdx.y * X6_MT;
  const int q_begin = blockIdx.x * p.tpb, q_end = min(p.nchunks, q_begin + p.tpb);
  if (q_begin >= q_end) return;

  // staging roles: dy tile 32 rows x 8 float4 (one per thread), x tile 144 rows x 8 float4 (4.5 per thread)
  const int arow = tid >> 3, av = tid & 7;
  const int a_goff = ((r0 + arow) * T + 4 * av) * 4;            // + n0 * 4
  const int a_loff = x6_swz(arow, av >> 1) + 4 * (av & 1);
  const bool a_rowok = r0 + arow < p.R;

  f32x4 acc[X6_NU][3];
#pragma unroll
  for (int k = 0; k < X6_NU; ++k)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[k][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bias_acc = 0.f;
  const int half = wave & 1, c_first = wave >> 1;     // this wave's units: rows 16 * half .. + 15, channel tiles c_first + 2k
  const int nunits = c_first == 0 ? 5 : 4;

  int b = __builtin_amdgcn_readfirstlane(q_begin / p.ntiles);
  int tile = q_begin - b * p.ntiles;
  f32x4 av4, xv4[X6_XP];
  float hl, hr;                                       // dy[n0 - 1] (lanes av == 0), dy[n0 + 32] (lanes av == 7)
  auto issue = [&](int bb, int tt) {
    const int n0 = tt * X6_NT;
    const srd_t ars = make_srd(p.dy + (long)bb * p.dy_bs, p.R * T * 4);
    const bool cin = n0 + 4 * av < T;                 // T % 4 == 0: a float4 lies wholly inside or outside
    av4 = buf_load4(ars, (a_rowok && cin) ? a_goff + n0 * 4 : 0x7f000000);
    hl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ars, (a_rowok && av == 0 && n0 > 0) ? a_goff + (n0 - 1) * 4 : 0x7f000000, 0, 0));
    hr = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ars, (a_rowok && av == 7 && n0 + X6_NT < T) ? a_goff + (n0 + 4) * 4 : 0x7f000000, 0, 0));
    const srd_t xrs = make_srd(p.x + (long)bb * p.x_bs, p.Cin * T * 4);      // rows >= Cin read as zero
#pragma unroll
    for (int i = 0; i < X6_XP; ++i) {
      const int e = tid + i * 256, row = e >> 3, v = e & 7;
      xv4[i] = buf_load4(xrs, (e < X6_XROWS * 8 && n0 + 4 * v < T) ? (row * T + n0 + 4 * v) * 4 : 0x7f000000);
    }
  };
  issue(b, tile);

  for (int q = q_begin; q < q_end; ++q) {
    int nb = b, ntile_i = tile + 1;
    if (ntile_i == p.ntiles) { ntile_i = 0; ++nb; }
    __syncthreads();                                  // the previous chunk's fragments are consumed
    // ---- dy tile: bias partial, the three shifted copies, split, store
    {
      const f32x4 d = av4;
      bias_acc += (d[0] + d[1]) + (d[2] + d[3]);
      float nx0 = __shfl_down(d[0], 1), px3 = __shfl_up(d[3], 1);
      nx0 = av == 7 ? hr : nx0;
      px3 = av == 0 ? hl : px3;
      const f32x4 sp = {d[1], d[2], d[3], nx0};       // dy[u + 1]
      const f32x4 sm = {px3, d[0], d[1], d[2]};       // dy[u - 1]
      u32x2 h, m, l;
      split4(sp, h, m, l);
      *reinterpret_cast<u32x2*>(as + (0 * 3 + 0) * X6_APL + a_loff) = h;
      *reinterpret_cast<u32x2*>(as + (0 * 3 + 1) * X6_APL + a_loff) = m;
      *reinterpret_cast<u32x2*>(as + (0 * 3 + 2) * X6_APL + a_loff) = l;
      split4(d, h, m, l);
      *reinterpret_cast<u32x2*>(as + (1 * 3 + 0) * X6_APL + a_loff) = h;
      *reinterpret_cast<u32x2*>(as + (1 * 3 + 1) * X6_APL + a_loff) = m;
      *reinterpret_cast<u32x2*>(as + (1 * 3 + 2) * X6_APL + a_loff) = l;
      split4(sm, h, m, l);
      *reinterpret_cast<u32x2*>(as + (2 * 3 + 0) * X6_APL + a_loff) = h;
      *reinterpret_cast<u32x2*>(as + (2 * 3 + 1) * X6_APL + a_loff) = m;
      *reinterpret_cast<u32x2*>(as + (2 * 3 + 2) * X6_APL + a_loff) = l;
    }
    // ---- x tile: LeakyReLU, split, store
#pragma unroll
    for (int i = 0; i < X6_XP; ++i) {
      const int e = tid + i * 256;
      if (e < X6_XROWS * 8) {
        f32x4 v = xv4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], v[k] * p.x_slope);     // slope in (0, 1]: LeakyReLU; 1: identity
        u32x2 h, m, l;
        split4(v, h, m, l);
        const int off = x6_swz(e >> 3, (e & 7) >> 1) + 4 * (e & 1);
        *reinterpret_cast<u32x2*>(xs + 0 * X6_XPL + off) = h;
        *reinterpret_cast<u32x2*>(xs + 1 * X6_XPL + off) = m;
        *reinterpret_cast<u32x2*>(xs + 2 * X6_XPL + off) = l;
      }
    }
    __syncthreads();
    if (q + 1 < q_end) issue(nb, ntile_i);

    // ---- one k-block (32 steps): lane (row / column ln, group g) holds k = 8g .. 8g + 7
    bf16x8 af[3][3];                                  // [shift][piece] of this wave's 16 rows
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        af[s][pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(as + (s * 3 + pc) * X6_APL + x6_swz(16 * half + ln, g)));
#pragma unroll
    for (int k = 0; k < X6_NU; ++k) {
      if (k < nunits) {                               // wave-uniform
        const int ct = c_first + 2 * k;
        bf16x8 bf[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
          bf[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xs + pc * X6_XPL + x6_swz(16 * ct + ln, g)));
#pragma unroll
        for (int j = 0; j < 3; ++j) {                 // tap j pairs x'[u] with dy[u + 1 - j] = shift copy j; smallest products first
          f32x4 c = acc[k][j];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j][2], bf[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j][0], bf[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j][1], bf[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j][1], bf[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j][0], bf[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j][0], bf[0], c, 0, 0, 0);
          acc[k][j] = c;
        }
      }
    }
    b = nb; tile = ntile_i;
  }

  // ---- slab: D[co = 4g + r][ci = ln] per (unit, tap), module layout [R][Cin][3]
  float* slab = p.slab + (long)blockIdx.x * p.slab_stride;
#pragma unroll
  for (int k = 0; k < X6_NU; ++k) {
    if (k < nunits) {
      const int ci = (c_first + 2 * k) * 16 + ln;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = r0 + 16 * half + 4 * g + r;
        if (co < p.R && ci < p.Cin) {
          float* dst = slab + ((long)co * p.Cin + ci) * 3;
#pragma unroll
          for (int j = 0; j < 3; ++j) dst[j] = acc[k][j][r];
        }
      }
    }
  }
  if (p.bias_off >= 0) {                              // a row's 8 float4 columns sit in 8 consecutive lanes
    float s = bias_acc;
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (av == 0 && a_rowok) slab[p.bias_off + r0 + arow] = s;
  }
}

// Contract check shared by the workspace query and the launch.
bool wgrad_x6_ok(int R, int Cin, int T, int K, int dil, int pad, int reflect) {
  return g_knob[5] == 0 && K == 3 && dil == 1 && pad == 1 && !reflect && Cin > 64 && Cin <= X6_XROWS && R >= X6_MT && R % X6_MT == 0 && T >= 64 && (T & 3) == 0 &&
         (long)R * T < (1L << 29) && (long)Cin * T < (1L << 29);
}
void wgrad_x6_plan(int R, int T, int B, int* ntiles, int* tpb, int* ngroups) {
  *ntiles = (T + X6_NT - 1) / X6_NT;
  const long nchunks = (long)B * (*ntiles);
  const int rowblocks = R / X6_MT;
  long grp = (g_knob[7] ? 512 : 768) / rowblocks;    // one resident round of blocks: 3 per CU (46 KB LDS, 165 VGPRs); knob 7: 2 per CU (A/B)
  if (grp < 1) grp = 1;
  if (grp > nchunks) grp = nchunks;
  *tpb = (int)((nchunks + grp - 1) / grp);
  *ngroups = (int)((nchunks + *tpb - 1) / *tpb);
}

hipError_t launch_conv_wgrad_x6(const WgLeanP& q, int B, hipStream_t st) {
  WgX6P p = {};
  p.dy = q.a.p; p.dy_bs = q.a.bs; p.x = q.x.p; p.x_bs = q.x.bs;
  p.slab = q.slab; p.slab_stride = q.slab_stride; p.bias_off = q.bias_off;
  p.R = q.R; p.Cin = q.Cin; p.T = q.N;
  p.x_slope = q.x.xf.kind == XF_LRELU ? q.x.xf.slope : 1.f;
  int ngroups;
  wgrad_x6_plan(p.R, p.T, B, &p.ntiles, &p.tpb, &ngroups);
  p.nchunks = B * p.ntiles;
  auto k = conv_wgrad_x6_kernel;
  TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
  hipLaunchKernelGGL(k, dim3(ngroups, p.R / X6_MT), dim3(256), (size_t)X6_LDS_BYTES, st, p);
  return hipGetLastError();
}

}  // namespace tdvc
