// Backward of one FiLM block's conditioning network (model/generator.py:86-92,103) behind cond_var.2's output gradient, in ONE
// kernel: the 136-channel gradient d_cv0 = dL/d(cond_var.0 output) is produced and consumed per (sample, 60-step) chunk inside
// the block and never touches HBM (the two-launch path wrote and re-read 4 * B * nc * T bytes: 557 MB at T = 16000).
//
//   d_cv0[c][t]      = mask[c][t] * sum_{co,j} W2[co][c][j] * dgb[co][t + 1 - j]        (input-grad of cond_var.2, LeakyReLU mask)
//   dexc[ce][u]      = sum_{c,j} W0x[c][ce][j] * d_cv0[c][u + 1 - j]                    (input-grad wrt the excitation)
//   dW0x[c][ce][j]  += sum_t d_cv0[c][t] * exc[ce][t + j - 1]                           (weight-grad of the excitation window)
//   dk3[b][c][0..2]  = d_cv0[c][0], sum_{0<t<T-1} d_cv0[c][t], d_cv0[c][T-1]            (adjoint of the 3-valued embedding bias)
//
// A block walks `tpb` consecutive chunks. Per chunk, all on v_mfma_f32_16x16x4_f32:
//   (p) D[t][c] over 64 computed columns t = n0 - 2 .. n0 + 61 (wave w owns columns 16w .. 16w + 15, all 9 channel tiles):
//       reduction over (co, j) in chunks of 16 output channels -- the W2^T chunk [144][48] and the dgb chunk [16][68] are staged
//       through registers one chunk ahead of the MFMA loop;
//   mask (1 sign bit per element from the forward, or the stored fp32 cv0) and zeroing outside [0, T) in the accumulator layout,
//   the tile goes to LDS once ([144][68], aliasing the staging buffers of (p)); from it
//   (b) dW0x / dk3 partials: the 18 (row half, channel tile) units of the [32][144] result are dealt to the 4 waves, each unit
//       reduces over the chunk's 60 steps (20 accumulator registers per wave, nothing to sum across waves);
//   (a) P[i][(ce, j)] = sum_c d_cv0[c][i] W0x[c][ce][j] over all 64 columns (rows i = 16w + ln), then the three shifted taps
//       are summed through a small LDS image: dexc[u][ce] = sum_j P[u + 1 - j][(ce, j)].
// dW partials stay in registers across the block's chunks (one slab per block, folded by slab_reduce); dk3 partials are
// flushed with atomics when the block moves to another sample.
#include "conv_common.h"
#include "api_util.h"
#include <type_traits>

PROF_DEFINE(tdvc_debug_condbwd_prof)

namespace tdvc {

hipError_t launch_slab_reduce(const float* slab, int nslab, long stride, long n, float* dw, int rowlen, long dst_row_stride,
                              hipStream_t st, long n_w, float* dbias);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct CondBwdP {
  const float* dgb; long dgb_bs;         // [B][C2][T]
  const float* wt2;                      // cond_var.2 effective weight, pre-transposed [nc][C2][3]
  const unsigned* bits; long bits_bs;    // sign bits of cv0 [B][nc][T/32] (bit set = positive) or null
  const float* cv0; long cv0_bs;         // fp32 mask source when bits == null
  const float* exc; long exc_bs;
  const float* w; int w_rs;              // excitation window of cond_var.0's weight: [c][ce][j] at w[c * w_rs + ce * 3 + j]
  float* dexc; long dexc_bs;             // may be null
  float* dk3;                            // [B][nc][3], zeroed before the launch
  float* slab; long slab_stride;         // may be null (no weight gradient wanted)
  int B, T, nc, C2, ntile, tpb, nchunks;
  float slope;
};

constexpr int FC_NT = 60;               // output steps per chunk
constexpr int FC_DS = 68;               // d_cv0 tile row stride: 36 * 68 = 16 (mod 32) -> (a)'s lane groups read disjoint banks; rows 16-byte aligned
constexpr int FC_GS = 80;               // dgb chunk row stride (16 mod 32: the k-lane groups of (p)'s A fragment read disjoint banks)
constexpr int FC_ES = 80;
constexpr int FC_CC = 16;               // output channels of cond_var.2 per reduction chunk
constexpr int FC_W2S = FC_CC * 3 + 2;   // 50 = 18 (mod 32): the 16 rows x 2 k-lanes of (p)'s B fragment hit 32 distinct banks
constexpr int FC_WS0 = 26;              // staged W0x row: 24 taps + 2 zeros
constexpr int FC_CT = 9, FC_ROWS = 144, FC_G = 36;
constexpr int FC_WNV = FC_CC * 3 / 4;   // 12 float4 per staged weight row
constexpr int FC_WRP = 256 / FC_WNV;    // 21 rows per pass
constexpr int FC_WNP = (FC_ROWS + FC_WRP - 1) / FC_WRP;   // 7 passes
constexpr int FC_GNV = 17;              // float4 columns of the staged window [n0 - 4, n0 + 64)
constexpr int FC_NU = 5;                // (b): 18 units = (row half, channel tile) dealt round-robin to the 4 waves: unit u = wave + 4k, k < 5
constexpr int FC_PS = 68;               // row stride of the (a) product's column-major image [24][FC_PS]
constexpr int FC_LDS_FLOATS = FC_ROWS * FC_DS + 8 * FC_ES + FC_ROWS * FC_WS0 + FC_ROWS * 3 + 24 * FC_PS;
static_assert(FC_ROWS * FC_W2S + FC_CC * FC_GS <= FC_ROWS * FC_DS, "staging buffers must fit under the d_cv0 tile they alias");

template <bool BITS>
__global__ __launch_bounds__(256, 2) void film_cond_bwd_kernel(const CondBwdP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ds = smem;                               // [FC_ROWS][FC_DS] masked d_cv0 tile, column i <-> position n0 - 2 + i
  float* w2s = smem;                              // (aliases ds) [FC_ROWS][FC_W2S] W2^T chunk
  float* gs = smem + FC_ROWS * FC_W2S;            // (aliases ds) [FC_CC][FC_GS] dgb chunk, column <-> position n0 - 4 + col
  float* es = smem + FC_ROWS * FC_DS;             // [8][FC_ES] excitation tile, same window
  float* wsm = es + 8 * FC_ES;                    // [FC_ROWS][FC_WS0] W0x, zero rows / zero pad columns
  unsigned* bsm = reinterpret_cast<unsigned*>(wsm + FC_ROWS * FC_WS0);   // [FC_ROWS][3] sign-bit words of the chunk's window
  float* ps = wsm + FC_ROWS * FC_WS0 + FC_ROWS * 3;                     // [24][FC_PS] (a): P[(ce, j)][i]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // provably wave-uniform: (b)'s unit table below stays in scalar registers
  const int ln = lane & 15, kq = lane >> 4;
  const int T = p.T, C2 = p.C2;
  const int q_begin = blockIdx.x * p.tpb, q_end = min(p.nchunks, q_begin + p.tpb);
  if (q_begin >= q_end) return;
  const int ncc = C2 / FC_CC;
  constexpr bool have_bits = BITS;
  const int nw = T >> 5;

  for (int i = tid; i < FC_ROWS * FC_WS0; i += 256) {
    const int c = i / FC_WS0, r = i - c * FC_WS0;
    wsm[i] = (c < p.nc && r < 24) ? p.w[(long)c * p.w_rs + r] : 0.f;
  }

  // ---- staging roles
  const int wr = tid / FC_WNV, wvv = tid - wr * FC_WNV;           // weight chunk: (row of the pass, float4 column)
  const bool wact = wr < FC_WRP;
  const int w_goff = wact ? (wr * C2 * 3 + 4 * wvv) * 4 : 0x7f000000;
  const int w_gstep = FC_WRP * C2 * 3 * 4;
  const int w_loff = wr * FC_W2S + 4 * wvv;
  const srd_t wrs = make_srd(p.wt2, p.nc * C2 * 3 * 4);           // rows >= nc read as zero
  const int gr0 = tid / FC_GNV, gv0 = tid - gr0 * FC_GNV;        // dgb chunk element tid (row, float4 column); tid < 16: also element 256 + tid
  const int e1 = 256 + tid;
  const int gr1 = e1 / FC_GNV, gv1 = e1 - gr1 * FC_GNV;
  const bool g1act = tid < FC_CC * FC_GNV - 256;
  const bool eact = tid < 8 * FC_GNV;                             // excitation tile: one float4 per thread
  const int bc0 = tid / 3, bw0 = tid - bc0 * 3;                   // sign-bit words: elements tid and 256 + tid of [144][3]
  const int bc1 = e1 / 3, bw1 = e1 - bc1 * 3;
  const bool b1act = e1 < FC_ROWS * 3;

  // (b) A-operand rows: r = ln (row half 0), 16 + ln (row half 1); rows < 24 are (ce, j) = divmod(r, 3), 24..26 the dk3 rows.
  // E'[(ce, j)][t] = exc[ce][t + j - 1]; computed column i <-> t = n0 - 2 + i <-> es column i + j + 1
  const int e0 = (ln / 3) * FC_ES + (ln % 3) + 1;
  const int r1 = 16 + ln;
  const int e1o = (r1 / 3) * FC_ES + (r1 % 3) + 1;
  const int wcol = ln < 8 ? ln * 3 : 24;           // (a) B-operand column: ce = ln (< 8), else the zero pad column

  f32x4 acc2[FC_NU];                               // (b) accumulators of this wave's units, kept across the block's chunks
#pragma unroll
  for (int k = 0; k < FC_NU; ++k) acc2[k] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int b = __builtin_amdgcn_readfirstlane(q_begin / p.ntile);
  int tile = q_begin - b * p.ntile;

  // Register prefetch: the W2^T chunk (served from L2) one reduction step ahead, the dgb chunk and the per-chunk operands
  // (excitation tile, sign-bit words: HBM streams) two steps ahead. A "step" is one (chunk, cc) pair of the block's walk.
  f32x4 wv[FC_WNP], gv[2][2], ev;
  unsigned bv[2];
  auto issue_w = [&](int cc) {
    int vo = w_goff + cc * FC_CC * 3 * 4;
    asm volatile("" : "+v"(vo));                     // keep the 7 row offsets transient: hoisted out of the loops they cost 14 live registers
#pragma unroll
    for (int i = 0; i < FC_WNP; ++i) { wv[i] = buf_load4(wrs, vo); vo += w_gstep; }
  };
  auto issue_g = [&](f32x4 (&g)[2], int bb, int tt, int cc, auto chunk_ops) {
    const int n0 = tt * FC_NT;
    const srd_t grs = make_srd(p.dgb + (long)bb * p.dgb_bs, C2 * T * 4);
    const int q0 = n0 - 4 + 4 * gv0;
    g[0] = buf_load4(grs, (q0 >= 0 && q0 < T) ? ((cc * FC_CC + gr0) * T + q0) * 4 : 0x7f000000);
    const int q1 = n0 - 4 + 4 * gv1;
    g[1] = buf_load4(grs, (g1act && q1 >= 0 && q1 < T) ? ((cc * FC_CC + gr1) * T + q1) * 4 : 0x7f000000);
    if (decltype(chunk_ops)::value) {     // (cc == 0) per-chunk operands: excitation tile and the sign-bit words covering positions n0 - 2 .. n0 + 61
      const srd_t ers = make_srd(p.exc + (long)bb * p.exc_bs, 8 * T * 4);
      ev = buf_load4(ers, (eact && q0 >= 0 && q0 < T) ? (gr0 * T + q0) * 4 : 0x7f000000);
      if (have_bits) {
        const srd_t brs = make_srd(reinterpret_cast<const float*>(p.bits + (long)bb * p.bits_bs), p.nc * nw * 4);   // channels >= nc read as zero
        const int wbase = max(n0 - 2, 0) >> 5;
        bv[0] = __builtin_amdgcn_raw_buffer_load_b32(brs, (wbase + bw0 < nw) ? (bc0 * nw + wbase + bw0) * 4 : 0x7f000000, 0, 0);
        bv[1] = __builtin_amdgcn_raw_buffer_load_b32(brs, (b1act && wbase + bw1 < nw) ? (bc1 * nw + wbase + bw1) * 4 : 0x7f000000, 0, 0);
      }
    }
  };
  using yes_t = std::integral_constant<bool, true>;
  using no_t = std::integral_constant<bool, false>;
  issue_w(0);
  issue_g(gv[0], b, tile, 0, yes_t{});
  issue_g(gv[1], b, tile, 1, no_t{});                // ncc >= 2 (host-checked: C2 % 32 == 0)
  PROF_DECL

  for (int q = q_begin; q < q_end; ++q) {
    const int n0 = tile * FC_NT;
    int nb = b, ntile_i = tile + 1;
    if (ntile_i == p.ntile) { ntile_i = 0; ++nb; }
    const bool more = q + 1 < q_end;

    // ---- (p) d_cv0 tile: D[t = 16w + 4kq + r][c = 16m + ln]
    f32x4 acc[FC_CT];
#pragma unroll
    for (int m = 0; m < FC_CT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // FIRST: cc == 0 (commits the per-chunk operands); LAST: cc >= ncc - 2 (its prefetches belong to the next chunk). Compile-time
    // flags: a run-time `if (cc == 0) ev = load` makes the merge of old and new register values wait on the loads just issued.
    auto reduce_step = [&](auto first_c, auto last_c, auto even_c, int cc, f32x4 (&g)[2]) {
      constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value, EVEN = decltype(even_c)::value;
      PROF(9)
      __syncthreads();          // cc == 0: (a) / (b) of the previous chunk have read the tile these buffers alias; else: the previous MFMA loop is done
      PROF(0)
      if (wact) {
#pragma unroll
        for (int i = 0; i < FC_WNP; ++i) {
          if (i * FC_WRP + wr < FC_ROWS) {
            f32x2* d2 = reinterpret_cast<f32x2*>(w2s + w_loff + i * FC_WRP * FC_W2S);
            d2[0] = (f32x2){wv[i][0], wv[i][1]}; d2[1] = (f32x2){wv[i][2], wv[i][3]};
          }
        }
      }
      *reinterpret_cast<f32x4*>(gs + gr0 * FC_GS + 4 * gv0) = g[0];
      if (g1act) *reinterpret_cast<f32x4*>(gs + gr1 * FC_GS + 4 * gv1) = g[1];
      if (FIRST) {
        if (eact) *reinterpret_cast<f32x4*>(es + gr0 * FC_ES + 4 * gv0) = ev;
        if (have_bits) { bsm[tid] = bv[0]; if (b1act) bsm[e1] = bv[1]; }
      }
      PROF(1)
      __syncthreads();
      PROF(2)
      if (!LAST || EVEN) issue_w(cc + 1); else if (more) issue_w(0);
      if (!LAST) issue_g(g, b, tile, cc + 2, no_t{});
      else if (more) { if (EVEN) issue_g(g, nb, ntile_i, 0, yes_t{}); else issue_g(g, nb, ntile_i, 1, no_t{}); }
      PROF(3)
      // 12 k-steps (4 channel groups x 3 taps); the fragments of step s + 1 are read from LDS before the MFMAs of step s issue
      const float* gl = gs + kq * FC_GS + 16 * wave + ln + 3;      // A: dgb[co = 4cs + kq][(n0 - 2 + 16w + ln) + 1 - j]
      const float* wl = w2s + ln * FC_W2S + kq * 3;                // B: W2^T[c = 16m + ln][(co = 4cs + kq, j)]
      // one fragment set: the reads of step s + 1 issue right behind the MFMAs of step s (which have latched their operands) and
      // land while those 9 x 32 pipe cycles run; a second register set bought nothing and cost 10 registers
      float av, bw[FC_CT];
#pragma unroll
      for (int st = 0; st < (FC_CC / 4) * 3; ++st) {
        const int cs = st / 3, j = st - cs * 3;
        av = gl[cs * 4 * FC_GS - j];
#pragma unroll
        for (int m = 0; m < FC_CT; ++m) bw[m] = wl[m * 16 * FC_W2S + cs * 12 + j];
#pragma unroll
        for (int m = 0; m < FC_CT; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[m], acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);           // keep the fragment reads one step ahead, not twelve: the prefetch registers need the room
      }
    };
    if (ncc == 2) {
      reduce_step(yes_t{}, yes_t{}, yes_t{}, 0, gv[0]); reduce_step(no_t{}, yes_t{}, no_t{}, 1, gv[1]);
    } else {
      reduce_step(yes_t{}, no_t{}, yes_t{}, 0, gv[0]); reduce_step(no_t{}, no_t{}, no_t{}, 1, gv[1]);
      for (int cc = 2; cc < ncc - 2; cc += 2) { reduce_step(no_t{}, no_t{}, yes_t{}, cc, gv[0]); reduce_step(no_t{}, no_t{}, no_t{}, cc + 1, gv[1]); }
      reduce_step(no_t{}, yes_t{}, yes_t{}, ncc - 2, gv[0]); reduce_step(no_t{}, yes_t{}, no_t{}, ncc - 1, gv[1]);
    }

    PROF(4)
    // ---- LeakyReLU mask of cond_var.0's output + zero outside [0, T), in the accumulator layout
    const int i0 = 16 * wave + 4 * kq;               // this lane's computed columns i0 .. i0 + 3
    const int pos0 = n0 - 2 + i0;                    // = 2 (mod 4): positions pos0, pos0 + 1 share a 32-step word, so do pos0 + 2, pos0 + 3
    if (have_bits) {
      const int wbase = max(n0 - 2, 0) >> 5;
      const int ia = (max(pos0, 0) >> 5) - wbase, ib = (max(pos0 + 2, 0) >> 5) - wbase;    // 0 .. 2
      const int sa = pos0 & 31, sb = (pos0 + 2) & 31;
#pragma unroll
      for (int m = 0; m < FC_CT; ++m) {
        const unsigned ma = bsm[(m * 16 + ln) * 3 + ia] >> sa, mb = bsm[(m * 16 + ln) * 3 + ib] >> sb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pos = pos0 + r;
          const unsigned bit = ((r < 2 ? ma >> r : mb >> (r - 2)) & 1u);
          const float v = acc[m][r];
          acc[m][r] = (pos >= 0 && pos < T) ? (bit ? v : v * p.slope) : 0.f;
        }
      }
    } else {       // no sign bits (T % 32 != 0): the stored fp32 cond_var.0 output; elements outside the tensor load as zero
      const srd_t crs = make_srd(p.cv0 + (long)b * p.cv0_bs, p.nc * T * 4);
#pragma unroll
      for (int m = 0; m < FC_CT; ++m) {
        float sv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pos = pos0 + r;
          sv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(crs, (pos >= 0 && pos < T) ? ((m * 16 + ln) * T + pos) * 4 : 0x7f000000, 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pos = pos0 + r;
          const float v = acc[m][r];
          acc[m][r] = (pos >= 0 && pos < T) ? (sv[r] > 0.f ? v : v * p.slope) : 0.f;
        }
      }
    }
    PROF(5)
    __syncthreads();                                 // every wave is done with the staging buffers the tile aliases
#pragma unroll
    for (int m = 0; m < FC_CT; ++m) *reinterpret_cast<f32x4*>(ds + (m * 16 + ln) * FC_DS + i0) = acc[m];
    __syncthreads();                                 // the tile is complete in LDS
    PROF(6)

    // ---- (b) dW0x / dk3 partials: D2[(ce, j) | dk3 row][c] += sum_t E'[row][t] * d_cv0[c][t] over the chunk's own 60 steps
    // (the halo columns belong to the neighbours); this wave's units, all 16 k-steps of 4 columns. Unit u = wave + 4k:
    // k = 0, 1 -> row half 0; k = 3 -> row half 1; k = 2 -> half 0 for wave 0, else half 1; k = 4 (half 1) exists for waves 0, 1.
    {
      const bool k2lo = wave == 0, k4 = wave < 2;               // wave-uniform
      const float* bpk[FC_NU];
#pragma unroll
      for (int k = 0; k < FC_NU; ++k) {
        const int u = min(wave + 4 * k, 2 * FC_CT - 1);
        bpk[k] = ds + ((u >= FC_CT ? u - FC_CT : u) * 16 + ln) * FC_DS + kq;
      }
      const float* ea0 = es + e0 + kq;
      const float* ea1 = es + (ln < 8 ? e1o : e0) + kq;          // rows 24..31 read a valid address; the value is replaced below
      // rows 24, 25, 26 of the row half 1 are indicators (-> dk3[..][0], [1], [2]). In a chunk whose 60 steps are all interior
      // positions (every chunk but a sample's first and last) the indicator is constant: 1 on row 25, else 0.
      const bool interior = n0 >= 1 && n0 + FC_NT - 1 <= T - 2;  // wave-uniform
      const float c25 = ln == 9 ? 1.f : 0.f;
      float a0v[2], a1v[2], bq[2][FC_NU];
      auto load_b = [&](int buf, int ks, bool ends) {
        float x0 = ea0[4 * ks], x1 = ea1[4 * ks];
        float ind = c25;
        if (ends) {                                             // generic form: the first / last k-step, or a chunk at a sequence end
          const int i = 4 * ks + kq, tg = n0 - 2 + i;
          const bool live = i >= 2 && i < FC_NT + 2;
          ind = ((ln == 8 && tg == 0) || (ln == 9 && tg >= 1 && tg <= T - 2) || (ln == 10 && tg == T - 1)) ? 1.f : 0.f;
          x0 = live ? x0 : 0.f; x1 = live ? x1 : 0.f; ind = live ? ind : 0.f;
        }
        a0v[buf] = x0;
        a1v[buf] = ln < 8 ? x1 : ind;
#pragma unroll
        for (int k = 0; k < FC_NU; ++k) bq[buf][k] = bpk[k][4 * ks];
      };
      auto mma_b = [&](int buf) {
        acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v[buf], bq[buf][0], acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0v[buf], bq[buf][1], acc2[1], 0, 0, 0);
        acc2[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(k2lo ? a0v[buf] : a1v[buf], bq[buf][2], acc2[2], 0, 0, 0);
        acc2[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v[buf], bq[buf][3], acc2[3], 0, 0, 0);
        if (k4) acc2[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1v[buf], bq[buf][4], acc2[4], 0, 0, 0);
      };
      load_b(0, 0, true);
      if (interior) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          if (ks + 1 < 16) load_b((ks + 1) & 1, ks + 1, ks + 1 == 15);
          mma_b(ks & 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll 2
        for (int ks = 0; ks < 16; ++ks) {
          if (ks + 1 < 16) load_b((ks + 1) & 1, ks + 1, true);
          mma_b(ks & 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    PROF(8)
    // ---- (a) dexc: P[i][(ce, j)] = sum_c d_cv0[c][i] * W0x[c][ce][j] for all 64 computed columns i (rows i = 16w + ln, the 24
    // (ce, j) columns as two 16-column tiles, reduction over the 144 channel rows), then dexc[u][ce] = sum_j P[u + 1 - j][(ce, j)]
    // through LDS: 72 MFMAs per wave instead of the 108 of three shifted 8-column products.
    if (p.dexc) {
      f32x4 pacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      const float* ap = ds + kq * FC_G * FC_DS + 16 * wave + ln;              // A: d_cv0[c = 36 kq + cs][i = 16w + ln]
      const float* wq0 = wsm + kq * FC_G * FC_WS0 + ln;                        // B tile 0: column (ce, j) = ln
      const float* wq1 = wsm + kq * FC_G * FC_WS0 + (ln < 8 ? 16 + ln : 24);   // B tile 1: column 16 + ln (< 24), else the zero pad column
#pragma unroll 6
      for (int cs = 0; cs < FC_G; ++cs) {
        const float a = ap[cs * FC_DS];
        pacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wq0[cs * FC_WS0], pacc[0], 0, 0, 0);
        pacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wq1[cs * FC_WS0], pacc[1], 0, 0, 0);
      }
      // lane (ln, kq) holds P[i = 16w + 4kq + r][col = ln | 16 + ln]: column-major image [24][FC_PS]
      *reinterpret_cast<f32x4*>(ps + ln * FC_PS + 16 * wave + 4 * kq) = pacc[0];
      if (ln < 8) *reinterpret_cast<f32x4*>(ps + (16 + ln) * FC_PS + 16 * wave + 4 * kq) = pacc[1];
      __syncthreads();
      if (tid < 8 * (FC_NT / 4)) {                   // thread = (ce, 4 consecutive steps): u = n0 + 4g + r <-> column i = 2 + 4g + r
        const int ce = tid / (FC_NT / 4), g = tid - ce * (FC_NT / 4);
        const float* pj = ps + ce * 3 * FC_PS + 2 + 4 * g;
        f32x4 d1;
#pragma unroll
        for (int r = 0; r < 4; ++r) d1[r] = (pj[r + 1] + pj[FC_PS + r]) + pj[2 * FC_PS + r - 1];
        if (n0 + 4 * g < T) *reinterpret_cast<f32x4*>(p.dexc + (long)b * p.dexc_bs + (long)ce * T + n0 + 4 * g) = d1;
      }
    }

    // ---- dk3 partials leave the registers when the sample changes: rows 24..26 = row half 1, lane group kq = 2, registers 0..2
    if (nb != b || q + 1 == q_end) {
#pragma unroll
      for (int k = 0; k < FC_NU; ++k) {
        const int u = wave + 4 * k;
        if (u >= FC_CT && u < 2 * FC_CT && kq == 2) {
          const int c = (u - FC_CT) * 16 + ln;
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            if (c < p.nc) atomicAdd(p.dk3 + ((long)b * p.nc + c) * 3 + e, acc2[k][e]);
            acc2[k][e] = 0.f;
          }
        }
      }
    }
    b = nb; tile = ntile_i;
  }

  PROF(9)
  PROF_END
  // ---- dW: each (row half, channel tile) unit belongs to one wave: one slab per block, layout [c][ce * 3 + j]
  if (p.slab) {
    float* slab = p.slab + (long)blockIdx.x * p.slab_stride;
#pragma unroll
    for (int k = 0; k < FC_NU; ++k) {
      const int u = wave + 4 * k;
      if (u < 2 * FC_CT) {
        const int mh = u >= FC_CT ? 1 : 0, c = (u - mh * FC_CT) * 16 + ln;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int r = mh * 16 + kq * 4 + rr;
          if (r < 24 && c < p.nc) slab[c * 24 + r] = acc2[k][rr];
        }
      }
    }
  }
}

static void cond_bwd_plan(int B, int T, int* ntile, int* tpb, int* nblocks) {
  *ntile = (T + FC_NT - 1) / FC_NT;
  const long nchunks = (long)B * (*ntile);
  long nb = nchunks < 512 ? nchunks : 512;          // one resident wave of blocks (2 per CU)
  *tpb = (int)((nchunks + nb - 1) / nb);
  *nblocks = (int)((nchunks + *tpb - 1) / *tpb);
}

}  // namespace tdvc

using namespace tdvc;

extern "C" size_t tdvc_film_cond_bwd_workspace(int32_t B, int32_t T, int32_t n_cond, int32_t n_var) {
  if (B <= 0 || T <= 0 || n_cond <= 0 || n_var != 8) return 0;
  int ntile, tpb, nblocks;
  cond_bwd_plan(B, T, &ntile, &tpb, &nblocks);
  return (size_t)nblocks * (size_t)n_cond * 24 * sizeof(float);
}

extern "C" int tdvc_film_cond_bwd(const tdvc_film_cond_bwd_args* a, void* stream) {
  if (!a || !a->dgb || !a->wt2 || !a->exc || !a->w0 || !a->dk3 || (!a->cv0_sign_bits && !a->cv0))
    return tdvc_fail(TDVC_EINVAL, "film_cond_bwd: null pointer");
  if (a->B <= 0 || a->T < 4 || (a->T & 3) || a->n_var != 8 || a->n_cond <= a->n_var || (a->n_cond & 3) || a->n_cond > FC_ROWS ||
      a->C2 < 2 * FC_CC || (a->C2 % (2 * FC_CC)) || (long)a->C2 * a->T >= (1L << 29) || (long)a->n_cond * a->C2 * 3 >= (1L << 29))
    return tdvc_fail(TDVC_EUNSUPPORTED, "film_cond_bwd: needs T % 4 == 0, 8 excitation channels, n_cond % 4 == 0, n_cond <= 144, C2 % 32 == 0");
  if (a->cv0_sign_bits && (a->T & 31)) return tdvc_fail(TDVC_EUNSUPPORTED, "film_cond_bwd: sign bits need T % 32 == 0");
  if ((a->dgb_bs & 3) || (a->exc_bs & 3) || (a->dexc_bs & 3) ||
      (((uintptr_t)a->dgb | (uintptr_t)a->exc | (uintptr_t)a->dexc | (uintptr_t)a->wt2) & 15))
    return tdvc_fail(TDVC_EUNSUPPORTED, "film_cond_bwd: rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  CondBwdP p = {};
  p.dgb = a->dgb; p.dgb_bs = a->dgb_bs; p.wt2 = a->wt2; p.C2 = a->C2;
  p.bits = a->cv0_sign_bits; p.bits_bs = a->cv0_sign_bits_bs; p.cv0 = a->cv0; p.cv0_bs = a->cv0_bs;
  p.exc = a->exc; p.exc_bs = a->exc_bs;
  p.w_rs = a->n_cond * 3; p.w = a->w0 + (long)(a->n_cond - a->n_var) * 3;
  p.dexc = a->dexc; p.dexc_bs = a->dexc_bs; p.dk3 = a->dk3;
  p.B = a->B; p.T = a->T; p.nc = a->n_cond; p.slope = a->slope;
  int nblocks;
  cond_bwd_plan(a->B, a->T, &p.ntile, &p.tpb, &nblocks);
  p.nchunks = a->B * p.ntile;
  const long sstride = (long)a->n_cond * 24;
  if (a->dw0) {
    if (!a->workspace || a->workspace_bytes < (size_t)nblocks * sstride * sizeof(float))
      return tdvc_fail(TDVC_EWORKSPACE, "film_cond_bwd: workspace too small");
    p.slab = (float*)a->workspace; p.slab_stride = sstride;
  }
  if (hipMemsetAsync(a->dk3, 0, (size_t)a->B * a->n_cond * 3 * sizeof(float), st) != hipSuccess)
    return tdvc_fail(TDVC_ELAUNCH, "film_cond_bwd: memset failed");
  const size_t lds = g_knob[3] ? (size_t)100 * 1024 : (size_t)FC_LDS_FLOATS * sizeof(float);   // knob 3 (diagnostic): one block per CU
  if (p.bits) {
    auto k = film_cond_bwd_kernel<true>;
    TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), lds, st, p);
  } else {
    auto k = film_cond_bwd_kernel<false>;
    TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), lds, st, p);
  }
  TDVC_CHECK_LAUNCH();
  if (a->dw0) {
    const hipError_t e = launch_slab_reduce(p.slab, nblocks, sstride, sstride, a->dw0 + (long)(a->n_cond - a->n_var) * 3, 24,
                                            (long)a->n_cond * 3, st, -1, nullptr);
    if (e != hipSuccess) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
  }
  return TDVC_OK;
}
