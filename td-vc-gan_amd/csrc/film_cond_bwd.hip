// Backward of FiLM's first conditioning conv (cond_var.0, model/generator.py:86-92) in the split formulation:
// everything that consumes d_cv0 = dL/d(cond_var.0 output) [B][nc][T] in ONE pass over it,
//   dexc[b][ce][u]   = sum_{c,j} W0x[c][ce][j] * d_cv0[b][c][u + 1 - j]          (input-grad wrt the excitation)
//   dW0x[c][ce][j]  += sum_{b,t} d_cv0[b][c][t] * exc[b][ce][t + j - 1]          (weight-grad of the excitation window)
//   dk3[b][c][0..2]  = d_cv0[b][c][0], sum_{0<t<T-1} d_cv0[b][c][t], d_cv0[b][c][T-1]   (adjoint of the 3-valued bias)
// instead of three kernels that each stream the 136-channel tensor from HBM (the largest tensor of the step).
//
// A block walks `tpb` consecutive (sample, 64-step) chunks. Per chunk the d_cv0 tile [nc][72] and the excitation tile
// [8][72] are staged by row walks through raw buffer descriptors (zero outside [0, T)), then
//   (a) dexc tile:  D1[t][ce]     = sum_{(c,j)} A[t][(c,j)] * W[(c,j)][ce]      wave w owns time rows 16w .. 16w+15
//   (b) dW partial: D2[(ce,j)][c] += sum_t E'[(ce,j)][t] * d_cv0[c][t]          K (time) split over the 4 waves
// run on v_mfma_f32_16x16x4_f32. E' carries three extra rows -- the interior mask and the two end deltas -- so the
// dk3 sums fall out of the same matrix product (rows 24..26 of D2). dW partials stay in registers across the block's
// chunks and go to one slab per block (folded by slab_reduce_kernel); dk3 partials are flushed with atomics when the
// block moves to another sample.
#include "conv_common.h"
#include "api_util.h"

namespace tdvc {

hipError_t launch_slab_reduce(const float* slab, int nslab, long stride, long n, float* dw, int rowlen, long dst_row_stride,
                              hipStream_t st, long n_w, float* dbias);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Cond0BwdP {
  const float* dcv; long dcv_bs;
  const float* exc; long exc_bs;
  const float* w; int w_rs;            // excitation window of the weight: element [c][ce][j] at w[c * w_rs + ce * 3 + j]
  float* dexc; long dexc_bs;           // may be null
  float* dk3;                          // [B][nc][3], zeroed before the launch
  float* slab; long slab_stride;       // may be null (no weight gradient wanted)
  int B, T, nc, ntile, tpb, nchunks;
};

constexpr int CB_NT = 64;              // time steps per chunk
constexpr int CB_NV = 18;              // float4 columns staged: aligned window [n0 - 4, n0 + 68)
constexpr int CB_S = 76;               // LDS row stride of the tiles: 12 (mod 64) -> the 16 rows x 4 columns of a (b) fragment read hit 64 distinct
                                       // banks (16 distinct multiples of 4, + kq), rows are 16-byte aligned
constexpr int CB_G = 36;               // (a): lane group kq reduces channels kq*36 + cs; 36 * 76 = 48 (mod 64) -> the 4 groups' 16-column reads are disjoint
                                       // (alternative 74 / 40 rows apart -- conflict-free under a 32-bank half-wave model -- measured 4 % slower on the same box)
constexpr int CB_WS = 26;              // staged weight row: 24 taps + 2 zeros
constexpr int CB_CT = 9;               // channel tiles of 16: nc <= 144
constexpr int CB_ROWS = CB_CT * 16;
constexpr int CB_RP = 14;              // rows per row-walk pass (256 / 18)
constexpr int CB_NP = (CB_ROWS + CB_RP - 1) / CB_RP;

__global__ __launch_bounds__(256, 2) void film_cond0_bwd_kernel(const Cond0BwdP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ds = smem;                               // [CB_ROWS][CB_S] d_cv0 tile, rows >= nc are zero
  float* es = ds + CB_ROWS * CB_S;                // [8][CB_S] excitation tile
  float* wsm = es + 8 * CB_S;                     // [CB_ROWS][CB_WS] weights, zero rows / zero pad columns
  float* red2 = wsm + CB_ROWS * CB_WS;            // [4][CB_ROWS][3] dk3 partials of the 4 waves

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, kq = lane >> 4;
  const int T = p.T;
  const int q_begin = blockIdx.x * p.tpb, q_end = min(p.nchunks, q_begin + p.tpb);
  if (q_begin >= q_end) return;

  // ---- weights, once per block
  for (int i = tid; i < CB_ROWS * CB_WS; i += 256) {
    const int c = i / CB_WS, r = i - c * CB_WS;
    wsm[i] = (c < p.nc && r < 24) ? p.w[(long)c * p.w_rs + r] : 0.f;
  }

  // ---- row-walk roles: thread = (row of the pass, float4 column)
  const int rsub = (int)(((float)tid + 0.5f) * (1.0f / (float)CB_NV));
  const int vv = tid - rsub * CB_NV;
  const bool ract = rsub < CB_RP;
  const int lds_off = rsub * CB_S + 4 * vv;

  // (b) A-operand rows: r = ln (tile 0), 16 + ln (tile 1); rows < 24 are (ce, j) = divmod(r, 3), 24..26 the dk3 rows
  const int e0 = (ln / 3) * CB_S + (ln % 3) - 1 + 4;
  const int r1 = 16 + ln;
  const int e1 = (r1 / 3) * CB_S + (r1 % 3) - 1 + 4;
  // (a) B-operand column: ce = ln (< 8), else the zero pad column
  const int wcol = ln < 8 ? ln * 3 : 24;   // (lanes ln >= 8 re-reading lane ln - 8's address instead of the zero column: no gain, 2 % slower on the same box)

  f32x4 acc2[2][CB_CT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int c = 0; c < CB_CT; ++c) acc2[m][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int b = __builtin_amdgcn_readfirstlane(q_begin / p.ntile);
  int tile = q_begin - b * p.ntile;
  // Register prefetch one chunk ahead (T14): the loads of chunk q+1 are issued before the MFMAs of chunk q and stored to
  // LDS after them, so their latency hides under the matrix work.
  f32x4 v[CB_NP], ex;
  const int er = tid / CB_NV, ev = tid - er * CB_NV;           // tid < 144: one float4 of the excitation tile
  auto issue = [&](int bb, int tt) {    // stage d_cv0 [nc][72] and exc [8][72]; columns outside [0, T) and rows >= nc load as zero
    const int n0 = tt * CB_NT;
    const srd_t drs = make_srd(p.dcv + (long)bb * p.dcv_bs, p.nc * T * 4);
    const int qc = n0 - 4 + 4 * vv;
    int vo = (ract && qc >= 0 && qc < T) ? (rsub * T + qc) * 4 : 0x7f000000;
#pragma unroll
    for (int i = 0; i < CB_NP; ++i) { v[i] = buf_load4(drs, vo); vo += CB_RP * T * 4; }
    const srd_t ers = make_srd(p.exc + (long)bb * p.exc_bs, 8 * T * 4);
    const int eq = n0 - 4 + 4 * ev;
    ex = buf_load4(ers, (er < 8 && eq >= 0 && eq < T) ? (er * T + eq) * 4 : 0x7f000000);
  };
  issue(b, tile);
  for (int q = q_begin; q < q_end; ++q) {
    const int n0 = tile * CB_NT;
    int nb = b, ntile_i = tile + 1;
    if (ntile_i == p.ntile) { ntile_i = 0; ++nb; }
    __syncthreads();                                // the previous chunk's fragments are consumed (and wsm is written)
    if (ract) {
#pragma unroll
      for (int i = 0; i < CB_NP; ++i) {
        if (i * CB_RP + rsub < CB_ROWS) {
          *reinterpret_cast<f32x4*>(ds + lds_off + i * CB_RP * CB_S) = v[i];
        }
      }
    }
    if (er < 8) {
      f32x2* d = reinterpret_cast<f32x2*>(es + er * CB_S + 4 * ev);
      d[0] = (f32x2){ex[0], ex[1]}; d[1] = (f32x2){ex[2], ex[3]};
    }
    __syncthreads();
    if (q + 1 < q_end) issue(nb, ntile_i);

    // ---- (b) dW / dk3 partials: this wave's 16 time steps
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int t = wave * 16 + ks * 4 + kq;        // k index of this lane
      const int tg = n0 + t;
      const float a0 = es[e0 + t];
      float a1;
      if (ln < 8) a1 = es[e1 + t];
      else if (ln == 8) a1 = tg == 0 ? 1.f : 0.f;                       // rows 24, 25, 26 -> dk3[..][0], [1], [2]
      else if (ln == 9) a1 = (tg >= 1 && tg <= T - 2) ? 1.f : 0.f;
      else if (ln == 10) a1 = tg == T - 1 ? 1.f : 0.f;
      else a1 = 0.f;
      const float* bp = ds + ln * CB_S + t + 4;
#pragma unroll
      for (int c = 0; c < CB_CT; ++c) {
        const float bv = bp[c * 16 * CB_S];
        acc2[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc2[0][c], 0, 0, 0);
        acc2[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc2[1][c], 0, 0, 0);
      }
    }

    // ---- (a) dexc tile: rows t = 16 * wave + ln, reduction over (channel group, tap)
    if (p.dexc) {
      f32x4 acc1[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // one chain per tap
      // the k index of a step is free: lane group kq takes channel kq * CB_G + cs (rows >= nc are zero in both operands)
      const float* ap = ds + kq * CB_G * CB_S + wave * 16 + ln + 4 + 1;      // d_cv0[c = 36 kq + cs][t + 1 - j]
      const float* wp = wsm + kq * CB_G * CB_WS + wcol;                       // W0x[c = 36 kq + cs][ce = ln][j]
      for (int cs = 0; cs < CB_G; ++cs) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[cs * CB_S - j], wp[cs * CB_WS + (ln < 8 ? j : 0)], acc1[j], 0, 0, 0);
      }
      const f32x4 d1 = acc1[0] + acc1[1] + acc1[2];
      const int t0 = n0 + wave * 16 + kq * 4;        // D1[t = 16w + 4 kq + r][ce = ln]
      if (ln < 8 && t0 < T) *reinterpret_cast<f32x4*>(p.dexc + (long)b * p.dexc_bs + (long)ln * T + t0) = d1;
    }

    // ---- dk3 partials leave the registers when the sample changes
    if (nb != b || q + 1 == q_end) {
      if (kq == 2) {
#pragma unroll
        for (int c = 0; c < CB_CT; ++c)
#pragma unroll
          for (int e = 0; e < 3; ++e) { red2[((wave * CB_ROWS) + c * 16 + ln) * 3 + e] = acc2[1][c][e]; acc2[1][c][e] = 0.f; }
      }
      __syncthreads();
      for (int i = tid; i < p.nc * 3; i += 256) {
        const float s = (red2[i] + red2[CB_ROWS * 3 + i]) + (red2[2 * CB_ROWS * 3 + i] + red2[3 * CB_ROWS * 3 + i]);
        atomicAdd(p.dk3 + (long)b * p.nc * 3 + i, s);
      }
    }
    b = nb; tile = ntile_i;
  }

  // ---- dW: sum the 4 waves through LDS, one slab per block, layout [c][ce * 3 + j]
  if (p.slab) {
    float* red = ds;                                 // [CB_ROWS][24]
    for (int w = 0; w < 4; ++w) {
      __syncthreads();
      if (wave == w) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int c = 0; c < CB_CT; ++c)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int r = m * 16 + kq * 4 + rr;
              if (r < 24) {
                float* d = red + (c * 16 + ln) * 24 + r;
                if (w == 0) *d = acc2[m][c][rr]; else *d += acc2[m][c][rr];
              }
            }
      }
    }
    __syncthreads();
    float* slab = p.slab + (long)blockIdx.x * p.slab_stride;
    for (int i = tid; i < p.nc * 24; i += 256) slab[i] = red[i];
  }
}


// ------------------------------------------------------------------------------------------------------------------
// k3 = cond_var.0 restricted to the time-constant speaker-embedding channels, evaluated on a length-3 constant signal
// (zero 'same' padding): position 0 misses tap 0, position 2 misses tap 2. A [B x n_const] . [n_const x 3 nc] product per
// FiLM block -- far too small for the conv kernels (three launches + a slab fold per block were ~3 ms per step).
__global__ __launch_bounds__(256) void film_k3_fwd_kernel(const float* emb, long emb_bs, const float* w0, const float* b0, float* k3,
                                                          int B, int n_const, int nc) {
  // one wave per output channel c: lanes run along the (contiguous) weight row, which stays in registers for all samples
  const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= nc) return;
  const float* w = w0 + (long)c * nc * 3;
  float wa[4], wb[4], wc[4];                       // n_const <= 256
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ci = lane + 64 * i;
    const bool ok = ci < n_const;
    wa[i] = ok ? w[ci * 3] : 0.f; wb[i] = ok ? w[ci * 3 + 1] : 0.f; wc[i] = ok ? w[ci * 3 + 2] : 0.f;
  }
  const float bias = b0 ? b0[c] : 0.f;
  const int b_end = min(B, ((int)blockIdx.y + 1) * 4);
  for (int b = blockIdx.y * 4; b < b_end; ++b) {       // 4 samples per block: enough blocks to cover the latency of this tiny op
    const float* e = emb + (long)b * emb_bs;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ci = lane + 64 * i;
      const float ev = ci < n_const ? e[ci] : 0.f;
      s0 += ev * (wb[i] + wc[i]); s1 += ev * ((wa[i] + wb[i]) + wc[i]); s2 += ev * (wa[i] + wb[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) { float* o3 = k3 + ((long)b * nc + c) * 3; o3[0] = s0 + bias; o3[1] = s1 + bias; o3[2] = s2 + bias; }
  }
}

// blocks [0, nA): weight / bias gradients, thread = (c, ci), sequential (deterministic) sum over the batch, accumulated into
// the arena; blocks [nA, ...): embedding gradient, thread = (b, ci).
__global__ __launch_bounds__(256) void film_k3_bwd_kernel(const float* dk3, const float* emb, long emb_bs, const float* w0, float* demb,
                                                          float* dw0, float* db0, int B, int n_const, int nc, int nA) {
  if ((int)blockIdx.x < nA) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nc * n_const || !dw0) return;
    const int c = idx / n_const, ci = idx - c * n_const;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, gb = 0.f;
    for (int b = 0; b < B; ++b) {
      const float* d = dk3 + ((long)b * nc + c) * 3;
      const float d0 = d[0], d1 = d[1], d2 = d[2], ev = emb[(long)b * emb_bs + ci];
      g0 += ev * (d1 + d2); g1 += ev * ((d0 + d1) + d2); g2 += ev * (d0 + d1); gb += (d0 + d1) + d2;
    }
    float* o = dw0 + ((long)c * nc + ci) * 3;
    o[0] += g0; o[1] += g1; o[2] += g2;
    if (ci == 0 && db0) db0[c] += gb;
  } else {
    // embedding gradient: block = (sample b, 32 embedding channels); the 8 thread groups split the sum over the nc
    // output channels and are folded through LDS in a fixed order (deterministic)
    __shared__ float part[8][32];
    const int blk = (int)blockIdx.x - nA;
    const int per_b = (n_const + 31) / 32;
    const int b = blk / per_b, ci = (blk - b * per_b) * 32 + (threadIdx.x & 31), cg = threadIdx.x >> 5;
    float a = 0.f;
    if (ci < n_const) {
      for (int c = cg; c < nc; c += 8) {
        const float* d = dk3 + ((long)b * nc + c) * 3;
        const float* w = w0 + ((long)c * nc + ci) * 3;
        const float d0 = d[0], d1 = d[1], d2 = d[2];
        a += w[0] * (d1 + d2) + w[1] * ((d0 + d1) + d2) + w[2] * (d0 + d1);
      }
    }
    part[cg][threadIdx.x & 31] = a;
    __syncthreads();
    if (cg == 0 && ci < n_const) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += part[g][threadIdx.x];
      demb[(long)b * n_const + ci] = t;
    }
  }
}

// The same two ops for ALL FiLM blocks of one MRF stage (9 blocks read the same embedding): one launch each way instead of
// nine, and the embedding gradient comes out already summed over the blocks (fixed order: deterministic).
constexpr int K3M_MAX = 16;
struct K3Multi { const float* w0[K3M_MAX]; const float* b0[K3M_MAX]; float* k3[K3M_MAX]; const float* dk3[K3M_MAX]; float* dw0[K3M_MAX]; float* db0[K3M_MAX]; int n; };

__global__ __launch_bounds__(256) void film_k3_multi_fwd_kernel(const float* emb, long emb_bs, const K3Multi m, int B, int n_const, int nc) {
  const int z = blockIdx.z;
  const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= nc) return;
  const float* w = m.w0[z] + (long)c * nc * 3;
  float wa[4], wb[4], wc[4];                       // n_const <= 256
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ci = lane + 64 * i;
    const bool ok = ci < n_const;
    wa[i] = ok ? w[ci * 3] : 0.f; wb[i] = ok ? w[ci * 3 + 1] : 0.f; wc[i] = ok ? w[ci * 3 + 2] : 0.f;
  }
  const float bias = m.b0[z] ? m.b0[z][c] : 0.f;
  float* k3 = m.k3[z];
  const int b_end = min(B, ((int)blockIdx.y + 1) * 4);
  for (int b = blockIdx.y * 4; b < b_end; ++b) {
    const float* e = emb + (long)b * emb_bs;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ci = lane + 64 * i;
      const float ev = ci < n_const ? e[ci] : 0.f;
      s0 += ev * (wb[i] + wc[i]); s1 += ev * ((wa[i] + wb[i]) + wc[i]); s2 += ev * (wa[i] + wb[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) { float* o3 = k3 + ((long)b * nc + c) * 3; o3[0] = s0 + bias; o3[1] = s1 + bias; o3[2] = s2 + bias; }
  }
}

// blocks [0, nA) x grid.y = FiLM block: weight / bias gradients of that block; blocks [nA, ...) (grid.y == 0 only): embedding
// gradient summed over the FiLM blocks
__global__ __launch_bounds__(256) void film_k3_multi_bwd_kernel(const float* emb, long emb_bs, const K3Multi m, float* demb, int B, int n_const,
                                                                int nc, int nA) {
  if ((int)blockIdx.x < nA) {
    const int z = blockIdx.y;
    float* dw0 = m.dw0[z];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nc * n_const || !dw0) return;
    const int c = idx / n_const, ci = idx - c * n_const;
    const float* dk3 = m.dk3[z];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, gb = 0.f;
    for (int b = 0; b < B; ++b) {
      const float* d = dk3 + ((long)b * nc + c) * 3;
      const float d0 = d[0], d1 = d[1], d2 = d[2], ev = emb[(long)b * emb_bs + ci];
      g0 += ev * (d1 + d2); g1 += ev * ((d0 + d1) + d2); g2 += ev * (d0 + d1); gb += (d0 + d1) + d2;
    }
    float* o = dw0 + ((long)c * nc + ci) * 3;
    o[0] += g0; o[1] += g1; o[2] += g2;
    if (ci == 0 && m.db0[z]) m.db0[z][c] += gb;
  } else {
    if (blockIdx.y != 0 || !demb) return;
    __shared__ float part[8][32];
    const int blk = (int)blockIdx.x - nA;
    const int per_b = (n_const + 31) / 32;
    const int b = blk / per_b, ci = (blk - b * per_b) * 32 + (threadIdx.x & 31), cg = threadIdx.x >> 5;
    float a = 0.f;
    if (ci < n_const) {
      for (int z = 0; z < m.n; ++z) {
        const float* dk3 = m.dk3[z]; const float* w0 = m.w0[z];
        for (int c = cg; c < nc; c += 8) {
          const float* d = dk3 + ((long)b * nc + c) * 3;
          const float* w = w0 + ((long)c * nc + ci) * 3;
          const float d0 = d[0], d1 = d[1], d2 = d[2];
          a += w[0] * (d1 + d2) + w[1] * ((d0 + d1) + d2) + w[2] * (d0 + d1);
        }
      }
    }
    part[cg][threadIdx.x & 31] = a;
    __syncthreads();
    if (cg == 0 && ci < n_const) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) t += part[g][threadIdx.x];
      demb[(long)b * n_const + ci] = t;
    }
  }
}

static void cond0_plan(int B, int T, int* ntile, int* tpb, int* nblocks) {
  *ntile = (T + CB_NT - 1) / CB_NT;
  const long nchunks = (long)B * (*ntile);
  long nb = nchunks < 512 ? nchunks : 512;          // one resident wave of blocks (2 per CU)
  *tpb = (int)((nchunks + nb - 1) / nb);
  *nblocks = (int)((nchunks + *tpb - 1) / *tpb);
}

}  // namespace tdvc

using namespace tdvc;

extern "C" size_t tdvc_film_cond0_bwd_workspace(int32_t B, int32_t T, int32_t n_cond, int32_t n_var) {
  if (B <= 0 || T <= 0 || n_cond <= 0 || n_var != 8) return 0;
  int ntile, tpb, nblocks;
  cond0_plan(B, T, &ntile, &tpb, &nblocks);
  return (size_t)nblocks * (size_t)n_cond * 24 * sizeof(float);
}

extern "C" int tdvc_film_cond0_bwd(const tdvc_film_cond0_bwd_args* a, void* stream) {
  if (!a || !a->dcv || !a->exc || !a->w0 || !a->dk3) return tdvc_fail(TDVC_EINVAL, "film_cond0_bwd: null pointer");
  if (a->B <= 0 || a->T < 4 || (a->T & 3) || a->n_var != 8 || a->n_cond <= a->n_var || (a->n_cond & 3) || a->n_cond > CB_ROWS)
    return tdvc_fail(TDVC_EUNSUPPORTED, "film_cond0_bwd: needs T % 4 == 0, 8 excitation channels, n_cond % 4 == 0, n_cond <= 144");
  if ((a->dcv_bs & 3) || (a->exc_bs & 3) || (a->dexc_bs & 3) || (((uintptr_t)a->dcv | (uintptr_t)a->exc | (uintptr_t)a->dexc) & 15))
    return tdvc_fail(TDVC_EUNSUPPORTED, "film_cond0_bwd: rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  Cond0BwdP p = {};
  p.dcv = a->dcv; p.dcv_bs = a->dcv_bs; p.exc = a->exc; p.exc_bs = a->exc_bs;
  p.w_rs = a->n_cond * 3; p.w = a->w0 + (long)(a->n_cond - a->n_var) * 3;
  p.dexc = a->dexc; p.dexc_bs = a->dexc_bs; p.dk3 = a->dk3;
  p.B = a->B; p.T = a->T; p.nc = a->n_cond;
  int nblocks;
  cond0_plan(a->B, a->T, &p.ntile, &p.tpb, &nblocks);
  p.nchunks = a->B * p.ntile;
  const long sstride = (long)a->n_cond * 24;
  if (a->dw0) {
    if (!a->workspace || a->workspace_bytes < (size_t)nblocks * sstride * sizeof(float))
      return tdvc_fail(TDVC_EWORKSPACE, "film_cond0_bwd: workspace too small");
    p.slab = (float*)a->workspace; p.slab_stride = sstride;
  }
  if (hipMemsetAsync(a->dk3, 0, (size_t)a->B * a->n_cond * 3 * sizeof(float), st) != hipSuccess)
    return tdvc_fail(TDVC_ELAUNCH, "film_cond0_bwd: memset failed");
  TDVC_BIG_LDS_ONCE(film_cond0_bwd_kernel); TDVC_TRACE(film_cond0_bwd_kernel);
  const size_t lds = (size_t)(CB_ROWS * CB_S + 8 * CB_S + CB_ROWS * CB_WS + 4 * CB_ROWS * 3) * sizeof(float);
  hipLaunchKernelGGL(film_cond0_bwd_kernel, dim3(nblocks), dim3(256), lds, st, p);
  TDVC_CHECK_LAUNCH();
  if (a->dw0) {
    const hipError_t e = launch_slab_reduce(p.slab, nblocks, sstride, sstride, a->dw0 + (long)(a->n_cond - a->n_var) * 3, 24,
                                            (long)a->n_cond * 3, st, -1, nullptr);
    if (e != hipSuccess) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
  }
  return TDVC_OK;
}

extern "C" int tdvc_film_k3_fwd(const float* emb, int64_t emb_bs, const float* w0, const float* b0, float* k3, int32_t B, int32_t n_const,
                                int32_t n_cond, void* stream) {
  if (!emb || !w0 || !k3 || B <= 0 || n_const <= 0 || n_const > 256 || n_cond < n_const) return tdvc_fail(TDVC_EINVAL, "film_k3_fwd: bad arguments");
  hipLaunchKernelGGL(film_k3_fwd_kernel, dim3((n_cond + 3) / 4, (B + 3) / 4), dim3(256), 0, (hipStream_t)stream, emb, (long)emb_bs, w0, b0, k3, B,
                     n_const, n_cond);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}

extern "C" int tdvc_film_k3_multi_fwd(const float* emb, int64_t emb_bs, const float* const* w0s, const float* const* b0s, float* const* k3s,
                                      int32_t nblk, int32_t B, int32_t n_const, int32_t n_cond, void* stream) {
  if (!emb || !w0s || !k3s || nblk < 1 || nblk > K3M_MAX || B <= 0 || n_const <= 0 || n_const > 256 || n_cond < n_const)
    return tdvc_fail(TDVC_EINVAL, "film_k3_multi_fwd: bad arguments");
  K3Multi m = {};
  m.n = nblk;
  for (int i = 0; i < nblk; ++i) {
    if (!w0s[i] || !k3s[i]) return tdvc_fail(TDVC_EINVAL, "film_k3_multi_fwd: null pointer");
    m.w0[i] = w0s[i]; m.b0[i] = b0s ? b0s[i] : nullptr; m.k3[i] = k3s[i];
  }
  hipLaunchKernelGGL(film_k3_multi_fwd_kernel, dim3((n_cond + 3) / 4, (B + 3) / 4, nblk), dim3(256), 0, (hipStream_t)stream, emb, (long)emb_bs, m, B,
                     n_const, n_cond);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}

extern "C" int tdvc_film_k3_multi_bwd(const float* const* dk3s, const float* emb, int64_t emb_bs, const float* const* w0s, float* demb,
                                      float* const* dw0s, float* const* db0s, int32_t nblk, int32_t B, int32_t n_const, int32_t n_cond, void* stream) {
  if (!dk3s || !emb || !w0s || nblk < 1 || nblk > K3M_MAX || B <= 0 || n_const <= 0 || n_cond < n_const)
    return tdvc_fail(TDVC_EINVAL, "film_k3_multi_bwd: bad arguments");
  K3Multi m = {};
  m.n = nblk;
  bool any_w = false;
  for (int i = 0; i < nblk; ++i) {
    if (!dk3s[i] || !w0s[i]) return tdvc_fail(TDVC_EINVAL, "film_k3_multi_bwd: null pointer");
    m.dk3[i] = dk3s[i]; m.w0[i] = w0s[i]; m.dw0[i] = dw0s ? dw0s[i] : nullptr; m.db0[i] = (db0s && m.dw0[i]) ? db0s[i] : nullptr;
    any_w = any_w || m.dw0[i];
  }
  const int nA = any_w ? (n_cond * n_const + 255) / 256 : 0, nB = demb ? B * ((n_const + 31) / 32) : 0;
  if (nA + nB == 0) return TDVC_OK;
  hipLaunchKernelGGL(film_k3_multi_bwd_kernel, dim3(nA + nB, any_w ? nblk : 1), dim3(256), 0, (hipStream_t)stream, emb, (long)emb_bs, m, demb, B,
                     n_const, n_cond, nA);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}

extern "C" int tdvc_film_k3_bwd(const float* dk3, const float* emb, int64_t emb_bs, const float* w0, float* demb, float* dw0, float* db0,
                                int32_t B, int32_t n_const, int32_t n_cond, void* stream) {
  if (!dk3 || !emb || !w0 || B <= 0 || n_const <= 0 || n_cond < n_const) return tdvc_fail(TDVC_EINVAL, "film_k3_bwd: bad arguments");
  const int nA = dw0 ? (n_cond * n_const + 255) / 256 : 0, nB = demb ? B * ((n_const + 31) / 32) : 0;
  if (nA + nB == 0) return TDVC_OK;
  hipLaunchKernelGGL(film_k3_bwd_kernel, dim3(nA + nB), dim3(256), 0, (hipStream_t)stream, dk3, emb, (long)emb_bs, w0, demb, dw0, db0, B,
                     n_const, n_cond, nA);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
