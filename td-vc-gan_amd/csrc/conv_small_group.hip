// Grouped strided Conv1d with 4 input and 4 output channels per group (model/discriminator.py:26-32, layer 4:
// 1024 -> 1024, kernel 41, stride 4, 256 groups), forward / input-grad / weight-grad on the vector ALU.
//
// Why not the matrix pipe: per group the product is a 4 x 164 by 164 x T matrix product. A 16-row MFMA tile is a quarter
// full, every (group, sample) is its own block of 44 MFMAs, and the 8192-16384 blocks of a launch spend their time in
// staging, barriers and epilogue (profiles/r02_f_generic_conv_phase_cycles.txt: 5 of 23 k cycles in MFMAs; 91 / 127 / 100
// us per launch for fwd / input-grad / weight-grad of a 41 MB problem). Here a wave owns one (group, 64 output steps) and
// a lane one output step: 656 FMAs per lane fed by LDS reads -- the input tile in time-to-depth layout (conflict free),
// the weights as broadcast float4.
#include "conv_common.h"
#include "conv_small_group.h"

namespace tdvc {

hipError_t launch_slab_reduce(const float* slab, int nslab, long stride, long n, float* dw, int rowlen, long dst_row_stride,
                              hipStream_t st, long n_w, float* dbias);


typedef float f32x4 __attribute__((ext_vector_type(4)));

// dword load through a raw buffer descriptor: elements that must read as zero get an out-of-range offset, so nothing selects
// on the loaded value and the loads of a batch stay in flight together (conv_common.h: tile_issue)
__device__ __forceinline__ float bload(srd_t rs, int elem, bool ok) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? elem * 4 : 0x7f000000, 0, 0));
}

// ---------------------------------------------------------------------------------------------- forward
// grid (ceil(G / 4), B, ceil(Tout / 64)); wave = one group, lane = one output step.
template <int S>
__global__ __launch_bounds__(256) void small_group_fwd_kernel(const SmallGroupP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x * 4 + wave, b = blockIdx.y, t0 = blockIdx.z * 64;
  const int xsz = 4 * S * p.CS, wsz = 16 * p.K;
  float* xs = smem + wave * (xsz + wsz);          // [ci][phi][CS]
  float* ws = xs + xsz;                           // [ci][k][co]
  const bool live = g < p.G;
  if (live) {
    // All global loads of the wave are issued before the first LDS store (a load -> store loop costs one memory round
    // trip per iteration: 31 of them here).
    // weights: module layout [co][ci][k] -> [ci][k][co]
    const float* wg = p.w + (long)g * 16 * p.K;
    float wv[12];                                  // 16 K <= 768
    const srd_t wrs = make_srd(wg, 16 * p.K * 4);
#pragma unroll
    for (int i = 0; i < 12; ++i) { const int e = lane + i * 64; wv[i] = bload(wrs, e, e < 16 * p.K); }
    // input window: positions q0 .. q0 + span - 1 of the 4 channels, q = q0 + col * s + phi
    const int q0 = t0 * S - p.pad, span = 63 * S + p.K;
    const float inv_s = 1.0f / (float)S;
    const srd_t xrs = make_srd(p.x + (long)b * p.x_bs + (long)(g * 4) * p.Tin, 4 * p.Tin * 4);
    for (int e0 = 0; e0 < span; e0 += 5 * 64) {   // one pass for s = 4, K = 41 (span 293)
      float xv[4][5];
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const int e = e0 + lane + i * 64, q = q0 + e;
          xv[ci][i] = bload(xrs, ci * p.Tin + q, e < span && q >= 0 && q < p.Tin);
        }
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const int e = e0 + lane + i * 64;
          if (e < span) {
            float v = xv[ci][i];
            if (p.act_in) v = fmaxf(v, v * p.slope_in);
            const int col = (int)(((float)e + 0.5f) * inv_s);
            xs[(ci * S + (e - col * S)) * p.CS + col] = v * p.in_scale;
          }
        }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int e = lane + i * 64;
      if (e < 16 * p.K) { const int co = e / (4 * p.K), r = e - co * 4 * p.K; ws[r * 4 + co] = wv[i]; }   // r = ci * K + k
    }
  }
  __syncthreads();
  if (!live) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int ci = 0; ci < 4; ++ci) {
    const float* xc = xs + ci * S * p.CS + lane;
    const f32x4* wc = reinterpret_cast<const f32x4*>(ws) + ci * p.K;
    for (int j = 0; j < p.J; ++j) {               // S taps per step (compile-time stride: no index arithmetic, independent reads)
      float xv4[S]; f32x4 w4[S];
#pragma unroll
      for (int phi = 0; phi < S; ++phi) {
        const bool ok = j * S + phi < p.K;
        // taps past K read columns of the tile that nothing staged (e >= span): mask the OPERAND too -- a zero weight alone
        // would turn a stale Inf / NaN bit pattern in LDS into NaN (0 * x)
        xv4[phi] = ok ? xc[phi * p.CS + j] : 0.f;
        w4[phi] = ok ? wc[j * S + phi] : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int phi = 0; phi < S; ++phi) acc += w4[phi] * xv4[phi];
    }
  }
  const int t = t0 + lane;
  if (t >= p.Tout) return;
#pragma unroll
  for (int co = 0; co < 4; ++co) {
    const int ch = g * 4 + co;
    float v = acc[co] + (p.bias ? p.bias[ch] : 0.f);
    if (p.post == POST_LRELU) v = lrelu_f(v, p.post_slope);
    else if (p.post == POST_TANH) v = tanhf(v);
    p.y[(long)b * p.y_bs + (long)ch * p.Tout + t] = v * p.out_scale;
  }
}

// ---------------------------------------------------------------------------------------------- input-grad
// dx[ci][m*s + r - pad] = sum_{co, j} w[co][ci][r + j*s] * dyM[co][m - j]: a lane owns one m and all s phases r, i.e. s
// consecutive input positions (one float4 store per channel when s == 4).
// grid (ceil(G / 4), B, ceil(M / 64)), M = number of m values = (Tin - 1 + pad) / s + 1.
template <int S>
__global__ __launch_bounds__(256) void small_group_dgrad_kernel(const SmallGroupP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x * 4 + wave, b = blockIdx.y, m0 = blockIdx.z * 64;
  const int DS = 64 + p.J;                         // dyM window [m0 - J + 1, m0 + 63] per output channel
  const int dsz = 4 * DS, wsz = 16 * p.J * S;
  float* ds = smem + wave * (dsz + wsz);          // [co][DS]
  float* ws = ds + dsz;                           // [co][j][r][ci]   (zero for r + j*s >= K)
  const bool live = g < p.G;
  if (live) {
    const float* wg = p.w + (long)g * 16 * p.K;
    float wv[12];                                  // 16 J s <= 768 (host-checked)
    const srd_t wrs = make_srd(wg, 16 * p.K * 4);
    const srd_t drs = make_srd(p.dy + (long)b * p.dy_bs + (long)(g * 4) * p.Tout, 4 * p.Tout * 4);
    const srd_t mrs = make_srd(p.mask ? p.mask + (long)b * p.mask_bs + (long)(g * 4) * p.Tout : p.dy, p.mask ? 4 * p.Tout * 4 : 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int e = lane + i * 64;
      const int ci = e & 3; int rest = e >> 2;
      const int r = rest % S; rest /= S;
      const int j = rest % p.J, co = rest / p.J;
      const int k = r + j * S;
      wv[i] = bload(wrs, (co * 4 + ci) * p.K + k, e < wsz && k < p.K);
    }
    float dv[4][2], mv[4][2];                      // DS <= 128
#pragma unroll
    for (int co = 0; co < 4; ++co)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = lane + i * 64, t = m0 - p.J + 1 + e;
        const bool ok = e < DS && t >= 0 && t < p.Tout;
        dv[co][i] = bload(drs, co * p.Tout + t, ok);
        mv[co][i] = bload(mrs, co * p.Tout + t, ok);      // no mask: the empty descriptor reads 0 -> handled below
      }
#pragma unroll
    for (int i = 0; i < 12; ++i) { const int e = lane + i * 64; if (e < wsz) ws[e] = wv[i]; }
#pragma unroll
    for (int co = 0; co < 4; ++co)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = lane + i * 64;
        if (e < DS) { const float v = dv[co][i] * p.dy_scale; ds[co * DS + e] = (!p.mask || mv[co][i] > 0.f) ? v : v * p.m_slope; }
      }
  }
  __syncthreads();
  if (!live) return;
  f32x4 acc[S];                                   // [r] -> 4 input channels
#pragma unroll
  for (int r = 0; r < S; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int co = 0; co < 4; ++co) {
    const float* dc = ds + co * DS + lane + p.J - 1;          // dyM[co][m - j] at dc[-j]
    const f32x4* wc = reinterpret_cast<const f32x4*>(ws) + co * p.J * S;
#pragma unroll 2
    for (int j = 0; j < p.J; ++j) {
      const float dv = dc[-j];
#pragma unroll
      for (int r = 0; r < S; ++r) acc[r] += wc[j * S + r] * dv;
    }
  }
  const int m = m0 + lane;
  const int u0 = m * S - p.pad;
  if (u0 >= p.Tin || u0 + S <= 0) return;
  float* dxb = p.y + (long)b * p.y_bs + (long)(g * 4) * p.Tin;
  const bool vec = S == 4 && u0 >= 0 && u0 + 3 < p.Tin && ((u0 | p.Tin) & 3) == 0 && (p.y_bs & 3) == 0 && (((uintptr_t)p.y) & 15) == 0;
#pragma unroll
  for (int ci = 0; ci < 4; ++ci) {
    if (vec) {
      if (S == 4) *reinterpret_cast<f32x4*>(dxb + (long)ci * p.Tin + u0) = (f32x4){acc[0][ci], acc[1 % S][ci], acc[2 % S][ci], acc[3 % S][ci]} * p.out_scale;
    } else {
#pragma unroll
      for (int r = 0; r < S; ++r)
        if (u0 + r >= 0 && u0 + r < p.Tin) dxb[(long)ci * p.Tin + u0 + r] = acc[r][ci] * p.out_scale;
    }
  }
}

// ---------------------------------------------------------------------------------------------- weight-grad
// dw[co][ci][k] += sum_{b, t} dyM[co][t] * x[ci][t*s + k - pad];  dbias[co] += sum dyM[co][t].
// grid (G, nsplit): a block owns one group and every nsplit-th sample; thread = weight elements (co, ci, k) in steps of
// 256; the block's partial goes to slab[split] in the module layout, bias partials behind the weights.
__global__ __launch_bounds__(256) void small_group_wgrad_kernel(const SmallGroupP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int g = blockIdx.x, split = blockIdx.y;
  const int XW = p.Tin + 2 * p.pad + p.s;          // padded input row: position q at xs[q + pad]
  float* xs = smem;                                // [4][XW]
  float* ds = xs + 4 * XW;                         // [4][Tout]
  const int nel = 16 * p.K;
  // thread e < 4 K owns the weight elements (co = 0..3, ci, k) of one (ci, k): one x read and one float4 dy read per step
  // for four products (a thread per (co, ci, k) element made the kernel LDS-issue bound)
  const int e_ci = tid / p.K, e_k = tid - e_ci * p.K;
  const bool e_live = tid < 4 * p.K;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float bacc = 0.f;
  // staging roles, the same for every sample: up to 6 input values and 1 dy (+ mask) value per thread
  int xo[6]; bool xk[6];
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    const int i = tid + u * 256;
    const int ci = i / XW, q = i - ci * XW - p.pad;
    xk[u] = i < 4 * XW && q >= 0 && q < p.Tin;
    xo[u] = ci * p.Tin + q;
  }
  const bool dk = tid < 4 * p.Tout;
  float xv[6], dv, mv;
  auto issue = [&](int b) {
    const srd_t xrs = make_srd(p.x + (long)b * p.x_bs + (long)(g * 4) * p.Tin, 4 * p.Tin * 4);
    const srd_t drs = make_srd(p.dy + (long)b * p.dy_bs + (long)(g * 4) * p.Tout, 4 * p.Tout * 4);
    const srd_t mrs = make_srd(p.mask ? p.mask + (long)b * p.mask_bs + (long)(g * 4) * p.Tout : p.dy, p.mask ? 4 * p.Tout * 4 : 0);
#pragma unroll
    for (int u = 0; u < 6; ++u) xv[u] = bload(xrs, xo[u], xk[u]);
    dv = bload(drs, tid, dk);
    mv = bload(mrs, tid, dk);
  };
  issue(split);
  for (int b = split; b < p.B; b += p.nsplit) {
    __syncthreads();                               // the previous sample's tiles are consumed
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = tid + u * 256;
      if (i < 4 * XW) { float v = xv[u]; if (p.act_in) v = fmaxf(v, v * p.slope_in); xs[i] = v * p.in_scale; }
    }
    if (dk) {                                     // ds layout [t][co]: one float4 per time step
      const float v = dv * p.dy_scale;
      const int co = tid / p.Tout, t = tid - co * p.Tout;
      ds[t * 4 + co] = (!p.mask || mv > 0.f) ? v : v * p.m_slope;
    }
    __syncthreads();
    if (b + p.nsplit < p.B) issue(b + p.nsplit);   // the next sample's loads fly under this sample's products
    if (e_live) {
      const f32x4* dr = reinterpret_cast<const f32x4*>(ds);
      const float* xr = xs + e_ci * XW + e_k;
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
      int t = 0;
      for (; t + 1 < p.Tout; t += 2) { s0 += dr[t] * xr[t * p.s]; s1 += dr[t + 1] * xr[(t + 1) * p.s]; }
      if (t < p.Tout) s0 += dr[t] * xr[t * p.s];
      acc += s0 + s1;
    }
    if (tid >= 252) {                              // bias partials on four otherwise idle threads
      float sb = 0.f;
      for (int t = 0; t < p.Tout; ++t) sb += ds[t * 4 + (tid - 252)];
      bacc += sb;
    }
  }
  float* slab = p.slab + (long)split * p.slab_stride;
  if (e_live) {
#pragma unroll
    for (int co = 0; co < 4; ++co) slab[(long)g * nel + (co * 4 + e_ci) * p.K + e_k] = acc[co];
  }
  if (tid >= 252) slab[(long)p.G * nel + g * 4 + (tid - 252)] = bacc;
}

// ---------------------------------------------------------------------------------------------- host
static inline int sg_cs(int s, int K) { return ((64 + (K + s - 1) / s + 1 + 15) / 16) * 16; }

hipError_t launch_small_group_fwd(SmallGroupP p, hipStream_t st) {
  p.J = (p.K + p.s - 1) / p.s;
  p.CS = sg_cs(p.s, p.K);
  const size_t lds = (size_t)4 * (4 * p.s * p.CS + 16 * p.K) * sizeof(float);
  if (lds > 64 * 1024 || 16 * p.K > 768) return hipErrorNotSupported;
  const dim3 grid((p.G + 3) / 4, p.B, (p.Tout + 63) / 64);
  if (p.s == 2) { TDVC_TRACE(small_group_fwd_kernel<2>); hipLaunchKernelGGL(small_group_fwd_kernel<2>, grid, dim3(256), lds, st, p); }
  else if (p.s == 4) { TDVC_TRACE(small_group_fwd_kernel<4>); hipLaunchKernelGGL(small_group_fwd_kernel<4>, grid, dim3(256), lds, st, p); }
  else if (p.s == 8) { TDVC_TRACE(small_group_fwd_kernel<8>); hipLaunchKernelGGL(small_group_fwd_kernel<8>, grid, dim3(256), lds, st, p); }
  else return hipErrorNotSupported;
  return hipGetLastError();
}

hipError_t launch_small_group_dgrad(SmallGroupP p, hipStream_t st) {
  p.J = (p.K + p.s - 1) / p.s;
  if (p.s > 8 || 16 * p.J * p.s > 768 || p.J > 64) return hipErrorNotSupported;
  const size_t lds = (size_t)4 * (4 * (64 + p.J) + 16 * p.J * p.s) * sizeof(float);
  if (lds > 64 * 1024) return hipErrorNotSupported;
  const int M = (p.Tin - 1 + p.pad) / p.s + 1;
  const dim3 grid((p.G + 3) / 4, p.B, (M + 63) / 64);
  if (p.s == 2) { TDVC_TRACE(small_group_dgrad_kernel<2>); hipLaunchKernelGGL(small_group_dgrad_kernel<2>, grid, dim3(256), lds, st, p); }
  else if (p.s == 4) { TDVC_TRACE(small_group_dgrad_kernel<4>); hipLaunchKernelGGL(small_group_dgrad_kernel<4>, grid, dim3(256), lds, st, p); }
  else if (p.s == 8) { TDVC_TRACE(small_group_dgrad_kernel<8>); hipLaunchKernelGGL(small_group_dgrad_kernel<8>, grid, dim3(256), lds, st, p); }
  else return hipErrorNotSupported;
  return hipGetLastError();
}

size_t small_group_wgrad_workspace(int B, int G, int K) {
  const int nsplit = B >= 8 ? 4 : 1;
  return (size_t)nsplit * ((size_t)G * 16 * K + (size_t)G * 4) * sizeof(float);
}

hipError_t launch_small_group_wgrad(SmallGroupP p, float* dw, float* dbias, void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (4 * p.K > 252 || 4 * (p.Tin + 2 * p.pad + p.s) > 6 * 256 || 4 * p.Tout > 256) return hipErrorNotSupported;
  p.nsplit = p.B >= 8 ? 4 : 1;
  const long nW = (long)p.G * 16 * p.K, n = nW + (long)p.G * 4;
  if (workspace_bytes < (size_t)p.nsplit * n * sizeof(float) || !workspace) return hipErrorNotSupported;
  const size_t lds = (size_t)(4 * (p.Tin + 2 * p.pad + p.s) + 4 * p.Tout) * sizeof(float);
  if (lds > 64 * 1024) return hipErrorNotSupported;
  p.slab = (float*)workspace; p.slab_stride = n;
  TDVC_TRACE(small_group_wgrad_kernel);
  hipLaunchKernelGGL(small_group_wgrad_kernel, dim3(p.G, p.nsplit), dim3(256), lds, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // bias partials ride behind the weights; without a bias gradient only the weights are folded
  return launch_slab_reduce(p.slab, p.nsplit, n, dbias ? n : nW, dw, (int)nW, nW, st, nW, dbias);
}

}  // namespace tdvc
