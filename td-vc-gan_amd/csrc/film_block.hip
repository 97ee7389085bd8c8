// Fused forward of one FiLM residual block (model/generator.py:96-111) for the narrow, long-sequence end of the trunk
// (16 channels at T = 16000): the dilated Conv1d and its consumers in ONE launch,
//     h   = conv_kd(LeakyReLU(x)) + b1                       (reflect padding; h is stored: the backward pass reads it)
//     out = scale * (conv_1x1(LeakyReLU(h * (1 + gamma) + beta)) + b2 + x) + acc
// instead of two launches that each pay the fixed cost of a short HBM-bound kernel (~8 us of a 22 us launch) and move h
// through HBM twice. Same arithmetic, in the same order, as conv_lean_kernel<1,4,1,4,ACT,FWD> followed by
// conv_lean_kernel<1,4,1,4,FILM,FWD>: the results are bit-identical to the two-launch path (tests/test_conv_ops_gpu.py).
//
// Block = 16 channels x 256 time steps of one sample, 4 waves x 64 columns. After the dilated conv's MFMAs every wave holds
// its h tile in the accumulator layout (lane = channel, 4 consecutive steps); FiLM + LeakyReLU are applied there, the result
// is exchanged through LDS (the input tile's space, after a block barrier) and the second product (K = 16: 4 MFMA steps) runs on it
// with the operand roles SWAPPED (D2[co][t], rows = channels): sub-tile n, column l stands for time step 4l + n, so a lane's four
// accumulators of one channel are 4 consecutive steps and every load / store of the final epilogue (residual, MRF sum, output)
// covers 4 channel rows x 256 contiguous bytes per wave instruction -- the shape tools/store_shape_bench.hip measures at 1.5x the
// accumulator layout's 16 rows x 64 B. For that the h2 patch is kept as 4 phase planes [n][channel][l] (time 4l + n).
#include "conv_common.h"
#include "film_block.h"

namespace tdvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FB_NT = 256, FB_PS = 17, FB_PL = 16 * FB_PS;   // time tile; h2 patch: 4 phase planes of [16 channels][16 + 1] per wave
constexpr int FB_PATCH = 4 * FB_PL;
constexpr int FB_XSP = 80, FB_XPL = 16 * FB_XSP;   // input tile: 4 phase planes of [16 channels][80] (16 mod 64: the 4 k-rows of a fragment read hit disjoint banks)
constexpr int FB_XVP = 8, FB_WVP = 4;       // max row-walk passes of the input tile / the conv weight tile

template <bool FILM>
__global__ __launch_bounds__(256, 5) void film_block_fwd_kernel(const FilmBlockP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int N_REP = 4;
  float* xs = smem;                               // [16][XS] input tile, later the 4 waves' h2 patches (4 phase planes each)
  float* ws = smem + p.xs_floats;                 // [16][WS] dilated-conv weights
  float* w2s = ws + p.wnp * p.wrp * p.WS;         // [16][18] 1x1 weights, row = output channel (behind every row the weight walk stages)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * FB_NT, b = blockIdx.z;
  const int wcol0 = wave * 16 * N_REP;
  const int q0 = n0 + p.lo;
  const int jc = p.K * 16;

  // ---- stage: input tile (LeakyReLU on the way in), conv weights, 1x1 weights
  const RowWalk xw = make_walk(tid, p.span >> 2, p.xrp, p.T, p.XS, q0, p.T);
  const RowWalk ww = make_walk(tid, jc >> 2, p.wrp, jc, p.WS, 0, 1 << 30);
  RegTile<FB_XVP> xr;
  RegTile<FB_WVP> wr;
  const float* xrow0 = p.x + (long)b * p.x_bs;
  {
    const srd_t rx = make_srd(xrow0, 16 * p.T * 4);
    walk_issue<FB_XVP>(xr, rx, xw, p.xnp);
    const srd_t rw = make_srd(p.w1, 16 * jc * 4);
    walk_issue<FB_WVP>(wr, rw, ww, p.wnp);
  }
  const float w2v = p.w2[tid];                     // 256 threads = 16 x 16 weights
  // input tile as 4 phase planes: staged column i (time q0 + i) lives at xs[(i & 3) * FB_XPL + row * FB_XSP + (i >> 2)], so that the
  // fragment read of sub-tile n / tap j -- time step 4l + n + j*d of lane l -- is one phase plane at consecutive indices
  {
    const int nvec = p.span >> 2;
    const int rsub = (int)(((float)tid + 0.5f) * (1.0f / (float)nvec));
    const int vv = tid - rsub * nvec;
    if (xw.active) {
#pragma unroll
      for (int i = 0; i < FB_XVP; ++i) {
        if (i < walk_opaque(p.xnp) && i * p.xrp + rsub < 16) {
          float* d = xs + (i * p.xrp + rsub) * FB_XSP + vv;
#pragma unroll
          for (int q = 0; q < 4; ++q) { const float v = xr.v[i][q]; d[q * FB_XPL] = fmaxf(v, v * p.slope); }
        }
      }
    }
  }
  if (q0 < 0 || q0 + p.span > p.T) {               // reflect halo of the sample's first / last tile
    const int nl = q0 < 0 ? -q0 : 0;
    const int rfirst = p.T - q0;
    const int nr = rfirst < p.span ? p.span - rfirst : 0;
    const int nh = nl + nr;
    __syncthreads();
    const float inv = 1.0f / (float)nh;
    for (int e = tid; e < 16 * nh; e += 256) {
      const int r = (int)(((float)e + 0.5f) * inv);
      const int hh = e - r * nh;
      const int i = hh < nl ? hh : rfirst + (hh - nl);
      int q = q0 + i;
      q = q < 0 ? -q : 2 * (p.T - 1) - q;
      float v = (q >= 0 && q < p.T) ? xrow0[(long)r * p.T + q] : 0.f;
      v = v > 0.f ? v : v * p.slope;
      xs[(i & 3) * FB_XPL + r * FB_XSP + (i >> 2)] = v;
    }
  }
  walk_commit_w<FB_WVP>(wr, ww, p.wnp, ws);
  w2s[(tid >> 4) * 18 + (tid & 15)] = w2v;
  __syncthreads();

  // ---- dilated conv: D[co][t] += W1[co][k] * X'[k][t], one step = 4 channels of one tap; sub-tile n, column l = time step 4l + n
  f32x4 acc[N_REP];
#pragma unroll
  for (int n = 0; n < N_REP; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* w_lane = ws + ln * p.WS + kq * p.K;
    const float* x_lane = xs + kq * FB_XSP + (wcol0 >> 2) + ln;
    const int nsteps = p.K * 4;
    int sj = 0, scs = 0, woff = 0, cso = 0;
    int xo[N_REP];                                  // per sub-tile: phase plane + index shift of the current tap (wave-uniform)
#pragma unroll
    for (int n = 0; n < N_REP; ++n) { const int e = n + p.i0; xo[n] = (e & 3) * FB_XPL + (e >> 2); }
    auto advance = [&]() {
      if (++scs == 4) {
        scs = 0; ++sj; woff = sj; cso = 0;
#pragma unroll
        for (int n = 0; n < N_REP; ++n) { const int e = n + sj * p.d + p.i0; xo[n] = (e & 3) * FB_XPL + (e >> 2); }
      } else { woff += 4 * p.K; cso += 4 * FB_XSP; }
    };
    float wv[2], xv[2][N_REP];
    auto load_frag = [&](int buf) {
      wv[buf] = w_lane[woff];
#pragma unroll
      for (int n = 0; n < N_REP; ++n) xv[buf][n] = x_lane[xo[n] + cso];
    };
    auto mma = [&](int buf) {
#pragma unroll
      for (int n = 0; n < N_REP; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[buf], xv[buf][n], acc[n], 0, 0, 0);
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
  }

  // ---- h = conv + b1 (stored), h2 = LeakyReLU(h * (1 + gamma) + beta) into the wave's LDS patch; lane = (channels 4*kq .. 4*kq+3,
  // time steps t0 .. t0+3 with t0 = 4*ln): every global access of this phase covers 4 rows x 256 B per wave instruction
  __syncthreads();                                  // every wave is done reading the input tile
  float* hp = xs + wave * FB_PATCH;
  {
    const int t0 = n0 + wcol0 + 4 * ln;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = 4 * kq + r;
      const float b1 = p.b1 ? p.b1[co] : 0.f;
      const long ro = (long)co * p.T;
      f32x4 v = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += b1;
      f32x4 h2 = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (t0 < p.T) {
        *reinterpret_cast<f32x4*>(p.h + (long)b * p.h_bs + ro + t0) = v;
        if (FILM) {
          const float* gp = p.gb + (long)b * p.gb_bs + ro + t0;
          const f32x4 ga = *reinterpret_cast<const f32x4*>(gp);
          const f32x4 be = *reinterpret_cast<const f32x4*>(gp + (long)16 * p.T);
#pragma unroll
          for (int q = 0; q < 4; ++q) { const float u = v[q] * (1.f + ga[q]) + be[q]; h2[q] = fmaxf(u, u * p.slope); }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) h2[q] = fmaxf(v[q], v[q] * p.slope);
        }
      }
      // element n of this float4 is time step 4*ln + n: phase plane n, column ln
#pragma unroll
      for (int n = 0; n < 4; ++n) hp[n * FB_PL + co * FB_PS + ln] = h2[n];
    }
  }
  // wave-private exchange through LDS: DS operations of one wave execute in order; the fences keep the compiler from moving
  // the fragment reads (other lanes' data) above the writes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- 1x1 conv: D2[co2][t] += W2[co2][k] * H2[k][t], K = 16 channels = 4 steps; sub-tile n, column l = time step 4l + n
  f32x4 acc2[N_REP];
#pragma unroll
  for (int n = 0; n < N_REP; ++n) acc2[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cs = 0; cs < 4; ++cs) {
    const float wv2 = w2s[ln * 18 + cs * 4 + kq];
#pragma unroll
    for (int n = 0; n < N_REP; ++n) {
      const float xv2 = hp[n * FB_PL + (cs * 4 + kq) * FB_PS + ln];
      acc2[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv2, xv2, acc2[n], 0, 0, 0);
    }
  }

  // ---- out = scale * (D2 + b2 + x) + acc: lane = (channels 4*kq .. 4*kq+3, time steps t0 .. t0+3 with t0 = 4*ln)
  {
    const int t0 = n0 + wcol0 + 4 * ln;
    if (t0 < p.T) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 4 * kq + r;
        const float b2 = p.b2 ? p.b2[co] : 0.f;
        const long oi = (long)co * p.T + t0;
        f32x4 v = (f32x4){acc2[0][r], acc2[1][r], acc2[2][r], acc2[3][r]};
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += b2;
        const f32x4 res = *reinterpret_cast<const f32x4*>(p.x + (long)b * p.x_bs + oi);
        v += res;
        v *= p.scale;
        if (p.add) { const f32x4 a4 = *reinterpret_cast<const f32x4*>(p.add + (long)b * p.add_bs + oi); v += a4 * 1.f; }
        *reinterpret_cast<f32x4*>(p.y + (long)b * p.y_bs + oi) = v;
      }
    }
  }
}

static inline void fb_walk_geometry(int rows, int nvec, int* rp, int* np) {
  *rp = nvec <= 256 ? 256 / nvec : 0;
  if (*rp > rows) *rp = rows;
  *np = *rp ? (rows + *rp - 1) / *rp : 1 << 20;
}

// hipErrorNotSupported: outside the fused kernel's contract (the caller runs the two-launch path).
hipError_t launch_film_block_fwd(FilmBlockP p, int B, hipStream_t st) {
  if (p.T < 2 * FB_NT || (p.T & 3) || p.K < 1 || p.K > 15 || p.d < 1) return hipErrorNotSupported;
  const int pad = (p.K - 1) * p.d / 2;
  if (2 * pad != (p.K - 1) * p.d || pad >= p.T) return hipErrorNotSupported;
  const int first = -pad;
  const int lo = -(((pad) + 3) / 4 * 4);
  const int hi = (p.K - 1) * p.d - pad;
  p.lo = lo; p.i0 = first - lo;
  p.span = ((FB_NT + hi - lo) + 3) / 4 * 4;
  p.XS = ((p.span + 15) / 32) * 32 + 16;
  p.WS = p.K * 16 + 2;
  fb_walk_geometry(16, p.span / 4, &p.xrp, &p.xnp);
  fb_walk_geometry(16, p.K * 16 / 4, &p.wrp, &p.wnp);
  if (p.xnp > FB_XVP || p.wnp > FB_WVP) return hipErrorNotSupported;
  if ((p.span >> 2) > FB_XSP) return hipErrorNotSupported;
  int xs_floats = 4 * FB_XPL;                                // input tile: 4 phase planes
  if (xs_floats < 4 * FB_PATCH) xs_floats = 4 * FB_PATCH;    // ... and the same space later holds the 4 waves' h2 patches
  p.xs_floats = xs_floats;
  const size_t lds = (size_t)(xs_floats + p.wnp * p.wrp * p.WS + 16 * 18) * sizeof(float);
  if (p.wnp * p.wrp < 16) return hipErrorNotSupported;
  dim3 grid((p.T + FB_NT - 1) / FB_NT, 1, B);
  if (p.gb) {
    auto k = film_block_fwd_kernel<true>;
    TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  } else {
    auto k = film_block_fwd_kernel<false>;
    TDVC_BIG_LDS_ONCE(k); TDVC_TRACE(k);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
  }
  return hipGetLastError();
}

}  // namespace tdvc

using namespace tdvc;
#include "../../include/tdvc.h"
#include "api_util.h"

extern "C" int tdvc_film_block_fwd(const tdvc_film_block_args* a, void* stream) {
  if (!a || !a->x || !a->w1 || !a->h || !a->w2 || !a->y) return tdvc_fail(TDVC_EINVAL, "film_block_fwd: null pointer");
  if (a->B <= 0 || a->T <= 0 || a->K <= 0 || a->dilation <= 0) return tdvc_fail(TDVC_EINVAL, "film_block_fwd: bad shape");
  auto ok = [](const void* ptr, int64_t bs) { return !ptr || ((((uintptr_t)ptr) & 15) == 0 && (bs & 3) == 0 && bs < (1LL << 31)); };
  if (a->C != 16 || !ok(a->x, a->x_bs) || !ok(a->h, a->h_bs) || !ok(a->gb, a->gb_bs) || !ok(a->add, a->add_bs) || !ok(a->y, a->y_bs) ||
      !ok(a->w1, 0) || !ok(a->w2, 0))
    return tdvc_fail(TDVC_EUNSUPPORTED, "film_block_fwd: needs 16 channels and 16-byte aligned operands");
  if (g_force_tile >= 0) return tdvc_fail(TDVC_EUNSUPPORTED, "film_block_fwd: a lean tile is pinned (test-only tdvc_debug_force_tile)");
  FilmBlockP p = {};
  p.x = a->x; p.w1 = a->w1; p.b1 = a->b1; p.h = a->h; p.gb = a->gb; p.w2 = a->w2; p.b2 = a->b2; p.add = a->add; p.y = a->y;
  p.x_bs = (int)a->x_bs; p.h_bs = (int)a->h_bs; p.gb_bs = (int)a->gb_bs; p.add_bs = (int)a->add_bs; p.y_bs = (int)a->y_bs;
  p.T = a->T; p.K = a->K; p.d = a->dilation; p.slope = a->slope; p.scale = a->scale == 0.f ? 1.f : a->scale;
  const hipError_t e = launch_film_block_fwd(p, a->B, (hipStream_t)stream);
  if (e == hipErrorNotSupported) return tdvc_fail(TDVC_EUNSUPPORTED, "film_block_fwd: shape outside the fused kernel's contract");
  return e == hipSuccess ? TDVC_OK : tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
}
