// Non-convolution kernels of the train step: multi-tensor weight norm, fused AdamW, channel
// L2-normalise, class-channel gather, conditioning concat, conditional instance norm, loss
// reductions (LSGAN MSE-to-constant, L1 pairs, log-mel helpers, contrastive InfoNCE).
// All are HBM-bound streaming / reduction kernels: coalesced along T, wave64 shuffle reductions.
#include "../../include/tdvc.h"
#include "api_util.h"
#include "conv_common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[256] = "";
int tdvc_fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
  return code;
}
extern "C" const char* tdvc_last_error(void) { return g_err; }
extern "C" int tdvc_version(void) { return 200; }

// ------------------------------------------------------------------------------ test-only debug hooks
// Process-global switches used by the parity tests to (a) pin the tile configuration of the lean conv kernel so that
// every template instance the train step can select is checked against the float64 reference at small shapes, and
// (b) record which kernel instantiations the launches of a test actually used. Never touched by the product path.
#include <cxxabi.h>
#include <mutex>
#include <set>
#include <string>
namespace tdvc {
int g_trace_on = 0;
int g_force_tile = -1;
int g_lds_cap = 0;
int g_knob[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static std::mutex g_trace_mu;
static std::set<std::string> g_trace_names;
void trace_kernel(const void* fn) {
  const char* mangled = hipKernelNameRefByPtr(fn, nullptr);
  std::string name;
  if (mangled) {
    int status = 0;
    char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    name = (status == 0 && dem) ? dem : mangled;
    free(dem);
  } else {
    char buf[32]; snprintf(buf, sizeof(buf), "kernel@%p", fn); name = buf;
  }
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_trace_names.insert(name);
}
}  // namespace tdvc
extern "C" void tdvc_debug_force_tile(int cfg) { tdvc::g_force_tile = cfg; }
extern "C" void tdvc_debug_lds_cap(int bytes) { tdvc::g_lds_cap = bytes; }
extern "C" void tdvc_debug_knob(int which, int value) { if (which >= 0 && which < 8) tdvc::g_knob[which] = value; }
namespace tdvc {
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned word, int nwords, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned* u = reinterpret_cast<unsigned*>(smem);
  for (int i = threadIdx.x; i < nwords; i += 256) u[i] = word;
  __syncthreads();
  // keep the stores alive and make the block stay long enough that every CU hosts several of them
  unsigned acc = 0;
  for (int i = threadIdx.x; i < nwords; i += 256) acc ^= u[i];
  if (acc == 0x12345u && sink) sink[0] = acc;
}
}  // namespace tdvc
extern "C" int tdvc_debug_poison_lds(uint32_t word, void* stream) {
  auto k = tdvc::poison_lds_kernel;
  TDVC_BIG_LDS_ONCE(k);
  // 80 KB per block: two blocks per CU cover its 160 KB; 8 rounds of 512 blocks so that every CU is visited repeatedly
  hipLaunchKernelGGL(k, dim3(4096), dim3(256), 80 * 1024, (hipStream_t)stream, (unsigned)word, 80 * 1024 / 4, (unsigned*)nullptr);
  TDVC_CHECK_LAUNCH();
  return TDVC_OK;
}
namespace tdvc { __global__ void pmc_marker_kernel(int*) {} }
// TOOLS-ONLY: an empty dispatch that separates the launches of one table entry from the next in a rocprofv3 trace (tools/pmc_traffic.py)
extern "C" int tdvc_debug_marker(void* stream) {
  hipLaunchKernelGGL(tdvc::pmc_marker_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (int*)nullptr);
  TDVC_CHECK_LAUNCH();
  return TDVC_OK;
}
extern "C" void tdvc_debug_trace(int on) {
  std::lock_guard<std::mutex> lk(tdvc::g_trace_mu);
  if (on == 1) tdvc::g_trace_names.clear();
  tdvc::g_trace_on = on ? 1 : 0;
}
extern "C" size_t tdvc_debug_trace_dump(char* buf, size_t cap) {
  std::lock_guard<std::mutex> lk(tdvc::g_trace_mu);
  std::string all;
  for (const std::string& n : tdvc::g_trace_names) { all += n; all += '\n'; }
  if (buf && cap) { const size_t n = all.size() < cap - 1 ? all.size() : cap - 1; memcpy(buf, all.data(), n); buf[n] = 0; }
  return all.size() + 1;
}

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide sum for 256-thread blocks; result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* sh /* >= 4 floats */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  const int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

// ------------------------------------------------------------------------------ weight norm
// one wave per weight row (dim-0 slice); rows of all tensors of a model in one launch
__global__ __launch_bounds__(256) void wn_fwd_kernel(const float* __restrict__ params, float* __restrict__ w,
                                                     const int64_t* row_voff, const int64_t* row_goff,
                                                     const int64_t* row_woff, const int32_t* row_len, int nrows,
                                                     float* __restrict__ wt, const int64_t* row_tbase,
                                                     const int64_t* row_tstride, const int32_t* row_k) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int lane = threadIdx.x & 63;
  const float* v = params + row_voff[row];
  float* o = w + row_woff[row];
  const int n = row_len[row];
  float ss = 0.f;
  for (int i = lane; i < n; i += 64) { float t = v[i]; ss += t * t; }
  ss = wave_sum(ss);
  const float sc = params[row_goff[row]] / sqrtf(ss);
  const int K = wt ? row_k[row] : 0;
  if (K > 0) {
    float* tb = wt + row_tbase[row];
    const long ts = row_tstride[row];
    for (int i = lane; i < n; i += 64) {
      const float val = v[i] * sc;
      o[i] = val;
      const int ci = i / K;
      tb[(long)ci * ts + (i - ci * K)] = val;
    }
  } else {
    for (int i = lane; i < n; i += 64) o[i] = v[i] * sc;
  }
}

__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ params, const float* __restrict__ dw,
                                                     float* __restrict__ grads, const int64_t* row_voff,
                                                     const int64_t* row_goff, const int64_t* row_woff,
                                                     const int32_t* row_len, int nrows, int accumulate) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int lane = threadIdx.x & 63;
  const float* v = params + row_voff[row];
  const float* d = dw + row_woff[row];
  float* dv = grads + row_voff[row];
  const int n = row_len[row];
  float ss = 0.f, dot = 0.f;
  for (int i = lane; i < n; i += 64) { float t = v[i]; ss += t * t; dot += t * d[i]; }
  ss = wave_sum(ss); dot = wave_sum(dot);
  const float nrm = sqrtf(ss), g = params[row_goff[row]];
  const float a = g / nrm, bcoef = g * dot / (nrm * ss);
  for (int i = lane; i < n; i += 64) {
    const float t = a * d[i] - bcoef * v[i];
    dv[i] = accumulate ? dv[i] + t : t;
  }
  if (lane == 0) {
    float* dg = grads + row_goff[row];
    const float t = dot / nrm;
    dg[0] = accumulate ? dg[0] + t : t;
  }
}

// ------------------------------------------------------------------------------ AdamW
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, float gscale, const int32_t* step_ptr,
                                                    const float* gscale_dev) {
  if (gscale_dev) gscale *= gscale_dev[0];   // gradient-clipping coefficient computed on the device (tdvc_grad_clip_coef)
  if (step_ptr) {   // device-resident step counter: keeps a captured hipGraph valid across replays
    const float st = (float)step_ptr[0];
    bc1 = 1.f - powf(b1, st); bc2_sqrt = sqrtf(1.f - powf(b2, st));
  }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gr = g[i] * gscale;
    float pv = p[i] * (1.f - lr * wd);
    const float mv = b1 * m[i] + (1.f - b1) * gr;
    const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pv -= (lr / bc1) * (mv / denom);
    p[i] = pv; m[i] = mv; v[i] = vv;
  }
}

// ------------------------------------------------------------------------------ element-wise
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* x, float* y, float* inv, int C, int T, float eps) {
  const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  const float* xb = x + (long)b * C * T + t;
  float ss = 0.f;
  for (int c = 0; c < C; ++c) { float v = xb[(long)c * T]; ss += v * v; }
  const float r = 1.f / fmaxf(sqrtf(ss), eps);
  inv[(long)b * T + t] = r;
  float* yb = y + (long)b * C * T + t;
  for (int c = 0; c < C; ++c) yb[(long)c * T] = xb[(long)c * T] * r;
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* y, const float* inv, const float* dy, float* dx, int C, int T) {
  const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  const long o = (long)b * C * T + t;
  float dot = 0.f;
  for (int c = 0; c < C; ++c) dot += dy[o + (long)c * T] * y[o + (long)c * T];
  const float r = inv[(long)b * T + t];
  for (int c = 0; c < C; ++c) dx[o + (long)c * T] = (dy[o + (long)c * T] - y[o + (long)c * T] * dot) * r;
}

__global__ __launch_bounds__(256) void gather_ch_kernel(const float* x, const int64_t* label, float* y, int C, int T, int bwd) {
  const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  const int l = (int)label[b];
  if (!bwd) y[(long)b * T + t] = x[((long)b * C + l) * T + t];
  else for (int c = 0; c < C; ++c) y[((long)b * C + c) * T + t] = (c == l) ? x[(long)b * T + t] : 0.f;
}

__global__ __launch_bounds__(256) void concat_cond_kernel(const float* emb, const float* exc, float* c, int Ce, int Cx, int T) {
  const int t = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float v = ch < Ce ? emb[(long)b * Ce + ch] : exc[((long)b * Cx + (ch - Ce)) * T + t];
  c[((long)b * (Ce + Cx) + ch) * T + t] = v;
}

// demb[b][ch] (+)= sum_t dc[b][ch][t]  (one block per (ch,b)); dexc = copy of the tail channels
__global__ __launch_bounds__(256) void concat_cond_bwd_kernel(const float* dc, float* demb, float* dexc, int Ce, int Cx, int T, int acc) {
  __shared__ float sh[4];
  const int ch = blockIdx.x, b = blockIdx.y;
  const float* src = dc + ((long)b * (Ce + Cx) + ch) * T;
  if (ch < Ce) {
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) s += src[t];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) { float* d = demb + (long)b * Ce + ch; *d = acc ? *d + s : s; }
  } else {
    float* dst = dexc + ((long)b * Cx + (ch - Ce)) * T;
    for (int t = threadIdx.x; t < T; t += 256) dst[t] = src[t];
  }
}

// one block per (c, b) row: first element, interior sum, last element
__global__ __launch_bounds__(256) void edge_sum3_kernel(const float* d, float* out, int C, int T) {
  __shared__ float sh[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const float* r = d + ((long)b * C + c) * T;
  float s = 0.f;
  for (int t = 1 + threadIdx.x; t < T - 1; t += 256) s += r[t];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) { float* o = out + ((long)b * C + c) * 3; o[0] = r[0]; o[1] = s; o[2] = r[T - 1]; }
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* a, const float* b, float* y, float alpha, float beta, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

// ------------------------------------------------------------------------------ WaveNet gate (SSL encoder WN stack)
// acts[b][c][t] = tanh(a) * sigmoid(s) with a = xin[b][c][t] (+ g), s = xin[b][H + c][t] (+ g') -- the reference's
// fused_add_tanh_sigmoid_multiply (model/ssl_encoder.py:8-15). One thread per (b, c, t); T fastest -> coalesced.
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* xin, long xin_bs, const float* g, long g_bs, float* acts, long acts_bs,
                                                       int H, int T, long n) {
  const long HT = (long)H * T;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / HT, r = i - b * HT;
    float a = xin[b * xin_bs + r], s = xin[b * xin_bs + HT + r];
    if (g) { a += g[b * g_bs + r]; s += g[b * g_bs + HT + r]; }
    acts[b * acts_bs + r] = tanhf(a) * (1.f / (1.f + expf(-s)));
  }
}
// d_xin[c] = d_acts * sig * (1 - tanh^2), d_xin[H + c] = d_acts * tanh * sig * (1 - sig)
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* xin, long xin_bs, const float* g, long g_bs, const float* dacts, long dacts_bs,
                                                       float* dxin, long dxin_bs, int H, int T, long n) {
  const long HT = (long)H * T;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / HT, r = i - b * HT;
    float a = xin[b * xin_bs + r], s = xin[b * xin_bs + HT + r];
    if (g) { a += g[b * g_bs + r]; s += g[b * g_bs + HT + r]; }
    const float th = tanhf(a), sg = 1.f / (1.f + expf(-s)), d = dacts[b * dacts_bs + r];
    dxin[b * dxin_bs + r] = d * sg * (1.f - th * th);
    dxin[b * dxin_bs + HT + r] = d * th * sg * (1.f - sg);
  }
}

__global__ __launch_bounds__(256) void fill_kernel(float* y, float v, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = v;
}

// ------------------------------------------------------------------------------ conditional instance norm
// one block per (c, b) row; biased variance, eps inside the sqrt (nn.InstanceNorm1d, affine=False)
__global__ __launch_bounds__(256) void cin_fwd_kernel(const float* x, const float* gb, float* y, float* mean, float* rstd,
                                                      int C, int T, int Tg, float eps) {
  __shared__ float sh[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const float* xr = x + ((long)b * C + c) * T;
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) s += xr[t];
  const float mu = block_sum(s, sh) / T;
  float q = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) { float d = xr[t] - mu; q += d * d; }
  const float rs = rsqrtf(block_sum(q, sh) / T + eps);
  if (threadIdx.x == 0) { mean[(long)b * C + c] = mu; rstd[(long)b * C + c] = rs; }
  const float* ga = gb + ((long)b * 2 * C + c) * Tg;
  const float* be = gb + ((long)b * 2 * C + C + c) * Tg;
  float* yr = y + ((long)b * C + c) * T;
  for (int t = threadIdx.x; t < T; t += 256) {
    const int tg = Tg == 1 ? 0 : t;
    yr[t] = (1.f + ga[tg]) * ((xr[t] - mu) * rs) + be[tg];
  }
}

__global__ __launch_bounds__(256) void cin_bwd_kernel(const float* x, const float* gb, const float* dy, const float* mean,
                                                      const float* rstd, float* dx, float* dgb, int C, int T, int Tg) {
  __shared__ float sh[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const long ro = ((long)b * C + c) * T;
  const float mu = mean[(long)b * C + c], rs = rstd[(long)b * C + c];
  const float* ga = gb + ((long)b * 2 * C + c) * Tg;
  float* dga = dgb + ((long)b * 2 * C + c) * Tg;
  float* dbe = dgb + ((long)b * 2 * C + C + c) * Tg;
  float s1 = 0.f, s2 = 0.f, sg = 0.f, sb = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    const int tg = Tg == 1 ? 0 : t;
    const float xh = (x[ro + t] - mu) * rs, d = dy[ro + t], dh = d * (1.f + ga[tg]);
    s1 += dh; s2 += dh * xh;
    if (Tg == 1) { sg += d * xh; sb += d; } else { dga[t] = d * xh; dbe[t] = d; }
  }
  const float m1 = block_sum(s1, sh) / T, m2 = block_sum(s2, sh) / T;
  if (Tg == 1) {
    const float tg_ = block_sum(sg, sh), tb_ = block_sum(sb, sh);
    if (threadIdx.x == 0) { dga[0] = tg_; dbe[0] = tb_; }
  }
  for (int t = threadIdx.x; t < T; t += 256) {
    const int tg = Tg == 1 ? 0 : t;
    const float xh = (x[ro + t] - mu) * rs, dh = dy[ro + t] * (1.f + ga[tg]);
    dx[ro + t] = rs * (dh - m1 - xh * m2);
  }
}

// Single-pass variants: the whole (b, c) row lives in registers (NV float4 per thread, T <= 1024 * NV, rows 16-byte aligned), so
// x (and dy) are read from HBM once instead of three (two) times: 8 B per element forward, 12-20 B backward. Same arithmetic
// as the kernels above (mean first, then the centred sum of squares).
typedef float cin_f4 __attribute__((ext_vector_type(4)));
template <int NV>
__global__ __launch_bounds__(256) void cin_fwd_row_kernel(const float* x, const float* gb, float* y, float* mean, float* rstd,
                                                          int C, int T, int Tg, float eps) {
  __shared__ float sh[4];
  const int c = blockIdx.x, b = blockIdx.y, nv = T >> 2;
  const cin_f4* xr = reinterpret_cast<const cin_f4*>(x + ((long)b * C + c) * T);
  cin_f4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = threadIdx.x + i * 256;
    v[i] = e < nv ? xr[e] : (cin_f4){0.f, 0.f, 0.f, 0.f};
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mu = block_sum(s, sh) / T;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (threadIdx.x + i * 256 < nv) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = v[i][k] - mu; q += d * d; }
    }
  }
  const float rs = rsqrtf(block_sum(q, sh) / T + eps);
  if (threadIdx.x == 0) { mean[(long)b * C + c] = mu; rstd[(long)b * C + c] = rs; }
  const float* ga = gb + ((long)b * 2 * C + c) * Tg;
  const float* be = gb + ((long)b * 2 * C + C + c) * Tg;
  cin_f4* yr = reinterpret_cast<cin_f4*>(y + ((long)b * C + c) * T);
  const float g1 = Tg == 1 ? 1.f + ga[0] : 0.f, b1 = Tg == 1 ? be[0] : 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e >= nv) continue;
    cin_f4 o;
    if (Tg == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = g1 * ((v[i][k] - mu) * rs) + b1;
    } else {
      const cin_f4 g4 = reinterpret_cast<const cin_f4*>(ga)[e], b4 = reinterpret_cast<const cin_f4*>(be)[e];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (1.f + g4[k]) * ((v[i][k] - mu) * rs) + b4[k];
    }
    yr[e] = o;
  }
}

template <int NV>
__global__ __launch_bounds__(256) void cin_bwd_row_kernel(const float* x, const float* gb, const float* dy, const float* mean,
                                                          const float* rstd, float* dx, float* dgb, int C, int T, int Tg) {
  __shared__ float sh[4];
  const int c = blockIdx.x, b = blockIdx.y, nv = T >> 2;
  const long ro = ((long)b * C + c) * T;
  const float mu = mean[(long)b * C + c], rs = rstd[(long)b * C + c];
  const float* ga = gb + ((long)b * 2 * C + c) * Tg;
  float* dga = dgb + ((long)b * 2 * C + c) * Tg;
  float* dbe = dgb + ((long)b * 2 * C + C + c) * Tg;
  const cin_f4* xr = reinterpret_cast<const cin_f4*>(x + ro);
  const cin_f4* dr = reinterpret_cast<const cin_f4*>(dy + ro);
  cin_f4 xh[NV], dh[NV];                           // normalised input, gradient wrt it
  float s1 = 0.f, s2 = 0.f, sg = 0.f, sb = 0.f;
  const float g1 = Tg == 1 ? 1.f + ga[0] : 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = threadIdx.x + i * 256;
    xh[i] = dh[i] = (cin_f4){0.f, 0.f, 0.f, 0.f};
    if (e >= nv) continue;
    const cin_f4 xv = xr[e], d = dr[e];
    cin_f4 g4 = {g1, g1, g1, g1};
    if (Tg != 1) { g4 = reinterpret_cast<const cin_f4*>(ga)[e]; g4 += 1.f; }
    cin_f4 og, ob;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xh[i][k] = (xv[k] - mu) * rs; dh[i][k] = d[k] * g4[k];
      s1 += dh[i][k]; s2 += dh[i][k] * xh[i][k];
      og[k] = d[k] * xh[i][k]; ob[k] = d[k];
      sg += og[k]; sb += ob[k];
    }
    if (Tg != 1) { reinterpret_cast<cin_f4*>(dga)[e] = og; reinterpret_cast<cin_f4*>(dbe)[e] = ob; }
  }
  const float m1 = block_sum(s1, sh) / T, m2 = block_sum(s2, sh) / T;
  if (Tg == 1) {
    const float tg_ = block_sum(sg, sh), tb_ = block_sum(sb, sh);
    if (threadIdx.x == 0) { dga[0] = tg_; dbe[0] = tb_; }
  }
  cin_f4* dxr = reinterpret_cast<cin_f4*>(dx + ro);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e >= nv) continue;
    cin_f4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = rs * (dh[i][k] - m1 - xh[i][k] * m2);
    dxr[e] = o;
  }
}

// ------------------------------------------------------------------------------ gradient clipping, batch roll
// torch.nn.utils.clip_grad_norm_ (train.py:289-290, 489-490) on the flat gradient arena: pass 1 leaves one partial sum of
// squares per block, pass 2 (one block, fixed order: deterministic) turns them into
//   out[0] = min(1, max_norm / (grad_scale * ||g||_2 + 1e-6))   (the clip coefficient),   out[1] = grad_scale * ||g||_2.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, long n, float* part) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += g[i] * g[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* part, int nparts, float max_norm, float gscale, float* out) {
  __shared__ double shd[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += (double)part[i];
  shd[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += shd[i];
    const float total = gscale * (float)sqrt(t);
    const float coef = max_norm / (total + 1e-6f);
    out[0] = coef < 1.f ? coef : 1.f;
    out[1] = total;
  }
}
// util.roll_batches (util/__init__.py:91-102) along the last axis: y[b][c][t] = x[b][c][(t - shift[b]) mod T]
__global__ __launch_bounds__(256) void roll_batches_kernel(const float* x, const int64_t* shift, float* y, int C, int T) {
  const int b = blockIdx.z, c = blockIdx.y;
  const long ro = ((long)b * C + c) * T;
  long sft = shift[b] % T;
  if (sft < 0) sft += T;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < T; t += gridDim.x * 256) {
    long src = t - sft;
    if (src < 0) src += T;
    y[ro + t] = x[ro + src];
  }
}

// ------------------------------------------------------------------------------ loss reductions
// loss_out[0] += weight * mean((x - target)^2)
__global__ __launch_bounds__(256) void mse_const_fwd_kernel(const float* x, long n, float target, float w_over_n, float* out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { float d = x[i] - target; s += d * d; }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) atomicAdd(out, s * w_over_n);
}
__global__ __launch_bounds__(256) void mse_const_bwd_kernel(const float* x, long n, float target, float w2_over_n, const float* up, float* dx) {
  const float u = up ? up[0] : 1.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dx[i] = (x[i] - target) * w2_over_n * u;
}
__global__ __launch_bounds__(256) void l1_fwd_kernel(const float* a, const float* b, long n, float w_over_n, float* out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += fabsf(a[i] - b[i]);
  s = block_sum(s, sh);
  if (threadIdx.x == 0) atomicAdd(out, s * w_over_n);
}
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* a, const float* b, long n, float w_over_n, const float* up, float* da, int acc) {
  const float u = (up ? up[0] : 1.f) * w_over_n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = a[i] - b[i];
    const float g = d > 0.f ? u : (d < 0.f ? -u : 0.f);
    da[i] = acc ? da[i] + g : g;
  }
}

// Multi-tensor L1 (feature matching, util/losses.py:55-68: one term per discriminator feature map): all pairs of a loss in one
// launch. A block owns one L1M_CHUNK-element chunk of one pair (found by a scan of the chunk prefix table, <= 64 entries).
constexpr int L1M_MAX = 64;
constexpr long L1M_CHUNK = 16384;
struct L1Batch { const float* a[L1M_MAX]; const float* b[L1M_MAX]; float* da[L1M_MAX]; long n[L1M_MAX]; float w[L1M_MAX]; int first[L1M_MAX + 1]; int np; };
__device__ __forceinline__ int l1m_find(const L1Batch& q, int chunk) {
  int d = 0;
  while (d + 1 < q.np && chunk >= q.first[d + 1]) ++d;
  return d;
}
__global__ __launch_bounds__(256) void l1_multi_fwd_kernel(const L1Batch q, float* out) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ float sh[4];
  const int d = l1m_find(q, blockIdx.x);
  const long o = (long)(blockIdx.x - q.first[d]) * L1M_CHUNK;
  const long len = min(L1M_CHUNK, q.n[d] - o);
  const float* a = q.a[d] + o; const float* b = q.b[d] + o;
  float s = 0.f;
  if (((((uintptr_t)a) | ((uintptr_t)b)) & 15) == 0) {
    const long n4 = len >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) {
      const f4 x = reinterpret_cast<const f4*>(a)[i], y = reinterpret_cast<const f4*>(b)[i];
      s += (fabsf(x[0] - y[0]) + fabsf(x[1] - y[1])) + (fabsf(x[2] - y[2]) + fabsf(x[3] - y[3]));
    }
    for (long i = (n4 << 2) + threadIdx.x; i < len; i += 256) s += fabsf(a[i] - b[i]);
  } else {
    for (long i = threadIdx.x; i < len; i += 256) s += fabsf(a[i] - b[i]);
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) atomicAdd(out, s * q.w[d]);
}
// da = sign(a - b) * w * upstream for a pair; b == nullptr: da = 0 (the samples of a batched map that the loss does not read)
__global__ __launch_bounds__(256) void l1_multi_bwd_kernel(const L1Batch q, const float* up) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int d = l1m_find(q, blockIdx.x);
  const long o = (long)(blockIdx.x - q.first[d]) * L1M_CHUNK;
  const long len = min(L1M_CHUNK, q.n[d] - o);
  float* da = q.da[d] + o;
  if (!q.b[d]) {
    if ((((uintptr_t)da) & 15) == 0) {
      const long n4 = len >> 2;
      for (long i = threadIdx.x; i < n4; i += 256) reinterpret_cast<f4*>(da)[i] = (f4){0.f, 0.f, 0.f, 0.f};
      for (long i = (n4 << 2) + threadIdx.x; i < len; i += 256) da[i] = 0.f;
    } else {
      for (long i = threadIdx.x; i < len; i += 256) da[i] = 0.f;
    }
    return;
  }
  const float* a = q.a[d] + o; const float* b = q.b[d] + o;
  const float u = (up ? up[0] : 1.f) * q.w[d];
  auto g1 = [u](float x, float y) { const float df = x - y; return df > 0.f ? u : (df < 0.f ? -u : 0.f); };
  if (((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)da)) & 15) == 0) {
    const long n4 = len >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) {
      const f4 x = reinterpret_cast<const f4*>(a)[i], y = reinterpret_cast<const f4*>(b)[i];
      reinterpret_cast<f4*>(da)[i] = (f4){g1(x[0], y[0]), g1(x[1], y[1]), g1(x[2], y[2]), g1(x[3], y[3])};
    }
    for (long i = (n4 << 2) + threadIdx.x; i < len; i += 256) da[i] = g1(a[i], b[i]);
  } else {
    for (long i = threadIdx.x; i < len; i += 256) da[i] = g1(a[i], b[i]);
  }
}

// ---- log-mel helpers (the two GEMMs run on the conv kernels: STFT = strided conv with the
// windowed DFT basis as weight, mel projection = 1x1 conv with the filterbank as weight)
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* x, float* y, int T, int pad) {
  const int Tp = T + 2 * pad, b = blockIdx.y;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < Tp; i += gridDim.x * 256) {
    int q = i - pad;
    if (q < 0) q = -q; else if (q >= T) q = 2 * (T - 1) - q;
    y[(long)b * Tp + i] = x[(long)b * T + q];
  }
}
__global__ __launch_bounds__(256) void reflect_pad_bwd_kernel(const float* dy, float* dx, int T, int pad) {
  const int Tp = T + 2 * pad, b = blockIdx.y;
  for (int q = blockIdx.x * 256 + threadIdx.x; q < T; q += gridDim.x * 256) {
    float s = dy[(long)b * Tp + q + pad];
    if (q >= 1 && q <= pad) s += dy[(long)b * Tp + pad - q];
    if (q >= T - 1 - pad && q <= T - 2) s += dy[(long)b * Tp + pad + 2 * (T - 1) - q];
    dx[(long)b * T + q] = s;
  }
}
// spec [B][2F][N] (re rows 0..F-1, im rows F..2F-1) -> power [B][F][N]
__global__ __launch_bounds__(256) void power_fwd_kernel(const float* spec, float* pw, int F, int N, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / ((long)F * N), r = i - b * (long)F * N;
    const float re = spec[b * 2L * F * N + r], im = spec[b * 2L * F * N + (long)F * N + r];
    pw[i] = re * re + im * im;
  }
}
__global__ __launch_bounds__(256) void power_bwd_kernel(const float* spec, const float* dpw, float* dspec, int F, int N, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / ((long)F * N), r = i - b * (long)F * N;
    const long o = b * 2L * F * N + r;
    const float g = 2.f * dpw[i];
    dspec[o] = g * spec[o];
    dspec[o + (long)F * N] = g * spec[o + (long)F * N];
  }
}
// loss += w/n * sum |log(max(a,floor)) - log(max(b,floor))|
__global__ __launch_bounds__(256) void log_l1_fwd_kernel(const float* a, const float* b, long n, float floor_, float w_over_n, float* out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    s += fabsf(logf(fmaxf(a[i], floor_)) - logf(fmaxf(b[i], floor_)));
  s = block_sum(s, sh);
  if (threadIdx.x == 0) atomicAdd(out, s * w_over_n);
}
__global__ __launch_bounds__(256) void log_l1_bwd_kernel(const float* a, const float* b, long n, float floor_, float w_over_n, const float* up, float* da) {
  const float u = (up ? up[0] : 1.f) * w_over_n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float av = a[i];
    const float d = logf(fmaxf(av, floor_)) - logf(fmaxf(b[i], floor_));
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    da[i] = av > floor_ ? sg * u / av : 0.f;     // clamp passes gradient only above the floor
  }
}

// ------------------------------------------------------------------------------ contrastive InfoNCE
// One block per (direction, sample, time slice): stages Z = anchor-side embedding [C][T] in LDS, loops over its t.
// logits[k] = cos(A[:,t], target_k), target_0 = P[:,t] (other side), target_{1+n} = Z[:, neg(t,n)] (detached).
// dA / dP accumulate with atomics (each tensor is anchor in one direction and positive in the other).
__global__ __launch_bounds__(128) void contrastive_kernel(const float* X, const float* Y, const int32_t* idx_x, const int32_t* idx_y,
                                                          int C, int T, int N, float coef, float* loss, float* dX, float* dY) {
  extern __shared__ float sm[];
  float* Z = sm;                 // [C][T]
  float* zn = Z + C * T;         // [T] column norms of Z
  float* cs = zn + T;            // [N+1] cosines (logits)
  float* dl = cs + N + 1;        // [N+1] dlogits * coef
  float* av = dl + N + 1;        // [C] anchor (normalised)
  float* pv = av + C;            // [C] positive (normalised)
  float* red = pv + C;           // [8]
  const int dir = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* A = (dir == 0 ? X : Y) + (long)b * C * T;
  const float* P = (dir == 0 ? Y : X) + (long)b * C * T;
  float* dA = (dir == 0 ? dX : dY) + (long)b * C * T;
  float* dP = (dir == 0 ? dY : dX) + (long)b * C * T;
  const int32_t* idx = (dir == 0 ? idx_x : idx_y) + (long)b * T * N;
  for (int i = tid; i < C * T; i += 128) Z[i] = A[i];
  __syncthreads();
  for (int t = tid; t < T; t += 128) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) { float v = Z[c * T + t]; s += v * v; }
    zn[t] = fmaxf(sqrtf(s), 1e-8f);
  }
  __syncthreads();
  float loss_acc = 0.f;
  for (int t = blockIdx.z; t < T; t += gridDim.z) {     // time steps are independent given Z
    float ps = 0.f;
    for (int c = tid; c < C; c += 128) { float v = P[c * T + t]; ps += v * v; }
    ps = wave_sum(ps);
    if ((tid & 63) == 0) red[tid >> 6] = ps;
    __syncthreads();
    const float pn = fmaxf(sqrtf(red[0] + red[1]), 1e-8f), an = zn[t];
    for (int c = tid; c < C; c += 128) { av[c] = Z[c * T + t] / an; pv[c] = P[c * T + t] / pn; }
    __syncthreads();
    for (int k = tid; k <= N; k += 128) {
      float s = 0.f;
      if (k == 0) { for (int c = 0; c < C; ++c) s += av[c] * pv[c]; }
      else {
        const int j = idx[t * N + k - 1];
        for (int c = 0; c < C; ++c) s += av[c] * Z[c * T + j];
        s /= zn[j];
      }
      cs[k] = s;
    }
    __syncthreads();
    float mx = -1e30f;
    for (int k = tid; k <= N; k += 128) mx = fmaxf(mx, cs[k]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[2 + (tid >> 6)] = mx;
    __syncthreads();
    mx = fmaxf(red[2], red[3]);
    float se = 0.f;
    for (int k = tid; k <= N; k += 128) se += expf(cs[k] - mx);
    se = wave_sum(se);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = se;
    __syncthreads();
    se = red[4] + red[5];
    if (tid == 0) loss_acc += logf(se) + mx - cs[0];
    float sdc = 0.f;
    for (int k = tid; k <= N; k += 128) {
      const float d = (expf(cs[k] - mx) / se - (k == 0 ? 1.f : 0.f)) * coef;
      dl[k] = d; sdc += d * cs[k];
    }
    sdc = wave_sum(sdc);
    if ((tid & 63) == 0) red[6 + (tid >> 6)] = sdc;
    __syncthreads();
    sdc = red[6] + red[7];
    const float dl0 = dl[0], c0 = cs[0];
    for (int c = tid; c < C; c += 128) {
      float ga = dl0 * pv[c];
      for (int k = 1; k <= N; ++k) { const int j = idx[t * N + k - 1]; ga += dl[k] * Z[c * T + j] / zn[j]; }
      ga = (ga - sdc * av[c]) / an;
      atomicAdd(&dA[c * T + t], ga);
      atomicAdd(&dP[c * T + t], dl0 * (av[c] - c0 * pv[c]) / pn);
    }
    __syncthreads();
  }
  if (tid == 0) atomicAdd(loss, loss_acc * coef);
}


// ------------------------------------------------------------------------------ sine + noise excitation from F0
// util/__init__.py:22-50 (f0_to_excitation): omega = 2 pi f0 / sr per frame (last frame dropped), nearest-upsampled by
// `step`, linearly interpolated (align_corners=False) where both neighbouring frames are voiced, cumulative phase,
// 0.1 sin(phase + phi0) + 0.003 n_v; unvoiced samples (omega == 0) are 0.1/3 n_u. The random draws are inputs.
// One block per sample: per-thread contiguous segment, block scan of the segment sums (double: 16000..71680 terms).
__device__ __forceinline__ double exc_omega(const float* f0, int n, int t, int step, double k, int linear) {
  const int fr = t / step;
  const double wn = k * (double)f0[fr];
  if (!linear) return wn;
  double src = ((double)t + 0.5) / (double)step - 0.5;
  if (src < 0.0) src = 0.0;
  int i0 = (int)src; if (i0 > n - 1) i0 = n - 1;
  const int i1 = i0 + 1 < n ? i0 + 1 : n - 1;
  const double lam = src - (double)i0;
  const double w0 = k * (double)f0[i0], w1 = k * (double)f0[i1];
  return (w0 > 0.0 && w1 > 0.0) ? w0 * (1.0 - lam) + w1 * lam : wn;
}

__global__ __launch_bounds__(256) void f0_excitation_kernel(const float* f0, const float* noise_v, const float* noise_u, const float* phi0,
                                                            float* exc, int nf1, int step, float sr, int linear) {
  __shared__ double part[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = nf1 - 1, T = n * step;
  const float* f = f0 + (long)b * nf1;
  const double k = 6.283185307179586 / (double)sr;
  const int per = (T + 255) / 256;
  const int t0 = tid * per, t1 = min(T, t0 + per);
  double s = 0.0;
  for (int t = t0; t < t1; ++t) s += exc_omega(f, n, t, step, k, linear);
  part[tid] = s;
  __syncthreads();
  if (tid == 0) { double run = 0.0; for (int i = 0; i < 256; ++i) { const double v = part[i]; part[i] = run; run += v; } }
  __syncthreads();
  double ph = part[tid] + (double)phi0[0];
  for (int t = t0; t < t1; ++t) {
    const double w = exc_omega(f, n, t, step, k, linear);
    ph += w;
    const long i = (long)b * T + t;
    exc[i] = w == 0.0 ? noise_u[i] * (0.1f / 3.0f) : 0.1f * (float)sin(ph) + noise_v[i] * 0.003f;
  }
}

// ------------------------------------------------------------------------------ softmax cross-entropy (mean over the batch)
// F.cross_entropy(logits [B][K], labels) of the latent classifier (train.py:302, :422). One wave per sample; the forward
// keeps the softmax probabilities for the backward.
__global__ __launch_bounds__(64) void cross_entropy_fwd_kernel(const float* logits, const int64_t* labels, int K, float w_over_b,
                                                               float* loss, float* prob) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* z = logits + (long)b * K;
  float mx = -1e30f;
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, z[k]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float se = 0.f;
  for (int k = lane; k < K; k += 64) se += expf(z[k] - mx);
  se = wave_sum(se);
  for (int k = lane; k < K; k += 64) prob[(long)b * K + k] = expf(z[k] - mx) / se;
  if (lane == 0) atomicAdd(loss, (logf(se) + mx - z[labels[b]]) * w_over_b);
}
__global__ __launch_bounds__(256) void cross_entropy_bwd_kernel(const float* prob, const int64_t* labels, int B, int K, float w_over_b,
                                                                const float* up, float* dlogits) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * K) return;
  const int b = i / K, k = i - b * K;
  dlogits[i] = (prob[i] - (labels[b] == k ? 1.f : 0.f)) * w_over_b * up[0];
}

}  // namespace

// ================================================================================================ C ABI
extern "C" int tdvc_weight_norm_fwd(const float* params, float* w, const int64_t* row_voff, const int64_t* row_goff,
                                    const int64_t* row_woff, const int32_t* row_len, int nrows, void* stream) {
  if (!params || !w || nrows <= 0) return tdvc_fail(TDVC_EINVAL, "weight_norm_fwd: bad arguments");
  hipLaunchKernelGGL(wn_fwd_kernel, dim3((nrows + 3) / 4), dim3(256), 0, (hipStream_t)stream, params, w, row_voff, row_goff, row_woff, row_len, nrows,
                     (float*)nullptr, (const int64_t*)nullptr, (const int64_t*)nullptr, (const int32_t*)nullptr);
  TDVC_CHECK_LAUNCH();
  return TDVC_OK;
}
extern "C" int tdvc_weight_norm_fwd_t(const float* params, float* w, float* wt, const int64_t* row_voff, const int64_t* row_goff,
                                      const int64_t* row_woff, const int32_t* row_len, const int64_t* row_tbase,
                                      const int64_t* row_tstride, const int32_t* row_k, int nrows, void* stream) {
  if (!params || !w || !wt || nrows <= 0) return tdvc_fail(TDVC_EINVAL, "weight_norm_fwd_t: bad arguments");
  hipLaunchKernelGGL(wn_fwd_kernel, dim3((nrows + 3) / 4), dim3(256), 0, (hipStream_t)stream, params, w, row_voff, row_goff, row_woff, row_len, nrows,
                     wt, row_tbase, row_tstride, row_k);
  TDVC_CHECK_LAUNCH();
  return TDVC_OK;
}
extern "C" int tdvc_weight_norm_bwd(const float* params, const float* dw, float* grads, const int64_t* row_voff,
                                    const int64_t* row_goff, const int64_t* row_woff, const int32_t* row_len, int nrows,
                                    int accumulate, void* stream) {
  if (!params || !dw || !grads || nrows <= 0) return tdvc_fail(TDVC_EINVAL, "weight_norm_bwd: bad arguments");
  hipLaunchKernelGGL(wn_bwd_kernel, dim3((nrows + 3) / 4), dim3(256), 0, (hipStream_t)stream, params, dw, grads, row_voff, row_goff, row_woff, row_len, nrows, accumulate);
  TDVC_CHECK_LAUNCH();
  return TDVC_OK;
}

extern "C" int tdvc_adamw(float* p, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int step, const int32_t* step_dev, float grad_scale, void* stream) {
  if (!p || !grad || !m || !v || n <= 0 || (step < 1 && !step_dev)) return tdvc_fail(TDVC_EINVAL, "adamw: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(tdvc_grid(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, grad, m, v, (long)n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2s, grad_scale == 0.f ? 1.f : grad_scale, step_dev, (const float*)nullptr);
  TDVC_CHECK_LAUNCH();
  return TDVC_OK;
}
extern "C" int tdvc_adamw_clipped(float* p, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, int step, const int32_t* step_dev, float grad_scale,
                                  const float* clip_coef_dev, void* stream) {
  if (!p || !grad || !m || !v || n <= 0 || (step < 1 && !step_dev)) return tdvc_fail(TDVC_EINVAL, "adamw_clipped: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(tdvc_grid(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, grad, m, v, (long)n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2s, grad_scale == 0.f ? 1.f : grad_scale, step_dev, clip_coef_dev);
  TDVC_CHECK_LAUNCH();
  return TDVC_OK;
}
extern "C" int tdvc_grad_clip_coef(const float* grad, int64_t n, float max_norm, float grad_scale, float* workspace, float* out, void* stream) {
  if (!grad || !workspace || !out || n <= 0 || !(max_norm > 0.f)) return tdvc_fail(TDVC_EINVAL, "grad_clip_coef: bad arguments");
  const int nparts = tdvc_grid(n, 256, 1024);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, grad, (long)n, workspace);
  TDVC_CHECK_LAUNCH();
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, nparts, max_norm, grad_scale == 0.f ? 1.f : grad_scale, out);
  TDVC_CHECK_LAUNCH();
  return TDVC_OK;
}
extern "C" int tdvc_roll_batches(const float* x, const int64_t* shift, float* y, int B, int C, int T, void* stream) {
  if (!x || !shift || !y || B <= 0 || C <= 0 || T <= 0) return tdvc_fail(TDVC_EINVAL, "roll_batches: bad arguments");
  hipLaunchKernelGGL(roll_batches_kernel, dim3(tdvc_grid(T, 256, 64), C, B), dim3(256), 0, (hipStream_t)stream, x, shift, y, C, T);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}

__global__ void inc_i32_kernel(int32_t* p, int32_t by) { if (threadIdx.x == 0) p[0] += by; }
extern "C" int tdvc_inc_i32(int32_t* p, int32_t by, void* stream) {
  hipLaunchKernelGGL(inc_i32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p, by);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}

extern "C" int tdvc_l2norm_fwd(const float* x, float* y, float* inv_norm, int B, int C, int T, float eps, void* stream) {
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((T + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, y, inv_norm, C, T, eps);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_l2norm_bwd(const float* y, const float* inv_norm, const float* dy, float* dx, int B, int C, int T, void* stream) {
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((T + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, y, inv_norm, dy, dx, C, T);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_gather_ch_fwd(const float* x, const int64_t* label, float* y, int B, int C, int T, void* stream) {
  hipLaunchKernelGGL(gather_ch_kernel, dim3((T + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, label, y, C, T, 0);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_gather_ch_bwd(const float* dy, const int64_t* label, float* dx, int B, int C, int T, void* stream) {
  hipLaunchKernelGGL(gather_ch_kernel, dim3((T + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, dy, label, dx, C, T, 1);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_concat_cond(const float* emb, const float* exc, float* c, int B, int Ce, int Cx, int T, void* stream) {
  hipLaunchKernelGGL(concat_cond_kernel, dim3((T + 255) / 256, Ce + Cx, B), dim3(256), 0, (hipStream_t)stream, emb, exc, c, Ce, Cx, T);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_concat_cond_bwd(const float* dc, float* demb, float* dexc, int B, int Ce, int Cx, int T, int accumulate_emb, void* stream) {
  hipLaunchKernelGGL(concat_cond_bwd_kernel, dim3(Ce + Cx, B), dim3(256), 0, (hipStream_t)stream, dc, demb, dexc, Ce, Cx, T, accumulate_emb);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_edge_sum3(const float* d, float* out, int B, int C, int T, void* stream) {
  if (T < 3) return tdvc_fail(TDVC_EINVAL, "edge_sum3: T must be >= 3");
  hipLaunchKernelGGL(edge_sum3_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, d, out, C, T);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
// y = sum of up to 16 tensors in one pass (reads n, writes 1): the gradient of an activation that fans out to several
// consumers, instead of autograd's chain of n - 1 two-operand adds (each reads 2, writes 1)
namespace { struct SumSrc { const float* p[16]; }; }
__global__ __launch_bounds__(256) void sum_n_kernel(const SumSrc s, int nsrc, float* y, long n, int vec) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  if (vec) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      f4 a = reinterpret_cast<const f4*>(s.p[0])[i];
      for (int k = 1; k < nsrc; ++k) a += reinterpret_cast<const f4*>(s.p[k])[i];
      reinterpret_cast<f4*>(y)[i] = a;
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
      float a = s.p[0][i];
      for (int k = 1; k < nsrc; ++k) a += s.p[k][i];
      y[i] = a;
    }
  }
}
extern "C" int tdvc_sum_n(const float* const* srcs, int nsrc, float* y, int64_t n, void* stream) {
  if (!srcs || !y || nsrc < 1 || nsrc > 16) return tdvc_fail(TDVC_EINVAL, "sum_n: 1..16 sources");
  if (n <= 0) return TDVC_OK;
  SumSrc s;
  bool vec = (n & 3) == 0 && (((uintptr_t)y) & 15) == 0;
  for (int k = 0; k < 16; ++k) {
    s.p[k] = k < nsrc ? srcs[k] : nullptr;
    if (k < nsrc && (!srcs[k] || (((uintptr_t)srcs[k]) & 15))) { if (!srcs[k]) return tdvc_fail(TDVC_EINVAL, "sum_n: null source"); vec = false; }
  }
  hipLaunchKernelGGL(sum_n_kernel, dim3(tdvc_grid(vec ? n / 4 : n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, s, nsrc, y, (long)n, vec ? 1 : 0);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_axpby(const float* a, const float* b, float* y, float alpha, float beta, int64_t n, void* stream) {
  if (n <= 0) return TDVC_OK;
  hipLaunchKernelGGL(axpby_kernel, dim3(tdvc_grid(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, a, b, y, alpha, beta, (long)n);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_gate_fwd(const float* xin, int64_t xin_bs, const float* g, int64_t g_bs, float* acts, int64_t acts_bs, int B, int H, int T,
                             void* stream) {
  if (!xin || !acts || B <= 0 || H <= 0 || T <= 0) return tdvc_fail(TDVC_EINVAL, "gate_fwd: bad argument");
  const long n = (long)B * H * T;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(tdvc_grid(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, xin, (long)xin_bs, g, (long)g_bs, acts,
                     (long)acts_bs, H, T, n);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_gate_bwd(const float* xin, int64_t xin_bs, const float* g, int64_t g_bs, const float* dacts, int64_t dacts_bs, float* dxin,
                             int64_t dxin_bs, int B, int H, int T, void* stream) {
  if (!xin || !dacts || !dxin || B <= 0 || H <= 0 || T <= 0) return tdvc_fail(TDVC_EINVAL, "gate_bwd: bad argument");
  const long n = (long)B * H * T;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(tdvc_grid(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, xin, (long)xin_bs, g, (long)g_bs, dacts,
                     (long)dacts_bs, dxin, (long)dxin_bs, H, T, n);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_fill(float* y, float value, int64_t n, void* stream) {
  if (n <= 0) return TDVC_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(tdvc_grid(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, y, value, (long)n);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}

extern "C" int tdvc_cin_fwd(const float* x, const float* gb, float* y, float* mean, float* rstd, int B, int C, int T, int Tg, float eps, void* stream) {
  if (Tg != 1 && Tg != T) return tdvc_fail(TDVC_EINVAL, "cin_fwd: Tg must be 1 or T");
  const bool al = (T & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)gb)) & 15) == 0;
  if (al && T <= 4096) hipLaunchKernelGGL(cin_fwd_row_kernel<4>, dim3(C, B), dim3(256), 0, (hipStream_t)stream, x, gb, y, mean, rstd, C, T, Tg, eps);
  else if (al && T <= 16384) hipLaunchKernelGGL(cin_fwd_row_kernel<16>, dim3(C, B), dim3(256), 0, (hipStream_t)stream, x, gb, y, mean, rstd, C, T, Tg, eps);
  else hipLaunchKernelGGL(cin_fwd_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, x, gb, y, mean, rstd, C, T, Tg, eps);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_cin_bwd(const float* x, const float* gb, const float* dy, const float* mean, const float* rstd,
                            float* dx, float* dgb, int B, int C, int T, int Tg, void* stream) {
  if (Tg != 1 && Tg != T) return tdvc_fail(TDVC_EINVAL, "cin_bwd: Tg must be 1 or T");
  const bool al = (T & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)gb) | ((uintptr_t)dgb)) & 15) == 0;
  if (al && T <= 4096) hipLaunchKernelGGL(cin_bwd_row_kernel<4>, dim3(C, B), dim3(256), 0, (hipStream_t)stream, x, gb, dy, mean, rstd, dx, dgb, C, T, Tg);
  else if (al && T <= 8192) hipLaunchKernelGGL(cin_bwd_row_kernel<8>, dim3(C, B), dim3(256), 0, (hipStream_t)stream, x, gb, dy, mean, rstd, dx, dgb, C, T, Tg);
  else hipLaunchKernelGGL(cin_bwd_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, x, gb, dy, mean, rstd, dx, dgb, C, T, Tg);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}

extern "C" int tdvc_mse_const_fwd(const float* x, int64_t n, float target, float weight, float* loss_out, void* stream) {
  hipLaunchKernelGGL(mse_const_fwd_kernel, dim3(tdvc_grid(n, 256, 256)), dim3(256), 0, (hipStream_t)stream, x, (long)n, target, weight / (float)n, loss_out);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_mse_const_bwd(const float* x, int64_t n, float target, float weight, const float* upstream, float* dx, void* stream) {
  hipLaunchKernelGGL(mse_const_bwd_kernel, dim3(tdvc_grid(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, x, (long)n, target, 2.f * weight / (float)n, upstream, dx);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_l1_fwd(const float* a, const float* b, int64_t n, float weight, float* loss_out, void* stream) {
  hipLaunchKernelGGL(l1_fwd_kernel, dim3(tdvc_grid(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, a, b, (long)n, weight / (float)n, loss_out);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_l1_bwd(const float* a, const float* b, int64_t n, float weight, const float* upstream, float* da, int accumulate, void* stream) {
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(tdvc_grid(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a, b, (long)n, weight / (float)n, upstream, da, accumulate);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}

// pairs: HOST array, consumed before the call returns. fwd: loss_out += sum_i weight_i / n_i * sum|a_i - b_i| (b_i != NULL);
// bwd: da_i = sign(a_i - b_i) * weight_i / n_i * upstream, or da_i = 0 where b_i == NULL.
static int l1_multi(const tdvc_l1_pair* pairs, int npairs, bool bwd, float* loss_out, const float* upstream, hipStream_t st) {
  if (!pairs || npairs < 0) return tdvc_fail(TDVC_EINVAL, "l1_multi: bad argument");
  for (int i = 0; i < npairs;) {                       // batches of up to L1M_MAX live entries
    L1Batch q = {};
    int chunks = 0;
    for (; i < npairs && q.np < L1M_MAX; ++i) {
      const tdvc_l1_pair& e = pairs[i];
      if (e.n <= 0) continue;
      if (bwd ? !e.da || (e.b && !e.a) : (!e.a || !e.b)) {
        if (!bwd && !e.b) continue;                  // a zero-fill entry has nothing to add to the loss
        return tdvc_fail(TDVC_EINVAL, "l1_multi: null pointer");
      }
      q.a[q.np] = e.a; q.b[q.np] = e.b; q.da[q.np] = e.da; q.n[q.np] = e.n; q.w[q.np] = e.weight / (float)e.n;
      q.first[q.np] = chunks;
      chunks += (int)((e.n + L1M_CHUNK - 1) / L1M_CHUNK);
      ++q.np;
    }
    q.first[q.np] = chunks;
    if (!chunks) continue;
    if (bwd) hipLaunchKernelGGL(l1_multi_bwd_kernel, dim3(chunks), dim3(256), 0, st, q, upstream);
    else hipLaunchKernelGGL(l1_multi_fwd_kernel, dim3(chunks), dim3(256), 0, st, q, loss_out);
    TDVC_CHECK_LAUNCH();
  }
  return TDVC_OK;
}
extern "C" int tdvc_l1_multi_fwd(const tdvc_l1_pair* pairs, int npairs, float* loss_out, void* stream) {
  if (!loss_out) return tdvc_fail(TDVC_EINVAL, "l1_multi_fwd: null output");
  return l1_multi(pairs, npairs, false, loss_out, nullptr, (hipStream_t)stream);
}
extern "C" int tdvc_l1_multi_bwd(const tdvc_l1_pair* pairs, int npairs, const float* upstream, void* stream) {
  return l1_multi(pairs, npairs, true, nullptr, upstream, (hipStream_t)stream);
}

extern "C" int tdvc_reflect_pad_fwd(const float* x, float* y, int B, int T, int pad, void* stream) {
  if (pad >= T) return tdvc_fail(TDVC_EINVAL, "reflect_pad: pad must be < T");
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(tdvc_grid(T + 2 * pad, 256, 256), B), dim3(256), 0, (hipStream_t)stream, x, y, T, pad);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_reflect_pad_bwd(const float* dy, float* dx, int B, int T, int pad, void* stream) {
  if (pad >= T) return tdvc_fail(TDVC_EINVAL, "reflect_pad: pad must be < T");
  hipLaunchKernelGGL(reflect_pad_bwd_kernel, dim3(tdvc_grid(T, 256, 256), B), dim3(256), 0, (hipStream_t)stream, dy, dx, T, pad);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_power_fwd(const float* spec, float* power, int B, int F, int N, void* stream) {
  const long total = (long)B * F * N;
  hipLaunchKernelGGL(power_fwd_kernel, dim3(tdvc_grid(total, 256, 2048)), dim3(256), 0, (hipStream_t)stream, spec, power, F, N, total);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_power_bwd(const float* spec, const float* dpower, float* dspec, int B, int F, int N, void* stream) {
  const long total = (long)B * F * N;
  hipLaunchKernelGGL(power_bwd_kernel, dim3(tdvc_grid(total, 256, 2048)), dim3(256), 0, (hipStream_t)stream, spec, dpower, dspec, F, N, total);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_log_l1_fwd(const float* a, const float* b, int64_t n, float floor_, float weight, float* loss_out, void* stream) {
  hipLaunchKernelGGL(log_l1_fwd_kernel, dim3(tdvc_grid(n, 256, 256)), dim3(256), 0, (hipStream_t)stream, a, b, (long)n, floor_, weight / (float)n, loss_out);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_log_l1_bwd(const float* a, const float* b, int64_t n, float floor_, float weight, const float* upstream, float* da, void* stream) {
  hipLaunchKernelGGL(log_l1_bwd_kernel, dim3(tdvc_grid(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, a, b, (long)n, floor_, weight / (float)n, upstream, da);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}

extern "C" int tdvc_contrastive_fwd_bwd(const float* X, const float* Y, const int32_t* idx_x, const int32_t* idx_y, int B, int C, int T, int N,
                                        float weight, float* loss_out, float* dX, float* dY, void* stream) {
  if (C <= 0 || T <= 1 || N <= 0) return tdvc_fail(TDVC_EINVAL, "contrastive: bad shape");
  const size_t lds = (size_t)(C * T + T + 2 * (N + 1) + 2 * C + 8) * sizeof(float);
  if (lds > 150 * 1024) return tdvc_fail(TDVC_EUNSUPPORTED, "contrastive: embedding tile exceeds LDS");
  TDVC_BIG_LDS_ONCE(contrastive_kernel);
  const float coef = weight / (2.f * (float)B * (float)T);
  const int tsplit = T < 16 ? T : 16;            // 2 * B * 16 blocks instead of 2 * B serial walks over T
  hipLaunchKernelGGL(contrastive_kernel, dim3(2, B, tsplit), dim3(128), lds, (hipStream_t)stream, X, Y, idx_x, idx_y, C, T, N, coef, loss_out, dX, dY);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}

extern "C" int tdvc_f0_to_excitation(const float* f0, const float* noise_v, const float* noise_u, const float* start_phase, float* exc,
                                     int B, int n_frames, int step, float sampling_rate, int linear, void* stream) {
  if (!f0 || !noise_v || !noise_u || !start_phase || !exc) return tdvc_fail(TDVC_EINVAL, "f0_to_excitation: null pointer");
  if (B <= 0 || n_frames < 2 || step <= 0 || sampling_rate <= 0.f) return tdvc_fail(TDVC_EINVAL, "f0_to_excitation: bad shape");
  hipLaunchKernelGGL(f0_excitation_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, f0, noise_v, noise_u, start_phase, exc, n_frames, step,
                     sampling_rate, linear);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}

extern "C" int tdvc_cross_entropy_fwd(const float* logits, const int64_t* labels, int B, int K, float weight, float* loss_out, float* prob,
                                      void* stream) {
  if (!logits || !labels || !loss_out || !prob || B <= 0 || K <= 0) return tdvc_fail(TDVC_EINVAL, "cross_entropy_fwd: bad arguments");
  hipLaunchKernelGGL(cross_entropy_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, labels, K, weight / (float)B, loss_out, prob);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
extern "C" int tdvc_cross_entropy_bwd(const float* prob, const int64_t* labels, int B, int K, float weight, const float* upstream,
                                      float* dlogits, void* stream) {
  if (!prob || !labels || !upstream || !dlogits || B <= 0 || K <= 0) return tdvc_fail(TDVC_EINVAL, "cross_entropy_bwd: bad arguments");
  hipLaunchKernelGGL(cross_entropy_bwd_kernel, dim3((B * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, prob, labels, B, K,
                     weight / (float)B, upstream, dlogits);
  TDVC_CHECK_LAUNCH(); return TDVC_OK;
}
