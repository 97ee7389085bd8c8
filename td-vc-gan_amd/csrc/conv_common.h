// Internal (device + host) definitions shared by the conv kernels. Not part of the C ABI.
//
// Every Conv1d / ConvTranspose1d forward, input-grad and weight-grad on the path is mapped by the
// host onto ONE "reduced" stride-1 problem per (group, sample):
//     Y'[r][n] = sum_{c < Cred} sum_{j < J}  A[r][c][j] * X'[c][n + j*d + lo]
// with three staging modes that define X' (and how rows/cols map back to tensors):
//   DIRECT : X'[c][q]        = xf(x[c][q])                 (stride-1 conv, also its dgrad with flipped taps)
//   DOWN   : X'[(c,phi)][t]  = xf(x[c][t*s + phi - pad])   (strided conv as a time-to-depth + J=ceil(K/s) tap conv)
//   UP     : rows r=(m,phi); out position u = n*s + phi - pad (transposed conv as J-tap conv + depth-to-time)
// so dilated, strided, grouped and transposed convolutions share one MFMA main loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

// Host-side launch helpers. TDVC_BIG_LDS_ONCE: raise the dynamic-LDS cap of one kernel instantiation exactly once
// (thread-safe: the C ABI is re-entrant). TDVC_TRACE: test-only record of which kernel instantiation a launch used
// (tdvc_debug_trace in include/tdvc.h; a single predictable branch when tracing is off).
#define TDVC_BIG_LDS_ONCE(k)                                                                                         \
  do {                                                                                                               \
    static std::once_flag f_;                                                                                        \
    std::call_once(f_, [&] { hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
  } while (0)
namespace tdvc {
extern int g_trace_on;
extern int g_force_tile;
extern int g_force_generic;      // conv_api.hip: tests route every conv to the scalar kernels
extern int g_lds_cap;
extern int g_knob[8];   // tuning knobs (tdvc_debug_knob): [0] XCD-aware block order of the lean conv kernel (default off: measured null, profiles/r02_e_xcd_remap_ab.txt)
void trace_kernel(const void* fn);
}
#define TDVC_TRACE(k) do { if (tdvc::g_trace_on) tdvc::trace_kernel(reinterpret_cast<const void*>(k)); } while (0)

namespace tdvc {

enum { MODE_DIRECT = 0, MODE_DOWN = 1, MODE_UP = 2 };
enum { XF_NONE = 0, XF_LRELU = 1, XF_FILM_LRELU = 2, XF_MASK_LRELU = 3, XF_MASK_TANH = 4 };
enum { EPI_FWD = 0, EPI_MASK = 1, EPI_FILM = 2, EPI_PLAIN = 3 };
enum { POST_NONE = 0, POST_LRELU = 1, POST_TANH = 2 };

struct Xf {
  int kind; float slope; float scale;
  const float* aux; long aux_bs;
};

// Tensor operand [B][groups*Cg][T], batch stride bs (elements), channel stride T.
struct Opnd {
  const float* p; long bs; int T; int Cg; Xf xf;
};

struct GemmConvP {
  Opnd x;                       // B-matrix source
  const float* w; long w_sg, w_sm, w_sc; int K; int tap_flip;   // A[r][c][k] = w[g*w_sg + m*w_sm + c*w_sc + k]
  int mode, s, d, pad, reflect, J;
  int R, Cred, Cc, N, groups, lo, span, XS, WS;
  int i0;                       // staged index read by column n0, tap 0 (= first tap position - lo)
  int mirror_pad;               // > 0: fold reflect-pad halo (dgrad of a reflect conv)
  int stage_rows;               // DOWN with a large stride: stage with lanes along the reduced rows
  int chan_stage;               // DOWN, chunk = whole channels (Cc % s == 0): coalesced time-to-depth staging per channel
  int w_nat;                    // weight rows are contiguous over (c,k) and float4-alignable (host-checked)
  int w_packed;                 // p.w is the reduced-layout copy [g][R][Cred4][J] made by weight_repack_kernel: rows copy as float4
  float* y; long y_bs; int Ty; int Cy_g;
  int epi;
  const float* bias;
  const float* bias3;           // [B][Ctot][3] edge/interior/edge bias
  const float* res; long res_bs;
  int post; float post_slope; float out_scale;
  const float* add; long add_bs; float add_scale;
  const float* mx; long mx_bs; float m_slope;
  const float* gb; long gb_bs; float* dgb; long dgb_bs;
};

struct WgradP {
  Opnd a;                       // rows operand (dy-like), N columns
  Opnd x;                       // column operand (x-like)
  int mode, s, d, pad, reflect, J, K;
  int R, Cred, N, groups, lo, span, NTc, XS, AS;
  int i0;
  int ntiles;                   // time chunks per sample
  float* slab; long slab_stride; // slab[(b*ntiles+tile)][groups*R*Cx_g*K (+ bias rows)]
  long w_sg, w_sm, w_sc;        // slab element index = g*w_sg + m*w_sm + c*w_sc + k
};

__device__ __forceinline__ float lrelu_f(float v, float s) { return v > 0.f ? v : v * s; }

// Value of the transformed operand element (b, ch, t) given its raw value v; ch is the absolute channel.
__device__ __forceinline__ float apply_xf(const Xf& xf, float v, int b, int ch, int t, int T, int Ctot) {
  switch (xf.kind) {
    case XF_LRELU: v = lrelu_f(v, xf.slope); break;
    case XF_FILM_LRELU: {
      const float* g = xf.aux + (long)b * xf.aux_bs + (long)ch * T + t;
      v = lrelu_f(v * (1.f + g[0]) + g[(long)Ctot * T], xf.slope);
    } break;
    case XF_MASK_LRELU: {
      float a = xf.aux[(long)b * xf.aux_bs + (long)ch * T + t];
      v = a > 0.f ? v : v * xf.slope;
    } break;
    case XF_MASK_TANH: {
      float a = xf.aux[(long)b * xf.aux_bs + (long)ch * T + t];
      v = v * (1.f - a * a);
    } break;
    default: break;
  }
  return v * xf.scale;
}

// Fetch transformed operand at absolute channel ch, time q with zero / reflect padding.
__device__ __forceinline__ float fetch_opnd(const Opnd& o, int b, int ch, int q, int reflect, int Ctot) {
  if (reflect) {
    if (q < 0) q = -q;
    else if (q >= o.T) q = 2 * (o.T - 1) - q;
  }
  if (q < 0 || q >= o.T) return 0.f;
  float v = o.p[(long)b * o.bs + (long)ch * o.T + q];
  return apply_xf(o.xf, v, b, ch, q, o.T, Ctot);
}


typedef float f32x4_t __attribute__((ext_vector_type(4)));

// True when rows [q0, q0+span) of operand `o` can be staged with aligned 16-byte loads and no padding logic.
__device__ __forceinline__ bool rows_align_ok(const Opnd& o, int q0, int span) {
  return (o.T & 3) == 0 && (o.bs & 3) == 0 && ((q0 | span) & 3) == 0 && (((uintptr_t)o.p) & 15) == 0 &&
         (o.xf.kind == XF_NONE || o.xf.kind == XF_LRELU || ((((uintptr_t)o.xf.aux) & 15) == 0 && (o.xf.aux_bs & 3) == 0));
}
__device__ __forceinline__ bool rows_fast_ok(const Opnd& o, int q0, int span) {
  return q0 >= 0 && q0 + span <= o.T && rows_align_ok(o, q0, span);
}

// Register-staged row tiles, split into ISSUE (all global loads of a batch in flight at once) and COMMIT
// (prologue transform + LDS write), so that HBM latency overlaps whatever runs in between (the MFMA loop
// on the previous channel chunk) — cdna_hip_programming.md T14. Tile = nrows x span floats; element
// e = ebase + tid + i*256 -> (row r, float4 v). Rows >= nvalid are zero-filled.
template <int NV>
struct RegTile { f32x4_t v[NV]; };

template <int NV>
__device__ __forceinline__ void tile_issue(RegTile<NV>& t, const float* base, int rowstride, int nvalid, int nrows, int span,
                                           int cvalid /* valid floats per row, multiple of 4 */, int ebase, int tid,
                                           int vlo = 0, int vhi = 1 << 30 /* valid float4 columns [vlo, vhi): zero padding at sequence ends */);
// kind NONE / LRELU use only `t`; MASK_* use `a` (activation output); FILM uses a = gamma, c = beta.
template <int NV>
__device__ __forceinline__ void tile_commit(const RegTile<NV>& t, const RegTile<NV>* a, const RegTile<NV>* c, const Xf& xf,
                                            float* dst, int XS, int nvalid, int nrows, int span, int ebase, int tid) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const int nvec = span >> 2, total = nrows * nvec;
  const float inv = 1.0f / (float)nvec;
  const int kind = xf.kind;
  const float sl = xf.slope, sc = xf.scale;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = ebase + tid + i * 256;
    if (e >= total) continue;
    const int r = (int)(((float)e + 0.5f) * inv);
    const int vv = e - r * nvec;
    f32x4_t val = t.v[i];
    if (r < nvalid) {
      if (kind == XF_LRELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) val[q] = fmaxf(val[q], val[q] * sl);
      } else if (kind == XF_FILM_LRELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float h = val[q] * (1.f + a->v[i][q]) + c->v[i][q]; val[q] = fmaxf(h, h * sl); }
      } else if (kind == XF_MASK_LRELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) val[q] = a->v[i][q] > 0.f ? val[q] : val[q] * sl;
      } else if (kind == XF_MASK_TANH) {
#pragma unroll
        for (int q = 0; q < 4; ++q) val[q] = val[q] * (1.f - a->v[i][q] * a->v[i][q]);
      }
      if (sc != 1.f) {
#pragma unroll
        for (int q = 0; q < 4; ++q) val[q] *= sc;
      }
    }
    f32x2_t* d2 = reinterpret_cast<f32x2_t*>(dst + r * XS + 4 * vv);   // rows are 8-byte aligned in every layout
    d2[0] = (f32x2_t){val[0], val[1]};
    d2[1] = (f32x2_t){val[2], val[3]};
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Row-walk staging. A tile of `nrows` rows x `nvec` float4 columns (rows `rowstride` floats apart in HBM, `ls` floats
// apart in LDS) is covered in passes of rp = 256 / nvec whole rows: thread (rsub, vv) touches row i*rp + rsub, column
// vv in pass i, so consecutive passes differ by a wave-uniform stride and the per-element cost is one buffer load,
// one add and one LDS store. Loads go through a raw buffer descriptor whose range check returns zero past the end of
// the tensor (cdna_hip_programming.md T8); threads beyond rp*nvec carry an out-of-range offset.
typedef __amdgpu_buffer_rsrc_t srd_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ srd_t make_srd(const float* base, int bytes) {   // wave-uniform inputs only
  // readfirstlane on the descriptor inputs makes the uniformity provable: without it a base that went through the
  // vector ALU (integer division, 64-bit multiply) puts every load behind a waterfall loop (cdna_hip_programming.md T20)
  const uintptr_t a = reinterpret_cast<uintptr_t>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(a & 0xffffffffu));
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int nb = __builtin_amdgcn_readfirstlane(bytes > 0 ? bytes : 0);
  float* ub = reinterpret_cast<float*>(((uintptr_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(ub, 0, nb, 0x00020000);
}
__device__ __forceinline__ f32x4_t buf_load4(srd_t rs, int voff) {
  return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
}

// Raw buffer loads with an out-of-range offset for the elements that must read as zero: nothing selects on the loaded
// value, so no s_waitcnt lands between the issue and the code that runs under the loads (a `cond ? load : 0` form made
// the compiler wait for every load right where it was issued).
template <int NV>
__device__ __forceinline__ void tile_issue(RegTile<NV>& t, const float* base, int rowstride, int nvalid, int nrows, int span,
                                           int cvalid, int ebase, int tid, int vlo, int vhi) {
  const int nvec = span >> 2, total = nrows * nvec;
  const float inv = 1.0f / (float)nvec;
  const srd_t rs = make_srd(base, nvalid > 0 ? ((nvalid - 1) * rowstride + cvalid) * 4 : 0);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = ebase + tid + i * 256;
    const int r = (int)(((float)e + 0.5f) * inv);
    const int vv = e - r * nvec;
    t.v[i] = buf_load4(rs, (e < total && r < nvalid && 4 * vv < cvalid && vv >= vlo && vv < vhi) ? (r * rowstride + 4 * vv) * 4 : 0x7f000000);
  }
}

struct RowWalk { int voff, loff, gstep, lstep, nk; bool active; };   // nk: valid components of the thread's float4 (rows with T % 4 != 0)

// The pass-count guards are loop invariant; hidden from LICM they stay one s_cmp + branch each instead of a table
// of precomputed lane masks that spills out of the SGPR file.
__device__ __forceinline__ int walk_opaque(int n) { asm volatile("" : "+s"(n)); return n; }

// q0 / T: the tile's first column sits at position q0 of rows T long; columns outside [0, T) load as zero.
__device__ __forceinline__ RowWalk make_walk(int tid, int nvec, int rp, int rowstride, int ls, int q0, int T) {
  RowWalk w;
  const int rsub = (int)(((float)tid + 0.5f) * (1.0f / (float)nvec));
  const int vv = tid - rsub * nvec;
  const int q = q0 + 4 * vv;
  w.active = rsub < rp;
  w.voff = (w.active && q >= 0 && q < T) ? (rsub * rowstride + q) * 4 : 0x7f000000;
  w.nk = T - q < 4 ? T - q : 4;
  w.loff = rsub * ls + 4 * vv;
  w.gstep = rp * rowstride * 4;
  w.lstep = rp * ls;
  return w;
}

template <int NP>
__device__ __forceinline__ void walk_issue(RegTile<NP>& t, srd_t rs, const RowWalk& w, int np) {
  int vo = w.voff;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (i < walk_opaque(np)) { t.v[i] = buf_load4(rs, vo); vo += w.gstep; }
  }
}

// Input tile: LeakyReLU (slope 1 = identity) and scale on the way into LDS; rows are 16-byte aligned (ls % 4 == 0).
// The activation is applied in place so that every LDS store has its own source registers (no store-to-store waits).
template <int NP>
__device__ __forceinline__ void walk_commit_act(RegTile<NP>& t, const RowWalk& w, int np, float* lds, float slope, float scale, bool tail = false) {
  if (!w.active) return;
  const int base = w.loff;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (i < walk_opaque(np)) {
      f32x4_t o;
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float v = t.v[i][q]; o[q] = fmaxf(v, v * slope); }
      if (scale != 1.f) o *= scale;
      if (tail) {                                        // rows with T % 4 != 0: the last float4 of a row runs into the next row
#pragma unroll
        for (int q = 1; q < 4; ++q) o[q] = q < w.nk ? o[q] : 0.f;
      }
      *reinterpret_cast<f32x4_t*>(lds + base + i * w.lstep) = o;
      __builtin_amdgcn_sched_barrier(0);                 // one element at a time: keeps the temporaries to one float4
    }
  }
}

// Weight tile: rows are only 8-byte aligned in LDS (row stride K*Cc + 2 keeps the fragment reads conflict-free).
template <int NP>
__device__ __forceinline__ void walk_commit_w(const RegTile<NP>& t, const RowWalk& w, int np, float* lds) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  if (!w.active) return;
  const int base = w.loff;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (i < walk_opaque(np)) {
      f32x2_t* d2 = reinterpret_cast<f32x2_t*>(lds + base + i * w.lstep);
      d2[0] = (f32x2_t){t.v[i][0], t.v[i][1]};
      d2[1] = (f32x2_t){t.v[i][2], t.v[i][3]};
    }
  }
}

// Whole tile, not pipelined: batches of NV float4 per thread (x up to 3 source tensors) in flight at a time.
template <int NV>
__device__ __forceinline__ void stage_rows_batched(const Opnd& o, float* dst, int XS, int b, int ch0, int nvalid, int nrows,
                                                   int q0, int span, int Ctot, int tid, int vlo = 0, int vhi = 1 << 30) {
  const int total = nrows * (span >> 2);
  const float* base = o.p + (long)b * o.bs + (long)ch0 * o.T + q0;
  const float* abase = o.xf.aux ? o.xf.aux + (long)b * o.xf.aux_bs + (long)ch0 * o.T + q0 : nullptr;
  for (int eb = 0; eb < total; eb += NV * 256) {
    RegTile<NV> t, a, c;
    tile_issue<NV>(t, base, o.T, nvalid, nrows, span, span, eb, tid, vlo, vhi);
    if (o.xf.kind >= XF_FILM_LRELU) tile_issue<NV>(a, abase, o.T, nvalid, nrows, span, span, eb, tid, vlo, vhi);
    if (o.xf.kind == XF_FILM_LRELU) tile_issue<NV>(c, abase + (long)Ctot * o.T, o.T, nvalid, nrows, span, span, eb, tid, vlo, vhi);
    tile_commit<NV>(t, &a, &c, o.xf, dst, XS, nvalid, nrows, span, eb, tid);
  }
}


// MODE_DOWN staging, general form (any position, prologue NONE / LeakyReLU): the s phase rows (c, 0..s-1) of one channel are
// the SAME contiguous run of span*s samples x[c][qbase ...], read as float4 by consecutive lanes and de-interleaved on the
// way into LDS (sample e of the run -> phase e % s, column e / s). NB float4 per thread are in flight before the first LDS
// store. xs rows: [nch * s][XS]; channels >= nchv and samples outside [0, T) are written as zero.
template <int NB>
__device__ __forceinline__ void stage_down_runs(const Opnd& o, float* xs, int XS, int b, int chan0, int nch, int nchv, int s, int qbase,
                                                int span, int Ctot, int tid) {
  const int nvr = (span * s) >> 2, tot = nch * nvr;
  const float inv_nvr = 1.0f / (float)nvr, inv_s = 1.0f / (float)s;
  const srd_t rs = make_srd(o.p + (long)b * o.bs, Ctot * o.T * 4);
  const bool act = o.xf.kind == XF_LRELU;
  const float sl = o.xf.slope, sc = o.xf.scale;
  for (int eb = 0; eb < tot; eb += NB * 256) {
    f32x4_t v[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = eb + tid + i * 256;
      const int ch = (int)(((float)e + 0.5f) * inv_nvr);
      const int q = qbase + 4 * (e - ch * nvr);
      const bool ok = e < tot && ch < nchv && q + 3 >= 0 && q < o.T;
      const int off = ((chan0 + ch) * o.T + q) * 4;
      if (ok && q < 0) {                 // straddles position 0: a negative offset fails the range check as a whole (first channel)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          v[i][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, q + k >= 0 ? off + 4 * k : 0x7f000000, 0, 0));
      } else {
        v[i] = buf_load4(rs, ok ? off : 0x7f000000);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = eb + tid + i * 256;
      if (e >= tot) continue;
      const int ch = (int)(((float)e + 0.5f) * inv_nvr);
      const int ee = 4 * (e - ch * nvr);
      const int q = qbase + ee;
      int col = (int)(((float)ee + 0.5f) * inv_s);
      int phi = ee - col * s;
      float* rows = xs + ch * s * XS;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float val = v[i][k];
        if (act) val = fmaxf(val, val * sl);
        val = (ch < nchv && q + k >= 0 && q + k < o.T) ? val * sc : 0.f;   // a float4 may straddle a row end
        rows[phi * XS + col] = val;
        if (++phi == s) { phi = 0; ++col; }
      }
    }
  }
}

// Scalar (dword) register tile with full padding logic: used for short / unaligned sequences (T = 50, 63, ...) and
// for the chunks at the sequence ends, so that those too are prefetched one chunk ahead instead of being staged
// element by element. Supports the prologues NONE / LRELU (aux unused) and MASK_LRELU (aux = activation output).
template <int NS>
struct RegS { float v[NS]; };

template <int NS>
__device__ __forceinline__ void tile_issue_s(RegS<NS>& t, const float* base, int T, int nvalid, int nrows, int span, int q0,
                                             int reflect, int tid) {
  const int total = nrows * span;
  const float inv = 1.0f / (float)span;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int e = tid + i * 256;
    const int r = (int)(((float)e + 0.5f) * inv);
    int q = q0 + (e - r * span);
    if (reflect) { if (q < 0) q = -q; else if (q >= T) q = 2 * (T - 1) - q; }
    t.v[i] = 0.f;
    if (e < total && r < nvalid && q >= 0 && q < T) t.v[i] = base[(long)r * T + q];
  }
}

template <int NS>
__device__ __forceinline__ void tile_commit_s(const RegS<NS>& t, const RegS<NS>* a, const Xf& xf, float* dst, int XS, int nrows,
                                              int span, int tid) {
  const int total = nrows * span;
  const float inv = 1.0f / (float)span;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int e = tid + i * 256;
    if (e >= total) continue;
    const int r = (int)(((float)e + 0.5f) * inv);
    float v = t.v[i];
    if (xf.kind == XF_LRELU) v = fmaxf(v, v * xf.slope);
    else if (xf.kind == XF_MASK_LRELU) v = a->v[i] > 0.f ? v : v * xf.slope;
    dst[r * XS + (e - r * span)] = v * xf.scale;
  }
}

// Phase-cycle instrumentation for tools/lean_phase_prof.py (diagnostic build only: `make prof`; no stamp executes in the product .so).
#ifdef LEAN_PROF
// each instrumented translation unit defines its own buffer pointer (no relocatable device code): PROF_DEFINE(setter)
#define PROF_DEFINE(setter) namespace tdvc { static __device__ unsigned long long* g_lean_prof = nullptr; } \
  extern "C" int setter(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tdvc::g_lean_prof), &buf, sizeof(buf)); }
#define PROF_DECL unsigned long long pt_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long pl_ = __builtin_amdgcn_s_memtime(); const unsigned long long pr0_ = __builtin_amdgcn_s_memrealtime();
#define PROF(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pt_[i] += n_ - pl_; pl_ = n_; }
#define PROF_WAITV() __builtin_amdgcn_s_waitcnt(0x0f70);   /* vmcnt(0) only (gfx9 encoding: lgkmcnt/expcnt fields left at max) */
#define PROF_END { pt_[7] = __builtin_amdgcn_s_memrealtime() - pr0_; if (g_lean_prof && threadIdx.x == 0) { \
    unsigned long long* o_ = g_lean_prof + 10 * ((unsigned long long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x); \
    for (int i_ = 0; i_ < 10; ++i_) o_[i_] = pt_[i_]; } }
#else
#define PROF_DEFINE(setter)
#define PROF_DECL
#define PROF(i)
#define PROF_WAITV()
#define PROF_END
#endif

}  // namespace tdvc
