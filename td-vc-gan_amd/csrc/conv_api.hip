// C-ABI entry points for the convolution family: maps a (tdvc_conv_desc, args) pair onto the
// reduced stride-1 problem of conv_common.h and picks the MFMA or the scalar kernel.
#include "../../include/tdvc.h"
#include "conv_common.h"
#include "api_util.h"

namespace tdvc {
template <int MODE> hipError_t launch_conv_gemm(GemmConvP, int, hipStream_t);
template <int MODE> hipError_t launch_conv_scalar(GemmConvP, int, hipStream_t);
template <int MODE> hipError_t launch_conv_wgrad(WgradP, int, int, hipStream_t);
template <int MODE> hipError_t launch_conv_wgrad_scalar(WgradP, int, long, float*, hipStream_t);
int wgrad_geometry(WgradP& p, int B, int* bpb_out);
bool wgrad_mfma_supported(int J);
hipError_t launch_slab_reduce(const float*, int, long, long, float*, int, long, hipStream_t);
hipError_t launch_bias_grad(const Opnd&, int, int, int, float*, hipStream_t);
}  // namespace tdvc

using namespace tdvc;

static int g_force_generic = 0;
extern "C" void tdvc_set_force_generic(int on) { g_force_generic = on; }

static Xf to_xf(const tdvc_xform& x) {
  Xf r; r.kind = x.kind; r.slope = x.slope; r.scale = x.scale == 0.f ? 1.f : x.scale;  // 0 = unset
  r.aux = x.aux; r.aux_bs = x.aux_bs; return r;
}

static int check_desc(const tdvc_conv_desc* d) {
  if (!d) return tdvc_fail(TDVC_EINVAL, "null desc");
  if (d->B <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->Tin <= 0 || d->Tout <= 0 || d->K <= 0 || d->stride <= 0 ||
      d->dilation <= 0 || d->groups <= 0 || d->pad < 0)
    return tdvc_fail(TDVC_EINVAL, "conv desc: non-positive dimension");
  if (d->Cin % d->groups || d->Cout % d->groups) return tdvc_fail(TDVC_EINVAL, "conv desc: channels not divisible by groups");
  if (d->stride > 1 && d->dilation != 1) return tdvc_fail(TDVC_EUNSUPPORTED, "strided conv with dilation");
  if (d->reflect && (d->stride != 1 || d->kind != TDVC_CONV)) return tdvc_fail(TDVC_EUNSUPPORTED, "reflect padding needs stride 1 conv");
  if (d->reflect && d->pad >= d->Tin) return tdvc_fail(TDVC_EINVAL, "reflect padding must be smaller than the input length");
  if (d->w_cin < 0 || d->w_cin_off < 0) return tdvc_fail(TDVC_EINVAL, "conv desc: negative weight channel window");
  if (d->w_cin > 0) {
    if (d->groups != 1 || d->kind != TDVC_CONV) return tdvc_fail(TDVC_EUNSUPPORTED, "weight channel window needs a plain conv with groups == 1");
    if (d->w_cin_off + d->Cin > d->w_cin) return tdvc_fail(TDVC_EINVAL, "conv desc: weight channel window out of range");
  }
  if (d->kind == TDVC_CONV) {
    long expect = ((long)d->Tin + 2L * d->pad - (long)d->dilation * (d->K - 1) - 1) / d->stride + 1;
    if (expect != d->Tout) return tdvc_fail(TDVC_EINVAL, "conv desc: Tout does not match conv arithmetic");
  } else if (d->kind == TDVC_CONV_TRANSPOSE) {
    if (d->stride < 2) return tdvc_fail(TDVC_EUNSUPPORTED, "transposed conv needs stride >= 2");
    long expect = ((long)d->Tin - 1) * d->stride - 2L * d->pad + (d->K - 1) + 1;
    if (d->Tout > expect + d->stride - 1 || d->Tout < expect) return tdvc_fail(TDVC_EINVAL, "conv desc: Tout does not match transposed conv arithmetic");
  } else return tdvc_fail(TDVC_EINVAL, "conv desc: unknown kind");
  return TDVC_OK;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

static bool use_mfma(const GemmConvP& p) {
  if (g_force_generic) return false;
  if (p.R <= 2 && p.x.Cg == 1 && p.Cy_g <= 2) return false;   // depthwise / single-channel FIR filters
  return (long)p.Cred * p.J >= 7;                             // Cred is zero-padded to a multiple of 4 in LDS
}

template <int MODE>
static int run_gemm(GemmConvP& p, int B, hipStream_t st) {
  hipError_t e = use_mfma(p) ? launch_conv_gemm<MODE>(p, B, st) : launch_conv_scalar<MODE>(p, B, st);
  return e == hipSuccess ? TDVC_OK : tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
}

static int dispatch_gemm(GemmConvP& p, int B, hipStream_t st) {
  switch (p.mode) {
    case MODE_DIRECT: return run_gemm<MODE_DIRECT>(p, B, st);
    case MODE_DOWN: return run_gemm<MODE_DOWN>(p, B, st);
    default: return run_gemm<MODE_UP>(p, B, st);
  }
}

extern "C" int tdvc_conv_fwd(const tdvc_conv_desc* d, const tdvc_conv_fwd_args* a, void* stream) {
  if (int rc = check_desc(d)) return rc;
  if (!a || !a->x || !a->w || !a->y) return tdvc_fail(TDVC_EINVAL, "conv_fwd: null pointer");
  const int Cin_g = d->Cin / d->groups, Cout_g = d->Cout / d->groups;
  GemmConvP p = {};
  p.x.p = a->x; p.x.bs = a->x_bs; p.x.T = d->Tin; p.x.Cg = Cin_g; p.x.xf = to_xf(a->x_xf);
  p.w = a->w; p.K = d->K; p.s = d->stride; p.pad = d->pad; p.groups = d->groups;
  p.y = a->y; p.y_bs = a->y_bs; p.Ty = d->Tout; p.Cy_g = Cout_g;
  p.epi = EPI_FWD; p.bias = a->bias; p.bias3 = a->bias3; p.res = a->res; p.res_bs = a->res_bs;
  p.post = a->post_act; p.post_slope = a->post_slope; p.out_scale = a->out_scale == 0.f ? 1.f : a->out_scale;
  p.add = a->add; p.add_bs = a->add_bs; p.add_scale = 1.f;
  p.w_sg = (long)Cout_g * Cin_g * d->K;
  if (d->kind == TDVC_CONV) {
    p.w_sm = (long)Cin_g * d->K; p.w_sc = d->K;
    if (d->w_cin > 0) { p.w_sm = (long)d->w_cin * d->K; p.w += (long)d->w_cin_off * d->K; }
    p.R = Cout_g; p.N = d->Tout;
    if (d->stride == 1) { p.mode = MODE_DIRECT; p.Cred = Cin_g; p.J = d->K; p.d = d->dilation; p.reflect = d->reflect; }
    else { p.mode = MODE_DOWN; p.Cred = Cin_g * d->stride; p.J = ceil_div(d->K, d->stride); p.d = 1; }
  } else {
    p.w_sm = d->K; p.w_sc = (long)Cout_g * d->K;
    p.mode = MODE_UP; p.R = Cout_g * d->stride; p.Cred = Cin_g; p.J = ceil_div(d->K, d->stride); p.d = 1;
    p.N = (d->Tout - 1 + d->pad) / d->stride + 1;
  }
  return dispatch_gemm(p, d->B, (hipStream_t)stream);
}

extern "C" int tdvc_conv_dgrad(const tdvc_conv_desc* d, const tdvc_conv_dgrad_args* a, void* stream) {
  if (int rc = check_desc(d)) return rc;
  if (!a || !a->dy || !a->w || !a->dx) return tdvc_fail(TDVC_EINVAL, "conv_dgrad: null pointer");
  const int Cin_g = d->Cin / d->groups, Cout_g = d->Cout / d->groups;
  GemmConvP p = {};
  p.x.p = a->dy; p.x.bs = a->dy_bs; p.x.T = d->Tout; p.x.Cg = Cout_g; p.x.xf = to_xf(a->dy_xf);
  p.w = a->w; p.K = d->K; p.s = d->stride; p.groups = d->groups;
  p.y = a->dx; p.y_bs = a->dx_bs; p.Ty = d->Tin; p.Cy_g = Cin_g; p.N = d->Tin;
  p.add = a->add; p.add_bs = a->add_bs; p.add_scale = a->add_scale;
  p.w_sg = (long)Cout_g * Cin_g * d->K;
  switch (a->epilogue) {
    case TDVC_DG_PLAIN: p.epi = EPI_PLAIN; break;
    case TDVC_DG_MASK_LRELU:
      if (!a->x_in) return tdvc_fail(TDVC_EINVAL, "conv_dgrad: mask epilogue needs x_in");
      p.epi = EPI_MASK; p.mx = a->x_in; p.mx_bs = a->x_in_bs; p.m_slope = a->slope; break;
    case TDVC_DG_FILM:
      if (!a->x_in || !a->gb || !a->dgb) return tdvc_fail(TDVC_EINVAL, "conv_dgrad: FiLM epilogue needs x_in, gb, dgb");
      p.epi = EPI_FILM; p.mx = a->x_in; p.mx_bs = a->x_in_bs; p.m_slope = a->slope;
      p.gb = a->gb; p.gb_bs = a->gb_bs; p.dgb = a->dgb; p.dgb_bs = a->dgb_bs; break;
    default: return tdvc_fail(TDVC_EINVAL, "conv_dgrad: unknown epilogue");
  }
  if (d->kind == TDVC_CONV) {
    p.w_sm = d->K; p.w_sc = (long)Cin_g * d->K;      // rows = input channel, reduced channel = output channel
    if (d->w_cin > 0) { p.w_sc = (long)d->w_cin * d->K; p.w += (long)d->w_cin_off * d->K; }
    if (d->stride == 1) {
      p.mode = MODE_DIRECT; p.R = Cin_g; p.Cred = Cout_g; p.J = d->K; p.d = d->dilation; p.tap_flip = 1;
      p.pad = (d->K - 1) * d->dilation - d->pad;
      if (p.pad < 0) return tdvc_fail(TDVC_EUNSUPPORTED, "conv_dgrad: padding larger than the receptive field");
      p.mirror_pad = d->reflect ? d->pad : 0;
    } else {
      p.mode = MODE_UP; p.R = Cin_g * d->stride; p.Cred = Cout_g; p.J = ceil_div(d->K, d->stride); p.d = 1; p.pad = d->pad;
      p.N = (d->Tin - 1 + d->pad) / d->stride + 1;
    }
  } else {
    p.w_sm = (long)Cout_g * d->K; p.w_sc = d->K;      // weight [Cin][Cout_g][K] read as a strided conv over dy
    p.mode = MODE_DOWN; p.R = Cin_g; p.Cred = Cout_g * d->stride; p.J = ceil_div(d->K, d->stride); p.d = 1; p.pad = d->pad;
  }
  return dispatch_gemm(p, d->B, (hipStream_t)stream);
}

static void fill_wgrad(const tdvc_conv_desc* d, const tdvc_conv_wgrad_args* a, WgradP& p) {
  const int Cin_g = d->Cin / d->groups, Cout_g = d->Cout / d->groups;
  p.groups = d->groups; p.K = d->K; p.s = d->stride; p.pad = d->pad;
  p.w_sg = (long)Cout_g * Cin_g * d->K;
  if (d->kind == TDVC_CONV) {
    if (a) { p.a.p = a->dy; p.a.bs = a->dy_bs; p.a.xf = to_xf(a->dy_xf); p.x.p = a->x; p.x.bs = a->x_bs; p.x.xf = to_xf(a->x_xf); }
    p.a.T = d->Tout; p.a.Cg = Cout_g; p.x.T = d->Tin; p.x.Cg = Cin_g;
    p.R = Cout_g; p.N = d->Tout; p.w_sm = (long)Cin_g * d->K; p.w_sc = d->K;
    if (d->stride == 1) { p.mode = MODE_DIRECT; p.Cred = Cin_g; p.J = d->K; p.d = d->dilation; p.reflect = d->reflect; }
    else { p.mode = MODE_DOWN; p.Cred = Cin_g * d->stride; p.J = ceil_div(d->K, d->stride); p.d = 1; }
  } else {
    // dW[ci][co][k] = sum_t x[ci][t] * dy[co][t*s - pad + k]: a strided-conv weight-grad with x and dy swapped
    if (a) { p.a.p = a->x; p.a.bs = a->x_bs; p.a.xf = to_xf(a->x_xf); p.x.p = a->dy; p.x.bs = a->dy_bs; p.x.xf = to_xf(a->dy_xf); }
    p.a.T = d->Tin; p.a.Cg = Cin_g; p.x.T = d->Tout; p.x.Cg = Cout_g;
    p.R = Cin_g; p.N = d->Tin; p.w_sm = (long)Cout_g * d->K; p.w_sc = d->K;
    p.mode = MODE_DOWN; p.Cred = Cout_g * d->stride; p.J = ceil_div(d->K, d->stride); p.d = 1;
  }
}

static bool wgrad_use_mfma(const WgradP& p) {
  if (g_force_generic) return false;
  if (p.R <= 2 && p.x.Cg == 1) return false;
  if (!wgrad_mfma_supported(p.J)) return false;
  return true;
}

extern "C" size_t tdvc_conv_wgrad_workspace(const tdvc_conv_desc* d) {
  if (check_desc(d)) return 0;
  WgradP p = {};
  fill_wgrad(d, nullptr, p);
  if (!wgrad_use_mfma(p)) return 0;
  int bpb;
  const int nslab = wgrad_geometry(p, d->B, &bpb);
  const long wsize = (long)d->groups * p.w_sg;
  return (size_t)nslab * (size_t)wsize * sizeof(float);
}

extern "C" int tdvc_conv_wgrad(const tdvc_conv_desc* d, const tdvc_conv_wgrad_args* a, void* stream) {
  if (int rc = check_desc(d)) return rc;
  if (!a || !a->x || !a->dy) return tdvc_fail(TDVC_EINVAL, "conv_wgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  WgradP p = {};
  fill_wgrad(d, a, p);
  const long wsize = (long)d->groups * p.w_sg;
  hipError_t e = hipSuccess;
  if (a->dw) {
    if (wgrad_use_mfma(p)) {
      int bpb;
      const int nslab = wgrad_geometry(p, d->B, &bpb);
      const size_t need = (size_t)nslab * (size_t)wsize * sizeof(float);
      if (!a->workspace || a->workspace_bytes < need) return tdvc_fail(TDVC_EWORKSPACE, "conv_wgrad: workspace too small");
      p.slab = (float*)a->workspace; p.slab_stride = wsize;
      e = p.mode == MODE_DIRECT ? launch_conv_wgrad<MODE_DIRECT>(p, d->B, bpb, st) : launch_conv_wgrad<MODE_DOWN>(p, d->B, bpb, st);
      if (e == hipSuccess) {
        const int rowlen = (int)p.w_sm;   // compact slab rows: [Cin_g*K]
        if (d->w_cin > 0) e = launch_slab_reduce(p.slab, nslab, wsize, wsize, a->dw + (long)d->w_cin_off * d->K, rowlen, (long)d->w_cin * d->K, st);
        else e = launch_slab_reduce(p.slab, nslab, wsize, wsize, a->dw, rowlen, rowlen, st);
      }
    } else {
      float* dw_base = a->dw;
      if (d->w_cin > 0) { p.w_sm = (long)d->w_cin * d->K; dw_base += (long)d->w_cin_off * d->K; }
      e = p.mode == MODE_DIRECT ? launch_conv_wgrad_scalar<MODE_DIRECT>(p, d->B, wsize, dw_base, st)
                                : launch_conv_wgrad_scalar<MODE_DOWN>(p, d->B, wsize, dw_base, st);
    }
    if (e != hipSuccess) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
  }
  if (a->dbias) {
    Opnd dy; dy.p = a->dy; dy.bs = a->dy_bs; dy.T = d->Tout; dy.Cg = d->Cout / d->groups; dy.xf = to_xf(a->dy_xf);
    e = launch_bias_grad(dy, d->Tout, d->Cout, d->B, a->dbias, st);
    if (e != hipSuccess) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
  }
  return TDVC_OK;
}
