// C-ABI entry points for the convolution family: maps a (tdvc_conv_desc, args) pair onto the
// reduced stride-1 problem of conv_common.h and picks the MFMA or the scalar kernel.
#include "../../include/tdvc.h"
#include "conv_common.h"
#include "conv_lean.h"
#include "conv_wgrad_lean.h"
#include "conv_small_group.h"
#include "api_util.h"

namespace tdvc {
template <int MODE> hipError_t launch_conv_gemm(GemmConvP, int, hipStream_t);
template <int MODE> hipError_t launch_conv_scalar(GemmConvP, int, hipStream_t);
template <int MODE> hipError_t launch_conv_wgrad(WgradP, int, int, hipStream_t);
template <int MODE> hipError_t launch_conv_wgrad_scalar(WgradP, int, long, float*, hipStream_t);
int wgrad_geometry(WgradP& p, int B, int* bpb_out);
bool wgrad_mfma_supported(int J);
hipError_t launch_slab_reduce(const float*, int, long, long, float*, int, long, hipStream_t, long n_w = -1, float* dbias = nullptr);
hipError_t launch_bias_grad(const Opnd&, int, int, int, float*, hipStream_t);
hipError_t fold_flush(hipStream_t st);
void fold_set_defer(int on);
void fold_reset(hipStream_t st);
}  // namespace tdvc

namespace tdvc {


}  // namespace tdvc

using namespace tdvc;

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline bool ok_bs(const void* p, long bs) { return !p || (bs < (1L << 31)); }
static inline bool vec_ptr(const void* p, long bs) { return !p || (al16(p) && (bs & 3) == 0); }

// Prologue kind of the lean kernel for an operand transform; -1 = not supported there.
static int lean_xfk(const tdvc_xform& x, float* slope, float* scale, const float** aux, long* aux_bs) {
  *scale = x.scale == 0.f ? 1.f : x.scale; *aux = x.aux; *aux_bs = x.aux_bs; *slope = x.slope;
  switch (x.kind) {
    case TDVC_XF_NONE: *slope = 1.f; *aux = nullptr; return LXF_ACT;
    case TDVC_XF_LRELU: *aux = nullptr; return (x.slope >= 0.f && x.slope <= 1.f) ? LXF_ACT : -1;
    case TDVC_XF_FILM_LRELU: return (x.slope >= 0.f && x.slope <= 1.f) ? LXF_FILM : -1;
    case TDVC_XF_MASK_LRELU: return LXF_MASK_LRELU;
    case TDVC_XF_MASK_TANH: return LXF_MASK_TANH;
    default: return -1;
  }
}

static bool lean_shape_ok(const tdvc_conv_desc* d) {
  return d->kind == TDVC_CONV && d->stride == 1 && d->groups == 1 && d->Tin == d->Tout && ((d->Tin & 3) == 0 || d->Tin <= 80);
}

namespace tdvc { int g_force_generic = 0; }   // test-only switch, like the tdvc_debug_* hooks (misc_kernels.hip)
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

// Grouped strided conv with 4 -> 4 channels per group (discriminator layer 4): vector-ALU kernels of conv_small_group.hip
static bool small_group_ok(const tdvc_conv_desc* d) {
  return !g_force_generic && g_knob[2] == 0 && d->kind == TDVC_CONV && d->groups >= 16 && d->Cin == 4 * d->groups && d->Cout == 4 * d->groups &&
         d->dilation == 1 && !d->reflect && d->w_cin == 0 && d->stride >= 2 && d->stride <= 8 && d->K <= 48;
}
static void small_group_base(const tdvc_conv_desc* d, SmallGroupP& q) {
  q.B = d->B; q.G = d->groups; q.Tin = d->Tin; q.Tout = d->Tout; q.K = d->K; q.s = d->stride; q.pad = d->pad;
  q.in_scale = 1.f; q.out_scale = 1.f; q.dy_scale = 1.f;
}

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
  if (small_group_ok(d) && a->x_xf.kind <= TDVC_XF_LRELU && !a->res && !a->add && !a->bias3) {
    SmallGroupP q = {};
    small_group_base(d, q);
    q.x = a->x; q.x_bs = a->x_bs; q.w = a->w; q.bias = a->bias; q.y = a->y; q.y_bs = a->y_bs;
    q.act_in = a->x_xf.kind == TDVC_XF_LRELU; q.slope_in = a->x_xf.slope; q.in_scale = a->x_xf.scale == 0.f ? 1.f : a->x_xf.scale;
    q.post = a->post_act; q.post_slope = a->post_slope; q.out_scale = a->out_scale == 0.f ? 1.f : a->out_scale;
    const hipError_t e = launch_small_group_fwd(q, (hipStream_t)stream);
    if (e == hipSuccess) return TDVC_OK;
    if (e != hipErrorNotSupported) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
  }
  if (!g_force_generic && lean_shape_ok(d) && (d->Cin & 3) == 0) {
    LeanP q = {};
    float slope, scale; const float* aux; long aux_bs;
    const int xfk = lean_xfk(a->x_xf, &slope, &scale, &aux, &aux_bs);
    const int cw = (d->w_cin > 0 ? d->w_cin : d->Cin) * d->K;
    const float* w = a->w + (long)d->w_cin_off * d->K;
    if (xfk >= 0 && (cw & 3) == 0 && al16(w) && ok_bs(a->x, a->x_bs) && ok_bs(a->y, a->y_bs) && ok_bs(a->res, a->res_bs) &&
        ok_bs(a->add, a->add_bs) && ok_bs(aux, aux_bs)) {
      q.x = a->x; q.w = w; q.y = a->y; q.bias = a->bias; q.bias3 = a->bias3; q.res = a->res; q.add = a->add; q.aux = aux;
      q.x_bs = (int)a->x_bs; q.y_bs = (int)a->y_bs; q.res_bs = (int)a->res_bs; q.add_bs = (int)a->add_bs; q.aux_bs = (int)aux_bs;
      q.T = d->Tin; q.Cin = d->Cin; q.Cout = d->Cout; q.Cw = cw; q.K = d->K; q.d = d->dilation; q.pad = d->pad; q.reflect = d->reflect;
      q.post = a->post_act; q.slope = slope; q.in_scale = scale; q.out_scale = a->out_scale == 0.f ? 1.f : a->out_scale;
      q.add_scale = 1.f; q.m_slope = a->post_slope;
      q.sbits = a->sign_bits; q.sb_bs = (int)a->sign_bits_bs;
      q.vec = ((d->Tin & 3) == 0 && vec_ptr(a->x, a->x_bs) && vec_ptr(a->y, a->y_bs) && vec_ptr(a->res, a->res_bs) &&
               vec_ptr(a->add, a->add_bs) && vec_ptr(aux, aux_bs)) ? 1 : 0;
      if (!q.vec && d->Tin > 80) goto generic_fwd;
      hipError_t e = launch_conv_lean(q, d->B, xfk, EPI_FWD, (hipStream_t)stream);
      if (e == hipSuccess) return TDVC_OK;
      if (e != hipErrorNotSupported) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
    }
  }
generic_fwd:
  if (a->sign_bits) return tdvc_fail(TDVC_EUNSUPPORTED, "conv_fwd: sign_bits needs a stride-1 conv with Tout % 32 == 0, Tout > 80 and 16-byte aligned operands");
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
  if (small_group_ok(d) && a->epilogue == TDVC_DG_PLAIN && !a->add &&
      (a->dy_xf.kind == TDVC_XF_NONE || (a->dy_xf.kind == TDVC_XF_MASK_LRELU && a->dy_xf.aux))) {
    SmallGroupP q = {};
    small_group_base(d, q);
    q.dy = a->dy; q.dy_bs = a->dy_bs; q.w = a->w; q.y = a->dx; q.y_bs = a->dx_bs;
    q.dy_scale = a->dy_xf.scale == 0.f ? 1.f : a->dy_xf.scale;
    if (a->dy_xf.kind == TDVC_XF_MASK_LRELU) { q.mask = a->dy_xf.aux; q.mask_bs = a->dy_xf.aux_bs; q.m_slope = a->dy_xf.slope; }
    const hipError_t e = launch_small_group_dgrad(q, (hipStream_t)stream);
    if (e == hipSuccess) return TDVC_OK;
    if (e != hipErrorNotSupported) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
  }
  if (!g_force_generic && a->wt && lean_shape_ok(d) && (d->Cout & 3) == 0 && (d->K - 1) * d->dilation - d->pad >= 0) {
    LeanP q = {};
    float slope, scale; const float* aux; long aux_bs;
    const int xfk = lean_xfk(a->dy_xf, &slope, &scale, &aux, &aux_bs);
    const int cw = d->Cout * d->K;                                  // wt rows: [ci] -> (co, k) contiguous
    const float* w = a->wt + (long)d->w_cin_off * cw;
    const int epi = a->epilogue == TDVC_DG_PLAIN ? EPI_PLAIN : (a->epilogue == TDVC_DG_MASK_LRELU ? EPI_MASK : EPI_FILM);
    if (xfk >= 0 && (cw & 3) == 0 && al16(w) && ok_bs(a->dy, a->dy_bs) && ok_bs(a->dx, a->dx_bs) && ok_bs(a->add, a->add_bs) &&
        ok_bs(aux, aux_bs) && ok_bs(a->x_in, a->x_in_bs) && ok_bs(a->gb, a->gb_bs) && ok_bs(a->dgb, a->dgb_bs) &&
        (epi == EPI_PLAIN || a->x_in || (epi == EPI_MASK && a->x_sign_bits)) && (epi != EPI_FILM || (a->gb && a->dgb))) {
      q.x = a->dy; q.w = w; q.y = a->dx; q.add = a->add; q.aux = aux; q.mx = a->x_in; q.gb = a->gb; q.dgb = a->dgb;
      q.x_bs = (int)a->dy_bs; q.y_bs = (int)a->dx_bs; q.add_bs = (int)a->add_bs; q.aux_bs = (int)aux_bs; q.mx_bs = (int)a->x_in_bs;
      q.gb_bs = (int)a->gb_bs; q.dgb_bs = (int)a->dgb_bs;
      q.T = d->Tin; q.Cin = d->Cout; q.Cout = d->Cin; q.Cw = cw; q.K = d->K; q.d = d->dilation;
      q.pad = (d->K - 1) * d->dilation - d->pad; q.flip = 1; q.mirror = d->reflect ? d->pad : 0;
      q.slope = slope; q.in_scale = scale; q.out_scale = 1.f; q.add_scale = a->add_scale; q.m_slope = a->slope;
      if (epi == EPI_MASK && a->x_sign_bits) { q.mbits = a->x_sign_bits; q.mb_bs = (int)a->x_sign_bits_bs; }
      q.vec = ((d->Tin & 3) == 0 && vec_ptr(a->dy, a->dy_bs) && vec_ptr(a->dx, a->dx_bs) && vec_ptr(a->add, a->add_bs) && vec_ptr(aux, aux_bs) &&
               vec_ptr(a->x_in, a->x_in_bs) && vec_ptr(a->gb, a->gb_bs) && vec_ptr(a->dgb, a->dgb_bs)) ? 1 : 0;
      if (!q.vec && d->Tin > 80) goto generic_dgrad;
      hipError_t e = launch_conv_lean(q, d->B, xfk, epi, (hipStream_t)stream);
      if (e == hipSuccess) return TDVC_OK;
      if (e != hipErrorNotSupported) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
      if (q.mbits && a->x_in) {   // shape outside the sign-bit path: the fp32 mask source does the same job
        q.mbits = nullptr;
        e = launch_conv_lean(q, d->B, xfk, epi, (hipStream_t)stream);
        if (e == hipSuccess) return TDVC_OK;
        if (e != hipErrorNotSupported) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
      }
    }
  }
generic_dgrad:
  if (a->epilogue == TDVC_DG_MASK_LRELU && !a->x_in && a->x_sign_bits)
    return tdvc_fail(TDVC_EUNSUPPORTED, "conv_dgrad: x_sign_bits without x_in needs a stride-1 conv with Tin % 32 == 0, Tin > 80 and 16-byte aligned operands");
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

static bool wgrad_lean_ok(const tdvc_conv_desc* d) {
  const bool wide = d->Cout >= 32 && d->Cin >= 32;     // register-tile kernel: walks (sample, tile) chunks, any length
  return !g_force_generic && d->kind == TDVC_CONV && d->stride == 1 && d->groups == 1 && d->Tin == d->Tout &&
         (d->Tout > 128 || wide) && wgrad_lean_supported(d->K, d->dilation) && (d->K != 15 || d->Cout <= 16 || d->Cin <= 16);
}

namespace tdvc {
int wgrad_lean_nslab(int R, int Cin, int N, int K, int B);
bool wgrad_x6_ok(int R, int Cin, int T, int K, int dil, int pad, int reflect);
void wgrad_x6_plan(int R, int T, int B, int* ntiles, int* tpb, int* ngroups);
hipError_t launch_conv_wgrad_x6(const WgLeanP& q, int B, hipStream_t st);
}
// split-bf16 weight-grad kernel (conv_wgrad_x6.hip): the layer geometry it takes
static bool wgrad_x6_desc_ok(const tdvc_conv_desc* d) {
  return !g_force_generic && d->kind == TDVC_CONV && d->stride == 1 && d->groups == 1 && d->Tin == d->Tout && d->w_cin == 0 &&
         wgrad_x6_ok(d->Cout, d->Cin, d->Tout, d->K, d->dilation, d->pad, d->reflect);
}

extern "C" size_t tdvc_conv_wgrad_workspace(const tdvc_conv_desc* d) {
  if (check_desc(d)) return 0;
  size_t small = small_group_ok(d) ? small_group_wgrad_workspace(d->B, d->groups, d->K) : 0;
  if (wgrad_lean_ok(d)) {
    size_t lean = (size_t)wgrad_lean_nslab(d->Cout, d->Cin, d->Tout, d->K, d->B) * ((size_t)d->Cout * d->Cin * d->K + d->Cout) * sizeof(float);
    if (wgrad_x6_desc_ok(d)) {       // which of the two kernels runs depends on the operand transforms: size for either
      int nt, tpb, ng;
      wgrad_x6_plan(d->Cout, d->Tout, d->B, &nt, &tpb, &ng);
      const size_t x6 = (size_t)ng * ((size_t)d->Cout * d->Cin * d->K + d->Cout) * sizeof(float);
      if (x6 > lean) lean = x6;
    }
    return lean;
  }
  WgradP p = {};
  fill_wgrad(d, nullptr, p);
  if (!wgrad_use_mfma(p)) return small;
  int bpb;
  const int nslab = wgrad_geometry(p, d->B, &bpb);
  const long wsize = (long)d->groups * p.w_sg;
  const size_t gen = (size_t)nslab * (size_t)wsize * sizeof(float);
  return gen > small ? gen : small;
}

extern "C" int tdvc_conv_wgrad(const tdvc_conv_desc* d, const tdvc_conv_wgrad_args* a, void* stream) {
  if (int rc = check_desc(d)) return rc;
  if (!a || !a->x || !a->dy) return tdvc_fail(TDVC_EINVAL, "conv_wgrad: null pointer");
  hipStream_t st = (hipStream_t)stream;
  WgradP p = {};
  fill_wgrad(d, a, p);
  const long wsize = (long)d->groups * p.w_sg;
  hipError_t e = hipSuccess;
  bool done = false, bias_done = false;
  if (a->dw && small_group_ok(d) && a->x_xf.kind <= TDVC_XF_LRELU &&
      (a->dy_xf.kind == TDVC_XF_NONE || (a->dy_xf.kind == TDVC_XF_MASK_LRELU && a->dy_xf.aux))) {
    SmallGroupP q = {};
    small_group_base(d, q);
    q.x = a->x; q.x_bs = a->x_bs; q.dy = a->dy; q.dy_bs = a->dy_bs;
    q.act_in = a->x_xf.kind == TDVC_XF_LRELU; q.slope_in = a->x_xf.slope; q.in_scale = a->x_xf.scale == 0.f ? 1.f : a->x_xf.scale;
    q.dy_scale = a->dy_xf.scale == 0.f ? 1.f : a->dy_xf.scale;
    if (a->dy_xf.kind == TDVC_XF_MASK_LRELU) { q.mask = a->dy_xf.aux; q.mask_bs = a->dy_xf.aux_bs; q.m_slope = a->dy_xf.slope; }
    e = launch_small_group_wgrad(q, a->dw, a->dbias, a->workspace, a->workspace_bytes, st);
    if (e == hipSuccess) return TDVC_OK;
    if (e != hipErrorNotSupported) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
    e = hipSuccess;
  }
  auto unit_scale = [](float sc) { return sc == 0.f || sc == 1.f; };
  if (a->dw && wgrad_lean_ok(d) && wgrad_x6_desc_ok(d) && a->x_xf.kind <= TDVC_XF_LRELU && a->dy_xf.kind == TDVC_XF_NONE &&
      unit_scale(a->x_xf.scale) && unit_scale(a->dy_xf.scale) && (a->x_xf.kind == TDVC_XF_NONE || (a->x_xf.slope > 0.f && a->x_xf.slope <= 1.f)) &&
      al16(a->x) && al16(a->dy) && (a->x_bs & 3) == 0 && (a->dy_bs & 3) == 0) {
    // 3-tap conv with 65..144 input channels (FiLM cond_var.2): the split-bf16 x6 kernel, dy and x read once per 32-row block
    WgLeanP q = {};
    q.a = p.a; q.x = p.x; q.R = d->Cout; q.Cin = d->Cin; q.N = d->Tout;
    int nt, tpb, nslab;
    wgrad_x6_plan(d->Cout, d->Tout, d->B, &nt, &tpb, &nslab);
    const long sstride = wsize + d->Cout;
    const size_t need = (size_t)nslab * (size_t)sstride * sizeof(float);
    if (!a->workspace || a->workspace_bytes < need) return tdvc_fail(TDVC_EWORKSPACE, "conv_wgrad: workspace too small");
    q.slab = (float*)a->workspace; q.slab_stride = sstride; q.bias_off = a->dbias ? wsize : -1;
    e = launch_conv_wgrad_x6(q, d->B, st);
    if (e != hipSuccess) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
    const int rowlen = d->Cin * d->K;
    e = launch_slab_reduce(q.slab, nslab, sstride, a->dbias ? sstride : wsize, a->dw, rowlen, rowlen, st, wsize, a->dbias);
    if (e != hipSuccess) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
    return TDVC_OK;
  }
  if (a->dw && wgrad_lean_ok(d)) {
    WgLeanP q = {};
    q.a = p.a; q.x = p.x; q.R = d->Cout; q.Cin = d->Cin; q.N = d->Tout; q.pad = d->pad; q.K = d->K; q.reflect = d->reflect;
    const int nslab = wgrad_lean_nslab(d->Cout, d->Cin, d->Tout, d->K, d->B);
    const long sstride = wsize + d->Cout;                       // weights + per-slab bias partials
    const size_t need = (size_t)nslab * (size_t)sstride * sizeof(float);
    if (!a->workspace || a->workspace_bytes < need) return tdvc_fail(TDVC_EWORKSPACE, "conv_wgrad: workspace too small");
    q.slab = (float*)a->workspace; q.slab_stride = sstride;
    auto okp = [](const Opnd& o) { return al16(o.p) && (o.bs & 3) == 0 && (o.T & 3) == 0 && (!o.xf.aux || (al16(o.xf.aux) && (o.xf.aux_bs & 3) == 0)); };
    q.vec = (okp(q.a) && okp(q.x)) ? 1 : 0;
    q.bias_off = a->dbias ? wsize : -1;
    e = launch_conv_wgrad_lean(q, d->B, d->K, d->dilation, st);
    if (e == hipSuccess) {
      const int rowlen = d->Cin * d->K;
      const long nred = a->dbias ? sstride : wsize;
      if (d->w_cin > 0) e = launch_slab_reduce(q.slab, nslab, sstride, nred, a->dw + (long)d->w_cin_off * d->K, rowlen, (long)d->w_cin * d->K, st, wsize, a->dbias);
      else e = launch_slab_reduce(q.slab, nslab, sstride, nred, a->dw, rowlen, rowlen, st, wsize, a->dbias);
      if (e != hipSuccess) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
      done = true; bias_done = true;
    } else if (e != hipErrorNotSupported) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
  }
  if (a->dw && !done) {
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
  if (a->dbias && !bias_done) {
    Opnd dy; dy.p = a->dy; dy.bs = a->dy_bs; dy.T = d->Tout; dy.Cg = d->Cout / d->groups; dy.xf = to_xf(a->dy_xf);
    e = launch_bias_grad(dy, d->Tout, d->Cout, d->B, a->dbias, st);
    if (e != hipSuccess) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
  }
  return TDVC_OK;
}


// FiLM conditioning forward (model/generator.py:86-92,103-104): gb = cond_var.2(LeakyReLU(cond_var.0(c))), where
// cond_var.0 is evaluated as [time-constant speaker part = k3] + [n_var-channel excitation window of its weight].
extern "C" int tdvc_film_cond_fwd(const tdvc_film_cond_args* a, void* stream) {
  if (!a || !a->exc || !a->w0 || !a->k3 || !a->w2 || !a->gb) return tdvc_fail(TDVC_EINVAL, "film_cond_fwd: null pointer");
  if (a->B <= 0 || a->T < 4 || a->n_cond <= a->n_var || a->n_var <= 0 || a->C2 <= 0) return tdvc_fail(TDVC_EINVAL, "film_cond_fwd: bad shape");
  LeanP q = {};
  q.x = a->exc; q.x_bs = (int)a->exc_bs; q.w = a->w2; q.bias = a->b2; q.y = a->gb; q.y_bs = (int)a->gb_bs;
  q.T = a->T; q.Cin = a->n_cond; q.Cout = a->C2; q.Cw = a->n_cond * 3; q.K = 3; q.d = 1; q.pad = 1;
  q.cw = a->w0 + (long)(a->n_cond - a->n_var) * 3; q.cw_stride = a->n_cond * 3; q.Cv = a->n_var;
  q.k3 = a->k3; q.cv0 = a->cv0; q.cv0_bs = (int)a->cv0_bs;
  q.slope = a->slope; q.in_scale = 1.f; q.out_scale = 1.f; q.add_scale = 1.f; q.m_slope = a->slope; q.post = TDVC_POST_NONE;
  q.vec = ((a->T & 3) == 0 && vec_ptr(a->gb, a->gb_bs) && vec_ptr(a->cv0, a->cv0_bs) && al16(a->w2) && ((a->n_cond * 3) & 3) == 0) ? 1 : 0;
  hipError_t e = launch_conv_lean_cond(q, a->B, (hipStream_t)stream);
  if (e == hipErrorNotSupported) return tdvc_fail(TDVC_EUNSUPPORTED, "film_cond_fwd: shape outside the fused kernel's contract");
  return e == hipSuccess ? TDVC_OK : tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
}

// Deferred weight-gradient folds (include/tdvc.h)
extern "C" void tdvc_fold_defer(int on) { tdvc::fold_set_defer(on); }
extern "C" void tdvc_fold_reset(void* stream) { tdvc::fold_reset((hipStream_t)stream); }
extern "C" int tdvc_fold_flush(void* stream) {
  const hipError_t e = tdvc::fold_flush((hipStream_t)stream);
  if (e != hipSuccess) return tdvc_fail(TDVC_ELAUNCH, hipGetErrorString(e));
  return TDVC_OK;
}
