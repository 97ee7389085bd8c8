/* tdvc.h — C ABI of libtdvc_hip.so: MI355X (gfx950) kernels for TD-VC-GAN's G+D train step.
 *
 * The reference (vicpc00/td-vc-gan) has NO native/FFI interface: its boundary is the Python
 * nn.Module surface. This ABI is what sits directly beneath that surface; each
 * entry point names the ATen operator instance(s) of the reference it replaces (file:line under
 * /root/reference). INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions (all entry points):
 *   - plain C, caller-owned DEVICE pointers, fp32, tensors [B][C][T] with T fastest; a batch stride
 *     (in elements) is passed wherever a channel-slice of a larger tensor may be used;
 *   - asynchronous on the hipStream_t passed as void* (NULL = default stream); no allocation, no
 *     synchronisation, no host read-back inside -> every call is hipGraph-capturable;
 *   - returns 0 (TDVC_OK) or a negative tdvc_status; never throws. tdvc_last_error() gives text;
 *   - re-entrant: the compute entry points keep no mutable state (per-kernel LDS-size attributes are set under
 *     std::call_once). The only process-global state is the TEST-ONLY switches tdvc_set_force_generic and
 *     tdvc_debug_*: they exist so that the parity tests can reach and name every kernel instance; the product path
 *     never calls them and they must not be flipped while another thread is launching.
 */
#ifndef TDVC_H
#define TDVC_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum { TDVC_OK = 0, TDVC_EINVAL = -1, TDVC_EWORKSPACE = -2, TDVC_ELAUNCH = -3, TDVC_EUNSUPPORTED = -4 } tdvc_status;

/* Operand transform applied while a tensor is staged into LDS (fused prologue). */
typedef enum {
  TDVC_XF_NONE = 0,       /* v */
  TDVC_XF_LRELU = 1,      /* leaky_relu(v, slope)                      — nn.LeakyReLU before every G conv (model/generator.py:74,82) */
  TDVC_XF_FILM_LRELU = 2, /* leaky_relu(v*(1+gamma)+beta, slope)       — FiLM (model/generator.py:104-107) fused into posconv's load */
  TDVC_XF_MASK_LRELU = 3, /* v * (aux>0 ? 1 : slope)                   — backward of an in-place LeakyReLU (model/discriminator.py:20) */
  TDVC_XF_MASK_TANH = 4   /* v * (1-aux^2)                             — backward of nn.Tanh (model/generator.py:337,347) */
} tdvc_xform_kind;

typedef struct {
  int32_t kind;        /* tdvc_xform_kind */
  float slope;         /* LeakyReLU slope */
  float scale;         /* multiplies the transformed value (1 = none) */
  const float* aux;    /* FILM: gamma/beta tensor [B][2C][T]; MASK_*: activation output, same shape as operand */
  int64_t aux_bs;      /* batch stride of aux (elements) */
} tdvc_xform;

typedef enum { TDVC_CONV = 0, TDVC_CONV_TRANSPOSE = 1 } tdvc_conv_kind;
typedef enum { TDVC_POST_NONE = 0, TDVC_POST_LRELU = 1, TDVC_POST_TANH = 2 } tdvc_post_act;
typedef enum {
  TDVC_DG_PLAIN = 0,     /* dx = acc */
  TDVC_DG_MASK_LRELU = 1,/* dx = acc * lrelu'(x_in)                       (pre-activation conv) */
  TDVC_DG_FILM = 2       /* x_in = h: m = lrelu'(h(1+g)+b); dh2 = acc*m; dx = dh2*(1+g); dgb = [dh2*h ; dh2] */
} tdvc_dgrad_epilogue;

/* One 1-D convolution layer instance. Replaces aten::conv1d / aten::conv_transpose1d /
 * aten::convolution_backward for nn.Conv1d / nn.ConvTranspose1d / F.conv1d call sites:
 * model/generator.py:17-22,75-92,146-157,165-168,214-249,300-362 ; model/discriminator.py:17-37,100-102. */
typedef struct {
  int32_t kind;          /* tdvc_conv_kind */
  int32_t B, Cin, Cout;  /* Cin/Cout are the module's in/out channels (for transpose: x has Cin, y has Cout) */
  int32_t Tin, Tout;
  int32_t K, stride, dilation, pad, groups;
  int32_t reflect;       /* 1: padding_mode='reflect' (stride 1 only), 0: zeros */
  /* Input-channel window of the weight tensor (groups == 1, kind == TDVC_CONV): the weight is
   * [Cout][w_cin][K] and this call uses input channels [w_cin_off, w_cin_off + Cin). 0/0 = whole tensor.
   * Lets FiLM's cond_var.0 (model/generator.py:88) split into its time-constant speaker-embedding part
   * (128 channels, evaluated on a length-3 signal) and its 8-channel excitation part (SURVEY §2.2 reduction 2). */
  int32_t w_cin, w_cin_off;
} tdvc_conv_desc;

typedef struct {
  const float* x; int64_t x_bs;      /* input activation */
  tdvc_xform x_xf;                   /* prologue on x */
  const float* w;                    /* effective weight, module layout: conv [Cout][Cin/g][K]; transpose [Cin][Cout/g][K] */
  const float* bias;                 /* [Cout] or NULL */
  const float* res; int64_t res_bs;  /* optional residual added before post_act (FiLM block shortcut) */
  int32_t post_act; float post_slope;
  float out_scale;                   /* y = out_scale * post(conv+bias+res) + (add ? add : 0) */
  const float* add; int64_t add_bs;  /* optional running sum (MRF mean, model/generator.py:192-193) */
  float* y; int64_t y_bs;
  const float* bias3;                /* optional [B][Cout][3]: per-sample bias for t==0 / interior / t==Tout-1 */
  uint32_t* sign_bits; int64_t sign_bits_bs; /* optional output [B][Cout][Tout/32] (batch stride in words): bit t%32 of word t/32 =
                                        (y[b][co][t] > 0) -- a 32x smaller LeakyReLU mask source for the input-grad of the NEXT layer
                                        (tdvc_conv_dgrad_args.x_sign_bits). Needs a stride-1 conv, Tout % 32 == 0, Tout > 80, 16-byte
                                        aligned operands (TDVC_EUNSUPPORTED otherwise) */
} tdvc_conv_fwd_args;

typedef struct {
  const float* dy; int64_t dy_bs;    /* upstream gradient w.r.t. y */
  tdvc_xform dy_xf;                  /* e.g. MASK_LRELU with aux = y for post-activated layers; scale = out_scale */
  const float* w;
  const float* wt;                   /* optional: the same weight pre-transposed to [Cin(_w)][Cout][K] (stride-1, groups==1 convs);
                                        lets the input-grad kernel stream contiguous weight rows (written by tdvc_weight_norm_fwd) */
  int32_t epilogue;                  /* tdvc_dgrad_epilogue */
  const float* x_in; int64_t x_in_bs; float slope; /* tensor the forward prologue read (mask / FiLM source) */
  const float* gb; int64_t gb_bs;    /* FiLM gamma/beta (TDVC_DG_FILM) */
  float* dgb; int64_t dgb_bs;        /* FiLM gradient out [B][2C][T] (TDVC_DG_FILM) */
  const float* add; int64_t add_bs; float add_scale; /* dx += add_scale*add (residual path gradient) */
  float* dx; int64_t dx_bs;
  const uint32_t* x_sign_bits; int64_t x_sign_bits_bs; /* optional, TDVC_DG_MASK_LRELU: the sign bits of x_in as written by
                                        tdvc_conv_fwd_args.sign_bits ([B][Cin][Tin/32]); read instead of x_in (which may then be NULL).
                                        Same shape limits as sign_bits */
} tdvc_conv_dgrad_args;

typedef struct {
  const float* x; int64_t x_bs; tdvc_xform x_xf;
  const float* dy; int64_t dy_bs; tdvc_xform dy_xf;
  float* dw;        /* accumulated: dw += sum, module layout */
  float* dbias;     /* accumulated, or NULL */
  void* workspace; size_t workspace_bytes;
} tdvc_conv_wgrad_args;

int tdvc_conv_fwd(const tdvc_conv_desc* d, const tdvc_conv_fwd_args* a, void* stream);
/* Split-bf16 x6 forward (conv_fwd_x6.hip): the same result as tdvc_conv_fwd at fp32 accuracy on the bf16 matrix pipe, for 3-tap
 * stride-1 'same' convs with 65..160 input channels and Cout % 32 == 0 (FiLM cond_var.2), prologue none / LeakyReLU, bias, no
 * other epilogue operand; TDVC_EUNSUPPORTED otherwise (call tdvc_conv_fwd). The weights come pre-split into three exact bf16
 * pieces: tdvc_conv_x6_weight_planes writes tdvc_conv_x6_weight_planes_bytes() bytes from the fp32 weight [Cout][Cin][3]; redo it
 * whenever the weight changes (once per optimizer step). */
size_t tdvc_conv_x6_weight_planes_bytes(int32_t Cout, int32_t Cin, int32_t K);
int tdvc_conv_x6_weight_planes(const float* w, int32_t Cout, int32_t Cin, int32_t K, void* planes, void* stream);
int tdvc_conv_fwd_x6(const tdvc_conv_desc* d, const tdvc_conv_fwd_args* a, const void* weight_planes, void* stream);
int tdvc_conv_dgrad(const tdvc_conv_desc* d, const tdvc_conv_dgrad_args* a, void* stream);
int tdvc_conv_wgrad(const tdvc_conv_desc* d, const tdvc_conv_wgrad_args* a, void* stream);
size_t tdvc_conv_wgrad_workspace(const tdvc_conv_desc* d);
/* Deferred weight-gradient folds. Every weight-grad entry point (tdvc_conv_wgrad, tdvc_film_cond0_bwd) ends with a fold of
 * its per-block partial slabs (in `workspace`) into dw / dbias. With tdvc_fold_defer(1) that fold is queued on the stream
 * instead of launched, and up to 24 queued folds run as ONE launch: when the queue is full, when a new fold targets a
 * gradient that is already queued, or on tdvc_fold_flush(stream). The caller then owns two duties: every call gets a
 * workspace region that stays untouched until the flush, and tdvc_fold_flush runs before anything reads dw / dbias or
 * reuses a region. Default: off (each call folds before it returns its stream position). Process-wide switch; the queues
 * are per stream. */
void tdvc_fold_defer(int on);
int tdvc_fold_flush(void* stream);
/* Drop the queued folds of `stream` WITHOUT running them: for the failure paths (a backward pass that raised, an abandoned
 * graph capture), after which the queued descriptors point into workspace regions that later calls overwrite. */
void tdvc_fold_reset(void* stream);

/* TEST-ONLY process-global switches (see the conventions above).
 * tdvc_set_force_generic: route convs to the scalar (non-MFMA) kernels, to cross-check the two code paths.
 * tdvc_debug_force_tile: pin the tile configuration of the lean stride-1 conv kernel (cfg 0..6 = <M_REP,N_REP,WM,WN> of
 *   conv_lean.hip: 0 <1,4,1,4>, 1 <2,4,1,4>, 2 <4,4,1,4>, 3 <1,1,1,4>, 4 <1,4,4,1>, 5 <3,4,1,4>, 6 <1,2,2,2>; -1 = automatic,
 *   grid-aware choice). Lets small test shapes run the instances that only large batches select.
 * tdvc_debug_trace(1) clears and starts, (2) resumes, (0) stops recording the demangled names of the conv-family kernel
 *   instantiations launched; tdvc_debug_trace_dump copies them ('\n'-separated, NUL-terminated) and returns the size needed. */
void tdvc_set_force_generic(int on);
void tdvc_debug_force_tile(int cfg);
void tdvc_debug_knob(int which, int value); /* tuning knobs for A/B measurements: 0 = XCD-aware block order of the lean conv kernel (1 = on; default 0: measured null on this path); 3 = one block per CU in the fused conditioning backward (diagnostic); 4 = 1: no sample folding of short sequences (T = 16 / 32) in the lean conv kernel; 5 = 1: exact-fp32 MFMA instead of the split-bf16 x6 weight-grad kernel (conv_wgrad_x6.hip); 6 = 1: tdvc_conv_fwd_x6 always answers TDVC_EUNSUPPORTED; 7 = 1: two instead of three resident blocks per CU in the x6 weight-grad's plan */
void tdvc_debug_lds_cap(int bytes);   /* tuning knob: LDS bytes per block the lean kernel's chunk-size choice may use (0 = built-in) */
void tdvc_debug_trace(int on);
size_t tdvc_debug_trace_dump(char* buf, size_t cap);
/* TEST-ONLY: fill the LDS of every CU with the bit pattern `word` (e.g. 0xFFFFFFFF = a NaN) so that a kernel which reads
 * LDS it never wrote shows up as NaN in its output instead of passing on whatever finite data the last kernel left. */
int tdvc_debug_poison_lds(uint32_t word, void* stream);
/* TOOLS-ONLY: an empty dispatch (pmc_marker_kernel) that separates table entries in a rocprofv3 trace (tools/microbench_kernels.py) */
int tdvc_debug_marker(void* stream);

/* Fused forward of one FiLM residual block (model/generator.py:96-111), narrow long-sequence case (C == 16, T % 4 == 0, T >= 512,
 * reflect padding (K-1)*dilation/2, 16-byte aligned operands; TDVC_EUNSUPPORTED otherwise -> run the two tdvc_conv_fwd calls):
 *   h = conv_kd(LeakyReLU(x)) + b1 (stored for the backward pass);  y = scale * (conv_1x1(LeakyReLU(h*(1+gamma)+beta)) + b2 + x) + add
 * gb = [gamma; beta] [B][2C][T] or NULL (no FiLM: LeakyReLU(h)); add = MRF running sum or NULL. Bit-identical to the two-launch path. */
typedef struct {
  int32_t B, C, T, K, dilation;
  const float* x; int64_t x_bs;
  const float* w1; const float* b1;        /* dilated conv [C][C][K], bias or NULL */
  float* h; int64_t h_bs;
  const float* gb; int64_t gb_bs;
  const float* w2; const float* b2;        /* 1x1 conv [C][C][1], bias or NULL */
  const float* add; int64_t add_bs;
  float scale, slope;
  float* y; int64_t y_bs;
} tdvc_film_block_args;
int tdvc_film_block_fwd(const tdvc_film_block_args* a, void* stream);

/* Fused FiLM conditioning forward (model/generator.py:86-92, 103): gb = cond_var.2(LeakyReLU(cond_var.0(c))) with
 * c = [speaker embedding (n_cond-n_var channels, constant in time) ; excitation (n_var channels)]. The time-constant part
 * enters as k3 [B][n_cond][3] (= cond_var.0 restricted to the embedding channels, evaluated on a length-3 signal, bias
 * included: left edge / interior / right edge); the excitation part is computed per tile in LDS, so the n_cond-channel
 * intermediate is written once (cv0, for the backward pass; may be NULL) and never re-read by the forward. */
typedef struct {
  int32_t B, T, n_cond, n_var, C2;      /* C2 = 2 * n_channel (gamma and beta) */
  const float* exc; int64_t exc_bs;     /* [B][n_var][T] */
  const float* w0;                      /* cond_var.0 effective weight [n_cond][n_cond][3] */
  const float* k3;                      /* [B][n_cond][3] */
  const float* w2; const float* b2;     /* cond_var.2 effective weight [C2][n_cond][3], bias [C2] */
  float* cv0; int64_t cv0_bs;           /* optional pre-activation cond_var.0 output [B][n_cond][T] */
  float* gb; int64_t gb_bs;             /* [B][C2][T] */
  float slope;
} tdvc_film_cond_args;
int tdvc_film_cond_fwd(const tdvc_film_cond_args* a, void* stream);
/* The same forward with cond_var.2 in split-bf16 x6 arithmetic (conv_fwd_x6.hip) and cond_var.0's excitation window computed per tile inside
 * that kernel: cv0 is written once (for the backward pass) and never read back, its sign bits (cv0_sign_bits [B][n_cond][T/32], optional,
 * T % 32 == 0) come out of the same registers. w2_planes = tdvc_conv_x6_weight_planes of cond_var.2's weight (a->w2 is not read).
 * Needs n_var == 8, 64 < n_cond <= 160, n_cond % 4 == 0, C2 % 32 == 0, T >= 128, T % 4 == 0; TDVC_EUNSUPPORTED otherwise. */
int tdvc_film_cond_fwd_x6(const tdvc_film_cond_args* a, const void* w2_planes, uint32_t* cv0_sign_bits, int64_t bits_bs, void* stream);

/* Backward of cond_var.0 in the split formulation, everything that consumes d_cv0 = dL/d(cond_var.0 output) in one pass:
 * dexc (input-grad wrt the excitation), the excitation window of dW (accumulated into dw0, module layout
 * [n_cond][n_cond][3]) and dk3 (adjoint of the 3-valued bias: first step, interior sum, last step). Replaces the
 * conv_wgrad + conv_dgrad + edge_sum3 calls on the same tensor. Needs T % 4 == 0, n_var == 8, n_cond <= 144. */
typedef struct {
  int32_t B, T, n_cond, n_var;
  const float* dcv; int64_t dcv_bs;     /* [B][n_cond][T] */
  const float* exc; int64_t exc_bs;     /* [B][n_var][T] */
  const float* w0;                      /* cond_var.0 effective weight [n_cond][n_cond][3] */
  float* dexc; int64_t dexc_bs;         /* [B][n_var][T], optional */
  float* dk3;                           /* [B][n_cond][3] */
  float* dw0;                           /* weight-gradient accumulator of cond_var.0, optional */
  void* workspace; size_t workspace_bytes;   /* tdvc_film_cond0_bwd_workspace() bytes when dw0 is given */
} tdvc_film_cond0_bwd_args;
size_t tdvc_film_cond0_bwd_workspace(int32_t B, int32_t T, int32_t n_cond, int32_t n_var);
int tdvc_film_cond0_bwd(const tdvc_film_cond0_bwd_args* a, void* stream);

/* Backward of the whole conditioning network behind cond_var.2's output gradient in ONE kernel (film_cond_fused_bwd.hip):
 * d_cv0 = LeakyReLU-mask * (cond_var.2 input-grad of dgb) is produced and consumed per tile on chip -- dexc, the excitation
 * window of dw0 and dk3 as in tdvc_film_cond0_bwd -- and never written to HBM. Replaces tdvc_conv_dgrad (cond_var.2) +
 * tdvc_film_cond0_bwd. The mask comes from the sign bits the forward stored (cv0_sign_bits, [B][n_cond][T/32], T % 32 == 0)
 * or, when that is NULL, from the stored fp32 cond_var.0 output (cv0). Needs T % 4 == 0, n_var == 8, n_cond <= 144,
 * C2 % 32 == 0, wt2 = cond_var.2's effective weight pre-transposed to [n_cond][C2][3]; TDVC_EUNSUPPORTED otherwise. */
typedef struct {
  int32_t B, T, n_cond, n_var, C2;
  const float* dgb; int64_t dgb_bs;                      /* [B][C2][T] */
  const float* wt2;                                      /* [n_cond][C2][3] */
  const uint32_t* cv0_sign_bits; int64_t cv0_sign_bits_bs;   /* batch stride in words; or NULL */
  const float* cv0; int64_t cv0_bs;                      /* [B][n_cond][T], read only when cv0_sign_bits is NULL */
  const float* exc; int64_t exc_bs;                      /* [B][n_var][T] */
  const float* w0;                                       /* cond_var.0 effective weight [n_cond][n_cond][3] */
  float* dexc; int64_t dexc_bs;                          /* [B][n_var][T], optional */
  float* dk3;                                            /* [B][n_cond][3] */
  float* dw0;                                            /* weight-gradient accumulator of cond_var.0, optional */
  void* workspace; size_t workspace_bytes;               /* tdvc_film_cond_bwd_workspace() bytes when dw0 is given */
  float slope;
} tdvc_film_cond_bwd_args;
size_t tdvc_film_cond_bwd_workspace(int32_t B, int32_t T, int32_t n_cond, int32_t n_var);
int tdvc_film_cond_bwd(const tdvc_film_cond_bwd_args* a, void* stream);

/* k3 [B][n_cond][3] = cond_var.0 restricted to the n_const time-constant speaker-embedding channels of its input,
 * evaluated on a length-3 constant signal with the conv's zero 'same' padding (bias b0 included; w0 = effective weight
 * [n_cond][n_cond][3], emb [B][n_const]). bwd: demb [B][n_const] (optional), and the matching window of dw0 / db0 is
 * accumulated (+=, optional). Replaces three conv launches per FiLM block on a 3-sample sequence. */
int tdvc_film_k3_fwd(const float* emb, int64_t emb_bs, const float* w0, const float* b0, float* k3, int32_t B, int32_t n_const,
                     int32_t n_cond, void* stream);
int tdvc_film_k3_bwd(const float* dk3, const float* emb, int64_t emb_bs, const float* w0, float* demb, float* dw0, float* db0,
                     int32_t B, int32_t n_const, int32_t n_cond, void* stream);
/* The same for all FiLM blocks of one MRF stage (they share `emb`): nblk <= 16 weight / bias / k3 pointers as HOST arrays (read
 * before the call returns); the backward writes `demb` already summed over the blocks and accumulates each block's dw0 / db0. */
int tdvc_film_k3_multi_fwd(const float* emb, int64_t emb_bs, const float* const* w0s, const float* const* b0s, float* const* k3s,
                           int32_t nblk, int32_t B, int32_t n_const, int32_t n_cond, void* stream);
int tdvc_film_k3_multi_bwd(const float* const* dk3s, const float* emb, int64_t emb_bs, const float* const* w0s, float* demb,
                           float* const* dw0s, float* const* db0s, int32_t nblk, int32_t B, int32_t n_const, int32_t n_cond, void* stream);

/* Multi-tensor weight norm (old-style nn.utils.weight_norm, dim=0; model/generator.py:14,
 * util/__init__.py:16-20, model/discriminator.py:11): w[row] = g[row] * v[row] / ||v[row]||, one wave per
 * dim-0 slice, every weight-normed tensor of a model in ONE launch. `params` is the model's flat parameter
 * arena (v and g live in it), `w` the flat effective-weight arena; the four DEVICE tables give, per row,
 * the element offsets of v / g in `params`, of w in `w`, and the row length. */
int tdvc_weight_norm_fwd(const float* params, float* w, const int64_t* row_voff, const int64_t* row_goff,
                         const int64_t* row_woff, const int32_t* row_len, int nrows, void* stream);
/* Same, additionally writing the [Cin][Cout][K] transposed copy of selected tensors into `wt`: per row, row_tbase =
 * offset of element (ci=0, co=row, k=0) in wt, row_tstride = Cout*K, row_k = K (0 = tensor has no transposed copy). */
int tdvc_weight_norm_fwd_t(const float* params, float* w, float* wt, const int64_t* row_voff, const int64_t* row_goff,
                           const int64_t* row_woff, const int32_t* row_len, const int64_t* row_tbase,
                           const int64_t* row_tstride, const int32_t* row_k, int nrows, void* stream);
/* grads[v] (+)= g/||v|| * (dw - v <dw,v>/||v||^2), grads[g] (+)= <dw,v>/||v||; `grads` mirrors `params`. */
int tdvc_weight_norm_bwd(const float* params, const float* dw, float* grads, const int64_t* row_voff,
                         const int64_t* row_goff, const int64_t* row_woff, const int32_t* row_len, int nrows,
                         int accumulate, void* stream);

/* Fused AdamW over a flat parameter arena (torch.optim.AdamW semantics, train.py:188-189:
 * lr, betas, eps 1e-8, weight_decay 1e-2). The 1-based step count of this update comes from step_dev
 * (a DEVICE int32, so that a captured hipGraph stays valid across replays) or, if NULL, from `step`. */
int tdvc_adamw(float* p, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
               float eps, float weight_decay, int step, const int32_t* step_dev, float grad_scale, void* stream);
/* The same with one more factor on the gradient read from DEVICE memory (clip_coef_dev[0], or NULL): the gradient-clipping
 * coefficient of tdvc_grad_clip_coef, so that clipping costs no pass over the gradients and no host synchronisation. */
int tdvc_adamw_clipped(float* p, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int step, const int32_t* step_dev, float grad_scale,
                       const float* clip_coef_dev, void* stream);
/* torch.nn.utils.clip_grad_norm_ on a flat gradient buffer (/root/reference/train.py:289-290, 489-490): out[0] = min(1, max_norm /
 * (grad_scale * ||grad||_2 + 1e-6)), out[1] = grad_scale * ||grad||_2 (the norm of the gradient the optimizer sees: with data
 * parallelism grad holds the all-reduced SUM and grad_scale = 1 / world). Deterministic two-pass reduction; workspace >= 1024
 * floats. The coefficient is applied by tdvc_adamw_clipped (the gradient buffer itself is left unscaled). */
int tdvc_grad_clip_coef(const float* grad, int64_t n, float max_norm, float grad_scale, float* workspace, float* out, void* stream);
/* util.roll_batches on the last axis (/root/reference/util/__init__.py:91-102, used by util.audio.add_jitter, train.py:335-336):
 * y[b][c][t] = x[b][c][(t - shift[b]) mod T], shift = int64 [B] on the device. */
int tdvc_roll_batches(const float* x, const int64_t* shift, float* y, int B, int C, int T, void* stream);
int tdvc_inc_i32(int32_t* p, int32_t by, void* stream);

/* Element-wise / reductions over [B][C][T] tensors. */
int tdvc_l2norm_fwd(const float* x, float* y, float* inv_norm, int B, int C, int T, float eps, void* stream);   /* F.normalize(dim=1), model/generator.py:271 */
int tdvc_l2norm_bwd(const float* y, const float* inv_norm, const float* dy, float* dx, int B, int C, int T, void* stream);
int tdvc_gather_ch_fwd(const float* x, const int64_t* label, float* y, int B, int C, int T, void* stream);       /* x.gather(1,label), model/discriminator.py:47-51 */
int tdvc_gather_ch_bwd(const float* dy, const int64_t* label, float* dx, int B, int C, int T, void* stream);
int tdvc_concat_cond(const float* emb, const float* exc, float* c, int B, int Ce, int Cx, int T, void* stream);   /* cat([emb.repeat(T), exc]), model/generator.py:387-399 */
int tdvc_concat_cond_bwd(const float* dc, float* demb, float* dexc, int B, int Ce, int Cx, int T, int accumulate_emb, void* stream);
/* out[b][c][0..2] = d[b][c][0], sum_{t=1..T-2} d[b][c][t], d[b][c][T-1]  (adjoint of the bias3 broadcast) */
int tdvc_edge_sum3(const float* d, float* out, int B, int C, int T, void* stream);
int tdvc_axpby(const float* a, const float* b, float* y, float alpha, float beta, int64_t n, void* stream);      /* y = alpha*a + beta*b (b may be NULL) */
int tdvc_fill(float* y, float value, int64_t n, void* stream);
int tdvc_sum_n(const float* const* srcs, int nsrc, float* y, int64_t n, void* stream);   /* y = srcs[0] + ... + srcs[nsrc-1], 1 <= nsrc <= 16 (srcs: HOST array of device pointers, read before the call returns): the gradient of a tensor with several consumers (autograd's AccumulateGrad chain) in one pass */
/* WaveNet gate of the SSL encoder's WN stack (model/ssl_encoder.py:8-15, fused_add_tanh_sigmoid_multiply):
 * acts[b][c][t] = tanh(xin[b][c][t] + g[b][c][t]) * sigmoid(xin[b][H+c][t] + g[b][H+c][t]); xin / g are [B][2H][T]
 * (g optional: NULL when the stack has no global conditioning, as in SSLEncoder), acts [B][H][T]; batch strides in elements.
 * bwd: dxin [B][2H][T] from dacts [B][H][T] (the gradient wrt g equals dxin). */
int tdvc_gate_fwd(const float* xin, int64_t xin_bs, const float* g, int64_t g_bs, float* acts, int64_t acts_bs, int B, int H, int T, void* stream);
int tdvc_gate_bwd(const float* xin, int64_t xin_bs, const float* g, int64_t g_bs, const float* dacts, int64_t dacts_bs, float* dxin,
                  int64_t dxin_bs, int B, int H, int T, void* stream);

/* InstanceNorm1d(affine=False, eps) + conditional scale/shift: y = (1+gamma)*IN(x)+beta
 * (model/conditional_instance_norm.py:4-19). gb is [B][2C][Tg] with Tg = 1 (Linear path) or T (Conv path). */
int tdvc_cin_fwd(const float* x, const float* gb, float* y, float* mean, float* rstd, int B, int C, int T, int Tg, float eps, void* stream);
int tdvc_cin_bwd(const float* x, const float* gb, const float* dy, const float* mean, const float* rstd,
                 float* dx, float* dgb, int B, int C, int T, int Tg, void* stream);

/* Loss reductions. Each writes loss_out[0] (+)= weight*loss and, where present, the gradient
 * already scaled by weight*upstream. */
int tdvc_mse_const_fwd(const float* x, int64_t n, float target, float weight, float* loss_out, void* stream);      /* F.mse_loss(x, const), train.py:273-279,329 */
int tdvc_mse_const_bwd(const float* x, int64_t n, float target, float weight, const float* upstream, float* dx, void* stream);
int tdvc_l1_fwd(const float* a, const float* b, int64_t n, float weight, float* loss_out, void* stream);          /* F.l1_loss, util/losses.py:63 */
int tdvc_l1_bwd(const float* a, const float* b, int64_t n, float weight, const float* upstream, float* da, int accumulate, void* stream);
/* All L1 terms of one loss in one launch (multiscale_feat_loss, util/losses.py:55-68: a term per discriminator feature map).
 * pairs: HOST array read before the call returns. fwd: loss_out += sum_i weight_i / n_i * sum|a_i - b_i|;
 * bwd: da_i = sign(a_i - b_i) * weight_i / n_i * upstream[0]; an entry with b == NULL zero-fills da (samples of a batched map
 * the loss does not read) and is ignored by fwd. */
typedef struct { const float* a; const float* b; float* da; int64_t n; float weight; } tdvc_l1_pair;
int tdvc_l1_multi_fwd(const tdvc_l1_pair* pairs, int npairs, float* loss_out, void* stream);
int tdvc_l1_multi_bwd(const tdvc_l1_pair* pairs, int npairs, const float* upstream, void* stream);

/* log-mel L1 (util/losses.py:28-53 + torchaudio MelSpectrogram semantics). The two GEMMs run on
 * tdvc_conv_*: STFT = Conv1d(1 -> 2F, K=n_fft, stride=hop) with the windowed DFT basis as weight over the
 * reflect-padded signal, mel projection = 1x1 Conv1d(F -> n_mels) with the filterbank as weight. These are
 * the element-wise pieces around them. spec is [B][2F][N] (real rows then imaginary rows). */
int tdvc_reflect_pad_fwd(const float* x, float* y, int B, int T, int pad, void* stream);       /* torch.stft(center=True, pad_mode='reflect') */
int tdvc_reflect_pad_bwd(const float* dy, float* dx, int B, int T, int pad, void* stream);
int tdvc_power_fwd(const float* spec, float* power, int B, int F, int N, void* stream);        /* |.|^2, power=2 */
int tdvc_power_bwd(const float* spec, const float* dpower, float* dspec, int B, int F, int N, void* stream);
int tdvc_log_l1_fwd(const float* a, const float* b, int64_t n, float floor_, float weight, float* loss_out, void* stream); /* l1(log(clamp(a)), log(clamp(b))) */
int tdvc_log_l1_bwd(const float* a, const float* b, int64_t n, float floor_, float weight, const float* upstream, float* da, void* stream);

/* Contrastive InfoNCE (util/losses.py:70-116): cosine logits, softmax, CE against class 0, both
 * directions, fused with its backward. idx_* [B][T][N] int32 are the negative positions (already skipping
 * self, util/losses.py:82-83). loss_out[0] += weight*CE; dX, dY += weight * dCE/d{X,Y} (zero them first). */
/* Softmax cross-entropy, mean over the batch (F.cross_entropy on the latent classifier's logits, train.py:302, :422).
 * fwd: *loss_out += weight * mean_b CE(logits[b], labels[b]) and the softmax probabilities go to prob [B][K];
 * bwd: dlogits = weight / B * (prob - onehot) * upstream[0]. labels are int64 class indices. */
int tdvc_cross_entropy_fwd(const float* logits, const int64_t* labels, int B, int K, float weight, float* loss_out, float* prob, void* stream);
int tdvc_cross_entropy_bwd(const float* prob, const int64_t* labels, int B, int K, float weight, const float* upstream, float* dlogits,
                           void* stream);

/* Sine + noise excitation from a frame-level F0 track (util/__init__.py:22-50, f0_to_excitation): f0 [B][n_frames] in Hz
 * (0 = unvoiced; the last frame is dropped as in the reference), output exc [B][(n_frames-1)*step]. The random draws are
 * inputs: noise_v / noise_u [B][T] standard normal (voiced / unvoiced samples), start_phase [1] in radians (device pointer). */
int tdvc_f0_to_excitation(const float* f0, const float* noise_v, const float* noise_u, const float* start_phase, float* exc,
                          int B, int n_frames, int step, float sampling_rate, int linear, void* stream);

int tdvc_contrastive_fwd_bwd(const float* X, const float* Y, const int32_t* idx_x, const int32_t* idx_y, int B, int C, int T, int N,
                             float weight, float* loss_out, float* dX, float* dY, void* stream);

const char* tdvc_last_error(void);
int tdvc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TDVC_H */
