"""Autograd operators over the C ABI (libtdvc_hip.so). torch is plumbing here: it owns device
memory, the stream and the autograd graph; every FLOP runs in the HIP kernels.

Weight gradients do not travel through autograd: each conv's wgrad kernel accumulates straight
into its model's flat `dW` / `G` arena (arena.py) and ParamArena.finish_grads() — queued to run at
the end of the backward pass — folds them into (weight_v, weight_g) gradients. That removes ~800
per-tensor accumulate launches per iteration.
"""
import ctypes as C
import os

import torch
from torch.autograd import Function

from . import _lib as L
from .arena import ConvSlot

SLOPE = 0.2


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class ConvSpec:
    """Static description of one conv layer + where its tensors live (ConvSlot)."""
    __slots__ = ('kind', 'cin', 'cout', 'k', 'stride', 'dil', 'pad', 'groups', 'reflect', 'slot', 'out_pad', 'w_cin', 'w_cin_off')

    def __init__(self, cin, cout, k, stride=1, pad=0, dil=1, groups=1, reflect=False, transposed=False, out_pad=0,
                 w_cin=0, w_cin_off=0):
        self.w_cin, self.w_cin_off = w_cin, w_cin_off
        self.kind = L.CONV_TRANSPOSE if transposed else L.CONV
        self.cin, self.cout, self.k, self.stride, self.dil, self.pad = cin, cout, k, stride, dil, pad
        self.groups, self.reflect, self.out_pad = groups, int(reflect), out_pad
        self.slot = None

    def tout(self, tin):
        if self.kind == L.CONV:
            return (tin + 2 * self.pad - self.dil * (self.k - 1) - 1) // self.stride + 1
        return (tin - 1) * self.stride - 2 * self.pad + self.k + self.out_pad

    def desc(self, B, tin):
        if self.reflect and self.pad >= tin:
            raise RuntimeError(f'Padding size should be less than the corresponding input dimension, but got: '
                               f'padding ({self.pad}, {self.pad}) at dimension 2 of input length {tin}')
        return L.ConvDesc(self.kind, B, self.cin, self.cout, tin, self.tout(tin), self.k, self.stride, self.dil,
                          self.pad, self.groups, self.reflect, self.w_cin, self.w_cin_off)


# ------------------------------------------------------------------------------- workspace
# Weight-gradient kernels leave per-block partial slabs in a workspace and fold them into dW. The folds are DEFERRED
# (include/tdvc.h: tdvc_fold_defer): the library queues them and runs up to 24 as one launch, which takes ~250 launches
# of mostly launch latency out of a backward pass. The duties that come with it live here: every weight-grad call gets
# its own region of a ring buffer (the slabs must survive until the flush), the ring flushes before it wraps, and
# fold_flush() runs before anything reads the gradients (ParamArena.finish_grads / segment hand-over, tests).
FOLD_DEFER = os.environ.get('TDVC_FOLD_DEFER', '1') == '1'
RING_MIN = 64 << 20                                                    # first allocation
RING_MAX = int(os.environ.get('TDVC_FOLD_RING_MB', '1024')) << 20      # the ring doubles on a wrap until it reaches this
_ws = {}


def fold_flush(device):
    """Launch the queued folds of `device`'s current stream (no-op when nothing is queued). Everything queued before is
    then ordered ahead of later launches on the stream, so the ring restarts at its beginning: its high-water mark is
    the slab volume between two flushes (one backward pass), not of the whole run."""
    if FOLD_DEFER:
        L.check(L.lib().tdvc_fold_flush(torch.cuda.current_stream(device).cuda_stream))
        ent = _ws.get(device)
        if ent is not None:
            ent[1] = 0


def workspace(device, nbytes):
    """A region of `nbytes` for the slabs of one weight-grad call -> (buffer, byte offset). The ring is sized by use: it
    starts at 64 MB and doubles (up to TDVC_FOLD_RING_MB) whenever a pass wraps it outside graph capture -- the eager
    warm-up step sizes it for the captured one; a wrap that still happens costs one extra fold launch, nothing else."""
    ent = _ws.get(device)
    need = max(nbytes, 1 << 20)
    capturing = torch.cuda.is_current_stream_capturing()
    grow = ent is None or ent[0].numel() < need
    if not grow and FOLD_DEFER and ent[1] + nbytes > ent[0].numel() and ent[0].numel() < RING_MAX and not capturing:
        grow = True                                  # wrapped: a larger ring saves the mid-pass flush from now on
        need = max(need, 2 * ent[0].numel())
    if grow:
        if capturing:
            raise L.TdvcError('wgrad workspace would have to grow during graph capture: run an eager step first')
        if ent is not None:
            fold_flush(device)
            torch.cuda.current_stream(device).synchronize()
        L.lib().tdvc_fold_defer(1 if FOLD_DEFER else 0)
        size = max(need * 2 if ent is None else need, RING_MIN) if FOLD_DEFER else need
        ent = [torch.empty(size, dtype=torch.uint8, device=device), 0]
        _ws[device] = ent
    if not FOLD_DEFER:
        return ent[0], 0
    off = ent[1]
    if off + nbytes > ent[0].numel():            # wrap: the queued folds still read the old regions
        fold_flush(device)
        off = 0
    ent[1] = (off + nbytes + 255) & ~255
    return ent[0], off


# ------------------------------------------------------------------------------- launch recorder (measurement only)
# bench.py sets RECORDER = [] around ONE eager iteration: every conv-family entry point then appends a description of its
# call (layer geometry, operand transforms, epilogue, operand shapes -- no tensors), from which the benchmark rebuilds each
# (op, launch shape) class on rotating operand sets for the roofline table. None = off (the product path: one branch).
RECORDER = None


def _spec_key(spec):
    return (spec.cin, spec.cout, spec.k, spec.stride, spec.pad, spec.dil, spec.groups, int(spec.reflect),
            int(spec.kind == L.CONV_TRANSPOSE), spec.out_pad, spec.w_cin, spec.w_cin_off)


def _xf(kind=L.XF_NONE, slope=SLOPE, scale=1.0, aux=None):
    if aux is None:
        return L.Xform(kind, slope, scale, None, 0)
    return L.Xform(kind, slope, scale, aux.data_ptr(), aux.stride(0))


def _bs(t):
    return t.stride(0)


def _check_layout(t):
    if t.stride(-1) != 1 or (t.dim() == 3 and t.stride(1) != t.shape[2]):
        raise L.TdvcError('operand must be [B,C,T] with contiguous (C,T) planes')


X6_FWD = os.environ.get('TDVC_X6_FWD', '1') == '1'      # split-bf16 x6 forward for FiLM's cond_var.2 (tdvc_conv_fwd_x6); 0 = exact-fp32 MFMA kernel
X6_FWD_MIN_COUT = 32  # smallest Cout routed to the split-bf16 forward (profiles/r03_d_fwd_x6_vs_fp32.txt: 1.19x the fp32 kernel at Cout = 32, 1.5-1.7x above)


def _weight_planes_x6(spec, device):
    """The three exact bf16 pieces of the layer's effective weight ([piece][Cout][tap][160] bf16). Cached on the arena that owns the
    weight and refreshed when it has re-materialised (arena.version: once per forward pass that may follow an optimizer step);
    stand-alone slots (tests, tools) split on every call."""
    s = spec.slot
    lib = L.lib()
    ent = s.arena.x6_planes.get(s.w) if s.arena is not None else None
    if ent is None or ent[0] != s.arena.version:
        buf = ent[1] if ent is not None else torch.empty(lib.tdvc_conv_x6_weight_planes_bytes(spec.cout, spec.cin, spec.k), dtype=torch.uint8, device=device)
        L.check(lib.tdvc_conv_x6_weight_planes(s.w, spec.cout, spec.cin, spec.k, buf.data_ptr(), torch.cuda.current_stream(device).cuda_stream))
        ent = (s.arena.version if s.arena is not None else None, buf)
        if s.arena is not None:
            s.arena.x6_planes[s.w] = ent
    return ent[1]


def conv_fwd_raw(spec: ConvSpec, x, x_xf, post=L.POST_NONE, res=None, add=None, out_scale=1.0, out=None, w_ptr=None, b_ptr=None,
                 bias3=None, sign_bits=None):
    B, _, tin = x.shape
    d = spec.desc(B, tin)
    y = out if out is not None else torch.empty((B, spec.cout, d.Tout), dtype=torch.float32, device=x.device)
    _check_layout(x); _check_layout(y)
    if (X6_FWD and spec.k == 3 and spec.kind == L.CONV and spec.stride == 1 and spec.dil == 1 and spec.pad == 1 and spec.groups == 1 and not spec.reflect
            and spec.w_cin == 0 and 64 < spec.cin <= 160 and spec.cout % 32 == 0 and spec.cout >= X6_FWD_MIN_COUT and tin >= 128 and tin % 4 == 0 and post == L.POST_NONE and res is None
            and add is None and bias3 is None and sign_bits is None and w_ptr is None and x_xf.kind in (L.XF_NONE, L.XF_LRELU) and out_scale == 1.0):
        a = L.ConvFwdArgs(x.data_ptr(), _bs(x), x_xf, spec.slot.w, (b_ptr if b_ptr is not None else spec.slot.b) or None, None, 0, post, SLOPE, 1.0,
                          None, 0, y.data_ptr(), _bs(y), None, None, 0)
        rc = L.lib().tdvc_conv_fwd_x6(C.byref(d), C.byref(a), _weight_planes_x6(spec, x.device).data_ptr(), _stream(x))
        if rc != L.EUNSUPPORTED:
            L.check(rc)
            if RECORDER is not None:
                RECORDER.append(('fwd_x6', _spec_key(spec), B, tin, x_xf.kind, bool((b_ptr if b_ptr is not None else spec.slot.b))))
            return y
    if RECORDER is not None:
        RECORDER.append(('fwd', _spec_key(spec), B, tin, x_xf.kind, post, res is not None, add is not None,
                         bool((b_ptr if b_ptr is not None else spec.slot.b)), bias3 is not None, sign_bits is not None))
    a = L.ConvFwdArgs(x.data_ptr(), _bs(x), x_xf, w_ptr if w_ptr is not None else spec.slot.w,
                      (b_ptr if b_ptr is not None else spec.slot.b) or None,
                      res.data_ptr() if res is not None else None, _bs(res) if res is not None else 0,
                      post, SLOPE, out_scale, add.data_ptr() if add is not None else None,
                      _bs(add) if add is not None else 0, y.data_ptr(), _bs(y),
                      bias3.data_ptr() if bias3 is not None else None,
                      sign_bits.data_ptr() if sign_bits is not None else None, _bs(sign_bits) if sign_bits is not None else 0)
    L.check(L.lib().tdvc_conv_fwd(C.byref(d), C.byref(a), _stream(x)))
    return y


def conv_dgrad_raw(spec: ConvSpec, dy, dy_xf, tin, epilogue=L.DG_PLAIN, x_in=None, gb=None, dgb=None, add=None,
                   add_scale=1.0, out=None, x_bits=None):
    B = dy.shape[0]
    d = spec.desc(B, tin)
    dx = out if out is not None else torch.empty((B, spec.cin, tin), dtype=torch.float32, device=dy.device)
    _check_layout(dy); _check_layout(dx)
    if RECORDER is not None:
        RECORDER.append(('dgrad', _spec_key(spec), B, tin, dy_xf.kind, epilogue, x_in is not None, x_bits is not None, add is not None,
                         bool(spec.slot.wt)))
    a = L.ConvDgradArgs(dy.data_ptr(), _bs(dy), dy_xf, spec.slot.w, spec.slot.wt or None, epilogue,
                        x_in.data_ptr() if x_in is not None else None, _bs(x_in) if x_in is not None else 0, SLOPE,
                        gb.data_ptr() if gb is not None else None, _bs(gb) if gb is not None else 0,
                        dgb.data_ptr() if dgb is not None else None, _bs(dgb) if dgb is not None else 0,
                        add.data_ptr() if add is not None else None, _bs(add) if add is not None else 0, add_scale,
                        dx.data_ptr(), _bs(dx),
                        x_bits.data_ptr() if x_bits is not None else None, _bs(x_bits) if x_bits is not None else 0)
    L.check(L.lib().tdvc_conv_dgrad(C.byref(d), C.byref(a), _stream(dy)))
    return dx


def conv_wgrad_raw(spec: ConvSpec, x, x_xf, dy, dy_xf):
    """Accumulates into the layer's dW / dbias arena slices. No-op for frozen layers / disabled arenas."""
    s = spec.slot
    if not s.trainable or (s.arena is not None and not s.arena.wgrad_enabled):
        return
    B, _, tin = x.shape
    d = spec.desc(B, tin)
    lib = L.lib()
    if RECORDER is not None:
        RECORDER.append(('wgrad', _spec_key(spec), B, tin, x_xf.kind, dy_xf.kind, bool(s.db)))
    nbytes = lib.tdvc_conv_wgrad_workspace(C.byref(d))
    ws, off = workspace(x.device, nbytes) if nbytes else (None, 0)
    a = L.ConvWgradArgs(x.data_ptr(), _bs(x), x_xf, dy.data_ptr(), _bs(dy), dy_xf, s.dw, s.db or None,
                        ws.data_ptr() + off if ws is not None else None, nbytes if ws is not None else 0)
    L.check(lib.tdvc_conv_wgrad(C.byref(d), C.byref(a), _stream(x)))
    if s.arena is not None:
        s.arena.note_grad(s)
    else:
        fold_flush(x.device)            # stand-alone use (tests, tools): the caller reads dw right away


PRE_NONE, PRE_LRELU = 0, 1


class ConvFn(Function):
    """y = post(conv(pre(x)) + bias [+ k3]) [+ add].  pre in {none, LeakyReLU}; post in {none, LeakyReLU, tanh};
    k3 [B,Cout,3] is a per-sample bias for (t == 0, interior, t == T-1) — the time-constant part of FiLM's
    conditioning conv evaluated on a length-3 signal."""

    @staticmethod
    def forward(ctx, x, add, token, spec, pre, post, k3=None):
        x = x.contiguous()
        xf = _xf(L.XF_LRELU if pre == PRE_LRELU else L.XF_NONE)
        if k3 is not None:
            if post != L.POST_NONE:
                raise L.TdvcError('k3 bias is only supported without a post-activation')
            k3 = k3.contiguous()
        y = conv_fwd_raw(spec, x, xf, post=post, add=add, bias3=k3)
        ctx.spec, ctx.pre, ctx.post, ctx.tin = spec, pre, post, x.shape[2]
        ctx.has_add, ctx.has_k3 = add is not None, k3 is not None
        ctx.save_for_backward(x, y if post != L.POST_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        spec = ctx.spec
        dy = dy.contiguous()
        if ctx.post == L.POST_LRELU:
            dy_xf = _xf(L.XF_MASK_LRELU, aux=y)
        elif ctx.post == L.POST_TANH:
            dy_xf = _xf(L.XF_MASK_TANH, aux=y)
        else:
            dy_xf = _xf()
        x_xf = _xf(L.XF_LRELU if ctx.pre == PRE_LRELU else L.XF_NONE)
        conv_wgrad_raw(spec, x, x_xf, dy, dy_xf)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.pre == PRE_LRELU:
                dx = conv_dgrad_raw(spec, dy, dy_xf, ctx.tin, L.DG_MASK_LRELU, x_in=x)
            else:
                dx = conv_dgrad_raw(spec, dy, dy_xf, ctx.tin, L.DG_PLAIN)
        dk3 = None
        if ctx.has_k3:
            B, Cc, T = dy.shape
            dk3 = torch.empty((B, Cc, 3), dtype=torch.float32, device=dy.device)
            L.check(L.lib().tdvc_edge_sum3(dy.data_ptr(), dk3.data_ptr(), B, Cc, T, _stream(dy)))
        return dx, (dy if ctx.has_add else None), None, None, None, None, dk3


def _token(*specs):
    """Weight gradients bypass autograd (they accumulate in the arena), so a trainable layer whose input
    carries no gradient would never get a backward call. A per-arena dummy leaf keeps it in the graph."""
    if not torch.is_grad_enabled():
        return None
    for sp in specs:
        s = sp.slot
        if s.trainable and s.arena is not None and s.arena.wgrad_enabled:
            return s.arena.token
    return None


def conv(x, spec, pre=PRE_NONE, post=L.POST_NONE, add=None, k3=None):
    return ConvFn.apply(x, add, _token(spec), spec, pre, post, k3)


class FilmBlockFn(Function):
    """One FiLM residual block, fused (model/generator.py:96-111):
         h   = conv_kd(lrelu(x)) + b1
         out = scale * (conv_1x1(lrelu(h*(1+gamma)+beta)) + b2 + x) + acc
    FiLM is applied while posconv stages its input tile, so h*(1+gamma)+beta never touches HBM; the
    backward emits dgamma/dbeta from the same dgrad epilogue that applies the LeakyReLU mask.
    `scale`/`acc` carry the MRF running mean (model/generator.py:192-193)."""

    @staticmethod
    def forward(ctx, x, gb, acc, token, conv_spec, pos_spec, scale):
        x = x.contiguous()
        if gb is not None:
            gb = gb.contiguous()
        h = out = None
        cs, ps = conv_spec, pos_spec
        if (FUSED_FILM_BLOCK and cs.cin == 16 and cs.cout == 16 and ps.cin == 16 and ps.cout == 16 and ps.k == 1 and cs.reflect and cs.stride == 1 and
                cs.groups == 1 and 2 * cs.pad == (cs.k - 1) * cs.dil and x.shape[2] % 4 == 0 and x.shape[2] >= 512):
            # narrow long-sequence blocks: the dilated conv, FiLM, the 1x1 conv and the residual in one launch (film_block.hip)
            B, _, T = x.shape
            h, out = torch.empty_like(x), torch.empty_like(x)
            accc = acc.contiguous() if acc is not None else None
            a = L.FilmBlockArgs(B, 16, T, cs.k, cs.dil, x.data_ptr(), _bs(x), cs.slot.w, cs.slot.b or None, h.data_ptr(), _bs(h),
                                gb.data_ptr() if gb is not None else None, _bs(gb) if gb is not None else 0, ps.slot.w, ps.slot.b or None,
                                accc.data_ptr() if accc is not None else None, _bs(accc) if accc is not None else 0, scale, SLOPE,
                                out.data_ptr(), _bs(out))
            rc = L.lib().tdvc_film_block_fwd(C.byref(a), _stream(x))
            if rc == L.EUNSUPPORTED:
                h = out = None
            else:
                L.check(rc)
                if RECORDER is not None:
                    RECORDER.append(('film_block_fwd', B, T, cs.k, cs.dil, gb is not None, accc is not None))
        if out is None:
            h = conv_fwd_raw(conv_spec, x, _xf(L.XF_LRELU))
            xf2 = _xf(L.XF_FILM_LRELU, aux=gb) if gb is not None else _xf(L.XF_LRELU)
            out = conv_fwd_raw(pos_spec, h, xf2, res=x, add=acc, out_scale=scale)
        ctx.cs, ctx.ps, ctx.scale = conv_spec, pos_spec, scale
        ctx.has_gb, ctx.has_acc = gb is not None, acc is not None
        ctx.save_for_backward(x, h, gb)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, h, gb = ctx.saved_tensors
        d_out = d_out.contiguous()
        s = ctx.scale
        T = x.shape[2]
        dy_xf = _xf(scale=s)
        if ctx.has_gb:
            dgb = torch.empty_like(gb)
            dh = conv_dgrad_raw(ctx.ps, d_out, dy_xf, T, L.DG_FILM, x_in=h, gb=gb, dgb=dgb)
            conv_wgrad_raw(ctx.ps, h, _xf(L.XF_FILM_LRELU, aux=gb), d_out, dy_xf)
        else:
            dgb = None
            dh = conv_dgrad_raw(ctx.ps, d_out, dy_xf, T, L.DG_MASK_LRELU, x_in=h)
            conv_wgrad_raw(ctx.ps, h, _xf(L.XF_LRELU), d_out, dy_xf)
        conv_wgrad_raw(ctx.cs, x, _xf(L.XF_LRELU), dh, _xf())
        dx = conv_dgrad_raw(ctx.cs, dh, _xf(), T, L.DG_MASK_LRELU, x_in=x, add=d_out, add_scale=s)
        return dx, dgb, (d_out if ctx.has_acc else None), None, None, None, None


FUSED_FILM_BLOCK = os.environ.get('TDVC_FUSED_FILM_BLOCK', '1') == '1'   # one-launch FiLM block forward at 16 channels (A/B switch)
FUSED_COND_FWD = os.environ.get('TDVC_FUSED_COND_FWD', '0') == '1'     # single-launch conditioning forward (tdvc_film_cond_fwd)
SIGN_BIT_MASKS = os.environ.get('TDVC_SIGN_BIT_MASKS', '1') == '1'     # cond_var.2 input-grad reads 1-bit LeakyReLU masks (A/B switch)
FUSED_COND_BWD = os.environ.get('TDVC_FUSED_COND_BWD', '1') == '1'     # cond_var.2 input-grad + cond_var.0 backward in one launch (tdvc_film_cond_bwd)


FUSED_COND_FWD_X6 = os.environ.get('TDVC_FUSED_COND_FWD_X6', '1') == '1'      # cond_var.0's window computed inside the split-bf16 cond_var.2 forward
FUSED_COND_FWD_X6_ALWAYS = os.environ.get('TDVC_FUSED_COND_FWD_X6_ALWAYS', '0') == '1'      # tests / tools / A-B: also where the two launches measure faster stand-alone


def _film_cond_fwd_x6(ctx, exc, k3, spec_var, spec2):
    """One launch for the conditioning forward (tdvc_film_cond_fwd_x6): returns gb and leaves cv0 / sign bits on ctx, or None when the
    shape is outside the kernel's contract (the caller then runs the two launches)."""
    B, nv, T = exc.shape
    nc, C2 = spec2.cin, spec2.cout
    if not (nv == 8 and nc % 4 == 0 and 64 < nc <= 160 and C2 % 32 == 0 and C2 >= X6_FWD_MIN_COUT and T >= 128 and T % 4 == 0 and spec2.k == 3):
        return None
    if C2 >= 128 and T >= 2048 and not FUSED_COND_FWD_X6_ALWAYS:
        # two or more output-channel blocks per tile each recompute the cv0 tile: at long sequences the two launches are faster
        # (tools/bench_cond_fwd_x6.py: 131 vs 122 us at 136 -> 128, T = 4000; 229 vs 248 / 138 vs 148 / 35 vs 43 us at the other stages)
        return None
    lib = L.lib()
    keep = any(ctx.needs_input_grad)      # inference / no_grad: the intermediate is not stored at all
    cv0 = torch.empty((B, nc, T), dtype=torch.float32, device=exc.device) if keep else None
    gb = torch.empty((B, C2, T), dtype=torch.float32, device=exc.device)
    bits = torch.empty((B, nc, T // 32), dtype=torch.int32, device=exc.device) if (keep and SIGN_BIT_MASKS and T % 32 == 0 and T >= 512) else None
    a = L.FilmCondArgs(B, T, nc, nv, C2, exc.data_ptr(), _bs(exc), spec_var.slot.w, k3.data_ptr(), spec2.slot.w, spec2.slot.b or None,
                       cv0.data_ptr() if keep else None, _bs(cv0) if keep else 0, gb.data_ptr(), _bs(gb), SLOPE)
    rc = lib.tdvc_film_cond_fwd_x6(C.byref(a), _weight_planes_x6(spec2, exc.device).data_ptr(), bits.data_ptr() if bits is not None else None,
                                   _bs(bits) if bits is not None else 0, _stream(exc))
    if rc == L.EUNSUPPORTED:
        return None
    L.check(rc)
    if RECORDER is not None:
        RECORDER.append(('film_cond_fwd_x6', B, T, nc, nv, C2, bits is not None))
    ctx.cv0_tmp, ctx.bits_tmp = cv0, bits
    return gb


class FilmCondFn(Function):
    """gb = cond_var.2(LeakyReLU(cond_var.0([emb; exc]))) (model/generator.py:86-92,103) in one forward launch:
    the time-constant embedding part enters as k3 [B,nc,3], the excitation part of cond_var.0 is an MFMA pre-pass
    that fills cond_var.2's LDS input tile directly. The nc-channel intermediate is written once for the backward
    (which is the unfused chain) and not re-read by the forward."""

    @staticmethod
    def forward(ctx, exc, k3, token, spec_var, spec2):
        exc, k3 = exc.contiguous(), k3.contiguous()
        B, nv, T = exc.shape
        nc = spec2.cin
        bits = None
        if FUSED_COND_FWD:
            cv0 = torch.empty((B, nc, T), dtype=torch.float32, device=exc.device)
            gb = torch.empty((B, spec2.cout, T), dtype=torch.float32, device=exc.device)
            a = L.FilmCondArgs(B, T, nc, nv, spec2.cout, exc.data_ptr(), _bs(exc), spec_var.slot.w, k3.data_ptr(),
                               spec2.slot.w, spec2.slot.b or None, cv0.data_ptr(), _bs(cv0), gb.data_ptr(), _bs(gb), SLOPE)
            L.check(L.lib().tdvc_film_cond_fwd(C.byref(a), _stream(exc)))
        elif FUSED_COND_FWD_X6 and X6_FWD and (gb := _film_cond_fwd_x6(ctx, exc, k3, spec_var, spec2)) is not None:
            cv0, bits = ctx.cv0_tmp, ctx.bits_tmp
            del ctx.cv0_tmp, ctx.bits_tmp
        else:
            # two launches: the 8-channel excitation window of cond_var.0 (HBM-bound, writes the 136-channel intermediate
            # once) and cond_var.2 on it with LeakyReLU-on-load. Measured faster than the single fused launch at every
            # decoder stage once the plain kernel runs 3 blocks per CU (tools/tile_sweep.py).
            # the input-grad of cond_var.2 only needs the SIGN of cv0 (LeakyReLU mask): one bit per element, written by the
            # launch that produces cv0, instead of re-reading the 136-channel fp32 tensor (the largest of the step)
            if SIGN_BIT_MASKS and T % 32 == 0 and T >= 512:
                bits = torch.empty((B, nc, T // 32), dtype=torch.int32, device=exc.device)
            cv0 = conv_fwd_raw(spec_var, exc, _xf(), bias3=k3, sign_bits=bits)
            gb = conv_fwd_raw(spec2, cv0, _xf(L.XF_LRELU))
        ctx.sv, ctx.s2 = spec_var, spec2
        ctx.bits = bits
        ctx.save_for_backward(exc, cv0)
        return gb

    @staticmethod
    def backward(ctx, dgb):
        exc, cv0 = ctx.saved_tensors
        dgb = dgb.contiguous()
        B, nc, T = cv0.shape
        conv_wgrad_raw(ctx.s2, cv0, _xf(L.XF_LRELU), dgb, _xf())
        lib = L.lib()
        sv = ctx.sv.slot
        nv = exc.shape[1]
        want_w = sv.trainable and (sv.arena is None or sv.arena.wgrad_enabled)
        dexc = torch.empty_like(exc) if ctx.needs_input_grad[0] else None
        dk3 = torch.empty((B, nc, 3), dtype=torch.float32, device=dgb.device)
        C2 = ctx.s2.cout
        if FUSED_COND_BWD and ctx.s2.slot.wt and nv == 8 and C2 % 32 == 0 and T % 4 == 0 and nc <= 144 and nc % 4 == 0:
            # one launch: the 136-channel gradient of cond_var.0's output lives in LDS / registers only (film_cond_fused_bwd.hip)
            nbytes = lib.tdvc_film_cond_bwd_workspace(B, T, nc, nv) if want_w else 0
            ws, off = workspace(dgb.device, nbytes) if nbytes else (None, 0)
            bits = ctx.bits
            a = L.FilmCondBwdArgs(B, T, nc, nv, C2, dgb.data_ptr(), _bs(dgb), ctx.s2.slot.wt,
                                  bits.data_ptr() if bits is not None else None, _bs(bits) if bits is not None else 0,
                                  cv0.data_ptr(), _bs(cv0), exc.data_ptr(), _bs(exc), sv.w,
                                  dexc.data_ptr() if dexc is not None else None, _bs(dexc) if dexc is not None else 0,
                                  dk3.data_ptr(), sv.dw if want_w else None,
                                  ws.data_ptr() + off if ws is not None else None, nbytes if ws is not None else 0, SLOPE)
            rc = lib.tdvc_film_cond_bwd(C.byref(a), _stream(dgb))
            if rc != L.EUNSUPPORTED:
                L.check(rc)
                if RECORDER is not None:
                    RECORDER.append(('film_cond_bwd', B, T, nc, nv, C2, bits is not None, dexc is not None, want_w))
                if want_w and sv.arena is not None:
                    sv.arena.note_grad(sv)
                elif want_w:
                    fold_flush(dgb.device)
                return dexc, dk3, None, None, None
        dcv = conv_dgrad_raw(ctx.s2, dgb, _xf(), T, L.DG_MASK_LRELU, x_in=cv0, x_bits=ctx.bits)
        # everything that consumes d_cv0 in one pass over it: dexc, the excitation window of cond_var.0's weight-grad, dk3
        nbytes = lib.tdvc_film_cond0_bwd_workspace(B, T, nc, nv) if want_w else 0
        ws, off = workspace(dgb.device, nbytes) if nbytes else (None, 0)
        a = L.FilmCond0BwdArgs(B, T, nc, nv, dcv.data_ptr(), _bs(dcv), exc.data_ptr(), _bs(exc), sv.w,
                               dexc.data_ptr() if dexc is not None else None, _bs(dexc) if dexc is not None else 0,
                               dk3.data_ptr(), sv.dw if want_w else None,
                               ws.data_ptr() + off if ws is not None else None, nbytes if ws is not None else 0)
        L.check(lib.tdvc_film_cond0_bwd(C.byref(a), _stream(dgb)))
        if RECORDER is not None:
            RECORDER.append(('film_cond0_bwd', B, T, nc, nv, dexc is not None, want_w))
        if want_w and sv.arena is not None:
            sv.arena.note_grad(sv)
        elif want_w:
            fold_flush(dgb.device)
        return dexc, dk3, None, None, None


class FilmK3Fn(Function):
    """k3 [B, nc, 3]: cond_var.0 on the time-constant speaker-embedding channels, as a length-3 constant signal
    (model/generator.py:86-92 with the exact split of SURVEY §2.2). emb: [B, n_const]."""

    @staticmethod
    def forward(ctx, emb, token, spec_const):
        emb = emb.contiguous()
        B, n_const = emb.shape
        nc = spec_const.cout
        s = spec_const.slot
        k3 = torch.empty((B, nc, 3), dtype=torch.float32, device=emb.device)
        L.check(L.lib().tdvc_film_k3_fwd(emb.data_ptr(), emb.stride(0), s.w, s.b or None, k3.data_ptr(), B, n_const, nc, _stream(emb)))
        ctx.spec = spec_const
        ctx.save_for_backward(emb)
        return k3

    @staticmethod
    def backward(ctx, dk3):
        (emb,) = ctx.saved_tensors
        dk3 = dk3.contiguous()
        B, n_const = emb.shape
        s = ctx.spec.slot
        want_w = s.trainable and (s.arena is None or s.arena.wgrad_enabled)
        demb = torch.empty_like(emb) if ctx.needs_input_grad[0] else None
        L.check(L.lib().tdvc_film_k3_bwd(dk3.data_ptr(), emb.data_ptr(), emb.stride(0), s.w, demb.data_ptr() if demb is not None else None,
                                         s.dw if want_w else None, (s.db or None) if want_w else None, B, n_const, ctx.spec.cout, _stream(dk3)))
        if want_w and s.arena is not None:
            s.arena.note_grad(s)
        return demb, None, None


def film_k3(emb, spec_const):
    return FilmK3Fn.apply(emb, _token(spec_const), spec_const)


class FilmK3MultiFn(Function):
    """FilmK3Fn for all FiLM blocks of one MRF stage at once (they read the same embedding): one launch forward, one backward
    (tdvc_film_k3_multi_*), and the embedding gradient comes back already summed over the blocks."""

    @staticmethod
    def forward(ctx, emb, token, *specs):
        emb = emb.contiguous()
        B, n_const = emb.shape
        nc = specs[0].cout
        n = len(specs)
        k3s = [torch.empty((B, nc, 3), dtype=torch.float32, device=emb.device) for _ in specs]
        w0 = (C.c_void_p * n)(*[sp.slot.w for sp in specs])
        b0 = (C.c_void_p * n)(*[(sp.slot.b or None) for sp in specs])
        out = (C.c_void_p * n)(*[t.data_ptr() for t in k3s])
        L.check(L.lib().tdvc_film_k3_multi_fwd(emb.data_ptr(), emb.stride(0), w0, b0, out, n, B, n_const, nc, _stream(emb)))
        ctx.specs = specs
        ctx.save_for_backward(emb)
        return tuple(k3s)

    @staticmethod
    def backward(ctx, *dk3s):
        (emb,) = ctx.saved_tensors
        specs = ctx.specs
        B, n_const = emb.shape
        n = len(specs)
        dk = [(g.contiguous() if g is not None else torch.zeros((B, specs[0].cout, 3), dtype=torch.float32, device=emb.device)) for g in dk3s]
        want = [sp.slot.trainable and (sp.slot.arena is None or sp.slot.arena.wgrad_enabled) for sp in specs]
        demb = torch.empty_like(emb) if ctx.needs_input_grad[0] else None
        pd = (C.c_void_p * n)(*[t.data_ptr() for t in dk])
        w0 = (C.c_void_p * n)(*[sp.slot.w for sp in specs])
        dw = (C.c_void_p * n)(*[(sp.slot.dw if w_ else None) for sp, w_ in zip(specs, want)])
        db = (C.c_void_p * n)(*[((sp.slot.db or None) if w_ else None) for sp, w_ in zip(specs, want)])
        L.check(L.lib().tdvc_film_k3_multi_bwd(pd, emb.data_ptr(), emb.stride(0), w0, demb.data_ptr() if demb is not None else None, dw, db,
                                               n, B, n_const, specs[0].cout, _stream(emb)))
        for sp, w_ in zip(specs, want):
            if w_ and sp.slot.arena is not None:
                sp.slot.arena.note_grad(sp.slot)
        return (demb, None) + (None,) * n


def film_k3_multi(emb, specs):
    """k3 of every spec in `specs` (<= 16, same shapes) from one embedding; returns a tuple of [B, nc, 3] tensors."""
    return FilmK3MultiFn.apply(emb, _token(*specs), *specs)


def film_cond(exc, k3, spec_var, spec2):
    return FilmCondFn.apply(exc, k3, _token(spec_var, spec2), spec_var, spec2)


def film_block(x, gb, acc, conv_spec, pos_spec, scale):
    return FilmBlockFn.apply(x, gb, acc, _token(conv_spec, pos_spec), conv_spec, pos_spec, scale)


# ------------------------------------------------------------------------------- small ops
class FanOutFn(Function):
    """n aliases of x for n consumers; the backward sums the n incoming gradients in ONE pass (tdvc_sum_n: reads n, writes 1)
    instead of autograd's chain of n - 1 two-operand adds. Same arithmetic as the reference's implicit gradient
    accumulation up to the order of the fp32 additions (left to right here, as autograd's chain)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g.contiguous() for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        out = torch.empty_like(gs[0])
        done = None
        for i in range(0, len(gs), 15):      # 16 sources per launch: the running sum + 15 new ones
            part = ([done] if done is not None else []) + gs[i:i + 15]
            ptrs = (C.c_void_p * len(part))(*[t.data_ptr() for t in part])
            L.check(L.lib().tdvc_sum_n(ptrs, len(part), out.data_ptr(), out.numel(), _stream(out)))
            done = out
        return out, None


def fanout(x, n):
    """x for n consumers (see FanOutFn); plain aliases when no gradient flows."""
    if n <= 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    return FanOutFn.apply(x, n)


class L2NormFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, Cc, T = x.shape
        y = torch.empty_like(x)
        inv = torch.empty((B, T), dtype=torch.float32, device=x.device)
        L.check(L.lib().tdvc_l2norm_fwd(x.data_ptr(), y.data_ptr(), inv.data_ptr(), B, Cc, T, 1e-12, _stream(x)))
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dy = dy.contiguous()
        B, Cc, T = y.shape
        dx = torch.empty_like(y)
        L.check(L.lib().tdvc_l2norm_bwd(y.data_ptr(), inv.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, Cc, T, _stream(y)))
        return dx


class GatherChFn(Function):
    @staticmethod
    def forward(ctx, x, label):
        x = x.contiguous()
        B, Cc, T = x.shape
        y = torch.empty((B, 1, T), dtype=torch.float32, device=x.device)
        L.check(L.lib().tdvc_gather_ch_fwd(x.data_ptr(), label.data_ptr(), y.data_ptr(), B, Cc, T, _stream(x)))
        ctx.save_for_backward(label)
        ctx.C = Cc
        return y

    @staticmethod
    def backward(ctx, dy):
        (label,) = ctx.saved_tensors
        dy = dy.contiguous()
        B, _, T = dy.shape
        dx = torch.empty((B, ctx.C, T), dtype=torch.float32, device=dy.device)
        L.check(L.lib().tdvc_gather_ch_bwd(dy.data_ptr(), label.data_ptr(), dx.data_ptr(), B, ctx.C, T, _stream(dy)))
        return dx, None


class ConcatCondFn(Function):
    """c = cat([emb.unsqueeze(2).repeat(1,1,T), exc], dim=1) (model/generator.py:387-399)."""

    @staticmethod
    def forward(ctx, emb, exc):
        emb, exc = emb.contiguous(), exc.contiguous()
        B, Ce = emb.shape
        _, Cx, T = exc.shape
        c = torch.empty((B, Ce + Cx, T), dtype=torch.float32, device=exc.device)
        L.check(L.lib().tdvc_concat_cond(emb.data_ptr(), exc.data_ptr(), c.data_ptr(), B, Ce, Cx, T, _stream(exc)))
        ctx.dims = (B, Ce, Cx, T)
        return c

    @staticmethod
    def backward(ctx, dc):
        B, Ce, Cx, T = ctx.dims
        dc = dc.contiguous()
        demb = torch.empty((B, Ce), dtype=torch.float32, device=dc.device)
        dexc = torch.empty((B, Cx, T), dtype=torch.float32, device=dc.device)
        L.check(L.lib().tdvc_concat_cond_bwd(dc.data_ptr(), demb.data_ptr(), dexc.data_ptr(), B, Ce, Cx, T, 0, _stream(dc)))
        return demb, dexc


class CinFn(Function):
    """(1+gamma) * InstanceNorm(x) + beta, gb = [gamma; beta] as [B,2C,1] or [B,2C,T]."""

    @staticmethod
    def forward(ctx, x, gb, eps):
        x, gb = x.contiguous(), gb.contiguous()
        B, Cc, T = x.shape
        Tg = gb.shape[2]
        y = torch.empty_like(x)
        mean = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        L.check(L.lib().tdvc_cin_fwd(x.data_ptr(), gb.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                     B, Cc, T, Tg, eps, _stream(x)))
        ctx.save_for_backward(x, gb, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gb, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        B, Cc, T = x.shape
        dx, dgb = torch.empty_like(x), torch.empty_like(gb)
        L.check(L.lib().tdvc_cin_bwd(x.data_ptr(), gb.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                     dx.data_ptr(), dgb.data_ptr(), B, Cc, T, gb.shape[2], _stream(x)))
        return dx, dgb, None


def axpby(a, b, alpha, beta):
    y = torch.empty_like(a)
    L.check(L.lib().tdvc_axpby(a.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(), alpha, beta,
                               a.numel(), _stream(a)))
    return y


# ------------------------------------------------------------------------------- SSL encoder: WaveNet gated stack
def _gate_fwd(xin, H):
    B, _, T = xin.shape
    acts = torch.empty((B, H, T), dtype=torch.float32, device=xin.device)
    L.check(L.lib().tdvc_gate_fwd(xin.data_ptr(), _bs(xin), None, 0, acts.data_ptr(), _bs(acts), B, H, T, _stream(xin)))
    return acts


def _gate_bwd(xin, dacts, H):
    B, _, T = xin.shape
    dxin = torch.empty_like(xin)
    L.check(L.lib().tdvc_gate_bwd(xin.data_ptr(), _bs(xin), None, 0, dacts.data_ptr(), _bs(dacts), dxin.data_ptr(), _bs(dxin), B, H, T, _stream(xin)))
    return dxin


class WnStackFn(Function):
    """The gated WaveNet stack of the SSL content encoder (model/ssl_encoder.py:52-82, WN.forward with g=None,
    x_mask=1, dropout 0). Per layer i:
        x_in = in_i(x);  acts = tanh(x_in[:H]) * sigmoid(x_in[H:]);  rs = res_skip_i(acts)
        i < n-1:  x += rs[:H], out += rs[H:]          last layer: out += rs  (H channels)
    The running pair lives in ONE state tensor S = [x ; out] ([B, 2H, T]), so res_skip's residual epilogue updates both
    halves in a single launch; the in-layer reads the x half through its batch stride. Three launches per layer forward
    (conv, gate, conv); backward = the convs' input/weight-grad kernels plus the gate's backward."""

    @staticmethod
    def forward(ctx, x, token, in_specs, rs_specs):
        x = x.contiguous()
        B, H, T = x.shape
        n = len(in_specs)
        S = torch.zeros((B, 2 * H, T), dtype=torch.float32, device=x.device)
        S[:, :H].copy_(x)
        states, xins, actss = [], [], []
        for i in range(n):
            states.append(S)
            xin = conv_fwd_raw(in_specs[i], S[:, :H], _xf())
            acts = _gate_fwd(xin, H)
            xins.append(xin); actss.append(acts)
            if i < n - 1:
                S = conv_fwd_raw(rs_specs[i], acts, _xf(), res=S)
            else:
                out = conv_fwd_raw(rs_specs[i], acts, _xf(), res=S[:, H:])
        ctx.in_specs, ctx.rs_specs, ctx.H = in_specs, rs_specs, H
        ctx.save_for_backward(*states, *xins, *actss)
        return out

    @staticmethod
    def backward(ctx, d_out):
        n, H = len(ctx.in_specs), ctx.H
        saved = ctx.saved_tensors
        states, xins, actss = saved[:n], saved[n:2 * n], saved[2 * n:]
        d_out = d_out.contiguous()
        B, _, T = d_out.shape
        # dS = gradient wrt the state [x ; out] entering layer i+1; its `out` half is d_out for every layer
        bufs = [torch.empty((B, 2 * H, T), dtype=torch.float32, device=d_out.device) for _ in range(2)]
        for b in bufs:
            b[:, H:].copy_(d_out)
        dS = None
        for i in reversed(range(n)):
            dy = d_out if i == n - 1 else dS
            conv_wgrad_raw(ctx.rs_specs[i], actss[i], _xf(), dy, _xf())
            d_acts = conv_dgrad_raw(ctx.rs_specs[i], dy, _xf(), T, L.DG_PLAIN)
            d_xin = _gate_bwd(xins[i], d_acts, H)
            x_i = states[i][:, :H]
            conv_wgrad_raw(ctx.in_specs[i], x_i, _xf(), d_xin, _xf())
            nxt = bufs[i & 1]
            if i == n - 1:      # the last layer leaves x untouched: no residual-path gradient
                conv_dgrad_raw(ctx.in_specs[i], d_xin, _xf(), T, L.DG_PLAIN, out=nxt[:, :H])
            else:
                conv_dgrad_raw(ctx.in_specs[i], d_xin, _xf(), T, L.DG_PLAIN, add=dS[:, :H], add_scale=1.0, out=nxt[:, :H])
            dS = nxt
        dx = dS[:, :H].contiguous() if ctx.needs_input_grad[0] else None
        return dx, None, None, None


def wn_stack(x, in_specs, rs_specs):
    return WnStackFn.apply(x, _token(*in_specs, *rs_specs), list(in_specs), list(rs_specs))
