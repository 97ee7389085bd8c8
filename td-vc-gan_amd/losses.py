"""Loss terms of the train step on the HIP kernels; mirrors util/losses.py of the reference
(`multiscale_spec_loss` :33-53, `multiscale_feat_loss` :55-68, `contrastive_loss` :70-116) plus the
LSGAN terms written inline in train.py:271-281, 327-331.

Reference quirks reproduced on purpose: Q1 — `multiscale_spec_loss` only ever uses
fft_sizes[0]; Q2 — `contrastive_loss` ignores its `temp` argument (temperature 1).
"""
import math

import numpy as np
import torch
from torch.autograd import Function

from . import _lib as L
from . import ops
from .arena import ConvSlot
from .ops import ConvSpec, _stream


class _ScalarOut:
    @staticmethod
    def new(ref):
        return torch.zeros(1, dtype=torch.float32, device=ref.device)


class MseConstFn(Function):
    """sum_i mean((x_i - target)^2) over a list of tensors (the 5 discriminator outputs)."""

    @staticmethod
    def forward(ctx, target, *xs):
        xs = [x.contiguous() for x in xs]
        out = _ScalarOut.new(xs[0])
        lib = L.lib()
        for x in xs:
            L.check(lib.tdvc_mse_const_fwd(x.data_ptr(), x.numel(), target, 1.0, out.data_ptr(), _stream(x)))
        ctx.target = target
        ctx.save_for_backward(*xs)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        lib = L.lib()
        grads = []
        for x in ctx.saved_tensors:
            dx = torch.empty_like(x)
            L.check(lib.tdvc_mse_const_bwd(x.data_ptr(), x.numel(), ctx.target, 1.0, g.data_ptr(), dx.data_ptr(), _stream(x)))
            grads.append(dx)
        return (None, *grads)


class CrossEntropyFn(Function):
    """F.cross_entropy(logits [B, K], labels [B]) — mean over the batch (latent classifier, train.py:302, :422)."""

    @staticmethod
    def forward(ctx, logits, labels):
        logits = logits.contiguous()
        labels = labels.contiguous().long()
        B, K = logits.shape
        out = _ScalarOut.new(logits)
        prob = torch.empty_like(logits)
        L.check(L.lib().tdvc_cross_entropy_fwd(logits.data_ptr(), labels.data_ptr(), B, K, 1.0, out.data_ptr(), prob.data_ptr(), _stream(logits)))
        ctx.save_for_backward(prob, labels)
        return out

    @staticmethod
    def backward(ctx, g):
        prob, labels = ctx.saved_tensors
        B, K = prob.shape
        d = torch.empty_like(prob)
        L.check(L.lib().tdvc_cross_entropy_bwd(prob.data_ptr(), labels.data_ptr(), B, K, 1.0, g.contiguous().data_ptr(), d.data_ptr(), _stream(prob)))
        return d, None


def cross_entropy_loss(logits, labels):
    return CrossEntropyFn.apply(logits, labels)


def _rows(x, lo, hi):
    """(byte offset, element count) of samples [lo, hi) of a batch-major contiguous tensor."""
    per = x[0].numel()
    return 4 * lo * per, (hi - lo) * per


def _zero_outside(dx, lo, hi, lib):
    """Zero the rows of dx outside [lo, hi) (the loss does not see them)."""
    N = dx.shape[0]
    st = _stream(dx)
    if lo > 0:
        off, n = _rows(dx, 0, lo)
        L.check(lib.tdvc_fill(dx.data_ptr() + off, 0.0, n, st))
    if hi < N:
        off, n = _rows(dx, hi, N)
        L.check(lib.tdvc_fill(dx.data_ptr() + off, 0.0, n, st))


class MseConstRangeFn(Function):
    """sum_i mean((x_i[lo:hi] - target)^2): the LSGAN term on a sample range of BATCHED discriminator outputs. The
    gradient is produced directly at full batch size (zero outside the range), so no slice copy / zero-pad / add kernels
    appear in the backward pass."""

    @staticmethod
    def forward(ctx, target, lo, hi, *xs):
        xs = [x.contiguous() for x in xs]
        out = _ScalarOut.new(xs[0])
        lib = L.lib()
        for x in xs:
            off, n = _rows(x, lo, hi)
            L.check(lib.tdvc_mse_const_fwd(x.data_ptr() + off, n, target, 1.0, out.data_ptr(), _stream(x)))
        ctx.args = (target, lo, hi)
        ctx.save_for_backward(*xs)
        return out

    @staticmethod
    def backward(ctx, g):
        target, lo, hi = ctx.args
        g = g.contiguous()
        lib = L.lib()
        grads = []
        for x in ctx.saved_tensors:
            dx = torch.empty_like(x)
            _zero_outside(dx, lo, hi, lib)
            off, n = _rows(x, lo, hi)
            L.check(lib.tdvc_mse_const_bwd(x.data_ptr() + off, n, target, 1.0, g.data_ptr(), dx.data_ptr() + off, _stream(x)))
            grads.append(dx)
        return (None, None, None, *grads)


class LsganSplitFn(Function):
    """D-step losses on outputs of ONE discriminator call over [real; fake] (train.py:271-281): returns
    (sum_i mean((x_i[:B] - 1)^2), sum_i mean(x_i[B:]^2)) and writes both gradient halves into one tensor per output."""

    @staticmethod
    def forward(ctx, B, *xs):
        xs = [x.contiguous() for x in xs]
        l_real, l_fake = _ScalarOut.new(xs[0]), _ScalarOut.new(xs[0])
        lib = L.lib()
        for x in xs:
            N = x.shape[0]
            o0, n0 = _rows(x, 0, B)
            o1, n1 = _rows(x, B, N)
            L.check(lib.tdvc_mse_const_fwd(x.data_ptr() + o0, n0, 1.0, 1.0, l_real.data_ptr(), _stream(x)))
            L.check(lib.tdvc_mse_const_fwd(x.data_ptr() + o1, n1, 0.0, 1.0, l_fake.data_ptr(), _stream(x)))
        ctx.B = B
        ctx.save_for_backward(*xs)
        return l_real, l_fake

    @staticmethod
    def backward(ctx, g_real, g_fake):
        lib = L.lib()
        xs = ctx.saved_tensors
        zero = None
        if g_real is None or g_fake is None:
            zero = torch.zeros(1, dtype=torch.float32, device=xs[0].device)
        g_real = g_real.contiguous() if g_real is not None else zero
        g_fake = g_fake.contiguous() if g_fake is not None else zero
        grads = []
        for x in xs:
            dx = torch.empty_like(x)
            o0, n0 = _rows(x, 0, ctx.B)
            o1, n1 = _rows(x, ctx.B, x.shape[0])
            L.check(lib.tdvc_mse_const_bwd(x.data_ptr() + o0, n0, 1.0, 1.0, g_real.data_ptr(), dx.data_ptr() + o0, _stream(x)))
            L.check(lib.tdvc_mse_const_bwd(x.data_ptr() + o1, n1, 0.0, 1.0, g_fake.data_ptr(), dx.data_ptr() + o1, _stream(x)))
            grads.append(dx)
        return (None, *grads)


def lsgan_loss(outs, target, rng=None):
    """rng = (lo, hi): the outputs are batched over several signals and only samples [lo, hi) enter this term."""
    if rng is None:
        return MseConstFn.apply(float(target), *outs)
    return MseConstRangeFn.apply(float(target), int(rng[0]), int(rng[1]), *outs)


def lsgan_split(outs, B):
    """(real term on samples [0, B), fake term on samples [B, N)) of one batched discriminator call."""
    return LsganSplitFn.apply(int(B), *outs)


class L1PairsFn(Function):
    """sum over pairs of mean|a[lo:hi] - b| (b carries no gradient). With a sample range the a's are BATCHED feature
    maps (several signals through the discriminator in one call); their gradient comes out at full batch size, zero
    outside the range — the slice / zero-pad / copy kernels autograd would otherwise run on 30 feature maps of up to
    33 MB each were 4 % of the iteration."""

    @staticmethod
    def forward(ctx, n, lo, hi, *ab):
        a = [t.contiguous() for t in ab[:n]]
        b = [t.contiguous() for t in ab[n:]]
        out = _ScalarOut.new(a[0])
        lib = L.lib()
        for x, y in zip(a, b):
            h = x.shape[0] if hi is None else hi
            if x[lo:h].shape != y.shape:
                raise RuntimeError(f'l1 pair shape mismatch {tuple(x[lo:h].shape)} vs {tuple(y.shape)}')
            off, cnt = _rows(x, lo, h)
            L.check(lib.tdvc_l1_fwd(x.data_ptr() + off, y.data_ptr(), cnt, 1.0, out.data_ptr(), _stream(x)))
        ctx.n, ctx.lo, ctx.hi = n, lo, hi
        ctx.save_for_backward(*a, *b)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        n = ctx.n
        a, b = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        lib = L.lib()
        grads = []
        for x, y in zip(a, b):
            h = x.shape[0] if ctx.hi is None else ctx.hi
            dx = torch.empty_like(x)
            _zero_outside(dx, ctx.lo, h, lib)
            off, cnt = _rows(x, ctx.lo, h)
            L.check(lib.tdvc_l1_bwd(x.data_ptr() + off, y.data_ptr(), cnt, 1.0, g.data_ptr(), dx.data_ptr() + off, 0, _stream(x)))
            grads.append(dx)
        return (None, None, None, *grads, *([None] * n))


def multiscale_feat_loss(feat_sig_list, feat_ref_list, norm_p=1, rng=None):
    """util/losses.py:55-68. rng = (lo, hi): `feat_sig_list` holds feature maps of a discriminator call batched over
    several signals and only samples [lo, hi) are the signal of this term (product-side extension, see L1PairsFn)."""
    if norm_p != 1:
        raise NotImplementedError('norm_p=2 calls a non-existent F.rms_loss in the reference (Q13)')
    a = [m for fl in feat_sig_list for m in fl]
    b = [m.detach() for fl in feat_ref_list for m in fl]
    lo, hi = (0, None) if rng is None else (int(rng[0]), int(rng[1]))
    return L1PairsFn.apply(len(a), lo, hi, *a, *b)


# ------------------------------------------------------------------------------- log-mel
class _ReflectPadFn(Function):
    @staticmethod
    def forward(ctx, x, pad):
        x = x.contiguous()
        B, _, T = x.shape
        y = torch.empty((B, 1, T + 2 * pad), dtype=torch.float32, device=x.device)
        L.check(L.lib().tdvc_reflect_pad_fwd(x.data_ptr(), y.data_ptr(), B, T, pad, _stream(x)))
        ctx.dims = (B, T, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, T, pad = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty((B, 1, T), dtype=torch.float32, device=dy.device)
        L.check(L.lib().tdvc_reflect_pad_bwd(dy.data_ptr(), dx.data_ptr(), B, T, pad, _stream(dy)))
        return dx, None


class _PowerFn(Function):
    @staticmethod
    def forward(ctx, spec):
        spec = spec.contiguous()
        B, F2, N = spec.shape
        pw = torch.empty((B, F2 // 2, N), dtype=torch.float32, device=spec.device)
        L.check(L.lib().tdvc_power_fwd(spec.data_ptr(), pw.data_ptr(), B, F2 // 2, N, _stream(spec)))
        ctx.save_for_backward(spec)
        return pw

    @staticmethod
    def backward(ctx, dpw):
        (spec,) = ctx.saved_tensors
        dpw = dpw.contiguous()
        B, F2, N = spec.shape
        ds = torch.empty_like(spec)
        L.check(L.lib().tdvc_power_bwd(spec.data_ptr(), dpw.data_ptr(), ds.data_ptr(), B, F2 // 2, N, _stream(spec)))
        return ds


class _LogL1Fn(Function):
    @staticmethod
    def forward(ctx, a, b, floor):
        a, b = a.contiguous(), b.contiguous()
        out = _ScalarOut.new(a)
        L.check(L.lib().tdvc_log_l1_fwd(a.data_ptr(), b.data_ptr(), a.numel(), floor, 1.0, out.data_ptr(), _stream(a)))
        ctx.floor = floor
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = torch.empty_like(a)
        L.check(L.lib().tdvc_log_l1_bwd(a.data_ptr(), b.data_ptr(), a.numel(), ctx.floor, 1.0, g.data_ptr(), da.data_ptr(), _stream(a)))
        return da, None, None


def _mel_filterbank(n_freqs, n_mels, sr):
    """[n_freqs, n_mels], HTK mel scale, Slaney area normalisation (torchaudio.functional.melscale_fbanks)."""
    hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    all_freqs = np.linspace(0, sr // 2, n_freqs)
    m_pts = np.linspace(hz2mel(0.0), hz2mel(float(sr // 2)), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    return fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels]))[None, :]


class MelSpec:
    """MelSpectrogram(sample_rate, n_fft, hop=n_fft//4, n_mels, norm='slaney') with torchaudio defaults
    (periodic Hann, centre reflect pad, power 2, HTK scale), as two convolutions on the HIP conv kernels:
    STFT = Conv1d(1 -> 2F, K=n_fft, stride=hop) whose weight is the windowed DFT basis, mel projection =
    1x1 Conv1d(F -> n_mels) whose weight is the filterbank. F = n_fft/2+1 padded to a multiple of 4."""

    def __init__(self, sr, n_fft, n_mels):
        self.n_fft, self.hop, self.n_mels = n_fft, n_fft // 4, n_mels
        F = n_fft // 2 + 1
        Fp = (F + 3) // 4 * 4
        n = np.arange(n_fft)
        win = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)
        ang = 2 * np.pi * np.outer(np.arange(F), n) / n_fft
        basis = np.zeros((2 * Fp, 1, n_fft), np.float64)
        basis[:F, 0] = np.cos(ang) * win
        basis[Fp:Fp + F, 0] = -np.sin(ang) * win
        fb = np.zeros((n_mels, Fp, 1), np.float64)
        fb[:, :F, 0] = _mel_filterbank(F, n_mels, sr).T
        self.basis = torch.from_numpy(basis.astype(np.float32))
        self.fb = torch.from_numpy(fb.astype(np.float32))
        self.stft_spec = ConvSpec(1, 2 * Fp, n_fft, stride=self.hop)
        self.mel_spec = ConvSpec(Fp, n_mels, 1)
        self.dev = None

    def _to(self, device):
        if self.dev != device:
            self.basis_d, self.fb_d = self.basis.to(device), self.fb.to(device)
            self.stft_spec.slot = ConvSlot(self.basis_d.data_ptr(), 0, 0, 0, trainable=False)
            self.mel_spec.slot = ConvSlot(self.fb_d.data_ptr(), 0, 0, 0, trainable=False)
            self.dev = device

    def __call__(self, x):
        """x [B,1,T] -> mel power [B,n_mels,1+T//hop]."""
        self._to(x.device)
        xp = _ReflectPadFn.apply(x, self.n_fft // 2)
        spec = ops.conv(xp, self.stft_spec)
        return ops.conv(_PowerFn.apply(spec), self.mel_spec)


_mel_cache = {}


def get_melspec_transform(sr, n_fft, n_mels, device=None):
    key = (sr, n_fft, n_mels)
    if key not in _mel_cache:
        _mel_cache[key] = MelSpec(sr, n_fft, n_mels)
    return _mel_cache[key]


def multiscale_spec_loss(signal, ref, fft_sizes, spectype='both', return_separated=False, norm_p=1, all_resolutions=False):
    """l1(log(clamp(mel(signal),1e-5)), log(clamp(mel(ref),1e-5))). Like the reference, only fft_sizes[0]
    contributes (the reference returns from inside its loop, Q1) unless all_resolutions=True."""
    if norm_p != 1:
        raise NotImplementedError('norm_p=2 calls a non-existent F.rms_loss in the reference (Q13)')
    sizes = list(fft_sizes) if all_resolutions else list(fft_sizes)[:1]
    losses = []
    for n_fft in sizes:
        t = get_melspec_transform(16000, n_fft, 80)
        with torch.no_grad():
            mref = t(ref)
        losses.append(_LogL1Fn.apply(t(signal), mref, 1e-5))
    total = losses[0]
    for l in losses[1:]:
        total = total + l
    if return_separated:
        return total, losses
    return total


# ------------------------------------------------------------------------------- contrastive
class _ContrastiveFn(Function):
    @staticmethod
    def forward(ctx, X, Y, idx_x, idx_y):
        X, Y = X.contiguous(), Y.contiguous()
        B, Cc, T = X.shape
        N = idx_x.shape[2]
        out = _ScalarOut.new(X)
        dX, dY = torch.zeros_like(X), torch.zeros_like(Y)
        L.check(L.lib().tdvc_contrastive_fwd_bwd(X.data_ptr(), Y.data_ptr(), idx_x.data_ptr(), idx_y.data_ptr(), B, Cc, T, N,
                                                 1.0, out.data_ptr(), dX.data_ptr(), dY.data_ptr(), _stream(X)))
        ctx.save_for_backward(dX, dY)
        return out

    @staticmethod
    def backward(ctx, g):
        dX, dY = ctx.saved_tensors
        return dX * g, dY * g, None, None


def sample_negative_indices(B, T, n_neg, device, generator=None):
    """The reference's draw (util/losses.py:79-83): randint(0, T-1) then skip self."""
    idx = torch.randint(0, T - 1, (B, T, n_neg), device=device, generator=generator)
    return idx


def _skip_self(idx):
    T = idx.shape[1]
    self_idx = torch.arange(T, device=idx.device).view(1, T, 1)
    return (idx + (idx >= self_idx).to(idx.dtype)).to(torch.int32).contiguous()


def contrastive_loss(sig_X, sig_Y, num_negatives=100, temp=1, idx_x=None, idx_y=None):
    """InfoNCE over time steps, both directions. `temp` is accepted and ignored, as in the reference (Q2).
    idx_x / idx_y ([B,T,N] draws in [0,T-1)) may be injected for reproducible parity tests."""
    B, _, T = sig_X.shape
    if idx_x is None:
        idx_x = sample_negative_indices(B, T, num_negatives, sig_X.device)
    if idx_y is None:
        idx_y = sample_negative_indices(B, T, num_negatives, sig_X.device)
    return _ContrastiveFn.apply(sig_X, sig_Y, _skip_self(idx_x.to(sig_X.device)), _skip_self(idx_y.to(sig_X.device)))
