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


class BatchView:
    """Samples [off, off + n) of a batch-major contiguous tensor `t`, WITHOUT materialising the slice.

    The train step pushes several signals through the discriminator in one call (real | fake, fake | identity |
    reconstruction, and inside discriminators 1 / 2 the filtered input | the sub-scale input on top). Every loss term
    reads a sample range of those batched outputs / feature maps. Slicing them as tensors makes autograd run, for each
    of the ~30 maps of up to 33 MB, a zero-fill of a full-size gradient, a copy of the slice gradient into it and an add
    of the partial gradients (5 % of the iteration). The loss functions below take views instead and write ONE
    full-size gradient per batched tensor: the kernels address the range through a pointer offset."""
    __slots__ = ('t', 'off', 'n')

    def __init__(self, t, off=0, n=None):
        self.t, self.off, self.n = t, off, (t.shape[0] - off) if n is None else n

    def rows(self, lo, hi):
        assert 0 <= lo <= hi <= self.n
        return BatchView(self.t, self.off + lo, hi - lo)

    def tensor(self):
        return self.t[self.off:self.off + self.n]

    @property
    def shape(self):
        return (self.n, *self.t.shape[1:])

    def detach(self):
        return BatchView(self.t.detach(), self.off, self.n)


def as_view(x):
    return x if isinstance(x, BatchView) else BatchView(x)


def _group(views):
    """Unique underlying tensors of `views` (contiguous) and, per view, the index of its tensor."""
    tensors, index, which = [], {}, []
    for v in views:
        k = id(v.t)
        if k not in index:
            index[k] = len(tensors)
            tensors.append(v.t.contiguous())
        which.append(index[k])
    return tensors, which


def _fill_gaps(dx, ranges, lib):
    """Zero the rows of dx that none of the (off, n) ranges covers."""
    per = dx[0].numel()
    st = _stream(dx)
    pos = 0
    for off, n in sorted(ranges):
        if off > pos:
            L.check(lib.tdvc_fill(dx.data_ptr() + 4 * pos * per, 0.0, (off - pos) * per, st))
        pos = max(pos, off + n)
    if pos < dx.shape[0]:
        L.check(lib.tdvc_fill(dx.data_ptr() + 4 * pos * per, 0.0, (dx.shape[0] - pos) * per, st))


class MseViewsFn(Function):
    """LSGAN terms on views of batched discriminator outputs: spec = [(tensor index, off, n, target, k)]; returns
    `nout` scalars, scalar k = sum over its entries of mean((x[off:off+n] - target)^2). One gradient per tensor."""

    @staticmethod
    def forward(ctx, spec, nout, *xs):
        outs = [_ScalarOut.new(xs[0]) for _ in range(nout)]
        lib = L.lib()
        for ti, off, n, target, k in spec:
            x = xs[ti]
            per = x[0].numel()
            L.check(lib.tdvc_mse_const_fwd(x.data_ptr() + 4 * off * per, n * per, target, 1.0, outs[k].data_ptr(), _stream(x)))
        ctx.spec = spec
        ctx.save_for_backward(*xs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        xs = ctx.saved_tensors
        lib = L.lib()
        zero = torch.zeros(1, dtype=torch.float32, device=xs[0].device) if any(g is None for g in gs) else None
        gs = [g.contiguous() if g is not None else zero for g in gs]
        grads = []
        for ti, x in enumerate(xs):
            mine = [e for e in ctx.spec if e[0] == ti]
            dx = torch.empty_like(x)
            _fill_gaps(dx, [(off, n) for _, off, n, _, _ in mine], lib)
            per = x[0].numel()
            for _, off, n, target, k in mine:
                L.check(lib.tdvc_mse_const_bwd(x.data_ptr() + 4 * off * per, n * per, target, 1.0, gs[k].data_ptr(),
                                               dx.data_ptr() + 4 * off * per, _stream(x)))
            grads.append(dx)
        return (None, None, *grads)


def lsgan_loss(outs, target, rng=None):
    """sum_i mean((o_i - target)^2) (train.py:273-279, 327-331). `outs` may hold BatchViews; rng = (lo, hi) restricts every
    output to samples [lo, hi) of its view (the outputs are batched over several signals)."""
    views = [as_view(o) for o in outs]
    if rng is not None:
        views = [v.rows(int(rng[0]), int(rng[1])) for v in views]
    tensors, which = _group(views)
    spec = [(ti, v.off, v.n, float(target), 0) for ti, v in zip(which, views)]
    return MseViewsFn.apply(spec, 1, *tensors)[0]


def lsgan_split(outs, B):
    """D-step pair (train.py:271-281) on outputs of ONE discriminator call over [real; fake]:
    (sum_i mean((o_i[:B] - 1)^2), sum_i mean(o_i[B:]^2))."""
    views = [as_view(o) for o in outs]
    tensors, which = _group(views)
    spec = []
    for ti, v in zip(which, views):
        spec.append((ti, v.off, B, 1.0, 0))
        spec.append((ti, v.off + B, v.n - B, 0.0, 1))
    return MseViewsFn.apply(spec, 2, *tensors)


class L1ViewsFn(Function):
    """sum over pairs of mean|a[off:off+n] - b[roff:roff+n]| on views of batched feature maps (b carries no gradient):
    spec = [(a tensor index, off, n, b tensor index, roff)]. One full-size gradient per batched `a` tensor."""

    @staticmethod
    def forward(ctx, spec, n_a, *ts):
        a, b = ts[:n_a], ts[n_a:]
        out = _ScalarOut.new(a[0])
        pairs = (L.L1Pair * len(spec))()
        for k, (ai, off, n, bi, roff) in enumerate(spec):
            x, y = a[ai], b[bi]
            if x.shape[1:] != y.shape[1:]:
                raise RuntimeError(f'l1 pair shape mismatch {tuple(x.shape[1:])} vs {tuple(y.shape[1:])}')
            per = x[0].numel()
            pairs[k] = L.L1Pair(x.data_ptr() + 4 * off * per, y.data_ptr() + 4 * roff * per, None, n * per, 1.0)
        L.check(L.lib().tdvc_l1_multi_fwd(pairs, len(spec), out.data_ptr(), _stream(a[0])))     # every term in one launch
        ctx.spec, ctx.n_a = spec, n_a
        ctx.save_for_backward(*ts)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        a, b = ctx.saved_tensors[:ctx.n_a], ctx.saved_tensors[ctx.n_a:]
        grads, ents = [], []
        for ai, x in enumerate(a):
            mine = sorted((off, n, bi, roff) for i_, off, n, bi, roff in ctx.spec if i_ == ai)
            dx = torch.empty_like(x)
            per = x[0].numel()
            pos = 0
            for off, n, bi, roff in mine:      # the term's gradient on its samples, zeros on the samples no term reads
                if off > pos:
                    ents.append(L.L1Pair(None, None, dx.data_ptr() + 4 * pos * per, (off - pos) * per, 0.0))
                ents.append(L.L1Pair(x.data_ptr() + 4 * off * per, b[bi].data_ptr() + 4 * roff * per, dx.data_ptr() + 4 * off * per, n * per, 1.0))
                pos = max(pos, off + n)
            if pos < x.shape[0]:
                ents.append(L.L1Pair(None, None, dx.data_ptr() + 4 * pos * per, (x.shape[0] - pos) * per, 0.0))
            grads.append(dx)
        pairs = (L.L1Pair * len(ents))(*ents)
        L.check(L.lib().tdvc_l1_multi_bwd(pairs, len(ents), g.data_ptr(), _stream(g)))             # all maps in one launch
        return (None, None, *grads, *([None] * len(b)))


def multiscale_feat_loss(feat_sig_list, feat_ref_list, norm_p=1, rng=None):
    """util/losses.py:55-68: sum over the 5 passes x 6 maps of mean|sig - ref.detach()|. Maps may be BatchViews;
    rng = (lo, hi) restricts every `sig` map to samples [lo, hi) of its view (product-side extension, see BatchView)."""
    if norm_p != 1:
        raise NotImplementedError('norm_p=2 calls a non-existent F.rms_loss in the reference (Q13)')
    a = [as_view(m) for fl in feat_sig_list for m in fl]
    b = [as_view(m).detach() for fl in feat_ref_list for m in fl]
    if rng is not None:
        a = [v.rows(int(rng[0]), int(rng[1])) for v in a]
    for v, w in zip(a, b):
        if v.n != w.n:
            raise RuntimeError(f'l1 pair batch mismatch {v.n} vs {w.n}')
    ta, wa = _group(a)
    tb, wb = _group(b)
    spec = [(ia, v.off, v.n, ib, w.off) for ia, v, ib, w in zip(wa, a, wb, b)]
    return L1ViewsFn.apply(spec, len(ta), *ta, *tb)


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
