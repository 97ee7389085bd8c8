"""ORACLE — test infrastructure only (see oracle/model.py header). CPU restatement of the
loss terms of the TD-VC-GAN train step.

Reference anchors:
  LSGAN terms            train.py:271-281, 327-331
  feature matching       util/losses.py:55-68
  log-mel L1             util/losses.py:28-53  (returns inside the loop: only fft_sizes[0])
  contrastive (InfoNCE)  util/losses.py:70-116 (temperature never forwarded: temp = 1)

PARITY UNPINNED for the log-mel term only: its arithmetic lives in torchaudio 2.1.0
(MelSpectrogram, requirements.txt:19), which is absent from the reference tree and from this
image. `mel_filterbank`/`log_mel` restate torchaudio's published algorithm (periodic Hann,
centre reflect pad, power spectrogram, HTK mel scale, Slaney area normalisation); the reference's
own call site (util/losses.py:30: sr 16000, n_fft, hop n_fft/4, 80 mels, norm='slaney') fixes the
parameters. Everything else in this file is pinned against the imported reference.
"""
import math

import torch
import torch.nn.functional as F


def lsgan_to_one(outs):
    return sum(((o - 1.0) ** 2).mean() for o in outs)


def lsgan_to_zero(outs):
    return sum((o ** 2).mean() for o in outs)


def feature_matching(feats_sig, feats_ref):
    tot = 0
    for fs, fr in zip(feats_sig, feats_ref):
        for a, b in zip(fs, fr):
            tot = tot + (a - b.detach()).abs().mean()
    return tot


def mel_filterbank(n_freqs, n_mels, sr, f_min=0.0, f_max=None, dtype=torch.float32):
    """[n_freqs, n_mels] triangular filters, HTK mel scale, Slaney normalisation."""
    f_max = float(sr // 2) if f_max is None else f_max
    hz2mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    all_freqs = torch.linspace(0, sr // 2, n_freqs, dtype=torch.float64)
    m_pts = torch.linspace(hz2mel(f_min), hz2mel(f_max), n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.minimum(down, up), min=0.0)
    fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
    return fb.to(dtype)


def log_mel(x, n_fft=2048, n_mels=80, sr=16000):
    """x [B,1,T] -> log(clamp(mel power spectrogram, 1e-5)) [B,1,n_mels,1+T//hop]."""
    hop = n_fft // 4
    B = x.shape[0]
    win = torch.hann_window(n_fft, periodic=True, dtype=x.dtype, device=x.device)
    spec = torch.stft(x.reshape(B, -1), n_fft, hop_length=hop, win_length=n_fft, window=win,
                      center=True, pad_mode='reflect', normalized=False, onesided=True,
                      return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                      # [B, n_freqs, frames]
    fb = mel_filterbank(n_fft // 2 + 1, n_mels, sr, dtype=x.dtype).to(x.device)
    mel = torch.matmul(power.transpose(1, 2), fb).transpose(1, 2)  # [B, n_mels, frames]
    return torch.log(torch.clamp(mel, min=1e-5)).unsqueeze(1)


def log_mel_l1(sig, ref, fft_sizes=(2048, 1024, 512), all_resolutions=False):
    """Reference behaviour (default): only fft_sizes[0] contributes (Q1)."""
    sizes = fft_sizes if all_resolutions else fft_sizes[:1]
    return sum((log_mel(sig, n) - log_mel(ref, n).detach()).abs().mean() for n in sizes)


def contrastive(X, Y, idx_x, idx_y):
    """X, Y [B,C,T]; idx_* [B,T,N] draws in [0,T-1) (the reference samples them inside).

    logits[b,t,0] = cos(X[b,:,t], Y[b,:,t]); logits[b,t,1+n] = cos(X[b,:,t], X[b,:,neg(b,t,n)]);
    both directions, CE against class 0, mean over 2B*T. Temperature 1 (Q2).
    """
    def negs(Z, idx):
        T = Z.shape[2]
        self_idx = torch.arange(T, device=Z.device).view(1, T, 1)
        idx = idx + (idx >= self_idx).to(idx.dtype)
        return Z.detach()[torch.arange(Z.shape[0]).view(-1, 1, 1), :, idx].permute(0, 3, 1, 2)  # B,C,T,N

    def sim(A, Bm, ng):
        tg = torch.cat([Bm.unsqueeze(-1), ng], dim=-1)           # B,C,T,1+N
        return F.cosine_similarity(A.unsqueeze(-1), tg, dim=1)   # B,T,1+N
    lg = torch.cat([sim(X, Y, negs(X, idx_x)), sim(Y, X, negs(Y, idx_y))], dim=0)
    tgt = torch.zeros(lg.shape[:-1], dtype=torch.long, device=lg.device)
    return F.cross_entropy(lg.transpose(1, 2), tgt)
