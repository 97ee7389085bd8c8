"""Deterministic synthetic weights and inputs for the TD-VC-GAN train-step path.

Nothing here touches the GPU or the HIP library: both the product side (bench.py,
tests) and the checker side (oracle/) use these so that "same weights, same
inputs" is true by construction without shipping weight blobs.

Reference facts restated here (not copied):
  * signal statistics: RMS normalisation to -30 dB + random gain/sign augmentation
    (data/dataset.py:119-125, util/__init__.py:53-62), 1e-9 noise floor (train.py:109)
  * numpy seed 1234 (train.py:78), in-batch target permutation (train.py:221-222)
  * F0 frame rate: hop 64 samples with one extra frame (util/crepe.py:10,40-45)
"""
import math
import zlib

import numpy as np
import torch

SAMPLE_RATE = 16000
NUM_SPK = 16
F0_HOP = 64


def _rs(key: str) -> np.random.RandomState:
    return np.random.RandomState(zlib.crc32(key.encode()) & 0x7FFFFFFF)


def fill_tensor_by_key(key: str, shape, v_for_g: np.ndarray = None) -> np.ndarray:
    """Deterministic value for one state_dict entry, from its key and shape only.

    weight_v / weight / bias ~ U(-b, b), b = 1/sqrt(fan_in) (the default-init scale,
    SURVEY.md App. D); weight_g = ||v|| * U(0.8, 1.2) per dim-0 slice so the effective
    weight differs from v (exercises the weight-norm kernel, Q12).
    """
    rs = _rs(key)
    shape = tuple(shape)
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'weight_g':
        assert v_for_g is not None
        nrm = np.sqrt((v_for_g.reshape(v_for_g.shape[0], -1).astype(np.float64) ** 2).sum(1))
        return (nrm * rs.uniform(0.8, 1.2, size=nrm.shape)).astype(np.float32).reshape(shape)
    if leaf in ('weight_v', 'weight'):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        b = 1.0 / math.sqrt(max(fan_in, 1))
        return rs.uniform(-b, b, size=shape).astype(np.float32)
    if leaf == 'bias':
        # bias bound uses the layer's fan_in, which the key alone does not carry;
        # a fixed small bound keeps it shape-independent and deterministic.
        return rs.uniform(-0.05, 0.05, size=shape).astype(np.float32)
    raise KeyError(f'unknown state_dict leaf {leaf!r} in {key!r}')


def fill_state_dict(shapes: dict) -> dict:
    """shapes: {key: shape} -> {key: torch.float32 tensor}. Handles (v, g) pairing."""
    out = {}
    for k, shp in shapes.items():
        if k.endswith('.weight_g'):
            continue
        out[k] = fill_tensor_by_key(k, shp)
    for k, shp in shapes.items():
        if k.endswith('.weight_g'):
            v = out[k[:-len('weight_g')] + 'weight_v']
            out[k] = fill_tensor_by_key(k, shp, v_for_g=v)
    return {k: torch.from_numpy(np.ascontiguousarray(out[k])) for k in shapes}


def make_f0(rs: np.random.RandomState, B: int, T: int) -> np.ndarray:
    """Piecewise-smooth F0 contour [B,1,T/64+1], 90-300 Hz with ~30 % unvoiced frames."""
    n = T // F0_HOP + 1
    f0 = np.zeros((B, 1, n), np.float32)
    for b in range(B):
        base = rs.uniform(100.0, 260.0)
        t = np.arange(n)
        contour = base * (1.0 + 0.15 * np.sin(2 * np.pi * t / rs.uniform(40, 120) + rs.uniform(0, 6.28)))
        contour = np.clip(contour, 90.0, 300.0)
        voiced = np.ones(n, bool)
        pos = 0
        while pos < n:  # alternating voiced / unvoiced runs
            run = int(rs.randint(8, 40))
            if rs.uniform() < 0.3:
                voiced[pos:pos + run] = False
            pos += run
        f0[b, 0] = np.where(voiced, contour, 0.0)
    return f0


def excitation_from_f0(f0: np.ndarray, rs: np.random.RandomState, step: int = F0_HOP,
                       sr: int = SAMPLE_RATE) -> np.ndarray:
    """numpy restatement of the sine+noise excitation (util/__init__.py:22-50).

    Drop the last frame, nearest-upsample omega = 2*pi*f0/sr by `step`, linearly
    interpolate inside voiced runs, cumulative phase, 0.1*sin + N(0, 0.003); unvoiced
    samples are N(0, 0.1/3). Random draws come from `rs`, so the result is an INPUT
    fixture for both sides rather than something either side regenerates.
    """
    f0 = f0[:, :, :-1]
    B, _, n = f0.shape
    w = 2 * np.pi * f0.astype(np.float64) / sr
    up = np.repeat(w, step, axis=-1)
    # linear interpolation, align_corners=False semantics of F.interpolate(mode='linear')
    T = n * step
    src = (np.arange(T) + 0.5) / step - 0.5
    src = np.clip(src, 0, None)
    i0 = np.minimum(np.floor(src).astype(int), n - 1)
    i1 = np.minimum(i0 + 1, n - 1)
    lam = src - i0
    lin = w[..., i0] * (1 - lam) + w[..., i1] * lam
    both_voiced = (w[..., i0] > 0) & (w[..., i1] > 0)
    up = np.where(both_voiced, lin, up)
    phase = np.cumsum(up, axis=-1) + rs.uniform() * 2 * np.pi
    exc = 0.1 * np.sin(phase) + rs.randn(B, 1, T) * 0.003
    unv = up == 0
    exc = np.where(unv, rs.randn(B, 1, T) * (0.1 / 3.0), exc)
    return exc.astype(np.float32)


def make_batch(B: int, T: int = SAMPLE_RATE, seed: int = 1234, num_spk: int = NUM_SPK,
               conversion: bool = True) -> dict:
    """One synthetic minibatch with the shapes/statistics of the reference's loader.

    Returns CPU tensors: signal_real, signal_corrupted [B,1,T]; label_src, label_tgt [B]
    int64; c_src, c_tgt one-hot [B,num_spk]; c_f0_src, c_f0_conv [B,1,T] excitations;
    perm [B].
    """
    assert T % 320 == 0 and T >= 8320, 'T must be a multiple of 320 and >= 8320 (SURVEY Q14)'
    rs = np.random.RandomState(seed)
    target_rms = 10.0 ** (-30.0 / 20.0)

    def sig():
        x = rs.randn(B, 1, T)
        x = x / np.sqrt((x ** 2).mean(axis=-1, keepdims=True)) * target_rms
        gain = rs.uniform(0.3, 1.0, size=(B, 1, 1)) * rs.choice([-1.0, 1.0], size=(B, 1, 1))
        return x * gain + 1e-9 * rs.randn(B, 1, T)

    real = sig()
    corrupted = sig()
    corrupted = corrupted * np.sqrt((real ** 2).mean(-1, keepdims=True)) / (
        np.sqrt((corrupted ** 2).mean(-1, keepdims=True)) + 1e-8)
    label_src = rs.randint(0, num_spk, size=B)
    perm = rs.permutation(B) if conversion else np.arange(B)
    label_tgt = label_src[perm]
    f0_src = make_f0(rs, B, T)
    if conversion:  # log-F0 mean shift towards the target speaker (train.py:245-251)
        f0_tgt = f0_src[perm]

        def mu(f):
            v = f > 0
            return (v * np.log(f + 1e-6)).sum(-1, keepdims=True) / (v.sum(-1, keepdims=True) + 1e-6)
        f0_conv = np.where(f0_src > 0, np.exp(np.log(f0_src + 1e-6) + mu(f0_tgt) - mu(f0_src)), 0.0)
        f0_conv = f0_conv.astype(np.float32)
    else:
        f0_conv = f0_src
    c_f0_conv = excitation_from_f0(f0_conv, rs)
    c_f0_src = excitation_from_f0(f0_src, rs)

    def onehot(lbl):
        o = np.zeros((B, num_spk), np.float32)
        o[np.arange(B), lbl] = 1.0
        return o

    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    return dict(signal_real=t(real), signal_corrupted=t(corrupted),
                label_src=t(label_src, torch.int64), label_tgt=t(label_tgt, torch.int64),
                c_src=t(onehot(label_src)), c_tgt=t(onehot(label_tgt)),
                c_f0_src=t(c_f0_src), c_f0_conv=t(c_f0_conv), perm=t(perm, torch.int64))


def contrastive_indices(B: int, T: int, n_neg: int, seed: int) -> torch.Tensor:
    """Negative-sample indices [B,T,n_neg] in [0,T-1) for the contrastive loss.

    The reference draws them with torch.randint inside the loss (util/losses.py:79-81);
    parity needs both sides to use the same draw, so it is lifted to an input.
    """
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.randint(0, T - 1, size=(B, T, n_neg)).astype(np.int64))


class FrameFeatureExtractor(torch.nn.Module):
    """Synthetic stand-in for the frozen SSL feature extractor of `encoder_model='wavlm'` (WavLM-Large: third-party, its
    checkpoint wavlm/WavLM-Large.pt does not ship with the reference). Same interface and framing as the module the
    reference calls (model/ssl_encoder.py:141-145): `extract_features(wave [B, L]) -> ([B, L', 1024], ...)` with a
    receptive field of 400 samples and a hop of 320, i.e. L' = (L - 400) // 320 + 1 -- one frame per 320 samples of the
    160-sample left-padded signal. Plain PyTorch (like the real extractor, it is not part of the HIP path); deterministic
    weights, frozen. Used by the tests / tools on both the product and the oracle side."""

    def __init__(self, dim: int = 1024, field: int = 400, hop: int = 320):
        super().__init__()
        rs = _rs('ssl_stand_in')
        w = rs.uniform(-1.0, 1.0, size=(dim, 1, field)).astype(np.float32) * (4.0 / math.sqrt(field))
        self.weight = torch.nn.Parameter(torch.from_numpy(w), requires_grad=False)
        self.hop = hop

    def extract_features(self, wave: torch.Tensor):
        # gain: the training signals sit at -30 dB RMS; bring the frames to O(1) before the squashing non-linearity
        f = torch.tanh(torch.nn.functional.conv1d(wave[:, None, :].float() * 30.0, self.weight, stride=self.hop))
        return (f.transpose(1, 2).contiguous(),)


class WavLMShapedExtractor(torch.nn.Module):
    """Frozen stand-in with the COMPUTE SHAPE of WavLM-Large (wavlm/WavLM.py in the reference; checkpoint not shipped), built
    from stock torch.nn modules only, for timing BASELINE config #5 (`wavlm-stage2_2`): a 7-layer strided conv front end to 512
    channels (kernels 10,3,3,3,3,2,2 / strides 5,2,2,2,2,2,2: hop 320, receptive field 400, layer-norm + GELU after each conv),
    LayerNorm + projection to 1024, a grouped positional conv (k 128, 16 groups), 24 pre-norm transformer layers (1024 wide,
    16 heads, 4096 FFN) and a final LayerNorm -- ~315 M parameters, random init, eval mode, no gradient. Same interface and
    framing as the module the reference calls (model/ssl_encoder.py:141-145): extract_features(wave [B, L]) -> ([B, L', 1024],).
    Not WavLM: no relative position bias / gating (a few percent of its FLOPs), random weights -> throughput only, no parity
    claim on the features."""

    def __init__(self, layers: int = 24, dim: int = 1024, heads: int = 16, ffn: int = 4096, seed: int = 20240):
        super().__init__()
        nn = torch.nn
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(seed)
            specs = [(10, 5)] + [(3, 2)] * 4 + [(2, 2)] * 2
            convs, norms, cin = [], [], 1
            for k, st in specs:
                convs.append(nn.Conv1d(cin, 512, k, st, bias=False)); norms.append(nn.LayerNorm(512)); cin = 512
            self.convs, self.norms = nn.ModuleList(convs), nn.ModuleList(norms)
            self.post_norm = nn.LayerNorm(512)
            self.proj = nn.Linear(512, dim)
            self.pos_conv = nn.Conv1d(dim, dim, 128, padding=64, groups=16)
            self.layers = nn.ModuleList([nn.TransformerEncoderLayer(dim, heads, ffn, dropout=0.0, activation='gelu', batch_first=True,
                                                                    norm_first=True) for _ in range(layers)])
            self.final_norm = nn.LayerNorm(dim)
        for p_ in self.parameters():
            p_.requires_grad = False
        self.eval()

    @torch.no_grad()
    def extract_features(self, wave: torch.Tensor):
        F = torch.nn.functional
        x = wave[:, None, :].float() * 30.0            # the training signals sit at -30 dB RMS
        for conv, norm in zip(self.convs, self.norms):
            x = F.gelu(norm(conv(x).transpose(1, 2)).transpose(1, 2))
        x = self.proj(self.post_norm(x.transpose(1, 2)))                    # [B, L', 1024]
        pos = F.gelu(self.pos_conv(x.transpose(1, 2))[:, :, :x.shape[1]]).transpose(1, 2)
        x = x + pos
        for layer in self.layers:
            x = layer(x)
        return (self.final_norm(x).contiguous(),)
