"""ORACLE — test infrastructure only. CPU restatement (plain PyTorch fp32/fp64 functional code)
of the TD-VC-GAN generator / discriminator forward path. Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this; the product path never does.

Parity status: pinned. The reference ships no golden vectors, so this
restatement is pinned against outputs of the reference itself, imported in the build
container by oracle/make_golden.py (fixtures under tests/golden/).

Every function works on a flat {state_dict key: tensor} mapping `sd` with the reference's
key names, so real checkpoints interchange. Autograd over the `sd` tensors gives the
gradient oracle.

Reference anchors (file:line under /root/reference):
  weight norm           model/generator.py:14, util/__init__.py:16-20   (old-style, dim=0)
  FiLM residual block   model/generator.py:69-111
  MRF block             model/generator.py:175-194
  excitation pyramid    model/generator.py:141-173, 364-372 ; util/__init__.py:104-113
  encoder               model/generator.py:197-272
  decoder               model/generator.py:276-406
  generator wiring      model/generator.py:409-508
  discriminator         model/discriminator.py:7-53, 77-118 ; util/dsp.py:5-16
  conditional IN        model/conditional_instance_norm.py:4-19
"""
import math

import torch
import torch.nn.functional as F

SLOPE = 0.2
MRF_KERNELS = (3, 7, 11)
MRF_DILATIONS = (1, 3, 5)


# --------------------------------------------------------------------------- primitives
def wn_weight(sd, p):
    """Effective weight of a weight-normed conv: g * v / ||v|| over all dims but 0."""
    v, g = sd[p + '.weight_v'], sd[p + '.weight_g']
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
    return v * (g / n)


def weight_of(sd, p):
    return wn_weight(sd, p) if (p + '.weight_v') in sd else sd[p + '.weight']


def conv(sd, p, x, stride=1, pad=0, dil=1, reflect=False, groups=1):
    w = weight_of(sd, p)
    b = sd.get(p + '.bias')
    if reflect and pad > 0:
        x = F.pad(x, (pad, pad), mode='reflect')
        pad = 0
    return F.conv1d(x, w, b, stride=stride, padding=pad, dilation=dil, groups=groups)


def conv_t(sd, p, x, stride, pad):
    return F.conv_transpose1d(x, weight_of(sd, p), sd.get(p + '.bias'), stride=stride, padding=pad)


def lrelu(x):
    return F.leaky_relu(x, SLOPE)


def kaiser_sinc_even(L, w):
    """L+1 taps, cutoff w (cycles/sample*2), Kaiser beta 2.5, unit DC gain (util/__init__.py:104-113)."""
    n = torch.arange(-(L // 2), L // 2 + 1, dtype=torch.float32)
    f = torch.sin(math.pi * w * n) / (math.pi * n + 1e-8)
    f[n.numel() // 2] = w
    f = f * torch.kaiser_window(L + 1, False, 2.5)
    return f / f.sum()


def kaiser_sinc_odd(L, fc, beta):
    """L taps (L odd) low-pass, unit DC gain (util/dsp.py:5-16)."""
    assert L % 2 == 1
    h = (L - 1) // 2
    n = torch.arange(-h, h + 1, dtype=torch.float32)
    f = torch.sin(math.pi * fc * n) / (math.pi * n + 1e-8)
    f[h] = fc
    f = f * torch.kaiser_window(L, False, beta)
    return f / f.sum()


# --------------------------------------------------------------------------- generator
def film_block(sd, p, x, c, k, d):
    """One FiLM residual block; `c` is None (encoder) or [B,136,T] (decoder)."""
    h = conv(sd, p + '.conv.1', lrelu(x), pad=(k * d - d) // 2, dil=d, reflect=True)
    if c is not None:
        cv = conv(sd, p + '.cond_var.0', c, pad=1)
        cv = conv(sd, p + '.cond_var.2', lrelu(cv), pad=1)
        gamma, beta = cv.chunk(2, dim=1)
        h = h * (1 + gamma) + beta
    return conv(sd, p + '.posconv.1', lrelu(h)) + x


def mrf(sd, p, x, c=None):
    acc = 0
    for i, k in enumerate(MRF_KERNELS):
        xs = x
        for j, d in enumerate(MRF_DILATIONS):
            xs = film_block(sd, f'{p}.blocks.{i}.{j}', xs, c, k, d)
        acc = acc + xs
    return acc / len(MRF_KERNELS)


def encoder(sd, x, ratios=(2, 2, 8, 10), p='encoder.encoder'):
    """x [B,1,T] -> L2-normalised content embedding [B,content_dim,T/prod(ratios)]."""
    x = conv(sd, f'{p}.0', x, pad=3, reflect=True)
    idx = 1
    for r in ratios:
        # idx: Identity, idx+1: LeakyReLU, idx+2: strided conv, idx+3: MRF
        x = conv(sd, f'{p}.{idx + 2}', lrelu(x), stride=r, pad=r // 2 + r % 2)
        x = mrf(sd, f'{p}.{idx + 3}', x)
        idx += 4
    x = conv(sd, f'{p}.{idx + 1}', lrelu(x), pad=3)
    x = conv(sd, f'{p}.{idx + 3}', lrelu(x), pad=3)
    return F.normalize(x, dim=1)


def excite_block(sd, p, x, r):
    sh = F.conv1d(x, sd[p + '.shortcut.weight'], sd[p + '.shortcut.bias'])
    C = sh.shape[1]
    fir = kaiser_sinc_even(16 * r, 1.0 / r).to(x.dtype).view(1, 1, -1).expand(C, 1, -1)
    sh = F.conv1d(sh, fir, stride=r, padding=8 * r, groups=C)
    h = conv(sd, p + '.block.0', x, stride=r, pad=r // 2)
    h = conv(sd, p + '.block.2', lrelu(h), pad=2)
    h = conv(sd, p + '.block.4', lrelu(h), pad=2)
    return h + sh


def excitation_pyramid(sd, c_var, ratios=(10, 8, 2, 2), p='decoder.excite_downsample'):
    """[B,1,T] -> list of 8-channel excitations at T, T/2, T/4, T/32 (, T/320 — dead, Q7)."""
    n = len(ratios)
    out = [conv(sd, f'{p}.{n}', c_var, pad=3, reflect=True)]
    for i in reversed(range(1, n)):  # stage 0 (the coarsest) is never consumed: skipped
        out.append(excite_block(sd, f'{p}.{i}', out[-1], ratios[i]))
    return out


def decoder(sd, x, emb, c_var, ratios=(10, 8, 2, 2), p='decoder'):
    """x [B,content,T/320], emb [B,128] speaker embedding, c_var [B,1,T].

    Returns (y [B,1,T], [sub T/4, sub T/2])."""
    pyr = excitation_pyramid(sd, c_var, ratios, p + '.excite_downsample')  # fine -> coarse
    dd = p + '.decoder'
    x = conv(sd, f'{dd}.1', lrelu(x), pad=3)
    x = conv(sd, f'{dd}.3', lrelu(x), pad=3)
    subs = []
    idx = 4
    for i, r in enumerate(ratios):
        x = conv_t(sd, f'{dd}.{idx + 2}', lrelu(x), stride=r, pad=r // 2 + r % 2)
        if i in (1, 2):  # sub-scale heads read the up-sampled tensor BEFORE its MRF block (:391-394)
            subs.append(torch.tanh(conv(sd, f'{p}.subsample_out_layers.{i}.1', lrelu(x), pad=3, reflect=True)))
        exc = pyr[len(ratios) - 1 - i]
        c = torch.cat([emb.unsqueeze(2).expand(-1, -1, x.shape[2]), exc], dim=1)
        x = mrf(sd, f'{dd}.{idx + 3}', x, c)
        idx += 4
    y = torch.tanh(conv(sd, f'{dd}.{idx + 2}', lrelu(x), pad=3, reflect=True))
    return y, subs


def generator(sd, x, c_tgt, c_var, ratios=(10, 8, 2, 2)):
    """Returns (y, [sub_T/4, sub_T/2], content_embedding). c_tgt is one-hot float [B,num_spk]."""
    emb = F.linear(c_tgt, sd['embedding.weight'], sd['embedding.bias'])
    content = encoder(sd, x, tuple(reversed(ratios)))
    y, subs = decoder(sd, content, emb, c_var, ratios)
    return y, subs, content


# --------------------------------------------------------------------------- discriminator
def disc_single(sd, p, x, label, n_layers=4, mult=4):
    feats = []
    x = lrelu(conv(sd, f'{p}.discriminator.0.0', x, pad=7, reflect=True))
    feats.append(x)
    for i in range(n_layers):
        cin = x.shape[1]
        x = lrelu(conv(sd, f'{p}.discriminator.{i + 1}.0', x, stride=4, pad=20, groups=cin // mult))
        feats.append(x)
    x = lrelu(conv(sd, f'{p}.discriminator.{n_layers + 1}.0', x, pad=2))
    feats.append(x)
    o = conv(sd, f'{p}.output', x, pad=1)
    idx = label.view(-1, 1, 1).expand(-1, 1, o.shape[2])
    return o.gather(1, idx), feats


def down2(x):
    f = kaiser_sinc_odd(129, 0.5, 10.0).to(x.dtype).view(1, 1, -1)
    return F.conv1d(x, f, stride=2, padding=64)


def disc_subsamples(x, num_disc=3):
    out = []
    for _ in range(num_disc - 1):
        x = down2(x)
        out.append(x)
    return out[::-1]  # [x/4, x/2]


def discriminator(sd, x, label, subscales=(), num_disc=3):
    outs, feats = [], []
    for i in range(num_disc):
        o, f = disc_single(sd, f'discriminators.{i}', x, label)
        outs.append(o); feats.append(f)
        x = down2(x)
    for xs, i in zip(subscales, reversed(range(num_disc))):
        o, f = disc_single(sd, f'discriminators.{i}', xs, label)
        outs.append(o); feats.append(f)
    return outs, feats


# --------------------------------------------------------------------------- cond. instance norm
def cond_instance_norm(sd, p, x, c, eps=1e-5):
    """(1+gamma(c)) * IN(x) + beta(c); c [B,n_cond] (Linear) or [B,n_cond+1,T] (Conv k=5 'same')."""
    p = p + '.' if p else ''
    if c.dim() == 2:
        h = F.linear(c, sd[p + 'embedding.weight'], sd[p + 'embedding.bias']).unsqueeze(2)
    else:
        h = F.conv1d(c, sd[p + 'embedding_conv.weight'], sd[p + 'embedding_conv.bias'], padding=2)
    gamma, beta = h.chunk(2, dim=1)
    mu = x.mean(dim=2, keepdim=True)
    var = x.var(dim=2, unbiased=False, keepdim=True)
    return (1 + gamma) * ((x - mu) / torch.sqrt(var + eps)) + beta


# --------------------------------------------------------------------------- latent classifier
def latent_classifier(sd, x, n_layers=3, p='classifier'):
    """Gradient-reversed speaker classifier on the content embedding
    (model/latent_classifier.py:8-39, model/grad_rev.py:3-17)."""
    class _Rev(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            return -g
    x = _Rev.apply(x)
    idx = 1
    for _ in range(n_layers):
        x = lrelu(conv(sd, f'{p}.{idx}', x, stride=2, pad=10))
        idx += 2
    x = lrelu(conv(sd, f'{p}.{idx}', x, pad=2))
    x = conv(sd, f'{p}.{idx + 2}', x, pad=1)
    return x.mean(dim=2)


# --------------------------------------------------------------------------- SSL (WavLM-side) content encoder
def wn_stack(sd, p, x, n_layers, hidden, kernel_size=5, dilation_rate=1):
    """WaveNet-style gated stack, g=None, x_mask=1, dropout 0 (model/ssl_encoder.py:17-90): per layer
    x_in = in_i(x); acts = tanh(x_in[:H]) * sigmoid(x_in[H:]); rs = res_skip_i(acts); x += rs[:H]; out += rs[H:]
    (the last layer's res_skip has H channels, all of which go to `out`)."""
    out = torch.zeros_like(x)
    for i in range(n_layers):
        d = dilation_rate ** i
        x_in = conv(sd, f'{p}.in_layers.{i}', x, pad=(kernel_size * d - d) // 2, dil=d)
        acts = torch.tanh(x_in[:, :hidden]) * torch.sigmoid(x_in[:, hidden:])
        rs = conv(sd, f'{p}.res_skip_layers.{i}', acts)
        if i < n_layers - 1:
            x = x + rs[:, :hidden]
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out


def ssl_content_encoder(sd, c, n_layers=16, emb_dim=128, kernel_size=5, dilation_rate=1, p='encoder.encoder'):
    """SSLEncoder.forward after the (frozen, external) WavLM feature extractor (model/ssl_encoder.py:93-148):
    pre (1x1) -> WN -> proj (1x1, 2*emb_dim channels); returns m = the first emb_dim channels (the sampled z is unused)."""
    x = conv(sd, p + '.pre', c)
    x = wn_stack(sd, p + '.enc', x, n_layers, emb_dim, kernel_size, dilation_rate)
    stats = conv(sd, p + '.proj', x)
    return stats[:, :emb_dim]


def generator_ssl(sd, c_feat, c_tgt, c_var, ratios=(10, 8, 2, 2), n_layers=16):
    """Generator with encoder_model='wavlm' (model/generator.py:453-454, 490-508), fed the SSL features [B,1024,T/320]."""
    emb = F.linear(c_tgt, sd['embedding.weight'], sd['embedding.bias'])
    content = ssl_content_encoder(sd, c_feat, n_layers)
    y, subs = decoder(sd, content, emb, c_var, ratios)
    return y, subs, content
