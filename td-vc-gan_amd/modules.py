"""Drop-in `Generator` / `CollaborativeMultibandDiscriminator` / `ConditionalInstanceNorm` /
`LatentClassifier` modules: the reference's constructor signatures, forward signatures and
state_dict keys, with every forward/backward FLOP in the HIP kernels.

Reference surface mirrored (file:line under the reference tree):
  Generator(...)                               model/generator.py:409-508
  Encoder / Decoder / MRFBlock / FiLMResnetBlock / ExciteDownsampleBlock   :69-111, :141-173, :175-194, :197-406
  CollaborativeMultibandDiscriminator(...)     model/discriminator.py:77-118 (Discriminator :7-53)
  ConditionalInstanceNorm(n_channel, n_cond)   model/conditional_instance_norm.py:4-19
  LatentClassifier(num_classes, C)             model/latent_classifier.py:8-39, model/grad_rev.py:3-17
Only the configuration space the shipped YAMLs reach is implemented (conv encoder, Identity norm
layers, weight_norm on, 'target' conditioning); anything else raises NotImplementedError.
"""
import math

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .arena import ConvSlot, ParamArena
from .ops import PRE_LRELU, PRE_NONE, ConvSpec

SLOPE = 0.2
MRF_KERNELS = (3, 7, 11)
MRF_DILATIONS = (1, 3, 5)


# ------------------------------------------------------------------------------- filters
def kaiser_filter_even(L_, w):
    """L_+1 taps (util/__init__.py:104-113): sinc(w) * kaiser(beta 2.5), unit DC gain."""
    n = torch.arange(-(L_ // 2), L_ // 2 + 1, dtype=torch.float32)
    f = torch.sin(math.pi * w * n) / (math.pi * n + 1e-8)
    f[n.numel() // 2] = w
    f = f * torch.kaiser_window(L_ + 1, False, 2.5)
    return f / f.sum()


def kaiser_filter_odd(L_, fc, beta=2.5):
    """L_ taps, L_ odd (util/dsp.py:5-16)."""
    if L_ % 2 == 0:
        raise Exception('Even length filter not implemented')
    h = (L_ - 1) // 2
    n = torch.arange(-h, h + 1, dtype=torch.float32)
    f = torch.sin(math.pi * fc * n) / (math.pi * n + 1e-8)
    f[h] = fc
    f = f * torch.kaiser_window(L_, False, beta)
    return f / f.sum()


# ------------------------------------------------------------------------------- parameter holders
class ConvParams(nn.Module):
    """Parameters of one conv with the reference's key names: weight_norm -> bias, weight_g, weight_v;
    plain -> weight, bias. For ConvTranspose1d dim 0 of the weight is the INPUT channel axis (Q12)."""

    def __init__(self, cin, cout, k, stride=1, pad=0, dil=1, groups=1, reflect=False, bias=True, wn=True,
                 transposed=False, out_pad=0):
        super().__init__()
        self.spec = ConvSpec(cin, cout, k, stride, pad, dil, groups, reflect, transposed, out_pad)
        shape = (cin, cout // groups, k) if transposed else (cout, cin // groups, k)
        fan_in = shape[1] * k
        bound = 1.0 / math.sqrt(fan_in)
        w = torch.empty(shape).uniform_(-bound, bound)
        self.has_bias, self.wn = bias, wn
        if not wn:
            self.weight = nn.Parameter(w)
        if bias:
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        if wn:
            self.weight_g = nn.Parameter(w.reshape(shape[0], -1).norm(dim=1).reshape(shape[0], 1, 1))
            self.weight_v = nn.Parameter(w)

    def forward(self, x, pre=PRE_NONE, post=L.POST_NONE, add=None):
        return ops.conv(x, self.spec, pre, post, add)


class LinearParams(nn.Module):
    """nn.Linear keys (weight [out,in], bias); runs as a K=1 conv over a length-1 sequence."""

    def __init__(self, cin, cout):
        super().__init__()
        bound = 1.0 / math.sqrt(cin)
        self.weight = nn.Parameter(torch.empty(cout, cin).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        self.spec = ConvSpec(cin, cout, 1)
        self.has_bias, self.wn = True, False

    def forward(self, x):
        return ops.conv(x.unsqueeze(2), self.spec).squeeze(2)


class FixedFIR:
    """Constant (non-trainable, non-persistent) FIR filter run through the conv kernels."""

    def __init__(self, taps: torch.Tensor, channels, stride, pad):
        self.taps = taps.reshape(1, 1, -1).expand(channels, 1, -1).contiguous()
        self.spec = ConvSpec(channels, channels, taps.numel(), stride, pad, 1, channels)
        self.dev_taps = None

    def to(self, device):
        if self.dev_taps is None or self.dev_taps.device != device:
            self.dev_taps = self.taps.to(device)
            self.spec.slot = ConvSlot(self.dev_taps.data_ptr(), 0, 0, 0, trainable=False)

    def __call__(self, x):
        self.to(x.device)
        return ops.conv(x, self.spec)


class ArenaModule(nn.Module):
    """Top-level model: owns the flat arenas and binds every conv's slot to them."""
    dead_prefixes = ()

    def _init_arena_state(self):
        self._arena = None
        self._frozen_weights = False
        # a checkpoint loaded AFTER the arena exists (e.g. while a TrainStep holds the weights frozen) must refresh the
        # effective weights w = g*v/||v||, otherwise forward() would keep using the old ones
        self.register_load_state_dict_post_hook(ArenaModule._after_load)

    @staticmethod
    def _after_load(module, incompatible_keys):
        if module._arena is not None:
            module._arena.materialize()

    @property
    def arena(self) -> ParamArena:
        return self._arena

    def ensure_arena(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise L.TdvcError('tdvc modules run on an MI355X device only (no CPU fallback): move inputs to cuda')
        ps = list(self.parameters())
        sig = tuple(p.requires_grad for p in ps)      # frozen sub-networks (requires_grad = False, train.py:195-197) change the arena layout
        if (self._arena is not None and self._arena.device == device and self._arena.owns(ps[0]) and self._arena.owns(ps[-1]) and
                getattr(self, '_arena_sig', None) == sig):
            return self._arena
        self._arena_sig = sig
        L.lib()   # fail loudly before touching anything if the extension is missing
        tr = [name for name, m in self.named_modules()
              if isinstance(m, ConvParams) and m.wn and m.spec.kind == L.CONV and m.spec.stride == 1 and m.spec.groups == 1]
        self._arena = ParamArena(self, device, self.dead_prefixes, transposable=tr)
        for name, m in self.named_modules():
            if isinstance(m, (ConvParams, LinearParams)):
                if any((name + '.').startswith(d) for d in self.dead_prefixes):
                    continue
                m.spec.slot = self._arena.slot(name, m.has_bias)
        for m in self.modules():
            if hasattr(m, 'on_slots_bound'):
                m.on_slots_bound()
        self._arena.materialize()
        return self._arena

    def begin_forward(self, x):
        a = self.ensure_arena(x.device)
        if not self._frozen_weights:
            a.materialize()
        return a

    def weights_frozen(self, flag=True):
        """While set, forward() reuses the effective weights of the last ParamArena.materialize()."""
        self._frozen_weights = flag


# ------------------------------------------------------------------------------- generator blocks
FUSED_COND = True      # cond_var.0 -> LeakyReLU -> cond_var.2 in one forward launch (ops.FilmCondFn)


class FiLMResnetBlock(nn.Module):
    def __init__(self, n_channel, n_cond_const, n_cond_var=0, dilation=1, kernel_size=3):
        super().__init__()
        pad = (kernel_size * dilation - dilation) // 2
        self.conv = nn.Sequential(nn.LeakyReLU(SLOPE), ConvParams(n_channel, n_channel, kernel_size, pad=pad, dil=dilation, reflect=True))
        self.posconv = nn.Sequential(nn.LeakyReLU(SLOPE), ConvParams(n_channel, n_channel, 1))
        self.has_cond = bool(n_cond_const or n_cond_var)
        if self.has_cond:
            nc = n_cond_const + n_cond_var
            self.cond_var = nn.Sequential(ConvParams(nc, nc, 3, pad=1), nn.LeakyReLU(SLOPE), ConvParams(nc, n_channel * 2, 3, pad=1))
            # exact split of cond_var.0 over its input channels (SURVEY §2.2 reduction 2): the n_cond_const
            # speaker-embedding channels are constant in time, so their contribution is a per-(sample, channel)
            # constant with a one-sample correction at each zero-padded edge == the same conv on a length-3 signal
            self.n_const, self.n_var = n_cond_const, n_cond_var
            self.spec_const = ConvSpec(n_cond_const, nc, 3, pad=1, w_cin=nc, w_cin_off=0)
            self.spec_var = ConvSpec(n_cond_var, nc, 3, pad=1, w_cin=nc, w_cin_off=n_cond_const)
        self.shortcut = nn.Identity()

    def on_slots_bound(self):
        if self.has_cond and self.cond_var[0].spec.slot is not None:
            s = self.cond_var[0].spec.slot
            self.spec_const.slot = s                                             # carries the bias
            self.spec_var.slot = ConvSlot(s.w, 0, s.dw, 0, s.trainable, s.arena, s.wt, s.seg)  # no bias on the time-varying part

    def forward(self, x, c=None, acc=None, scale=1.0, k3=None):
        """c: None (encoder), a dense [B,n_const+n_var,T] conditioning tensor (reference formulation), or a
        (emb [B,n_const] or emb3 [B,n_const,3], exc [B,n_var,T]) pair for the split formulation; k3: the embedding part of
        cond_var.0 when the caller has already computed it for all blocks of the stage (ops.film_k3_multi)."""
        gb = None
        if isinstance(c, tuple):
            emb3, exc = c
            if k3 is None:
                k3 = ops.film_k3(emb3, self.spec_const) if emb3.dim() == 2 else ops.conv(emb3, self.spec_const)
            if FUSED_COND and exc.shape[2] % 4 == 0:
                gb = ops.film_cond(exc, k3, self.spec_var, self.cond_var[2].spec)
            else:
                cv = ops.conv(exc, self.spec_var, k3=k3)
                gb = self.cond_var[2](cv, pre=PRE_LRELU)
        elif c is not None:
            cv = self.cond_var[0](c)
            gb = self.cond_var[2](cv, pre=PRE_LRELU)
        return ops.film_block(x, gb, acc, self.conv[1].spec, self.posconv[1].spec, scale)


class MRFBlock(nn.Module):
    def __init__(self, n_channel, n_cond_const=0, n_cond_var=0):
        super().__init__()
        self.blocks = nn.ModuleList([nn.ModuleList([FiLMResnetBlock(n_channel, n_cond_const, n_cond_var, d, k)
                                                    for d in MRF_DILATIONS]) for k in MRF_KERNELS])

    def forward(self, x, c=None):
        acc, s = None, 1.0 / len(self.blocks)
        # x feeds every branch, the conditioning every block: their gradients come back summed in one pass each (ops.fanout)
        nblk = sum(len(br) for br in self.blocks)
        xin = ops.fanout(x, len(self.blocks))
        cs, k3s = [c] * nblk, [None] * nblk
        if isinstance(c, tuple):
            if c[0].dim() == 2 and nblk <= 16:      # the embedding part of every block's cond_var.0 in one launch each way
                k3s = ops.film_k3_multi(c[0], [blk.spec_const for br in self.blocks for blk in br])
                cs = [(c[0], e) for e in ops.fanout(c[1], nblk)]
            else:
                cs = list(zip(ops.fanout(c[0], nblk), ops.fanout(c[1], nblk)))
        elif c is not None:
            cs = list(ops.fanout(c, nblk))
        i = 0
        for bi, branch in enumerate(self.blocks):
            xs = xin[bi]
            for j, blk in enumerate(branch):
                last = j == len(branch) - 1
                xs = blk(xs, cs[i], acc if last else None, s if last else 1.0, k3=k3s[i])
                i += 1
            acc = xs
        return acc


class ExciteDownsampleBlock(nn.Module):
    def __init__(self, in_channel, out_channel, scale_factor, n_layers=2, kernel_size=5):
        super().__init__()
        r = scale_factor
        self.scale_factor = r
        self.block = nn.ModuleList([ConvParams(in_channel, out_channel, 2 * r, stride=r, pad=r // 2)])
        for _ in range(n_layers):
            self.block += [nn.LeakyReLU(SLOPE), ConvParams(out_channel, out_channel, kernel_size, pad=(kernel_size - 1) // 2)]
        self.shortcut = ConvParams(in_channel, out_channel, 1, wn=False)
        self.fir = FixedFIR(kaiser_filter_even(16 * r, 1.0 / r), out_channel, r, 8 * r)

    def forward(self, x):
        sh = self.fir(self.shortcut(x))
        h = self.block[0](x)
        h = self.block[2](h, pre=PRE_LRELU)
        return self.block[4](h, pre=PRE_LRELU, add=sh)


class Encoder(nn.Module):
    def __init__(self, ratios, channels, embedding_dim):
        super().__init__()
        m = nn.ModuleList([ConvParams(1, channels[0], 7, pad=3, reflect=True)])
        for i, r in enumerate(ratios):
            m += [nn.Identity(), nn.LeakyReLU(SLOPE),
                  ConvParams(channels[i], channels[i + 1], 2 * r, stride=r, pad=r // 2 + r % 2),
                  MRFBlock(channels[i + 1])]
        m += [nn.LeakyReLU(SLOPE), ConvParams(channels[-1], channels[-1], 7, pad=3)]
        if embedding_dim:
            m += [nn.LeakyReLU(SLOPE), ConvParams(channels[-1], embedding_dim, 7, pad=3, bias=False)]
        self.encoder = m
        self._top = None     # set by Generator: lets `G.encoder(x)` (train.py:405) find the arena

    def forward(self, x, c=None):
        if self._top is not None:
            self._top().begin_forward(x)
        x = x.contiguous().float()
        first = True
        for mod in self.encoder:
            if isinstance(mod, ConvParams):
                x = mod(x, pre=PRE_NONE if first else PRE_LRELU)
                first = False
            elif isinstance(mod, MRFBlock):
                x = mod(x)
        return ops.L2NormFn.apply(x)


class Decoder(nn.Module):
    def __init__(self, ratios, channels, conditional_dim, embedding_dim):
        super().__init__()
        self.upsample_ratios = list(ratios)
        self.split_cond = True      # False = the reference's dense 136-channel conditioning convs
        excite = [8] * (len(ratios) + 1)
        sub_out = [False, True, True, False]
        m = nn.ModuleList()
        if embedding_dim:
            m += [nn.LeakyReLU(SLOPE), ConvParams(embedding_dim, channels[0], 7, pad=3, bias=False)]
        m += [nn.LeakyReLU(SLOPE), ConvParams(channels[0], channels[0], 7, pad=3)]
        self.upsample_idxs = []
        self.subsample_out_layers = nn.ModuleList()
        for i, r in enumerate(ratios):
            m += [nn.Identity(), nn.LeakyReLU(SLOPE),
                  ConvParams(channels[i], channels[i + 1], 2 * r, stride=r, pad=r // 2 + r % 2, transposed=True, out_pad=r % 2)]
            self.upsample_idxs.append(len(m))
            m += [MRFBlock(channels[i + 1], conditional_dim, excite[i + 1])]
            if i < len(sub_out) and sub_out[i]:
                self.subsample_out_layers.append(nn.Sequential(nn.LeakyReLU(SLOPE), ConvParams(channels[i + 1], 1, 7, pad=3, reflect=True), nn.Tanh()))
            else:
                self.subsample_out_layers.append(None)
        m += [nn.Identity(), nn.LeakyReLU(SLOPE), ConvParams(channels[-1], 1, 7, pad=3, reflect=True), nn.Tanh()]
        self.upsample_idxs.append(len(m))
        self.decoder = m
        self.excite_downsample = nn.ModuleList()
        for r, ci, co in zip(ratios, excite[:-1], excite[1:]):
            self.excite_downsample += [ExciteDownsampleBlock(ci, co, r)]
        self.excite_downsample += [ConvParams(1, excite[0], 7, pad=3, reflect=True)]

    def get_scaled_conditioning(self, c):
        """Fine -> coarse excitation pyramid. The coarsest stage (excite_downsample[0]) is never consumed by
        forward(): it is not computed and its parameters keep grad None."""
        out = []
        mods = list(reversed(self.excite_downsample))
        for mod in mods[:-1]:
            c = mod(c)
            out.append(c)
        return out

    def forward(self, x, c=None, c_var=None, out_subsample=False):
        if c_var is None:
            # the reference raises UnboundLocalError here (Q8); be explicit instead
            raise RuntimeError('Decoder.forward needs c_var (the F0 excitation): the reference has no path without it')
        pyr = self.get_scaled_conditioning(c_var.contiguous().float())
        if self.split_cond:   # the speaker embedding [B,128] itself: its conv over a length-3 constant signal is tdvc_film_k3
            emb3 = c.contiguous().float()
            n_mrf = sum(isinstance(m_, MRFBlock) for m_ in self.decoder)
            emb3s = list(ops.fanout(emb3, n_mrf))      # one alias per decoder stage
        subs = []
        scale = 0
        final_conv = len(self.decoder) - 2
        for i, mod in enumerate(self.decoder):
            if i == self.upsample_idxs[scale]:
                head = self.subsample_out_layers[scale]
                if head is not None:
                    subs.append(head[1](x, pre=PRE_LRELU, post=L.POST_TANH))
                exc = pyr[len(pyr) - 1 - scale]
                cond = (emb3s[scale], exc) if self.split_cond else ops.ConcatCondFn.apply(c, exc)
                scale += 1
            if isinstance(mod, MRFBlock):
                x = mod(x, cond)
            elif isinstance(mod, ConvParams):
                x = mod(x, pre=PRE_LRELU, post=L.POST_TANH if i == final_conv else L.POST_NONE)
        if out_subsample:
            return x, subs
        return x


class Generator(ArenaModule):
    dead_prefixes = ('decoder.excite_downsample.0.',)

    def __init__(self, decoder_ratios, decoder_channels, num_bottleneck_layers, num_classes, conditional_dim,
                 content_dim=None, num_res_blocks=3, num_enc_layers=0, encoder_model=None, norm_layer=None,
                 weight_norm=None, bot_cond='target', enc_cond=None, dec_cond=None, output_content_emb=False, cmodel=None):
        """Reference signature (model/generator.py:410-415) + `cmodel`: the frozen SSL feature extractor for
        encoder_model='wavlm' (the reference loads wavlm/WavLM-Large.pt itself, model/ssl_encoder.py:126-133; that
        checkpoint does not ship, so the module is injected — see ssl_encoder.SSLEncoder)."""
        super().__init__()
        self._init_arena_state()
        nls = norm_layer if isinstance(norm_layer, tuple) else (norm_layer,) * 3
        wns = weight_norm if isinstance(weight_norm, tuple) else (weight_norm,) * 3
        if any(n is not None for n in nls):
            raise NotImplementedError('norm layers other than Identity are not reachable from the shipped configs')
        if any(w != 'weight_norm' for w in wns):
            raise NotImplementedError('only weight_norm="weight_norm" is implemented')
        if encoder_model not in (None, 'conv', 'wavlm'):
            raise NotImplementedError(f'unknown encoder model {encoder_model!r}')
        if num_bottleneck_layers != 0 or bot_cond != 'target' or enc_cond is not None or dec_cond is None:
            raise NotImplementedError('only the shipped conditioning layout is implemented '
                                      '(0 bottleneck layers, encoder unconditioned, decoder on target)')
        self.output_content_emb = output_content_emb
        self.decoder = Decoder(decoder_ratios, list(decoder_channels), conditional_dim, content_dim)
        if encoder_model == 'wavlm':      # model/generator.py:453-454
            from .ssl_encoder import SSLEncoder
            self.encoder = SSLEncoder(encoder_model, num_enc_layers, content_dim, cmodel=cmodel)
            self.dead_prefixes = Generator.dead_prefixes + ('encoder.cmodel.',)     # frozen: never in the optimizer's live prefix
        else:
            self.encoder = Encoder(decoder_ratios[::-1], list(decoder_channels)[::-1], content_dim)
        import weakref
        self.encoder._top = weakref.ref(self)
        self.bottleneck = nn.ModuleList()
        self.embedding = LinearParams(num_classes, conditional_dim)

    def forward(self, x, c_tgt, c_src=None, c_var=None, out_subsample=False):
        self.begin_forward(x)
        emb = self.embedding(c_tgt.contiguous().float())
        top, self.encoder._top = self.encoder._top, None     # arena already prepared for this call
        try:
            content = self.encoder(x)
        finally:
            self.encoder._top = top
        if self.output_content_emb:
            self.content_embedding = content
        return self.decoder(content, emb, c_var, out_subsample=out_subsample)


def generator_forward_pair(G, x, x_other, conds, c_vars):
    """Product-side fast path for one training iteration (not part of the reference surface):
    the encoder runs ONCE on cat([x, x_other]) (real + corrupted signals) and the decoder runs ONCE on the
    real-signal content replicated per conditioning in `conds` / `c_vars` (target-speaker pass and identity pass).
    Exactly the arithmetic of separate G(x, c, c_var) calls — the reference recomputes the identical encoder
    output for each conditioning (train.py:322,367) — with 2x the work per launch.
    Returns ([(y, subs)] per conditioning, emb_x, emb_other)."""
    G.begin_forward(x)
    B = x.shape[0]
    top, G.encoder._top = G.encoder._top, None
    try:
        both = G.encoder(torch.cat([x, x_other], dim=0) if x_other is not None else x)
    finally:
        G.encoder._top = top
    emb_x = both[:B]
    emb_other = both[B:] if x_other is not None else None
    n = len(conds)
    emb = G.embedding(torch.cat([c.contiguous().float() for c in conds], dim=0))
    content = torch.cat([emb_x] * n, dim=0) if n > 1 else emb_x
    y, subs = G.decoder(content, emb, torch.cat(list(c_vars), dim=0) if n > 1 else c_vars[0], out_subsample=True)
    outs = _PairList((y[i * B:(i + 1) * B], [s_[i * B:(i + 1) * B] for s_ in subs]) for i in range(n))
    outs.full = (y, subs)          # the batched tensors themselves ([cond 0; cond 1; ...] along the batch axis)
    G.content_embedding = emb_x
    return outs, emb_x, emb_other


class _PairList(list):
    """List of per-conditioning (y, subs) pairs that also carries the batched tensors they are slices of (`.full`)."""


# ------------------------------------------------------------------------------- discriminator
class Discriminator(nn.Module):
    def __init__(self, num_classes, num_layers, num_channels_base, num_channel_mult=4, downsampling_factor=4,
                 conditional_dim=32, conditional='both'):
        super().__init__()
        ds = downsampling_factor
        self.discriminator = nn.ModuleList()
        self.discriminator += [nn.Sequential(ConvParams(1, num_channels_base, 15, pad=7, reflect=True), nn.LeakyReLU(SLOPE))]
        nf = num_channels_base
        for _ in range(num_layers):
            nf_prev, nf = nf, min(nf * num_channel_mult, 1024)
            self.discriminator += [nn.Sequential(ConvParams(nf_prev, nf, ds * 10 + 1, stride=ds, pad=ds * 5,
                                                            groups=nf_prev // num_channel_mult), nn.LeakyReLU(SLOPE))]
        self.discriminator += [nn.Sequential(ConvParams(nf, nf, 5, pad=2), nn.LeakyReLU(SLOPE))]
        self.output = ConvParams(nf, num_classes, 3, pad=1, bias=False)

    def forward(self, x, label_tgt):
        feats = []
        for layer in self.discriminator:
            x = layer[0](x, post=L.POST_LRELU)      # in-place LeakyReLU: the stored map is post-activation
            feats.append(x)
        x = self.output(x)
        return ops.GatherChFn.apply(x, label_tgt), feats


class CollaborativeMultibandDiscriminator(ArenaModule):
    def __init__(self, num_disc, num_classes, num_layers, num_channels_base, num_channel_mult=4, downsampling_factor=4,
                 conditional_dim=32, conditional='both'):
        super().__init__()
        self._init_arena_state()
        self.discriminators = nn.ModuleList([Discriminator(num_classes, num_layers, num_channels_base, num_channel_mult,
                                                           downsampling_factor, conditional_dim, conditional)
                                             for _ in range(num_disc)])
        self.L = 129
        self.down = FixedFIR(kaiser_filter_odd(self.L, 0.5, 10), 1, 2, (self.L - 1) // 2)

    def forward(self, x, label_tgt, subscales=[], views=False):
        """Returns (outs, features) in the reference's order: disc0@T, disc1@T/2, disc2@T/4, then the sub-scale
        passes disc2@sub[T/4], disc1@sub[T/2]. A discriminator that sees both a filtered and a sub-scale input of
        the same length runs them as ONE pass with the batch doubled (exact: D has no cross-sample op) — the
        short, latency-bound layers get twice the work per launch.
        views=True (train step): the two halves of such a pass are returned as losses.BatchView objects on the
        batched tensors instead of tensor slices, so that the loss terms produce one gradient per batched tensor."""
        self.begin_forward(x)
        x = x.contiguous().float()
        n = len(self.discriminators)
        B = x.shape[0]
        xs = [x]
        for _ in range(n - 1):
            xs.append(self.down(xs[-1]))
        subs = {}                       # discriminator index -> sub-scale input
        for x_sub, i in zip(subscales, reversed(range(n))):
            subs[i] = x_sub.contiguous().float()
        main, extra = [None] * n, {}
        for i, disc in enumerate(self.discriminators):
            if i in subs and subs[i].shape == xs[i].shape:
                o, f = disc(torch.cat([xs[i], subs[i]], dim=0), torch.cat([label_tgt, label_tgt], dim=0))
                if views:
                    from .losses import BatchView
                    main[i] = (BatchView(o, 0, B), [BatchView(m, 0, B) for m in f])
                    extra[i] = (BatchView(o, B, B), [BatchView(m, B, B) for m in f])
                else:
                    main[i] = (o[:B], [m[:B] for m in f])
                    extra[i] = (o[B:], [m[B:] for m in f])
            else:
                main[i] = disc(xs[i], label_tgt)
                if i in subs:
                    extra[i] = disc(subs[i], label_tgt)
        ret = list(main)
        for _, i in zip(subscales, reversed(range(n))):
            ret.append(extra[i])
        out, features = zip(*ret)
        return list(out), list(features)

    def get_subsamples(self, x):
        x = x.contiguous().float()
        ret = []
        for _ in range(len(self.discriminators) - 1):
            x = self.down(x)
            ret.append(x)
        return list(reversed(ret))


# ------------------------------------------------------------------------------- conditional instance norm
class ConditionalInstanceNorm(ArenaModule):
    def __init__(self, n_channel, n_cond, n_conf_var=0):
        super().__init__()
        self._init_arena_state()
        self.n_channel = n_channel
        self.embedding = LinearParams(n_cond, n_channel * 2)
        self.embedding_conv = ConvParams(n_cond + 1, n_channel * 2, 5, pad=2, wn=False)

    def forward(self, x, c):
        self.begin_forward(x)
        if c.dim() == 2:
            gb = self.embedding(c.contiguous().float()).unsqueeze(2)
        else:
            gb = self.embedding_conv(c.contiguous().float())
        return ops.CinFn.apply(x.contiguous().float(), gb, 1e-5)


# ------------------------------------------------------------------------------- latent classifier
class _GradRev(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return ops.axpby(g.contiguous(), None, -1.0, 0.0)


class LatentClassifier(ArenaModule):
    def __init__(self, num_classes, num_channels_input, num_layers=3, num_channel_mult=2, downsampling_factor=2):
        super().__init__()
        self._init_arena_state()
        ds = downsampling_factor
        self.classifier = nn.ModuleList([nn.Identity()])
        nf = num_channels_input
        for _ in range(num_layers):
            nf_prev, nf = nf, nf * num_channel_mult
            self.classifier += [ConvParams(nf_prev, nf, ds * 10 + 1, stride=ds, pad=ds * 5), nn.LeakyReLU(SLOPE)]
        self.classifier += [ConvParams(nf, nf, 5, pad=2), nn.LeakyReLU(SLOPE)]
        self.classifier += [ConvParams(nf, num_classes, 3, pad=1, bias=False)]
        self._pool = {}

    def forward(self, x):
        self.begin_forward(x)
        x = _GradRev.apply(x.contiguous().float())
        convs = [m for m in self.classifier if isinstance(m, ConvParams)]
        for m in convs[:-1]:
            x = m(x, post=L.POST_LRELU)
        x = convs[-1](x)
        # F.avg_pool1d over the full length, as a fixed 1-tap-per-sample FIR: mean over T
        T = x.shape[2]
        key = (T, x.shape[1])
        if key not in self._pool:        # kept alive: the autograd graph holds only the raw pointer of the taps
            self._pool[key] = FixedFIR(torch.full((T,), 1.0 / T), x.shape[1], 1, 0)
        return self._pool[key](x).squeeze(2)
