"""ORACLE — test infrastructure only (see oracle/model.py header). CPU restatement of one
TD-VC-GAN training iteration: D-step then G-step with both AdamW updates.

Reference anchors: train.py:209-521 (loop body), :259-316 (D-step), :320-510 (G-step),
:188-189 (AdamW built positionally: eps 1e-8, weight_decay 1e-2 — SURVEY Q11),
:477-482 (loss assembly). The CREPE-backed F0 term (train.py:429-470) is excluded on both
sides (torchcrepe absent); F0 enters only as the excitation inputs.

Dead work the reference performs but that reaches no parameter update is not reproduced
(SURVEY Q3, Q5, Q6): D grads in the G-step, G grads in the D-step. The second, bit-identical
generator forward of the G-step (Q4) IS recomputed here, as the reference does.
"""
from dataclasses import dataclass, field

import torch

from . import losses as L
from . import model as M


@dataclass
class StepConfig:
    no_conv: bool = False
    lambda_rec: float = 0.0
    lambda_idt: float = 5.0
    lambda_feat: float = 2.0
    lambda_spec: float = 5.0
    lambda_cont_emb: float = 10.0
    lambda_corrupted: float = 1.0
    lambda_latcls: float = 0.0
    lr_g: float = 1e-4
    lr_d: float = 1e-4
    betas: tuple = (0.8, 0.99)
    eps: float = 1e-8
    weight_decay: float = 1e-2
    n_neg: int = 100
    fft_sizes: tuple = (2048, 1024, 512)
    # loop switches off in every shipped YAML (train.py:357-360,381-384 / :409-413 / :335-336 / :289-290,489-490 / :259,320 / :195-197)
    lambda_wave: float = 0.0
    lambda_converted: float = 0.0
    jitter_amp: int = 0
    grad_max_norm_d: float = None
    grad_max_norm_g: float = None
    d_step_interval: int = 1
    g_step_interval: int = 1
    freeze_subnets: tuple = ()

    @staticmethod
    def from_hparams(train: dict) -> 'StepConfig':
        g = train.get
        return StepConfig(no_conv=bool(g('no_conv', False)), lambda_rec=float(g('lambda_rec', 0)),
                          lambda_idt=float(g('lambda_idt', 0)), lambda_feat=float(g('lambda_feat', 0)),
                          lambda_spec=float(g('lambda_spec', 0)), lambda_cont_emb=float(g('lambda_cont_emb', 0)),
                          lambda_corrupted=float(g('lambda_corrupted', 0)), lambda_latcls=float(g('lambda_latcls', 0)),
                          lr_g=float(g('lr_g', 1e-4)),
                          lr_d=float(g('lr_d', 1e-4)), betas=tuple(g('adam_beta', (0.8, 0.99))),
                          lambda_wave=float(g('lambda_wave', 0) or 0), lambda_converted=float(g('lambda_converted', 0) or 0),
                          jitter_amp=int(g('jitter_amp', 0) or 0),
                          grad_max_norm_d=(float(g('grad_max_norm_D')) if g('grad_max_norm_D') is not None else None),
                          grad_max_norm_g=(float(g('grad_max_norm_G')) if g('grad_max_norm_G') is not None else None),
                          d_step_interval=int(g('D_step_interval', 1) or 1), g_step_interval=int(g('G_step_interval', 1) or 1),
                          freeze_subnets=tuple(g('freeze_subnets') or ()))


class AdamW:
    """Decoupled weight decay Adam, torch.optim.AdamW semantics."""

    def __init__(self, params: dict, lr, betas, eps, wd):
        self.p, self.lr, self.b1, self.b2, self.eps, self.wd = params, lr, betas[0], betas[1], eps, wd
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = {k: 0 for k in params}

    @torch.no_grad()
    def step(self):
        for k, p in self.p.items():
            if p.grad is None:      # skipped entirely, weight decay included
                continue
            self.t[k] += 1
            t = self.t[k]
            p.mul_(1 - self.lr * self.wd)
            self.m[k].mul_(self.b1).add_(p.grad, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(p.grad, p.grad, value=1 - self.b2)
            denom = (self.v[k].sqrt() / (1 - self.b2 ** t) ** 0.5).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-self.lr / (1 - self.b1 ** t))

    def zero_grad(self):
        for p in self.p.values():
            p.grad = None


class TrainStep:
    def __init__(self, sd_g: dict, sd_d: dict, cfg: StepConfig, sd_c: dict = None, ssl_extractor=None):
        """ssl_extractor: for a generator with encoder_model='wavlm' (model/generator.py:453-454) the frozen feature
        extractor `wave [B, L] -> [B, L', 1024]` features (model/ssl_encoder.py:141-145: 160-sample left pad, no grad);
        None = the conv content encoder."""
        self.ssl = ssl_extractor
        # train.py:195-197: a frozen encoder = requires_grad False on G.encoder.parameters() (AdamW then skips them: grad None)
        frozen = (lambda k: k.startswith('encoder.')) if 'encoder' in cfg.freeze_subnets else (lambda k: False)
        self.g = {k: v.clone().requires_grad_(not frozen(k)) for k, v in sd_g.items()}
        self.d = {k: v.clone().requires_grad_(True) for k, v in sd_d.items()}
        self.cfg = cfg
        self.iter_count = 0
        self.opt_g = AdamW(self.g, cfg.lr_g, cfg.betas, cfg.eps, cfg.weight_decay)
        self.opt_d = AdamW(self.d, cfg.lr_d, cfg.betas, cfg.eps, cfg.weight_decay)
        # latent classifier (train.py:153-154, optimizer :192 = torch.optim.Adam(lr_d, adam_beta): no weight decay)
        self.c = None
        if cfg.lambda_latcls != 0:
            if sd_c is None:
                raise ValueError('lambda_latcls != 0 needs the latent classifier state_dict')
            self.c = {k: v.clone().requires_grad_(True) for k, v in sd_c.items()}
            self.opt_c = AdamW(self.c, cfg.lr_d, cfg.betas, cfg.eps, 0.0)

    # -- generator / content encoder, conv or SSL-conditioned ------------------------
    def _ssl_features(self, x):
        with torch.no_grad():      # model/ssl_encoder.py:141-145
            c = self.ssl.extract_features(torch.nn.functional.pad(x, (160, 0)).squeeze(1))[0]
        return c.transpose(1, 2)

    def _generator(self, x, c, c_var):
        if self.ssl is None:
            return M.generator(self.g, x, c, c_var)
        return M.generator_ssl(self.g, self._ssl_features(x), c, c_var)

    def _encoder(self, x):
        if self.ssl is None:
            return M.encoder(self.g, x)
        return M.ssl_content_encoder(self.g, self._ssl_features(x))

    # -- D-step ---------------------------------------------------------------------
    def d_losses(self, batch):
        with torch.no_grad():  # G is not updated by the D-step (Q3)
            fake, fake_subs, _ = self._generator(batch['signal_real'], batch['c_tgt'], batch['c_f0_conv'])
        real_subs = M.disc_subsamples(batch['signal_real'])
        out_real, _ = M.discriminator(self.d, batch['signal_real'], batch['label_src'], real_subs)
        out_fake, _ = M.discriminator(self.d, fake, batch['label_tgt'], fake_subs)
        l_real, l_fake = L.lsgan_to_one(out_real), L.lsgan_to_zero(out_fake)
        return dict(D_loss_adv_real=l_real, D_loss_adv_fake=l_fake, D_loss=l_real + l_fake)

    # -- G-step ---------------------------------------------------------------------
    def g_losses(self, batch, idx_x, idx_y):
        c = self.cfg
        x = batch['signal_real']
        fake, fake_subs, emb_real = self._generator(x, batch['c_tgt'], batch['c_f0_conv'])
        out_fake, feats_fake = M.discriminator(self.d, fake, batch['label_tgt'], fake_subs)
        adv = L.lsgan_to_one(out_fake)
        out = dict(G_loss_adv_fake=adv)
        total = adv
        # train.py:333-341: the loss target is the real signal rolled by a per-sample random shift when jitter_amp > 0
        # (util.audio.add_jitter -> util.roll_batches: y[b, :, t] = x[b, :, (t - shift_b) mod T]); the draw is an input here
        xt = x
        if c.jitter_amp > 0 and (c.lambda_rec > 0 or c.lambda_idt > 0):
            xt = torch.stack([torch.roll(x[b_], int(batch['jitter'][b_]), dims=-1) for b_ in range(x.shape[0])])
        feats_real = None
        if (c.lambda_rec > 0 or c.lambda_idt > 0) and c.lambda_feat > 0:
            with torch.no_grad():  # only used detached (Q6)
                _, feats_real = M.discriminator(self.d, xt, batch['label_src'], M.disc_subsamples(xt))
        wave_idt = None
        if not c.no_conv and c.lambda_rec > 0:
            # cycle reconstruction (train.py:344-361): the converted signal, detached, converted back to the source speaker
            rec, rec_subs, _ = self._generator(fake.detach(), batch['c_src'], batch['c_f0_src'])
            rec_loss = 0
            if c.lambda_feat > 0:
                _, feats_rec = M.discriminator(self.d, rec, batch['label_src'], rec_subs)
                out['G_loss_rec_feat'] = L.feature_matching(feats_rec, feats_real)
                rec_loss = rec_loss + c.lambda_feat * out['G_loss_rec_feat']
            if c.lambda_spec > 0:
                out['G_loss_rec_spec'] = L.log_mel_l1(rec, xt, c.fft_sizes)
                rec_loss = rec_loss + c.lambda_spec * out['G_loss_rec_spec']
            if c.lambda_wave > 0:       # train.py:357-360
                out['G_loss_rec_wave'] = (x - rec).abs().mean()
                rec_loss = rec_loss + c.lambda_wave * out['G_loss_rec_wave']
            out['G_loss_rec'] = rec_loss
            total = total + c.lambda_rec * rec_loss
        if c.lambda_idt > 0:
            if c.no_conv:
                idt, idt_subs, feats_idt_src = fake, fake_subs, None
            else:
                idt, idt_subs, _ = self._generator(x, batch['c_src'], batch['c_f0_src'])
            idt_loss = 0
            if c.lambda_feat > 0:
                _, feats_idt = M.discriminator(self.d, idt, batch['label_src'], idt_subs)
                out['G_loss_idt_feat'] = L.feature_matching(feats_idt, feats_real)
                idt_loss = idt_loss + c.lambda_feat * out['G_loss_idt_feat']
            if c.lambda_spec > 0:
                out['G_loss_idt_spec'] = L.log_mel_l1(idt, xt, c.fft_sizes)
                idt_loss = idt_loss + c.lambda_spec * out['G_loss_idt_spec']
            if c.lambda_wave > 0:       # train.py:381-384: `g_loss_rec += lambda_wave * ...` -> weight lambda_rec * lambda_wave in the total
                out['G_loss_idt_wave'] = wave_idt = (x - idt).abs().mean()
            out['G_loss_idt'] = idt_loss
            total = total + c.lambda_idt * idt_loss
        if wave_idt is not None and c.lambda_rec != 0:
            total = total + (c.lambda_rec * c.lambda_wave) * wave_idt
        # lambda_converted (train.py:409-413) accumulates its term into itself, never into the loss: nothing to add
        if self.c is not None:      # train.py:420-422, :480 — gradient-reversed into the encoder
            out['G_loss_lat_cls'] = torch.nn.functional.cross_entropy(M.latent_classifier(self.c, emb_real), batch['label_src'])
            total = total + c.lambda_latcls * out['G_loss_lat_cls']
        if c.lambda_cont_emb > 0 and c.lambda_corrupted:
            emb_cor = self._encoder(batch['signal_corrupted'])
            out['G_loss_cont_emb'] = L.contrastive(emb_real, emb_cor, idx_x, idx_y)
            total = total + c.lambda_cont_emb * out['G_loss_cont_emb']
        out['G_loss'] = total
        return out

    def _clip(self, params, max_norm):
        if max_norm is not None:      # train.py:289-290, 489-490
            torch.nn.utils.clip_grad_norm_([p for p in params.values() if p.grad is not None], max_norm)

    def run(self, batch, idx_x, idx_y):
        """One full iteration; returns {name: float} of every logged scalar."""
        c, it = self.cfg, self.iter_count
        self.iter_count += 1
        dl, cl, gl = {}, {}, {}
        if it % c.d_step_interval == 0:      # train.py:259
            self.opt_d.zero_grad()
            dl = self.d_losses(batch)
            dl['D_loss'].backward()
            self._clip(self.d, c.grad_max_norm_d)
            self.opt_d.step()
            self.opt_d.zero_grad()
            if self.c is not None:      # latent-classifier step (train.py:300-308) on the detached content embedding:
                with torch.no_grad():   # its gradient into G is dead work (G is not stepped here, grads are zeroed before the G-step)
                    emb = self._encoder(batch['signal_real'])
                self.opt_c.zero_grad()
                logits = M.latent_classifier(self.c, emb)
                cl['C_loss'] = torch.nn.functional.cross_entropy(logits, batch['label_src'])
                cl['C_loss'].backward()
                self.opt_c.step()
                self.opt_c.zero_grad()
        if it % c.g_step_interval == 0:      # train.py:320
            self.opt_g.zero_grad()
            for p in list(self.d.values()) + (list(self.c.values()) if self.c is not None else []):
                p.requires_grad_(False)      # D / C grads from the G-step are dead work (Q5): not computed
            try:
                gl = self.g_losses(batch, idx_x, idx_y)
                gl['G_loss'].backward()
            finally:
                for p in list(self.d.values()) + (list(self.c.values()) if self.c is not None else []):
                    p.requires_grad_(True)
            self._clip(self.g, c.grad_max_norm_g)
            self.opt_g.step()
        return {k: float(v.detach()) for k, v in {**dl, **cl, **gl}.items()}
