"""One TD-VC-GAN training iteration (D-step + G-step, both AdamW updates) on the HIP path.

Mirrors the loop body of the reference's train.py:209-521 (D-step :259-316, G-step :320-510,
loss assembly :477-482, optimizers :188-189; cycle-reconstruction branch :344-361 when lambda_rec > 0)
for the configuration space of the shipped YAMLs,
minus the CREPE-backed F0 term (:429-470, torchcrepe unavailable: SURVEY §8c) and minus work
that reaches no parameter update:
  Q3  the D-step's generator forward runs without a graph (the reference back-propagates into G
      through the sub-scale heads and then discards those gradients);
  Q4  the G-step reuses the D-step's generator forward (identical inputs and weights) instead of
      recomputing it — bit-identical by construction, `reuse_fake=False` recomputes like the reference;
  Q5  discriminator weight gradients are not computed in the G-step;
  Q6  D(real) feature maps of the G-step run forward-only.
Scalar names follow the reference's tensorboard tags (D_loss_adv_real, G_loss_idt_feat, ...).
"""
from dataclasses import dataclass

import torch

from . import losses as LS
from .arena import FlatAdamW


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
    lambda_f0: float = 0.0
    lr_g: float = 1e-4
    lr_d: float = 1e-4
    betas: tuple = (0.8, 0.99)
    eps: float = 1e-8
    weight_decay: float = 1e-2
    n_neg: int = 100
    fft_sizes: tuple = (2048, 1024, 512)
    # loop switches that every shipped YAML leaves off; semantics = the reference's, file:line at each use
    lambda_wave: float = 0.0            # train.py:357-360, 381-384: waveform L1 terms
    lambda_converted: float = 0.0       # train.py:409-413: accepted, reaches no loss in the reference (see _g_fwd_bwd)
    jitter_amp: int = 0                 # train.py:335-336: the real signal rolled by a random per-sample shift as the loss target
    grad_max_norm_d: float = None       # train.py:289-290: clip_grad_norm_ on D before its optimizer step
    grad_max_norm_g: float = None       # train.py:489-490
    d_step_interval: int = 1            # train.py:259
    g_step_interval: int = 1            # train.py:320
    freeze_subnets: tuple = ()          # train.py:195-197: ('encoder',) -> G.encoder parameters get requires_grad = False

    @staticmethod
    def from_hparams(train: dict) -> 'StepConfig':
        """Build from the `train` document of a reference YAML (config/*.yaml). `lambda_f0` (the CREPE-backed F0 term,
        train.py:429-470) is excluded by contract (torchcrepe is not available: SURVEY §8c) and only warns; a sub-network
        name in `freeze_subnets` other than 'encoder' is ignored by the reference too (train.py:195)."""
        g = train.get
        if float(g('lambda_f0', 0) or 0) != 0:
            import warnings
            warnings.warn('lambda_f0 != 0: the CREPE-backed F0 loss term (train.py:429-470) is not part of this path '
                          '(torchcrepe unavailable); the iteration runs without it', stacklevel=2)
        return StepConfig(no_conv=bool(g('no_conv', False)), lambda_rec=float(g('lambda_rec', 0)),
                          lambda_idt=float(g('lambda_idt', 0)), lambda_feat=float(g('lambda_feat', 0)),
                          lambda_spec=float(g('lambda_spec', 0)), lambda_cont_emb=float(g('lambda_cont_emb', 0)),
                          lambda_corrupted=float(g('lambda_corrupted', 0)), lambda_latcls=float(g('lambda_latcls', 0)),
                          lambda_f0=float(g('lambda_f0', 0)),
                          lr_g=float(g('lr_g', 1e-4)), lr_d=float(g('lr_d', 1e-4)),
                          betas=tuple(g('adam_beta', (0.8, 0.99))),
                          lambda_wave=float(g('lambda_wave', 0) or 0), lambda_converted=float(g('lambda_converted', 0) or 0),
                          jitter_amp=int(g('jitter_amp', 0) or 0),
                          grad_max_norm_d=(float(g('grad_max_norm_D')) if g('grad_max_norm_D') is not None else None),
                          grad_max_norm_g=(float(g('grad_max_norm_G')) if g('grad_max_norm_G') is not None else None),
                          d_step_interval=int(g('D_step_interval', 1) or 1), g_step_interval=int(g('G_step_interval', 1) or 1),
                          freeze_subnets=tuple(g('freeze_subnets') or ()))


class TrainStep:
    def __init__(self, G, D, cfg: StepConfig, device, reuse_fake=True, grad_sync=None, C=None):
        self.G, self.D, self.cfg, self.device = G, D, cfg, torch.device(device)
        self.reuse_fake = reuse_fake
        self.grad_sync = grad_sync            # parallel.GradSync or None
        self.comm_events = None               # set to [] to record (backward done, all-reduce joined) event pairs per optimizer step
        self.iter_count = 0                   # train.py:211: the step intervals count iterations from 0
        if 'encoder' in cfg.freeze_subnets:   # train.py:195-197
            for p_ in G.encoder.parameters():
                p_.requires_grad = False
        ga, da = G.ensure_arena(self.device), D.ensure_arena(self.device)
        self.opt_g = FlatAdamW(ga, cfg.lr_g, cfg.betas, cfg.eps, cfg.weight_decay)
        self.opt_d = FlatAdamW(da, cfg.lr_d, cfg.betas, cfg.eps, cfg.weight_decay)
        G.weights_frozen(True); D.weights_frozen(True)     # effective weights are rebuilt right after each update
        ga.materialize(); da.materialize()
        if grad_sync is not None:     # completed gradient segments go to the all-reduce while backward continues
            grad_sync.attach(ga); grad_sync.attach(da)
        # latent classifier (train.py:153-154; optimizer :192 is torch.optim.Adam(lr_d, adam_beta): no weight decay)
        self.C = None
        if cfg.lambda_latcls != 0:
            if C is None:
                raise ValueError('lambda_latcls != 0 needs the LatentClassifier module')
            self.C = C
            ca = C.ensure_arena(self.device)
            self.opt_c = FlatAdamW(ca, cfg.lr_d, cfg.betas, cfg.eps, 0.0)
            C.weights_frozen(True)
            ca.materialize()
            if grad_sync is not None:
                grad_sync.attach(ca)

    # -------------------------------------------------------------------------------------------
    def _generate(self, batch):
        """All generator work of the iteration in one batched pass (G's weights do not change between the D-step and
        the G-step): encoder on [real; corrupted], decoder on [target-conditioned; identity-conditioned]."""
        from .modules import generator_forward_pair
        c = self.cfg
        want_idt = c.lambda_idt > 0 and not c.no_conv
        want_cor = c.lambda_cont_emb > 0 and bool(c.lambda_corrupted)
        conds = [batch['c_tgt']] + ([batch['c_src']] if want_idt else [])
        cvars = [batch['c_f0_conv']] + ([batch['c_f0_src']] if want_idt else [])
        outs, emb_real, emb_cor = generator_forward_pair(self.G, batch['signal_real'], batch['signal_corrupted'] if want_cor else None,
                                                         conds, cvars)
        fake = outs[0]
        idt = outs[1] if want_idt else (fake if c.lambda_idt > 0 else None)
        self._gen_full = outs.full if want_idt else None      # (y, subs) over [target; identity]: the D input of the G-step as is
        return fake, idt, emb_real, emb_cor

    def d_step(self, batch, log):
        self._d_fwd_bwd(batch, log)       # data-parallel: segment all-reduces were issued during / at the end of backward
        self._join_sync(self.D.arena)
        self._d_update()

    def _join_sync(self, arena):
        """Before an optimizer step of a data-parallel run: the compute stream waits for the arena's gradient all-reduces (side
        stream). With `comm_events` set, the wait is bracketed by two events on the compute stream: their distance is the time
        the stream sat idle for communication that the backward pass did not cover (the exposed part of the exchange)."""
        if self.grad_sync is None:
            return
        if self.comm_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.grad_sync.wait(arena)
            e1.record()
            self.comm_events.append((e0, e1))
        else:
            self.grad_sync.wait(arena)

    def _d_update(self):
        self.opt_d.step(grad_scale=1.0 if self.grad_sync is None else self.grad_sync.scale, max_norm=self.cfg.grad_max_norm_d)
        self.D.arena.materialize()

    def _d_fwd_bwd(self, batch, log):
        D = self.D
        if self.reuse_fake:
            self._gen_out = self._generate(batch)
            (fake, fake_subs) = self._gen_out[0]
        else:
            with torch.no_grad():
                gen = self._generate(batch)
                (fake, fake_subs) = gen[0]
                self._emb_for_c = gen[2]
        real = batch['signal_real']
        B = real.shape[0]
        self._real_subs = D.get_subsamples(real)
        # real and fake in ONE discriminator call (batch 2B): exact, D has no cross-sample op
        x2 = torch.cat([real, fake.detach()], dim=0)
        l2 = torch.cat([batch['label_src'], batch['label_tgt']], dim=0)
        s2 = [torch.cat([r, f.detach()], dim=0) for r, f in zip(self._real_subs, fake_subs)]
        outs, _ = D(x2, l2, s2, views=True)
        l_real, l_fake = LS.lsgan_split(outs, B)          # both halves of the batched outputs, gradients written in place
        d_loss = l_real + l_fake
        self.opt_d.zero_grad()
        d_loss.backward()
        log.update(D_loss_adv_real=l_real.detach(), D_loss_adv_fake=l_fake.detach(), D_loss=d_loss.detach())

    def g_step(self, batch, log, idx_x=None, idx_y=None):
        self._g_fwd_bwd(batch, log, idx_x, idx_y)
        self._join_sync(self.G.arena)
        self._g_update()

    def _g_update(self):
        self.opt_g.step(grad_scale=1.0 if self.grad_sync is None else self.grad_sync.scale, max_norm=self.cfg.grad_max_norm_g)
        self.G.arena.materialize()

    def _g_fwd_bwd(self, batch, log, idx_x=None, idx_y=None):
        G, D, c = self.G, self.D, self.cfg
        real = batch['signal_real']
        B = real.shape[0]
        D.arena.wgrad_enabled = False                     # Q5
        if self.C is not None:
            self.C.arena.wgrad_enabled = False            # the classifier's own grads from the G-step are dead work too
        try:
            if self.reuse_fake and getattr(self, '_gen_out', None) is not None:
                (fake, fake_subs), idt_pair, emb_real, emb_cor = self._gen_out
                self._gen_out = None
            else:       # reuse off, or no D-step ran in this iteration (D_step_interval > 1)
                (fake, fake_subs), idt_pair, emb_real, emb_cor = self._generate(batch)
                self._real_subs = D.get_subsamples(real)
            need_feat = c.lambda_idt > 0 and c.lambda_feat > 0
            separate_idt = need_feat and idt_pair is not None and idt_pair[0] is not fake
            want_rec = c.lambda_rec > 0 and not c.no_conv
            rec = rec_subs = feats_rec = None
            if want_rec:
                # cycle reconstruction (train.py:344-361): G(fake.detach(), c_src, c_var=c_f0_src) -- a second, sequentially
                # dependent generator pass; its encoder sees the converted signal, its decoder the source conditioning
                from .modules import generator_forward_pair
                (rec_pair,), _, _ = generator_forward_pair(G, fake.detach(), None, [batch['c_src']], [batch['c_f0_src']])
                rec, rec_subs = rec_pair
            if separate_idt or (want_rec and c.lambda_feat > 0):
                # fake, identity and reconstructed signals through D in ONE call (batch 2B / 3B): exact, D has no cross-sample op
                sigs, subs, labels = [fake], [fake_subs], [batch['label_tgt']]
                if separate_idt:
                    sigs.append(idt_pair[0]); subs.append(idt_pair[1]); labels.append(batch['label_src'])
                    full = getattr(self, '_gen_full', None)
                    if full is not None and self.reuse_fake:      # [fake; idt] IS the decoder's batched output: no re-concatenation
                        sigs, subs = [full[0]], [full[1]]
                if want_rec and c.lambda_feat > 0:
                    sigs.append(rec); subs.append(rec_subs); labels.append(batch['label_src'])
                cat = lambda ts: ts[0] if len(ts) == 1 else torch.cat(ts, dim=0)
                outs, feats = D(cat(sigs), torch.cat(labels, dim=0), [cat([s_[i] for s_ in subs]) for i in range(len(fake_subs))], views=True)
                # the loss terms read their sample range of the BATCHED outputs / feature maps (losses.BatchView)
                out_fake, adv_rng = outs, (0, B)
                pos = 1
                idt, idt_subs = (fake, fake_subs) if idt_pair is not None else (None, None)
                feats_idt = feats_rec = None
                idt_rng = rec_rng = None
                if separate_idt:
                    idt, idt_subs = idt_pair
                    feats_idt, idt_rng = feats, (pos * B, (pos + 1) * B)
                    pos += 1
                if want_rec and c.lambda_feat > 0:
                    feats_rec, rec_rng = feats, (pos * B, (pos + 1) * B)
            else:
                adv_rng = idt_rng = rec_rng = None
                out_fake, feats_fake = D(fake, batch['label_tgt'], fake_subs, views=True)
                idt, idt_subs = (fake, fake_subs) if idt_pair is not None else (None, None)
                # no_conv: the identity signal IS the converted signal, but it is judged with label_src (train.py:374)
                feats_idt = None
                if need_feat and idt is not None:
                    # no_conv: label_tgt IS label_src (train.py:217), so D(idt, label_src) == D(fake, label_tgt)
                    if c.no_conv:
                        feats_idt = feats_fake
                    else:
                        _, feats_idt = D(idt, batch['label_src'], idt_subs, views=True)
            adv = LS.lsgan_loss(out_fake, 1.0, rng=adv_rng)
            total = adv
            log['G_loss_adv_fake'] = adv.detach()
            # loss target of the reconstruction / identity terms: the real signal, rolled by a random per-sample shift when
            # jitter_amp > 0 (train.py:333-341, util.audio.add_jitter; the draw comes in as batch['jitter'] for reproducible runs)
            real_t, real_t_subs = real, self._real_subs
            if c.jitter_amp > 0 and (c.lambda_rec > 0 or c.lambda_idt > 0):
                shifts = batch.get('jitter')
                if shifts is None:
                    shifts = torch.randint(-c.jitter_amp, c.jitter_amp + 1, (B,), device=real.device)
                real_t = torch.empty_like(real)
                LS.L.check(LS.L.lib().tdvc_roll_batches(real.data_ptr(), shifts.to(torch.int64).contiguous().data_ptr(), real_t.data_ptr(), B, real.shape[1],
                                                    real.shape[2], torch.cuda.current_stream(real.device).cuda_stream))
                real_t_subs = D.get_subsamples(real_t) if c.lambda_feat > 0 else None
            feats_real = None
            if (need_feat and idt is not None) or feats_rec is not None:
                with torch.no_grad():                      # Q6: D(real) feature maps are only ever used detached
                    _, feats_real = D(real_t, batch['label_src'], real_t_subs, views=True)
            wave_idt = None
            if want_rec:                                   # train.py:344-361
                l_rec = None
                if feats_rec is not None:
                    l_rf = LS.multiscale_feat_loss(feats_rec, feats_real, norm_p=1, rng=rec_rng)
                    log['G_loss_rec_feat'] = l_rf.detach()
                    l_rec = c.lambda_feat * l_rf
                if c.lambda_spec > 0:
                    l_rs = LS.multiscale_spec_loss(rec, real_t, list(c.fft_sizes))
                    log['G_loss_rec_spec'] = l_rs.detach()
                    l_rec = c.lambda_spec * l_rs if l_rec is None else l_rec + c.lambda_spec * l_rs
                if c.lambda_wave > 0:                      # train.py:357-360: mean |real - rec| on the un-jittered signal
                    l_rw = LS.multiscale_feat_loss([[rec]], [[real]], norm_p=1)
                    log['G_loss_rec_wave'] = l_rw.detach()
                    l_rec = c.lambda_wave * l_rw if l_rec is None else l_rec + c.lambda_wave * l_rw
                if l_rec is not None:
                    log['G_loss_rec'] = l_rec.detach()
                    total = total + c.lambda_rec * l_rec
            if c.lambda_idt > 0 and idt is not None:
                l_idt = None
                if need_feat:
                    l_feat = LS.multiscale_feat_loss(feats_idt, feats_real, norm_p=1, rng=idt_rng)
                    log['G_loss_idt_feat'] = l_feat.detach()
                    l_idt = c.lambda_feat * l_feat
                if c.lambda_spec > 0:
                    l_spec = LS.multiscale_spec_loss(idt, real_t, list(c.fft_sizes))
                    log['G_loss_idt_spec'] = l_spec.detach()
                    l_idt = c.lambda_spec * l_spec if l_idt is None else l_idt + c.lambda_spec * l_spec
                if c.lambda_wave > 0:
                    # train.py:381-384 adds the identity waveform term to g_loss_REC (`g_loss_rec += ...`), so it enters the total
                    # with the weight lambda_rec * lambda_wave -- reproduced as is (zero weight when lambda_rec == 0)
                    wave_idt = LS.multiscale_feat_loss([[idt]], [[real]], norm_p=1)
                    log['G_loss_idt_wave'] = wave_idt.detach()
                if l_idt is not None:
                    log['G_loss_idt'] = l_idt.detach()
                    total = total + c.lambda_idt * l_idt
            if wave_idt is not None and c.lambda_rec != 0:
                total = total + (c.lambda_rec * c.lambda_wave) * wave_idt
            # lambda_converted (train.py:409-413): the reference encodes the converted signal and computes a contrastive term, but
            # accumulates it into itself (`g_loss_cont_emb_conv += g_loss_cont_emb_conv`), never into g_loss_cont_emb: it reaches no
            # loss and no gradient. Nothing to compute here.
            if self.C is not None:      # train.py:420-422, :480 -- gradient-reversed into the encoder
                l_cls = LS.cross_entropy_loss(self.C(emb_real), batch['label_src'])
                log['G_loss_lat_cls'] = l_cls.detach()
                total = total + c.lambda_latcls * l_cls
            if emb_cor is not None:
                l_con = LS.contrastive_loss(emb_real, emb_cor, num_negatives=c.n_neg, temp=0.1, idx_x=idx_x, idx_y=idx_y)
                log['G_loss_cont_emb'] = l_con.detach()
                total = total + c.lambda_cont_emb * l_con
            self.opt_g.zero_grad()
            total.backward()
        finally:
            D.arena.wgrad_enabled = True
            if self.C is not None:
                self.C.arena.wgrad_enabled = True
        log['G_loss'] = total.detach()

    def run(self, batch, idx_x=None, idx_y=None):
        """One iteration. Returns {tag: 1-element device tensor}; no host synchronisation inside."""
        log = {}
        c, it = self.cfg, self.iter_count
        self.iter_count += 1
        do_d, do_g = it % c.d_step_interval == 0, it % c.g_step_interval == 0      # train.py:259, :320
        self._gen_out = None
        if do_d:
            self.d_step(batch, log)
            if self.C is not None:      # the classifier step sits inside the D-step block (train.py:298-308)
                self.c_step(batch, log)
        if do_g:
            self.g_step(batch, log, idx_x, idx_y)
        self._gen_out = None
        return log

    def c_step(self, batch, log):
        """Latent-classifier step (train.py:300-308) on the detached content embedding of the D-step's generator forward:
        the gradient it would send into G is dead work (G is not stepped here and its grads are zeroed before the G-step)."""
        C = self.C
        emb = self._gen_out[2] if self.reuse_fake else self._emb_for_c
        logits = C(emb.detach())
        c_loss = LS.cross_entropy_loss(logits, batch['label_src'])
        self.opt_c.zero_grad()
        c_loss.backward()
        self._join_sync(C.arena)
        self.opt_c.step(grad_scale=1.0 if self.grad_sync is None else self.grad_sync.scale)
        C.arena.materialize()
        log['C_loss'] = c_loss.detach()

    def capture(self, batch, idx_x, idx_y, warmup=2):
        """Capture the whole iteration (≈3.5k kernel launches, both backward passes, both AdamW updates) into one
        hipGraph and return `replay() -> log`. The batch tensors are static inputs: copy new data into them
        between replays. Everything the step touches is graph-safe by construction: no host sync, no allocation
        outside the torch caching allocator, wgrad workspace sized during the eager warm-up, AdamW step counter
        on the device. Data-parallel runs stay eager: with the RCCL process group alive its watchdog thread touches HIP
        events during capture (hipErrorStreamCaptureUnsupported), and the eager step measured as fast as the replay
        (72.5 vs 72.2 ms, `bench.py --force-dp --no-graph`), i.e. it is not host-bound."""
        if self.grad_sync is not None:
            raise RuntimeError('graph capture is single-GPU only; run data-parallel steps eagerly')
        if self.cfg.d_step_interval != 1 or self.cfg.g_step_interval != 1:
            raise RuntimeError('graph capture needs D_step_interval = G_step_interval = 1 (iterations would differ in their work)')
        if self.cfg.jitter_amp > 0 and 'jitter' not in batch:
            raise RuntimeError("graph capture with jitter_amp > 0 needs the shifts as a static input: batch['jitter'] (int64 [B])")
        dev = self.device
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.run(batch, idx_x, idx_y)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            log = self.run(batch, idx_x, idx_y)

        def replay():
            graph.replay()
            return log
        replay.graph = graph
        return replay
