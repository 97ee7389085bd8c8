"""ORACLE tooling — BUILD container only (imports /root/reference). Golden iteration WITH the latent classifier
(SURVEY §8f-2): reference Generator / Discriminator / LatentClassifier modules driven with the loop body of train.py
(:259-316 D-step, :300-308 classifier step with torch.optim.Adam, :320-491 G-step incl. :420-422 / :480 the
gradient-reversed classification term), conv_enc-stage1 with lambda_latcls = 1, B=2, T=8960, 2 iterations. Stores loss
scalars and parameter checksums (tests/golden/step_latcls.json), sampled post-update parameters (step_latcls_update.npz) and pins oracle/step.py against it.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_latcls
"""
import json
import sys

import torch
import yaml

from oracle import make_golden as MG
from oracle import losses as OL, step as OS

synth = MG.synth
OUT, REF = MG.OUT, MG.REF


def main():
    torch.manual_seed(0)
    Gc, Dc, _, LatCls, RL, _ = MG.import_reference()
    G, D, C = Gc(**MG.G_ARGS), Dc(**MG.D_ARGS), LatCls(16, 128)
    sd_g, sd_d, sd_c = MG.load_filled(G), MG.load_filled(D), MG.load_filled(C)
    docs = {}
    for d in yaml.safe_load_all(open(f'{REF}/config/conv_enc-stage1.yaml')):
        docs.update(d)
    docs['train']['lambda_latcls'] = 1.0
    cfg = OS.StepConfig.from_hparams(docs['train'])
    B, T, iters = 2, 8960, 2
    opt_g = torch.optim.AdamW(G.parameters(), cfg.lr_g, cfg.betas)
    opt_d = torch.optim.AdamW(D.parameters(), cfg.lr_d, cfg.betas)
    opt_c = torch.optim.Adam(C.parameters(), cfg.lr_d, cfg.betas)
    ost = OS.TrainStep(sd_g, sd_d, cfg, sd_c)
    bt = synth.make_batch(B, T, seed=1234, conversion=True)
    Tn = T // 320
    F = torch.nn.functional
    log, pin = [], {}
    for it in range(iters):
        ix = synth.contrastive_indices(B, Tn, cfg.n_neg, seed=100 + 2 * it)
        iy = synth.contrastive_indices(B, Tn, cfg.n_neg, seed=101 + 2 * it)
        fake, fsubs = G(bt['signal_real'], bt['c_tgt'], c_var=bt['c_f0_conv'], out_subsample=True)
        emb_d = G.content_embedding
        rsubs = D.get_subsamples(bt['signal_real'])
        o_r, _ = D(bt['signal_real'], bt['label_src'], rsubs)
        o_f, _ = D(fake.detach(), bt['label_tgt'], [s.detach() for s in fsubs])
        l_r = sum(((o - 1) ** 2).mean() for o in o_r); l_f = sum((o ** 2).mean() for o in o_f)
        opt_d.zero_grad(); (l_r + l_f).backward(); opt_d.step()
        c_loss = F.cross_entropy(C(emb_d), bt['label_src'])                       # train.py:300-308
        opt_c.zero_grad(); c_loss.backward(); opt_c.step()
        fake, fsubs = G(bt['signal_real'], bt['c_tgt'], c_var=bt['c_f0_conv'], out_subsample=True)
        emb_real = G.content_embedding.clone()
        o_f, _ = D(fake, bt['label_tgt'], fsubs)
        adv = sum(((o - 1) ** 2).mean() for o in o_f)
        _, f_real = D(bt['signal_real'], bt['label_src'], D.get_subsamples(bt['signal_real']))
        idt, isubs = G(bt['signal_real'], bt['c_src'], c_var=bt['c_f0_src'], out_subsample=True)
        _, f_idt = D(idt, bt['label_src'], isubs)
        l_feat = RL.multiscale_feat_loss(f_idt, f_real, norm_p=1)
        l_spec = OL.log_mel_l1(idt, bt['signal_real'], cfg.fft_sizes)
        l_idt = cfg.lambda_feat * l_feat + cfg.lambda_spec * l_spec
        l_cls = F.cross_entropy(C(emb_real), bt['label_src'])                     # train.py:420-422
        emb_cor = G.encoder(bt['signal_corrupted'])
        l_con = OL.contrastive(emb_real, emb_cor, ix, iy)
        g_loss = adv + cfg.lambda_idt * l_idt + cfg.lambda_latcls * l_cls + cfg.lambda_cont_emb * l_con
        opt_d.zero_grad(); opt_c.zero_grad(); opt_g.zero_grad(); g_loss.backward(); opt_g.step()
        ref_log = dict(D_loss_adv_real=float(l_r), D_loss_adv_fake=float(l_f), D_loss=float(l_r + l_f), C_loss=float(c_loss),
                       G_loss_adv_fake=float(adv), G_loss_idt_feat=float(l_feat), G_loss_idt_spec=float(l_spec),
                       G_loss_idt=float(l_idt), G_loss_lat_cls=float(l_cls), G_loss_cont_emb=float(l_con), G_loss=float(g_loss))
        ora = ost.run(bt, ix, iy)
        pin[f'it{it}'] = {k: abs(ora[k] - v) / (abs(v) + 1e-12) for k, v in ref_log.items()}
        log.append(ref_log)
    pin['params'] = dict(G=max(MG.rel(ost.g[k], p) for k, p in G.state_dict().items()),
                         D=max(MG.rel(ost.d[k], p) for k, p in D.state_dict().items()),
                         C=max(MG.rel(ost.c[k], p) for k, p in C.state_dict().items()))
    chk = lambda sd: {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items()}
    json.dump(dict(config='conv_enc-stage1 + lambda_latcls=1', B=B, T=T, iters=iters, losses=log, params_G=chk(G.state_dict()),
                   params_D=chk(D.state_dict()), params_C=chk(C.state_dict())), open(f'{OUT}/step_latcls.json', 'w'))
    # sampled post-update parameter values + per-tensor update norms, as for the other step fixtures (make_golden.py): the
    # UPDATE is what a test can observe (an lr = 1e-4 step moves sum|p| by ~1e-5 relative)
    import numpy as np
    upd = {}
    for tag_, sd_after, sd_before in (('G', G.state_dict(), sd_g), ('D', D.state_dict(), sd_d), ('C', C.state_dict(), sd_c)):
        for k, v in sd_after.items():
            idx = MG.param_sample_idx(k, v.numel())
            upd[f'{tag_}/{k}'] = v.detach().reshape(-1)[idx].numpy().astype(np.float32)
            upd[f'{tag_}/{k}/dnorm'] = np.float64((v.detach().double() - sd_before[k].double()).norm())
    np.savez_compressed(f'{OUT}/step_latcls_update.npz', **upd)
    allpin = json.load(open(f'{OUT}/PINNING.json'))
    allpin['step_latcls'] = pin
    json.dump(allpin, open(f'{OUT}/PINNING.json', 'w'), indent=1)
    print(json.dumps(pin, indent=1))


if __name__ == '__main__':
    main()
