"""GPU parity of the drop-in Generator / Discriminator / train step against (a) the committed golden
fixtures generated from the reference itself and (b) the CPU oracle on the same seeded inputs.
Tolerance: 1e-3 relative (north_star), per-tensor rel-L2 for gradients (SURVEY §7 hard part (f))."""
import json
import os

import numpy as np
import pytest
import torch

from common import (GOLDEN, assert_grads_close, assert_update_matches_fixture, build_models, feat_sample_idx, filled_sd, pkg, rel_l2,
                    to_dev)

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope='module')
def models(dev):
    return build_models(dev)


@pytest.mark.parametrize('tag,seed', [('B2_T8960', 7), ('B2_T16000', 1234)])
def test_generator_forward_backward_vs_golden(models, dev, tag, seed):
    G, _ = models
    T = int(tag.split('_T')[1])
    bt = to_dev(pkg().synth.make_batch(2, T, seed=seed), dev)
    gold = np.load(os.path.join(GOLDEN, f'gen_fwd_{tag}.npz'))
    G.arena.zero_grad()
    y, subs = G(bt['signal_real'], bt['c_tgt'], c_var=bt['c_f0_conv'], out_subsample=True)
    emb = G.content_embedding
    errs = dict(y=rel_l2(y, torch.from_numpy(gold['y'])), sub4=rel_l2(subs[0], torch.from_numpy(gold['sub4'])),
                sub2=rel_l2(subs[1], torch.from_numpy(gold['sub2'])), emb=rel_l2(emb, torch.from_numpy(gold['emb'])))
    assert max(errs.values()) < TOL, errs
    # same cotangents as oracle/make_golden.py
    rs = np.random.RandomState(99)
    outs = (y, subs[0], subs[1], emb)
    cot = [torch.from_numpy(rs.randn(*t.shape).astype(np.float32)).to(dev) for t in outs]
    loss = sum((t * c).mean() for t, c in zip(outs, cot))
    loss.backward()
    torch.cuda.synchronize()
    gg = json.load(open(os.path.join(GOLDEN, f'gen_grad_{tag}.json')))
    assert abs(float(loss) - gg['loss']) <= TOL * max(1e-3, abs(gg['loss']))
    errs, samp_bad = {}, {}
    for k, p in G.named_parameters():
        n_ref = gg['norms'][k]
        if n_ref < 0:
            assert p.grad is None, f'{k}: reference leaves grad None (Q7)'
            continue
        assert p.grad is not None, k
        n = float(p.grad.double().norm())
        idx, vals = gg['samples'][k]
        samp = p.grad.reshape(-1)[torch.tensor(idx, device=dev)].double().cpu()
        e_norm = abs(n - n_ref) / (n_ref + 1e-30)
        e_samp = float((samp - torch.tensor(vals)).norm()) / (n_ref / max(1.0, p.numel() ** 0.5) * 2 + 1e-30)
        tol = max(TOL, 5.0 * gg.get('noise', {}).get(k, 0.0))    # fp32 noise floor of this tensor (oracle fp32 vs fp64)
        errs[k] = (e_norm, tol)
        if e_samp > 0.05:
            samp_bad[k] = e_samp
    assert not samp_bad, dict(list(samp_bad.items())[:8])
    assert_grads_close(errs, TOL, f'generator gradients vs golden {tag}', max_outliers=4)       # observed 0 / 732 (r03)


def test_generator_grads_vs_oracle_full(models, dev):
    """Every gradient element of every parameter against the CPU oracle (rel-L2 per tensor)."""
    from oracle import model as OM
    G, _ = models
    bt_cpu = pkg().synth.make_batch(2, 8960, seed=11)
    bt = to_dev(bt_cpu, dev)
    sg = {k: v.clone().requires_grad_(True) for k, v in filled_sd('G').items()}
    oy, osubs, oemb = OM.generator(sg, bt_cpu['signal_real'], bt_cpu['c_tgt'], bt_cpu['c_f0_conv'])
    rs = np.random.RandomState(5)
    cot = [torch.from_numpy(rs.randn(*t.shape).astype(np.float32)) for t in (oy, osubs[0], osubs[1], oemb)]
    sum((t * c).mean() for t, c in zip((oy, osubs[0], osubs[1], oemb), cot)).backward()
    G.arena.zero_grad()
    y, subs = G(bt['signal_real'], bt['c_tgt'], c_var=bt['c_f0_conv'], out_subsample=True)
    outs = (y, subs[0], subs[1], G.content_embedding)
    sum((t * c.to(dev)).mean() for t, c in zip(outs, cot)).backward()
    torch.cuda.synchronize()
    errs = {k: rel_l2(p.grad, sg[k].grad) for k, p in G.named_parameters() if p.grad is not None}
    assert_grads_close(errs, TOL, 'generator gradients vs oracle', max_outliers=4)                # observed 0 / 732 (r03)
    assert rel_l2(y, oy) < TOL


@pytest.mark.parametrize('tag,seed', [('B2_T8960', 7), ('B2_T16000', 1234)])
def test_discriminator_vs_golden(models, dev, tag, seed):
    _, D = models
    LS = pkg().losses
    T = int(tag.split('_T')[1])
    bt = to_dev(pkg().synth.make_batch(2, T, seed=seed), dev)
    gold = np.load(os.path.join(GOLDEN, f'disc_{tag}.npz'))
    gj = json.load(open(os.path.join(GOLDEN, f'disc_{tag}.json')))
    gen = np.load(os.path.join(GOLDEN, f'gen_fwd_{tag}.npz'))
    fake = torch.from_numpy(gen['y']).to(dev)
    fsubs = [torch.from_numpy(gen['sub4']).to(dev), torch.from_numpy(gen['sub2']).to(dev)]
    rsubs = D.get_subsamples(bt['signal_real'])
    for i, s in enumerate(rsubs):
        assert rel_l2(s, torch.from_numpy(gold[f'sub_real_{i}'])) < TOL
    D.arena.zero_grad()
    o_r, f_r = D(bt['signal_real'], bt['label_src'], rsubs)
    o_f, f_f = D(fake, bt['label_tgt'], fsubs)
    for i in range(5):
        assert rel_l2(o_r[i], torch.from_numpy(gold[f'out_real_{i}'])) < TOL, ('real', i)
        assert rel_l2(o_f[i], torch.from_numpy(gold[f'out_fake_{i}'])) < TOL, ('fake', i)
    for fl, ref in ((f_r, gj['feat_stats_real']), (f_f, gj['feat_stats_fake'])):
        for p_, (maps, stats) in enumerate(zip(fl, ref)):
            for m, st in zip(maps, stats):
                md = m.double()
                got = [float(md.mean()), float(md.abs().mean()), float(md.std())]
                assert abs(got[1] - st[1]) <= TOL * abs(st[1]) and abs(got[2] - st[2]) <= TOL * abs(st[2]), (p_, got, st)
    # sampled elements of every feature map (5 passes x 6 maps, real and fake), element-wise against the reference
    for fl, key in ((f_r, 'feat_samples_real'), (f_f, 'feat_samples_fake')):
        for p_, maps in enumerate(fl):
            for m_, m in enumerate(maps):
                vals, rms = gj[key][p_][m_]
                got = m.reshape(-1)[torch.from_numpy(feat_sample_idx(p_, m_, m.numel())).to(dev)].double().cpu()
                err = float((got - torch.tensor(vals, dtype=torch.float64)).abs().max())
                assert err <= TOL * rms, (key, p_, m_, err, rms)
    loss = LS.lsgan_loss(o_r, 1.0) + LS.lsgan_loss(o_f, 0.0)
    assert abs(float(loss) - gj['loss']) <= TOL * abs(gj['loss'])
    loss.backward()
    torch.cuda.synchronize()
    bad = {}
    for k, p in D.named_parameters():
        n_ref = gj['norms'][k]
        e = abs(float(p.grad.double().norm()) - n_ref) / (n_ref + 1e-30)
        if e > max(TOL, 5.0 * gj.get('noise', {}).get(k, 0.0)):
            bad[k] = e
    assert not bad, bad


@pytest.mark.parametrize('cfg_name,T', [('conv_enc-stage1', 8960), ('conv_enc-stage2_1', 8960), ('conv_enc-stage1', 16000),
                                        ('conv_enc-stage2_2', 8960)])
def test_train_step_vs_golden(dev, cfg_name, T):
    """Full iteration(s): every logged loss scalar vs the fixture produced by the reference's own modules
    + torch.optim.AdamW; post-update parameters vs the fixture checksums."""
    P = pkg()
    gold = json.load(open(os.path.join(GOLDEN, f'step_{cfg_name}_T{T}.json')))
    hp = P.hparams.HParam(os.path.join(os.path.dirname(GOLDEN), '..', 'config', f'{cfg_name}.yaml'))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')          # lambda_f0: the CREPE-backed term is excluded by contract
        cfg = P.train_step.StepConfig.from_hparams(hp.train)
    G, D = build_models(dev)
    ts = P.train_step.TrainStep(G, D, cfg, dev)
    bt = to_dev(P.synth.make_batch(gold['B'], T, seed=1234, conversion=not cfg.no_conv), dev)
    Tn = T // 320
    for it, ref in enumerate(gold['losses']):
        ix = P.synth.contrastive_indices(gold['B'], Tn, cfg.n_neg, seed=100 + 2 * it)
        iy = P.synth.contrastive_indices(gold['B'], Tn, cfg.n_neg, seed=101 + 2 * it)
        log = ts.run(bt, ix, iy)
        torch.cuda.synchronize()
        errs = {k: abs(float(log[k]) - v) / (abs(v) + 1e-12) for k, v in ref.items()}
        assert max(errs.values()) < TOL, (it, errs)
    assert len(gold['losses']) >= 2, 'fixture must hold >= 2 iterations: the second one observes the first update'
    # the parameter UPDATE of the iterations above against the reference's (sampled elements + per-tensor update norms)
    upd = np.load(os.path.join(GOLDEN, f'step_{cfg_name}_T{T}_update.npz'))
    assert_update_matches_fixture(dict(G=G, D=D), upd, dict(G=filled_sd('G'), D=filled_sd('D')), cfg.lr_g, f'{cfg_name} T={T}')


def test_split_conditioning_equals_dense(models, dev):
    """The exact algebraic split of cond_var.0 (time-constant speaker channels on a length-3 signal + 8-channel
    excitation conv) against the reference's dense 136-channel formulation, forward and parameter gradients."""
    G, _ = models
    bt = to_dev(pkg().synth.make_batch(2, 8960, seed=21), dev)
    res = {}
    for split in (False, True):
        G.decoder.split_cond = split
        G.arena.zero_grad()
        y, subs = G(bt['signal_real'], bt['c_tgt'], c_var=bt['c_f0_conv'], out_subsample=True)
        (y.square().mean() + subs[0].mean() + subs[1].square().mean()).backward()
        torch.cuda.synchronize()
        res[split] = (y.detach().clone(), {k: p.grad.detach().clone() for k, p in G.named_parameters() if p.grad is not None})
    G.decoder.split_cond = True
    assert rel_l2(res[True][0], res[False][0]) < 1e-5
    errs = {k: rel_l2(res[True][1][k], g) for k, g in res[False][1].items()}
    assert_grads_close(errs, 2e-4, 'gradients, split vs dense conditioning', max_outliers=22)       # observed 11 / 732 (r03): one kink flip, see common.py


def test_fused_conditioning_equals_unfused(models, dev):
    """tdvc_film_cond_fwd (cond_var.0 evaluated inside cond_var.2's forward kernel) against the two-launch chain."""
    G, _ = models
    M = pkg().modules
    bt = to_dev(pkg().synth.make_batch(2, 8960, seed=22), dev)
    res = {}
    try:
        for fused in (False, True):
            M.FUSED_COND = fused
            G.arena.zero_grad()
            y, subs = G(bt['signal_real'], bt['c_tgt'], c_var=bt['c_f0_conv'], out_subsample=True)
            (y.square().mean() + subs[0].mean() + subs[1].square().mean()).backward()
            torch.cuda.synchronize()
            res[fused] = (y.detach().clone(), {k: p.grad.detach().clone() for k, p in G.named_parameters() if p.grad is not None})
    finally:
        M.FUSED_COND = True
    assert rel_l2(res[True][0], res[False][0]) < 1e-5
    errs = {k: rel_l2(res[True][1][k], g) for k, g in res[False][1].items()}
    assert_grads_close(errs, 2e-4, 'gradients, fused vs chained conditioning ops', max_outliers=4)    # observed 0 / 732 (r03)


def test_graph_replay_matches_eager(dev):
    """The whole iteration captured into one hipGraph replays to the same losses / parameters as eager launches."""
    P = pkg()
    cfg = P.train_step.StepConfig()
    bt = to_dev(P.synth.make_batch(2, 8960, seed=5), dev)
    ix = P.synth.contrastive_indices(2, 28, cfg.n_neg, 1).to(dev)
    iy = P.synth.contrastive_indices(2, 28, cfg.n_neg, 2).to(dev)
    res = {}
    for mode in ('eager', 'graph'):
        G, D = build_models(dev)
        ts = P.train_step.TrainStep(G, D, cfg, dev)
        if mode == 'graph':
            step = ts.capture(bt, ix, iy, warmup=2)       # 2 eager warm-up iterations (capture itself executes nothing)
            for _ in range(2):
                log = step()
        else:
            for _ in range(4):
                log = ts.run(bt, ix, iy)
        torch.cuda.synchronize()
        res[mode] = ({k: float(v) for k, v in log.items()}, G.arena.P.detach().clone(), D.arena.P.detach().clone())
    for k, v in res['eager'][0].items():
        # not bitwise: bias / slab folds use float atomics whose order differs from launch to launch
        assert abs(res['graph'][0][k] - v) <= TOL * (abs(v) + 1e-6), (k, res['graph'][0][k], v)
    # parameters after 4 AdamW steps: early Adam updates are ~lr*sign(g), so last-bit gradient differences (atomics)
    # on near-zero gradient elements move single parameters by 2*lr
    assert rel_l2(res['graph'][1], res['eager'][1]) < 5e-4 and rel_l2(res['graph'][2], res['eager'][2]) < 5e-4


def test_f0_to_excitation_device_vs_reference_golden(dev):
    """tdvc_f0_to_excitation against the reference's own output with the same random draws."""
    g = np.load(os.path.join(GOLDEN, 'f0_excitation.npz'))
    U = pkg().util
    t = lambda a: torch.from_numpy(a).to(dev)
    exc = U.f0_to_excitation(t(g['f0']), int(g['step']), 16000, True, noise=(t(g['noise_v']), t(g['noise_u'])), start_phase=t(g['start_phase']))
    torch.cuda.synchronize()
    ref = torch.from_numpy(g['exc'])
    assert exc.shape == ref.shape
    assert float((exc.cpu() - ref).abs().max()) < 1e-4          # 1e-3 of the 0.1 sine amplitude (north star tolerance)
    # default path: draws on the device, same statistics
    exc2 = U.f0_to_excitation(t(g['f0']), int(g['step']))
    assert exc2.shape == ref.shape and abs(float(exc2.std()) - float(ref.std())) < 0.1 * float(ref.std())


def test_train_step_with_latent_classifier_vs_golden(dev):
    """the iteration with lambda_latcls = 1 (latent-classifier step with Adam + gradient-reversed
    classification term in the G loss) vs the fixture produced by the reference's own Generator / Discriminator /
    LatentClassifier modules (tests/golden/step_latcls.json, oracle/make_golden_latcls.py)."""
    P = pkg()
    gold = json.load(open(os.path.join(GOLDEN, 'step_latcls.json')))
    hp = P.hparams.HParam(os.path.join(os.path.dirname(GOLDEN), '..', 'config', 'conv_enc-stage1.yaml'))
    train = dict(hp.train); train['lambda_latcls'] = 1.0
    cfg = P.train_step.StepConfig.from_hparams(train)
    G, D = build_models(dev)
    C = P.modules.LatentClassifier(16, 128)
    C.load_state_dict(filled_sd('C')); C.ensure_arena(dev)
    ts = P.train_step.TrainStep(G, D, cfg, dev, C=C)
    T, B = gold['T'], gold['B']
    bt = to_dev(P.synth.make_batch(B, T, seed=1234, conversion=True), dev)
    for it, ref in enumerate(gold['losses']):
        ix = P.synth.contrastive_indices(B, T // 320, cfg.n_neg, seed=100 + 2 * it)
        iy = P.synth.contrastive_indices(B, T // 320, cfg.n_neg, seed=101 + 2 * it)
        log = ts.run(bt, ix, iy)
        torch.cuda.synchronize()
        errs = {k: abs(float(log[k]) - v) / (abs(v) + 1e-12) for k, v in ref.items()}
        tol = {k: TOL for k in errs}
        assert all(errs[k] < tol[k] for k in errs), (it, errs)
    # the parameter UPDATE of G, D and the classifier (Adam, no weight decay) against the reference's: sampled elements +
    # per-tensor update norms (a checksum of p cannot see whether an lr = 1e-4 optimizer ran)
    upd = np.load(os.path.join(GOLDEN, 'step_latcls_update.npz'))
    assert_update_matches_fixture(dict(G=G, D=D, C=C), upd, dict(G=filled_sd('G'), D=filled_sd('D'), C=filled_sd('C')), cfg.lr_g,
                                  'conv_enc-stage1 + lambda_latcls=1')


def test_latent_classifier_grads_vs_oracle(dev):
    """LatentClassifier (gradient-reversal + 3 strided convs + 2 convs + mean) + cross-entropy: loss, input gradient and every
    parameter gradient against the CPU oracle (model/latent_classifier.py:8-39, model/grad_rev.py:3-17)."""
    from oracle import model as OM
    P = pkg()
    C = P.modules.LatentClassifier(16, 128)
    C.load_state_dict(filled_sd('C')); C.ensure_arena(dev)
    rs = np.random.RandomState(9)
    emb = torch.from_numpy(rs.randn(4, 128, 50).astype(np.float32))
    lab = torch.from_numpy(rs.randint(0, 16, size=4).astype(np.int64))
    sc = {k: v.clone().requires_grad_(True) for k, v in filled_sd('C').items()}
    eo = emb.clone().requires_grad_(True)
    lo = torch.nn.functional.cross_entropy(OM.latent_classifier(sc, eo), lab)
    lo.backward()
    ed = emb.to(dev).requires_grad_(True)
    C.arena.zero_grad()
    l = P.losses.cross_entropy_loss(C(ed), lab.to(dev))
    l.backward()
    torch.cuda.synchronize()
    assert abs(float(l) - float(lo)) < 1e-5 * abs(float(lo))
    errs = {k: rel_l2(p.grad, sc[k].grad) for k, p in C.named_parameters()}
    errs['d_emb'] = rel_l2(ed.grad, eo.grad)
    assert max(errs.values()) < 1e-4, errs


def test_inference_path_long_utterance_vs_oracle(dev, tmp_path):
    """checkpoint interchange (a state_dict file with the reference's keys, loaded with weights_only=True),
    generator built from the reference config section, excitation on the device, ONE forward-only pass of a
    test.max_segment-long utterance (71680 samples, batch 1) against the CPU oracle."""
    from oracle import model as OM
    P = pkg()
    ckpt = tmp_path / 'latest-G.pt'
    torch.save(filled_sd('G'), ckpt)
    hp = P.hparams.HParam(os.path.join(os.path.dirname(GOLDEN), '..', 'config', 'conv_enc-stage1.yaml'))
    G = P.infer.build_generator(hp.model.generator, 16, dev)
    P.infer.load_generator_checkpoint(G, str(ckpt))
    T = 71680
    rs = np.random.RandomState(31)
    f0_src, f0_tgt = P.synth.make_f0(rs, 1, T), P.synth.make_f0(rs, 1, T)
    x = torch.from_numpy((rs.randn(1, 1, T) * 0.03).astype(np.float32))
    c_tgt = torch.zeros(1, 16); c_tgt[0, 5] = 1.0
    f0c = P.infer.shift_f0(torch.from_numpy(f0_src).to(dev), torch.from_numpy(f0_tgt).to(dev))
    noise = (torch.from_numpy(rs.randn(1, 1, T).astype(np.float32)).to(dev), torch.from_numpy(rs.randn(1, 1, T).astype(np.float32)).to(dev))
    phi0 = torch.tensor([1.234], device=dev)
    y = P.infer.convert(G, x.to(dev), c_tgt.to(dev), f0c, noise=noise, start_phase=phi0)
    exc = P.util.f0_to_excitation(f0c, 64, noise=noise, start_phase=phi0)
    torch.cuda.synchronize()
    assert y.shape == (1, 1, T) and not y.requires_grad
    with torch.no_grad():
        yo, _, _ = OM.generator(filled_sd('G'), x, c_tgt, exc.cpu())
    assert rel_l2(y, yo) < TOL


@pytest.mark.parametrize('B,T', [(1, 8320), (3, 9600)], ids=['B1_Tmin8320', 'B3_T9600'])
def test_generator_edge_sizes_vs_oracle(dev, B, T):
    """Smallest legal utterance (T = 8320: 26 frames) with a single sample, and an odd batch at a length whose
    per-stage sequence lengths are not multiples of the tile sizes (30 / 300 / 2400 / 4800 / 9600): forward outputs and the
    content embedding against the CPU oracle, plus one backward pass for finiteness of every gradient."""
    from oracle import model as OM
    P = pkg()
    G, _ = build_models(dev)
    bt_cpu = P.synth.make_batch(B, T, seed=40 + B)
    bt = to_dev(bt_cpu, dev)
    with torch.no_grad():
        yo, osubs, oemb = OM.generator(filled_sd('G'), bt_cpu['signal_real'], bt_cpu['c_tgt'], bt_cpu['c_f0_conv'])
    G.arena.zero_grad()
    y, subs = G(bt['signal_real'], bt['c_tgt'], c_var=bt['c_f0_conv'], out_subsample=True)
    errs = dict(y=rel_l2(y, yo), sub4=rel_l2(subs[0], osubs[0]), sub2=rel_l2(subs[1], osubs[1]), emb=rel_l2(G.content_embedding, oemb))
    assert max(errs.values()) < TOL, errs
    (y.square().mean() + subs[0].square().mean() + subs[1].square().mean()).backward()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p.grad).all() for p in G.parameters() if p.grad is not None)
