"""CPU: the oracle (oracle/) against the committed golden fixtures that were generated from the reference
itself (oracle/make_golden.py). This is what pins the oracle on a machine without /root/reference."""
import json
import os

import numpy as np
import pytest
import torch

from common import GOLDEN, filled_sd, pkg, rel_l2

TOL = 1e-4


def _sd(which, grad=False):
    return {k: v.clone().requires_grad_(grad) for k, v in filled_sd(which).items()}


def test_filters_match_reference():
    from oracle import model as OM
    f = np.load(os.path.join(GOLDEN, 'filters.npz'))
    assert rel_l2(OM.kaiser_sinc_odd(129, 0.5, 10.0), torch.from_numpy(f['d_down'])) < 1e-6
    for r in (2, 8, 10):
        assert rel_l2(OM.kaiser_sinc_even(16 * r, 1.0 / r), torch.from_numpy(f[f'exc_r{r}'])) < 1e-6
    # the product-side filter builders are the same arithmetic
    M = pkg().modules
    assert rel_l2(M.kaiser_filter_odd(129, 0.5, 10), torch.from_numpy(f['d_down'])) < 1e-6
    assert rel_l2(M.kaiser_filter_even(32, 0.5), torch.from_numpy(f['exc_r2'])) < 1e-6


def test_generator_forward_and_grads_vs_golden():
    from oracle import model as OM
    tag, seed, T = 'B2_T8960', 7, 8960
    bt = pkg().synth.make_batch(2, T, seed=seed)
    gold = np.load(os.path.join(GOLDEN, f'gen_fwd_{tag}.npz'))
    sd = _sd('G', grad=True)
    y, subs, emb = OM.generator(sd, bt['signal_real'], bt['c_tgt'], bt['c_f0_conv'])
    for name, t in (('y', y), ('sub4', subs[0]), ('sub2', subs[1]), ('emb', emb)):
        assert rel_l2(t, torch.from_numpy(gold[name])) < TOL, name
    rs = np.random.RandomState(99)
    cot = [torch.from_numpy(rs.randn(*t.shape).astype(np.float32)) for t in (y, subs[0], subs[1], emb)]
    sum((t * c).mean() for t, c in zip((y, subs[0], subs[1], emb), cot)).backward()
    gg = json.load(open(os.path.join(GOLDEN, f'gen_grad_{tag}.json')))
    for k, n_ref in gg['norms'].items():
        g = sd[k].grad
        if n_ref < 0:     # the reference leaves these parameters without a gradient
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        assert abs(float(g.double().norm()) - n_ref) <= max(1e-3, 5 * gg['noise'][k]) * n_ref, k


def test_discriminator_vs_golden():
    from oracle import losses as OL, model as OM
    tag, seed, T = 'B2_T8960', 7, 8960
    bt = pkg().synth.make_batch(2, T, seed=seed)
    gold = np.load(os.path.join(GOLDEN, f'disc_{tag}.npz'))
    gj = json.load(open(os.path.join(GOLDEN, f'disc_{tag}.json')))
    gen = np.load(os.path.join(GOLDEN, f'gen_fwd_{tag}.npz'))
    sd = _sd('D', grad=True)
    fake = torch.from_numpy(gen['y'])
    fsubs = [torch.from_numpy(gen['sub4']), torch.from_numpy(gen['sub2'])]
    rsubs = OM.disc_subsamples(bt['signal_real'])
    o_r, _ = OM.discriminator(sd, bt['signal_real'], bt['label_src'], rsubs)
    o_f, _ = OM.discriminator(sd, fake, bt['label_tgt'], fsubs)
    for i in range(5):
        assert rel_l2(o_r[i], torch.from_numpy(gold[f'out_real_{i}'])) < TOL
        assert rel_l2(o_f[i], torch.from_numpy(gold[f'out_fake_{i}'])) < TOL
    loss = OL.lsgan_to_one(o_r) + OL.lsgan_to_zero(o_f)
    assert abs(float(loss) - gj['loss']) < 1e-5 * abs(gj['loss'])
    loss.backward()
    for k, n_ref in gj['norms'].items():
        assert abs(float(sd[k].grad.double().norm()) - n_ref) <= max(1e-3, 5 * gj['noise'][k]) * n_ref, k


def test_cin_and_latent_classifier_vs_golden():
    from oracle import model as OM
    g = np.load(os.path.join(GOLDEN, 'cin.npz'))
    sd = pkg().synth.fill_state_dict(json.load(open(os.path.join(GOLDEN, 'shapes_CIN.json'))))
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.randn(3, 32, 500).astype(np.float32)).requires_grad_(True)
    c2 = torch.from_numpy(rs.randn(3, 128).astype(np.float32))
    c3 = torch.from_numpy(rs.randn(3, 129, 500).astype(np.float32))
    cot = torch.from_numpy(rs.randn(3, 32, 500).astype(np.float32))
    for name, c in (('2d', c2), ('3d', c3)):
        x.grad = None
        y = OM.cond_instance_norm(sd, '', x, c)
        (y * cot).mean().backward()
        assert rel_l2(y, torch.from_numpy(g[f'y_{name}'])) < 1e-5 and rel_l2(x.grad, torch.from_numpy(g[f'dx_{name}'])) < 1e-5
    lc = np.load(os.path.join(GOLDEN, 'latcls.npz'))
    sdc = pkg().synth.fill_state_dict(json.load(open(os.path.join(GOLDEN, 'shapes_C.json'))))
    e = torch.from_numpy(np.random.RandomState(3).randn(2, 128, 50).astype(np.float32)).requires_grad_(True)
    out = OM.latent_classifier(sdc, e)
    out.square().mean().backward()
    assert rel_l2(out, torch.from_numpy(lc['y'])) < 1e-5 and rel_l2(e.grad, torch.from_numpy(lc['dx'])) < 1e-4


def test_losses_vs_golden():
    from oracle import losses as OL, model as OM
    tag = 'B2_T8960'
    g = np.load(os.path.join(GOLDEN, f'losses_{tag}.npz'))
    gen = np.load(os.path.join(GOLDEN, f'gen_fwd_{tag}.npz'))
    emb, emb_cor = torch.from_numpy(gen['emb']), torch.from_numpy(g['emb_cor'])
    ix, iy = torch.from_numpy(g['idx_x'].astype(np.int64)), torch.from_numpy(g['idx_y'].astype(np.int64))
    assert abs(float(OL.contrastive(emb, emb_cor, ix, iy)) - float(g['contrastive'])) < 1e-5 * float(g['contrastive'])


def test_full_iteration_vs_golden():
    """One D+G iteration of conv_enc-stage1 at B=2, T=8960 against the fixture produced by the reference's own
    modules + torch.optim.AdamW (train.py:259-491 semantics)."""
    from oracle import step as OS
    gold = json.load(open(os.path.join(GOLDEN, 'step_conv_enc-stage1_T8960.json')))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    st = OS.TrainStep(filled_sd('G'), filled_sd('D'), OS.StepConfig())
    bt = pkg().synth.make_batch(2, 8960, seed=1234)
    ix, iy = pkg().synth.contrastive_indices(2, 28, 100, 100), pkg().synth.contrastive_indices(2, 28, 100, 101)
    log = st.run(bt, ix, iy)
    for k, v in gold['losses'][0].items():
        assert abs(log[k] - v) <= 1e-5 * (abs(v) + 1e-12), k


def test_excitation_restatement_vs_reference_golden():
    """synth.excitation_from_f0 (numpy restatement of util/__init__.py:22-50) against the output of the reference's own
    f0_to_excitation with its three random draws replayed (tests/golden/f0_excitation.npz, oracle/make_golden_f0.py)."""
    import importlib
    import numpy as np
    synth = importlib.import_module('td-vc-gan_amd.synth')
    g = np.load(os.path.join(GOLDEN, 'f0_excitation.npz'))

    class Replay:
        k = 0
        def uniform(self): return float(g['start_phase'][0]) / (2 * np.pi)
        def randn(self, *shape):
            self.k += 1
            return (g['noise_v'] if self.k == 1 else g['noise_u']).astype(np.float64)
    ours = synth.excitation_from_f0(g['f0'], Replay(), int(g['step']))
    assert ours.shape == g['exc'].shape
    assert float(np.abs(ours - g['exc']).max()) < 2e-5      # amplitude 0.1; the reference accumulates the phase in fp32


def test_oracle_step_with_latent_classifier_vs_golden():
    """oracle.step.TrainStep with lambda_latcls = 1 against the reference-module iteration (tests/golden/step_latcls.json)."""
    import importlib
    from oracle import step as OS
    synth = importlib.import_module('td-vc-gan_amd.synth')
    hparams = importlib.import_module('td-vc-gan_amd.hparams')
    gold = json.load(open(os.path.join(GOLDEN, 'step_latcls.json')))
    hp = hparams.HParam(os.path.join(os.path.dirname(GOLDEN), '..', 'config', 'conv_enc-stage1.yaml'))
    train = dict(hp.train); train['lambda_latcls'] = 1.0
    cfg = OS.StepConfig.from_hparams(train)
    ost = OS.TrainStep(filled_sd('G'), filled_sd('D'), cfg, filled_sd('C'))
    bt = synth.make_batch(gold['B'], gold['T'], seed=1234, conversion=True)
    ref = gold['losses'][0]
    log = ost.run(bt, synth.contrastive_indices(gold['B'], gold['T'] // 320, cfg.n_neg, seed=100),
                  synth.contrastive_indices(gold['B'], gold['T'] // 320, cfg.n_neg, seed=101))
    errs = {k: abs(log[k] - v) / (abs(v) + 1e-12) for k, v in ref.items()}
    assert max(errs.values()) < 1e-5, errs
