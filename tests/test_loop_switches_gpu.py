"""GPU: the train-loop switches every shipped YAML leaves off -- gradient clipping by the global norm (train.py:289-290, 489-490),
a frozen encoder (:195-197), D / G step intervals (:259, :320), the waveform L1 terms (:357-360, 381-384), the jittered loss
target (:335-336, util.audio.add_jitter) and lambda_converted (:409-413, which reaches no loss in the reference) -- through the
product TrainStep.run() over several iterations against the CPU oracle's run() with the same switches (oracle/step.py uses
torch.nn.utils.clip_grad_norm_ / requires_grad / torch.roll directly, i.e. the reference's own calls)."""
import os

import numpy as np
import pytest
import torch

from common import build_models, filled_sd, pkg, to_dev
from test_step_launch_shape_gpu import update_stats

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _run_both(train_overrides, iters, dev, B=2, T=8960, base='conv_enc-stage2_2'):
    from oracle import step as OS
    P = pkg()
    hp = P.hparams.HParam(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'config', f'{base}.yaml'))
    train = dict(hp.train); train.update(train_overrides); train['lambda_f0'] = 0
    cfg, ocfg = P.train_step.StepConfig.from_hparams(train), OS.StepConfig.from_hparams(train)
    G, D = build_models(dev)
    ts = P.train_step.TrainStep(G, D, cfg, dev)
    sd_g, sd_d = filled_sd('G'), filled_sd('D')
    ost = OS.TrainStep(sd_g, sd_d, ocfg)
    bt_cpu = P.synth.make_batch(B, T, seed=99, conversion=True)
    if cfg.jitter_amp > 0:
        bt_cpu['jitter'] = torch.from_numpy(np.random.RandomState(5).randint(-cfg.jitter_amp, cfg.jitter_amp + 1, size=B).astype(np.int64))
    bt = to_dev(bt_cpu, dev)
    logs = []
    for it in range(iters):
        ix = P.synth.contrastive_indices(B, T // 320, cfg.n_neg, seed=300 + 2 * it)
        iy = P.synth.contrastive_indices(B, T // 320, cfg.n_neg, seed=301 + 2 * it)
        log = ts.run(bt, ix.to(dev), iy.to(dev))
        torch.cuda.synchronize()
        ref = ost.run(bt_cpu, ix, iy)
        assert set(log) == set(ref), (it, sorted(set(log) ^ set(ref)))
        bad = {k: (float(log[k]), v) for k, v in ref.items() if abs(float(log[k]) - v) > TOL * (abs(v) + 1e-12)}
        assert not bad, (it, bad)
        logs.append(ref)
    return G, D, ts, ost, sd_g, sd_d, cfg, logs


def _check_updates(model, ref_params, before, lr, what):
    fl, rest = [], []
    for k, p in model.named_parameters():
        r = ref_params[k].detach()
        if torch.equal(r, before[k]):      # the oracle never moved it (frozen / dead / skipped step): bit-identical here too
            assert torch.equal(p.detach().cpu(), before[k]), f'{what}: {k} moved but the reference leaves it untouched'
            continue
        f, rs, _ = update_stats(before[k], p.detach(), r, lr)
        fl.append(f); rest.append(rs)
    assert fl, what
    assert float(np.median(fl)) <= 5e-3 and float(np.max(fl)) <= 0.08, (what, float(np.median(fl)), float(np.max(fl)))
    assert float(np.median(rest)) <= 1e-2, (what, float(np.median(rest)))


def test_gradient_clipping_and_waveform_terms(dev):
    """clip_grad_norm_ with thresholds far below the actual norms (every step is clipped), lambda_wave on both branches."""
    G, D, ts, ost, sd_g, sd_d, cfg, logs = _run_both(dict(grad_max_norm_D=0.05, grad_max_norm_G=0.5, lambda_wave=2.0), 2, dev)
    assert 'G_loss_rec_wave' in logs[0] and 'G_loss_idt_wave' in logs[0]
    # both optimizers really were clipped: the device-side total norms exceed the thresholds, and the oracle's generator gradients
    # (scaled IN PLACE by torch's clip_grad_norm_; the D ones are zeroed after its step) now have exactly the threshold norm
    assert float(ts.opt_d.last_grad_norm) > 0.05 and float(ts.opt_g.last_grad_norm) > 0.5, (float(ts.opt_d.last_grad_norm), float(ts.opt_g.last_grad_norm))
    ref_norm = float(torch.sqrt(torch.stack([(p.grad.double() ** 2).sum() for p in ost.g.values() if p.grad is not None]).sum()))
    assert abs(ref_norm - 0.5) <= 1e-3 * 0.5, ref_norm
    _check_updates(D, ost.d, sd_d, cfg.lr_d, 'D (clipped)')
    _check_updates(G, ost.g, sd_g, cfg.lr_g, 'G (clipped)')


def test_frozen_encoder_intervals_and_jitter(dev):
    """freeze_subnets: [encoder], G-step every 2nd iteration, jittered loss target, lambda_converted (a no-op, like in the reference)."""
    G, D, ts, ost, sd_g, sd_d, cfg, logs = _run_both(dict(freeze_subnets=['encoder'], G_step_interval=2, jitter_amp=37, lambda_converted=1.0), 3, dev)
    assert 'G_loss' in logs[0] and 'G_loss' not in logs[1] and 'G_loss' in logs[2] and all('D_loss' in l for l in logs)
    for k, p in G.named_parameters():
        if k.startswith('encoder.'):
            assert p.grad is None and torch.equal(p.detach().cpu(), sd_g[k]), f'{k}: frozen encoder parameter changed'
    assert G.arena.n_live < sum(p.numel() for k, p in G.named_parameters() if not k.startswith('decoder.excite_downsample.0.'))
    _check_updates(D, ost.d, sd_d, cfg.lr_d, 'D')
    _check_updates(G, ost.g, sd_g, cfg.lr_g, 'G (decoder + embedding only)')
    with pytest.raises(RuntimeError, match='interval'):
        ts.capture({}, None, None)


def test_cin_single_pass_matches_three_pass(dev):
    """tdvc_cin_fwd / bwd: the register-resident single-pass kernels (T % 4 == 0, T <= 16384 / 8192) against the three-pass
    kernels they replace (an unaligned view of the same data takes the old route): same statistics, same outputs."""
    P = pkg()
    torch.manual_seed(3)
    for B, C, T, Tg in ((2, 8, 4096, 1), (2, 5, 12000, 12000), (3, 4, 2048, 2048)):
        x_al = torch.randn(B, C, T, device=dev)
        gb_al = torch.randn(B, 2 * C, Tg, device=dev)
        res = {}
        for name in ('single', 'three'):
            xr = x_al.clone().requires_grad_(True)
            gr = gb_al.clone().requires_grad_(True)
            y = P.ops.CinFn.apply(xr, gr, 1e-5) if name == 'single' else _cin_unaligned(P, xr, gr)
            cot = torch.sin(torch.arange(y.numel(), device=dev, dtype=torch.float32)).view_as(y)
            (y * cot).sum().backward()
            res[name] = (y.detach(), xr.grad.detach().clone(), gr.grad.detach().clone())
        for a, b_, what in zip(res['single'], res['three'], ('y', 'dx', 'dgb')):
            err = float((a.double() - b_.double()).norm() / (b_.double().norm() + 1e-30))
            assert err < 2e-6, (B, C, T, Tg, what, err)


def _cin_unaligned(P, x, gb):
    """CinFn on operands whose rows are NOT 16-byte aligned (CinFn itself makes them contiguous = aligned): call the C ABI directly."""
    import ctypes as C_
    L = P._lib

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, gb):
            B, Cc, T = x.shape
            Tg = gb.shape[2]
            hold = torch.empty(x.numel() + 1, device=x.device); xs = hold[1:].view_as(x); xs.copy_(x)
            yh = torch.empty(x.numel() + 1, device=x.device); y = yh[1:].view_as(x)
            mean, rstd = torch.empty(B, Cc, device=x.device), torch.empty(B, Cc, device=x.device)
            st = torch.cuda.current_stream(x.device).cuda_stream
            L.check(L.lib().tdvc_cin_fwd(xs.data_ptr(), gb.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, Cc, T, Tg, 1e-5, st))
            ctx.save_for_backward(xs, gb, mean, rstd)
            return y
        @staticmethod
        def backward(ctx, dy):
            xs, gb, mean, rstd = ctx.saved_tensors
            B, Cc, T = xs.shape
            dh = torch.empty(dy.numel() + 1, device=dy.device); d = dh[1:].view_as(dy); d.copy_(dy)
            dxh = torch.empty(dy.numel() + 1, device=dy.device); dx = dxh[1:].view_as(dy)
            dgb = torch.empty_like(gb)
            st = torch.cuda.current_stream(dy.device).cuda_stream
            L.check(L.lib().tdvc_cin_bwd(xs.data_ptr(), gb.data_ptr(), d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dgb.data_ptr(),
                                         B, Cc, T, gb.shape[2], st))
            return dx.clone(), dgb
    return Fn.apply(x, gb.contiguous())
