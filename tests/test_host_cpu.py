"""CPU: host-side logic of the product package — C-ABI symbols, state_dict layout, configs, synthetic inputs,
and that the product path refuses to run without a GPU (no CPU fallback)."""
import json
import os
import re

import pytest
import torch

from common import D_ARGS, G_ARGS, GOLDEN, ROOT, pkg, shapes


def test_library_exports_every_declared_symbol():
    L = pkg()._lib
    hdr = open(os.path.join(ROOT, 'include', 'tdvc.h')).read()
    declared = set(re.findall(r'\b(tdvc_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'tdvc_status'}
    lib = L.lib()                      # dlopen; raises if the .so is missing
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in include/tdvc.h but not exported by libtdvc_hip.so'
    assert declared <= set(L.SIGNATURES), declared - set(L.SIGNATURES)
    assert lib.tdvc_version() >= 100
    # argument validation happens on the host, before any launch
    d = L.ConvDesc(L.CONV, 2, 16, 16, 100, 99, 3, 1, 1, 1, 1, 0, 0, 0)
    assert lib.tdvc_conv_wgrad_workspace(d) == 0            # Tout inconsistent with the conv arithmetic
    import ctypes as C
    a = L.ConvFwdArgs()
    assert lib.tdvc_conv_fwd(C.byref(d), C.byref(a), None) == -1
    assert b'Tout' in lib.tdvc_last_error()


def test_state_dict_matches_reference_layout():
    M = pkg().modules
    G = M.Generator(**{**G_ARGS, 'decoder_channels': list(G_ARGS['decoder_channels'])})
    D = M.CollaborativeMultibandDiscriminator(**D_ARGS)
    assert {k: list(v.shape) for k, v in G.state_dict().items()} == shapes('G')
    assert {k: list(v.shape) for k, v in D.state_dict().items()} == shapes('D')
    assert sum(p.numel() for p in G.parameters()) == 14619806 and sum(p.numel() for p in D.parameters()) == 17062368
    cin = M.ConditionalInstanceNorm(32, 128)
    assert {k: list(v.shape) for k, v in cin.state_dict().items()} == shapes('CIN')
    C = M.LatentClassifier(16, 128)
    assert {k: list(v.shape) for k, v in C.state_dict().items()} == shapes('C')


def test_unsupported_configurations_raise():
    M = pkg().modules
    with pytest.raises(NotImplementedError):
        M.Generator(**{**G_ARGS, 'encoder_model': 'hubert'})       # only 'conv' and 'wavlm' exist (model/generator.py:453)
    G = M.Generator(**{**G_ARGS, 'encoder_model': 'wavlm'})         # SSL content encoder: state_dict layout of model/ssl_encoder.py
    ref_keys = set(json.load(open(os.path.join(GOLDEN, 'shapes_SSLENC.json'))))
    assert {k[len('encoder.encoder.'):] for k in G.state_dict() if k.startswith('encoder.')} == ref_keys
    with pytest.raises(NotImplementedError):
        M.Generator(**{**G_ARGS, 'norm_layer': ('instance_norm',) * 3})


def test_no_cpu_fallback():
    M = pkg().modules
    D = M.CollaborativeMultibandDiscriminator(**D_ARGS)
    with pytest.raises(RuntimeError):          # TdvcError: modules run on an MI355X device only
        D(torch.zeros(1, 1, 8960), torch.zeros(1, dtype=torch.int64))


def test_configs_and_step_config():
    P = pkg()
    for name, no_conv, idt in (('conv_enc-stage1', False, 5.0), ('conv_enc-stage2_1', True, 20.0)):
        hp = P.hparams.HParam(os.path.join(ROOT, 'config', f'{name}.yaml'))
        cfg = P.train_step.StepConfig.from_hparams(hp.train)
        assert cfg.no_conv is no_conv and cfg.lambda_idt == idt and cfg.lambda_feat == 2 and cfg.lambda_spec == 5
        assert cfg.betas == (0.8, 0.99) and cfg.weight_decay == 1e-2 and cfg.eps == 1e-8
        assert hp.model.generator.decoder_ratios == [10, 8, 2, 2] and hp.model.discriminator.num_disc == 3


def test_synthetic_batch_is_deterministic_and_shaped():
    S = pkg().synth
    a, b = S.make_batch(3, 8960, seed=5), S.make_batch(3, 8960, seed=5)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert a['signal_real'].shape == (3, 1, 8960) and a['c_f0_conv'].shape == (3, 1, 8960)
    assert a['c_tgt'].shape == (3, 16) and torch.equal(a['c_tgt'].argmax(1), a['label_tgt'])
    assert torch.equal(a['label_tgt'], a['label_src'][a['perm']])
    rms = a['signal_real'].pow(2).mean(-1).sqrt()
    assert float(rms.max()) <= 10 ** (-30 / 20) * 1.01 and float(rms.min()) >= 10 ** (-30 / 20) * 0.29
    with pytest.raises(AssertionError):
        S.make_batch(1, 3200)                                  # reflect padding needs T >= 8320
    sd = S.fill_state_dict({'a.weight_v': [4, 3, 5], 'a.weight_g': [4, 1, 1], 'a.bias': [4]})
    assert sd['a.weight_g'].shape == (4, 1, 1) and float(sd['a.weight_g'].min()) > 0


def test_mel_filterbank_matches_oracle():
    """Product-side mel filterbank / DFT basis (numpy float64) against the oracle's torch restatement."""
    from oracle import losses as OL
    LS = pkg().losses
    fb = torch.from_numpy(LS._mel_filterbank(1025, 80, 16000)).float()
    assert float((fb - OL.mel_filterbank(1025, 80, 16000)).abs().max()) < 1e-6
    m = LS.MelSpec(16000, 512, 80)
    # the DFT basis reproduces torch.stft on a centred frame
    x = torch.randn(1, 1, 2048)
    xp = torch.nn.functional.pad(x, (256, 256), mode='reflect')
    spec = torch.nn.functional.conv1d(xp, m.basis, stride=128)          # [1, 2*Fp, N]
    ref = torch.stft(x.reshape(1, -1), 512, hop_length=128, window=torch.hann_window(512), center=True,
                     pad_mode='reflect', return_complex=True)
    Fp = m.basis.shape[0] // 2
    assert float((spec[0, :257] - ref[0].real).abs().max()) < 2e-3 and float((spec[0, Fp:Fp + 257] - ref[0].imag).abs().max()) < 2e-3


def test_x6_lds_swizzle_is_conflict_free_under_b128_lane_groups():
    """The LDS image of the split-bf16 kernels (conv_fwd_x6.hip `swz`, conv_wgrad_x6.hip `x6_swz`): 64-byte rows, 16-byte slot q of row r
    stored at slot q ^ ((r >> 1) & 3). Under gfx950's ds_read_b128 lane grouping (four NON-contiguous 16-lane groups, MI355X_MICROARCH.md)
    the MFMA fragment read `row = base + (lane & 15), slot = lane >> 4` must be conflict-free for every base row; the padded 80-byte
    rows the kernels started with are not (measured: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50, profiles/r03_pmc.txt history)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('lds_swizzle_check', os.path.join(ROOT, 'tools', 'lds_swizzle_check.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    swz = lambda r, q: r * 64 + 16 * (q ^ ((r >> 1) & 3))
    for base in range(64):
        assert m.extra_cycles(lambda l, b=base: swz(b + (l & 15), l >> 4)) == 0, base
    assert m.extra_cycles(lambda l: (l & 15) * 80 + 16 * (l >> 4)) == 4          # one extra cycle in each of the four groups
    # the swizzle is an involution on the 4 slots of a row: writes and reads use the same formula
    for r in range(16):
        assert sorted((q ^ ((r >> 1) & 3)) for q in range(4)) == [0, 1, 2, 3]
