"""Parity of the kernel INSTANCES the train step runs at its real launch shapes.

`launch_conv_lean` picks its tile from the grid size (blocks >= 512), so small test shapes only ever reach the
16x64 tile; the benchmark's launch shape (2 x 16 samples per conv) selects the 16/32/48/64-row x 256-column
instances instead. Two complementary sweeps, both against float64 CPU autograd at 2e-5 rel-L2 per tensor:

  (a) forced tiles: every lean tile configuration (tdvc_debug_force_tile) x every (prologue, epilogue) pair the step
      uses, at small shapes with ragged ends (T % NT != 0), reflect-pad mirror folds up to pad 25 and 136-row tiles;
  (b) launch shapes: the op cases of the B = 16 step (every trunk conv sees 32 samples, D layer 5 sees 64) with the
      automatic tile choice — these also put the weight-grad kernels (conv_wgrad_pipe / _tile / _lean) in their
      multi-chunk steady state and cover the generic strided / grouped / transposed kernel at its grid-aware tiles.

Each case records which kernel instantiations it launched (tdvc_debug_trace) and asserts the expected one is among them,
so a silent fall-back to another kernel cannot pass. test_zz_profile_coverage_gpu.py then checks that every tdvc kernel
instance named in profiles/*kernel_stats.csv was launched by a passing test of this session.
"""
import importlib

import pytest
import torch

import test_conv_ops_gpu as OPS
from common import rel_l2, traced

pytestmark = pytest.mark.gpu
TOL = 2e-5

LEAN_CFG = {0: (1, 4, 1, 4), 1: (2, 4, 1, 4), 2: (4, 4, 1, 4), 3: (1, 1, 1, 4), 4: (1, 4, 4, 1), 5: (3, 4, 1, 4), 6: (1, 2, 2, 2), 7: (9, 1, 1, 4)}
LXF_ACT, LXF_FILM, LXF_MASK_LRELU = 0, 1, 2
EPI_FWD, EPI_MASK, EPI_FILM, EPI_PLAIN = 0, 1, 2, 3


def _L():
    return importlib.import_module('td-vc-gan_amd')._lib


def lean_name(cfg, xfk, epi):
    m, n, wm, wn = LEAN_CFG[cfg]
    return f'conv_lean_kernel<{m},{n},{wm},{wn},{xfk},{epi}>'


def _run_forced(cfg, fn):
    L = _L()
    L.lib().tdvc_debug_force_tile(cfg)
    try:
        with traced() as tr:
            errs = fn()
    finally:
        L.lib().tdvc_debug_force_tile(-1)
    return errs, tr.names


# ------------------------------------------------------------------------------------------------ (a) forced tiles
# FiLM block = dilated reflect conv (ACT/FWD) -> FiLM 1x1 posconv (FILM/FWD); backward: posconv input-grad (ACT/FILM),
# dilated input-grad with the reflect mirror fold and the residual add (ACT/MASK).  (C, k, d, T, cond, with_acc)
FILM_SHAPES = [(16, 3, 1, 520, True, True), (16, 11, 5, 332, True, False), (32, 7, 3, 520, True, True), (64, 11, 5, 332, True, True),
               (64, 3, 1, 268, False, False), (128, 7, 5, 140, True, True)]


@pytest.mark.parametrize('shape', FILM_SHAPES, ids=[f'C{c}k{k}d{d}T{t}' + ('' if cond else '_nocond') for c, k, d, t, cond, _ in FILM_SHAPES])
@pytest.mark.parametrize('cfg', sorted(LEAN_CFG), ids=[f'tile{c}_' + 'x'.join(map(str, LEAN_CFG[c])) for c in sorted(LEAN_CFG)])
def test_forced_tile_film_block(cfg, shape, dev):
    errs, names = _run_forced(cfg, lambda: OPS.film_block_errors(shape, dev, B=2))
    assert max(errs.values()) < TOL, errs
    want = [lean_name(cfg, LXF_ACT, EPI_FWD), lean_name(cfg, LXF_ACT, EPI_MASK)]
    want += [lean_name(cfg, LXF_FILM, EPI_FWD), lean_name(cfg, LXF_ACT, EPI_FILM)] if shape[4] else []
    missing = [w for w in want if w not in names]
    assert not missing, (missing, sorted(n for n in names if 'lean_kernel' in n))


# Plain conv cases (tests/test_conv_ops_gpu.py layout): (name, cin, cout, k, stride, pad, dil, groups, reflect, transposed, T, pre, post)
FORCED_CONV = [
    # D-style layer: no prologue, in-place LeakyReLU after -> input-grad = MASK_LRELU prologue, PLAIN epilogue
    ('d5_like_128_k5_lrelu', 128, 128, 5, 1, 2, 1, 1, False, False, 252, 0, 1, (LXF_MASK_LRELU, EPI_PLAIN)),
    ('d5_like_256_k5_T63', 256, 256, 5, 1, 2, 1, 1, False, False, 63, 0, 1, (LXF_MASK_LRELU, EPI_PLAIN)),
    # cond_var.0 dense (no activation either side) -> ACT/PLAIN input-grad; 136 rows = the 48-row tile's reason to exist
    ('cond0_136_136', 136, 136, 3, 1, 1, 1, 1, False, False, 268, 0, 0, (LXF_ACT, EPI_PLAIN)),
    # cond_var.2: LeakyReLU before, 136-row input-grad with the LeakyReLU mask epilogue
    ('cond2_136_64', 136, 64, 3, 1, 1, 1, 1, False, False, 332, 1, 0, (LXF_ACT, EPI_MASK)),
    ('cond2_136_256', 136, 256, 3, 1, 1, 1, 1, False, False, 140, 1, 0, (LXF_ACT, EPI_MASK)),
    # reflect conv with the widest halo at a short sequence: both mirror folds inside one tile
    ('dil_c32_k11_d5_T52', 32, 32, 11, 1, 25, 5, 1, True, False, 52, 1, 0, (LXF_ACT, EPI_MASK)),
]


@pytest.mark.parametrize('case', FORCED_CONV, ids=[c[0] for c in FORCED_CONV])
@pytest.mark.parametrize('cfg', sorted(LEAN_CFG), ids=[f'tile{c}_' + 'x'.join(map(str, LEAN_CFG[c])) for c in sorted(LEAN_CFG)])
def test_forced_tile_conv(cfg, case, dev):
    T = case[10]
    if T <= 80 and LEAN_CFG[cfg][1] * LEAN_CFG[cfg][3] * 16 == 256:
        pytest.skip('256-column tiles are never selected for T <= 80 (launch_conv_lean)')
    errs, names = _run_forced(cfg, lambda: OPS.conv_case_errors(case[:13], dev, 0, B=3))
    assert max(errs.values()) < TOL, errs
    want = [lean_name(cfg, LXF_ACT, EPI_FWD), lean_name(cfg, *case[13])]
    missing = [w for w in want if w not in names]
    assert not missing, (missing, sorted(n for n in names if 'lean_kernel' in n))


# ------------------------------------------------------------------------------------------------ (b) launch shapes
# FiLM blocks of the decoder MRFs at the step's launch shape: 2 x 16 samples, all (k, d) of the 16-channel stage,
# the corner (k, d) pairs of the wider stages.  (C, k, d, T, cond, with_acc), B
LAUNCH_FILM = [((16, k, d, 16000, True, d == 5), 32) for k in (3, 7, 11) for d in (1, 3, 5)] + \
              [((32, 3, 1, 8000, True, False), 32), ((32, 11, 5, 8000, True, True), 32), ((32, 7, 3, 8000, False, False), 32),
               ((64, 3, 1, 4000, True, False), 32), ((64, 11, 5, 4000, True, True), 32), ((64, 7, 3, 4000, False, False), 32),
               ((128, 3, 1, 500, True, False), 32), ((128, 11, 5, 500, True, True), 32), ((128, 7, 3, 500, False, False), 32),
               ((256, 3, 1, 50, False, False), 32), ((256, 11, 5, 50, False, True), 32)]


@pytest.mark.parametrize('shape,B', LAUNCH_FILM, ids=[f'C{s[0]}k{s[1]}d{s[2]}T{s[3]}B{b}' + ('' if s[4] else '_enc') for s, b in LAUNCH_FILM])
def test_launch_shape_film_block(shape, B, dev):
    with traced() as tr:
        errs = OPS.film_block_errors(shape, dev, B=B)
    assert max(errs.values()) < TOL, (errs, sorted(tr.names))
    assert any('conv_lean_kernel' in n for n in tr.names), sorted(tr.names)
    if shape[0] == 16 and shape[3] >= 512:      # the 16-channel long-sequence blocks run the one-launch forward (film_block.hip)
        want = 'film_block_fwd_kernel<true>' if shape[4] else 'film_block_fwd_kernel<false>'
        assert want in tr.names, (want, sorted(tr.names))


# (name, cin, cout, k, stride, pad, dil, groups, reflect, transposed, T, pre, post), B
LAUNCH_CONV = [
    # FiLM conditioning convs, dense formulation (cond_var.0) and cond_var.2 at every decoder stage
    (('cond0_136_136_T4000', 136, 136, 3, 1, 1, 1, 1, False, False, 4000, 0, 0), 16),
    (('cond2_136_32_T16000', 136, 32, 3, 1, 1, 1, 1, False, False, 16000, 1, 0), 32),
    (('cond2_136_64_T8000', 136, 64, 3, 1, 1, 1, 1, False, False, 8000, 1, 0), 32),
    (('cond2_136_128_T4000', 136, 128, 3, 1, 1, 1, 1, False, False, 4000, 1, 0), 32),
    (('cond2_136_256_T500', 136, 256, 3, 1, 1, 1, 1, False, False, 500, 1, 0), 32),
    # discriminator: layer 0, grouped strided layers 1-4, dense layer 5, output conv (real + fake = 2 x 16, with the
    # sub-scale pass of discs 1 / 2 batched on top = 64)
    (('d0_1_16_k15', 1, 16, 15, 1, 7, 1, 1, True, False, 16000, 0, 1), 32),
    (('d1_grp_16_64', 16, 64, 41, 4, 20, 1, 4, False, False, 16000, 0, 1), 32),
    (('d2_grp_64_256', 64, 256, 41, 4, 20, 1, 16, False, False, 4000, 0, 1), 32),
    (('d3_grp_256_1024', 256, 1024, 41, 4, 20, 1, 64, False, False, 1000, 0, 1), 32),
    (('d4_grp_1024_1024', 1024, 1024, 41, 4, 20, 1, 256, False, False, 250, 0, 1), 32),
    (('d5_1024_1024_k5_T63', 1024, 1024, 5, 1, 2, 1, 1, False, False, 63, 0, 1), 64),
    (('d5_1024_1024_k5_T32', 1024, 1024, 5, 1, 2, 1, 1, False, False, 32, 0, 1), 64),
    (('dout_1024_16_k3_T63', 1024, 16, 3, 1, 1, 1, 1, False, False, 63, 0, 0), 64),
    # encoder / decoder resampling convs, bottleneck convs, heads
    (('enc0_1_16_k7', 1, 16, 7, 1, 3, 1, 1, True, False, 16000, 0, 0), 32),
    (('down_16_32_s2', 16, 32, 4, 2, 1, 1, 1, False, False, 16000, 1, 0), 32),
    (('down_32_64_s2', 32, 64, 4, 2, 1, 1, 1, False, False, 8000, 1, 0), 32),
    (('down_64_128_s8', 64, 128, 16, 8, 4, 1, 1, False, False, 4000, 1, 0), 32),
    (('down_128_256_s10', 128, 256, 20, 10, 5, 1, 1, False, False, 500, 1, 0), 32),
    (('bott_256_256_k7_T50', 256, 256, 7, 1, 3, 1, 1, False, False, 50, 1, 0), 32),
    (('bott_256_128_k7_T50', 256, 128, 7, 1, 3, 1, 1, False, False, 50, 1, 0), 32),
    (('up_256_128_s10', 256, 128, 20, 10, 5, 1, 1, False, True, 50, 1, 0), 32),
    (('up_128_64_s8', 128, 64, 16, 8, 4, 1, 1, False, True, 500, 1, 0), 32),
    (('up_64_32_s2', 64, 32, 4, 2, 1, 1, 1, False, True, 4000, 1, 0), 32),
    (('up_32_16_s2', 32, 16, 4, 2, 1, 1, 1, False, True, 8000, 1, 0), 32),
    (('head_16_1_tanh', 16, 1, 7, 1, 3, 1, 1, True, False, 16000, 1, 2), 32),
    (('head_64_1_tanh', 64, 1, 7, 1, 3, 1, 1, True, False, 4000, 1, 2), 32),
    # excitation pyramid
    (('exc_in_1_8_k7', 1, 8, 7, 1, 3, 1, 1, True, False, 16000, 0, 0), 32),
    (('exc_8_8_k5', 8, 8, 5, 1, 2, 1, 1, False, False, 8000, 1, 0), 32),
    (('exc_down_8_8_s2', 8, 8, 4, 2, 1, 1, 1, False, False, 16000, 0, 0), 32),
    (('fir_dw8_k33_s2', 8, 8, 33, 2, 16, 1, 8, False, False, 16000, 0, 0), 32),
    (('fir_1_k129_s2', 1, 1, 129, 2, 64, 1, 1, False, False, 16000, 0, 0), 32),
    # STFT of the log-mel loss: Conv1d(1 -> 2*1028, K = 2048, stride 512) on the reflect-padded second of audio
    (('stft_1_2056_k2048_s512', 1, 2056, 2048, 512, 0, 1, 1, False, False, 18048, 0, 0), 16),
    (('mel_1028_80_k1', 1028, 80, 1, 1, 0, 1, 1, False, False, 32, 0, 0), 16),
]


@pytest.mark.parametrize('case,B', LAUNCH_CONV, ids=[f'{c[0]}_B{b}' for c, b in LAUNCH_CONV])
def test_launch_shape_conv(case, B, dev):
    with traced() as tr:
        errs = OPS.conv_case_errors(case, dev, 0, B=B)
    assert max(errs.values()) < TOL, (errs, sorted(tr.names))


# fused conditioning forward (LXF_COND) + its one-pass backward at the four decoder stages: (C, T, B)
LAUNCH_COND = [(16, 16000, 16), (32, 8000, 16), (64, 4000, 16), (128, 500, 32)]


@pytest.mark.parametrize('fused_bwd', [True, False], ids=['bwd1launch', 'bwd2launch'])
@pytest.mark.parametrize('fused_fwd', [False, True], ids=['fwd2launch', 'fwd1launch'])
@pytest.mark.parametrize('cfg', LAUNCH_COND, ids=[f'C{c}_T{t}_B{b}' for c, t, b in LAUNCH_COND])
def test_launch_shape_fused_conditioning(cfg, fused_fwd, fused_bwd, dev):
    """FiLM conditioning path (ops.FilmCondFn) at the four decoder stages: forward as the two-launch default and as the single
    fused launch (tdvc_film_cond_fwd, prologue kind LXF_COND); backward as the step's ONE launch behind cond_var.2's output
    gradient (tdvc_film_cond_bwd: the 136-channel gradient stays on chip; mask from the forward's sign bits, or from the fp32
    intermediate where T % 32 != 0) and as the two-launch path it replaces (cond_var.2 input-grad + tdvc_film_cond0_bwd)."""
    ops = importlib.import_module('td-vc-gan_amd').ops
    old = ops.FUSED_COND_FWD, ops.FUSED_COND_BWD
    ops.FUSED_COND_FWD, ops.FUSED_COND_BWD = fused_fwd, fused_bwd
    try:
        with traced() as tr:
            errs = OPS.film_cond_errors(cfg, dev)
    finally:
        ops.FUSED_COND_FWD, ops.FUSED_COND_BWD = old
    assert max(errs.values()) < TOL, (errs, sorted(tr.names))
    assert any(n.startswith('conv_lean_kernel') and n.endswith(',4,0>') for n in tr.names) == fused_fwd, sorted(tr.names)
    if fused_bwd:
        bits = ops.SIGN_BIT_MASKS and not fused_fwd and cfg[1] % 32 == 0 and cfg[1] >= 512
        assert ('film_cond_bwd_kernel<true>' if bits else 'film_cond_bwd_kernel<false>') in tr.names, sorted(tr.names)
        assert 'film_cond0_bwd_kernel' not in tr.names, sorted(tr.names)
    else:
        assert 'film_cond0_bwd_kernel' in tr.names, sorted(tr.names)


# ------------------------------------------------------------------------------------------------ folded short sequences
# conv_lean_kernel<..., FOLD = true>: T = 16 / 32 put 4 / 2 samples into one 64-column tile (discriminator layer 5 of the two
# sub-sampled discriminators). Ragged batches on purpose (B % fold != 0), the step's own (prologue, epilogue) pairs, and the
# launch shapes of the step. (name, cin, cout, k, stride, pad, dil, groups, reflect, transposed, T, pre, post), B
FOLD_CASES = [(('fold_c64_k5_T16_lrelu', 64, 64, 5, 1, 2, 1, 1, False, False, 16, 0, 1), 7),
              (('fold_c128_k5_T32_lrelu', 128, 128, 5, 1, 2, 1, 1, False, False, 32, 0, 1), 5),
              (('fold_c64_k3_T16_plain', 64, 48, 3, 1, 1, 1, 1, False, False, 16, 0, 0), 9),
              (('d5_1024_k5_T16', 1024, 1024, 5, 1, 2, 1, 1, False, False, 16, 0, 1), 64),
              (('d5_1024_k5_T32', 1024, 1024, 5, 1, 2, 1, 1, False, False, 32, 0, 1), 64)]


@pytest.mark.parametrize('case,B', FOLD_CASES, ids=[c[0][0] for c in FOLD_CASES])
def test_folded_short_sequences(case, B, dev):
    with traced() as tr:
        errs = OPS.conv_case_errors(case, dev, 0, B=B)
    assert max(errs.values()) < TOL, (errs, sorted(tr.names))
    folded = [n for n in tr.names if n.startswith('conv_lean_kernel') and n.endswith(',true>')]
    assert len(folded) >= 2, sorted(tr.names)          # forward and input-grad both run folded tiles
    lib = importlib.import_module('td-vc-gan_amd')._lib.lib()
    lib.tdvc_debug_knob(4, 1)                          # the same case unfolded: same results up to the accumulation order
    try:
        errs1 = OPS.conv_case_errors(case, dev, 0, B=B)
    finally:
        lib.tdvc_debug_knob(4, 0)
    assert max(errs1.values()) < TOL, errs1


# ------------------------------------------------------------------------------------------------ split-bf16 x6 weight-grad
# conv_wgrad_x6_kernel: the weight gradient of 3-tap convs with 65..144 input channels (FiLM cond_var.2) on the bf16 matrix pipe
# at fp32 accuracy (three exact bf16 pieces per operand, six piece products). Same 2e-5 bar against float64 as the exact-fp32
# kernels, at ragged shapes (T not a multiple of the 32-step chunk, channel counts off the 16-tile grid, bias on / off through the
# conv cases' bias gradient) and at the step's launch shapes; knob 5 runs the exact-fp32 MFMA kernel on the same case.
X6_CASES = [(('x6_cond2_136_32_T500', 136, 32, 3, 1, 1, 1, 1, False, False, 500, 1, 0), 3),
            (('x6_72_96_T260', 72, 96, 3, 1, 1, 1, 1, False, False, 260, 0, 0), 2),
            (('x6_144_64_T2048', 144, 64, 3, 1, 1, 1, 1, False, False, 2048, 1, 0), 2),
            (('x6_cond2_136_32_T16000', 136, 32, 3, 1, 1, 1, 1, False, False, 16000, 1, 0), 32),
            (('x6_cond2_136_256_T500', 136, 256, 3, 1, 1, 1, 1, False, False, 500, 1, 0), 32)]


@pytest.mark.parametrize('case,B', X6_CASES, ids=[c[0][0] for c in X6_CASES])
def test_split_bf16_x6_weight_grad(case, B, dev):
    with traced() as tr:
        errs = OPS.conv_case_errors(case, dev, 0, B=B)
    assert max(errs.values()) < TOL, (errs, sorted(tr.names))
    assert 'conv_wgrad_x6_kernel' in tr.names, sorted(tr.names)
    lib = importlib.import_module('td-vc-gan_amd')._lib.lib()
    lib.tdvc_debug_knob(5, 1)                          # exact-fp32 MFMA path on the same case
    try:
        with traced() as tr1:
            errs1 = OPS.conv_case_errors(case, dev, 0, B=B)
    finally:
        lib.tdvc_debug_knob(5, 0)
    assert max(errs1.values()) < TOL and 'conv_wgrad_x6_kernel' not in tr1.names, (errs1, sorted(tr1.names))
    assert errs['dw'] <= 4 * errs1['dw'] + 1e-7, (errs['dw'], errs1['dw'])      # not worse than the fp32 kernel's own rounding


# conv_fwd_x6.hip: the same arithmetic for the forward of a 3-tap conv with 64 < Cin <= 160 (cond_var.2): x split on the fly while it is
# transposed into LDS, the weight pieces cached once per optimizer step (ops._weight_planes_x6). Ragged T (not a multiple of the 128-step
# tile, down to the minimum 128), Cin at both ends of the window (65, 160), Cout 32-multiples handled by the 2- and the 4-tile variant,
# with / without LeakyReLU on load; knob 6 = the exact-fp32 MFMA kernel on the same case.
X6_FWD_CASES = [(('x6f_136_32_T500', 136, 32, 3, 1, 1, 1, 1, False, False, 500, 1, 0), 3),
                (('x6f_65_64_T132', 65, 64, 3, 1, 1, 1, 1, False, False, 132, 0, 0), 2),
                (('x6f_160_96_T128', 160, 96, 3, 1, 1, 1, 1, False, False, 128, 1, 0), 2),
                (('x6f_136_128_T1000', 136, 128, 3, 1, 1, 1, 1, False, False, 1000, 1, 0), 3),
                (('x6f_cond2_136_64_T8000', 136, 64, 3, 1, 1, 1, 1, False, False, 8000, 1, 0), 32),
                (('x6f_cond2_136_256_T500', 136, 256, 3, 1, 1, 1, 1, False, False, 500, 1, 0), 32)]


@pytest.mark.parametrize('case,B', X6_FWD_CASES, ids=[c[0][0] for c in X6_FWD_CASES])
def test_split_bf16_x6_forward(case, B, dev):
    P = importlib.import_module('td-vc-gan_amd')
    lib = P._lib.lib()
    old = P.ops.X6_FWD_MIN_COUT
    P.ops.X6_FWD_MIN_COUT = 32                         # also the variants the step does not route (kept correct, selectable)
    try:
        with traced() as tr:
            errs = OPS.conv_case_errors(case, dev, 0, B=B)
        assert max(errs.values()) < TOL, (errs, sorted(tr.names))
        assert any(n.startswith('conv_fwd_x6_kernel') for n in tr.names), sorted(tr.names)
        lib.tdvc_debug_knob(6, 1)                      # exact-fp32 MFMA path on the same case
        try:
            with traced() as tr1:
                errs1 = OPS.conv_case_errors(case, dev, 0, B=B)
        finally:
            lib.tdvc_debug_knob(6, 0)
    finally:
        P.ops.X6_FWD_MIN_COUT = old
    assert max(errs1.values()) < TOL and not any(n.startswith('conv_fwd_x6_kernel') for n in tr1.names), (errs1, sorted(tr1.names))
    assert errs['y'] <= 4 * errs1['y'] + 1e-7, (errs['y'], errs1['y'])


def test_split_bf16_x6_forward_weight_cache(dev):
    """The cached bf16 weight pieces follow the arena's weights: they key on arena.version, which every materialize() bumps. A stale
    cache would reproduce the old output exactly; here the layer's weight is scaled by 1.5 in place between two forwards."""
    from common import build_models
    P = importlib.import_module('td-vc-gan_amd')
    ops, L = P.ops, P._lib
    G, _ = build_models(dev)
    ar = G.ensure_arena(dev)
    name, mod = next((n, m) for n, m in G.named_modules() if n.endswith('cond_var.2') and m.spec.cout >= ops.X6_FWD_MIN_COUT and m.spec.cout % 32 == 0)
    spec = mod.spec
    torch.manual_seed(3)
    x = torch.randn(2, spec.cin, 512, device=dev)
    with traced() as tr:
        y1 = ops.conv_fwd_raw(spec, x, ops._xf(L.XF_NONE)).clone()
    assert any(n.startswith('conv_fwd_x6_kernel') for n in tr.names), sorted(tr.names)
    wkey = name + ('.weight_g' if name + '.weight_g' in ar.params else '.weight')
    bias = ar.params[name + '.bias'].detach().view(1, -1, 1)
    with torch.no_grad():
        ar.params[wkey].mul_(1.5)
    try:
        ar.materialize()
        y2 = ops.conv_fwd_raw(spec, x, ops._xf(L.XF_NONE))
        ref = 1.5 * (y1 - bias) + bias
        assert float((y2 - ref).norm() / ref.norm()) < 1e-6, float((y2 - ref).norm() / ref.norm())
    finally:
        with torch.no_grad():
            ar.params[wkey].div_(1.5)
        ar.materialize()


# tdvc_film_cond_fwd_x6 (conv_fwd_x6.hip, film_cond_fwd_x6_kernel): the conditioning forward in ONE launch -- cond_var.0's excitation window
# computed per tile inside the split-bf16 cond_var.2 forward. Against float64: gb, the stored cv0; the sign bits must equal (stored cv0 > 0)
# bit for bit (they are packed from the very registers that are stored). Ragged T (not a multiple of the 128-step tile, of 32; the minimum 128),
# n_cond at both ends of the window, 1..4 output-channel blocks (only the first stores cv0 / bits), edge tiles (3-valued bias at t = 0, T - 1).
COND_FWD_X6 = [(136, 32, 1024, 3, True), (136, 64, 500, 2, False), (72, 96, 128, 2, True), (160, 256, 640, 2, True), (136, 128, 4000, 4, True),
               (68, 32, 132, 1, False), (144, 64, 260, 1, False), (100, 32, 2048, 1, True)]


@pytest.mark.parametrize('nc,C2,T,B,with_bits', COND_FWD_X6, ids=[f'nc{a}_C{b}_T{c}' for a, b, c, _, _ in COND_FWD_X6])
def test_fused_cond_forward_x6(nc, C2, T, B, with_bits, dev):
    import ctypes as C
    P = importlib.import_module('td-vc-gan_amd')
    ops, arena, L = P.ops, P.arena, P._lib
    lib = L.lib()
    torch.manual_seed(nc + C2 + T)
    nv = 8
    w0 = torch.randn(nc, nc, 3, dtype=torch.float64) / (nc * 3) ** 0.5
    w2 = torch.randn(C2, nc, 3, dtype=torch.float64) / (nc * 3) ** 0.5
    b2 = torch.randn(C2, dtype=torch.float64) * 0.1
    exc = torch.randn(B, nv, T, dtype=torch.float64)
    k3 = torch.randn(B, nc, 3, dtype=torch.float64) * 0.3
    # float64 reference: the 3-valued bias is k3[..., 0] at t = 0, k3[..., 2] at t = T - 1, k3[..., 1] between
    bias3 = k3[:, :, 1:2].repeat(1, 1, T).clone()
    bias3[:, :, 0] = k3[:, :, 0]; bias3[:, :, T - 1] = k3[:, :, 2]
    cv_ref = torch.nn.functional.conv1d(exc, w0[:, nc - nv:, :], padding=1) + bias3
    gb_ref = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(cv_ref, 0.2), w2, b2, padding=1)
    d = lambda t: t.float().to(dev).contiguous()
    w0d, w2d, b2d, excd, k3d = d(w0), d(w2), d(b2), d(exc), d(k3)
    spec2 = ops.ConvSpec(nc, C2, 3, 1, 1, 1, 1, False)
    spec2.slot = arena.ConvSlot(w2d.data_ptr(), b2d.data_ptr(), 0, 0, False, None, 0)
    planes = ops._weight_planes_x6(spec2, dev)
    cv0 = torch.full((B, nc, T), float('nan'), device=dev)
    gb = torch.full((B, C2, T), float('nan'), device=dev)
    bits = torch.zeros((B, nc, T // 32), dtype=torch.int32, device=dev) if with_bits else None
    a = L.FilmCondArgs(B, T, nc, nv, C2, excd.data_ptr(), excd.stride(0), w0d.data_ptr(), k3d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(),
                       cv0.data_ptr(), cv0.stride(0), gb.data_ptr(), gb.stride(0), 0.2)
    st = torch.cuda.current_stream(dev).cuda_stream
    lib.tdvc_debug_poison_lds(0xFFFFFFFF, st)
    with traced() as tr:
        L.check(lib.tdvc_film_cond_fwd_x6(C.byref(a), planes.data_ptr(), bits.data_ptr() if with_bits else None, bits.stride(0) if with_bits else 0, st))
        torch.cuda.synchronize()
    assert any(n.startswith('film_cond_fwd_x6_kernel') for n in tr.names), sorted(tr.names)
    assert bool(torch.isfinite(gb).all()) and bool(torch.isfinite(cv0).all())
    e_cv, e_gb = rel_l2(cv0, cv_ref), rel_l2(gb, gb_ref)
    assert e_cv < TOL and e_gb < TOL, (e_cv, e_gb)
    # inference form: no intermediate, no bits -> the same gb bit for bit
    gb2 = torch.full((B, C2, T), float('nan'), device=dev)
    a2 = L.FilmCondArgs(B, T, nc, nv, C2, excd.data_ptr(), excd.stride(0), w0d.data_ptr(), k3d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(),
                        None, 0, gb2.data_ptr(), gb2.stride(0), 0.2)
    L.check(lib.tdvc_film_cond_fwd_x6(C.byref(a2), planes.data_ptr(), None, 0, st))
    torch.cuda.synchronize()
    assert torch.equal(gb2, gb)
    if with_bits:
        want = (cv0 > 0).view(B, nc, T // 32, 32).to(torch.int64)
        words = (want << torch.arange(32, device=dev, dtype=torch.int64)).sum(-1)
        got = bits.to(torch.int64) & 0xFFFFFFFF
        assert torch.equal(got, words), int((got != words).sum())


def test_fused_cond_forward_x6_matches_two_launches(dev):
    """ops.film_cond with the fused forward vs the two-launch forward: same gb and same gradients (dexc, dk3, weight gradients) within fp32
    rounding -- the backward consumes the fused kernel's cv0 / sign bits."""
    P = importlib.import_module('td-vc-gan_amd')
    ops, arena, L = P.ops, P.arena, P._lib
    torch.manual_seed(77)
    B, nc, nv, C2, T = 3, 136, 8, 64, 1024
    w0 = (torch.randn(nc, nc, 3) / (nc * 3) ** 0.5).to(dev)
    w2 = (torch.randn(C2, nc, 3) / (nc * 3) ** 0.5).to(dev)
    b2 = (torch.randn(C2) * 0.1).to(dev)
    w2t = w2.permute(1, 0, 2).contiguous()
    res = {}
    for fused in (True, False):
        dw0, dw2, db2 = torch.zeros_like(w0), torch.zeros_like(w2), torch.zeros_like(b2)
        sv = ops.ConvSpec(nv, nc, 3, 1, 1, 1, 1, False, w_cin=nc, w_cin_off=nc - nv)
        sv.slot = arena.ConvSlot(w0.data_ptr(), 0, dw0.data_ptr(), 0, True, None, 0)
        s2 = ops.ConvSpec(nc, C2, 3, 1, 1, 1, 1, False)
        s2.slot = arena.ConvSlot(w2.data_ptr(), b2.data_ptr(), dw2.data_ptr(), db2.data_ptr(), True, None, w2t.data_ptr())
        torch.manual_seed(5)
        exc = torch.randn(B, nv, T, device=dev, requires_grad=True)
        k3 = (torch.randn(B, nc, 3, device=dev) * 0.3).requires_grad_(True)
        cot = torch.randn(B, C2, T, device=dev)
        old = ops.FUSED_COND_FWD_X6
        ops.FUSED_COND_FWD_X6 = fused
        try:
            with traced() as tr:
                gb = ops.film_cond(exc, k3, sv, s2)
                gb.backward(cot)
                ops.fold_flush(dev)
                torch.cuda.synchronize()
        finally:
            ops.FUSED_COND_FWD_X6 = old
        assert any(n.startswith('film_cond_fwd_x6_kernel') for n in tr.names) == fused, sorted(tr.names)
        res[fused] = dict(gb=gb.detach().clone(), dexc=exc.grad.clone(), dk3=k3.grad.clone(), dw0=dw0.clone(), dw2=dw2.clone(), db2=db2.clone())
    for k in res[True]:
        e = rel_l2(res[True][k], res[False][k].double().cpu())
        assert e < 2e-5, (k, e)        # (a LeakyReLU element within rounding of zero may flip between the two cv0 roundings: 1024 x 136 x 3 elements, none expected)


# ------------------------------------------------------------------------------------------------ sign-bit masks
# tdvc_conv_fwd_args.sign_bits / tdvc_conv_dgrad_args.x_sign_bits: the forward epilogue packs (y > 0) into one bit per element,
# the LeakyReLU-mask epilogue of the next layer's input-grad reads those words instead of the fp32 tensor (FiLM conditioning:
# cond_var.0 -> LeakyReLU -> cond_var.2). Exact by construction: the bits must equal (y > 0) and the input-grad must be
# bit-identical to the one computed from the fp32 mask source -- on every tile that can carry the words (two or four
# 16-column sub-tiles per wave), with a partial channel tile (40 = 2.5 x 16), a partial time tile and a 3-valued edge bias.
SIGN_CFG = [-1, 0, 1, 2, 4, 5, 6, 3]      # 3 = one sub-tile per wave: the launcher must move to the 16 x 256 tile


@pytest.mark.parametrize('cfg', SIGN_CFG, ids=[f'tile{c}' for c in SIGN_CFG])
def test_sign_bit_masks(cfg, dev):
    P = importlib.import_module('td-vc-gan_amd')
    ops, arena, L = P.ops, P.arena, P._lib
    torch.manual_seed(100 + cfg)
    B, cin, cmid, cout, T = 3, 8, 40, 24, 544
    w0 = (torch.randn(cmid, cin, 3) / (cin * 3) ** 0.5).to(dev)
    b0 = (torch.randn(cmid) * 0.1).to(dev)
    k3 = (torch.randn(B, cmid, 3) * 0.3).to(dev)
    w2 = (torch.randn(cout, cmid, 3) / (cmid * 3) ** 0.5).to(dev)
    w2t = w2.permute(1, 0, 2).contiguous()
    s0 = ops.ConvSpec(cin, cmid, 3, 1, 1, 1, 1, False)
    s0.slot = arena.ConvSlot(w0.data_ptr(), b0.data_ptr(), 0, 0, False, None, 0)
    s2 = ops.ConvSpec(cmid, cout, 3, 1, 1, 1, 1, False)
    s2.slot = arena.ConvSlot(w2.data_ptr(), 0, 0, 0, False, None, w2t.data_ptr())
    x = torch.randn(B, cin, T, device=dev)
    dy = torch.randn(B, cout, T, device=dev)
    bits = torch.zeros(B, cmid, T // 32, dtype=torch.int32, device=dev)

    def run():
        y = ops.conv_fwd_raw(s0, x, ops._xf(), bias3=k3, sign_bits=bits)
        dx_bits = ops.conv_dgrad_raw(s2, dy, ops._xf(), T, L.DG_MASK_LRELU, x_in=None, x_bits=bits)
        dx_f32 = ops.conv_dgrad_raw(s2, dy, ops._xf(), T, L.DG_MASK_LRELU, x_in=y)
        return y, dx_bits, dx_f32
    if cfg >= 0:
        (y, dx_bits, dx_f32), names = _run_forced(cfg, run)
    else:
        with traced() as tr:
            y, dx_bits, dx_f32 = run()
        names = tr.names
    torch.cuda.synchronize()
    want = ((y > 0).reshape(B, cmid, T // 32, 32).long() << torch.arange(32, device=dev)).sum(-1)
    got = bits.long() & 0xFFFFFFFF
    assert torch.equal(got, want), f'{int((got != want).sum())} sign words differ'
    assert float((y > 0).float().mean()) > 0.2 and float((y <= 0).float().mean()) > 0.2      # the mask is not trivial
    if cfg in (-1, 3):      # the bit launch moved to another tile than the fp32 one: another channel-chunk order, same math
        assert float((dx_bits - dx_f32).norm() / dx_f32.norm()) < 1e-6
    else:
        assert torch.equal(dx_bits, dx_f32)
    assert float(dx_bits.abs().max()) > 0
    eff = 0 if cfg == 3 else cfg
    if cfg >= 0:
        assert lean_name(eff, LXF_ACT, EPI_FWD) in names and lean_name(eff, LXF_ACT, EPI_MASK) in names, sorted(names)


def test_sign_bits_refused_outside_contract(dev):
    """Shapes that cannot carry the words (T % 32 != 0) are refused, not silently computed without the bits."""
    P = importlib.import_module('td-vc-gan_amd')
    ops, arena, L = P.ops, P.arena, P._lib
    w = torch.randn(16, 8, 3, device=dev)
    s = ops.ConvSpec(8, 16, 3, 1, 1, 1, 1, False)
    s.slot = arena.ConvSlot(w.data_ptr(), 0, 0, 0, False, None, 0)
    x = torch.randn(2, 8, 500, device=dev)
    bits = torch.zeros(2, 16, 16, dtype=torch.int32, device=dev)
    with pytest.raises(L.TdvcError):
        ops.conv_fwd_raw(s, x, ops._xf(), sign_bits=bits)
