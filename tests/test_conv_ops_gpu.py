"""GPU parity of the conv C-ABI (tdvc_conv_fwd / dgrad / wgrad) against torch CPU float64 autograd,
for every conv shape class on the path, on both the MFMA and the scalar kernels."""
import importlib

import pytest
import torch
import torch.nn.functional as F

from common import traced

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 2e-5   # fp32 accumulation vs float64 reference, rel-L2 per tensor


def _mods():
    pkg = importlib.import_module('td-vc-gan_amd')
    return pkg.ops, pkg._lib, pkg.arena


# (name, cin, cout, k, stride, pad, dil, groups, reflect, transposed, T, pre, post)
CASES = [
    ('dil_c16_k3_d1', 16, 16, 3, 1, 1, 1, 1, True, False, 1000, 1, 0),
    ('dil_c16_k11_d5', 16, 16, 11, 1, 25, 5, 1, True, False, 700, 1, 0),
    ('dil_c32_k7_d3', 32, 32, 7, 1, 9, 3, 1, True, False, 520, 1, 0),
    ('dil_c64_k3_d5', 64, 64, 3, 1, 5, 5, 1, True, False, 300, 1, 0),
    ('dil_c128_k11_d1', 128, 128, 11, 1, 5, 1, 1, True, False, 100, 1, 0),
    ('dil_c256_k11_d5_T50', 256, 256, 11, 1, 25, 5, 1, True, False, 50, 1, 0),
    ('pw_c64', 64, 64, 1, 1, 0, 1, 1, False, False, 333, 1, 0),
    ('pw_c16', 16, 16, 1, 1, 0, 1, 1, False, False, 1000, 1, 0),
    ('cond0_136', 136, 136, 3, 1, 1, 1, 1, False, False, 260, 0, 0),
    ('cond2_136_32', 136, 32, 3, 1, 1, 1, 1, False, False, 260, 1, 0),
    ('in_1_16_k7', 1, 16, 7, 1, 3, 1, 1, True, False, 900, 0, 0),
    ('head_16_1_tanh', 16, 1, 7, 1, 3, 1, 1, True, False, 900, 1, 2),
    ('d0_1_16_k15_lrelu', 1, 16, 15, 1, 7, 1, 1, True, False, 800, 0, 1),
    ('k7_256_128_T50', 256, 128, 7, 1, 3, 1, 1, False, False, 50, 1, 0),
    ('d5_256_256_k5_lrelu', 256, 256, 5, 1, 2, 1, 1, False, False, 63, 0, 1),
    ('dout_256_16_k3', 256, 16, 3, 1, 1, 1, 1, False, False, 63, 0, 0),
    ('exc_8_8_k5', 8, 8, 5, 1, 2, 1, 1, False, False, 500, 1, 0),
    ('down_16_32_s2', 16, 32, 4, 2, 1, 1, 1, False, False, 1000, 1, 0),
    ('down_64_128_s8', 64, 128, 16, 8, 4, 1, 1, False, False, 800, 1, 0),
    ('down_128_256_s10', 128, 256, 20, 10, 5, 1, 1, False, False, 500, 1, 0),
    ('exc_down_8_8_s8', 8, 8, 16, 8, 4, 1, 1, False, False, 640, 0, 0),
    ('grp_16_64_k41_s4', 16, 64, 41, 4, 20, 1, 4, False, False, 1000, 0, 1),
    ('grp_64_256_k41_s4', 64, 256, 41, 4, 20, 1, 16, False, False, 500, 0, 1),
    ('grp_64_64_k41_s4_g16', 64, 64, 41, 4, 20, 1, 16, False, False, 250, 0, 1),
    ('up_256_128_s10', 256, 128, 20, 10, 5, 1, 1, False, True, 50, 1, 0),
    ('up_128_64_s8', 128, 64, 16, 8, 4, 1, 1, False, True, 100, 1, 0),
    ('up_32_16_s2', 32, 16, 4, 2, 1, 1, 1, False, True, 700, 1, 0),
    ('fir_dw8_k33_s2', 8, 8, 33, 2, 16, 1, 8, False, False, 1000, 0, 0),
    ('fir_dw8_k129_s8', 8, 8, 129, 8, 64, 1, 8, False, False, 1024, 0, 0),
    ('fir_1_k129_s2', 1, 1, 129, 2, 64, 1, 1, False, False, 1000, 0, 0),
    ('stft_like_1_40_k64_s16', 1, 40, 64, 16, 0, 1, 1, False, False, 640, 0, 0),
    ('linear_16_128_T1', 16, 128, 1, 1, 0, 1, 1, False, False, 1, 0, 0),
]


def _torch_ref(x, w, b, add, c):
    (_, cin, cout, k, s, p, d, g, reflect, transposed, T, pre, post) = c
    h = F.leaky_relu(x, 0.2) if pre else x
    if transposed:
        y = F.conv_transpose1d(h, w, b, stride=s, padding=p)
    else:
        if reflect and p > 0:
            h = F.pad(h, (p, p), mode='reflect')
            y = F.conv1d(h, w, b, stride=s, dilation=d, groups=g)
        else:
            y = F.conv1d(h, w, b, stride=s, padding=p, dilation=d, groups=g)
    if post == 1:
        y = F.leaky_relu(y, 0.2)
    elif post == 2:
        y = torch.tanh(y)
    if add is not None:
        y = y + add
    return y


def conv_case_errors(case, dev, generic=0, B=3):
    """Forward, input-grad, weight-grad and bias-grad of one conv case through the C ABI vs float64 CPU autograd."""
    ops, L, arena = _mods()
    (name, cin, cout, k, s, p, d, g, reflect, transposed, T, pre, post) = case
    torch.manual_seed(sum(map(ord, name)))
    wshape = (cin, cout // g, k) if transposed else (cout, cin // g, k)
    x = torch.randn(B, cin, T, dtype=torch.float64)
    w = torch.randn(wshape, dtype=torch.float64) / (wshape[1] * k) ** 0.5
    b = torch.randn(cout, dtype=torch.float64) * 0.1
    spec = ops.ConvSpec(cin, cout, k, s, p, d, g, reflect, transposed)
    tout = spec.tout(T)
    add = torch.randn(B, cout, tout, dtype=torch.float64) if name.startswith('exc_8') else None
    wd, bd = w.float().to(dev).contiguous(), b.float().to(dev).contiguous()
    dw, db = torch.zeros_like(wd), torch.zeros_like(bd)
    wt = wd.permute(1, 0, 2).contiguous() if (not transposed and g == 1 and s == 1) else None   # [Cin][Cout][K] copy for the lean dgrad
    spec.slot = arena.ConvSlot(wd.data_ptr(), bd.data_ptr(), dw.data_ptr(), db.data_ptr(), True, None, wt.data_ptr() if wt is not None else 0)
    L.lib().tdvc_set_force_generic(generic)
    try:
        xd = x.float().to(dev).requires_grad_(True)
        addd = add.float().to(dev) if add is not None else None
        y = ops.ConvFn.apply(xd, addd, None, spec, pre, post)
        # float64 reference. A post-LeakyReLU layer stores its post-activation map and the backward kernels take the
        # LeakyReLU mask from THAT map (model/discriminator.py:20: in-place activation); the reference uses the same mask,
        # so an output within rounding of the kink cannot land on different sides of it in the two computations (with
        # ~10^7 outputs per launch-shape case one such element is expected, and one flip moves dx/dw/db by ~1e-4).
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        if post == 1:
            mask = ((y.detach() - (addd if addd is not None else 0)) > 0).cpu()
            yr = _torch_ref(xr, wr, br, None, case[:12] + (0,))
            yr = yr * torch.where(mask, 1.0, 0.2)
            if add is not None:
                yr = yr + add
        else:
            yr = _torch_ref(xr, wr, br, add, case)
        cot = torch.randn_like(yr)
        (yr * cot).sum().backward()
        assert y.shape == yr.shape
        e_y = rel_l2(y, yr)
        y.backward(cot.float().to(dev))
        torch.cuda.synchronize()
        e_dx, e_dw, e_db = rel_l2(xd.grad, xr.grad), rel_l2(dw, wr.grad), rel_l2(db, br.grad)
    finally:
        L.lib().tdvc_set_force_generic(0)
    return dict(y=e_y, dx=e_dx, dw=e_dw, db=e_db)


@pytest.mark.parametrize('generic', [0, 1], ids=['mfma', 'scalar'])
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_conv_fwd_bwd(case, generic, dev):
    errs = conv_case_errors(case, dev, generic)
    assert max(errs.values()) < TOL, errs


POISON_CASES = ['grp_64_64_k41_s4_g16', 'grp_16_64_k41_s4', 'dil_c16_k11_d5', 'dil_c64_k3_d5', 'cond2_136_32', 'd5_256_256_k5_lrelu',
                'down_64_128_s8', 'up_128_64_s8', 'fir_dw8_k33_s2', 'k7_256_128_T50']


@pytest.mark.parametrize('case', [c for c in CASES if c[0] in POISON_CASES], ids=[c[0] for c in CASES if c[0] in POISON_CASES])
def test_conv_with_poisoned_lds(case, dev):
    """Every kernel family with the LDS of all CUs pre-filled with NaN bit patterns (tdvc_debug_poison_lds): a kernel that
    reads LDS words it never staged -- masked by a zero WEIGHT only -- turns them into NaN (0 * NaN) instead of passing on
    whatever finite values the previous kernel happened to leave (small_group_fwd_kernel, taps past K)."""
    _, L, _ = _mods()
    st = torch.cuda.current_stream(dev).cuda_stream
    L.check(L.lib().tdvc_debug_poison_lds(0xFFFFFFFF, st))
    errs = conv_case_errors(case, dev, 0)
    assert all(e == e for e in errs.values()) and max(errs.values()) < TOL, errs


@pytest.mark.parametrize('cfg', [(16, 3, 1, 900, True, True), (32, 7, 3, 400, True, False), (64, 11, 5, 260, False, True),
                                 (128, 3, 1, 100, True, True), (256, 11, 5, 50, False, False)],
                         ids=['c16k3', 'c32k7d3', 'c64k11d5_enc', 'c128k3', 'c256k11d5_enc'])
def test_film_block(cfg, dev):
    """Fused FiLM residual block (model/generator.py:96-111) incl. MRF running-mean epilogue."""
    errs = film_block_errors(cfg, dev)
    assert max(errs.values()) < TOL, errs


def film_block_errors(cfg, dev, B=2):
    ops, L, arena = _mods()
    C, k, d, T, cond, with_acc = cfg
    torch.manual_seed(C + k)
    pad, scale = (k * d - d) // 2, 1.0 / 3.0
    x = torch.randn(B, C, T, dtype=torch.float64)
    gb = torch.randn(B, 2 * C, T, dtype=torch.float64) * 0.5 if cond else None
    acc = torch.randn(B, C, T, dtype=torch.float64) if with_acc else None
    w1 = torch.randn(C, C, k, dtype=torch.float64) / (C * k) ** 0.5
    b1 = torch.randn(C, dtype=torch.float64) * 0.1
    w2 = torch.randn(C, C, 1, dtype=torch.float64) / C ** 0.5
    b2 = torch.randn(C, dtype=torch.float64) * 0.1
    f = lambda t: t.float().to(dev).contiguous()
    # The second LeakyReLU acts on a COMPUTED tensor (h2 = h*(1+gamma)+beta, or h itself without FiLM): an element within
    # fp32 rounding of 0 may land on the other side of the kink on the GPU, and one such flip among ~10^7 elements (the
    # launch-shape cases) moves the gradients by ~1e-4 -- a property of the data, not of the kernel. Keep the data off
    # the kink: with FiLM, nudge beta where |h2| is tiny; without, take the mask from the GPU's own h (bit-identical to
    # what the fused block computes: same kernel, same inputs).
    with torch.no_grad():
        h0 = F.conv1d(F.pad(F.leaky_relu(x, 0.2), (pad, pad), mode='reflect'), w1, b1, dilation=d)
    mask2 = None
    if cond:
        ga0, be0 = gb.chunk(2, dim=1)
        h20 = h0 * (1 + ga0) + be0
        tau = 1e-4 * float(h20.pow(2).mean().sqrt())
        near = h20.abs() < tau
        gb[:, C:][near] += torch.where(h20[near] >= 0, 2 * tau, -2 * tau)
    else:
        cs0 = ops.ConvSpec(C, C, k, 1, pad, d, 1, True)
        w1d0, b1d0 = f(w1), f(b1)
        cs0.slot = arena.ConvSlot(w1d0.data_ptr(), b1d0.data_ptr(), 0, 0, False, None, 0)
        mask2 = (ops.conv_fwd_raw(cs0, f(x), ops._xf(L.XF_LRELU)) > 0).cpu()
    leaves = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)] + \
             [gb.clone().requires_grad_(True) if cond else None, acc.clone().requires_grad_(True) if with_acc else None]
    xr, w1r, b1r, w2r, b2r, gbr, accr = leaves
    h = F.conv1d(F.pad(F.leaky_relu(xr, 0.2), (pad, pad), mode='reflect'), w1r, b1r, dilation=d)
    if cond:
        ga, be = gbr.chunk(2, dim=1)
        h = h * (1 + ga) + be
    a2 = F.leaky_relu(h, 0.2) if mask2 is None else h * torch.where(mask2, 1.0, 0.2)
    out = scale * (F.conv1d(a2, w2r, b2r) + xr)
    if with_acc:
        out = out + accr
    cot = torch.randn_like(out)
    (out * cot).sum().backward()

    dts = {n: f(t) for n, t in dict(w1=w1, b1=b1, w2=w2, b2=b2).items()}
    grads = {n: torch.zeros_like(t) for n, t in dts.items()}
    cs = ops.ConvSpec(C, C, k, 1, pad, d, 1, True)
    ps = ops.ConvSpec(C, C, 1)
    w1t, w2t = dts['w1'].permute(1, 0, 2).contiguous(), dts['w2'].permute(1, 0, 2).contiguous()
    cs.slot = arena.ConvSlot(dts['w1'].data_ptr(), dts['b1'].data_ptr(), grads['w1'].data_ptr(), grads['b1'].data_ptr(), True, None, w1t.data_ptr())
    ps.slot = arena.ConvSlot(dts['w2'].data_ptr(), dts['b2'].data_ptr(), grads['w2'].data_ptr(), grads['b2'].data_ptr(), True, None, w2t.data_ptr())
    xd = f(x).requires_grad_(True)
    gbd = f(gb).requires_grad_(True) if cond else None
    accd = f(acc).requires_grad_(True) if with_acc else None
    y = ops.FilmBlockFn.apply(xd, gbd, accd, None, cs, ps, scale)
    y.backward(f(cot))
    torch.cuda.synchronize()
    errs = dict(y=rel_l2(y, out), dx=rel_l2(xd.grad, xr.grad), dw1=rel_l2(grads['w1'], w1r.grad), db1=rel_l2(grads['b1'], b1r.grad),
                dw2=rel_l2(grads['w2'], w2r.grad), db2=rel_l2(grads['b2'], b2r.grad))
    if cond:
        errs['dgb'] = rel_l2(gbd.grad, gbr.grad)
    if with_acc:
        errs['dacc'] = rel_l2(accd.grad, accr.grad)
    return errs


@pytest.mark.parametrize('cfg', [(16, 2048, 2), (64, 600, 3), (128, 260, 2), (32, 64, 2)], ids=['C16_T2048', 'C64_T600', 'C128_T260', 'C32_T64'])
@pytest.mark.parametrize('fused_bwd', [True, False], ids=['bwd1launch', 'bwd2launch'])
@pytest.mark.parametrize('fused_fwd', [False, True], ids=['fwd2launch', 'fwd1launch'])
def test_film_conditioning_fused(cfg, fused_fwd, fused_bwd, dev):
    """tdvc_film_cond_fwd / tdvc_film_cond_bwd / tdvc_film_cond0_bwd (FiLM conditioning path of model/generator.py:86-92,103 in
    the split formulation) against float64 autograd of the dense reference formulation
        gb = cond_var.2(LeakyReLU(cond_var.0(cat([emb.repeat(T), exc])))).
    Ragged cases on purpose: T = 600 / 260 / 64 are not multiples of the fused backward's 60-step chunk or of 32 (fp32 mask
    source), T = 2048 runs the sign-bit mask with a partial last chunk."""
    ops = _mods()[0]
    old = ops.FUSED_COND_FWD, ops.FUSED_COND_BWD
    ops.FUSED_COND_FWD, ops.FUSED_COND_BWD = fused_fwd, fused_bwd
    try:
        with traced() as tr:
            errs = film_cond_errors(cfg, dev)
    finally:
        ops.FUSED_COND_FWD, ops.FUSED_COND_BWD = old
    assert max(errs.values()) < TOL, errs
    assert any(n.startswith('film_cond_bwd_kernel') for n in tr.names) == fused_bwd, sorted(tr.names)


def film_cond_errors(cfg, dev):
    ops, L, arena = _mods()
    C, T, B = cfg
    n_const, n_var = 128, 8
    nc = n_const + n_var
    torch.manual_seed(C + T)
    emb = torch.randn(B, n_const, dtype=torch.float64)
    exc = torch.randn(B, n_var, T, dtype=torch.float64)
    w0 = torch.randn(nc, nc, 3, dtype=torch.float64) / (nc * 3) ** 0.5
    b0 = torch.randn(nc, dtype=torch.float64) * 0.1
    w2 = torch.randn(2 * C, nc, 3, dtype=torch.float64) / (nc * 3) ** 0.5
    b2 = torch.randn(2 * C, dtype=torch.float64) * 0.1
    f = lambda t: t.float().to(dev).contiguous()
    w0d, b0d, w2d, b2d = f(w0), f(b0), f(w2), f(b2)
    dw0, db0, dw2, db2 = (torch.zeros_like(t) for t in (w0d, b0d, w2d, b2d))
    w0t, w2t = w0d.permute(1, 0, 2).contiguous(), w2d.permute(1, 0, 2).contiguous()
    spec_const = ops.ConvSpec(n_const, nc, 3, pad=1, w_cin=nc, w_cin_off=0)
    spec_var = ops.ConvSpec(n_var, nc, 3, pad=1, w_cin=nc, w_cin_off=n_const)
    spec2 = ops.ConvSpec(nc, 2 * C, 3, pad=1)
    spec_const.slot = arena.ConvSlot(w0d.data_ptr(), b0d.data_ptr(), dw0.data_ptr(), db0.data_ptr(), True, None, w0t.data_ptr())
    spec_var.slot = arena.ConvSlot(w0d.data_ptr(), 0, dw0.data_ptr(), 0, True, None, w0t.data_ptr())
    spec2.slot = arena.ConvSlot(w2d.data_ptr(), b2d.data_ptr(), dw2.data_ptr(), db2.data_ptr(), True, None, w2t.data_ptr())
    embd, excd = f(emb).requires_grad_(True), f(exc).requires_grad_(True)
    if C % 32 == 0:     # the module's route: dedicated k3 kernel on the embedding itself
        k3 = ops.film_k3(embd, spec_const)
    else:               # generic route: the same conv on an explicit length-3 constant signal
        emb3 = embd.unsqueeze(2).expand(B, n_const, 3).contiguous()
        k3 = ops.conv(emb3, spec_const)
    gb = ops.film_cond(excd, k3, spec_var, spec2)
    # float64 reference of the dense formulation. The LeakyReLU between the two convs acts on the computed 136-channel
    # intermediate (up to 3.5e7 elements at the launch shapes): its mask is taken from the intermediate the GPU forward
    # stored for its own backward pass, so an element within rounding of the kink sits on the same side in both
    # computations (one flip would move every gradient by ~3e-4; see film_block_errors).
    mask = (gb.grad_fn.saved_tensors[1] > 0).cpu()
    embr, excr, w0r, b0r, w2r, b2r = [t.clone().requires_grad_(True) for t in (emb, exc, w0, b0, w2, b2)]
    c = torch.cat([embr.unsqueeze(2).expand(B, n_const, T), excr], dim=1)
    cv0r = F.conv1d(c, w0r, b0r, padding=1)
    gbr = F.conv1d(cv0r * torch.where(mask, 1.0, 0.2), w2r, b2r, padding=1)
    cot = torch.randn_like(gbr)
    (gbr * cot).sum().backward()
    assert gb.shape == gbr.shape and rel_l2(gb.grad_fn.saved_tensors[1], cv0r) < TOL
    gb.backward(f(cot))
    torch.cuda.synchronize()
    errs = dict(gb=rel_l2(gb, gbr), dexc=rel_l2(excd.grad, excr.grad), demb=rel_l2(embd.grad, embr.grad),
                dw0=rel_l2(dw0, w0r.grad), db0=rel_l2(db0, b0r.grad), dw2=rel_l2(dw2, w2r.grad), db2=rel_l2(db2, b2r.grad))
    return errs


def test_fanout_sum(dev):
    """ops.fanout: n aliases forward, ONE n-ary sum backward (tdvc_sum_n), incl. the > 16-source chaining and a consumer that
    sends no gradient."""
    ops, L, _ = _mods()
    for n, shape in ((3, (2, 16, 520)), (9, (3, 8, 333)), (36, (4, 128))):
        x = torch.randn(*shape, device=dev, requires_grad=True)
        ws = [torch.randn(*shape, device=dev) for _ in range(n)]
        outs = ops.fanout(x, n)
        assert all(o.data_ptr() == x.data_ptr() for o in outs)
        sum((o * w).sum() for o, w in list(zip(outs, ws))[1:]).backward()      # consumer 0 unused: its gradient is None
        ref = torch.stack([w.double() for w in ws[1:]]).sum(0)
        assert rel_l2(x.grad, ref) < 1e-6
    with torch.no_grad():
        y = torch.randn(2, 3, device=dev)
        assert all(o is y for o in ops.fanout(y, 4))


def test_l1_multi(dev):
    """tdvc_l1_multi_fwd / _bwd (all feature-matching terms in one launch) against torch: ragged sizes incl. an unaligned
    pair, a pair longer than one chunk, zero-fill entries, more pairs than one launch holds."""
    import ctypes as C
    _, L, _ = _mods()
    lib = L.lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    rs = torch.Generator().manual_seed(5)
    sizes = [40000, 7, 16384, 16385, 1023] + [300 + 3 * i for i in range(70)]
    big = torch.randn(sum(sizes) + 8, generator=rs).to(dev)
    a, b, off = [], [], 1                                  # offset 1 float: the first pairs are NOT 16-byte aligned
    for n in sizes:
        a.append(big[off:off + n]); off += n
    b = [torch.randn(n, generator=rs).to(dev) for n in sizes]
    da = [torch.full((n,), 7.0, device=dev) for n in sizes]
    zero = torch.full((5000,), 3.0, device=dev)
    w = [0.5 + 0.01 * i for i in range(len(sizes))]
    ents = [L.L1Pair(x.data_ptr(), y.data_ptr(), d.data_ptr(), x.numel(), wi) for x, y, d, wi in zip(a, b, da, w)]
    ents.insert(3, L.L1Pair(None, None, zero.data_ptr(), zero.numel(), 0.0))
    pairs = (L.L1Pair * len(ents))(*ents)
    out = torch.zeros(1, device=dev)
    up = torch.tensor([1.7], device=dev)
    L.check(lib.tdvc_l1_multi_fwd(pairs, len(ents), out.data_ptr(), st))
    L.check(lib.tdvc_l1_multi_bwd(pairs, len(ents), up.data_ptr(), st))
    torch.cuda.synchronize()
    ref = sum(wi * (x.double() - y.double()).abs().mean() for x, y, wi in zip(a, b, w))
    assert abs(float(out) - float(ref)) < 1e-5 * float(ref)
    for x, y, d, wi in zip(a, b, da, w):
        assert torch.allclose(d, torch.sign(x - y) * (wi * 1.7 / x.numel()), rtol=1e-6, atol=0)
    assert float(zero.abs().max()) == 0.0


def test_film_k3_multi_matches_single(dev):
    """ops.film_k3_multi (all FiLM blocks of a stage in one launch each way) against the per-block FilmK3Fn: outputs, weight /
    bias gradients per block, and the embedding gradient summed over the blocks."""
    ops, L, arena = _mods()
    torch.manual_seed(3)
    B, n_const, nc, nblk = 5, 128, 136, 9
    ws = [(torch.randn(nc, nc, 3, device=dev) / (nc * 3) ** 0.5) for _ in range(nblk)]
    bs = [torch.randn(nc, device=dev) * 0.1 for _ in range(nblk)]
    cots = [torch.randn(B, nc, 3, device=dev) for _ in range(nblk)]
    emb = torch.randn(B, n_const, device=dev)

    def run(multi):
        dws, dbs = [torch.zeros_like(w) for w in ws], [torch.zeros_like(b) for b in bs]
        specs = []
        for w, b, dw, db in zip(ws, bs, dws, dbs):
            sp = ops.ConvSpec(n_const, nc, 3, 1, 1, 1, 1, False, w_cin=nc, w_cin_off=0)
            sp.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), dw.data_ptr(), db.data_ptr(), True, None, 0)
            specs.append(sp)
        e = emb.clone().requires_grad_(True)
        k3s = ops.film_k3_multi(e, specs) if multi else [ops.film_k3(e, sp) for sp in specs]
        sum((k * c).sum() for k, c in zip(k3s, cots)).backward()
        torch.cuda.synchronize()
        return [k.detach() for k in k3s], e.grad, dws, dbs
    k_m, de_m, dw_m, db_m = run(True)
    k_s, de_s, dw_s, db_s = run(False)
    for a, b_ in zip(k_m, k_s):
        assert torch.equal(a, b_)
    assert rel_l2(de_m, de_s) < 1e-6
    for a, b_ in zip(dw_m + db_m, dw_s + db_s):
        assert torch.equal(a, b_)
    # and the forward against plain torch: cond_var.0 on the constant embedding channels as a length-3 signal
    x3 = emb[:, :, None].expand(B, n_const, 3).double().cpu()
    ref = torch.nn.functional.conv1d(x3, ws[0][:, :n_const].double().cpu(), bs[0].double().cpu(), padding=1)
    assert rel_l2(k_m[0], ref) < 1e-6


@pytest.mark.parametrize('cfg', [(3, 1, 1040, True, True), (7, 3, 16000, True, False), (11, 5, 2052, False, True), (3, 5, 516, True, False)],
                         ids=['k3d1_T1040', 'k7d3_T16000', 'k11d5_T2052_nofilm', 'k3d5_T516'])
def test_fused_film_block_forward_is_bit_identical(cfg, dev):
    """tdvc_film_block_fwd (dilated conv + FiLM + 1x1 conv + residual in one launch, 16 channels) against the two-launch path:
    h and out must be bit-identical (same arithmetic in the same order), over ragged time tiles, with / without FiLM and MRF sum."""
    ops, L, arena = _mods()
    k, d, T, film, with_acc = cfg
    torch.manual_seed(k * 100 + d)
    B, Cc = 3, 16
    w1 = (torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5).to(dev); b1 = (torch.randn(Cc) * 0.1).to(dev)
    w2 = (torch.randn(Cc, Cc, 1) / Cc ** 0.5).to(dev); b2 = (torch.randn(Cc) * 0.1).to(dev)
    cs = ops.ConvSpec(Cc, Cc, k, 1, (k - 1) * d // 2, d, 1, True)
    cs.slot = arena.ConvSlot(w1.data_ptr(), b1.data_ptr(), 0, 0, False, None, 0)
    ps = ops.ConvSpec(Cc, Cc, 1)
    ps.slot = arena.ConvSlot(w2.data_ptr(), b2.data_ptr(), 0, 0, False, None, 0)
    x0 = torch.randn(B, Cc, T, device=dev)
    gb0 = torch.randn(B, 2 * Cc, T, device=dev) * 0.5 if film else None
    acc = torch.randn(B, Cc, T, device=dev) if with_acc else None
    cot = torch.randn(B, Cc, T, device=dev)
    outs = {}
    for fused in (True, False):
        old = ops.FUSED_FILM_BLOCK
        ops.FUSED_FILM_BLOCK = fused
        x = x0.clone().requires_grad_(True)
        gb = gb0.clone().requires_grad_(True) if film else None
        try:
            with traced() as tr:
                y = ops.FilmBlockFn.apply(x, gb, acc, None, cs, ps, 1.0 / 3)
                y.backward(cot)      # reads the stored h: a wrong h shows up in both gradients
        finally:
            ops.FUSED_FILM_BLOCK = old
        assert any(n.startswith('film_block_fwd_kernel') for n in tr.names) == fused, sorted(tr.names)
        outs[fused] = (y.detach().clone(), x.grad.clone(), gb.grad.clone() if film else None)
    for a_, b_ in zip(outs[True], outs[False]):
        if a_ is not None:
            assert torch.equal(a_, b_), float((a_ - b_).abs().max())
