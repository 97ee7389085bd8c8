"""Lean kernel vs generic kernel on single conv launches (debug aid): python tools/lean_vs_generic.py"""
import importlib, os, sys, itertools
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('td-vc-gan_amd')
ops, arena, L = pkg.ops, pkg.arena, pkg._lib
dev = torch.device('cuda:0')
lib = L.lib()
torch.manual_seed(0)


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(cin, cout, k, d, T, reflect, pre, B):
    pad = (k - 1) * d // 2
    spec = ops.ConvSpec(cin, cout, k, 1, pad, d, 1, reflect)
    w = torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    wt = w.permute(1, 0, 2).contiguous()
    spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), dw.data_ptr(), db.data_ptr(), True, None, wt.data_ptr())
    x = torch.randn(B, cin, T, device=dev)
    dy = torch.randn(B, cout, T, device=dev)
    xf = ops._xf(L.XF_LRELU if pre else L.XF_NONE)
    out = {}
    for g in (0, 1):
        lib.tdvc_set_force_generic(g)
        y = ops.conv_fwd_raw(spec, x, xf)
        dx = ops.conv_dgrad_raw(spec, dy, ops._xf(), T, L.DG_MASK_LRELU if pre else L.DG_PLAIN, x_in=x if pre else None)
        torch.cuda.synchronize()
        out[g] = (y.clone(), dx.clone())
    lib.tdvc_set_force_generic(0)
    ef, ed = rel(out[0][0], out[1][0]), rel(out[0][1], out[1][1])
    flag = '  <-- BAD' if max(ef, ed) > 1e-5 else ''
    print(f'cin={cin:4d} cout={cout:4d} k={k:2d} d={d} T={T:5d} refl={int(reflect)} pre={pre} B={B:2d}  fwd {ef:.2e}  dgrad {ed:.2e}{flag}', flush=True)
    if flag:
        dfy = (out[0][0] - out[1][0]).abs().amax(dim=(0, 1))
        bad = torch.nonzero(dfy > 1e-4 * out[1][0].abs().max()).flatten().tolist()
        print('     fwd bad t:', bad[:12], '...', bad[-4:], 'n=', len(bad))
        dfx = (out[0][1] - out[1][1]).abs().amax(dim=(0, 1))
        bad = torch.nonzero(dfx > 1e-4 * out[1][1].abs().max()).flatten().tolist()
        print('     dgrad bad t:', bad[:12], '...', bad[-4:], 'n=', len(bad))
        dfc = (out[0][0] - out[1][0]).abs().amax(dim=(0, 2))
        print('     fwd bad ch:', torch.nonzero(dfc > 1e-4 * out[1][0].abs().max()).flatten().tolist()[:20])


def run_film(C, k, d, T, B, with_gb=True):
    pad = (k - 1) * d // 2
    cs = ops.ConvSpec(C, C, k, 1, pad, d, 1, True)
    ps = ops.ConvSpec(C, C, 1, 1, 0, 1, 1, False)
    keep = []
    for sp, kk in ((cs, k), (ps, 1)):
        w = torch.randn(C, C, kk, device=dev) / (C * kk) ** 0.5
        b = torch.randn(C, device=dev) * 0.1
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        wt = w.permute(1, 0, 2).contiguous()
        keep += [w, b, dw, db, wt]
        sp.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), dw.data_ptr(), db.data_ptr(), True, None, wt.data_ptr())
    x0 = torch.randn(B, C, T, device=dev)
    gb0 = torch.randn(B, 2 * C, T, device=dev) * 0.5
    acc0 = torch.randn(B, C, T, device=dev)
    do = torch.randn(B, C, T, device=dev)
    out = {}
    for g in (0, 1):
        lib.tdvc_set_force_generic(g)
        x = x0.clone().requires_grad_(True); gb = gb0.clone().requires_grad_(True); acc = acc0.clone().requires_grad_(True)
        y = ops.film_block(x, gb if with_gb else None, acc, cs, ps, 1.0 / 3)
        y.backward(do)
        torch.cuda.synchronize()
        out[g] = (y.detach().clone(), x.grad.clone(), gb.grad.clone() if with_gb else torch.zeros(1, device=dev))
    lib.tdvc_set_force_generic(0)
    e = [rel(a, b) for a, b in zip(out[0], out[1])]
    flag = '  <-- BAD' if max(e) > 1e-5 else ''
    print(f'film C={C:4d} k={k:2d} d={d} T={T:5d} B={B:2d} gb={int(with_gb)}  y {e[0]:.2e}  dx {e[1]:.2e}  dgb {e[2]:.2e}{flag}', flush=True)
    if flag:
        for nm, a, b in zip(('y', 'dx', 'dgb'), out[0], out[1]):
            df = (a - b).abs().amax(dim=(0, 1)) if a.dim() == 3 else None
            if df is None: continue
            bad = torch.nonzero(df > 1e-4 * b.abs().max()).flatten().tolist()
            dc = (a - b).abs().amax(dim=(0, 2))
            print('    ', nm, 'bad t:', bad[:10], '...', bad[-4:], 'n=', len(bad), ' bad ch:', torch.nonzero(dc > 1e-4 * b.abs().max()).flatten().tolist()[:24])


for B in (2,):
    for (c, T) in ((64, 2240), (32, 4480), (16, 8960), (128, 280)):
        for k in (3, 7, 11):
            for d in (1, 3, 5):
                run(c, c, k, d, T, True, 1, B)
sys.exit(0)
for B in (2, 32):
    for (c, T) in ((64, 2240), (32, 4480), (16, 8960), (128, 280)):
        for k, d in ((3, 1), (7, 3), (11, 5), (1, 1)):
            run(c, c, k, d, T, k > 1, 1, B)
        run(136, 2 * c, 3, 1, T, False, 1, B)
        run(8, 136, 3, 1, T, False, 0, B)
