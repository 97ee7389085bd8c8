"""Dump generator outputs + parameter gradients of the golden B2_T8960 case (debug aid for A/B library builds):
   TDVC_HIP_LIB=.../libtdvc_hip_old.so python tools/grad_dump.py out_old.pt ; python tools/grad_dump.py out_new.pt ; python tools/grad_dump.py --cmp out_old.pt out_new.pt"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import GOLDEN, build_models, pkg, to_dev

if sys.argv[1] == '--cmp':
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    gg = json.load(open(os.path.join(GOLDEN, 'gen_grad_B2_T8960.json')))
    for k in a:
        if k.startswith('hook.0120') and k.endswith('.dx'):
            d = (a[k].double() - b[k].double()).abs()
            print('DETAIL', k, 'max abs', float(d.max()), 'ref max', float(a[k].abs().max()))
            flat = d.flatten().topk(12)
            for v, i in zip(flat.values.tolist(), flat.indices.tolist()):
                bb, rem = divmod(i, d.shape[1] * d.shape[2]); c, t = divmod(rem, d.shape[2])
                print(f'    b={bb} c={c} t={t} diff={v:.3e} old={float(a[k][bb, c, t]):.4e} new={float(b[k][bb, c, t]):.4e}')
            dt = d.amax(dim=(0, 1)); bad = torch.nonzero(dt > 1e-3 * float(a[k].abs().max())).flatten().tolist()
            print('    bad t count', len(bad), bad[:40])
            for kk in a:
                if kk.startswith('fwd.hook.0119'):
                    print('    x old/new at flip:', [(float(a[kk][bb_, c_, t_]), float(b[kk][bb_, c_, t_])) for (bb_, c_, t_) in [(0, 4, 3106)]], 'x rms', float(a[kk].pow(2).mean().sqrt()), 'fwd rel diff', float((a[kk] - b[kk]).norm() / a[kk].norm()))
            y = a['out.sub2']; print('    sub2 |y|>0.999 frac', float((y.abs() > 0.999).float().mean()), ' |y| mean', float(y.abs().mean()))
    rows = []
    for k in a:
        d = float((a[k].double() - b[k].double()).norm() / (a[k].double().norm() + 1e-30))
        na, nb = float(a[k].double().norm()), float(b[k].double().norm())
        ref = gg['norms'].get(k, None)
        rows.append((d, k, na, nb, ref))
    for r in sorted(rows, key=lambda r: r[1]):
        if r[1].startswith('out.') or r[1].startswith('hook.'): print(f'{r[0]:.3e}  {r[1]}')
    import collections
    grp = collections.defaultdict(float)
    for r in rows:
        if r[1].startswith('hook.'): continue
        key = '.'.join(r[1].split('.')[:3])
        grp[key] = max(grp[key], r[0])
    for k, v in sorted(grp.items(), key=lambda kv: -kv[1]): print(f'   group {k:40s} max diff {v:.3e}')
    rows.sort(reverse=True)
    for d, k, na, nb, ref in rows[:10]:
        extra = '' if ref is None else f'  e_norm old {abs(na - ref) / ref:.2e} new {abs(nb - ref) / ref:.2e}'
        print(f'{d:.3e}  {k}{extra}')
    sys.exit(0)

dev = torch.device('cuda:0')
G, _ = build_models(dev)
if os.environ.get('TDVC_FORCE_GENERIC'):
    pkg()._lib.lib().tdvc_set_force_generic(1)
ops = pkg().ops
res = {}
counter = [0]
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **kw):
        idx = counter[0]; counter[0] += 1
        if isinstance(a[0], torch.Tensor) and a[0].requires_grad:
            a = (a[0].view_as(a[0]),) + tuple(a[1:])      # own autograd edge -> per-consumer input gradient
        x = a[0]
        out = orig(*a, **kw)
        tag = f'hook.{idx:04d}.{name}.{tuple(out.shape)}'
        if idx in (118, 119): res['fwd.' + tag] = out.detach().cpu()
        if out.requires_grad:
            out.register_hook(lambda g, t=tag: res.__setitem__(t + '.dout', g.detach().cpu()))
        if isinstance(x, torch.Tensor) and x.requires_grad:
            x.register_hook(lambda g, t=tag: res.__setitem__(t + '.dx', g.detach().cpu()))
        return out
    setattr(ops, name, f)
for nm in ('conv', 'film_block', 'film_cond'):
    wrap(nm)
bt = to_dev(pkg().synth.make_batch(2, 8960, seed=7), dev)
G.arena.zero_grad()
y, subs = G(bt['signal_real'], bt['c_tgt'], c_var=bt['c_f0_conv'], out_subsample=True)
emb = G.content_embedding
rs = np.random.RandomState(99)
outs = (y, subs[0], subs[1], emb)
cot = [torch.from_numpy(rs.randn(*t.shape).astype(np.float32)).to(dev) for t in outs]
loss = sum((t * c).mean() for t, c in zip(outs, cot))
loss.backward()
torch.cuda.synchronize()
res.update({'out.y': y.detach().cpu(), 'out.sub4': subs[0].detach().cpu(), 'out.sub2': subs[1].detach().cpu(), 'out.emb': emb.detach().cpu()})
for k, p in G.named_parameters():
    if p.grad is not None:
        res[k] = p.grad.detach().cpu()
torch.save(res, sys.argv[1])
print('saved', sys.argv[1], float(loss))
