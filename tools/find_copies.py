"""Diagnostic: where do device-to-device copies / ATen kernels come from in one train step?"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('td-vc-gan_amd')
from common import build_models, to_dev
dev = torch.device('cuda:0')
G, D = build_models(dev)
ts = pkg.train_step.TrainStep(G, D, pkg.train_step.StepConfig(), dev)
B, T = 4, 16000
bt = to_dev(pkg.synth.make_batch(B, T, seed=1), dev)
ix = pkg.synth.contrastive_indices(B, T // 320, 100, 1).to(dev); iy = pkg.synth.contrastive_indices(B, T // 320, 100, 2).to(dev)
ts.run(bt, ix, iy); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    ts.run(bt, ix, iy); torch.cuda.synchronize()
ev = [e for e in prof.events() if e.name in ('aten::copy_', 'aten::add', 'aten::add_', 'aten::mul', 'aten::clone', 'aten::contiguous', 'aten::zero_', 'aten::fill_', 'aten::sum', 'aten::zeros', 'aten::zeros_like')]
import collections
c = collections.Counter()
for e in ev:
    st = [s for s in (e.stack or []) if 'td-vc-gan_amd' in s or 'tools/' in s or 'autograd' in s]
    c[(e.name, str(e.input_shapes)[:80], st[0][-70:] if st else '?')] += 1
for k, v in c.most_common(40):
    print(v, k)
