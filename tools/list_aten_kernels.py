"""Diagnostic: the ATen (non-tdvc) kernels of one eager training iteration, with input shapes and the autograd node that
issued them (torch.profiler). Shows where autograd's own gradient accumulation / glue still costs launches.

    python tools/list_aten_kernels.py
"""
import collections
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from common import build_models, to_dev  # noqa: E402

pkg = importlib.import_module('td-vc-gan_amd')
dev = torch.device('cuda:0')
G, D = build_models(dev)
ts = pkg.train_step.TrainStep(G, D, pkg.train_step.StepConfig(), dev)
B, T = 16, 16000
bt = to_dev(pkg.synth.make_batch(B, T, seed=1), dev)
ix = pkg.synth.contrastive_indices(B, T // 320, 100, 1).to(dev); iy = pkg.synth.contrastive_indices(B, T // 320, 100, 2).to(dev)
for _ in range(2):
    ts.run(bt, ix, iy)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    ts.run(bt, ix, iy)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, 'self_device_time_total', None)
    if t is None:
        t = getattr(e, 'self_cuda_time_total', 0.0)
    if e.key.startswith('aten::') and t > 0:
        rows.append((t, e.count, e.key, str(e.input_shapes)[:100]))
tot = 0.0
for t, n, name, shp in sorted(rows, reverse=True)[:45]:
    print(f'{t / 1e3:7.3f} ms  n={n:4d}  {name:28s} {shp}')
    tot += t
print(f'total self device time of the listed ATen ops: {tot / 1e3:.2f} ms')
