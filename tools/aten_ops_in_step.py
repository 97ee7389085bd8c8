"""Which ATen kernels does one eager training iteration launch, and from which Python lines? (diagnostic: torch.profiler with stacks)"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import build_models, to_dev
pkg = importlib.import_module('td-vc-gan_amd')
dev = torch.device('cuda:0')
G, D = build_models(dev)
cfg = pkg.train_step.StepConfig()
ts = pkg.train_step.TrainStep(G, D, cfg, dev)
bt = to_dev(pkg.synth.make_batch(16, 16000, seed=1), dev)
ix = pkg.synth.contrastive_indices(16, 50, cfg.n_neg, 1).to(dev); iy = pkg.synth.contrastive_indices(16, 50, cfg.n_neg, 2).to(dev)
for _ in range(2): ts.run(bt, ix, iy)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    ts.run(bt, ix, iy); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6) if e.key.startswith('aten::') and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    st = [s for s in e.stack if 'td-vc-gan_amd' in s or 'tests/' in s][:3]
    print(f'{e.device_time_total:9.1f} us n={e.count:3d} {e.key:28s} {str(e.input_shapes)[:70]:70s} {st}')
