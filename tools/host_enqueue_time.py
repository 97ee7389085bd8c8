"""Host time to ENQUEUE one eager training iteration (no synchronisation inside): the slack the data-parallel (eager) path has before it
becomes launch-bound. Diagnostic."""
import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import build_models, to_dev
pkg = importlib.import_module('td-vc-gan_amd')
dev = torch.device('cuda:0')
G, D = build_models(dev)
cfg = pkg.train_step.StepConfig()
ts = pkg.train_step.TrainStep(G, D, cfg, dev)
bt = to_dev(pkg.synth.make_batch(16, 16000, seed=1), dev)
ix = pkg.synth.contrastive_indices(16, 50, cfg.n_neg, 1).to(dev); iy = pkg.synth.contrastive_indices(16, 50, cfg.n_neg, 2).to(dev)
for _ in range(3): ts.run(bt, ix, iy)
torch.cuda.synchronize()
host, total = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts.run(bt, ix, iy)                      # returns when everything is enqueued
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print('host enqueue ms per iteration:', [round(h, 1) for h in host])
print('enqueue + drain ms:           ', [round(t, 1) for t in total])
print(f'cores available to this process: {len(os.sched_getaffinity(0))}')
