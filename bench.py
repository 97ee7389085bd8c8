"""Headline benchmark: audio-seconds/sec of one full TD-VC-GAN training iteration (D-step + G-step,
both AdamW updates) on the HIP path, config/conv_enc-stage1.yaml, 16 x 1 s of 16 kHz audio per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement). `roofline` is measured live with HIP
events on the launch stream for the kernel named in it; `cpu_baseline` times the CPU oracle (a port of
the reference's algorithm, pinned against the reference: oracle/) on this host's cores on a bounded sample.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

SR = 16000
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3    # fp32-input MFMA = fp32 vector peak
PMC_TRAFFIC_BYTES = None     # filled from profiles/r01_pmc_traffic.txt below ((2*FETCH_SIZE + WRITE_SIZE) KiB per launch)


def time_conv_kernel(pkg, dev, B, cin, cout, k, dil, T, iters=50):
    """Average device time of ONE forward launch of the stride-1 dilated Conv1d kernel (reflect pad,
    fused LeakyReLU-on-load + bias), HIP events on the launch stream."""
    ops, arena, L = pkg.ops, pkg.arena, pkg._lib
    pad = (k - 1) * dil // 2
    spec = ops.ConvSpec(cin, cout, k, 1, pad, dil, 1, True)
    w = torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), 0, 0, False, None)
    x = torch.randn(B, cin, T, device=dev)
    y = torch.empty(B, cout, T, device=dev)
    xf = ops._xf(L.XF_LRELU)
    for _ in range(5):
        ops.conv_fwd_raw(spec, x, xf, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        ops.conv_fwd_raw(spec, x, xf, out=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    alg_bytes = 4.0 * B * (cin * T + cout * T) + 4.0 * (cout * cin * k + cout)
    flops = 2.0 * B * T * cout * cin * k
    return ms, alg_bytes, flops


def pmc_traffic_bytes():
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE,
    gfx950 correction per MI355X_MICROARCH.md); None when the profile file is absent."""
    try:
        for line in open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.txt')):
            if line.startswith('ROOFLINE_KERNEL_HBM_BYTES_PER_LAUNCH'):
                return float(line.split('=')[1])
    except OSError:
        pass
    return None


def usable_cores():
    """CPUs this process may actually run on (the GPU box exposes more logical CPUs than its cgroup share)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 quota
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(pkg, cfg_train, iters=2):
    """CPU oracle timed on this host: B=2 x 1 s, 1 warm-up + `iters` timed iterations (~10-30 s)."""
    from common import filled_sd
    from oracle import step as OS
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = OS.StepConfig.from_hparams(cfg_train)
    st = OS.TrainStep(filled_sd('G'), filled_sd('D'), cfg)
    B, T = 2, SR
    bt = pkg.synth.make_batch(B, T, seed=1234, conversion=not cfg.no_conv)
    ix, iy = pkg.synth.contrastive_indices(B, T // 320, cfg.n_neg, 1), pkg.synth.contrastive_indices(B, T // 320, cfg.n_neg, 2)
    st.run(bt, ix, iy)
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter(); st.run(bt, ix, iy); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    return dict(value=B * T / SR / t, unit='audio-seconds/sec', cores=cores, kind='port',
                sample=f'conv_enc-stage1 full D+G iteration, B={B} x 1 s, median of {iters} after 1 warm-up; '
                       f'CPU oracle (torch {torch.__version__} CPU, {torch.get_num_threads()} threads)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16, help='samples per GPU (1 s each)')
    ap.add_argument('--config', default='conv_enc-stage1')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='run eagerly instead of replaying a captured hipGraph')
    ap.add_argument('--force-dp', action='store_true', help='exercise the data-parallel code path (RCCL group, segmented graphs) even with one rank')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    dev = torch.device(f'cuda:{local}')
    torch.cuda.set_device(dev)
    dp = world > 1 or args.force_dp
    if dp:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
            dist.init_process_group('nccl', device_id=dev, rank=0, world_size=1)
        else:
            dist.init_process_group('nccl', device_id=dev)

    pkg = importlib.import_module('td-vc-gan_amd')
    from common import build_models, to_dev
    hp = pkg.hparams.HParam(os.path.join(ROOT, 'config', f'{args.config}.yaml'))
    cfg = pkg.train_step.StepConfig.from_hparams(hp.train)
    G, D = build_models(dev)
    sync = pkg.parallel.GradSync() if dp else None
    if sync is not None:
        sync.broadcast_params(G.arena); sync.broadcast_params(D.arena)
    ts = pkg.train_step.TrainStep(G, D, cfg, dev, grad_sync=sync)
    B, T = args.batch, SR
    bt = to_dev(pkg.synth.make_batch(B, T, seed=1234 + rank, conversion=not cfg.no_conv), dev)
    ix = pkg.synth.contrastive_indices(B, T // 320, cfg.n_neg, 17 + rank).to(dev)
    iy = pkg.synth.contrastive_indices(B, T // 320, cfg.n_neg, 917 + rank).to(dev)

    use_graph = (sync is None) and not args.no_graph      # data-parallel steps run eagerly (see TrainStep.capture)
    launch = 'eager'
    step = None
    if use_graph:
        try:
            step = ts.capture(bt, ix, iy, warmup=2)      # whole iteration as one hipGraph
            launch = 'hipGraph replay'
        except Exception as e:      # noqa: BLE001 -- fall back to eager launches rather than lose the measurement
            print(f'[bench] rank {rank}: graph capture failed ({type(e).__name__}: {e}); running eagerly', file=sys.stderr, flush=True)
            step = None
    if step is None:
        def step():
            return ts.run(bt, ix, iy)

    for _ in range(args.warmup):
        log = step()
    torch.cuda.synchronize()
    if rank == 0:
        print(f'[bench] warm-up done ({args.warmup} steps)', file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        log = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    g_loss = float(log['G_loss'])
    if rank == 0:
        print(f'[bench] timed region: {dt / args.steps * 1e3:.2f} ms/step', file=sys.stderr, flush=True)
    if not (g_loss == g_loss):
        raise SystemExit('non-finite loss in the timed region')

    if rank == 0:
        value = world * B * (T / SR) * args.steps / dt
        # roofline of the dilated Conv1d kernel the north star names (16->16, k=3, T=16000: the HBM-bound end).
        # traffic: HBM bytes per launch from the committed rocprofv3 PMC pass (profiles/, FETCH_SIZE doubled per
        # MI355X_MICROARCH.md) — not re-measured live (needs the profiler).
        # Launch shape = the step's: every trunk conv runs on 2*B samples (decoder: [target; identity] conditionings,
        # encoder: [real; corrupted], discriminator: [real; fake]), one launch per layer.
        BL = 2 * B
        ms, alg_bytes, flops = time_conv_kernel(pkg, dev, BL, 16, 16, 3, 1, T)
        roof = dict(kernel=f'conv_lean_kernel<1,4,1,4,ACT,FWD> dilated Conv1d 16->16 k3 d1 T=16000 B={BL} fwd (fused LeakyReLU+bias)',
                    bound='hbm', achieved=alg_bytes / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=pmc_traffic_bytes(), ms_per_launch=ms,
                    algorithmic_bytes=alg_bytes)
        # the FLOP-dominant kernel class of the step: FiLM conditioning conv 136 -> 2C, k3 (here C=64, T=4000); the fused
        # conditioning forward and the cond_var.2 input-grad run the same tile / MFMA loop
        ms2, _, fl2 = time_conv_kernel(pkg, dev, BL, 136, 128, 3, 1, 4000)
        roof['mfma_kernel'] = dict(kernel=f'conv_lean_kernel<4,4,1,4,ACT,FWD> FiLM cond_var.2 136->128 k3 T=4000 B={BL} fwd', bound='mfma',
                                   achieved=fl2 / (ms2 * 1e-3) / 1e12, peak=MFMA_F32_PEAK_TF, unit='TFLOP/s',
                                   frac=fl2 / (ms2 * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, ms_per_launch=ms2)
        stage = 'stage-1' if 'stage1' in args.config else args.config
        out = dict(metric=f'audio-seconds/sec (G+D train step, {stage})', value=value, unit='audio-seconds/sec', n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3, higher_is_better=True,
                   scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload=f'config/{args.config}.yaml full D+G iteration, {B} x 1 s @16 kHz per GPU, NUM_SPK=16, '
                                        'F0 (CREPE) loss term excluded', global_batch=world * B, parallelism=f'dp{world}'),
                   roofline=roof, final_G_loss=g_loss, launch=launch)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(pkg, hp.train)
            out['speedup_vs_cpu'] = value / out['cpu_baseline']['value']
        print(json.dumps(out))
    if dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
