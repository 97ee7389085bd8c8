"""Headline benchmark: audio-seconds/sec of one full TD-VC-GAN training iteration (D-step + G-step,
both AdamW updates) on the HIP path, config/conv_enc-stage1.yaml, 16 x 1 s of 16 kHz audio per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 ...          # spawns its own ranks (torch.distributed.run, one fresh process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement). `roofline` is measured live with HIP events on the
launch stream: every kernel class of the table below is launched at the step's own launch shape on ROTATING buffer sets
(> 512 MB per rotation, so nothing is served from the 256 MB Infinity Cache), its instance name is read back from the
library (tdvc_debug_trace), and the entries are ranked by their share of the measured step time — the dominant kernel
first. `cpu_baseline` times the CPU oracle (a port of the reference's algorithm, pinned against the reference: oracle/)
on this host's cores on a bounded sample.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
SR = 16000
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3    # fp32-input MFMA = fp32 vector peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16, help='samples per GPU (1 s each)')
    ap.add_argument('--seconds', type=float, default=1.0, help='length of each sample in seconds (BASELINE config #5: 2)')
    ap.add_argument('--config', default='conv_enc-stage1', help='conv_enc-stage1 | conv_enc-stage2_1 | conv_enc-stage2_2 | wavlm-stage2_2 (frozen SSL '
                                                                "extractor = synth.FrameFeatureExtractor, the stand-in for WavLM-Large whose checkpoint is absent)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-table', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='run eagerly instead of replaying a captured hipGraph')
    ap.add_argument('--force-dp', action='store_true', help='exercise the data-parallel code path (RCCL group, eager launches, '
                                                            'segment-pipelined gradient all-reduce) even with one rank')
    return ap.parse_args()


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes through
    torch.distributed.run and exit with their code. Decided before this process imports torch or touches a GPU."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ or 'RANK' in os.environ:
        return
    port = 29400 + os.getpid() % 2000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


# ----------------------------------------------------------------------------------------------- kernel table
class Bufs:
    """Rotating operand sets for one micro-benchmark: `n` independent copies of every tensor, sized so that one
    rotation moves > 512 MB — a buffer is long gone from the 256 MB Infinity Cache when its turn comes again."""

    def __init__(self, torch, dev, shapes, min_bytes=600e6, max_sets=16):
        per = sum(4 * int(torch.Size(s).numel()) for s in shapes.values())
        self.n = int(max(2, min(max_sets, -(-min_bytes // per))))
        self.sets = [{k: torch.randn(s, device=dev) * 0.5 for k, s in shapes.items()} for _ in range(self.n)]
        self.bytes_per_rotation = per * self.n


def time_launches(torch, calls, iters):
    """Average device time of one launch over `iters` launches rotating through `calls`, HIP events on the launch stream."""
    n = len(calls)
    for i in range(n + 2):
        calls[i % n]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        calls[i % n]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_table(pkg, dev, B, step_ms, iters=40, manifest=None):
    """Per-kernel roofline table at the step's launch shapes (B samples per GPU -> every trunk conv sees BL = 2B:
    decoder [target; identity] conditionings, encoder [real; corrupted], discriminator [real; fake] / [fake; identity];
    discs 1 and 2 see another x2 from the batched sub-scale pass). `n` = launches of that (op, shape) per iteration of
    config/conv_enc-stage1.yaml: 9 FiLM blocks per decoder stage, one generator pass and three discriminator calls."""
    import torch
    ops, arena, L = pkg.ops, pkg.arena, pkg._lib
    lib = L.lib()
    BL = 2 * B
    rows = []

    def traced(call):
        lib.tdvc_debug_trace(1)
        call()
        torch.cuda.synchronize()
        names = sorted(L.traced_kernels())
        lib.tdvc_debug_trace(0)
        return ' + '.join(names)

    def add(label, n, bound, alg_bytes, flops, calls, rot_bytes):
        name = traced(calls[0])
        ms = time_launches(torch, calls, iters)
        if manifest is not None:      # tools/microbench_kernels.py: lets the PMC summary map dispatches back to this entry
            manifest.append(dict(op=label, kernels=name.split(' + '), calls=1 + len(calls) + 2 + iters))
        ach_b, ach_f = alg_bytes / (ms * 1e-3) / 1e9, flops / (ms * 1e-3) / 1e12
        e = dict(kernel=name, op=label, launches_per_step=n, ms_per_launch=ms, share_of_step=n * ms / step_ms, bound=bound,
                 achieved=ach_b if bound == 'hbm' else ach_f, peak=HBM_PEAK_GBS if bound == 'hbm' else MFMA_F32_PEAK_TF,
                 unit='GB/s' if bound == 'hbm' else 'TFLOP/s', algorithmic_bytes=alg_bytes, algorithmic_flops=flops,
                 rotation_bytes=rot_bytes)
        e['frac'] = e['achieved'] / e['peak']
        rows.append(e)

    def conv_case(label, n, bound, cin, cout, k, dil, T, reflect, pre, which, Bc, post=0, film=False, bias3=False, bits=False):
        pad = (k - 1) * dil // 2
        spec = ops.ConvSpec(cin, cout, k, 1, pad, dil, 1, reflect)
        w = torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5
        b = torch.randn(cout, device=dev) * 0.1
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        wt = w.permute(1, 0, 2).contiguous()
        spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), dw.data_ptr(), db.data_ptr(), True, None, wt.data_ptr())
        k3b = torch.randn(Bc, cout, 3, device=dev) * 0.1 if bias3 else None
        keep.extend([w, b, dw, db, wt, k3b])
        # operands of the launch exactly as the step passes them (distinct tensors for input / mask source / output)
        if which == 'fwd':
            shapes = dict(x=(Bc, cin, T), y=(Bc, cout, T))
            if film:
                shapes.update(gb=(Bc, 2 * cin, T), res=(Bc, cout, T))
            words = cin + cout + ((2 * cin + cout) if film else 0) + (cout / 32.0 if bits else 0)
        elif which == 'dgrad':
            shapes = dict(dy=(Bc, cout, T), dx=(Bc, cin, T))
            if pre and not bits:
                shapes['x_in'] = (Bc, cin, T)
            if post:
                shapes['act'] = (Bc, cout, T)
            # bits: the LeakyReLU mask comes as 1 bit per element (tdvc_conv_dgrad_args.x_sign_bits), like the step passes it
            words = cout + cin + ((cin / 32.0 if bits else cin) if pre else 0) + (cout if post else 0)
        else:
            shapes = dict(x=(Bc, cin, T), dy=(Bc, cout, T))
            if post:
                shapes['act'] = (Bc, cout, T)
            words = cin + cout + (cout if post else 0)
        bufs = Bufs(torch, dev, shapes)
        keep.append(bufs)
        calls = []
        nbw = (cin if which == 'dgrad' else cout)
        bitbufs = [torch.randint(-2 ** 31, 2 ** 31 - 1, (Bc, nbw, T // 32), dtype=torch.int32, device=dev) for _ in bufs.sets] if bits else [None] * bufs.n
        keep.append(bitbufs)
        for s, bw in zip(bufs.sets, bitbufs):
            dyxf = (lambda s=s: ops._xf(L.XF_MASK_LRELU, aux=s['act'])) if post else (lambda s=s: ops._xf())
            if which == 'fwd':
                xf = ops._xf(L.XF_FILM_LRELU, aux=s['gb']) if film else ops._xf(L.XF_LRELU if pre else L.XF_NONE)
                calls.append(lambda s=s, xf=xf, bw=bw: ops.conv_fwd_raw(spec, s['x'], xf, post=post, res=s.get('res'), out=s['y'], bias3=k3b, sign_bits=bw))
            elif which == 'dgrad':
                calls.append(lambda s=s, f=dyxf, bw=bw: ops.conv_dgrad_raw(spec, s['dy'], f(), T, L.DG_MASK_LRELU if pre else L.DG_PLAIN,
                                                                           x_in=s.get('x_in'), out=s['dx'], x_bits=bw))
            else:
                xf = ops._xf(L.XF_LRELU if pre else L.XF_NONE)
                calls.append(lambda s=s, xf=xf, f=dyxf: ops.conv_wgrad_raw(spec, s['x'], xf, s['dy'], f()))
        alg = 4.0 * Bc * T * words + 4.0 * (w.numel() + cout)
        add(label, n, bound, alg, 2.0 * Bc * T * cin * cout * k, calls, bufs.bytes_per_rotation)

    def cond_fwd_case(label, n, C2, T, Bc):
        nc, nv = 136, 8
        w0 = torch.randn(nc, nc, 3, device=dev) / (nc * 3) ** 0.5
        w2 = torch.randn(C2, nc, 3, device=dev) / (nc * 3) ** 0.5
        b2 = torch.randn(C2, device=dev) * 0.1
        keep.extend([w0, w2, b2])
        bufs = Bufs(torch, dev, dict(exc=(Bc, nv, T), k3=(Bc, nc, 3), cv0=(Bc, nc, T), gb=(Bc, C2, T)))
        keep.append(bufs)
        st = torch.cuda.current_stream(dev).cuda_stream
        calls = []
        for s in bufs.sets:
            a = L.FilmCondArgs(Bc, T, nc, nv, C2, s['exc'].data_ptr(), s['exc'].stride(0), w0.data_ptr(), s['k3'].data_ptr(), w2.data_ptr(),
                               b2.data_ptr(), s['cv0'].data_ptr(), s['cv0'].stride(0), s['gb'].data_ptr(), s['gb'].stride(0), 0.2)
            keep.append(a)
            calls.append(lambda a=a: L.check(lib.tdvc_film_cond_fwd(C.byref(a), st)))
        alg = 4.0 * Bc * T * (nv + nc + C2) + 4.0 * (w2.numel() + nc * nv * 3)
        add(label, n, 'mfma', alg, 2.0 * Bc * T * (C2 * nc * 3 + nc * nv * 3), calls, bufs.bytes_per_rotation)

    def cond0_bwd_case(label, n, T, Bc):
        nc, nv = 136, 8
        w0 = torch.randn(nc, nc, 3, device=dev) / (nc * 3) ** 0.5
        dw0 = torch.zeros_like(w0)
        keep.extend([w0, dw0])
        bufs = Bufs(torch, dev, dict(dcv=(Bc, nc, T), exc=(Bc, nv, T), dexc=(Bc, nv, T), dk3=(Bc, nc, 3)))
        keep.append(bufs)
        nbytes = lib.tdvc_film_cond0_bwd_workspace(Bc, T, nc, nv)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        keep.append(ws)
        st = torch.cuda.current_stream(dev).cuda_stream
        calls = []
        for s in bufs.sets:
            a = L.FilmCond0BwdArgs(Bc, T, nc, nv, s['dcv'].data_ptr(), s['dcv'].stride(0), s['exc'].data_ptr(), s['exc'].stride(0), w0.data_ptr(),
                                   s['dexc'].data_ptr(), s['dexc'].stride(0), s['dk3'].data_ptr(), dw0.data_ptr(), ws.data_ptr(),
                                   ws.numel() * ws.element_size())
            keep.append(a)
            calls.append(lambda a=a: (L.check(lib.tdvc_film_cond0_bwd(C.byref(a), st)), L.check(lib.tdvc_fold_flush(st))))
        add(label, n, 'hbm', 4.0 * Bc * T * (nc + 2 * nv), 2.0 * Bc * T * nc * nv * 3 * 2, calls, bufs.bytes_per_rotation)

    keep = []
    stages = [(32, 16000), (64, 8000), (128, 4000), (256, 500)]          # (2C, T) of the four decoder stages
    for C2, T in stages:
        tag = f'136->{C2} k3 T={T} B={BL}'
        sb = ops.SIGN_BIT_MASKS and T % 32 == 0 and T >= 512      # the step's formulation (ops.FilmCondFn)
        conv_case(f'FiLM cond_var.2 input-grad {tag}' + (' (1-bit LeakyReLU mask)' if sb else ''), 9, 'mfma', 136, C2, 3, 1, T, False, 1, 'dgrad', BL, bits=sb)
        if ops.FUSED_COND_FWD:
            cond_fwd_case(f'FiLM conditioning fwd (cond_var.0 fused into cond_var.2) {tag}', 9, C2, T, BL)
        else:       # the step's default: two launches
            conv_case(f'FiLM cond_var.0 excitation window 8->136 k3 T={T} B={BL} fwd (+ 3-valued embedding bias' + (', sign bits out)' if sb else ')'), 9, 'hbm',
                      8, 136, 3, 1, T, False, 0, 'fwd', BL, bias3=True, bits=sb)
            conv_case(f'FiLM cond_var.2 fwd {tag} (LeakyReLU on load)', 9, 'mfma', 136, C2, 3, 1, T, False, 1, 'fwd', BL)
        conv_case(f'FiLM cond_var.2 weight-grad {tag}', 9, 'mfma', 136, C2, 3, 1, T, False, 1, 'wgrad', BL)
        cond0_bwd_case(f'FiLM cond_var.0 backward (dexc + dW window + dk3) 136ch T={T} B={BL}', 9, T, BL)
        keep.clear()
    # discriminator layer 5 (1024 -> 1024, k5, T' = 63): disc 0 sees 2B, discs 1/2 see 4B (sub-scale pass batched on top)
    conv_case(f'D layer5 1024->1024 k5 T=63 B={2 * BL} fwd', 4, 'mfma', 1024, 1024, 5, 1, 63, False, 0, 'fwd', 2 * BL, post=1)
    conv_case(f'D layer5 1024->1024 k5 T=63 B={2 * BL} input-grad', 4, 'mfma', 1024, 1024, 5, 1, 63, False, 0, 'dgrad', 2 * BL, post=1)
    conv_case(f'D layer5 1024->1024 k5 T=63 B={2 * BL} weight-grad', 2, 'mfma', 1024, 1024, 5, 1, 63, False, 0, 'wgrad', 2 * BL, post=1)
    keep.clear()
    # dilated trunk convs (decoder stage 4: C=16, T=16000; one launch per (k, d) and direction) and the FiLM 1x1 posconv
    ns = {}
    for k, d in ((3, 1), (7, 3), (11, 5)):
        tag = f'dilated Conv1d 16->16 k{k} d{d} T=16000 B={BL}'
        conv_case(f'{tag} fwd (reflect pad, LeakyReLU-on-load, bias)' + (' [stand-alone launch: replaced in the step by the fused FiLM block]' if ops.FUSED_FILM_BLOCK else ''),
                  0 if ops.FUSED_FILM_BLOCK else 1, 'hbm', 16, 16, k, d, 16000, True, 1, 'fwd', BL)
        ns[(k, d)] = rows[-1]
        conv_case(f'{tag} input-grad (mirror fold + LeakyReLU mask)', 1, 'hbm', 16, 16, k, d, 16000, True, 1, 'dgrad', BL)
        conv_case(f'{tag} weight-grad', 1, 'hbm', 16, 16, k, d, 16000, True, 1, 'wgrad', BL)
        keep.clear()
    def film_block_case(label, n, k, d, T, Bc):
        """The fused FiLM-block forward (film_block.hip): dilated Conv1d + FiLM + 1x1 conv + residual in one launch."""
        w1 = torch.randn(16, 16, k, device=dev) / (16 * k) ** 0.5
        w2 = torch.randn(16, 16, 1, device=dev) / 4.0
        b1, b2 = torch.randn(16, device=dev) * 0.1, torch.randn(16, device=dev) * 0.1
        keep.extend([w1, w2, b1, b2])
        bufs = Bufs(torch, dev, dict(x=(Bc, 16, T), gb=(Bc, 32, T), h=(Bc, 16, T), y=(Bc, 16, T)))
        keep.append(bufs)
        st = torch.cuda.current_stream(dev).cuda_stream
        calls = []
        for s in bufs.sets:
            a = L.FilmBlockArgs(Bc, 16, T, k, d, s['x'].data_ptr(), s['x'].stride(0), w1.data_ptr(), b1.data_ptr(), s['h'].data_ptr(), s['h'].stride(0),
                                s['gb'].data_ptr(), s['gb'].stride(0), w2.data_ptr(), b2.data_ptr(), None, 0, 1.0, 0.2, s['y'].data_ptr(), s['y'].stride(0))
            keep.append(a)
            calls.append(lambda a=a: L.check(lib.tdvc_film_block_fwd(C.byref(a), st)))
        alg = 4.0 * Bc * T * (16 + 32 + 16 + 16) + 4.0 * (w1.numel() + w2.numel() + 32)
        add(label, n, 'hbm', alg, 2.0 * Bc * T * 16 * 16 * (k + 1), calls, bufs.bytes_per_rotation)

    fused = {}
    if ops.FUSED_FILM_BLOCK:      # the step's formulation of the 16-channel FiLM blocks: one launch per block and (k, d)
        for k, d in ((3, 1), (7, 3), (11, 5)):
            film_block_case(f'FiLM block fwd 16ch k{k} d{d} T=16000 B={BL}: dilated Conv1d (reflect, LeakyReLU-on-load) + FiLM + 1x1 conv + residual, one launch',
                            1, k, d, 16000, BL)
            fused[(k, d)] = rows[-1]
            keep.clear()
    else:
        conv_case(f'FiLM 1x1 posconv 16->16 T=16000 B={BL} fwd (h*(1+gamma)+beta on load, + residual)', 9, 'hbm', 16, 16, 1, 1, 16000, False, 1, 'fwd', BL,
                  film=True)
    conv_case(f'dilated Conv1d 64->64 k7 d3 T=4000 B={BL} fwd', 2, 'mfma', 64, 64, 7, 3, 4000, True, 1, 'fwd', BL)
    keep.clear()
    torch.cuda.empty_cache()
    # north star: the kernel that runs the 16 -> 16 k3 dilated Conv1d of the step -- the fused FiLM-block forward when it is on
    # (the stand-alone conv launch, which the step then no longer issues for these blocks, stays in the table next to it)
    north = dict(fused[(3, 1)], standalone_conv=dict(ns[(3, 1)])) if fused else ns[(3, 1)]
    rows.sort(key=lambda e: -e['share_of_step'])
    return rows, north


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


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(pkg, cfg_train, iters=3, sd_g=None, ssl=False):
    """CPU oracle timed on this host: B=2 x 1 s, 1 warm-up + `iters` timed iterations (~10-30 s)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from common import filled_sd
    from oracle import step as OS
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = OS.StepConfig.from_hparams(cfg_train)
    st = OS.TrainStep(sd_g if sd_g is not None else filled_sd('G'), filled_sd('D'), cfg, ssl_extractor=pkg.synth.FrameFeatureExtractor() if ssl else None)
    B, T = 2, SR
    bt = pkg.synth.make_batch(B, T, seed=1234, conversion=not cfg.no_conv)
    ix, iy = pkg.synth.contrastive_indices(B, T // 320, cfg.n_neg, 1), pkg.synth.contrastive_indices(B, T // 320, cfg.n_neg, 2)
    st.run(bt, ix, iy)
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter(); st.run(bt, ix, iy); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    return dict(value=B * T / SR / t, unit='audio-seconds/sec', cores=cores, kind='port', cpu=cpu_model(),
                sample=f'full D+G iteration of the same config, B={B} x 1 s, median of {iters} after 1 warm-up; '
                       f'CPU oracle (torch {torch.__version__} CPU, {torch.get_num_threads()} threads; recomputes the G-step '
                       'generator forward like the reference does, anomaly detection off)')


def pmc_traffic(op):
    """HBM bytes per launch of table entry `op` from the committed rocprofv3 PMC passes of this round (separate
    --pmc FETCH_SIZE / --pmc WRITE_SIZE runs over tools/microbench_kernels.py = this very table; FETCH_SIZE doubled per
    MI355X_MICROARCH.md): the latest profiles/r*_pmc.json, written by tools/pmc_traffic.py. None when the entry is not in it."""
    import glob
    try:
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc.json')))
        tab = json.load(open(files[-1]))
        return tab.get('entries', {}).get(op, {}).get('hbm_bytes_per_launch')
    except (OSError, ValueError, IndexError):
        return None


def main():
    args = parse_args()
    spawn_ranks_if_needed(args)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch
    import torch.distributed as dist

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
    from common import build_models, build_ssl_models, to_dev
    import warnings
    hp = pkg.hparams.HParam(os.path.join(ROOT, 'config', f'{args.config}.yaml'))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')      # lambda_f0: the CREPE-backed term is excluded by contract (named in config.workload)
        cfg = pkg.train_step.StepConfig.from_hparams(hp.train)
    ssl = args.config.startswith('wavlm')
    sd_g = None
    if ssl:
        G, D, sd_g = build_ssl_models(dev)
    else:
        G, D = build_models(dev)
    sync = pkg.parallel.GradSync() if dp else None
    if sync is not None:
        sync.broadcast_params(G.arena); sync.broadcast_params(D.arena)
    ts = pkg.train_step.TrainStep(G, D, cfg, dev, grad_sync=sync)
    B, T = args.batch, int(round(SR * args.seconds))
    if T % 320:
        raise SystemExit('--seconds must give a multiple of 320 samples (the content encoder hop)')
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
        step_ms = dt / args.steps * 1e3
        value = world * B * (T / SR) * args.steps / dt
        stage = 'stage-1' if 'stage1' in args.config else args.config
        out = dict(metric=f'audio-seconds/sec (G+D train step, {stage})', value=value, unit='audio-seconds/sec', n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=step_ms, higher_is_better=True,
                   scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload=f'config/{args.config}.yaml full D+G iteration, {B} x {args.seconds:g} s @16 kHz per GPU, NUM_SPK=16, '
                                        'F0 (CREPE) loss term excluded' + (', frozen SSL extractor = synthetic stand-in for WavLM-Large '
                                                                           '(plain PyTorch, inside the timed step)' if ssl else ''),
                               global_batch=world * B, parallelism=f'dp{world}'),
                   final_G_loss=g_loss, launch=launch)
        if world == 1 and not args.no_kernel_table and 'stage1' in args.config:      # rank 0 of a multi-rank run goes straight to the JSON line
            table, north = kernel_table(pkg, dev, B, step_ms)
            dom = table[0]
            roof = {k: dom[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'kernel', 'op', 'ms_per_launch', 'launches_per_step',
                                        'share_of_step', 'algorithmic_bytes', 'algorithmic_flops')}
            roof['traffic'] = pmc_traffic(dom['op'])
            roof['how'] = ('dominant kernel of the step by launches x measured time; HIP events on the launch stream over launches that rotate '
                           f'through {dom["rotation_bytes"] / 1e6:.0f} MB of operand sets (nothing served from the 256 MB Infinity Cache); '
                           'traffic = rocprofv3 PMC passes committed under profiles/ (null when absent)')
            # the kernel the north star names: stride-1 dilated Conv1d, 16 -> 16, k3 (HBM-bound end of the trunk)
            roof['north_star_kernel'] = dict(north, traffic=pmc_traffic(north['op']))
            roof['kernels'] = [{k: e[k] for k in ('kernel', 'op', 'launches_per_step', 'ms_per_launch', 'share_of_step', 'bound', 'achieved',
                                                  'peak', 'unit', 'frac')} for e in table[:12]]
            roof['table_share_of_step'] = sum(e['share_of_step'] for e in table)
            out['roofline'] = roof
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(pkg, hp.train, sd_g=sd_g, ssl=ssl)
            out['speedup_vs_cpu'] = value / out['cpu_baseline']['value']
        print(json.dumps(out), flush=True)
    if dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
