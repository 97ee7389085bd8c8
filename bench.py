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
MFMA_BF16_PEAK_TF = 2516.6  # dense bf16 MFMA (MI355X_MICROARCH.md: 16x the fp32-input rate)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16, help='samples per GPU (1 s each)')
    ap.add_argument('--seconds', type=float, default=1.0, help='length of each sample in seconds (BASELINE config #5: 2)')
    ap.add_argument('--config', default='conv_enc-stage1', help='conv_enc-stage1 | conv_enc-stage2_1 | conv_enc-stage2_2 | wavlm-stage2_2 (frozen SSL '
                                                                "extractor = synth.FrameFeatureExtractor, the stand-in for WavLM-Large whose checkpoint is absent)")
    ap.add_argument('--ssl-extractor', default='wavlm-shape', choices=['wavlm-shape', 'frame'],
                    help="wavlm-* configs: the frozen extractor stand-in: 'wavlm-shape' = synth.WavLMShapedExtractor (stock torch.nn modules with "
                         "WavLM-Large's compute shape: 7 strided convs to 512 ch + 24 transformer layers 1024/16/4096, ~315 M parameters, random "
                         "init), 'frame' = the one-conv synth.FrameFeatureExtractor of the parity tests")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-table', action='store_true')
    ap.add_argument('--exact-fp32', action='store_true', help='exact-fp32 MFMA kernels everywhere (debug knobs 5, 6): no split-bf16 x6 kernels')
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


RIDGE_F_PER_B = MFMA_F32_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)      # 19.7 FLOP/byte: below it a kernel is priced against HBM


def record_launches(pkg, run_step):
    """One eager iteration with the launch recorder on (td-vc-gan_amd/ops.py: RECORDER): every conv-family entry point of
    the step reports its (op, geometry, operand transforms, launch shape). Returns {class: launches per iteration}."""
    import collections
    import torch
    pkg.ops.RECORDER = rec = []
    try:
        run_step()
        torch.cuda.synchronize()
    finally:
        pkg.ops.RECORDER = None
    return collections.Counter(rec)


def _conv_label(op, key, B, tin, extra):
    cin, cout, k, stride, pad, dil, groups, reflect, transposed, out_pad, w_cin, w_cin_off = key
    role = ''
    if cin == 136 and not transposed:
        role = 'FiLM cond_var.2 '
    elif cin == 8 and cout == 136:
        role = 'FiLM cond_var.0 excitation window '
    elif cin == 1024 and cout == 1024 and k == 5:
        role = 'D layer5 '
    elif groups > 1 and k == 41:
        role = 'D grouped strided '
    elif cin == cout and stride == 1 and k > 1 and reflect:
        role = 'dilated '
    name = {'fwd': 'fwd', 'dgrad': 'input-grad', 'wgrad': 'weight-grad'}[op]
    geo = f'k{k}' + (f' s{stride}' if stride > 1 else '') + (f' d{dil}' if dil > 1 else '') + (f' g{groups}' if groups > 1 else '')
    return f"{role}{'ConvTranspose1d' if transposed else 'Conv1d'} {cin}->{cout} {geo} T={tin} B={B} {name}" + (f' ({extra})' if extra else '')


def kernel_table(pkg, dev, classes, step_ms, iters=20, manifest=None, only_ops=None):
    """Per-kernel roofline table of EVERY conv-family launch class the recorded iteration issued, each rebuilt at its own
    launch shape on rotating operand sets (> 600 MB per rotation: nothing is served from the 256 MB Infinity Cache) and timed
    with HIP events on the launch stream. `classes` = record_launches(): {class tuple: launches per iteration}.
    Algorithmic bytes of a launch = 4 B x every element of every tensor operand it must read or write once (mask / FiLM /
    residual operands included; sign-bit masks as 1 bit) + the weights; FLOPs = 2 * B * Tout * Cout * Cin/groups * K
    (ConvTranspose: Tin). A class is priced against the HBM peak when its arithmetic intensity is below the fp32 ridge
    (157.3 TF / 8 TB/s = 19.7 FLOP/B), else against the fp32 MFMA peak."""
    import torch
    ops, arena, L = pkg.ops, pkg.arena, pkg._lib
    lib = L.lib()
    rows = []
    keep = []
    if manifest is not None:      # closes the segment of everything launched before the table (the recorded iteration)
        L.check(lib.tdvc_debug_marker(torch.cuda.current_stream(dev).cuda_stream))

    def traced(call):
        lib.tdvc_debug_trace(1)
        call()
        torch.cuda.synchronize()
        names = sorted(L.traced_kernels())
        lib.tdvc_debug_trace(0)
        return ' + '.join(names)

    def add(label, n, alg_bytes, flops, calls, rot_bytes, bound=None):
        if only_ops is not None and label not in only_ops:
            return None
        name = traced(calls[0])
        ms = time_launches(torch, calls, iters)
        if manifest is not None:      # tools/microbench_kernels.py: lets the PMC summary map dispatches back to this entry.
            # Deferred weight-grad folds are flushed into the entry, then an empty marker dispatch closes it (the number of fold
            # launches per call is not fixed, so the summary splits the trace at the markers instead of counting dispatches)
            ops.fold_flush(dev)
            L.check(lib.tdvc_debug_marker(torch.cuda.current_stream(dev).cuda_stream))
            manifest.append(dict(op=label, kernels=name.split(' + '), calls=1 + len(calls) + 2 + iters))
        if bound is None:
            bound = 'hbm' if flops / max(alg_bytes, 1.0) < RIDGE_F_PER_B else 'mfma'
        ach_b, ach_f = alg_bytes / (ms * 1e-3) / 1e9, flops / (ms * 1e-3) / 1e12
        e = dict(kernel=name, op=label, launches_per_step=n, ms_per_launch=ms, share_of_step=n * ms / step_ms, bound=bound,
                 achieved=ach_b if bound == 'hbm' else ach_f, peak=HBM_PEAK_GBS if bound == 'hbm' else MFMA_F32_PEAK_TF,
                 unit='GB/s' if bound == 'hbm' else 'TFLOP/s', algorithmic_bytes=alg_bytes, algorithmic_flops=flops,
                 rotation_bytes=rot_bytes)
        e['frac'] = e['achieved'] / e['peak']
        if '_x6_kernel' in name and 'film_cond_fwd' not in name:      # split-bf16 x6: six bf16 MFMA products per fp32-equivalent product; both accountings are shown
            e['bf16_pipe'] = dict(executed_tflops=6.0 * ach_f, peak=MFMA_BF16_PEAK_TF, frac=6.0 * ach_f / MFMA_BF16_PEAK_TF,
                                  note='`achieved` / `frac` above count the fp32-equivalent product once and price it against the fp32 MFMA '
                                       'peak (what the exact-fp32 kernel it replaces is priced against); this is the same launch as executed '
                                       'bf16 matrix FLOPs against the dense bf16 peak')
        rows.append(e)
        return e

    def rand_bits(shape):
        return torch.randint(-2 ** 31, 2 ** 31 - 1, shape, dtype=torch.int32, device=dev)

    def conv_class(rec, n):
        op, key, B, tin = rec[:4]
        cin, cout, k, stride, pad, dil, groups, reflect, transposed, out_pad, w_cin, w_cin_off = key
        spec = ops.ConvSpec(cin, cout, k, stride, pad, dil, groups, bool(reflect), bool(transposed), out_pad, w_cin, w_cin_off)
        tout = spec.tout(tin)
        wshape = (cin, cout // groups, k) if transposed else (cout, (w_cin or cin) // groups, k)
        w = torch.randn(wshape, device=dev) / (wshape[1] * k) ** 0.5
        b = torch.randn(cout, device=dev) * 0.1
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        wt = w.permute(1, 0, 2).contiguous() if (not transposed and groups == 1 and stride == 1) else None
        xs, ys = (B, cin, tin), (B, cout, tout)
        FILM, MASKS = L.XF_FILM_LRELU, (L.XF_MASK_LRELU, L.XF_MASK_TANH)
        shapes, bitshape, extra = {}, None, []
        if op == 'fwd':
            _, _, _, _, xk, post, has_res, has_add, has_bias, has_b3, has_bits = rec
            shapes = dict(x=xs, y=ys)
            if xk == FILM:
                shapes['aux'] = (B, 2 * cin, tin); extra.append('h*(1+gamma)+beta, LeakyReLU on load')
            elif xk in MASKS:
                shapes['aux'] = xs
            elif xk == L.XF_LRELU:
                extra.append('LeakyReLU on load')
            if has_res:
                shapes['res'] = ys; extra.append('+ residual')
            if has_add:
                shapes['add'] = ys; extra.append('+ add')
            if has_b3:
                extra.append('3-valued embedding bias')
            if has_bits:
                bitshape = (B, cout, tout // 32); extra.append('sign bits out')
            if post:
                extra.append({1: 'LeakyReLU', 2: 'tanh'}[post] + ' out')
        elif op == 'dgrad':
            _, _, _, _, dyk, epi, has_xin, has_bits, has_add, has_wt = rec
            has_bias = False
            shapes = dict(dy=ys, dx=xs)
            if dyk in MASKS:
                shapes['act'] = ys; extra.append('activation-grad mask on load')
            if epi == L.DG_MASK_LRELU:
                if has_bits:
                    bitshape = (B, cin, tin // 32); extra.append('1-bit LeakyReLU mask')
                elif has_xin:
                    shapes['x_in'] = xs; extra.append('LeakyReLU mask')
            elif epi == L.DG_FILM:
                shapes.update(x_in=xs, gb=(B, 2 * cin, tin), dgb=(B, 2 * cin, tin)); extra.append('FiLM backward epilogue')
            if has_add:
                shapes['add'] = xs; extra.append('+ add')
            if not has_wt:
                wt = None
        else:
            _, _, _, _, xk, dyk, has_bias = rec
            shapes = dict(x=xs, dy=ys)
            if xk == FILM:
                shapes['aux'] = (B, 2 * cin, tin); extra.append('FiLM + LeakyReLU on load')
            elif xk in MASKS:
                shapes['aux'] = xs
            if dyk in MASKS:
                shapes['act'] = ys; extra.append('masked dy')
            extra.append('+ slab fold')
        spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr() if has_bias else 0, dw.data_ptr(), db.data_ptr() if has_bias else 0, True, None,
                                   wt.data_ptr() if wt is not None else 0)
        k3b = torch.randn(B, cout, 3, device=dev) * 0.1 if (op == 'fwd' and rec[9]) else None
        bufs = Bufs(torch, dev, shapes)
        bitbufs = [rand_bits(bitshape) if bitshape else None for _ in bufs.sets]
        keep[:] = [w, b, dw, db, wt, k3b, bufs, bitbufs, spec]
        calls = []
        for sset, bw in zip(bufs.sets, bitbufs):
            g = sset.get
            if op == 'fwd':
                xf = ops._xf(rec[4], aux=g('aux'))
                calls.append(lambda s=sset, xf=xf, bw=bw: ops.conv_fwd_raw(spec, s['x'], xf, post=rec[5], res=s.get('res'), add=s.get('add'), out=s['y'],
                                                                           bias3=k3b, sign_bits=bw))
            elif op == 'dgrad':
                calls.append(lambda s=sset, bw=bw: ops.conv_dgrad_raw(spec, s['dy'], ops._xf(rec[4], aux=s.get('act')), tin, rec[5], x_in=s.get('x_in'),
                                                                      gb=s.get('gb'), dgb=s.get('dgb'), add=s.get('add'), out=s['dx'], x_bits=bw))
            else:
                calls.append(lambda s=sset: ops.conv_wgrad_raw(spec, s['x'], ops._xf(rec[4], aux=s.get('aux')), s['dy'], ops._xf(rec[5], aux=s.get('act'))))
        words = sum(int(torch.Size(sh).numel()) for sh in shapes.values()) + (int(torch.Size(bitshape).numel()) if bitshape else 0)
        alg = 4.0 * words + 4.0 * (w.numel() + (cout if has_bias else 0))
        flops = 2.0 * B * (tin if transposed else tout) * cout * (cin // groups) * k
        return add(_conv_label(op, key, B, tin, ', '.join(extra)), n, alg, flops, calls, bufs.bytes_per_rotation)

    def fwd_x6_class(rec, n):
        """Split-bf16 x6 forward of a 3-tap conv (conv_fwd_x6.hip): same algorithmic bytes / flops as the fp32 launch it replaces."""
        _, key, B, tin, xk, has_bias = rec
        cin, cout = key[0], key[1]
        spec = ops.ConvSpec(cin, cout, 3, 1, 1, 1, 1, False)
        w = torch.randn(cout, cin, 3, device=dev) / (cin * 3) ** 0.5
        b = torch.randn(cout, device=dev) * 0.1
        spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr() if has_bias else 0, 0, 0, True, None, 0)
        bufs = Bufs(torch, dev, dict(x=(B, cin, tin), y=(B, cout, tin)))
        planes = ops._weight_planes_x6(spec, dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        d = spec.desc(B, tin)
        calls, args = [], []
        for s in bufs.sets:
            a = L.ConvFwdArgs(s['x'].data_ptr(), s['x'].stride(0), ops._xf(xk), w.data_ptr(), b.data_ptr() if has_bias else None, None, 0, 0, 0.2, 1.0, None, 0,
                              s['y'].data_ptr(), s['y'].stride(0), None, None, 0)
            args.append(a)
            calls.append(lambda a=a: L.check(lib.tdvc_conv_fwd_x6(C.byref(d), C.byref(a), planes.data_ptr(), st)))
        keep[:] = [w, b, bufs, planes, args, spec, d]
        label = _conv_label('fwd', key, B, tin, ('LeakyReLU on load, ' if xk == L.XF_LRELU else '') + 'split-bf16 x6 MFMA, fp32 accumulate')
        return add(label, n, 4.0 * B * tin * (cin + cout) + 4.0 * (w.numel() + cout), 2.0 * B * tin * cin * cout * 3, calls, bufs.bytes_per_rotation)

    def film_block_class(rec, n):
        """The fused FiLM-block forward (film_block.hip): dilated Conv1d + FiLM + 1x1 conv + residual in one launch."""
        _, Bc, T, k, d, has_gb, has_acc = rec
        w1 = torch.randn(16, 16, k, device=dev) / (16 * k) ** 0.5
        w2 = torch.randn(16, 16, 1, device=dev) / 4.0
        b1, b2 = torch.randn(16, device=dev) * 0.1, torch.randn(16, device=dev) * 0.1
        shapes = dict(x=(Bc, 16, T), h=(Bc, 16, T), y=(Bc, 16, T))
        if has_gb:
            shapes['gb'] = (Bc, 32, T)
        if has_acc:
            shapes['acc'] = (Bc, 16, T)
        bufs = Bufs(torch, dev, shapes)
        st = torch.cuda.current_stream(dev).cuda_stream
        calls, args = [], []
        for s in bufs.sets:
            gb, acc = s.get('gb'), s.get('acc')
            a = L.FilmBlockArgs(Bc, 16, T, k, d, s['x'].data_ptr(), s['x'].stride(0), w1.data_ptr(), b1.data_ptr(), s['h'].data_ptr(), s['h'].stride(0),
                                gb.data_ptr() if gb is not None else None, gb.stride(0) if gb is not None else 0, w2.data_ptr(), b2.data_ptr(),
                                acc.data_ptr() if acc is not None else None, acc.stride(0) if acc is not None else 0, 1.0, 0.2, s['y'].data_ptr(), s['y'].stride(0))
            args.append(a)
            calls.append(lambda a=a: L.check(lib.tdvc_film_block_fwd(C.byref(a), st)))
        keep[:] = [w1, w2, b1, b2, bufs, args]
        words = sum(int(torch.Size(sh).numel()) for sh in shapes.values())
        label = (f'FiLM block fwd 16ch k{k} d{d} T={T} B={Bc}: dilated Conv1d (reflect, LeakyReLU-on-load) + ' + ('FiLM + ' if has_gb else '') +
                 '1x1 conv + residual' + (' + MRF sum' if has_acc else '') + ', one launch')
        return add(label, n, 4.0 * words + 4.0 * (w1.numel() + w2.numel() + 32), 2.0 * Bc * T * 16 * 16 * (k + 1), calls, bufs.bytes_per_rotation)

    def cond_bwd_class(rec, n):
        """Fused backward of the conditioning network behind cond_var.2's output gradient (film_cond_fused_bwd.hip)."""
        _, Bc, T, nc, nv, C2, has_bits, has_dexc, want_w = rec
        w0 = torch.randn(nc, nc, 3, device=dev) / (nc * 3) ** 0.5
        wt2 = torch.randn(nc, C2, 3, device=dev) / (nc * 3) ** 0.5
        dw0 = torch.zeros_like(w0)
        shapes = dict(dgb=(Bc, C2, T), exc=(Bc, nv, T), dexc=(Bc, nv, T), dk3=(Bc, nc, 3))
        if not has_bits:
            shapes['cv0'] = (Bc, nc, T)
        bufs = Bufs(torch, dev, shapes)
        bitbufs = [rand_bits((Bc, nc, T // 32)) if has_bits else None for _ in bufs.sets]
        ws = torch.empty(max(lib.tdvc_film_cond_bwd_workspace(Bc, T, nc, nv), 1), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        calls, args = [], []
        for s, bw in zip(bufs.sets, bitbufs):
            cv0 = s.get('cv0')
            a = L.FilmCondBwdArgs(Bc, T, nc, nv, C2, s['dgb'].data_ptr(), s['dgb'].stride(0), wt2.data_ptr(),
                                  bw.data_ptr() if bw is not None else None, bw.stride(0) if bw is not None else 0,
                                  cv0.data_ptr() if cv0 is not None else None, cv0.stride(0) if cv0 is not None else 0,
                                  s['exc'].data_ptr(), s['exc'].stride(0), w0.data_ptr(), s['dexc'].data_ptr() if has_dexc else None, s['dexc'].stride(0),
                                  s['dk3'].data_ptr(), dw0.data_ptr() if want_w else None, ws.data_ptr(), ws.numel(), 0.2)
            args.append(a)
            calls.append(lambda a=a: (L.check(lib.tdvc_film_cond_bwd(C.byref(a), st)), L.check(lib.tdvc_fold_flush(st))))
        keep[:] = [w0, wt2, dw0, bufs, bitbufs, ws, args]
        words = Bc * T * (C2 + 2 * nv + (nc / 32.0 if has_bits else nc))
        label = (f'FiLM conditioning backward {C2}->136->8 T={T} B={Bc}: cond_var.2 input-grad + ' + ('1-bit ' if has_bits else 'fp32 ') +
                 'LeakyReLU mask + cond_var.0 backward (dexc, dW window, dk3), one launch (+ slab fold)')
        return add(label, n, 4.0 * words + 4.0 * (wt2.numel() + nc * nv * 3), 2.0 * Bc * T * nc * (C2 * 3 + 2 * nv * 3), calls, bufs.bytes_per_rotation)

    def cond_fwd_x6_class(rec, n):
        """Conditioning forward in one launch (conv_fwd_x6.hip, film_cond_fwd_x6_kernel): cond_var.0's excitation window on the fly + cond_var.2
        in split-bf16 x6. Algorithmic bytes: exc and k3 read; cv0 (fp32, for the backward), its sign bits and gb written; weights."""
        _, Bc, T, nc, nv, C2, has_bits = rec
        w0 = torch.randn(nc, nc, 3, device=dev) / (nc * 3) ** 0.5
        w2 = torch.randn(C2, nc, 3, device=dev) / (nc * 3) ** 0.5
        b2 = torch.randn(C2, device=dev) * 0.1
        spec2 = ops.ConvSpec(nc, C2, 3, 1, 1, 1, 1, False)
        spec2.slot = arena.ConvSlot(w2.data_ptr(), b2.data_ptr(), 0, 0, True, None, 0)
        planes = ops._weight_planes_x6(spec2, dev)
        bufs = Bufs(torch, dev, dict(exc=(Bc, nv, T), k3=(Bc, nc, 3), cv0=(Bc, nc, T), gb=(Bc, C2, T)))
        bitbufs = [torch.empty((Bc, nc, T // 32), dtype=torch.int32, device=dev) if has_bits else None for _ in bufs.sets]
        st = torch.cuda.current_stream(dev).cuda_stream
        calls, args = [], []
        for s, bw in zip(bufs.sets, bitbufs):
            a = L.FilmCondArgs(Bc, T, nc, nv, C2, s['exc'].data_ptr(), s['exc'].stride(0), w0.data_ptr(), s['k3'].data_ptr(), w2.data_ptr(), b2.data_ptr(),
                               s['cv0'].data_ptr(), s['cv0'].stride(0), s['gb'].data_ptr(), s['gb'].stride(0), 0.2)
            args.append(a)
            calls.append(lambda a=a, bw=bw: L.check(lib.tdvc_film_cond_fwd_x6(C.byref(a), planes.data_ptr(), bw.data_ptr() if bw is not None else None,
                                                                              bw.stride(0) if bw is not None else 0, st)))
        keep[:] = [w0, w2, b2, spec2, planes, bufs, bitbufs, args]
        words = Bc * T * (nv + nc + C2 + (nc / 32.0 if has_bits else 0.0))
        label = (f'FiLM conditioning forward 8->{nc}->{C2} T={T} B={Bc}: cond_var.0 excitation window + LeakyReLU + cond_var.2 (split-bf16 x6), one launch; '
                 'cv0' + (' + sign bits' if has_bits else '') + ' stored for the backward')
        return add(label, n, 4.0 * words + 4.0 * (w2.numel() + nc * nv * 3 + Bc * nc * 3), 2.0 * Bc * T * nc * 3 * (C2 + nv), calls, bufs.bytes_per_rotation)

    def cond0_bwd_class(rec, n):
        _, Bc, T, nc, nv, has_dexc, want_w = rec
        w0 = torch.randn(nc, nc, 3, device=dev) / (nc * 3) ** 0.5
        dw0 = torch.zeros_like(w0)
        bufs = Bufs(torch, dev, dict(dcv=(Bc, nc, T), exc=(Bc, nv, T), dexc=(Bc, nv, T), dk3=(Bc, nc, 3)))
        ws = torch.empty(max(lib.tdvc_film_cond0_bwd_workspace(Bc, T, nc, nv), 1), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        calls, args = [], []
        for s in bufs.sets:
            a = L.FilmCond0BwdArgs(Bc, T, nc, nv, s['dcv'].data_ptr(), s['dcv'].stride(0), s['exc'].data_ptr(), s['exc'].stride(0), w0.data_ptr(),
                                   s['dexc'].data_ptr() if has_dexc else None, s['dexc'].stride(0), s['dk3'].data_ptr(), dw0.data_ptr() if want_w else None,
                                   ws.data_ptr(), ws.numel())
            args.append(a)
            calls.append(lambda a=a: (L.check(lib.tdvc_film_cond0_bwd(C.byref(a), st)), L.check(lib.tdvc_fold_flush(st))))
        keep[:] = [w0, dw0, bufs, ws, args]
        return add(f'FiLM cond_var.0 backward (dexc + dW window + dk3) 136ch T={T} B={Bc} (+ slab fold)', n, 4.0 * Bc * T * (nc + 2 * nv),
                   2.0 * Bc * T * nc * nv * 3 * 2, calls, bufs.bytes_per_rotation)

    handlers = dict(fwd=conv_class, dgrad=conv_class, wgrad=conv_class, fwd_x6=fwd_x6_class, film_cond_fwd_x6=cond_fwd_x6_class, film_block_fwd=film_block_class, film_cond_bwd=cond_bwd_class,
                    film_cond0_bwd=cond0_bwd_class)
    north = None
    for rec, n in sorted(classes.items(), key=lambda kv: str(kv[0])):
        e = handlers[rec[0]](rec, n)
        if e is not None and rec[0] == 'film_block_fwd' and rec[3] == 3 and rec[4] == 1 and rec[5] and (north is None or e['launches_per_step'] >= north['launches_per_step']):
            north = e
        keep.clear()
        torch.cuda.empty_cache()
    # the kernel the north star names: the stride-1 dilated Conv1d 16 -> 16 k3 (HBM-bound end of the trunk). In the step it runs
    # inside the fused FiLM-block forward; the stand-alone conv launch (which the step no longer issues for these blocks) stays
    # next to it as the un-fused per-conv figure of SURVEY §8(d)
    BL = max((rec[1] for rec in classes if rec[0] == 'film_block_fwd'), default=32)      # the generator's trunk batch ([cond 0; cond 1])
    alone = conv_class(('fwd', (16, 16, 3, 1, 1, 1, 1, 1, 0, 0, 0, 0), BL, 16000, L.XF_LRELU, 0, False, False, True, False, False), 0)
    keep.clear()
    torch.cuda.empty_cache()
    if alone is not None:
        alone['op'] += ' [stand-alone launch: replaced in the step by the fused FiLM block]' if north is not None else ''
        north = dict(north, standalone_conv=dict(alone)) if north is not None else alone
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


def cpu_baseline(pkg, cfg_train, iters=3, sd_g=None, ssl=False, extractor=None):
    """CPU oracle timed on this host: B=2 x 1 s, 1 warm-up + `iters` timed iterations (~10-30 s)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from common import filled_sd
    from oracle import step as OS
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = OS.StepConfig.from_hparams(cfg_train)
    st = OS.TrainStep(sd_g if sd_g is not None else filled_sd('G'), filled_sd('D'), cfg, ssl_extractor=(extractor if extractor is not None else pkg.synth.FrameFeatureExtractor()) if ssl else None)
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


def lib_sha16():
    """sha256 (first 16 hex digits) of the HIP library this process runs: ties PMC tables to the build they were taken on."""
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, 'td-vc-gan_amd', 'csrc', 'libtdvc_hip.so'), 'rb').read()).hexdigest()[:16]
    except OSError:
        return None


def pmc_traffic(op):
    """HBM bytes per launch of table entry `op` from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE /
    --pmc WRITE_SIZE runs over tools/microbench_kernels.py = this very table; FETCH_SIZE doubled per MI355X_MICROARCH.md):
    the latest profiles/r*_pmc.json written by tools/pmc_traffic.py -- but ONLY when that file was taken on the very library
    build that is running now (its `lib_sha16`); otherwise None: counters of another build say nothing about this one."""
    import glob
    try:
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc.json')))
        tab = json.load(open(files[-1]))
        if not tab.get('lib_sha16') or tab['lib_sha16'] != lib_sha16():
            return None
        return tab.get('entries', {}).get(op, {}).get('hbm_bytes_per_launch')
    except (OSError, ValueError, IndexError):
        return None


DTYPE_EXACT = 'f32'
DTYPE_DEFAULT = ('f32 (fp32 storage and accumulation everywhere; FiLM cond_var.2 forward and weight-grad as split-bf16 x6 products -- three exact bf16 '
                 'pieces per operand, six piece products on the bf16 MFMA, fp32 accumulate: fp32-level accuracy; every other kernel exact-fp32 MFMA)')


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
    if args.exact_fp32:
        pkg._lib.lib().tdvc_debug_knob(5, 1); pkg._lib.lib().tdvc_debug_knob(6, 1)
        pkg.ops.X6_FWD = False      # (knob 6 alone would still split the weights before the entry point declines)
    from common import build_models, build_ssl_models, to_dev
    import warnings
    hp = pkg.hparams.HParam(os.path.join(ROOT, 'config', f'{args.config}.yaml'))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')      # lambda_f0: the CREPE-backed term is excluded by contract (named in config.workload)
        cfg = pkg.train_step.StepConfig.from_hparams(hp.train)
    ssl = args.config.startswith('wavlm')
    sd_g = None
    extractor = None
    if ssl:
        extractor = pkg.synth.WavLMShapedExtractor() if args.ssl_extractor == 'wavlm-shape' else pkg.synth.FrameFeatureExtractor()
        G, D, sd_g = build_ssl_models(dev, extractor)
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

    # which conv-family launches does one iteration of THIS configuration issue? (recorded on an eager warm-up iteration; the
    # roofline table below rebuilds every recorded (op, launch shape) class)
    classes = None
    if rank == 0 and world == 1 and not args.no_kernel_table:
        classes = record_launches(pkg, lambda: ts.run(bt, ix, iy))

    use_graph = (sync is None) and not args.no_graph      # data-parallel steps run eagerly (see TrainStep.capture)
    launch = 'eager'
    step = None
    if use_graph:
        try:
            step = ts.capture(bt, ix, iy, warmup=2)      # whole iteration as one hipGraph
            launch = 'hipGraph replay'
        except Exception as e:      # noqa: BLE001 -- fall back to eager launches rather than lose the measurement
            print(f'[bench] rank {rank}: graph capture failed ({type(e).__name__}: {e}); running eagerly', file=sys.stderr, flush=True)
            pkg._lib.lib().tdvc_fold_reset(torch.cuda.current_stream(dev).cuda_stream)     # folds queued by the abandoned capture
            step = None
    if step is None:
        def step():
            return ts.run(bt, ix, iy)

    for _ in range(args.warmup):
        log = step()
    torch.cuda.synchronize()
    if rank == 0:
        print(f'[bench] warm-up done ({args.warmup} steps)', file=sys.stderr, flush=True)
    if dp:
        ts.comm_events = []          # bracket every pre-optimizer wait for the gradient all-reduce with an event pair
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # per-step timings (median), no host sync
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        log = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2]
    my_ms = dt / args.steps * 1e3
    dp_info = None
    if dp:
        ev = ts.comm_events or []
        exposed = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)
        # proof of the collective's world size: an all-reduce of ones over the process group that carried the gradients
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        stats = torch.tensor([my_ms, exposed], device=dev, dtype=torch.float64)
        allst = [torch.zeros_like(stats) for _ in range(world)]
        dist.all_gather(allst, stats)
        dp_info = dict(rccl_world=int(round(float(ones))), backend=dist.get_backend(), per_rank_ms=[round(float(t_[0]), 3) for t_ in allst],
                       exposed_comm_ms=[round(float(t_[1]), 4) for t_ in allst], allreduce_calls_per_step=sync.calls / max(1, args.steps + args.warmup),
                       grad_segments=dict(G=G.arena.nseg, D=D.arena.nseg), grad_bytes=4 * (G.arena.n_live + D.arena.n_live),
                       how='exposed_comm_ms = per step, time the compute stream waited for the side-stream all-reduces right before the two optimizer '
                           'launches (event pair around GradSync.wait); everything else of the exchange ran under the backward pass')
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    g_loss = float(log['G_loss'])
    # the same K steps on the exact-fp32 kernels (split-bf16 x6 kernels off), reported beside the headline number
    fp32_ms = None
    if rank == 0 and world == 1 and not dp and not args.exact_fp32 and use_graph:
        lib = pkg._lib.lib()
        lib.tdvc_debug_knob(5, 1); lib.tdvc_debug_knob(6, 1)
        pkg.ops.X6_FWD = False
        try:
            step32 = ts.capture(bt, ix, iy, warmup=1)
            for _ in range(2):
                step32()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step32()
            torch.cuda.synchronize()
            fp32_ms = (time.perf_counter() - t1) / args.steps * 1e3
        except Exception as e:      # noqa: BLE001
            print(f'[bench] exact-fp32 comparison run failed ({type(e).__name__}: {e})', file=sys.stderr, flush=True)
            lib.tdvc_fold_reset(torch.cuda.current_stream(dev).cuda_stream)
        finally:
            lib.tdvc_debug_knob(5, 0); lib.tdvc_debug_knob(6, 0)
            pkg.ops.X6_FWD = True
    if rank == 0:
        print(f'[bench] timed region: {dt / args.steps * 1e3:.2f} ms/step (median of the per-step event timings {ms_median:.2f})', file=sys.stderr, flush=True)
    if not (g_loss == g_loss):
        raise SystemExit('non-finite loss in the timed region')

    if rank == 0:
        step_ms = dt / args.steps * 1e3
        value = world * B * (T / SR) * args.steps / dt
        stage = 'stage-1' if 'stage1' in args.config else args.config
        out = dict(metric=f'audio-seconds/sec (G+D train step, {stage})', value=value, unit='audio-seconds/sec', n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=step_ms, higher_is_better=True,
                   scaling='weak', vs_baseline=None, dtype=DTYPE_EXACT if args.exact_fp32 else DTYPE_DEFAULT, data='synthetic',
                   config=dict(workload=f'config/{args.config}.yaml full D+G iteration, {B} x {args.seconds:g} s @16 kHz per GPU, NUM_SPK=16, '
                                        'F0 (CREPE) loss term excluded' + ((', frozen SSL extractor = ' + ('stock-torch.nn stand-in with WavLM-Large\'s compute '
                                                                           'shape (7 strided convs + 24 transformer layers 1024/16/4096, random init)'
                                                                           if args.ssl_extractor == 'wavlm-shape' else 'one-conv synthetic stand-in') +
                                                                           ', plain PyTorch-ROCm, inside the timed step') if ssl else ''),
                               global_batch=world * B, parallelism=f'dp{world}'),
                   final_G_loss=g_loss, launch=launch, ms_per_step_median=ms_median,
                   timing='value / ms_per_step: K steps between two device synchronisations (the contract); ms_per_step_median: median of the K '
                          'per-step HIP-event timings inside that region')
        out['arithmetic'] = dict(default=DTYPE_DEFAULT, this_run=DTYPE_EXACT if args.exact_fp32 else DTYPE_DEFAULT,
                                 exact_fp32_ms_per_step=step_ms if args.exact_fp32 else fp32_ms,
                                 note='op-level error vs float64 of the split-bf16 kernels: 1.2e-7 (weight-grad) / 3.0e-7 (forward), of the fp32 MFMA '
                                      'kernels they replace: 1.6e-7 / 3.5e-7 (tests/test_kernel_instances_gpu.py, 2e-5 gate); --exact-fp32 selects the latter')
        if dp_info is not None:
            out['data_parallel'] = dp_info
        if ssl:
            # how much of the step is the frozen extractor (plain PyTorch-ROCm, not this repo's code)? One iteration calls it on
            # [real; corrupted] (2B signals) and, with the cycle branch, on the converted signal (B signals): model/ssl_encoder.py:141-145
            with torch.no_grad():
                pads = [torch.nn.functional.pad(torch.cat([bt['signal_real'], bt['signal_corrupted']], 0), (160, 0)).squeeze(1)]
                if cfg.lambda_rec > 0 and not cfg.no_conv:
                    pads.append(torch.nn.functional.pad(bt['signal_real'], (160, 0)).squeeze(1))
                def ext_once():
                    for w_ in pads:
                        G.encoder.cmodel.extract_features(w_)
                for _ in range(3):
                    ext_once()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(10):
                    ext_once()
                e1.record(); torch.cuda.synchronize()
            ext_ms = e0.elapsed_time(e1) / 10
            out['ssl_extractor'] = dict(kind=args.ssl_extractor, parameters=sum(p.numel() for p in G.encoder.cmodel.parameters()),
                                        extractor_ms_per_step=ext_ms, hip_path_ms_per_step=step_ms - ext_ms,
                                        how='extractor calls of one iteration timed on their own (HIP events, 10 repeats); hip_path = step - extractor')
        if classes is not None:      # rank 0 of a multi-rank run goes straight to the JSON line
            table, north = kernel_table(pkg, dev, classes, step_ms)
            try:
                os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
                json.dump(dict(step_ms=step_ms, lib_sha16=lib_sha16(), table=table), open(os.path.join(ROOT, 'gpurun_out', 'kernel_table.json'), 'w'), indent=1)
            except OSError:
                pass
            dom = table[0]
            roof = {k: dom[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'kernel', 'op', 'ms_per_launch', 'launches_per_step',
                                        'share_of_step', 'algorithmic_bytes', 'algorithmic_flops')}
            roof['traffic'] = pmc_traffic(dom['op'])
            if roof['traffic'] is not None and dom['bound'] == 'mfma' and roof['traffic'] / (dom['ms_per_launch'] * 1e-3) > 0.4 * HBM_PEAK_GBS * 1e9:
                roof['bound_note'] = 'also HBM-limited: measured traffic / time exceeds 0.4 of the HBM peak'
            roof['how'] = ('dominant kernel of the step by launches x measured time, out of EVERY conv-family launch class one recorded iteration '
                           'issues; HIP events on the launch stream over launches that rotate '
                           f'through {dom["rotation_bytes"] / 1e6:.0f} MB of operand sets (nothing served from the 256 MB Infinity Cache); '
                           'traffic = rocprofv3 PMC passes committed under profiles/, only when taken on this very library build (else null)')
            # the kernel the north star names: stride-1 dilated Conv1d, 16 -> 16, k3 (HBM-bound end of the trunk)
            roof['north_star_kernel'] = dict(north, traffic=pmc_traffic(north['op']))
            roof['kernels'] = [{k: e[k] for k in ('kernel', 'op', 'launches_per_step', 'ms_per_launch', 'share_of_step', 'bound', 'achieved',
                                                  'peak', 'unit', 'frac', 'bf16_pipe') if k in e} for e in table[:12]]
            roof['table_share_of_step'] = sum(e['share_of_step'] for e in table)
            roof['table_entries'] = len(table)
            by = {}
            for e in table:
                by.setdefault(e['bound'], [0.0, 0.0])
                by[e['bound']][0] += e['share_of_step']; by[e['bound']][1] += e['share_of_step'] * e['frac']
            roof['share_weighted_frac'] = {k: dict(share_of_step=v[0], mean_frac_of_peak=v[1] / max(v[0], 1e-12)) for k, v in by.items()}
            out['roofline'] = roof
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(pkg, hp.train, sd_g=sd_g, ssl=ssl, extractor=((pkg.synth.WavLMShapedExtractor() if args.ssl_extractor == 'wavlm-shape' else pkg.synth.FrameFeatureExtractor()) if ssl else None))
            out['speedup_vs_cpu'] = value / out['cpu_baseline']['value']
        print(json.dumps(out), flush=True)
    if dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
