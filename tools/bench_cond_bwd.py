"""Micro-benchmark: backward of the FiLM conditioning network behind cond_var.2's output gradient -- the fused launch
(tdvc_film_cond_bwd) against the two-launch path (cond_var.2 input-grad + tdvc_film_cond0_bwd) at the step's launch shapes,
rotating operand sets (> 600 MB per rotation). Diagnostic tool:  python tools/bench_cond_bwd.py [B]"""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('td-vc-gan_amd')
ops, L, arena = pkg.ops, pkg._lib, pkg.arena
lib = L.lib()
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nc, nv = 136, 8


def timeit(calls, iters=40):
    n = len(calls)
    for i in range(n + 2):
        calls[i % n]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters):
        calls[i % n]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for C2, T in ((32, 16000), (64, 8000), (128, 4000), (256, 500)):
    w0 = torch.randn(nc, nc, 3, device=dev) / (nc * 3) ** 0.5
    w2 = torch.randn(C2, nc, 3, device=dev) / (nc * 3) ** 0.5
    wt2 = w2.permute(1, 0, 2).contiguous()
    dw0 = torch.zeros_like(w0)
    spec2 = ops.ConvSpec(nc, C2, 3, pad=1)
    spec2.slot = arena.ConvSlot(w2.data_ptr(), 0, 0, 0, True, None, wt2.data_ptr())
    per = 4 * B * T * (C2 + nc + 2 * nv + nc / 32)
    nset = int(max(2, min(16, -(-600e6 // per))))
    st = torch.cuda.current_stream(dev).cuda_stream
    use_bits = T % 32 == 0
    sets = []
    for _ in range(nset):
        s = dict(dgb=torch.randn(B, C2, T, device=dev), exc=torch.randn(B, nv, T, device=dev), dexc=torch.empty(B, nv, T, device=dev),
                 dk3=torch.empty(B, nc, 3, device=dev), dcv=torch.empty(B, nc, T, device=dev))
        s['bits'] = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, nc, T // 32), dtype=torch.int32, device=dev) if use_bits else None
        s['cv0'] = None if use_bits else torch.randn(B, nc, T, device=dev)
        sets.append(s)
    nb_new = lib.tdvc_film_cond_bwd_workspace(B, T, nc, nv)
    nb_old = lib.tdvc_film_cond0_bwd_workspace(B, T, nc, nv)
    ws = torch.empty(max(nb_new, nb_old, 1), dtype=torch.uint8, device=dev)
    new_calls, old_calls, keep = [], [], []
    for s in sets:
        bits, cv0 = s['bits'], s['cv0']
        a = L.FilmCondBwdArgs(B, T, nc, nv, C2, s['dgb'].data_ptr(), s['dgb'].stride(0), wt2.data_ptr(),
                              bits.data_ptr() if bits is not None else None, bits.stride(0) if bits is not None else 0,
                              cv0.data_ptr() if cv0 is not None else None, cv0.stride(0) if cv0 is not None else 0,
                              s['exc'].data_ptr(), s['exc'].stride(0), w0.data_ptr(), s['dexc'].data_ptr(), s['dexc'].stride(0),
                              s['dk3'].data_ptr(), dw0.data_ptr(), ws.data_ptr(), ws.numel(), 0.2)
        a0 = L.FilmCond0BwdArgs(B, T, nc, nv, s['dcv'].data_ptr(), s['dcv'].stride(0), s['exc'].data_ptr(), s['exc'].stride(0), w0.data_ptr(),
                                s['dexc'].data_ptr(), s['dexc'].stride(0), s['dk3'].data_ptr(), dw0.data_ptr(), ws.data_ptr(), ws.numel())
        keep += [a, a0]
        new_calls.append(lambda a=a: (L.check(lib.tdvc_film_cond_bwd(C.byref(a), st)), L.check(lib.tdvc_fold_flush(st))))
        old_calls.append(lambda s=s, a0=a0, bits=bits, cv0=cv0: (
            ops.conv_dgrad_raw(spec2, s['dgb'], ops._xf(), T, L.DG_MASK_LRELU, x_in=cv0, out=s['dcv'], x_bits=bits),
            L.check(lib.tdvc_film_cond0_bwd(C.byref(a0), st)), L.check(lib.tdvc_fold_flush(st))))
    t_new, t_old = timeit(new_calls), timeit(old_calls)
    flops = 2.0 * B * T * nc * (C2 * 3 + 2 * nv * 3)
    alg = 4.0 * B * T * (C2 + 2 * nv + nc / 32.0)
    print(f'C2={C2:4d} T={T:6d} B={B}: fused {t_new:7.1f} us = {flops / t_new / 1e6:6.1f} TF ({flops / t_new / 1e6 / 157.3:.3f} of peak, '
          f'{alg / t_new / 1e3:6.0f} GB/s algorithmic) | two launches {t_old:7.1f} us | x{t_old / t_new:.2f}', flush=True)
    del sets, new_calls, old_calls, keep
    torch.cuda.empty_cache()
