"""Micro-benchmark of single conv launches (for rocprofv3 / tuning). Not part of the product path.

    python tools/microbench_conv.py --shape c16k3 --which fwd --iters 20
"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('td-vc-gan_amd')
ops, arena, L = pkg.ops, pkg.arena, pkg._lib

SHAPES = {  # name: (cin, cout, k, dil, T, reflect, pre)
    'c16k3': (16, 16, 3, 1, 16000, True, 1), 'c16k11d5': (16, 16, 11, 5, 16000, True, 1), 'c16k1': (16, 16, 1, 1, 16000, False, 1),
    'c32k7d3': (32, 32, 7, 3, 8000, True, 1), 'c64k11': (64, 64, 11, 1, 4000, True, 1), 'c128k7': (128, 128, 7, 1, 500, True, 1),
    'cond0': (136, 136, 3, 1, 16000, False, 0), 'cond2_c16': (136, 32, 3, 1, 16000, False, 1), 'cond2_c64': (136, 128, 3, 1, 4000, False, 1),
    'cond0x': (8, 136, 3, 1, 16000, False, 0), 'c16k1_al': (16, 16, 1, 1, 16128, False, 1), 'c16k3_al': (16, 16, 3, 1, 16128, True, 1), 'd5': (1024, 1024, 5, 1, 63, False, 0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='c16k3')
    ap.add_argument('--which', default='fwd,dgrad,wgrad')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=16)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    for name in a.shape.split(','):
        cin, cout, k, dil, T, reflect, pre = SHAPES[name]
        B = a.batch
        pad = (k - 1) * dil // 2
        spec = ops.ConvSpec(cin, cout, k, 1, pad, dil, 1, reflect)
        w = torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5
        b = torch.randn(cout, device=dev) * 0.1
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        wt = w.permute(1, 0, 2).contiguous()
        spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), dw.data_ptr(), db.data_ptr(), True, None, wt.data_ptr())
        x = torch.randn(B, cin, T, device=dev)
        y = torch.empty(B, cout, T, device=dev)
        dy = torch.randn(B, cout, T, device=dev)
        dx = torch.empty_like(x)
        xf = ops._xf(L.XF_LRELU if pre else L.XF_NONE)
        fns = {
            'fwd': lambda: ops.conv_fwd_raw(spec, x, xf, out=y),
            'dgrad': lambda: ops.conv_dgrad_raw(spec, dy, ops._xf(), T, L.DG_MASK_LRELU if pre else L.DG_PLAIN, x_in=x if pre else None, out=dx),
            'wgrad': lambda: ops.conv_wgrad_raw(spec, x, xf, dy, ops._xf()),
        }
        for which in a.which.split(','):
            f = fns[which]
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(a.iters):
                f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            byt = 4.0 * B * T * (cin + cout) + 4.0 * w.numel()
            fl = 2.0 * B * T * cin * cout * k
            print(f'{name:10s} {which:6s} {ms * 1e3:9.1f} us  {byt / ms / 1e6:8.1f} GB/s  {fl / ms / 1e9:8.2f} TF/s', flush=True)


if __name__ == '__main__':
    main()
