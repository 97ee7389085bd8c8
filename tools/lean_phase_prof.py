"""Phase-cycle breakdown of the lean conv kernel (diagnostic; needs `make -C td-vc-gan_amd/csrc prof`).

Per block (thread 0) the instrumented build accumulates s_memtime deltas per phase:
  0 prologue  1 barrier(top)  2 commit x+w  3 barrier(staged)  4 issue next  5 MFMA loop  6 epilogue  7 realtime(100 MHz)
"""
import argparse
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('td-vc-gan_amd')
ops, arena, L = pkg.ops, pkg.arena, pkg._lib
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), 'libtdvc_hip_prof.so')
L.SIGNATURES['tdvc_debug_lean_prof'] = (C.c_int, [C.c_void_p])
L.SIGNATURES['tdvc_debug_wgrad_prof'] = (C.c_int, [C.c_void_p])

SHAPES = {  # name: (cin, cout, k, dil, T, reflect, pre)
    'c16k3': (16, 16, 3, 1, 16000, True, 1), 'c16k11d5': (16, 16, 11, 5, 16000, True, 1), 'c16k1': (16, 16, 1, 1, 16000, False, 1),
    'c32k7d3': (32, 32, 7, 3, 8000, True, 1), 'c64k11': (64, 64, 11, 1, 4000, True, 1), 'c128k7': (128, 128, 7, 1, 500, True, 1),
    'cond2_c16': (136, 32, 3, 1, 16000, False, 1), 'cond2_c32': (136, 64, 3, 1, 8000, False, 1), 'cond2_c64': (136, 128, 3, 1, 4000, False, 1),
    'd5': (1024, 1024, 5, 1, 63, False, 0), 'cond2_c64s': (136, 128, 3, 1, 1024, False, 1), 'cond2_c64m': (136, 128, 3, 1, 2048, False, 1),
}
NAMES = ['prologue', 'bar_top', 'commit', 'bar_staged', 'issue', 'mfma', 'epilogue']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='cond2_c64,cond2_c16,c16k3,c64k11')
    ap.add_argument('--which', default='fwd,dgrad')
    ap.add_argument('--batch', type=int, default=32)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = L.lib()
    buf = torch.zeros(10 * 1 << 20, dtype=torch.int64, device=dev)
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
            setp = lib.tdvc_debug_wgrad_prof if which == 'wgrad' else lib.tdvc_debug_lean_prof
            setp(None)
            for _ in range(20):
                f()
            buf.zero_()
            setp(buf.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            f()
            e1.record(); torch.cuda.synchronize()
            setp(None)
            r = buf.view(-1, 10).cpu()
            r = r[r[:, 7] > 0].double()
            tot = r[:, :7].sum(1) + r[:, 8]
            clk = (tot / r[:, 7]).median().item() * 100.0      # MHz
            fl = 2.0 * B * T * cin * cout * k
            print(f'{name:10s} {which:6s} blocks={r.shape[0]:6d} wall={e0.elapsed_time(e1) * 1e3:7.1f} us  block cycles mean={tot.mean():9.0f} '
                  f'(= {tot.mean() / clk:6.1f} us @ {clk:5.0f} MHz)  ideal MFMA/SIMD-wave={fl / (r.shape[0] * 4) / 2048 * 32:8.0f} cyc', flush=True)
            print('     mean ' + '  '.join(f'{n}={r[:, i].mean():8.0f}' for i, n in enumerate(NAMES)) + f'  vmwait={r[:, 8].mean():8.0f}', flush=True)
            if which != 'wgrad':
                full = buf.view(-1, 10).cpu().double()
                nt = (T + 255) // 256 if T > 80 else 1
                for cand in (256, 64):
                    gx = (T + cand - 1) // cand
                    if r.shape[0] % gx == 0: nt = gx
                nb = r.shape[0]
                ids = torch.arange(nb) % nt
                fb = full[:nb]
                for nm, sel in (('end ', (ids == 0) | (ids == nt - 1)), ('mid ', (ids != 0) & (ids != nt - 1))):
                    if sel.any():
                        print(f'     {nm} ' + '  '.join(f'{n}={fb[sel][:, i].mean():8.0f}' for i, n in enumerate(NAMES)) + f'  total={(fb[sel][:, :7].sum(1) + fb[sel][:, 8]).mean():8.0f}  (gx={nt})', flush=True)
            print('     p50  ' + '  '.join(f'{n}={r[:, i].median():8.0f}' for i, n in enumerate(NAMES)) + f'  total p50={tot.median():8.0f} max={tot.max():8.0f}', flush=True)


if __name__ == '__main__':
    main()
