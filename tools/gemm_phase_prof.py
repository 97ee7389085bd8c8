"""Phase-cycle breakdown of the generic (strided / grouped / transposed) conv kernel (diagnostic; needs `make -C td-vc-gan_amd/csrc prof`).

Per block (thread 0) the instrumented build accumulates s_memtime deltas per phase:
  0 prologue  1 barrier(top)  2 stage x  8 stage w  3 barrier(staged)  4 issue next  5 MFMA loop  6 epilogue  7 realtime(100 MHz)
"""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
pkg = importlib.import_module('td-vc-gan_amd')
ops, arena, L = pkg.ops, pkg.arena, pkg._lib
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), 'libtdvc_hip_prof.so')
L.SIGNATURES['tdvc_debug_gemm_prof'] = (C.c_int, [C.c_void_p])
import generic_table as gt  # noqa: E402

NAMES = {0: 'prologue', 1: 'bar_top', 2: 'stage_x', 8: 'stage_w', 3: 'bar_staged', 4: 'issue', 5: 'mfma', 6: 'epilogue'}


def main():
    want = sys.argv[1].split(',') if len(sys.argv) > 1 else ['d1_grp_16_64', 'down_64_128_s8', 'up_128_64_s8', 'down_16_32_s2', 'd4_grp_1024_1024']
    dev = torch.device('cuda:0')
    lib = L.lib()
    buf = torch.zeros(10 * (1 << 20), dtype=torch.int64, device=dev)
    for (name, cin, cout, k, s, p, d, g, reflect, transposed, T, pre, post), B in gt.CASES:
        if name not in want:
            continue
        spec = ops.ConvSpec(cin, cout, k, s, p, d, g, reflect, transposed)
        tout = spec.tout(T)
        wshape = (cin, cout // g, k) if transposed else (cout, cin // g, k)
        w = torch.randn(wshape, device=dev) / (wshape[1] * k) ** 0.5
        b = torch.randn(cout, device=dev) * 0.1
        spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), 0, 0, False, None, 0)
        x, y, dx = torch.randn(B, cin, T, device=dev), torch.randn(B, cout, tout, device=dev), torch.empty(B, cin, T, device=dev)
        xf = ops._xf(L.XF_LRELU) if pre else ops._xf()
        dyxf = ops._xf(L.XF_MASK_LRELU, aux=y) if post == 1 else ops._xf()
        fns = {'fwd': lambda: ops.conv_fwd_raw(spec, x, xf, post=post, out=y),
               'dgrad': lambda: ops.conv_dgrad_raw(spec, y, dyxf, T, L.DG_MASK_LRELU if pre else L.DG_PLAIN, x_in=x if pre else None, out=dx)}
        for which, f in fns.items():
            lib.tdvc_debug_gemm_prof(None)
            for _ in range(10):
                f()
            buf.zero_()
            lib.tdvc_debug_gemm_prof(buf.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            f()
            e1.record(); torch.cuda.synchronize()
            lib.tdvc_debug_gemm_prof(None)
            r = buf.view(-1, 10).cpu()
            r = r[r[:, 7] > 0].double()
            if not r.shape[0]:
                print(f'{name} {which}: no stamped blocks (not the generic kernel)')
                continue
            tot = r[:, [0, 1, 2, 3, 4, 5, 6, 8]].sum(1)
            clk = (tot / r[:, 7]).median().item() * 100.0
            print(f'{name:18s} {which:6s} blocks={r.shape[0]:6d} wall={e0.elapsed_time(e1) * 1e3:7.1f} us  block cycles mean={tot.mean():9.0f} (= {tot.mean() / clk:6.1f} us @ {clk:5.0f} MHz)', flush=True)
            print('     mean ' + '  '.join(f'{n}={r[:, i].mean():8.0f}' for i, n in NAMES.items()), flush=True)


if __name__ == '__main__':
    main()
