"""Per-shape timing of the strided / grouped / transposed convs of the step (the generic conv_gemm / conv_wgrad kernels):
forward, input-grad and weight-grad at the launch shapes, rotating operand sets, with algorithmic bytes and FLOPs.

    python tools/generic_table.py
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

pkg = importlib.import_module('td-vc-gan_amd')
ops, arena, L = pkg.ops, pkg.arena, pkg._lib
lib = L.lib()
dev = torch.device('cuda:0')

# (name, cin, cout, k, stride, pad, dil, groups, reflect, transposed, T, pre, post), B   -- tests/test_kernel_instances_gpu.py LAUNCH_CONV
CASES = [
    (('d0_1_16_k15', 1, 16, 15, 1, 7, 1, 1, True, False, 16000, 0, 1), 32),
    (('d1_grp_16_64', 16, 64, 41, 4, 20, 1, 4, False, False, 16000, 0, 1), 32),
    (('d2_grp_64_256', 64, 256, 41, 4, 20, 1, 16, False, False, 4000, 0, 1), 32),
    (('d3_grp_256_1024', 256, 1024, 41, 4, 20, 1, 64, False, False, 1000, 0, 1), 32),
    (('d4_grp_1024_1024', 1024, 1024, 41, 4, 20, 1, 256, False, False, 250, 0, 1), 32),
    (('dout_1024_16_k3_T63', 1024, 16, 3, 1, 1, 1, 1, False, False, 63, 0, 0), 64),
    (('enc0_1_16_k7', 1, 16, 7, 1, 3, 1, 1, True, False, 16000, 0, 0), 32),
    (('down_16_32_s2', 16, 32, 4, 2, 1, 1, 1, False, False, 16000, 1, 0), 32),
    (('down_32_64_s2', 32, 64, 4, 2, 1, 1, 1, False, False, 8000, 1, 0), 32),
    (('down_64_128_s8', 64, 128, 16, 8, 4, 1, 1, False, False, 4000, 1, 0), 32),
    (('down_128_256_s10', 128, 256, 20, 10, 5, 1, 1, False, False, 500, 1, 0), 32),
    (('up_256_128_s10', 256, 128, 20, 10, 5, 1, 1, False, True, 50, 1, 0), 32),
    (('up_128_64_s8', 128, 64, 16, 8, 4, 1, 1, False, True, 500, 1, 0), 32),
    (('up_64_32_s2', 64, 32, 4, 2, 1, 1, 1, False, True, 4000, 1, 0), 32),
    (('up_32_16_s2', 32, 16, 4, 2, 1, 1, 1, False, True, 8000, 1, 0), 32),
    (('head_16_1_tanh', 16, 1, 7, 1, 3, 1, 1, True, False, 16000, 1, 2), 32),
    (('exc_in_1_8_k7', 1, 8, 7, 1, 3, 1, 1, True, False, 16000, 0, 0), 32),
    (('exc_8_8_k5', 8, 8, 5, 1, 2, 1, 1, False, False, 8000, 1, 0), 32),
    (('exc_down_8_8_s2', 8, 8, 4, 2, 1, 1, 1, False, False, 16000, 0, 0), 32),
    (('fir_dw8_k33_s2', 8, 8, 33, 2, 16, 1, 8, False, False, 16000, 0, 0), 32),
]


def traced_time(calls, iters=30):
    lib.tdvc_debug_trace(1)
    calls[0]()
    torch.cuda.synchronize()
    names = '+'.join(sorted(L.traced_kernels()))
    lib.tdvc_debug_trace(0)
    return bench.time_launches(torch, calls, iters) * 1e3, names


def main():
    for (name, cin, cout, k, s, p, d, g, reflect, transposed, T, pre, post), B in CASES:
        spec = ops.ConvSpec(cin, cout, k, s, p, d, g, reflect, transposed)
        tout = spec.tout(T)
        wshape = (cin, cout // g, k) if transposed else (cout, cin // g, k)
        w = torch.randn(wshape, device=dev) / (wshape[1] * k) ** 0.5
        b = torch.randn(cout, device=dev) * 0.1
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        wt = w.permute(1, 0, 2).contiguous() if (not transposed and g == 1 and s == 1) else None
        spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), dw.data_ptr(), db.data_ptr(), True, None, wt.data_ptr() if wt is not None else 0)
        bufs = bench.Bufs(torch, dev, dict(x=(B, cin, T), y=(B, cout, tout), dx=(B, cin, T)))
        xf = ops._xf(L.XF_LRELU) if pre else ops._xf()
        fwd = [lambda q=q: ops.conv_fwd_raw(spec, q['x'], xf, post=post, out=q['y']) for q in bufs.sets]
        # input-grad: dy masked by the stored post-activation (post LeakyReLU) / plain, LeakyReLU mask epilogue when pre-activated
        dyxf = (lambda q: ops._xf(L.XF_MASK_LRELU, aux=q['y'])) if post == 1 else (lambda q: ops._xf())
        dg = [lambda q=q: ops.conv_dgrad_raw(spec, q['y'], dyxf(q), T, L.DG_MASK_LRELU if pre else L.DG_PLAIN, x_in=q['x'] if pre else None, out=q['dx'])
              for q in bufs.sets]
        wg = [lambda q=q: ops.conv_wgrad_raw(spec, q['x'], xf, q['y'], dyxf(q)) for q in bufs.sets]
        flops = 2.0 * B * tout * cout * (cin // g) * k
        by_f = 4.0 * B * (cin * T + cout * tout)
        print(f'== {name} B={B} T={T}->{tout}  {flops / 1e9:.2f} GF  {by_f / 1e6:.1f} MB', flush=True)
        for label, calls, by in (('fwd', fwd, by_f), ('dgrad', dg, by_f + (4.0 * B * cin * T if pre else 0) + (4.0 * B * cout * tout if post == 1 else 0)),
                                 ('wgrad', wg, by_f + (4.0 * B * cout * tout if post == 1 else 0))):
            try:
                us, names = traced_time(calls)
            except Exception as e:      # noqa: BLE001
                print(f'   {label:5s} FAILED {str(e)[:80]}', flush=True)
                continue
            print(f'   {label:5s} {us:8.1f} us  {by / us / 1e6:6.2f} TB/s  {flops / us / 1e6:6.1f} TF   {names}', flush=True)
        del fwd, dg, wg, bufs
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
