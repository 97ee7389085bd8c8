"""Tuning sweep (diagnostic): time the FiLM conditioning convs at their launch shapes under every lean tile configuration
and several LDS budgets per block (tdvc_debug_force_tile / tdvc_debug_lds_cap), rotating operand sets like bench.py."""
import ctypes as C
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
BL = 32


def conv_calls(cin, cout, T, which, pre=1, k=3, BL=BL):
    spec = ops.ConvSpec(cin, cout, k, 1, (k - 1) // 2, 1, 1, False)
    w = torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    wt = w.permute(1, 0, 2).contiguous()
    spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), 0, 0, False, None, wt.data_ptr())
    if which == 'fwd':
        bufs = bench.Bufs(torch, dev, dict(x=(BL, cin, T), y=(BL, cout, T)))
        calls = [lambda s=s: ops.conv_fwd_raw(spec, s['x'], ops._xf(L.XF_LRELU), out=s['y']) for s in bufs.sets]
    else:
        bufs = bench.Bufs(torch, dev, dict(dy=(BL, cout, T), dx=(BL, cin, T), x_in=(BL, cin, T)))
        calls = [lambda s=s: ops.conv_dgrad_raw(spec, s['dy'], ops._xf(), T, L.DG_MASK_LRELU, x_in=s['x_in'], out=s['dx']) for s in bufs.sets]
    return calls, (w, b, wt, bufs, spec)


def cond_calls(C2, T):
    nc, nv = 136, 8
    w0 = torch.randn(nc, nc, 3, device=dev) / (nc * 3) ** 0.5
    w2 = torch.randn(C2, nc, 3, device=dev) / (nc * 3) ** 0.5
    b2 = torch.randn(C2, device=dev) * 0.1
    bufs = bench.Bufs(torch, dev, dict(exc=(BL, nv, T), k3=(BL, nc, 3), cv0=(BL, nc, T), gb=(BL, C2, T)))
    st = torch.cuda.current_stream(dev).cuda_stream
    calls, keep = [], [w0, w2, b2, bufs]
    for s in bufs.sets:
        a = L.FilmCondArgs(BL, T, nc, nv, C2, s['exc'].data_ptr(), s['exc'].stride(0), w0.data_ptr(), s['k3'].data_ptr(), w2.data_ptr(),
                           b2.data_ptr(), s['cv0'].data_ptr(), s['cv0'].stride(0), s['gb'].data_ptr(), s['gb'].stride(0), 0.2)
        keep.append(a)
        calls.append(lambda a=a: L.check(lib.tdvc_film_cond_fwd(C.byref(a), st)))
    return calls, keep


def main():
    stages = [(32, 16000), (64, 8000), (128, 4000), (256, 500)]
    jobs = [(136, C2, T, w, 3, BL) for C2, T in stages for w in ('fwd', 'dgrad', 'cond')]
    if len(sys.argv) > 1 and sys.argv[1] == 'trunk':
        jobs = [(1024, 1024, 63, 'fwd', 5, 64), (1024, 1024, 63, 'dgrad', 5, 64), (1024, 1024, 32, 'fwd', 5, 64),
                (64, 64, 4000, 'fwd', 7, 32), (64, 64, 4000, 'dgrad', 7, 32), (128, 128, 500, 'fwd', 7, 32), (128, 128, 500, 'dgrad', 7, 32),
                (32, 32, 8000, 'fwd', 7, 32), (32, 32, 8000, 'dgrad', 7, 32), (32, 32, 8000, 'fwd', 1, 32), (64, 64, 4000, 'fwd', 1, 32),
                (8, 136, 16000, 'fwd', 3, 32), (8, 136, 4000, 'fwd', 3, 32), (256, 256, 50, 'fwd', 7, 32)]
    caps, cfgs = (0, 52 * 1024, 40 * 1024), (-1, 0, 1, 2, 4, 5, 6, 7)
    if len(sys.argv) > 1 and sys.argv[1] == 'tall':      # the 136-row problems: 16- / 48-row tiles vs the 144 x 64 one
        jobs = [(136, C2, T, 'dgrad', 3, BL) for C2, T in stages] + [(8, 136, T, 'fwd', 3, BL) for _, T in stages]
        caps, cfgs = (0,), (-1, 0, 5, 7)
    if len(sys.argv) > 1 and sys.argv[1] == 'd5':        # D layer 5 and the short-sequence trunk convs on the 64 x 64 tile
        jobs = [(1024, 1024, 63, 'fwd', 5, 64), (1024, 1024, 63, 'dgrad', 5, 64), (1024, 1024, 32, 'fwd', 5, 64), (1024, 1024, 32, 'dgrad', 5, 64),
                (128, 128, 500, 'fwd', 7, 32), (128, 128, 500, 'dgrad', 7, 32), (256, 256, 50, 'fwd', 7, 32), (136, 256, 500, 'dgrad', 3, 32)]
        caps, cfgs = (0, 52 * 1024), (-1, 4, 6)
    for cin, C2, T, which, k, bl in jobs:
        if True:
            calls, keep = cond_calls(C2, T) if which == 'cond' else conv_calls(cin, C2, T, which, k=k, BL=bl)
            res = []
            for cap in caps:
                for cfg in ((-1,) if which == 'cond' else cfgs):
                    lib.tdvc_debug_lds_cap(cap); lib.tdvc_debug_force_tile(cfg)
                    lib.tdvc_debug_trace(1)
                    try:
                        calls[0]()
                    except Exception as e:      # noqa: BLE001
                        res.append((cap, cfg, None, str(e)[:40]))
                        continue
                    torch.cuda.synchronize()
                    name = '+'.join(sorted(L.traced_kernels()))
                    lib.tdvc_debug_trace(0)
                    ms = bench.time_launches(torch, calls, 30)
                    res.append((cap, cfg, ms, name))
            lib.tdvc_debug_lds_cap(0); lib.tdvc_debug_force_tile(-1)
            print(f'== {cin}->{C2} k{k} T={T} B={bl} {which}', flush=True)
            for cap, cfg, ms, name in res:
                print(f'   cap={cap // 1024:3d}K tile={cfg:2d}  ' + (f'{ms * 1e3:8.1f} us  {name}' if ms is not None else f'FAILED {name}'), flush=True)
            del calls, keep
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
