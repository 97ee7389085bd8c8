"""Phase-cycle breakdown of the fused conditioning backward kernel (diagnostic; needs `make -C td-vc-gan_amd/csrc prof`).
Per block (thread 0 = wave 0) the instrumented build accumulates s_memtime deltas per phase."""
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
L.SIGNATURES['tdvc_debug_condbwd_prof'] = (C.c_int, [C.c_void_p])
NAMES = ['bar_top', 'commit', 'bar_staged', 'issue', 'mfma(p)', 'mask', 'tile->lds', 'realtime', '(b)', '(a)+tail']
dev = torch.device('cuda:0')
lib = L.lib()
B, nc, nv = 32, 136, 8
buf = torch.zeros(10 * 4096, dtype=torch.int64, device=dev)
if len(sys.argv) > 1:
    lib.tdvc_debug_knob(3, int(sys.argv[1]))      # 1: one block per CU (uncontended per-wave phase times)
for C2, T in ((32, 16000), (128, 4000)):
    w0 = torch.randn(nc, nc, 3, device=dev) / (nc * 3) ** 0.5
    wt2 = torch.randn(nc, C2, 3, device=dev) / (nc * 3) ** 0.5
    dw0 = torch.zeros_like(w0)
    dgb, exc = torch.randn(B, C2, T, device=dev), torch.randn(B, nv, T, device=dev)
    dexc, dk3 = torch.empty_like(exc), torch.empty(B, nc, 3, device=dev)
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (B, nc, T // 32), dtype=torch.int32, device=dev)
    ws = torch.empty(lib.tdvc_film_cond_bwd_workspace(B, T, nc, nv), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    a = L.FilmCondBwdArgs(B, T, nc, nv, C2, dgb.data_ptr(), dgb.stride(0), wt2.data_ptr(), bits.data_ptr(), bits.stride(0), None, 0,
                          exc.data_ptr(), exc.stride(0), w0.data_ptr(), dexc.data_ptr(), dexc.stride(0), dk3.data_ptr(), dw0.data_ptr(),
                          ws.data_ptr(), ws.numel(), 0.2)
    for _ in range(3):
        L.check(lib.tdvc_film_cond_bwd(C.byref(a), st))
    torch.cuda.synchronize()
    buf.zero_()
    lib.tdvc_debug_condbwd_prof(buf.data_ptr())
    L.check(lib.tdvc_film_cond_bwd(C.byref(a), st))
    torch.cuda.synchronize()
    lib.tdvc_debug_condbwd_prof(None)
    t = buf.view(-1, 10).cpu().double()
    t = t[t[:, 4] > 0]
    tot = t[:, [0, 1, 2, 3, 4, 5, 6, 8, 9]].sum(1)
    print(f'C2={C2} T={T}: {t.shape[0]} blocks, cycles/block median {tot.median():.0f}, realtime {t[:, 7].median() / 100:.1f} us -> {tot.median() / (t[:, 7].median() / 100) / 1e3:.2f} GHz')
    for i, n in enumerate(NAMES):
        if i != 7:
            print(f'   {n:12s} {t[:, i].median():10.0f}  {100 * t[:, i].median() / tot.median():5.1f} %')
