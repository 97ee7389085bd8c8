"""Phase-cycle breakdown of the split-bf16 forward kernel (diagnostic; needs `make -C td-vc-gan_amd/csrc prof`)."""
import ctypes as C, importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module('td-vc-gan_amd'); ops, arena, L = pkg.ops, pkg.arena, pkg._lib
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), 'libtdvc_hip_prof.so')
L.SIGNATURES['tdvc_debug_fwdx6_prof'] = (C.c_int, [C.c_void_p])
NAMES = ['issue0', 'bar_top', 'wait loads', 'split+store', 'bar_staged', 'issue', 'mfma', 'realtime', 'epilogue', '-']
dev = torch.device('cuda:0'); lib = L.lib(); ops.X6_FWD_MIN_COUT = 32
buf = torch.zeros(10 * 8192, dtype=torch.int64, device=dev)
for C2, T in ((32, 16000), (64, 8000), (128, 4000)):
    B = 32
    spec = ops.ConvSpec(136, C2, 3, pad=1)
    w = torch.randn(C2, 136, 3, device=dev) / 20; b = torch.randn(C2, device=dev)
    spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), 0, 0, True, None, 0)
    x, y = torch.randn(B, 136, T, device=dev), torch.empty(B, C2, T, device=dev)
    for _ in range(3): ops.conv_fwd_raw(spec, x, ops._xf(L.XF_LRELU), out=y)
    torch.cuda.synchronize(); buf.zero_()
    lib.tdvc_debug_fwdx6_prof(buf.data_ptr())
    ops.conv_fwd_raw(spec, x, ops._xf(L.XF_LRELU), out=y); torch.cuda.synchronize()
    lib.tdvc_debug_fwdx6_prof(None)
    t = buf.view(-1, 10).cpu().double(); t = t[t[:, 6] > 0]
    idx = [0, 1, 2, 3, 4, 5, 6, 8]
    tot = t[:, idx].sum(1)
    print(f'C2={C2} T={T}: {t.shape[0]} blocks, cycles/block median {tot.median():.0f}, realtime {t[:, 7].median() / 100:.1f} us')
    for i in idx: print(f'   {NAMES[i]:12s} {t[:, i].median():10.0f}  {100 * t[:, i].median() / tot.median():5.1f} %')
