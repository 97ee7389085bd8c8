import importlib, sys, os, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
pkg = importlib.import_module('td-vc-gan_amd'); ops, L, arena = pkg.ops, pkg._lib, pkg.arena
lib = L.lib(); dev = torch.device('cuda:0')
def timeit(f, n=30):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for T, B in ((16, 64), (32, 64), (16, 32), (63, 32)):
    spec = ops.ConvSpec(1024, 1024, 5, 1, 2, 1, 1, False)
    w = torch.randn(1024, 1024, 5, device=dev) / 72; b = torch.randn(1024, device=dev) * 0.1
    wt = w.permute(1, 0, 2).contiguous()
    spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), 0, 0, True, None, wt.data_ptr())
    x = torch.randn(B, 1024, T, device=dev); y = torch.empty(B, 1024, T, device=dev); dy = torch.randn_like(y); dx = torch.empty_like(x)
    for knob in (0, 1):
        lib.tdvc_debug_knob(4, knob)
        tf = timeit(lambda: ops.conv_fwd_raw(spec, x, ops._xf(), post=1, out=y))
        td = timeit(lambda: ops.conv_dgrad_raw(spec, dy, ops._xf(L.XF_MASK_LRELU, aux=y), T, L.DG_PLAIN, out=dx))
        print(f'T={T} B={B} fold={"off" if knob else "on "}: fwd {tf:7.1f} us  dgrad {td:7.1f} us', flush=True)
    lib.tdvc_debug_knob(4, 0)
