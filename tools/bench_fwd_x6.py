"""Split-bf16 x6 forward (conv_fwd_x6.hip) vs the exact-fp32 MFMA kernel on FiLM's cond_var.2 shapes: accuracy vs float64 at a small
shape, time at the step's launch shapes on rotating operand sets (diagnostic)."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module('td-vc-gan_amd'); ops, L, arena = pkg.ops, pkg._lib, pkg.arena
lib = L.lib(); dev = torch.device('cuda:0'); ops.X6_FWD_MIN_COUT = 32
def make(C2, T, B, nset=1, cin=136):
    spec = ops.ConvSpec(cin, C2, 3, pad=1)
    w = torch.randn(C2, cin, 3, device=dev) / (cin * 3) ** 0.5; b = torch.randn(C2, device=dev) * 0.1
    spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), 0, 0, True, None, 0)
    return spec, [(torch.randn(B, cin, T, device=dev), torch.empty(B, C2, T, device=dev)) for _ in range(nset)], (w, b)
torch.manual_seed(0)
for C2, T, B, cin in ((32, 500, 3, 136), (64, 2048, 2, 136), (96, 260, 2, 72)):
    spec, sets, (w, b) = make(C2, T, B, cin=cin)
    x, y = sets[0]
    ref = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(x.double().cpu(), 0.2), w.double().cpu(), b.double().cpu(), padding=1)
    for on in (True, False):
        ops.X6_FWD = on
        lib.tdvc_debug_trace(1); ops.conv_fwd_raw(spec, x, ops._xf(L.XF_LRELU), out=y); torch.cuda.synchronize(); names = sorted(L.traced_kernels()); lib.tdvc_debug_trace(0)
        print(f'C2={C2} T={T} B={B} cin={cin} {"bf16x6" if on else "fp32  "}: rel-L2 {float((y.double().cpu() - ref).norm() / ref.norm()):.2e}  {names}', flush=True)
ops.X6_FWD = True
def timeit(calls, iters=30):
    n = len(calls)
    for i in range(n + 2): calls[i % n]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters): calls[i % n]()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
for C2, T in ((32, 16000), (64, 8000), (128, 4000), (256, 500)):
    B = 32; per = 4 * B * T * (136 + C2)
    spec, sets, keep = make(C2, T, B, nset=max(2, min(12, int(600e6 // per) + 1)))
    res = {}
    for on in (True, False):
        ops.X6_FWD = on
        res[on] = timeit([lambda s=s: ops.conv_fwd_raw(spec, s[0], ops._xf(L.XF_LRELU), out=s[1]) for s in sets])
    ops.X6_FWD = True
    fl = 2.0 * B * T * 136 * C2 * 3
    print(f'C2={C2:4d} T={T:6d}: bf16x6 {res[True]:7.1f} us ({fl / res[True] / 1e6:6.1f} TF fp32-equivalent, {per / res[True] / 1e3:5.0f} GB/s)   fp32 MFMA {res[False]:7.1f} us   x{res[False] / res[True]:.2f}', flush=True)
    del sets; torch.cuda.empty_cache()
