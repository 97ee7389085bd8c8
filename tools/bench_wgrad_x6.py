"""Split-bf16 x6 weight-grad kernel (conv_wgrad_x6.hip) vs the exact-fp32 MFMA path (knob 5) on FiLM's cond_var.2 shapes: accuracy
against float64 at a small shape, time at the step's launch shapes on rotating operand sets (diagnostic)."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module('td-vc-gan_amd'); ops, L, arena = pkg.ops, pkg._lib, pkg.arena
lib = L.lib(); dev = torch.device('cuda:0')

def run(spec, x, dy, dw, db):
    dw.zero_(); db.zero_()
    ops.conv_wgrad_raw(spec, x, ops._xf(L.XF_LRELU), dy, ops._xf())

def make(C2, T, B, nset=1):
    cin = 136
    spec = ops.ConvSpec(cin, C2, 3, pad=1)
    w = torch.randn(C2, cin, 3, device=dev); b = torch.zeros(C2, device=dev)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    spec.slot = arena.ConvSlot(w.data_ptr(), b.data_ptr(), dw.data_ptr(), db.data_ptr(), True, None, 0)
    sets = [(torch.randn(B, cin, T, device=dev), torch.randn(B, C2, T, device=dev)) for _ in range(nset)]
    return spec, sets, dw, db, (w, b)

# ---- accuracy
torch.manual_seed(0)
for C2, T, B in ((32, 500, 3), (64, 2048, 2)):
    spec, sets, dw, db, keep = make(C2, T, B)
    x, dy = sets[0]
    xr = torch.nn.functional.leaky_relu(x.double().cpu(), 0.2)
    ref = torch.stack([torch.einsum('bot,bct->oc', dy.double().cpu()[:, :, max(0, 1 - j):T - max(0, j - 1)], xr[:, :, max(0, j - 1):T - max(0, 1 - j)]) for j in range(3)], dim=2)
    refb = dy.double().cpu().sum((0, 2))
    for knob in (0, 1):
        lib.tdvc_debug_knob(5, knob)
        with_tr = pkg._lib
        lib.tdvc_debug_trace(1); run(spec, x, dy, dw, db); torch.cuda.synchronize(); names = sorted(L.traced_kernels()); lib.tdvc_debug_trace(0)
        e = float((dw.double().cpu() - ref).norm() / ref.norm()); eb = float((db.double().cpu() - refb).norm() / refb.norm())
        print(f'C2={C2} T={T} B={B} {"fp32 " if knob else "bf16x6"}: rel-L2 dW {e:.2e}  db {eb:.2e}  {[n for n in names if "wgrad" in n]}', flush=True)
    lib.tdvc_debug_knob(5, 0)

# ---- time
def timeit(calls, iters=30):
    n = len(calls)
    for i in range(n + 2): calls[i % n]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters): calls[i % n]()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
for C2, T in ((32, 16000), (64, 8000), (128, 4000), (256, 500)):
    B = 32
    per = 4 * B * T * (136 + C2)
    spec, sets, dw, db, keep = make(C2, T, B, nset=max(2, min(12, int(600e6 // per) + 1)))
    res = {}
    for knob in (0, 1):
        lib.tdvc_debug_knob(5, knob)
        res[knob] = timeit([lambda s=s: ops.conv_wgrad_raw(spec, s[0], ops._xf(L.XF_LRELU), s[1], ops._xf()) for s in sets])
    lib.tdvc_debug_knob(5, 0)
    fl = 2.0 * B * T * 136 * C2 * 3
    print(f'C2={C2:4d} T={T:6d}: bf16x6 {res[0]:7.1f} us ({fl / res[0] / 1e6:6.1f} TF fp32-equivalent, {per / res[0] / 1e3:5.0f} GB/s)   fp32 MFMA {res[1]:7.1f} us   x{res[1] / res[0]:.2f}', flush=True)
    del sets; torch.cuda.empty_cache()
