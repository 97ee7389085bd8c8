"""Conditioning forward: one launch (tdvc_film_cond_fwd_x6) vs two (cond_var.0 window producer + split-bf16 cond_var.2 forward), at the
step's launch shapes on rotating operand sets (diagnostic)."""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module('td-vc-gan_amd'); ops, L, arena = pkg.ops, pkg._lib, pkg.arena
lib = L.lib(); dev = torch.device('cuda:0'); ops.FUSED_COND_FWD_X6_ALWAYS = True
def timeit(calls, iters=30):
    n = len(calls)
    for i in range(n + 2): calls[i % n]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters): calls[i % n]()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
nc, nv, B = 136, 8, 32
for C2, T in ((32, 16000), (64, 8000), (128, 4000), (256, 500)):
    w0 = torch.randn(nc, nc, 3, device=dev) / 20; w2 = torch.randn(C2, nc, 3, device=dev) / 20; b2 = torch.randn(C2, device=dev)
    sv = ops.ConvSpec(nv, nc, 3, 1, 1, 1, 1, False, w_cin=nc, w_cin_off=nc - nv); sv.slot = arena.ConvSlot(w0.data_ptr(), 0, 0, 0, False, None, 0)
    s2 = ops.ConvSpec(nc, C2, 3, 1, 1, 1, 1, False); s2.slot = arena.ConvSlot(w2.data_ptr(), b2.data_ptr(), 0, 0, False, None, 0)
    per = 4 * B * T * (nc + C2 + nv)
    nset = max(2, min(12, int(600e6 // per) + 1))
    sets = [(torch.randn(B, nv, T, device=dev), torch.randn(B, nc, 3, device=dev)) for _ in range(nset)]
    res = {}
    with torch.no_grad():
        for fused in (True, False):
            ops.FUSED_COND_FWD_X6 = fused
            lib.tdvc_debug_trace(1); ops.film_cond(sets[0][0], sets[0][1], sv, s2); torch.cuda.synchronize(); names = sorted(L.traced_kernels()); lib.tdvc_debug_trace(0)
            res[fused] = (timeit([lambda s=s: ops.film_cond(s[0], s[1], sv, s2) for s in sets]), names)
    ops.FUSED_COND_FWD_X6 = True
    print(f'C2={C2:4d} T={T:6d}: one launch {res[True][0]:7.1f} us {res[True][1]}   two launches {res[False][0]:7.1f} us {res[False][1]}   x{res[False][0] / res[True][0]:.2f}', flush=True)
    del sets; torch.cuda.empty_cache()
