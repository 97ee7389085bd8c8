"""Sizing probe for DESIGN.md §8.1 (not part of the product): accuracy and library-GEMM speed of the split-bf16 scheme on the
D-layer-5 weight-grad / forward GEMM shape (M = 1024, N = 64 x 63, K = 5120), against the fp32 GEMM.
x = hi + mid + lo (bf16 each); 3 products = hi.hi + hi.mid + mid.hi; 6 products add hi.lo + lo.hi + mid.mid; fp32 accumulation.

    python tools/split_bf16_probe.py
"""
import torch

dev = torch.device('cuda:0')
M, N, K = 1024, 64 * 63, 5120
g = torch.Generator(device='cpu').manual_seed(0)
a = (torch.randn(M, K, generator=g) / K ** 0.5).to(dev)
b = torch.randn(K, N, generator=g).to(dev)
ref = a.double() @ b.double()


def split3(x):
    hi = x.to(torch.bfloat16)
    r1 = x - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def mm32(x, y):      # bf16 x bf16 -> fp32 accumulate and fp32 output
    try:
        return torch.mm(x, y, out_dtype=torch.float32)
    except TypeError:
        return (x.float() @ y.float())      # fallback (numerically the same products, not the bf16 pipe)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def rel(x):
    return float((x.double() - ref).norm() / ref.norm())


ah, am, al = split3(a)
bh, bm, bl = split3(b)
c32 = a @ b
c1 = mm32(ah, bh)
c3 = c1 + mm32(ah, bm) + mm32(am, bh)
c6 = c3 + mm32(ah, bl) + mm32(al, bh) + mm32(am, bm)
print(f'shape {M} x {N} x {K}: {2.0 * M * N * K / 1e9:.1f} GFLOP')
print(f'rel-L2 vs float64:  fp32 GEMM {rel(c32):.2e} | bf16 x1 {rel(c1):.2e} | split x3 {rel(c3):.2e} | split x6 {rel(c6):.2e}')
t32 = timeit(lambda: a @ b)
t16 = timeit(lambda: mm32(ah, bh))
tsp = timeit(lambda: split3(b))
print(f'library GEMM time: fp32 {t32:.0f} us ({2.0 * M * N * K / t32 / 1e6:.0f} TFLOP/s) | one bf16->fp32 product {t16:.0f} us '
      f'({2.0 * M * N * K / t16 / 1e6:.0f} TFLOP/s) | splitting the activation operand {tsp:.0f} us')
print(f'-> 3 products {3 * t16:.0f} us, 6 products {6 * t16:.0f} us (+ split) against {t32:.0f} us fp32; tdvc lean kernel on this layer: ~392 us')
