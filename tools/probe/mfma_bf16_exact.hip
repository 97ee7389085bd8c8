// Is v_mfma_f32_16x16x32_bf16 exact for bf16 inputs (exact products, fp32 accumulation)? One wave, one instruction, checked on the host.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* A, const unsigned short* B, float* D, int reps) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  u16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = A[r * 32 + 8 * g + e]; b[e] = B[(8 * g + e) * 16 + r]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < reps; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  for (int q = 0; q < 4; ++q) D[(4 * g + q) * 16 + r] = c[q];
}
static unsigned short bf(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
static float fb(unsigned short h) { unsigned u = (unsigned)h << 16; float x; memcpy(&x, &u, 4); return x; }
int main() {
  unsigned short hA[16 * 32], hB[32 * 16]; float hD[256];
  unsigned short *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  for (int mode = 0; mode < 4; ++mode) {
    srand(1 + mode);
    for (int i = 0; i < 512; ++i) {
      float a = 1.0f + (rand() % 128) / 128.0f;                 // 8 significant bits in [1, 2)
      if (mode >= 1) a *= (rand() & 1) ? 1.f : -1.f;            // sign mix
      if (mode >= 2) a *= ldexpf(1.f, rand() % 8);              // exponent mix
      float b = (mode == 3) ? 1.0f + (rand() % 128) / 128.0f : (float)(1 + rand() % 3);
      hA[i] = bf(a); hB[i] = bf(b);
    }
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    for (int reps = 1; reps <= 8; reps *= 8) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, reps);
      hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
      double maxrel = 0, maxabs = 0;
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int kk = 0; kk < 32; ++kk) s += (double)fb(hA[i * 32 + kk]) * (double)fb(hB[kk * 16 + j]);
        s *= reps;
        const double e = fabs(hD[i * 16 + j] - s); if (e > maxabs) maxabs = e; if (fabs(s) > 1e-9 && e / fabs(s) > maxrel) maxrel = e / fabs(s);
      }
      printf("mode %d reps %d: max abs err %.3e max rel err %.3e\n", mode, reps, maxabs, maxrel);
    }
  }
  return 0;
}
