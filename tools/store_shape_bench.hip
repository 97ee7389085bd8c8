// Diagnostic (not part of the product): HBM write rate of a [B][C][T] fp32 tensor for different shapes of one wave-level
// dwordx4 store instruction. A: 16 rows x 64 B (the lean conv epilogue: lane = (channel ln, 4 consecutive t at kq*4));
// B: 4 rows x 256 B; C: 1 row x 1 KB. Same bytes, same number of store instructions, same grid.
//   hipcc --offload-arch=gfx950 -O3 tools/store_shape_bench.hip -o /tmp/ssb && /tmp/ssb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// block = 256 threads = 4 waves; tile = 16 channels x 256 t (one wave: 16 ch x 64 t = 4 store instructions)
template <int SHAPE, bool READ>
__global__ __launch_bounds__(256) void k(float* __restrict__ y, const float* __restrict__ x, int C, int T, int ntx) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 15, kq = lane >> 4;
  const int tile = blockIdx.x, b = blockIdx.z, c0 = blockIdx.y * 16;
  const int n0 = tile * 256 + wave * 64;
  const long base = ((long)b * C + c0) * T;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c, t;
    if (SHAPE == 0) { c = ln; t = n0 + i * 16 + kq * 4; }              // 16 rows x 64 B
    else if (SHAPE == 1) { c = i * 4 + kq; t = n0 + ln * 4; }          // 4 rows x 256 B
    else { c = i * 4 + wave; t = tile * 256 + lane * 4; }              // 1 row x 1 KB (wave = row within the group of 4)
    if (c0 + c < C && t < T) {
      f32x4 v = {1.f * c, 2.f, 3.f, 4.f * t};
      if (READ) { const f32x4 r = *reinterpret_cast<const f32x4*>(x + base + (long)c * T + t); v += r; }
      *reinterpret_cast<f32x4*>(y + base + (long)c * T + t) = v;
    }
  }
}

template <int SHAPE, bool READ>
float run(float** ys, float** xs, int nb, int B, int C, int T, int iters) {
  const int ntx = (T + 255) / 256;
  dim3 grid(ntx, (C + 15) / 16, B);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < nb; ++i) hipLaunchKernelGGL((k<SHAPE, READ>), grid, dim3(256), 0, 0, ys[i], xs[i], C, T, ntx);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k<SHAPE, READ>), grid, dim3(256), 0, 0, ys[i % nb], xs[i % nb], C, T, ntx);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters * 1e3f;
}

int main() {
  const int cases[3][3] = {{32, 136, 16000}, {32, 16, 16000}, {32, 32, 8000}};
  for (auto& cs : cases) {
    const int B = cs[0], C = cs[1], T = cs[2];
    const size_t bytes = (size_t)B * C * T * 4;
    const int nb = (int)(700e6 / bytes) + 2;       // rotate: nothing stays in the 256 MB Infinity Cache
    std::vector<float*> ys(nb), xs(nb);
    for (int i = 0; i < nb; ++i) { hipMalloc(&ys[i], bytes); hipMalloc(&xs[i], bytes); hipMemset(xs[i], 0, bytes); }
    const char* names[3] = {"16 rows x 64 B ", "4 rows x 256 B ", "1 row x 1 KB   "};
    for (int rep = 0; rep < 2; ++rep) {
      float w[3] = {run<0, false>(ys.data(), xs.data(), nb, B, C, T, 40), run<1, false>(ys.data(), xs.data(), nb, B, C, T, 40), run<2, false>(ys.data(), xs.data(), nb, B, C, T, 40)};
      float rw[3] = {run<0, true>(ys.data(), xs.data(), nb, B, C, T, 40), run<1, true>(ys.data(), xs.data(), nb, B, C, T, 40), run<2, true>(ys.data(), xs.data(), nb, B, C, T, 40)};
      for (int s = 0; s < 3; ++s)
        printf("[%d x %d x %d] %s write-only %7.1f us %5.2f TB/s | read+write %7.1f us %5.2f TB/s\n", B, C, T, names[s], w[s], bytes / w[s] / 1e6,
               rw[s], 2.0 * bytes / rw[s] / 1e6);
    }
    for (int i = 0; i < nb; ++i) { hipFree(ys[i]); hipFree(xs[i]); }
  }
  return 0;
}
