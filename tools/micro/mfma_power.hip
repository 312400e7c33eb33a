// What the matrix pipe SUSTAINS: v_mfma_f32_32x32x16_f16 back to back on every SIMD of the chip for several
// milliseconds, wall-clock TFLOP/s, by operand content.  The issue-rate probes (issue_rate.hip, mfma_rate.hip)
// count cycles; this one counts seconds: the chip clocks to its power budget (MI355X_MICROARCH.md "DVFS give-back"),
// and a dense stream of MFMAs on random operands is the most power-hungry thing it can run.
//   operands: 0 = zeros, 1 = small integers (as mfma_rate.hip), 2 = random fp16 in [-1, 1) (8 operand sets rotating,
//   as a convolution's fragments do), 3 = random, two fp16 pieces of an fp32 value (h = 11 bits, m = the residual:
//   the fp16 x 2 arithmetic's actual operand statistics)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o tools/micro/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(const h16x8* __restrict__ ops, float* out, int iters) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  h16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = ops[(i * 2) * 256 + threadIdx.x];
    b[i] = ops[(i * 2 + 1) * 256 + threadIdx.x];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 24 / NACC; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u * NACC + j) & 7], b[(u + j) & 7], acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  h16x8* ops; hipMalloc(&ops, 16 * 256 * sizeof(h16x8));
  const char* names[4] = {"zeros", "small integers", "random fp16", "random fp16 x 2 pieces"};
  for (int mode = 0; mode < 4; ++mode) {
    std::vector<_Float16> h(16 * 256 * 8);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) {
      const float r = (float)rand() / RAND_MAX * 2.f - 1.f;
      float v = 0.f;
      if (mode == 1) v = (float)(i % 13);
      if (mode == 2) v = r;
      if (mode == 3) { const _Float16 hh = (_Float16)(r * 8192.f); v = (i & 8) ? (float)hh : (r * 8192.f - (float)hh); }
      h[i] = (_Float16)v;
    }
    hipMemcpy(ops, h.data(), h.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    for (int w : {1, 2}) {
      const int iters = 20000, grid = 256 * w;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, ops, d, 2000);     // (warm: clocks settle under load)
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, ops, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double mfma = (double)grid * 4 * iters * 24;
      const double tf = mfma * 32 * 32 * 16 * 2 / (ms * 1e-3) / 1e12;
      printf("%-24s waves/SIMD %d : %7.3f ms  %5.0f TFLOP/s = %.2f of the 2516.6 nominal  (%.2f GHz if the pipe never idles)\n",
             names[mode], w, ms, tf, tf / 2516.6, 2.4 * tf / 2516.6);
    }
  }
  return 0;
}
