// MFMA issue-rate probe: v_mfma_f32_32x32x16_bf16 in (a) one dependent chain per wave,
// (b) two alternating accumulators, (c) four; with 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 24 / NACC; ++u)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)grid * 4 * iters * 24;
  const double tf = mfma * 32 * 32 * 16 * 2 / (ms * 1e-3) / 1e12;
  printf("NACC %d  waves/SIMD %d : %.3f ms  %.0f TFLOP/s bf16 (%.1f%% of 2516)\n", NACC, blocks_per_cu, ms, tf, 100 * tf / 2516.6);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4}) { run<1>(w, d); run<2>(w, d); run<4>(w, d); }
  return 0;
}
