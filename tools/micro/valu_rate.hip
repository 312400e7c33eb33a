// VALU issue cost on gfx950, one and two waves per SIMD: cycles per wave-instruction of the
// operations the Winograd input transform is made of (bf16 split, fp32 add), alone and
// interleaved with MFMAs.  s_memtime around an unrolled stream of independent ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  f32x4 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x4{(float)threadIdx.x + i, 1.5f * i, 0.25f, (float)i};
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {                       // fp32 adds: 4 per f32x4
        v[u] = v[u] + v[(u + 1) & 7];
      } else if (MODE == 1) {                // the 3-way split of one f32x4 (22 VALU expected)
        bf16x4 h = __builtin_convertvector(v[u], bf16x4);
        f32x4 r1 = v[u] - __builtin_convertvector(h, f32x4);
        bf16x4 m = __builtin_convertvector(r1, bf16x4);
        f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);
        bf16x4 l = __builtin_convertvector(r2, bf16x4);
        v[u] = __builtin_convertvector(h, f32x4) + __builtin_convertvector(m, f32x4) * 3.f + __builtin_convertvector(l, f32x4) * 5.f + r2;
      } else if (MODE == 2) {                // one MFMA + the split (interleaving)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        bf16x4 h = __builtin_convertvector(v[u], bf16x4);
        f32x4 r1 = v[u] - __builtin_convertvector(h, f32x4);
        bf16x4 m = __builtin_convertvector(r1, bf16x4);
        f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);
        v[u] = r2 + __builtin_convertvector(m, f32x4);
      } else {                               // MFMA only
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, float* d, long long* c) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, c, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, c, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-28s waves/SIMD %d : %.3f ms, %.1f clk-counter ticks per loop body of 8 (%.2f us per 1000 bodies)\n",
         name, blocks_per_cu, ms, (double)h / iters, ms * 1e3 / iters * 1000 / 1000);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  long long* c; hipMalloc(&c, 64);
  for (int w : {1, 2}) {
    run<0>("8 x f32x4 add (32 VALU)", w, d, c);
    run<1>("8 x split3 (+7 fma)", w, d, c);
    run<2>("8 x (MFMA + 2-piece split)", w, d, c);
    run<3>("8 x MFMA", w, d, c);
  }
  return 0;
}
