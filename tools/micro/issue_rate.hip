// Issue model of a gfx950 SIMD around v_mfma_f32_32x32x16_bf16: cycles per MFMA of a loop of
// [MFMA, NF filler instructions] with one or two waves per SIMD, by filler type.  Answers: how
// many non-MFMA instructions fit in the 32-cycle shadow of an MFMA, per wave and per SIMD.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/issue_rate.hip -o tools/micro/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int TYPE>
__device__ __forceinline__ void filler(unsigned& a, unsigned& b, f32x2& p, f32x2& q, int i, const float* lds) {
  if (TYPE == 0) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(a));
  else if (TYPE == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a) : "v"(p.x), "v"(p.y));
  else if (TYPE == 2) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p) : "v"(q));
  else if (TYPE == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(p.x) : "v"(q.x));
  else if (TYPE == 4) {                                  // the split stage mix, one of 5 per call
    const int m = i % 5;
    if (m == 0) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(b) : "v"(a));
    else if (m == 1) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(a) : "v"(b));
    else if (m == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(p.x) : "v"(q.x));
    else if (m == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(p.y) : "v"(q.y));
    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a) : "v"(p.x), "v"(p.y));
  } else if (TYPE == 5) asm volatile("s_nop 0");
  else if (TYPE == 6) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
  else if (TYPE == 7) asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(p.x) : "v"(a), "s"(0x0000BF80u));
  else if (TYPE == 9) asm volatile("v_dot2c_f32_bf16 %0, 0xbf800000, %1" : "+v"(p.x) : "v"(a));
  else if (TYPE == 10) asm volatile("v_dot2c_f32_bf16 %0, -1.0, %1" : "+v"(p.x) : "v"(a));
  else if (TYPE == 11) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a) : "v"(p.x), "v"(p.y));
  else if (TYPE == 12) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(p.x) : "v"(a));
  else if (TYPE == 13) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(p.x) : "v"(a), "v"(q.x));
  else if (TYPE == 14) {                                 // fp16 x 2 split of a pair, cvt + sub form: 5 per call
    const int m = i % 5;
    if (m == 0) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(q.x) : "v"(a));
    else if (m == 1) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(q.y) : "v"(a));
    else if (m == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(p.x) : "v"(q.x));
    else if (m == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(p.y) : "v"(q.y));
    else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a) : "v"(p.x), "v"(p.y));
  } else if (TYPE == 15) {                               // fp16 x 2 split of a pair, v_fma_mix form: 3 per call
    const int m = i % 3;
    if (m == 0) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(p.x) : "v"(a), "v"(q.x));
    else if (m == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(p.y) : "v"(a), "v"(q.x));
    else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a) : "v"(p.x), "v"(p.y));
  }
  else if (TYPE == 8) {                                  // the split stage with v_dot2c residuals, one of 3 per call
    const int m = i % 3;
    if (m == 0) asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(p.x) : "v"(a), "s"(0x0000BF80u));
    else if (m == 1) asm volatile("v_dot2c_f32_bf16 %0, %2, %1" : "+v"(p.y) : "v"(a), "s"(0xBF800000u));
    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a) : "v"(p.x), "v"(p.y));
  }
}

template <int TYPE, int NF, bool MFMA>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  __shared__ float lds[256];
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  bf16x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(float)(threadIdx.x + i); bv[i] = (__bf16)(float)(i + 1); }
  unsigned a[4] = {threadIdx.x, 2, 3, 4}, b[4] = {5, 6, 7, 8};
  f32x2 p[4], q[4];
  for (int i = 0; i < 4; ++i) { p[i] = f32x2{1.f + i, 2.f}; q[i] = f32x2{0.5f, 0.25f * threadIdx.x}; }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MFMA) {
        if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc0, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < NF; ++f) filler<TYPE>(a[f & 3], b[f & 3], p[f & 3], q[f & 3], f, lds);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y + (float)a[i] + (float)b[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int TYPE, int NF, bool MFMA>
void run(const char* name, float* d, long long* c) {
  const int iters = 2000;
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL((k<TYPE, NF, MFMA>), dim3(256), dim3(threads), 0, 0, d, c, 50);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<TYPE, NF, MFMA>), dim3(256), dim3(threads), 0, 0, d, c, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-10s NF %2d mfma %d waves/SIMD %d: %7.1f ticks per [MFMA+fillers] per wave  (%.3f ms)\n", name, NF, (int)MFMA,
           threads / 256, (double)h / (iters * 8.0), ms);
  }
}
#define ROW(T, NAME)                                                                                   \
  run<T, 0, true>(NAME, d, c); run<T, 2, true>(NAME, d, c); run<T, 4, true>(NAME, d, c);               \
  run<T, 6, true>(NAME, d, c); run<T, 8, true>(NAME, d, c); run<T, 12, true>(NAME, d, c);              \
  run<T, 8, false>(NAME, d, c);
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  long long* c; hipMalloc(&c, 64);
  // warm the clocks
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<0, 4, true>), dim3(256), dim3(512), 0, 0, d, c, 2000);
  hipDeviceSynchronize();
  ROW(0, "v_and") ROW(1, "cvt_pk") ROW(2, "pk_add") ROW(3, "v_sub") ROW(4, "splitmix") ROW(5, "s_nop") ROW(6, "s_add") ROW(7, "dot2c") ROW(8, "dotmix") ROW(9, "dot2c_lit") ROW(10, "dot2c_inl") ROW(11, "cvt_pk_f16") ROW(12, "cvt_f32_f16") ROW(13, "fma_mix") ROW(14, "h2split5") ROW(15, "h2split3")
  return 0;
}
