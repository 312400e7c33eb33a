// fp16 x 2 operand split on gfx950: x ~ h + m with two round-to-nearest fp16 pieces.
//  (a) representation error of the split over |x| in [2^-10, 2^15] and below (denormal pieces)
//  (b) is v_cvt_pk_f16_f32 round-to-nearest-even?  (c) residual by v_fma_mix_f32 == by cvt + sub?
//  (d) does v_mfma_f32_32x32x16_f16 honour denormal fp16 inputs?
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/h2_probe.hip -o tools/micro/h2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void split_err(double* maxrel, unsigned long long* mixbad, int emin, int emax) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned u = mix(i + 1);
  const int e = emin + (int)(mix(i * 3 + 7) % (unsigned)(emax - emin + 1));
  const float v0 = __builtin_bit_cast(float, ((unsigned)(e + 127) << 23) | (u & 0x807fffffu));
  const f32x2 v = {v0, -v0 * 0.73f};
  const h2 h = __builtin_convertvector(v, h2);
  const f32x2 w = __builtin_convertvector(h, f32x2);
  const f32x2 r = {v.x - w.x, v.y - w.y};
  float rx, ry;
  const float neg1 = -1.0f;
  asm volatile("v_fma_mix_f32 %0, %1, %3, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(h), "v"(v.x), "v"(neg1));
  asm volatile("v_fma_mix_f32 %0, %1, %3, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(h), "v"(v.y), "v"(neg1));
  if (__builtin_bit_cast(unsigned, rx) != __builtin_bit_cast(unsigned, r.x) ||
      __builtin_bit_cast(unsigned, ry) != __builtin_bit_cast(unsigned, r.y)) atomicAdd(mixbad, 1ull);
  const h2 m = __builtin_convertvector(r, h2);
  const double back = (double)(float)h.x + (double)(float)m.x;
  const double ab = fabs(back - (double)v.x);
  // atomic max on doubles via the bit pattern (positive values): [0] relative, [1] absolute
  atomicMax(reinterpret_cast<unsigned long long*>(maxrel), __builtin_bit_cast(unsigned long long, ab / fabs((double)v.x)));
  atomicMax(reinterpret_cast<unsigned long long*>(maxrel) + 1, __builtin_bit_cast(unsigned long long, ab));
}
__global__ void ties(unsigned* out) {   // 1 + 2^-11 (tie, even below), 1 + 3*2^-11 (tie, even above)
  const f32x2 v = {1.0f + 0.00048828125f, 1.0f + 3 * 0.00048828125f};
  const h2 h = __builtin_convertvector(v, h2);
  out[0] = __builtin_bit_cast(unsigned, h);
}
__global__ void mfma_denorm(float* out) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1024.f; }   // 2^-20 (denormal), 2^10
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
  double* mr; unsigned long long* mb; unsigned* t; float* o;
  (void)hipMalloc(&mr, 16); (void)hipMalloc(&mb, 8); (void)hipMalloc(&t, 8); (void)hipMalloc(&o, 8);
  const int ranges[6][2] = {{12, 14}, {0, 11}, {-10, -1}, {-14, -11}, {-20, -15}, {-30, -21}};
  for (auto& rg : ranges) {
    (void)hipMemset(mr, 0, 16); (void)hipMemset(mb, 0, 8);
    hipLaunchKernelGGL(split_err, dim3(1 << 16), dim3(256), 0, 0, mr, mb, rg[0], rg[1]);
    double h2v[2]; unsigned long long b;
    (void)hipMemcpy(h2v, mr, 16, hipMemcpyDeviceToHost); const double h = h2v[0]; (void)hipMemcpy(&b, mb, 8, hipMemcpyDeviceToHost);
    printf("|x| in [2^%d, 2^%d): max |x - (h+m)| relative 2^%.1f, absolute 2^%.1f;  v_fma_mix residual differs in %llu cases\n",
           rg[0], rg[1] + 1, log2(h), log2(h2v[1]), b);
  }
  hipLaunchKernelGGL(ties, dim3(1), dim3(1), 0, 0, t);
  unsigned ht; (void)hipMemcpy(&ht, t, 4, hipMemcpyDeviceToHost);
  printf("v_cvt_pk_f16_f32 ties: 1+2^-11 -> 0x%04x (RNE: 0x3c00), 1+3*2^-11 -> 0x%04x (RNE: 0x3c02)\n", ht & 0xffff, ht >> 16);
  hipLaunchKernelGGL(mfma_denorm, dim3(1), dim3(64), 0, 0, o);
  float ho; (void)hipMemcpy(&ho, o, 4, hipMemcpyDeviceToHost);
  printf("mfma f16 with denormal A (2^-20) x 2^10, K=16: %.6e (exact: %.6e)\n", ho, 16 * ldexp(1.0, -10));
  return 0;
}
