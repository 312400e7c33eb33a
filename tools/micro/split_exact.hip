// Is the residual of a bf16 rounding exact when taken with v_dot2c_f32_bf16 (r = v + h * -1)?
// Compares, bit for bit, the three pieces of x = h + m + l formed with (a) shift / mask + v_sub_f32
// and (b) v_dot2c_f32_bf16 against each other on 2^26 values: random bit patterns (every exponent,
// denormals, both signs) and values next to rounding ties.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/split_exact.hip -o tools/micro/split_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 widen2(const bf16x2 p) {
  const unsigned u = __builtin_bit_cast(unsigned, p);
  return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}
__device__ __forceinline__ f32x2 resid_sub(f32x2 v, bf16x2 p) { const f32x2 w = widen2(p); return f32x2{v.x - w.x, v.y - w.y}; }
__device__ __forceinline__ f32x2 resid_dot(f32x2 v, bf16x2 p) {
#ifdef INLINE_CONSTANT   // hipcc emits (-1, 0) as the inline constant "-1.0"; the hardware reads 0xBF800000 = (0, -1)
  const bf16x2 s0 = {(__bf16)-1.0f, (__bf16)0.0f}, s1 = {(__bf16)0.0f, (__bf16)-1.0f};
  return f32x2{__builtin_amdgcn_fdot2_f32_bf16(p, s0, v.x, false), __builtin_amdgcn_fdot2_f32_bf16(p, s1, v.y, false)};
#else
  unsigned s_lo = 0x0000BF80u, s_hi = 0xBF800000u;
  asm("" : "+s"(s_lo));
  asm("" : "+s"(s_hi));
  return f32x2{__builtin_amdgcn_fdot2_f32_bf16(p, __builtin_bit_cast(bf16x2, s_lo), v.x, false),
               __builtin_amdgcn_fdot2_f32_bf16(p, __builtin_bit_cast(bf16x2, s_hi), v.y, false)};
#endif
}
__device__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k(unsigned long long* bad, unsigned* first, int mode) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned u0 = mix(2 * i + 1), u1 = mix(2 * i + 2);
  if (mode == 1) { u0 = (u0 & 0xffff0000u) | 0x8000u | (u0 & 1u); u1 = (u1 & 0xffff0000u) | 0x7fffu + (u1 & 3u); }   // ties
  if (mode == 2) { u0 &= 0x807fffffu; u1 = (u1 & 0x80ffffffu); }                                                  // denormals / tiny
  f32x2 v = {__builtin_bit_cast(float, u0), __builtin_bit_cast(float, u1)};
  if (!isfinite(v.x) || !isfinite(v.y)) return;
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const f32x2 a1 = resid_sub(v, h), b1 = resid_dot(v, h);
  const bf16x2 ma = __builtin_convertvector(a1, bf16x2), mb = __builtin_convertvector(b1, bf16x2);
  const f32x2 a2 = resid_sub(a1, ma), b2 = resid_dot(b1, mb);
  const bool ok = __builtin_bit_cast(unsigned, a1.x) == __builtin_bit_cast(unsigned, b1.x) &&
                  __builtin_bit_cast(unsigned, a1.y) == __builtin_bit_cast(unsigned, b1.y) &&
                  __builtin_bit_cast(unsigned, a2.x) == __builtin_bit_cast(unsigned, b2.x) &&
                  __builtin_bit_cast(unsigned, a2.y) == __builtin_bit_cast(unsigned, b2.y);
  if (!ok) { if (atomicAdd(bad, 1ull) == 0) { first[0] = u0; first[1] = u1; first[2] = __builtin_bit_cast(unsigned, a1.x); first[3] = __builtin_bit_cast(unsigned, b1.x);
                                               first[4] = __builtin_bit_cast(unsigned, a1.y); first[5] = __builtin_bit_cast(unsigned, b1.y); } }
}
int main() {
  unsigned long long* bad; unsigned* first;
  hipMalloc(&bad, 8); hipMalloc(&first, 32);
  for (int mode = 0; mode < 3; ++mode) {
    hipMemset(bad, 0, 8); hipMemset(first, 0, 32);
    hipLaunchKernelGGL(k, dim3(1 << 17), dim3(256), 0, 0, bad, first, mode);
    unsigned long long hb; unsigned hf[8];
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 32, hipMemcpyDeviceToHost);
    printf("mode %d (%s): %llu of %u pairs differ", mode, mode == 0 ? "random bits" : mode == 1 ? "ties" : "denormal / tiny", hb, 1u << 25);
    if (hb) printf("  first: v %08x %08x  r1.x sub %08x dot %08x  r1.y sub %08x dot %08x", hf[0], hf[1], hf[2], hf[3], hf[4], hf[5]);
    printf("\n");
  }
  return 0;
}
