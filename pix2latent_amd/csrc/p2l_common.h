// Shared helpers for libp2l_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "p2l.h"
#include "p2l_test.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

extern thread_local int g_p2l_last_hip_error;

static inline int p2l_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_p2l_last_hip_error = (int)e;
    return P2L_ELAUNCH;
  }
  return P2L_OK;
}

// fold n >= 256 partial maxima per image to 256 (p2l_plan.hip; max is exact): out [B][256]
int p2l_amax_compact(const float* in, int B, int n, float* out, void* st);

static inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Bijective XCD-aware remap of a 1-D grid: MI355X dispatches block b to XCD b%8;
// give every XCD a contiguous range of logical tiles so neighbouring tiles
// (shared halos / shared A panels) hit the same per-XCD L2.
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = id & 7, pos = id >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + pos;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
