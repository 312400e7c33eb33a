// Grouped small GEMMs of the StyleGAN2 plan (see p2l_sg2_k.h).  HBM/latency-bound work:
// every layer's [512 x Cin] modulation matrix is read exactly once per pass; one block
// per (64-column tile, layer), so a whole generator's styles fill the chip in one launch.
#include "p2l_common.h"
#include "p2l_sg2_k.h"

#define ST(s) ((hipStream_t)(s))

namespace p2lsg2 {
namespace {

constexpr int BG = 16;   // batch rows per block (grid.z walks larger batches)

// block = 64 output columns x 4 K-groups; x rows staged in LDS (broadcast reads);
// fixed-order combine of the 4 K-groups -> bit-reproducible, batch-composition independent
__global__ __launch_bounds__(256) void grouped_linear_fwd_kernel(const GLinFwdK k) {
  extern __shared__ float sm[];
  const GLinItem g = k.g[blockIdx.y];
  if ((int)blockIdx.x * 64 >= g.N) return;
  float* xs = sm;                 // [BG][K]
  float* red = sm + BG * g.K;     // [4][BG][64]
  const int tid = threadIdx.x, nl = tid & 63, kg = tid >> 6;
  const int n = blockIdx.x * 64 + nl;
  const int b_begin = blockIdx.z * BG;
  const int nb = min(BG, k.Bn - b_begin);
  for (int i = tid; i < BG * g.K; i += 256) {
    const int b = i / g.K, kk = i - b * g.K;
    float v = (b < nb) ? g.x[(size_t)(b_begin + b) * g.x_ld + kk] : 0.f;
    xs[i] = k.mode ? v * v : v;
  }
  __syncthreads();
  float acc[BG];
#pragma unroll
  for (int b = 0; b < BG; ++b) acc[b] = 0.f;
  const int kper = g.K >> 2, k0 = kg * kper;
  if (n < g.N) {
#pragma unroll 8
    for (int kk = k0; kk < k0 + kper; ++kk) {
      const float w = g.W[(size_t)kk * g.N + n];
#pragma unroll
      for (int b = 0; b < BG; ++b) acc[b] = fmaf(xs[b * g.K + kk], w, acc[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < BG; ++b) red[(kg * BG + b) * 64 + nl] = acc[b];
  __syncthreads();
  for (int i = tid; i < BG * 64; i += 256) {
    const int b = i >> 6, c = i & 63;
    const int nn = blockIdx.x * 64 + c;
    if (b < nb && nn < g.N) {
      float v = (red[(0 * BG + b) * 64 + c] + red[(1 * BG + b) * 64 + c]) +
                (red[(2 * BG + b) * 64 + c] + red[(3 * BG + b) * 64 + c]);
      if (k.mode) v = rsqrtf(v + 1e-8f);
      else if (g.bias) v += g.bias[nn];
      g.y[(size_t)(b_begin + b) * g.y_ld + nn] = v;
    }
  }
}

// one 256-thread block per (k, layer): dx[b][k] = sum_n dy'[b][n] W[k][n]
__global__ __launch_bounds__(256) void grouped_linear_bwd_kernel(const GLinBwdK k) {
  __shared__ float red[BG][4];
  const GLinBwdItem g = k.g[blockIdx.y];
  const int kk = blockIdx.x, tid = threadIdx.x;
  if (kk >= g.K) return;
  const int b_begin = blockIdx.z * BG;
  const int nb = min(BG, k.Bn - b_begin);
  float acc[BG];
#pragma unroll
  for (int b = 0; b < BG; ++b) acc[b] = 0.f;
  const float* wrow = g.W + (size_t)kk * g.N;
  const int N4 = g.N >> 2;
  for (int i = tid; i < N4; i += 256) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(wrow + 4 * i);
#pragma unroll
    for (int b = 0; b < BG; ++b) {
      if (b < nb) {
        const size_t o = (size_t)(b_begin + b) * g.N + 4 * i;
        f32x4 d = *reinterpret_cast<const f32x4*>(g.dy + o);
        if (k.mode) {
          const f32x4 s = *reinterpret_cast<const f32x4*>(g.d + o);
          d.x *= -0.5f * s.x * s.x * s.x; d.y *= -0.5f * s.y * s.y * s.y;
          d.z *= -0.5f * s.z * s.z * s.z; d.w *= -0.5f * s.w * s.w * s.w;
        }
        acc[b] += (d.x * w.x + d.y * w.y) + (d.z * w.z + d.w * w.w);
      }
    }
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int b = 0; b < BG; ++b) {
    const float v = wave_sum(acc[b]);
    if (lane == 0) red[b][wave] = v;
  }
  __syncthreads();
  if (tid < nb) {
    float s = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
    if (k.mode) s *= 2.f * g.x[(size_t)(b_begin + tid) * g.K + kk];
    float* p = g.dx + (size_t)(b_begin + tid) * g.dx_ld + kk;
    *p = g.accumulate ? (*p + s) : s;
  }
}

}  // namespace

int grouped_linear_fwd(const GLinFwdK& k, void* stream) {
  if (k.n < 1 || k.n > GL_MAX || k.Bn < 1) return P2L_EINVAL;
  int maxN = 0, maxK = 0;
  for (int i = 0; i < k.n; ++i) {
    const GLinItem& g = k.g[i];
    if (!g.W || !g.x || !g.y || g.K % 4 || g.K > 1024 || g.N < 1) return P2L_EINVAL;
    if (g.N > maxN) maxN = g.N;
    if (g.K > maxK) maxK = g.K;
  }
  const size_t lds = (size_t)(BG * maxK + 4 * BG * 64) * sizeof(float);
  hipLaunchKernelGGL(grouped_linear_fwd_kernel, dim3(cdiv(maxN, 64), k.n, cdiv(k.Bn, BG)), dim3(256),
                     lds, ST(stream), k);
  return p2l_check_launch();
}

int grouped_linear_bwd(const GLinBwdK& k, void* stream) {
  if (k.n < 1 || k.n > GL_MAX || k.Bn < 1) return P2L_EINVAL;
  int maxK = 0;
  for (int i = 0; i < k.n; ++i) {
    const GLinBwdItem& g = k.g[i];
    if (!g.W || !g.dy || !g.dx || g.N % 4 || (k.mode && (!g.d || !g.x))) return P2L_EINVAL;
    if (g.K > maxK) maxK = g.K;
  }
  hipLaunchKernelGGL(grouped_linear_bwd_kernel, dim3(maxK, k.n, cdiv(k.Bn, BG)), dim3(256), 0,
                     ST(stream), k);
  return p2l_check_launch();
}

}  // namespace p2lsg2
