// Internal (not part of the C ABI): grouped small-GEMM launches of the StyleGAN2 plan.
// All per-layer style projections (EqualLinear modulation, reference call site
// pix2latent/model/stylegan2.py:118,125 -> ModulatedConv2d.modulation) and all
// demodulation scales depend only on the latents, so the plan computes them for every
// layer in ONE launch each instead of 2 latency-bound launches per layer; the backward
// does the same for the demodulation / latent gradients once every layer's ds is known.
#pragma once
#include <cstddef>

namespace p2lsg2 {

constexpr int GL_MAX = 32;

struct GLinItem {
  const float* W;      // [K][N]
  const float* bias;   // [N] or null
  const float* x;      // row b at x + b*x_ld
  float* y;            // row b at y + b*y_ld
  int K, N, x_ld, y_ld;
};
// mode 0: y = x W + bias          mode 1: y = rsqrt((x*x) W + 1e-8)
struct GLinFwdK {
  GLinItem g[GL_MAX];
  int n, Bn, mode;
};
int grouped_linear_fwd(const GLinFwdK& k, void* stream);

struct GLinBwdItem {
  const float* W;      // [K][N]
  const float* dy;     // [B][N]
  const float* d;      // mode 1: demod scale [B][N]
  const float* x;      // mode 1: style s [B][K]
  float* dx;           // row b at dx + b*dx_ld
  int K, N, dx_ld, accumulate;
};
// mode 0: dx (+)= dy W^T          mode 1: dx (+)= 2 x * ((dy * -0.5 d^3) W^T)
struct GLinBwdK {
  GLinBwdItem g[GL_MAX];
  int n, Bn, mode;
};
int grouped_linear_bwd(const GLinBwdK& k, void* stream);

}  // namespace p2lsg2
