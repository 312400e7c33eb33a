// StyleGAN2-specific HBM-bound kernels (rosinality stylegan2-pytorch ops reached from
// reference pix2latent/model/stylegan2.py:116-125; its two CUDA extensions
// fused_bias_act / upfirdn2d are replaced by the kernels below):
//   * mapping network pieces: PixelNorm, bias + leaky-ReLU*sqrt2 (fused_bias_act)
//   * weight demodulation factors d[b,o] = rsqrt(sum_i s^2 Wsq + eps) and their gradient
//   * the 4x4 FIR blur that follows the stride-2 transposed conv, fused with the
//     demodulation scale, noise injection, bias and leaky ReLU (forward), and its
//     transpose fused with the activation backward and the per-(b,c) / per-pixel
//     reductions (backward)
//   * the RGB skip upsample (upfirdn2d up=2) forward / transpose
//   * clamp(-1,1) forward / backward on NHWC16 images
// All reductions are fixed-order (CMA-ES ranks by the resulting losses).
#include "p2l_common.h"

namespace {

#define ST(s) ((hipStream_t)(s))
constexpr float kSqrt2 = 1.41421356237f;
constexpr float kSlope = 0.2f;

__global__ void pixelnorm_fwd_kernel(const float* z, float* y, int Bn, int D) {
  // one wave per row
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= Bn) return;
  float ss = 0.f;
  for (int i = lane; i < D; i += 64) { const float v = z[(size_t)row * D + i]; ss += v * v; }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / D + 1e-8f);
  for (int i = lane; i < D; i += 64) y[(size_t)row * D + i] = z[(size_t)row * D + i] * r;
}
__global__ void pixelnorm_bwd_kernel(const float* z, const float* dy, float* dz, int Bn, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= Bn) return;
  float ss = 0.f, dot = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float v = z[(size_t)row * D + i];
    ss += v * v;
    dot += v * dy[(size_t)row * D + i];
  }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float r = rsqrtf(ss / D + 1e-8f);
  // y = z r, r = (mean z^2 + eps)^-1/2  =>  dz = r dy - z r^3 (z.dy)/D
  const float c = r * r * r * dot / D;
  for (int i = lane; i < D; i += 64)
    dz[(size_t)row * D + i] = r * dy[(size_t)row * D + i] - z[(size_t)row * D + i] * c;
}

__global__ void bias_lrelu_fwd_kernel(float* x, const float* bias, float bias_mul, int n, int D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i] + bias[i % D] * bias_mul;
  x[i] = v * (v > 0.f ? kSqrt2 : kSlope * kSqrt2);
}
__global__ void lrelu_bwd_kernel(const float* y, float* g, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  g[i] *= (y[i] > 0.f ? kSqrt2 : kSlope * kSqrt2);
}

// d[b,o] = rsqrt(sum_i s[b,i]^2 Wsq[i][o] + 1e-8)
__global__ void demod_fwd_kernel(const float* s, const float* Wsq, float* d, int Bn, int Cin,
                                 int Cout) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Bn * Cout) return;
  const int b = i / Cout, o = i - b * Cout;
  float q = 0.f;
  for (int c = 0; c < Cin; ++c) {
    const float sv = s[(size_t)b * Cin + c];
    q = fmaf(sv * sv, Wsq[(size_t)c * Cout + o], q);
  }
  d[i] = rsqrtf(q + 1e-8f);
}
// ds[b,i] (+)= 2 s[b,i] sum_o (dd[b,o] * -0.5 d^3) Wsq[i][o]
__global__ void demod_bwd_kernel(const float* s, const float* Wsq, const float* d,
                                 const float* dd, float* ds, int Bn, int Cin, int Cout,
                                 int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Bn * Cin) return;
  const int b = i / Cin, c = i - b * Cin;
  float a = 0.f;
  for (int o = 0; o < Cout; ++o) {
    const float dv = d[(size_t)b * Cout + o];
    a = fmaf(dd[(size_t)b * Cout + o] * (-0.5f * dv * dv * dv), Wsq[(size_t)c * Cout + o], a);
  }
  const float r = 2.f * s[i] * a;
  ds[i] = accumulate ? ds[i] + r : r;
}

// ---- per-image maxima for the fp16 x 2 conv that READS a tensor these kernels write (P2LAmax.in) ----
// The plan hands the reader [B][P2L_SG2_AMAX_SLOTS] partial maxima it zeroed before the producer ran.  A wave that
// sits inside one image (all but the few at an image boundary) reduces its lanes and issues ONE atomic max on the
// bit pattern (non-negative floats order like unsigned integers).  max is exact, so the result does not depend
// on the order the waves arrive in: the reader's power-of-two scale -- and with it every bit of its output --
// is what its own pass over the tensor would have given (tests/test_sg2_amax_gpu.py).
__device__ __forceinline__ float absmax4s(float m, const f32x4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ void amax_commit(float m, int b, unsigned slot, float* amax_out) {
  unsigned* dst = reinterpret_cast<unsigned*>(amax_out);
  slot &= (P2L_SG2_AMAX_SLOTS - 1);
  const int b0 = __builtin_amdgcn_readfirstlane(b);
  // (the ballot counts ACTIVE lanes only: all 64 bits set = nobody left the kernel early and one image)
  if (__ballot(b == b0) == ~0ull) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(dst + (size_t)b0 * P2L_SG2_AMAX_SLOTS + slot, __float_as_uint(m));
  } else {
    atomicMax(dst + (size_t)b * P2L_SG2_AMAX_SLOTS + slot, __float_as_uint(m));
  }
}

// ---- FIR blur after the transposed conv ------------------------------------
// u: [B, H+2, W+2, C] (rows/cols 0..H real, H+1 zero), y: [B, H, W, C]
// pre = blur(u)[Y,X] * d[b,c] + nw * noise[b,Y*W+X] + bias[c] ;  y = lrelu(pre)*sqrt2
// blur(u)[Y,X] = sum_{j,i} a_j a_i u[Y+j-1, X+i-1],  a = [.25 .75 .75 .25]
__device__ __forceinline__ float fir_a(int j) { return (j == 0 || j == 3) ? 0.25f : 0.75f; }

// A thread produces a 4 x 4 patch of outputs x 4 channels from the 7 x 7 inputs under it: 49 loads per 16
// outputs.  (Round 2: one output per thread, 16 loads each, 1.8 TB/s; round 3: four vertically adjacent outputs,
// 7 loads each, 2.3 TB/s -- 512^2 x 64 x 32 candidates, request-bound: every input element was fetched by four
// threads through L2.)  The sums keep the order of those forms: horizontal taps i = 0..3 into a row sum,
// vertical taps j = 0..3 over the row sums -- the same bits.
// amax_out: the maxima of |y * next_s[b,c]| (next_s = the style the next conv fuses into its prologue; NULL: of |y|)
// go there (amax_commit above).  ONE kernel for both forms (a run-time test, not a template parameter): the
// arithmetic that produces y must be the same instructions with and without the hand-over.
__global__ __launch_bounds__(256) void blur_fwd_kernel(const float* u, const float* d, const float* noise, float nw,
                                                       const float* bias, float* y, int Bn, int H, int W, int C,
                                                       const float* next_s, float* amax_out) {
  const int C4 = C >> 2, H4 = H >> 2, W4 = W >> 2;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)Bn * H4 * W4 * C4) return;
  const int c = (int)(idx % C4) * 4;
  size_t p = idx / C4;
  const int X0 = (int)(p % W4) * 4;
  p /= W4;
  const int Y0 = (int)(p % H4) * 4;
  const int b = (int)(p / H4);
  const int UW = W + 2, UH = H + 2;
  const f32x4 zero = {0, 0, 0, 0};
  f32x4 out[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int x = 0; x < 4; ++x) out[k][x] = zero;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const int yy = Y0 + r - 1;                       // (<= H + 1: inside the frame)
    f32x4 h[4] = {zero, zero, zero, zero};
    if (yy >= 0) {
      const float* row = u + (((size_t)b * UH + yy) * UW) * C + c;
      f32x4 in[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int xx = X0 + i - 1;                   // (<= W + 1)
        in[i] = xx >= 0 ? *reinterpret_cast<const f32x4*>(row + (size_t)xx * C) : zero;
      }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (X0 + x + i - 1 >= 0) h[x] += in[x + i] * fir_a(i);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = r - k;
      if (j < 0 || j > 3) continue;
#pragma unroll
      for (int x = 0; x < 4; ++x) out[k][x] += h[x] * fir_a(j);
    }
  }
  const f32x4 d4 = *reinterpret_cast<const f32x4*>(d + (size_t)b * C + c);
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + c);
  f32x4 ns4 = {1.f, 1.f, 1.f, 1.f};
  if (amax_out && next_s) ns4 = *reinterpret_cast<const f32x4*>(next_s + (size_t)b * C + c);
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int Y = Y0 + k;
    f32x4 nz4 = zero;
    if (noise) nz4 = *reinterpret_cast<const f32x4*>(noise + ((size_t)b * H + Y) * W + X0) * nw;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      f32x4 v = out[k][x] * d4 + b4 + nz4[x];
      v.x *= v.x > 0.f ? kSqrt2 : kSlope * kSqrt2;
      v.y *= v.y > 0.f ? kSqrt2 : kSlope * kSqrt2;
      v.z *= v.z > 0.f ? kSqrt2 : kSlope * kSqrt2;
      v.w *= v.w > 0.f ? kSqrt2 : kSlope * kSqrt2;
      *reinterpret_cast<f32x4*>(y + (((size_t)b * H + Y) * W + X0 + x) * C + c) = v;
      if (amax_out) m = absmax4s(m, next_s ? v * ns4 : v);
    }
  }
  if (amax_out) amax_commit(m, b, blockIdx.x * 4u + (threadIdx.x >> 6), amax_out);
}

// activation backward of a styled conv (with or without blur): given dy and the saved
// output y:  g1 = dy * lrelu'(y) ;  gd = g1 * d[b,c] (gradient w.r.t. the un-scaled conv /
// blur result, written to gd);  partial sums over the slab of
//   sum_p g1 * c_val   with c_val = (pre - nw*noise - bias)/d  (the un-scaled conv result)
// for the demodulation gradient, and per-pixel channel sums for the noise gradient.
struct ActBwdK {
  const float* dy; const float* y; const float* d; const float* noise; const float* bias;
  float* gd; float* partial; float* dnoise;
  float nw; int Bn, P, C, nblk;
  float* amax_out;     // [Bn][P2L_SG2_AMAX_SLOTS] maxima of |gd| (zeroed by the caller), or NULL
};
constexpr int AB_SLAB = 256;
// SW = channel strip width per block (64, or 32 for the 32-channel FFHQ 1024^2 layers)
template <int SW>
__global__ __launch_bounds__(256) void styled_act_bwd_kernel(const ActBwdK k) {
  constexpr int CL = SW / 4, PL = 256 / CL;
  __shared__ f32x4 red[256];
  const int tid = threadIdx.x, cl = tid % CL, pl = tid / CL;
  const int slab = blockIdx.x, c = blockIdx.y * SW + cl * 4, b = blockIdx.z;
  const f32x4 d4 = *reinterpret_cast<const f32x4*>(k.d + (size_t)b * k.C + c);
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(k.bias + c);
  f32x4 acc = {0, 0, 0, 0};
  float amax = 0.f;
  const int p_end = min(k.P, (slab + 1) * AB_SLAB);
  for (int p = slab * AB_SLAB + pl; p < p_end; p += PL) {
    const size_t o = ((size_t)b * k.P + p) * k.C + c;
    const f32x4 yv = *reinterpret_cast<const f32x4*>(k.y + o);
    f32x4 g = *reinterpret_cast<const f32x4*>(k.dy + o);
    f32x4 pre;
    g.x *= yv.x > 0.f ? kSqrt2 : kSlope * kSqrt2; pre.x = yv.x / (yv.x > 0.f ? kSqrt2 : kSlope * kSqrt2);
    g.y *= yv.y > 0.f ? kSqrt2 : kSlope * kSqrt2; pre.y = yv.y / (yv.y > 0.f ? kSqrt2 : kSlope * kSqrt2);
    g.z *= yv.z > 0.f ? kSqrt2 : kSlope * kSqrt2; pre.z = yv.z / (yv.z > 0.f ? kSqrt2 : kSlope * kSqrt2);
    g.w *= yv.w > 0.f ? kSqrt2 : kSlope * kSqrt2; pre.w = yv.w / (yv.w > 0.f ? kSqrt2 : kSlope * kSqrt2);
    const float nz = k.noise ? k.nw * k.noise[(size_t)b * k.P + p] : 0.f;
    const f32x4 cval = (pre - b4 - nz) / d4;
    acc += g * cval;
    const f32x4 gd4 = g * d4;
    *reinterpret_cast<f32x4*>(k.gd + o) = gd4;
    amax = absmax4s(amax, gd4);
    if (k.dnoise) {
      // channel sum of g1 for this pixel: CL lanes x float4 of this SW-channel strip
      float sn = (g.x + g.y) + (g.z + g.w);
      sn += __shfl_xor(sn, 1, 64); sn += __shfl_xor(sn, 2, 64);
      sn += __shfl_xor(sn, 4, 64);
      if (CL == 16) sn += __shfl_xor(sn, 8, 64);
      if (cl == 0) k.dnoise[((size_t)blockIdx.y * k.Bn + b) * k.P + p] = sn;   // per strip
    }
  }
  red[tid] = acc;
  __syncthreads();
  if (pl == 0) {
    f32x4 a = red[cl];
#pragma unroll
    for (int j = 1; j < PL; ++j) a += red[j * CL + cl];
    *reinterpret_cast<f32x4*>(k.partial + ((size_t)b * k.nblk + slab) * k.C + c) = a;
  }
  if (k.amax_out) amax_commit(amax, b, (slab * gridDim.y + blockIdx.y) * 4u + (tid >> 6), k.amax_out);
}
// grid (cdiv(C,64), B); 16 segments x 16 channel-float4 lanes: each thread adds its
// segment's slabs in order, then the segments are combined in a fixed order
__global__ __launch_bounds__(256) void rows_sum_finish_kernel(const float* partial, float* out,
                                                              int Bn, int nblk, int C) {
  __shared__ f32x4 red[256];
  const int tid = threadIdx.x, cl = tid & 15, seg = tid >> 4;
  const int c = blockIdx.x * 64 + cl * 4, b = blockIdx.y;
  const bool live = c < C;
  f32x4 a = {0, 0, 0, 0};
  for (int j = seg; live && j < nblk; j += 16)
    a += *reinterpret_cast<const f32x4*>(partial + ((size_t)b * nblk + j) * C + c);
  red[tid] = a;
  __syncthreads();
  if (seg == 0 && live) {
#pragma unroll
    for (int j = 1; j < 16; ++j) a += red[j * 16 + cl];
    *reinterpret_cast<f32x4*>(out + (size_t)b * C + c) = a;
  }
}
// ... for every recorded layer in ONE launch (p2l_sg2_rows_defer_*): grid (all channel groups of all layers, B).
// The backward pass of the synthesis network needs the demodulation gradients only at its end; 17 launches of a
// few blocks each were 0.33 ms of the 22.6 ms FFHQ-1024 step.  Same body, same order: same bits.
constexpr int ROWS_GROUP_MAX = P2L_SG2_MAX_CONVS;
struct RowsFin { const float* partial; float* out; int nblk, C, first; };
struct RowsFinGroup { RowsFin e[ROWS_GROUP_MAX]; int n, Bn; };
__global__ __launch_bounds__(256) void rows_sum_finish_group_kernel(const RowsFinGroup g) {
  __shared__ f32x4 red[256];
  int l = 0;
  while (l + 1 < g.n && (int)blockIdx.x >= g.e[l + 1].first) ++l;      // (uniform)
  const RowsFin& e = g.e[l];
  const int tid = threadIdx.x, cl = tid & 15, seg = tid >> 4;
  const int c = ((int)blockIdx.x - e.first) * 64 + cl * 4, b = blockIdx.y;
  const bool live = c < e.C;
  f32x4 a = {0, 0, 0, 0};
  for (int j = seg; live && j < e.nblk; j += 16)
    a += *reinterpret_cast<const f32x4*>(e.partial + ((size_t)b * e.nblk + j) * e.C + c);
  red[tid] = a;
  __syncthreads();
  if (seg == 0 && live) {
#pragma unroll
    for (int j = 1; j < 16; ++j) a += red[j * 16 + cl];
    *reinterpret_cast<f32x4*>(e.out + (size_t)b * e.C + c) = a;
  }
}
thread_local bool g_rows_defer = false;
thread_local RowsFinGroup g_rows_group;
int rows_group_launch(void* stream) {
  RowsFinGroup& g = g_rows_group;
  if (g.n == 0) return P2L_OK;
  int total = 0;
  for (int i = 0; i < g.n; ++i) { g.e[i].first = total; total += cdiv(g.e[i].C, 64); }
  hipLaunchKernelGGL(rows_sum_finish_group_kernel, dim3(total, g.Bn), dim3(256), 0, ST(stream), g);
  g.n = 0;
  return p2l_check_launch();
}

// dnoise[b,p] = nw * sum over 64-channel strips
__global__ void noise_grad_finish_kernel(const float* strips, float* dnoise, float nw, int nstrip,
                                         size_t BP) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= BP) return;
  float a = 0.f;
  for (int s = 0; s < nstrip; ++s) a += strips[(size_t)s * BP + i];
  dnoise[i] = nw * a;
}

// transpose of the blur: du[B,H+2,W+2,C] from g[B,H,W,C] (g already scaled by d).  As the forward kernel: a
// 4 x 4 patch of du per thread from the 7 x 7 gradients that reach it (the frame is (H+2)^2: patches past its
// end are cut), rows walked downwards so that the vertical taps add in the order j = 0..3 of the earlier forms.
__global__ __launch_bounds__(256) void blur_bwd_kernel(const float* g, float* du, int Bn, int H, int W, int C,
                                                       float* amax_out) {
  const int C4 = C >> 2, UH = H + 2, UW = W + 2, UH4 = (UH + 3) >> 2, UW4 = (UW + 3) >> 2;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)Bn * UH4 * UW4 * C4) return;
  const int c = (int)(idx % C4) * 4;
  size_t p = idx / C4;
  const int xx0 = (int)(p % UW4) * 4;
  p /= UW4;
  const int yy0 = (int)(p % UH4) * 4;
  const int b = (int)(p / UH4);
  const f32x4 zero = {0, 0, 0, 0};
  f32x4 out[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int x = 0; x < 4; ++x) out[k][x] = zero;
  // u[yy,xx] feeds y[Y,X] with Y = yy - j + 1, X = xx - i + 1: rows Y = yy0 - 2 .. yy0 + 4, columns likewise
#pragma unroll
  for (int r = 6; r >= 0; --r) {
    const int Y = yy0 + r - 2;
    f32x4 h[4] = {zero, zero, zero, zero};
    if (Y >= 0 && Y < H) {
      const float* row = g + (((size_t)b * H + Y) * W) * C + c;
      f32x4 in[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const int X = xx0 + q - 2;
        in[q] = (X >= 0 && X < W) ? *reinterpret_cast<const f32x4*>(row + (size_t)X * C) : zero;
      }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int i = 0; i < 4; ++i) {                 // X = (xx0 + x) - i + 1 = xx0 + (x - i + 3) - 2
          const int X = xx0 + x - i + 1;
          if (X >= 0 && X < W) h[x] += in[x - i + 3] * fir_a(i);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = k + 3 - r;                        // Y = yy - j + 1 = yy0 + (k + 3 - j) - 2
      if (j < 0 || j > 3) continue;
#pragma unroll
      for (int x = 0; x < 4; ++x) out[k][x] += h[x] * fir_a(j);
    }
  }
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int yy = yy0 + k;
    if (yy >= UH) break;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int xx = xx0 + x;
      if (xx >= UW) break;
      *reinterpret_cast<f32x4*>(du + (((size_t)b * UH + yy) * UW + xx) * C + c) = out[k][x];
      if (amax_out) m = absmax4s(m, out[k][x]);
    }
  }
  if (amax_out) amax_commit(m, b, blockIdx.x * 4u + (threadIdx.x >> 6), amax_out);
}

// RGB skip: out[B,2h,2w,16] = upfirdn2d(skip[B,h,w,16], up=2, pad=(2,1)); per dim:
//   even 2q: .25 s[q-1] + .75 s[q] ; odd 2q+1: .75 s[q] + .25 s[q+1]
__device__ __forceinline__ f32x4 ld4c(const float* s, int y, int x, int w) {
  return *reinterpret_cast<const f32x4*>(s + ((size_t)y * w + x) * 16);
}
__device__ __forceinline__ void up_taps(int Y, int h, int& q0, float& w0, int& q1, float& w1) {
  const int q = Y >> 1;
  if (Y & 1) { q0 = q; w0 = 0.75f; q1 = q + 1; w1 = (q + 1 < h) ? 0.25f : 0.f; }
  else { q0 = q - 1; w0 = (q - 1 >= 0) ? 0.25f : 0.f; q1 = q; w1 = 0.75f; }
  if (q0 < 0) q0 = 0;
  if (q1 >= h) q1 = h - 1;
}
__global__ void rgb_up_fwd_kernel(const float* skip, float* out, int Bn, int h, int w) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int H = 2 * h, W = 2 * w;
  if (idx >= (size_t)Bn * H * W) return;
  const int X = (int)(idx % W), Y = (int)((idx / W) % H), b = (int)(idx / ((size_t)W * H));
  int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
  up_taps(Y, h, y0, wy0, y1, wy1);
  up_taps(X, w, x0, wx0, x1, wx1);
  const float* s = skip + (size_t)b * h * w * 16;
  const f32x4 v = (ld4c(s, y0, x0, w) * wx0 + ld4c(s, y0, x1, w) * wx1) * wy0 +
                  (ld4c(s, y1, x0, w) * wx0 + ld4c(s, y1, x1, w) * wx1) * wy1;
  f32x4* o = reinterpret_cast<f32x4*>(out + idx * 16);
  const f32x4 z = {0, 0, 0, 0};
  o[0] = v; o[1] = z; o[2] = z; o[3] = z;
}
// transpose: dskip[m] = sum over outputs it feeds
__global__ void rgb_up_bwd_kernel(const float* dout, float* dskip, int Bn, int h, int w,
                                  int accumulate) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)Bn * h * w) return;
  const int x = (int)(idx % w), y = (int)((idx / w) % h), b = (int)(idx / ((size_t)w * h));
  const int H = 2 * h, W = 2 * w;
  // rows fed by y: 2y-1 (.25), 2y (.75), 2y+1 (.75), 2y+2 (.25)
  const int ry[4] = {2 * y - 1, 2 * y, 2 * y + 1, 2 * y + 2};
  const int rx[4] = {2 * x - 1, 2 * x, 2 * x + 1, 2 * x + 2};
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (ry[j] < 0 || ry[j] >= H) continue;
    f32x4 row = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (rx[i] < 0 || rx[i] >= W) continue;
      row += *reinterpret_cast<const f32x4*>(dout + (((size_t)b * H + ry[j]) * W + rx[i]) * 16) * fir_a(i);
    }
    acc += row * fir_a(j);
  }
  f32x4* o = reinterpret_cast<f32x4*>(dskip + idx * 16);
  const f32x4 z = {0, 0, 0, 0};
  o[0] = accumulate ? o[0] + acc : acc;
  if (!accumulate) { o[1] = z; o[2] = z; o[3] = z; }
}

__global__ void clamp16_fwd_kernel(const float* x, float* y, size_t P) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 16);
  v.x = fminf(fmaxf(v.x, -1.f), 1.f); v.y = fminf(fmaxf(v.y, -1.f), 1.f);
  v.z = fminf(fmaxf(v.z, -1.f), 1.f); v.w = 0.f;
  f32x4* o = reinterpret_cast<f32x4*>(y + i * 16);
  const f32x4 z = {0, 0, 0, 0};
  o[0] = v; o[1] = z; o[2] = z; o[3] = z;
}
// torch clamp backward: gradient passes where min <= x <= max
__global__ void clamp16_bwd_kernel(const float* x, const float* dy, float* dx, size_t P) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 16);
  f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 16);
  g.x = (v.x >= -1.f && v.x <= 1.f) ? g.x : 0.f;
  g.y = (v.y >= -1.f && v.y <= 1.f) ? g.y : 0.f;
  g.z = (v.z >= -1.f && v.z <= 1.f) ? g.z : 0.f;
  g.w = 0.f;
  f32x4* o = reinterpret_cast<f32x4*>(dx + i * 16);
  const f32x4 z = {0, 0, 0, 0};
  o[0] = g; o[1] = z; o[2] = z; o[3] = z;
}

// The per-layer noise between its two layouts: per sample [B][noise_total] (the reference's `noises` variable,
// model/stylegan2.py:128-138 reshape_noise) <-> layer-major (layer l at Bn*off[l], [Bn][h*w]: what the synthesis
// entry points take).  One launch instead of 17 strided slice copies + a cat (and 17 zero-fills, scatters and adds
// in the backward): 1.3 ms of the 25 ms FFHQ-1024 step.
struct NoiseLayoutK {
  int n_layers, total, Bn;
  int off[P2L_SG2_MAX_CONVS + 1];          // per-sample offset of layer l; off[n_layers] = total
};
__global__ void noise_relayout_kernel(const float* src, float* dst, const NoiseLayoutK k, int to_layer_major) {
  // the offsets in LDS (indexed per thread: out of the kernel-argument segment that was a chain of dependent
  // memory loads per step of the search), searched from the LAST layer down: the two highest resolutions hold
  // three quarters of a sample (268 -> 3x GB/s: 0.25 ms per call of the FFHQ-1024 step before)
  __shared__ int off[P2L_SG2_MAX_CONVS + 1];
  if ((int)threadIdx.x <= k.n_layers) off[threadIdx.x] = k.off[threadIdx.x];
  __syncthreads();
  const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;      // (every layer is 4^2 ... : multiples of 4)
  if (i4 >= (size_t)k.Bn * k.total) return;
  const int b = (int)(i4 / k.total), j = (int)(i4 - (size_t)b * k.total);
  int l = k.n_layers - 1;
  while (l > 0 && j < off[l]) --l;
  const int hw = off[l + 1] - off[l];
  const size_t lm = (size_t)k.Bn * off[l] + (size_t)b * hw + (j - off[l]);
  if (to_layer_major) *reinterpret_cast<f32x4*>(dst + lm) = *reinterpret_cast<const f32x4*>(src + i4);
  else *reinterpret_cast<f32x4*>(dst + i4) = *reinterpret_cast<const f32x4*>(src + lm);
}

// repeat a [1,h,w,C] constant over the batch
__global__ void broadcast_rows_kernel(const float* src, float* dst, size_t n, int Bn) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * Bn) return;
  dst[i] = src[i % n];
}
__global__ void add_inplace_kernel(float* a, const float* b, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] += b[i];
}

}  // namespace

extern "C" int p2l_sg2_pixelnorm_fwd(const float* z, float* y, int Bn, int D, void* stream) {
  hipLaunchKernelGGL(pixelnorm_fwd_kernel, dim3(cdiv(Bn, 4)), dim3(256), 0, ST(stream), z, y, Bn, D);
  return p2l_check_launch();
}
extern "C" int p2l_sg2_pixelnorm_bwd(const float* z, const float* dy, float* dz, int Bn, int D,
                                     void* stream) {
  hipLaunchKernelGGL(pixelnorm_bwd_kernel, dim3(cdiv(Bn, 4)), dim3(256), 0, ST(stream), z, dy, dz, Bn, D);
  return p2l_check_launch();
}
extern "C" int p2l_sg2_bias_lrelu_fwd(float* x, const float* bias, float bias_mul, int Bn, int D,
                                      void* stream) {
  hipLaunchKernelGGL(bias_lrelu_fwd_kernel, dim3(cdiv(Bn * D, 256)), dim3(256), 0, ST(stream), x, bias,
                     bias_mul, Bn * D, D);
  return p2l_check_launch();
}
extern "C" int p2l_sg2_lrelu_bwd(const float* y, float* g, int n, void* stream) {
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), y, g, n);
  return p2l_check_launch();
}
extern "C" int p2l_sg2_demod_fwd(const float* s, const float* Wsq, float* d, int Bn, int Cin,
                                 int Cout, void* stream) {
  hipLaunchKernelGGL(demod_fwd_kernel, dim3(cdiv(Bn * Cout, 256)), dim3(256), 0, ST(stream), s, Wsq, d,
                     Bn, Cin, Cout);
  return p2l_check_launch();
}
extern "C" int p2l_sg2_demod_bwd(const float* s, const float* Wsq, const float* d, const float* dd,
                                 float* ds, int Bn, int Cin, int Cout, int accumulate,
                                 void* stream) {
  hipLaunchKernelGGL(demod_bwd_kernel, dim3(cdiv(Bn * Cin, 256)), dim3(256), 0, ST(stream), s, Wsq, d,
                     dd, ds, Bn, Cin, Cout, accumulate);
  return p2l_check_launch();
}
extern "C" int p2l_sg2_blur_fwd_amax(const float* u, const float* d, const float* noise, float nw,
                                     const float* bias, float* y, int Bn, int H, int W, int C,
                                     const float* next_s, float* amax_out, void* stream) {
  if (C % 4) return P2L_EINVAL;
  if (H % 4 || W % 4) return P2L_EINVAL;
  const dim3 grid(cdiv((size_t)Bn * (H / 4) * (W / 4) * (C / 4), 256));
  hipLaunchKernelGGL(blur_fwd_kernel, grid, dim3(256), 0, ST(stream), u, d, noise, nw, bias, y, Bn, H, W, C,
                     amax_out ? next_s : nullptr, amax_out);
  return p2l_check_launch();
}
extern "C" int p2l_sg2_blur_fwd(const float* u, const float* d, const float* noise, float nw,
                                const float* bias, float* y, int Bn, int H, int W, int C,
                                void* stream) {
  return p2l_sg2_blur_fwd_amax(u, d, noise, nw, bias, y, Bn, H, W, C, nullptr, nullptr, stream);
}
extern "C" int p2l_sg2_act_bwd_nblk(int P) { return cdiv(P, AB_SLAB); }
// partial: Bn*nblk*C floats; strips: (C/sw)*Bn*P floats, sw = C%64 ? 32 : 64 (only when dnoise != NULL)
extern "C" int p2l_sg2_styled_act_bwd(const float* dy, const float* y, const float* d,
                                      const float* noise, float nw, const float* bias, float* gd,
                                      float* dd, float* dnoise, float* partial, float* strips,
                                      int Bn, int P, int C, void* stream) {
  return p2l_sg2_styled_act_bwd_amax(dy, y, d, noise, nw, bias, gd, dd, dnoise, partial, strips, Bn, P, C, nullptr,
                                     stream);
}
extern "C" int p2l_sg2_styled_act_bwd_amax(const float* dy, const float* y, const float* d,
                                           const float* noise, float nw, const float* bias, float* gd,
                                           float* dd, float* dnoise, float* partial, float* strips,
                                           int Bn, int P, int C, float* amax_out, void* stream) {
  if (C % 32) return P2L_EINVAL;
  const int sw = (C % 64) ? 32 : 64;
  ActBwdK k{};
  k.dy = dy; k.y = y; k.d = d; k.noise = noise; k.bias = bias; k.gd = gd; k.partial = partial;
  k.dnoise = dnoise ? strips : nullptr;
  k.nw = nw; k.Bn = Bn; k.P = P; k.C = C; k.nblk = cdiv(P, AB_SLAB);
  k.amax_out = amax_out;
  if (sw == 64)
    hipLaunchKernelGGL(styled_act_bwd_kernel<64>, dim3(k.nblk, C / 64, Bn), dim3(256), 0, ST(stream), k);
  else
    hipLaunchKernelGGL(styled_act_bwd_kernel<32>, dim3(k.nblk, C / 32, Bn), dim3(256), 0, ST(stream), k);
  if (g_rows_defer) {
    RowsFinGroup& g = g_rows_group;
    if (g.n == ROWS_GROUP_MAX || (g.n > 0 && g.Bn != Bn)) {
      const int rc = rows_group_launch(stream);
      if (rc) return rc;
    }
    g.Bn = Bn;
    g.e[g.n++] = RowsFin{partial, dd, k.nblk, C, 0};
  } else {
    hipLaunchKernelGGL(rows_sum_finish_kernel, dim3(cdiv(C, 64), Bn), dim3(256), 0, ST(stream), partial,
                       dd, Bn, k.nblk, C);
  }
  if (dnoise)
    hipLaunchKernelGGL(noise_grad_finish_kernel, dim3(cdiv((size_t)Bn * P, 256)), dim3(256), 0,
                       ST(stream), strips, dnoise, nw, C / sw, (size_t)Bn * P);
  return p2l_check_launch();
}
extern "C" void p2l_sg2_rows_defer_begin(void) { g_rows_defer = true; g_rows_group.n = 0; }
extern "C" void p2l_sg2_rows_defer_cancel(void) { g_rows_defer = false; g_rows_group.n = 0; }
extern "C" int p2l_sg2_rows_defer_flush(void* stream) {
  g_rows_defer = false;
  return rows_group_launch(stream);
}
extern "C" int p2l_sg2_blur_bwd_amax(const float* g, float* du, int Bn, int H, int W, int C, float* amax_out,
                                     void* stream) {
  if (C % 4) return P2L_EINVAL;
  const dim3 grid(cdiv((size_t)Bn * ((H + 5) / 4) * ((W + 5) / 4) * (C / 4), 256));
  hipLaunchKernelGGL(blur_bwd_kernel, grid, dim3(256), 0, ST(stream), g, du, Bn, H, W, C, amax_out);
  return p2l_check_launch();
}
extern "C" int p2l_sg2_blur_bwd(const float* g, float* du, int Bn, int H, int W, int C,
                                void* stream) {
  return p2l_sg2_blur_bwd_amax(g, du, Bn, H, W, C, nullptr, stream);
}
extern "C" int p2l_sg2_rgb_up_fwd(const float* skip, float* out, int Bn, int h, int w, void* stream) {
  hipLaunchKernelGGL(rgb_up_fwd_kernel, dim3(cdiv((size_t)Bn * 4 * h * w, 256)), dim3(256), 0,
                     ST(stream), skip, out, Bn, h, w);
  return p2l_check_launch();
}
extern "C" int p2l_sg2_rgb_up_bwd(const float* dout, float* dskip, int Bn, int h, int w,
                                  int accumulate, void* stream) {
  hipLaunchKernelGGL(rgb_up_bwd_kernel, dim3(cdiv((size_t)Bn * h * w, 256)), dim3(256), 0, ST(stream),
                     dout, dskip, Bn, h, w, accumulate);
  return p2l_check_launch();
}
extern "C" int p2l_sg2_clamp16_fwd(const float* x, float* y, int64_t P, void* stream) {
  hipLaunchKernelGGL(clamp16_fwd_kernel, dim3(cdiv(P, 256)), dim3(256), 0, ST(stream), x, y, (size_t)P);
  return p2l_check_launch();
}
extern "C" int p2l_sg2_clamp16_bwd(const float* x, const float* dy, float* dx, int64_t P,
                                   void* stream) {
  hipLaunchKernelGGL(clamp16_bwd_kernel, dim3(cdiv(P, 256)), dim3(256), 0, ST(stream), x, dy, dx,
                     (size_t)P);
  return p2l_check_launch();
}
extern "C" int p2l_sg2_noise_relayout(const P2LStyleGAN2* m, const float* src, float* dst, int Bn,
                                      int to_layer_major, void* stream) {
  if (!m || !src || !dst || Bn < 1 || m->n_conv < 1 || m->n_conv > P2L_SG2_MAX_CONVS) return P2L_EINVAL;
  NoiseLayoutK k{};
  k.n_layers = m->n_conv; k.Bn = Bn;
  for (int l = 0; l < m->n_conv; ++l) {
    k.off[l] = (int)m->conv[l].noise_off;
    if (k.off[l] % 4 || (l && k.off[l] <= k.off[l - 1])) return P2L_EINVAL;
  }
  k.total = (int)m->noise_total;
  k.off[m->n_conv] = k.total;
  if (k.total % 4 || k.total <= k.off[m->n_conv - 1]) return P2L_EINVAL;
  hipLaunchKernelGGL(noise_relayout_kernel, dim3(cdiv((size_t)Bn * k.total / 4, 256)), dim3(256), 0, ST(stream),
                     src, dst, k, to_layer_major);
  return p2l_check_launch();
}
extern "C" int p2l_broadcast_rows(const float* src, float* dst, int64_t n, int Bn, void* stream) {
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3(cdiv(n * Bn, 256)), dim3(256), 0, ST(stream), src, dst,
                     (size_t)n, Bn);
  return p2l_check_launch();
}
extern "C" int p2l_add_inplace(float* a, const float* b, int64_t n, void* stream) {
  hipLaunchKernelGGL(add_inplace_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), a, b, (size_t)n);
  return p2l_check_launch();
}
