// Kernels of the LPIPS-AlexNet feature network (reference call site
// pix2latent/loss_functions.py:87,131 `lpips.LPIPS(net='alex', spatial=True)`, the default
// of ProjectionLoss).  AlexNet's layers do not fit the tiled 3x3/1x1 kernel of p2l_conv.hip
// (11x11 stride 4, 5x5, odd 63/31/15 grids, overlapping 3x3/2 max-pools) and are cheap
// (0.87 GMAC per 256^2 image, < 3 % of a candidate evaluation), so they get ONE generic
// gather-based implicit-GEMM kernel on the same exact-fp32 MFMA instead of tuned variants.
#include "p2l_common.h"

#define ST(s) ((hipStream_t)(s))

namespace {

struct GConvK {
  const float* x; const float* w; const float* bias; const float* res; const float* mask;
  const float* pro_s; const float* pro_t;
  float* y;
  int B, Hi, Wi, Cin, x_ld;
  int Ho, Wo, Cout, y_ld, res_ld, mask_ld, n_store;
  int KH, KW, stride, pad, relu;
  int M;
};

constexpr int GP = 20;   // LDS row pitch in floats (16 + 4: conflict-free b128 reads)

// M = B*Ho*Wo output pixels x N = Cout x K = KH*KW*Cin.  Block = 64 pixels x 64 channels,
// 4 waves in a 2x2 grid of 32x32 accumulators; one K-chunk = 16 input channels of one tap.
// A rows are gathered (arbitrary stride / padding), B rows come from the packed
// [tap][Cin/16][Cout][16] weights; next chunk's global loads are in flight during the MFMAs.
__global__ __launch_bounds__(256) void gconv_mfma_kernel(const GConvK k) {
  __shared__ float As[64 * GP];
  __shared__ float Bs[64 * GP];
  const int tid = threadIdx.x, r = tid >> 2, q = tid & 3;
  const int lane = tid & 63, wave = tid >> 6, wm = wave & 1, wn = wave >> 1;
  const int row = lane & 31, h = lane >> 5;

  const int m = blockIdx.x * 64 + r;
  const bool mv = m < k.M;
  const int mm = mv ? m : 0;
  const int b = mm / (k.Ho * k.Wo);
  const int rem = mm - b * (k.Ho * k.Wo);
  const int oy = rem / k.Wo, ox = rem - oy * k.Wo;
  const int iy0 = oy * k.stride - k.pad, ix0 = ox * k.stride - k.pad;
  const int n_row = blockIdx.y * 64 + r;
  const int nch = k.Cin >> 4;
  const int nchunks = k.KH * k.KW * nch;

  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 ra, rb;
  auto load = [&](int ci) {
    const int tap = ci / nch, cc = ci - tap * nch;
    const int ky = tap / k.KW, kx = tap - ky * k.KW;
    const int iy = iy0 + ky, ix = ix0 + kx;
    const bool ok = mv && iy >= 0 && iy < k.Hi && ix >= 0 && ix < k.Wi;
    const int cy = min(max(iy, 0), k.Hi - 1), cx = min(max(ix, 0), k.Wi - 1);
    const int c = cc * 16 + q * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(k.x + (((size_t)b * k.Hi + cy) * k.Wi + cx) * k.x_ld + c);
    if (k.pro_s) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(k.pro_s + c);
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(k.pro_t + c);
      v = v * s4 + t4;
    }
    const f32x4 z = {0, 0, 0, 0};
    ra = ok ? v : z;
    rb = *reinterpret_cast<const f32x4*>(k.w + (((size_t)tap * nch + cc) * k.Cout + n_row) * 16 + q * 4);
  };
  load(0);
  for (int ci = 0; ci < nchunks; ++ci) {
    *reinterpret_cast<f32x4*>(As + r * GP + q * 4) = ra;
    *reinterpret_cast<f32x4*>(Bs + r * GP + q * 4) = rb;
    __syncthreads();
    if (ci + 1 < nchunks) load(ci + 1);
    const float* ap = As + (wm * 32 + row) * GP + h * 8;
    const float* bp = Bs + (wn * 32 + row) * GP + h * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(ap + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bp + 4);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
    __syncthreads();
  }
  // C layout: column (channel) = lane & 31, row (pixel) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int n = blockIdx.y * 64 + wn * 32 + row;
  if (n >= k.n_store) return;
  const float bv = k.bias ? k.bias[n] : 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const int i = (g & 3) + 8 * (g >> 2) + 4 * h;
    const int mo = blockIdx.x * 64 + wm * 32 + i;
    if (mo >= k.M) continue;
    float v = acc[g] + bv;
    if (k.res) v += k.res[(size_t)mo * k.res_ld + n];
    if (k.relu) v = fmaxf(v, 0.f);
    if (k.mask) v = (k.mask[(size_t)mo * k.mask_ld + n] > 0.f) ? v : 0.f;
    k.y[(size_t)mo * k.y_ld + n] = v;
  }
}

// 3x3 stride-2 max-pool (no padding), NHWC, C % 4 == 0
__global__ void maxpool3s2_fwd_kernel(const float* x, float* y, int Bn, int Hi, int Wi, int Ho,
                                      int Wo, int C) {
  const int C4 = C >> 2;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)Bn * Ho * Wo * C4) return;
  const int c = (int)(idx % C4) * 4;
  size_t p = idx / C4;
  const int ox = (int)(p % Wo); p /= Wo;
  const int oy = (int)(p % Ho);
  const int b = (int)(p / Ho);
  f32x4 m = *reinterpret_cast<const f32x4*>(x + (((size_t)b * Hi + 2 * oy) * Wi + 2 * ox) * C + c);
#pragma unroll
  for (int t = 1; t < 9; ++t) {
    const int dy = t / 3, dx = t - dy * 3;
    const f32x4 v = *reinterpret_cast<const f32x4*>(
        x + (((size_t)b * Hi + 2 * oy + dy) * Wi + 2 * ox + dx) * C + c);
    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
  }
  *reinterpret_cast<f32x4*>(y + idx * 4) = m;
}

// backward of the pool fused with what follows in the LPIPS backward:
//   dx = ( sum over the <= 4 windows containing the pixel [pixel is the window's first
//          maximum in scan order] * gp[window]  +  gtap ) * (x > 0)
// gather form -> no atomics, deterministic
__global__ void maxpool3s2_bwd_kernel(const float* x, const float* gp, const float* gtap,
                                      float* dx, int Bn, int Hi, int Wi, int Ho, int Wo, int C) {
  const int C4 = C >> 2;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)Bn * Hi * Wi * C4) return;
  const int c = (int)(idx % C4) * 4;
  size_t p = idx / C4;
  const int ix = (int)(p % Wi); p /= Wi;
  const int iy = (int)(p % Hi);
  const int b = (int)(p / Hi);
  const f32x4 me = *reinterpret_cast<const f32x4*>(x + idx * 4);
  f32x4 g = gtap ? *reinterpret_cast<const f32x4*>(gtap + idx * 4) : f32x4{0, 0, 0, 0};
  const int oy_lo = max(0, (iy - 1) >> 1), oy_hi = min(Ho - 1, iy >> 1);
  const int ox_lo = max(0, (ix - 1) >> 1), ox_hi = min(Wo - 1, ix >> 1);
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    if (iy - 2 * oy > 2) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      if (ix - 2 * ox > 2) continue;
      const int my = (iy - 2 * oy) * 3 + (ix - 2 * ox);     // my scan position in the window
      bool wx = true, wy = true, wz = true, ww = true;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t == my) continue;
        const int dy = t / 3, dxx = t - dy * 3;
        const f32x4 v = *reinterpret_cast<const f32x4*>(
            x + (((size_t)b * Hi + 2 * oy + dy) * Wi + 2 * ox + dxx) * C + c);
        // earlier positions win ties, later ones lose them
        if (t < my) { wx &= me.x > v.x; wy &= me.y > v.y; wz &= me.z > v.z; ww &= me.w > v.w; }
        else        { wx &= me.x >= v.x; wy &= me.y >= v.y; wz &= me.z >= v.z; ww &= me.w >= v.w; }
      }
      const f32x4 gw = *reinterpret_cast<const f32x4*>(gp + (((size_t)b * Ho + oy) * Wo + ox) * C + c);
      if (wx) g.x += gw.x;
      if (wy) g.y += gw.y;
      if (wz) g.z += gw.z;
      if (ww) g.w += gw.w;
    }
  }
  g.x = me.x > 0.f ? g.x : 0.f; g.y = me.y > 0.f ? g.y : 0.f;
  g.z = me.z > 0.f ? g.z : 0.f; g.w = me.w > 0.f ? g.w : 0.f;
  *reinterpret_cast<f32x4*>(dx + idx * 4) = g;
}

// input gradient of the stride-S KxK first conv to the 3 image channels (direct form: per
// image pixel only ceil(K/S)^2 taps are live).  w: [K*K][3][Co] with the LPIPS 1/scale
// folded in; g: [B,Ho,Wo,Co] (already ReLU-masked); dimg16: [B,H,W,16] (ch 3.. zeroed)
__global__ void conv1_dgrad_kernel(const float* g, const float* w, float* dimg16, int Bn, int H,
                                   int W, int Ho, int Wo, int Co, int K, int S, int pad) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)Bn * H * W) return;
  const int ix = (int)(idx % W);
  const int iy = (int)((idx / W) % H);
  const int b = (int)(idx / ((size_t)W * H));
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int ky = (iy + pad) % S; ky < K; ky += S) {
    const int oy = (iy + pad - ky) / S;
    if (iy + pad - ky < 0 || oy >= Ho) continue;
    for (int kx = (ix + pad) % S; kx < K; kx += S) {
      const int ox = (ix + pad - kx) / S;
      if (ix + pad - kx < 0 || ox >= Wo) continue;
      const float* gp = g + (((size_t)b * Ho + oy) * Wo + ox) * Co;
      const float* wp = w + (size_t)(ky * K + kx) * 3 * Co;
      for (int co = 0; co < Co; co += 4) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gp + co);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp + co);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wp + Co + co);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(wp + 2 * Co + co);
        a0 += (gv.x * w0.x + gv.y * w0.y) + (gv.z * w0.z + gv.w * w0.w);
        a1 += (gv.x * w1.x + gv.y * w1.y) + (gv.z * w1.z + gv.w * w1.w);
        a2 += (gv.x * w2.x + gv.y * w2.y) + (gv.z * w2.z + gv.w * w2.w);
      }
    }
  }
  f32x4* dp = reinterpret_cast<f32x4*>(dimg16 + idx * 16);
  const f32x4 z = {0, 0, 0, 0};
  dp[0] = f32x4{a0, a1, a2, 0.f}; dp[1] = z; dp[2] = z; dp[3] = z;
}

}  // namespace

extern "C" int p2l_gconv_fwd(const P2LGConv* d, const float* x, const float* w, const float* bias,
                             const float* pro_s, const float* pro_t, const float* res,
                             const float* mask, float* y, void* stream) {
  if (!d || !x || !w || !y) return P2L_EINVAL;
  if (d->Cin % 16 || d->Cout % 64 || d->x_ld % 4 || d->x_ld < d->Cin) return P2L_EINVAL;
  if (d->KH < 1 || d->KW < 1 || d->stride < 1 || d->pad < 0 || d->B < 1) return P2L_EINVAL;
  const int Ho = (d->Hi + 2 * d->pad - d->KH) / d->stride + 1;
  const int Wo = (d->Wi + 2 * d->pad - d->KW) / d->stride + 1;
  if (Ho < 1 || Wo < 1) return P2L_EINVAL;
  if ((pro_s == nullptr) != (pro_t == nullptr)) return P2L_EINVAL;
  const int64_t M = (int64_t)d->B * Ho * Wo;
  if (M * (d->y_ld > d->Cout ? d->y_ld : d->Cout) >= ((int64_t)1 << 31)) return P2L_EUNSUP;
  GConvK k{};
  k.x = x; k.w = w; k.bias = bias; k.res = res; k.mask = mask; k.pro_s = pro_s; k.pro_t = pro_t;
  k.y = y; k.B = d->B; k.Hi = d->Hi; k.Wi = d->Wi; k.Cin = d->Cin; k.x_ld = d->x_ld;
  k.Ho = Ho; k.Wo = Wo; k.Cout = d->Cout; k.y_ld = d->y_ld; k.res_ld = d->res_ld;
  k.mask_ld = d->mask_ld; k.n_store = d->n_store > 0 ? d->n_store : d->Cout;
  k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad; k.relu = d->relu;
  k.M = (int)M;
  hipLaunchKernelGGL(gconv_mfma_kernel, dim3(cdiv(M, 64), d->Cout / 64), dim3(256), 0, ST(stream), k);
  return p2l_check_launch();
}

extern "C" int p2l_maxpool3s2_fwd(const float* x, float* y, int Bn, int Hi, int Wi, int C,
                                  void* stream) {
  if (C % 4 || Hi < 3 || Wi < 3) return P2L_EINVAL;
  const int Ho = (Hi - 3) / 2 + 1, Wo = (Wi - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(cdiv((size_t)Bn * Ho * Wo * (C / 4), 256)), dim3(256),
                     0, ST(stream), x, y, Bn, Hi, Wi, Ho, Wo, C);
  return p2l_check_launch();
}

extern "C" int p2l_maxpool3s2_bwd(const float* x, const float* gpooled, const float* gtap, float* dx,
                                  int Bn, int Hi, int Wi, int C, void* stream) {
  if (C % 4 || Hi < 3 || Wi < 3) return P2L_EINVAL;
  const int Ho = (Hi - 3) / 2 + 1, Wo = (Wi - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(cdiv((size_t)Bn * Hi * Wi * (C / 4), 256)), dim3(256),
                     0, ST(stream), x, gpooled, gtap, dx, Bn, Hi, Wi, Ho, Wo, C);
  return p2l_check_launch();
}

extern "C" int p2l_conv1_dgrad(const float* g, const float* w_t3, float* dimg16, int Bn, int H, int W,
                               int Co, int K, int S, int pad, void* stream) {
  if (Co % 4 || K < 1 || S < 1) return P2L_EINVAL;
  const int Ho = (H + 2 * pad - K) / S + 1, Wo = (W + 2 * pad - K) / S + 1;
  hipLaunchKernelGGL(conv1_dgrad_kernel, dim3(cdiv((size_t)Bn * H * W, 256)), dim3(256), 0, ST(stream),
                     g, w_t3, dimg16, Bn, H, W, Ho, Wo, Co, K, S, pad);
  return p2l_check_launch();
}
