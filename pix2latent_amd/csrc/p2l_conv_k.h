// Shared by the conv kernels (p2l_conv.hip, p2l_conv2.hip): kernel argument block and
// the fused epilogue.
#pragma once
#include "p2l_common.h"

namespace p2lconv {

struct ConvK {
  const float* x;
  const float* w;
  const float* bias;
  const float* pro_s;
  const float* pro_t;
  const float* res;
  const float* mask;
  float* y;
  float* yp;
  float* ws;
  int B, H, W, Cin, Cout;
  int x_ld, y_ld, yp_ld, res_ld, mask_ld, n_store;
  int pro_bstride;
  float alpha;
  int act, pool, res_ups, ups;
  int nchunks, chunks_per_split, splitk;
  int tw_log, th_log, tb_log;
  int tiles_x_log, tiles_y_log;
  int n_mtiles, n_ntiles;
  // general (non power-of-two) M grid: k.H x k.W grid points, tiles_x x tiles_y tiles per
  // image; input coordinates are valid in [0,iH) x [0,iW) and live in an ibH x ibW pixel
  // buffer; outputs go to an obH x obW pixel buffer.  partial != 0 when the grid is not a
  // multiple of the tile (border sub-pixels are skipped in the epilogue).
  int tiles_x, tiles_y, iH, iW, ibH, ibW, obH, obW, partial;
  // StyleGAN2 epilogue terms: v = acc * oscale[b][n] + noise_w * noise[b][pixel] + bias[n]
  const float* oscale; const float* noise; float noise_w; int oscale_bstride;
  // fused backward of a = max(x*s+t, 0) applied to the conv result (dgrad epilogue)
  const float* arb_x; const float* arb_s; const float* arb_t; const float* arb_skip;
  float* arb_partial;
  int arb_x_ld, arb_bstride, arb_skip_ld, arb_skip_C, arb_skip_ups, arb_nblk;
  int arb_nomask;   // 1: plain scale backward (g = da), StyleGAN2 modulation
  // sub-pixel mode of the TAPS=4 kernel (nearest-x2 upsample folded into the weights):
  //   1 = forward: low-res input, 4 output phases (blockIdx.y), output stride 2
  //   2 = input-gradient: the 4 phase planes of the high-res dY are 4 K-slices
  int sp_mode, sp_ncc;   // sp_ncc = channel chunks per phase plane (mode 2)
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == P2L_ACT_RELU) return fmaxf(v, 0.f);
  if (act == P2L_ACT_TANH) return tanhf(v);
  return v;
}

// Epilogue for one quad (4 sub-pixels) of one output channel.
// pix0 = linear index of the quad's top-left pixel ((b*H + oy0)*W + ox0); all
// element offsets fit in 32 bits (B*H*W*ld < 2^31 is checked on the host).
// SIMPLE = no residual / mask / pool / second output: the common conv->conv case.
template <bool SIMPLE>
__device__ __forceinline__ void epilogue_quad(const ConvK& k, const float a[4], int pix0,
                                              int b, int oy0, int ox0, int n,
                                              float bias_n) {
  const int W = k.W;
  const int sub[4] = {0, 1, W, W + 1};
  if (SIMPLE) {
    float* yp = k.y + (size_t)((unsigned)pix0 * (unsigned)k.y_ld + (unsigned)n);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float t = apply_act(k.alpha * a[s] + bias_n, k.act);
      yp[(unsigned)sub[s] * (unsigned)k.y_ld] = t;
    }
    return;
  }
  float v[4];
  int rp0 = pix0;
  if (k.res && k.res_ups)
    rp0 = (b * (k.H >> 1) + (oy0 >> 1)) * (W >> 1) + (ox0 >> 1);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned pix = (unsigned)(pix0 + sub[s]);
    float t = k.alpha * a[s] + bias_n;
    if (k.res) {
      const unsigned rp = k.res_ups ? (unsigned)rp0 : pix;
      t += k.res[(size_t)(rp * (unsigned)k.res_ld + (unsigned)n)];
    }
    t = apply_act(t, k.act);
    if (k.mask) t = (k.mask[(size_t)(pix * (unsigned)k.mask_ld + (unsigned)n)] > 0.f) ? t : 0.f;
    if (k.y) k.y[(size_t)(pix * (unsigned)k.y_ld + (unsigned)n)] = t;
    v[s] = t;
  }
  if (k.pool) {
    float p;
    if (k.pool == P2L_POOL_MAX)
      p = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    else
      p = (v[0] + v[1]) + (v[2] + v[3]);
    const unsigned pp = (unsigned)((b * (k.H >> 1) + (oy0 >> 1)) * (W >> 1) + (ox0 >> 1));
    k.yp[(size_t)(pp * (unsigned)k.yp_ld + (unsigned)n)] = p;
  }
}


}  // namespace p2lconv

// v2 (persistent, LDS double-buffered) 3x3 kernel, defined in p2l_conv2.hip
