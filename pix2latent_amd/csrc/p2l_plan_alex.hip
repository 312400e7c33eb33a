// ProjectionLoss with the LPIPS-AlexNet network (the reference default,
// pix2latent/loss_functions.py:86-100 `ProjectionLoss(lpips_net='alex', beta=10)`,
// :126-148 PerceptualLoss -> lpips.LPIPS(net='alex', spatial=True)).  Same structure as
// the VGG16 plan in p2l_plan.hip: target features / adjoint-resized weight maps cached per
// target, weighted spatial sum evaluated at tap resolution, dgrad-only backward.
#include "p2l_common.h"

namespace {

struct Arena {
  size_t off = 0;
  size_t take(size_t n) {
    const size_t o = off;
    off += (n + 63) & ~(size_t)63;
    return o;
  }
};

#define RET_IF(x)          \
  do {                     \
    int _rc = (x);         \
    if (_rc) return _rc;   \
  } while (0)

const int kCin[5] = {16, 64, 192, 384, 256};
const int kCout[5] = {64, 192, 384, 256, 256};
const int kK[5] = {11, 5, 3, 3, 3};
const int kS[5] = {4, 1, 1, 1, 1};
const int kP[5] = {2, 2, 1, 1, 1};

struct AxLayout {
  int h[5], w[5];          // output grid of conv i
  int hp[2], wp[2];        // pooled grids (input of conv 1 and conv 2)
  size_t y[5], p[2], tgt16, wsrc, part, lp, l1, gs, ga, gb, gtap, total;
};

int ax_layout(int B, int H, int W, AxLayout& L) {
  if (B < 1 || H < 35 || W < 35) return P2L_EINVAL;     // two 3/2 pools behind an 11/4 conv
  L.h[0] = (H + 4 - 11) / 4 + 1; L.w[0] = (W + 4 - 11) / 4 + 1;
  L.hp[0] = (L.h[0] - 3) / 2 + 1; L.wp[0] = (L.w[0] - 3) / 2 + 1;
  L.h[1] = L.hp[0]; L.w[1] = L.wp[0];
  L.hp[1] = (L.h[1] - 3) / 2 + 1; L.wp[1] = (L.w[1] - 3) / 2 + 1;
  for (int i = 2; i < 5; ++i) { L.h[i] = L.hp[1]; L.w[i] = L.wp[1]; }
  if (L.hp[1] < 1 || L.wp[1] < 1) return P2L_EINVAL;
  Arena a;
  size_t max_act = 0;
  for (int i = 0; i < 5; ++i) {
    const size_t n = (size_t)B * L.h[i] * L.w[i] * kCout[i];
    L.y[i] = a.take(n);
    if (n > max_act) max_act = n;
  }
  L.p[0] = a.take((size_t)B * L.hp[0] * L.wp[0] * kCout[0]);
  L.p[1] = a.take((size_t)B * L.hp[1] * L.wp[1] * kCout[1]);
  L.tgt16 = a.take((size_t)B * H * W * 16);
  L.wsrc = a.take((size_t)B * H * W);
  size_t maxpart = (size_t)p2l_l1_loss_nblk(H, W);
  for (int k = 0; k < 5; ++k) {
    const size_t nb = (size_t)p2l_lpips_tap_nblk(L.h[k] * L.w[k], kCout[k]);
    if (nb > maxpart) maxpart = nb;
  }
  L.part = a.take((size_t)B * maxpart);
  L.lp = a.take(B);
  L.l1 = a.take(B);
  L.gs = a.take(B);
  L.ga = a.take(max_act);
  L.gb = a.take(max_act);
  L.gtap = a.take(max_act);
  L.total = a.off;
  return P2L_OK;
}

P2LGConv mk(int B, int Hi, int Wi, int Cin, int Cout, int K, int S, int P) {
  P2LGConv d{};
  d.B = B; d.Hi = Hi; d.Wi = Wi; d.Cin = Cin; d.Cout = Cout; d.KH = K; d.KW = K; d.stride = S;
  d.pad = P; d.x_ld = Cin; d.y_ld = Cout; d.res_ld = Cout; d.mask_ld = Cout;
  return d;
}

// AlexNet features on an NHWC16 image: y[i] = relu(conv_i), p[j] = pooled inputs
int alex_forward(const P2LAlexLpips* v, const float* img16, int B, int H, int W, float* Wk,
                 const AxLayout& L, void* st) {
  {
    P2LGConv d = mk(B, H, W, 16, 64, 11, 4, 2);
    d.relu = 1;
    RET_IF(p2l_gconv_fwd(&d, img16, v->w[0], v->b[0], v->in_s, v->in_t, nullptr, nullptr,
                         Wk + L.y[0], st));
  }
  RET_IF(p2l_maxpool3s2_fwd(Wk + L.y[0], Wk + L.p[0], B, L.h[0], L.w[0], 64, st));
  {
    P2LGConv d = mk(B, L.hp[0], L.wp[0], 64, 192, 5, 1, 2);
    d.relu = 1;
    RET_IF(p2l_gconv_fwd(&d, Wk + L.p[0], v->w[1], v->b[1], nullptr, nullptr, nullptr, nullptr,
                         Wk + L.y[1], st));
  }
  RET_IF(p2l_maxpool3s2_fwd(Wk + L.y[1], Wk + L.p[1], B, L.h[1], L.w[1], 192, st));
  const float* x = Wk + L.p[1];
  for (int i = 2; i < 5; ++i) {
    P2LGConv d = mk(B, L.h[i], L.w[i], kCin[i], kCout[i], 3, 1, 1);
    d.relu = 1;
    RET_IF(p2l_gconv_fwd(&d, x, v->w[i], v->b[i], nullptr, nullptr, nullptr, nullptr,
                         Wk + L.y[i], st));
    x = Wk + L.y[i];
  }
  return P2L_OK;
}

}  // namespace

extern "C" size_t p2l_alex_cache_floats(int B, int H, int W, size_t nft_off[5], size_t wt_off[5],
                                        size_t* wsum_off) {
  AxLayout L;
  if (ax_layout(B, H, W, L)) return 0;
  Arena a;
  for (int k = 0; k < 5; ++k) {
    const size_t P = (size_t)L.h[k] * L.w[k];
    nft_off[k] = a.take((size_t)B * P * kCout[k]);
    wt_off[k] = a.take((size_t)B * P);
  }
  *wsum_off = a.take(B);
  return a.off;
}

extern "C" size_t p2l_alexloss_ws_bytes(int B, int H, int W) {
  AxLayout L;
  if (ax_layout(B, H, W, L)) return 0;
  return L.total * sizeof(float);
}

extern "C" int p2l_alexloss_prepare(const P2LAlexLpips* v, const float* target,
                                    const float* weight, const float* loss_mask, int B, int H,
                                    int W, const P2LLossCache* cache, void* ws, size_t ws_bytes,
                                    void* st) {
  AxLayout L;
  RET_IF(ax_layout(B, H, W, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !cache || !target) return P2L_EWS;
  float* Wk = (float*)ws;
  if (weight) {
    RET_IF(p2l_weight_sum(weight, loss_mask, cache->wsum, B, 3 * H * W, st));
    RET_IF(p2l_weight_map(weight, loss_mask, Wk + L.wsrc, B, H, W, st));
    for (int k = 0; k < 5; ++k)
      RET_IF(p2l_bilinear_adjoint(Wk + L.wsrc, cache->wt[k], B, H, W, L.h[k], L.w[k], st));
  }
  if (v) {
    RET_IF(p2l_nchw3_to_nhwc16(target, Wk + L.tgt16, B, H, W, st));
    RET_IF(alex_forward(v, Wk + L.tgt16, B, H, W, Wk, L, st));
    for (int k = 0; k < 5; ++k)
      RET_IF(p2l_lpips_normalize(Wk + L.y[k], cache->nft[k], (int64_t)B * L.h[k] * L.w[k],
                                 kCout[k], st));
  }
  return P2L_OK;
}

extern "C" int p2l_alexloss_fwd(const P2LAlexLpips* v, const float* img16, const float* target,
                                const float* weight, const float* loss_mask,
                                const P2LLossCache* cache, float beta, int use_lpips, int B,
                                int H, int W, void* ws, size_t ws_bytes, float* loss,
                                float* loss_l1, float* loss_lpips, void* st) {
  AxLayout L;
  RET_IF(ax_layout(B, H, W, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !cache || !img16 || !loss) return P2L_EWS;
  float* Wk = (float*)ws;
  float* l1 = loss_l1 ? loss_l1 : Wk + L.l1;
  float* lp = loss_lpips ? loss_lpips : Wk + L.lp;
  RET_IF(p2l_l1_loss_fwd(img16, target, weight, loss_mask, cache->wsum, l1, Wk + L.part, B, H, W,
                         st));
  RET_IF(p2l_vec_scale_div(l1, nullptr, loss, B, 1.f, st));
  if (use_lpips) {
    if (!v) return P2L_EINVAL;
    RET_IF(alex_forward(v, img16, B, H, W, Wk, L, st));
    for (int k = 0; k < 5; ++k) {
      const int P = L.h[k] * L.w[k], C = kCout[k];
      const int nblk = p2l_lpips_tap_nblk(P, C);
      RET_IF(p2l_lpips_tap_fwd(Wk + L.y[k], cache->nft[k], (int64_t)P * C, v->lin[k], cache->wt[k],
                               P, Wk + L.part, B, P, C, st));
      RET_IF(p2l_reduce_rows(Wk + L.part, lp, B, nblk, 1.f, cache->wsum, k > 0, st));
    }
    RET_IF(p2l_reduce_rows(lp, loss, B, 1, beta, nullptr, 1, st));
  }
  return P2L_OK;
}

extern "C" int p2l_alexloss_bwd(const P2LAlexLpips* v, const float* img16, const float* target,
                                const float* weight, const float* loss_mask,
                                const P2LLossCache* cache, float beta, int use_lpips,
                                const float* gloss, int B, int H, int W, void* ws,
                                size_t ws_bytes, float* dimg16, void* st) {
  AxLayout L;
  RET_IF(ax_layout(B, H, W, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !cache || !img16 || !gloss || !dimg16)
    return P2L_EWS;
  float* Wk = (float*)ws;
  if (!use_lpips)
    return p2l_l1_loss_bwd(img16, target, weight, loss_mask, cache->wsum, gloss, dimg16, B, H, W, 0,
                           st);
  if (!v) return P2L_EINVAL;
  RET_IF(p2l_vec_scale_div(gloss, cache->wsum, Wk + L.gs, B, beta, st));
  float* ga = Wk + L.ga;     // gradient w.r.t. the PRE-ReLU output of the current conv
  float* gb = Wk + L.gb;
  float* gtap = Wk + L.gtap;
  auto tap = [&](int k) {
    const int P = L.h[k] * L.w[k], C = kCout[k];
    return p2l_lpips_tap_bwd(Wk + L.y[k], cache->nft[k], (int64_t)P * C, v->lin[k], cache->wt[k],
                             P, Wk + L.gs, gtap, B, P, C, st);
  };
  // relu5 is only consumed by its tap
  RET_IF(tap(4));
  RET_IF(p2l_relu_mask(Wk + L.y[4], 256, gtap, 256, ga, 256, (int64_t)B * L.h[4] * L.w[4], 256, st));
  // conv4, conv3 (3x3 chain): dgrad conv + tap gradient as residual + ReLU mask of the input
  for (int i = 4; i >= 3; --i) {
    RET_IF(tap(i - 1));
    P2LGConv d = mk(B, L.h[i], L.w[i], kCout[i], kCin[i], 3, 1, 1);
    RET_IF(p2l_gconv_fwd(&d, ga, v->wt[i], nullptr, nullptr, nullptr, gtap, Wk + L.y[i - 1], gb, st));
    float* t = ga; ga = gb; gb = t;
  }
  // conv2 -> pooled relu2 -> relu2 (+ tap) -> conv1 ...
  {
    P2LGConv d = mk(B, L.h[2], L.w[2], kCout[2], kCin[2], 3, 1, 1);
    RET_IF(p2l_gconv_fwd(&d, ga, v->wt[2], nullptr, nullptr, nullptr, nullptr, nullptr, gb, st));
    RET_IF(tap(1));
    RET_IF(p2l_maxpool3s2_bwd(Wk + L.y[1], gb, gtap, ga, B, L.h[1], L.w[1], 192, st));
  }
  {
    P2LGConv d = mk(B, L.h[1], L.w[1], kCout[1], kCin[1], 5, 1, 2);
    RET_IF(p2l_gconv_fwd(&d, ga, v->wt[1], nullptr, nullptr, nullptr, nullptr, nullptr, gb, st));
    RET_IF(tap(0));
    RET_IF(p2l_maxpool3s2_bwd(Wk + L.y[0], gb, gtap, ga, B, L.h[0], L.w[0], 64, st));
  }
  RET_IF(p2l_conv1_dgrad(ga, v->wt[0], dimg16, B, H, W, 64, 11, 4, 2, st));
  if (use_lpips == 2) return P2L_OK;   // PerceptualLoss on its own: no L1 term
  return p2l_l1_loss_bwd(img16, target, weight, loss_mask, cache->wsum, gloss, dimg16, B, H, W, 1,
                         st);
}
