// ProjectionLoss / PerceptualLoss with the LPIPS-SqueezeNet network (round 6).  The reference passes any lpips net
// name through (pix2latent/loss_functions.py:128-131 `lpips.LPIPS(net=net, spatial=True)`); 'squeeze' was the one
// of the three lpips ships that had no native plan.  torchvision squeezenet1_1.features as
// lpips.pretrained_networks.squeezenet slices it [3P-recall]:
//   conv(3,64,k3,s2) relu | pool fire(64,16,64) fire(128,16,64) | pool fire(128,32,128) fire(256,32,128) |
//   pool fire(256,48,192) | fire(384,48,192) | fire(384,64,256) | fire(512,64,256)         (7 taps)
//   Fire(in, s, e): x -> q = relu(conv1x1(in, s)) -> cat(relu(conv1x1(s, e)), relu(conv3x3(s, e, pad 1)))
//   pool = 3x3 stride-2 max-pool with ceil_mode: for the image sizes this plan accepts every window is whole
//   ((n - 3) even at each of the three pools: all powers of two >= 64), so it is p2l_maxpool3s2.
// Same structure as the AlexNet plan (p2l_plan_alex.hip): the generic gather conv (p2l_gconv_fwd, exact fp32
// MFMA), target features / adjoint-resized weight maps cached per target, weighted spatial sum at tap
// resolution, dgrad-only backward with fixed-order reductions.  The two expand branches of a fire write the two
// channel halves of one tensor; their input gradients meet in one launch (the second takes the first as residual
// and applies the squeeze ReLU's mask).
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

constexpr int NF = 8, NT = 7;
const int kIn[NF] = {64, 128, 128, 256, 256, 384, 384, 512};
const int kSq[NF] = {16, 16, 32, 32, 48, 48, 64, 64};
const int kEx[NF] = {64, 64, 128, 128, 192, 192, 256, 256};
const int kStage[NF] = {0, 0, 1, 1, 2, 2, 2, 2};       // which pooled grid a fire runs on
const int kTapOfFire[NF] = {-1, 1, -1, 2, 3, 4, 5, 6};  // LPIPS tap its output is (tap 0 = relu(conv0))
const int kTapC[NT] = {64, 128, 256, 384, 384, 512, 512};

struct SqLayout {
  int h0, w0;              // conv0 output grid
  int h[3], w[3];          // the three pooled grids
  size_t y0, p[3], q[NF], out[NF];
  size_t tgt16, wsrc, part, lp, l1, gs, ga, gb, gq, gq2, gtap, total;
  int tap_h[NT], tap_w[NT];
  size_t tap_y[NT];
};

int pool_ok(int n) { return n >= 3 && ((n - 3) & 1) == 0; }

int sq_layout(int B, int H, int W, SqLayout& L) {
  if (B < 1 || H < 35 || W < 35) return P2L_EINVAL;
  L.h0 = (H - 3) / 2 + 1; L.w0 = (W - 3) / 2 + 1;
  int hh = L.h0, ww = L.w0;
  for (int s = 0; s < 3; ++s) {
    // ceil_mode pools: only sizes whose last window is whole (floor == ceil) are taken
    if (!pool_ok(hh) || !pool_ok(ww)) return P2L_EINVAL;
    hh = (hh - 3) / 2 + 1; ww = (ww - 3) / 2 + 1;
    L.h[s] = hh; L.w[s] = ww;
  }
  Arena a;
  size_t max_act = (size_t)B * L.h0 * L.w0 * 64;
  L.y0 = a.take(max_act);
  const int pc[3] = {64, 128, 256};
  for (int s = 0; s < 3; ++s) L.p[s] = a.take((size_t)B * L.h[s] * L.w[s] * pc[s]);
  size_t max_q = 0;
  for (int i = 0; i < NF; ++i) {
    const size_t px = (size_t)B * L.h[kStage[i]] * L.w[kStage[i]];
    L.q[i] = a.take(px * kSq[i]);
    L.out[i] = a.take(px * 2 * kEx[i]);
    if (px * 2 * kEx[i] > max_act) max_act = px * 2 * kEx[i];
    if (px * kIn[i] > max_act) max_act = px * kIn[i];
    if (px * kSq[i] > max_q) max_q = px * kSq[i];
  }
  L.tap_h[0] = L.h0; L.tap_w[0] = L.w0; L.tap_y[0] = L.y0;
  for (int i = 0; i < NF; ++i)
    if (kTapOfFire[i] >= 0) {
      const int k = kTapOfFire[i];
      L.tap_h[k] = L.h[kStage[i]]; L.tap_w[k] = L.w[kStage[i]]; L.tap_y[k] = L.out[i];
    }
  L.tgt16 = a.take((size_t)B * H * W * 16);
  L.wsrc = a.take((size_t)B * H * W);
  size_t maxpart = (size_t)p2l_l1_loss_nblk(H, W);
  for (int k = 0; k < NT; ++k) {
    const size_t nb = (size_t)p2l_lpips_tap_nblk(L.tap_h[k] * L.tap_w[k], kTapC[k]);
    if (nb > maxpart) maxpart = nb;
  }
  L.part = a.take((size_t)B * maxpart);
  L.lp = a.take(B);
  L.l1 = a.take(B);
  L.gs = a.take(B);
  L.ga = a.take(max_act);
  L.gb = a.take(max_act);
  L.gtap = a.take(max_act);
  L.gq = a.take(max_q);
  L.gq2 = a.take(max_q);
  L.total = a.off;
  return P2L_OK;
}

P2LGConv mk(int B, int Hi, int Wi, int Cin, int Cout, int K, int S, int P) {
  P2LGConv d{};
  d.B = B; d.Hi = Hi; d.Wi = Wi; d.Cin = Cin; d.Cout = Cout; d.KH = K; d.KW = K; d.stride = S;
  d.pad = P; d.x_ld = Cin; d.y_ld = Cout; d.res_ld = Cout; d.mask_ld = Cout;
  return d;
}

// SqueezeNet1.1 features on an NHWC16 image: y0 = relu(conv0), q[i] / out[i] per fire, p[s] pooled inputs
int sq_forward(const P2LSqueezeLpips* v, const float* img16, int B, int H, int W, float* Wk, const SqLayout& L,
               void* st) {
  {
    P2LGConv d = mk(B, H, W, 16, 64, 3, 2, 0);
    d.relu = 1;
    RET_IF(p2l_gconv_fwd(&d, img16, v->w0, v->b0, v->in_s, v->in_t, nullptr, nullptr, Wk + L.y0, st));
  }
  const float* x = Wk + L.y0;
  int xh = L.h0, xw = L.w0;
  for (int i = 0; i < NF; ++i) {
    const int s = kStage[i], h = L.h[s], w = L.w[s];
    if (i == 0 || kStage[i - 1] != s) {                // a pool in front of the first fire of a stage
      RET_IF(p2l_maxpool3s2_fwd(x, Wk + L.p[s], B, xh, xw, kIn[i], st));
      x = Wk + L.p[s]; xh = h; xw = w;
    }
    {
      P2LGConv d = mk(B, h, w, kIn[i], 64, 1, 1, 0);   // squeeze: output channels padded to 64, sq stored
      d.relu = 1; d.n_store = kSq[i]; d.y_ld = kSq[i];
      RET_IF(p2l_gconv_fwd(&d, x, v->sq_w[i], v->sq_b[i], nullptr, nullptr, nullptr, nullptr, Wk + L.q[i], st));
    }
    {
      P2LGConv d = mk(B, h, w, kSq[i], kEx[i], 1, 1, 0);
      d.relu = 1; d.y_ld = 2 * kEx[i];
      RET_IF(p2l_gconv_fwd(&d, Wk + L.q[i], v->e1_w[i], v->e1_b[i], nullptr, nullptr, nullptr, nullptr,
                           Wk + L.out[i], st));
    }
    {
      P2LGConv d = mk(B, h, w, kSq[i], kEx[i], 3, 1, 1);
      d.relu = 1; d.y_ld = 2 * kEx[i];
      RET_IF(p2l_gconv_fwd(&d, Wk + L.q[i], v->e3_w[i], v->e3_b[i], nullptr, nullptr, nullptr, nullptr,
                           Wk + L.out[i] + kEx[i], st));
    }
    x = Wk + L.out[i];
  }
  return P2L_OK;
}

}  // namespace

extern "C" size_t p2l_sqz_cache_floats(int B, int H, int W, size_t nft_off[7], size_t wt_off[7],
                                       size_t* wsum_off) {
  SqLayout L;
  if (sq_layout(B, H, W, L)) return 0;
  Arena a;
  for (int k = 0; k < NT; ++k) {
    const size_t P = (size_t)L.tap_h[k] * L.tap_w[k];
    nft_off[k] = a.take((size_t)B * P * kTapC[k]);
    wt_off[k] = a.take((size_t)B * P);
  }
  *wsum_off = a.take(B);
  return a.off;
}

// test hook (include/p2l_test.h): idx 0 = relu(conv0), 1..8 = the squeeze outputs q of fires 0..7,
// 9..16 = the (concatenated) outputs of fires 0..7; NHWC
extern "C" int p2l_sqzloss_ws_lookup(int B, int H, int W, int idx, size_t* float_off, int32_t shape[4]) {
  if (idx < 0 || idx > 2 * NF || !float_off || !shape) return P2L_EINVAL;
  SqLayout L;
  RET_IF(sq_layout(B, H, W, L));
  shape[0] = B;
  if (idx == 0) {
    *float_off = L.y0; shape[1] = L.h0; shape[2] = L.w0; shape[3] = 64;
    return P2L_OK;
  }
  const int i = (idx - 1) % NF, s = kStage[i];
  shape[1] = L.h[s]; shape[2] = L.w[s];
  if (idx <= NF) { *float_off = L.q[i]; shape[3] = kSq[i]; }
  else { *float_off = L.out[i]; shape[3] = 2 * kEx[i]; }
  return P2L_OK;
}

extern "C" size_t p2l_sqzloss_ws_bytes(int B, int H, int W) {
  SqLayout L;
  if (sq_layout(B, H, W, L)) return 0;
  return L.total * sizeof(float);
}

extern "C" int p2l_sqzloss_prepare(const P2LSqueezeLpips* v, const float* target, const float* weight,
                                   const float* loss_mask, int B, int H, int W, const P2LLossCache7* cache,
                                   void* ws, size_t ws_bytes, void* st) {
  SqLayout L;
  RET_IF(sq_layout(B, H, W, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !cache || !target) return P2L_EWS;
  float* Wk = (float*)ws;
  if (weight) {
    RET_IF(p2l_weight_sum(weight, loss_mask, cache->wsum, B, 3 * H * W, st));
    RET_IF(p2l_weight_map(weight, loss_mask, Wk + L.wsrc, B, H, W, st));
    for (int k = 0; k < NT; ++k)
      RET_IF(p2l_bilinear_adjoint(Wk + L.wsrc, cache->wt[k], B, H, W, L.tap_h[k], L.tap_w[k], st));
  }
  if (v) {
    RET_IF(p2l_nchw3_to_nhwc16(target, Wk + L.tgt16, B, H, W, st));
    RET_IF(sq_forward(v, Wk + L.tgt16, B, H, W, Wk, L, st));
    for (int k = 0; k < NT; ++k)
      RET_IF(p2l_lpips_normalize(Wk + L.tap_y[k], cache->nft[k], (int64_t)B * L.tap_h[k] * L.tap_w[k], kTapC[k],
                                 st));
  }
  return P2L_OK;
}

extern "C" int p2l_sqzloss_fwd(const P2LSqueezeLpips* v, const float* img16, const float* target,
                               const float* weight, const float* loss_mask, const P2LLossCache7* cache,
                               float beta, int use_lpips, int B, int H, int W, void* ws, size_t ws_bytes,
                               float* loss, float* loss_l1, float* loss_lpips, void* st) {
  SqLayout L;
  RET_IF(sq_layout(B, H, W, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !cache || !img16 || !loss) return P2L_EWS;
  float* Wk = (float*)ws;
  float* l1 = loss_l1 ? loss_l1 : Wk + L.l1;
  float* lp = loss_lpips ? loss_lpips : Wk + L.lp;
  RET_IF(p2l_l1_loss_fwd(img16, target, weight, loss_mask, cache->wsum, l1, Wk + L.part, B, H, W, st));
  RET_IF(p2l_vec_scale_div(l1, nullptr, loss, B, 1.f, st));
  if (use_lpips) {
    if (!v) return P2L_EINVAL;
    RET_IF(sq_forward(v, img16, B, H, W, Wk, L, st));
    for (int k = 0; k < NT; ++k) {
      const int P = L.tap_h[k] * L.tap_w[k], C = kTapC[k];
      const int nblk = p2l_lpips_tap_nblk(P, C);
      RET_IF(p2l_lpips_tap_fwd(Wk + L.tap_y[k], cache->nft[k], (int64_t)P * C, v->lin[k], cache->wt[k], P,
                               Wk + L.part, B, P, C, st));
      RET_IF(p2l_reduce_rows(Wk + L.part, lp, B, nblk, 1.f, cache->wsum, k > 0, st));
    }
    RET_IF(p2l_reduce_rows(lp, loss, B, 1, beta, nullptr, 1, st));
  }
  return P2L_OK;
}

extern "C" int p2l_sqzloss_bwd(const P2LSqueezeLpips* v, const float* img16, const float* target,
                               const float* weight, const float* loss_mask, const P2LLossCache7* cache,
                               float beta, int use_lpips, const float* gloss, int B, int H, int W, void* ws,
                               size_t ws_bytes, float* dimg16, void* st) {
  SqLayout L;
  RET_IF(sq_layout(B, H, W, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !cache || !img16 || !gloss || !dimg16) return P2L_EWS;
  float* Wk = (float*)ws;
  if (!use_lpips)
    return p2l_l1_loss_bwd(img16, target, weight, loss_mask, cache->wsum, gloss, dimg16, B, H, W, 0, st);
  if (!v) return P2L_EINVAL;
  RET_IF(p2l_vec_scale_div(gloss, cache->wsum, Wk + L.gs, B, beta, st));
  float* ga = Wk + L.ga;     // M_i: gradient w.r.t. the PRE-ReLU outputs of fire i (both halves), masked
  float* gb = Wk + L.gb;
  float* gtap = Wk + L.gtap;
  float* gq = Wk + L.gq;
  float* gq2 = Wk + L.gq2;
  auto tap = [&](int k) {
    const int P = L.tap_h[k] * L.tap_w[k], C = kTapC[k];
    return p2l_lpips_tap_bwd(Wk + L.tap_y[k], cache->nft[k], (int64_t)P * C, v->lin[k], cache->wt[k], P,
                             Wk + L.gs, gtap, B, P, C, st);
  };
  // the last fire's output is consumed by its tap only
  RET_IF(tap(6));
  {
    const int i = NF - 1, C2 = 2 * kEx[i];
    RET_IF(p2l_relu_mask(Wk + L.out[i], C2, gtap, C2, ga, C2, (int64_t)B * L.h[2] * L.w[2], C2, st));
  }
  for (int i = NF - 1; i >= 0; --i) {
    const int s = kStage[i], h = L.h[s], w = L.w[s], C2 = 2 * kEx[i];
    // d q = dgrad(expand1x1)(M[:, :ex]) + dgrad(expand3x3)(M[:, ex:]), then the squeeze ReLU's mask
    {
      P2LGConv d = mk(B, h, w, kEx[i], 64, 1, 1, 0);
      d.x_ld = C2; d.n_store = kSq[i]; d.y_ld = kSq[i];
      RET_IF(p2l_gconv_fwd(&d, ga, v->e1_wt[i], nullptr, nullptr, nullptr, nullptr, nullptr, gq, st));
    }
    {
      P2LGConv d = mk(B, h, w, kEx[i], 64, 3, 1, 1);
      d.x_ld = C2; d.n_store = kSq[i]; d.y_ld = kSq[i]; d.res_ld = kSq[i]; d.mask_ld = kSq[i];
      RET_IF(p2l_gconv_fwd(&d, ga + kEx[i], v->e3_wt[i], nullptr, nullptr, nullptr, gq, Wk + L.q[i], gq2, st));
    }
    // d x = dgrad(squeeze)(d q): the gradient w.r.t. the fire's input tensor
    const bool pooled_in = (i == 0 || kStage[i - 1] != s);
    const int tk = (i > 0) ? kTapOfFire[i - 1] : 0;      // tap on the tensor in front (fire i-1's output | y0)
    if (tk >= 0) RET_IF(tap(tk));
    P2LGConv d = mk(B, h, w, kSq[i], kIn[i], 1, 1, 0);
    if (!pooled_in) {
      // the input IS fire i-1's output: + its tap gradient, x its ReLU mask, in the same launch
      RET_IF(p2l_gconv_fwd(&d, gq2, v->sq_wt[i], nullptr, nullptr, nullptr, tk >= 0 ? gtap : nullptr,
                           Wk + L.out[i - 1], gb, st));
    } else {
      // the input is the pooled tensor: gradient to it, then pool backward + tap gradient + ReLU mask in one pass
      RET_IF(p2l_gconv_fwd(&d, gq2, v->sq_wt[i], nullptr, nullptr, nullptr, nullptr, nullptr, gb, st));
      const float* src = (i == 0) ? Wk + L.y0 : Wk + L.out[i - 1];
      const int sh = (i == 0) ? L.h0 : L.h[s - 1], sw = (i == 0) ? L.w0 : L.w[s - 1];
      RET_IF(p2l_maxpool3s2_bwd(src, gb, tk >= 0 ? gtap : nullptr, ga, B, sh, sw, kIn[i], st));
      float* t = ga; ga = gb; gb = t;                    // (result in the old ga: swap so that it is `gb` below)
    }
    float* t = ga; ga = gb; gb = t;                      // M_{i-1} (or the masked gradient of y0) is now `ga`
  }
  RET_IF(p2l_conv1_dgrad(ga, v->wt0, dimg16, B, H, W, 64, 3, 2, 0, st));
  if (use_lpips == 2) return P2L_OK;   // PerceptualLoss on its own: no L1 term
  return p2l_l1_loss_bwd(img16, target, weight, loss_mask, cache->wsum, gloss, dimg16, B, H, W, 1, st);
}
