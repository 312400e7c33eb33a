// Native runtime for the StyleGAN2 generator (rosinality architecture) reached from
// pix2latent/model/stylegan2.py:116-125: mapping network, synthesis forward and the
// input-gradient backward to the latents (w / w+) and, optionally, the per-layer noise.
//
//   * modulated conv = the MFMA conv kernel with the style as a per-(sample, channel)
//     scale in the staging prologue, shared (un-modulated) weights, and the
//     demodulation factor d[b,o] as a per-(sample, channel) scale in the epilogue,
//     where noise injection, bias and leaky-ReLU*sqrt2 are fused as well;
//   * up-sampling conv = stride-2 transposed conv in sub-pixel form (4 phase 2x2 convs
//     on the low-res grid of H+1 points, P2LConv.ups=2/ext=1), followed by ONE fused
//     FIR kernel (4x4 blur + demod + noise + bias + lrelu);
//   * ToRGB = 1x1 modulated conv (3 outputs padded to 32, stored as NHWC16) with the
//     FIR-upsampled previous skip added through the conv's residual input;
//   * backward: activation backward + demod/noise reductions in one kernel, blur
//     transpose, input-gradient convs (sub-pixel ups=3 for the up convs) with the
//     modulation backward (dx = dx'*s, ds = sum dx'*x) fused into their epilogue.
// Noise layout at this boundary: layer-major, noise + Bn*noise_off[l] is [Bn][h*w].
#include "p2l_common.h"
#include "p2l_sg2_k.h"

namespace {

struct Arena {
  size_t off = 0;
  size_t take(size_t n) { const size_t o = off; off += (n + 63) & ~(size_t)63; return o; }
};
#define RET_IF(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

struct SgLayout {
  size_t s[P2L_SG2_MAX_CONVS], d[P2L_SG2_MAX_CONVS], y[P2L_SG2_MAX_CONVS];
  size_t ds[P2L_SG2_MAX_CONVS], dd[P2L_SG2_MAX_CONVS];
  size_t rs[P2L_SG2_MAX_RGBS], rds[P2L_SG2_MAX_RGBS], skip[P2L_SG2_MAX_RGBS];
  size_t x0, zeros, ubuf, upbuf, g_a, g_b, g_c, g_d, gs_a, gs_b, strips, scratch;
  // partial sums of the two deferred second-stage reductions (modulation backward: ds; demodulation: dd): one
  // buffer per layer, untouched until the flush at the end of the backward pass
  size_t part_l[P2L_SG2_MAX_CONVS], part_r[P2L_SG2_MAX_RGBS], part2_l[P2L_SG2_MAX_CONVS];
  size_t cws, cws_floats;   // conv workspace: the per-image maxima of the fp16 x 2 Winograd form
  // maxima handed from the kernel that writes a tensor to the fp16 x 2 conv that reads it (P2LAmax, round 6):
  //   amax_f [n_conv][B][64]     y[l] x s[l+1] left by the blur kernel of an up conv (atomic slots)
  //   amax_b [2][n_conv][B][64]  gd of styled_act_bwd | the blur transpose's frame
  //   amax_c                     partials of a plain conv's epilogue ([B][slots]) + [B][256] for their folded form
  size_t amax_f, amax_b, amax_c, amax_c_floats, amax_zero_floats;
  size_t total;
};
constexpr int kAmaxCompactFrom = 4096;

thread_local int g_sg_wfmt = P2L_WFMT_F32;   // the model's weight format: set by sg_layout

P2LConv mk(int B, int H, int Cin, int Cout, int taps) {
  P2LConv d{};
  d.wfmt = (taps == 9) ? (g_sg_wfmt & 0xF) : P2L_WFMT_F32;
  d.B = B; d.H = H; d.W = H; d.Cin = Cin; d.Cout = Cout; d.taps = taps;
  d.x_ld = Cin; d.alpha = 1.f; d.y_ld = Cout; d.yp_ld = Cout; d.n_store = Cout; d.splitk = 1;
  return d;
}
// The 4^2 ... 16^2 layers of a few candidates are a handful of blocks each running the whole K loop (8^2
// 512 -> 512 at 3 candidates: 81 - 93 us for 3 us of work): K slices as the BigGAN plan takes them, a function
// of the layer SHAPE only (p2l_conv_suggest_splitk), the slices summed in fixed order by the finish kernel
// that also runs the styled epilogue.  Stride-1 convs and the input-gradient form of the up convs (the forward
// sub-pixel kernel does not slice: its blockIdx.y is the output phase).
void suggest_split(P2LConv& d, size_t ws_floats_have) {
  d.splitk = 1;
  if (d.taps != 9 || !(d.ups == 0 || (d.ups == 3 && d.ext))) return;
  d.splitk = p2l_conv_suggest_splitk(&d);
  if (d.splitk > 1 && ws_floats_have && p2l_conv_workspace_bytes(&d) > ws_floats_have * sizeof(float)) d.splitk = 1;
}

int sg_layout(const P2LStyleGAN2* m, int B, SgLayout& L) {
  if (!m || B < 1 || m->n_conv < 1 || m->n_conv > P2L_SG2_MAX_CONVS || m->n_rgb > P2L_SG2_MAX_RGBS)
    return P2L_EINVAL;
  g_sg_wfmt = m->wfmt;
  Arena a;
  size_t max_act = 0, max_u = 0, max_c = 0, max_strips = 0;
  size_t max_cws = (size_t)B * 64;
  for (int l = 0; l < m->n_conv; ++l) {
    const P2LSg2Conv& c = m->conv[l];
    const size_t P = (size_t)c.res * c.res;
    L.s[l] = a.take((size_t)B * c.cin);
    L.d[l] = a.take((size_t)B * c.cout);
    L.y[l] = a.take((size_t)B * P * c.cout);
    L.ds[l] = a.take((size_t)B * c.cin);
    L.dd[l] = a.take((size_t)B * c.cout);
    const size_t cm = (size_t)(c.cin > c.cout ? c.cin : c.cout);
    if (B * P * cm > max_act) max_act = B * P * cm;
    if (c.up) {
      const size_t u = (size_t)B * (c.res + 2) * (c.res + 2) * c.cout;
      if (u > max_u) max_u = u;
    }
    if (cm > max_c) max_c = cm;
    size_t p1 = 2 * (size_t)B * cdiv(P, 128) * cm;                  // fused arb partials
    if (!c.up) {
      // K-sliced forms of the stride-1 layers: slices behind the 256 B of maxima per image, one partial of the
      // fused modulation backward per 2x2 quad
      P2LConv f = mk(B, c.res, c.cin, c.cout, 9);
      f.pro = P2L_PRO_AFFINE; f.pro_bstride = c.cin;
      suggest_split(f, 0);
      size_t w = p2l_conv_workspace_bytes(&f) / sizeof(float);
      if (w > max_cws) max_cws = w;
      P2LConv g = mk(B, c.res, c.cout, c.cin, 9);
      suggest_split(g, 0);
      w = p2l_conv_workspace_bytes(&g) / sizeof(float);
      if (w > max_cws) max_cws = w;
      const size_t pq = 2 * (size_t)B * p2l_conv_arb_nblk_ws(&g) * cm;
      if (pq > p1) p1 = pq;
    } else {
      P2LConv g = mk(B, c.res, c.cout, c.cin, 9);
      g.ups = 3; g.ext = 1;
      suggest_split(g, 0);
      const size_t w = p2l_conv_workspace_bytes(&g) / sizeof(float);
      if (w > max_cws) max_cws = w;
      const size_t pq = 2 * (size_t)B * p2l_conv_arb_nblk_ws(&g) * cm;
      if (pq > p1) p1 = pq;
    }
    L.part_l[l] = a.take(p1);
    L.part2_l[l] = a.take((size_t)B * p2l_sg2_act_bwd_nblk((int)P) * c.cout);
    const size_t st = (size_t)(c.cout / ((c.cout % 64) ? 32 : 64)) * B * P;
    if (st > max_strips) max_strips = st;
  }
  for (int j = 0; j < m->n_rgb; ++j) {
    const P2LSg2Rgb& r = m->rgb[j];
    L.rs[j] = a.take((size_t)B * r.cin);
    L.rds[j] = a.take((size_t)B * r.cin);
    L.skip[j] = a.take((size_t)B * r.res * r.res * 16);
    L.part_r[j] = a.take(2 * (size_t)B * cdiv((size_t)r.res * r.res, 128) * r.cin);
  }
  L.x0 = a.take((size_t)B * 16 * m->conv[0].cin);
  L.zeros = a.take((size_t)B * max_c);
  L.ubuf = a.take(max_u ? max_u : 64);
  const size_t img = (size_t)B * m->size * m->size * 16;
  L.upbuf = a.take(img);
  L.g_a = a.take(max_act);
  L.g_b = a.take(max_act);
  L.g_c = a.take(max_act);
  L.g_d = a.take(max_u > max_act ? max_u : max_act);
  L.gs_a = a.take(img);
  L.gs_b = a.take(img);
  L.strips = a.take(max_strips);
  L.scratch = a.take((size_t)B * max_c * 2);
  L.cws_floats = max_cws;
  L.cws = a.take(L.cws_floats);
  L.amax_zero_floats = (size_t)3 * m->n_conv * B * P2L_SG2_AMAX_SLOTS;
  L.amax_f = a.take(L.amax_zero_floats);
  L.amax_b = L.amax_f + (size_t)m->n_conv * B * P2L_SG2_AMAX_SLOTS;
  size_t mc = 0;
  for (int l = 0; l + 1 < m->n_conv; ++l) {
    // (an upper bound of p2l_conv_amax_slots for every kernel form: one partial per wave of a 128-pixel x
    //  32-channel tile -- the format of the run is not known when the workspace is sized)
    const P2LSg2Conv& c = m->conv[l];
    const size_t n = (size_t)c.res * c.res * c.cout / 1024;
    if (!c.up && n > mc) mc = n;
  }
  L.amax_c_floats = (size_t)B * (mc + 256);
  L.amax_c = a.take(L.amax_c_floats);
  L.total = a.off;
  return P2L_OK;
}


// input-gradient conv + modulation backward (dx = dx' * s + extra ; ds = sum_p dx' * x)
// amax_in: [B][P2L_SG2_AMAX_SLOTS] maxima of gin left by the kernel that wrote it, or NULL
int dgrad_scale(P2LConv& d, const float* gin, const float* w, const float* x, const float* s,
                int C, const float* extra, float* dx, float* ds, float* tmp, float* part,
                float* scratch, int B, int Hout, float* cws, size_t cws_floats, const float* amax_in, void* st) {
  suggest_split(d, cws_floats);
  const bool split_fused = d.splitk > 1 && p2l_conv_arb_split_fusable(&d);
  if (split_fused || (d.splitk == 1 && p2l_conv_arb_fusable(&d))) {
    P2LArb a{};
    a.amax.in = amax_in; a.amax.in_n = amax_in ? P2L_SG2_AMAX_SLOTS : 0;
    a.x = x; a.x_ld = C; a.s = s; a.t = s; a.st_bstride = C;
    a.skip = extra; a.skip_ld = C; a.skip_C = extra ? C : 0; a.skip_ups = 0;
    a.ds = ds; a.dt = scratch; a.dsdt_bstride = C; a.partial = part; a.nomask = 1;
    return p2l_conv_dgrad_arb_ws(&d, &a, gin, w, dx, cws, cws_floats * sizeof(float), st);
  }
  P2LConvExtra ex{};
  ex.amax.in = amax_in; ex.amax.in_n = amax_in ? P2L_SG2_AMAX_SLOTS : 0;
  RET_IF(p2l_conv_fwd_ex(&d, &ex, gin, w, nullptr, nullptr, nullptr, nullptr, nullptr, tmp, nullptr,
                         cws, cws_floats * sizeof(float), st));
  return p2l_scale_bwd(tmp, C, x, C, s, C, extra, C, extra ? C : 0, dx, C, ds, scratch, C, part, B,
                       Hout, Hout, C, st);
}

}  // namespace

extern "C" size_t p2l_sg2_ws_bytes(const P2LStyleGAN2* m, int Bn) {
  SgLayout L;
  if (sg_layout(m, Bn, L)) return 0;
  return L.total * sizeof(float);
}

// test hook (include/p2l_test.h): where the post-activation output of styled conv l lives in ws after
// p2l_sg2_synthesis_fwd, NHWC [B, res, res, cout] -- its signs are the leaky-ReLU decisions of the run
// (oracle/replay.py replays them in the CPU oracle)
extern "C" int p2l_sg2_ws_lookup(const P2LStyleGAN2* m, int Bn, int l, size_t* float_off, int32_t shape[4]) {
  SgLayout L;
  if (!m || !float_off || !shape || sg_layout(m, Bn, L) || l < 0 || l >= m->n_conv) return P2L_EINVAL;
  *float_off = L.y[l];
  shape[0] = Bn; shape[1] = m->conv[l].res; shape[2] = m->conv[l].res; shape[3] = m->conv[l].cout;
  return P2L_OK;
}

// acts: [9][B][D]: slot 0 = PixelNorm(z), slot i+1 = output of mapping layer i
extern "C" int p2l_sg2_mapping_fwd(const P2LStyleGAN2* m, const float* z, float* w, float* acts,
                                   int B, void* st) {
  const int D = m->style_dim;
  RET_IF(p2l_sg2_pixelnorm_fwd(z, acts, B, D, st));
  for (int i = 0; i < 8; ++i) {
    float* out = acts + (size_t)(i + 1) * B * D;
    RET_IF(p2l_linear_fwd(acts + (size_t)i * B * D, m->map_w[i], nullptr, out, B, D, D, st));
    RET_IF(p2l_sg2_bias_lrelu_fwd(out, m->map_b[i], 1.f, B, D, st));
  }
  if (hipMemcpyAsync(w, acts + (size_t)8 * B * D, (size_t)B * D * sizeof(float),
                     hipMemcpyDeviceToDevice, (hipStream_t)st) != hipSuccess)
    return P2L_ELAUNCH;
  return P2L_OK;
}

extern "C" int p2l_sg2_mapping_bwd(const P2LStyleGAN2* m, const float* z, const float* acts,
                                   const float* dw, float* dz, float* scratch, int B, void* st) {
  const int D = m->style_dim;
  float* g = scratch;
  float* g2 = scratch + (size_t)B * D;
  if (hipMemcpyAsync(g, dw, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice,
                     (hipStream_t)st) != hipSuccess)
    return P2L_ELAUNCH;
  for (int i = 7; i >= 0; --i) {
    RET_IF(p2l_sg2_lrelu_bwd(acts + (size_t)(i + 1) * B * D, g, B * D, st));
    RET_IF(p2l_linear_bwd(g, m->map_w[i], g2, B, D, D, 0, st));
    float* t = g; g = g2; g2 = t;
  }
  return p2l_sg2_pixelnorm_bwd(z, g, dz, B, D, st);
}

extern "C" int p2l_sg2_synthesis_fwd(const P2LStyleGAN2* m, const float* latent,
                                     const float* noise, int B, void* ws, size_t ws_bytes,
                                     float* img16, void* st) {
  SgLayout L;
  RET_IF(sg_layout(m, B, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !latent || !noise || !img16) return P2L_EWS;
  float* W = (float*)ws;
  const int D = m->style_dim, lat_ld = m->n_latent * D;
  if (hipMemsetAsync(W + L.zeros, 0, (L.ubuf - L.zeros) * sizeof(float), (hipStream_t)st) != hipSuccess)
    return P2L_ELAUNCH;
  RET_IF(p2l_broadcast_rows(m->const_input, W + L.x0, (int64_t)16 * m->conv[0].cin, B, st));
  const float* x = W + L.x0;
  {
    // every layer's style s = latent . mod_w + mod_b and demodulation scale
    // d = rsqrt(s^2 . wsq + eps) depend only on the latents: two grouped launches
    p2lsg2::GLinFwdK gs{}, gd{};
    gs.Bn = gd.Bn = B; gs.mode = 0; gd.mode = 1;
    for (int l = 0; l < m->n_conv; ++l) {
      const P2LSg2Conv& c = m->conv[l];
      p2lsg2::GLinItem& a = gs.g[gs.n++];
      a.W = c.mod_w; a.bias = c.mod_b; a.x = latent + (size_t)c.latent_idx * D; a.x_ld = lat_ld;
      a.y = W + L.s[l]; a.y_ld = c.cin; a.K = D; a.N = c.cin;
      p2lsg2::GLinItem& e = gd.g[gd.n++];
      e.W = c.wsq; e.bias = nullptr; e.x = W + L.s[l]; e.x_ld = c.cin;
      e.y = W + L.d[l]; e.y_ld = c.cout; e.K = c.cin; e.N = c.cout;
    }
    for (int j = 0; j < m->n_rgb; ++j) {
      const P2LSg2Rgb& r = m->rgb[j];
      p2lsg2::GLinItem& a = gs.g[gs.n++];
      a.W = r.mod_w; a.bias = r.mod_b; a.x = latent + (size_t)r.latent_idx * D; a.x_ld = lat_ld;
      a.y = W + L.rs[j]; a.y_ld = r.cin; a.K = D; a.N = r.cin;
    }
    RET_IF(p2lsg2::grouped_linear_fwd(gs, st));
    RET_IF(p2lsg2::grouped_linear_fwd(gd, st));
  }
  // Maxima of every conv's MODULATED input (y[l-1] * s[l]), left by the kernel that wrote y[l-1]: the fp16 x 2
  // launches then need no pass of their own over it (35 such passes were 6.6 % of the FFHQ-1024 step,
  // profiles/round6_sg2_1024_kernel_stats.csv).  P2L_WFMT_FLAG_NO_AMAX: every launch reduces its own.
  const bool hand = !(m->wfmt & P2L_WFMT_FLAG_NO_AMAX);
  if (hand && hipMemsetAsync(W + L.amax_f, 0, (size_t)m->n_conv * B * P2L_SG2_AMAX_SLOTS * sizeof(float),
                             (hipStream_t)st) != hipSuccess)
    return P2L_ELAUNCH;
  const float* am_in = nullptr;          // maxima of x with s[l] applied, [B][am_n]
  int am_n = 0;
  int rj = 0;
  for (int l = 0; l < m->n_conv; ++l) {
    const P2LSg2Conv& c = m->conv[l];
    const float* nz = noise + (size_t)B * c.noise_off;
    P2LConv d = mk(B, c.res, c.cin, c.cout, 9);
    d.pro = P2L_PRO_AFFINE; d.pro_bstride = c.cin;
    P2LConvExtra ex{};
    ex.amax.in = am_in; ex.amax.in_n = am_n; ex.amax.in_applied = am_in ? 1 : 0;
    am_in = nullptr; am_n = 0;
    const bool to_next = hand && l + 1 < m->n_conv;
    const float* next_s = to_next ? W + L.s[l + 1] : nullptr;
    if (!c.up) {
      d.act = P2L_ACT_LRELU_SQRT2;
      suggest_split(d, L.cws_floats);
      ex.oscale = W + L.d[l]; ex.oscale_bstride = c.cout; ex.noise = nz; ex.noise_w = c.noise_w;
      int ns = to_next ? p2l_conv_amax_slots(&d) : 0;
      if (ns > 0 && (size_t)B * (ns + 256) <= L.amax_c_floats) {
        ex.amax.out = W + L.amax_c;
        ex.amax.next_s = next_s; ex.amax.next_t = W + L.zeros; ex.amax.next_bstride = m->conv[l + 1].cin;
      } else {
        ns = 0;
      }
      RET_IF(p2l_conv_fwd_ex(&d, &ex, x, c.w, c.act_b, W + L.s[l], W + L.zeros, nullptr, nullptr,
                             W + L.y[l], nullptr, W + L.cws, L.cws_floats * sizeof(float), st));
      if (ns > 0) {
        am_in = W + L.amax_c; am_n = ns;
        if (ns >= kAmaxCompactFrom) {     // (every block of the reader reduces ALL partials of its image)
          float* folded = W + L.amax_c + (size_t)B * ns;
          RET_IF(p2l_amax_compact(am_in, B, ns, folded, st));
          am_in = folded; am_n = 256;
        }
      }
    } else {
      d.ups = 2; d.ext = 1;
      // (the workspace holds the per-image maxima of the fp16 x 2 form: without it the launch is bf16 x 3)
      RET_IF(p2l_conv_fwd_ex(&d, &ex, x, c.w, nullptr, W + L.s[l], W + L.zeros, nullptr, nullptr,
                             W + L.ubuf, nullptr, W + L.cws, L.cws_floats * sizeof(float), st));
      float* slots = to_next ? W + L.amax_f + (size_t)l * B * P2L_SG2_AMAX_SLOTS : nullptr;
      RET_IF(p2l_sg2_blur_fwd_amax(W + L.ubuf, W + L.d[l], nz, c.noise_w, c.act_b, W + L.y[l], B, c.res,
                                   c.res, c.cout, next_s, slots, st));
      if (slots) { am_in = slots; am_n = P2L_SG2_AMAX_SLOTS; }
    }
    x = W + L.y[l];
    while (rj < m->n_rgb && m->rgb[rj].after_conv == l) {
      const P2LSg2Rgb& r = m->rgb[rj];
      const float* res = nullptr;
      if (rj > 0) {
        RET_IF(p2l_sg2_rgb_up_fwd(W + L.skip[rj - 1], W + L.upbuf, B, r.res / 2, r.res / 2, st));
        res = W + L.upbuf;
      }
      P2LConv t = mk(B, r.res, r.cin, 32, 1);
      t.pro = P2L_PRO_AFFINE; t.pro_bstride = r.cin; t.n_store = 16; t.y_ld = 16; t.res_ld = 16;
      t.algo_flops = 2.0 * B * r.res * r.res * (double)r.cin * 3;
      RET_IF(p2l_conv_fwd(&t, x, r.w, r.bias, W + L.rs[rj], W + L.zeros, res, nullptr,
                          W + L.skip[rj], nullptr, nullptr, 0, st));
      ++rj;
    }
  }
  return p2l_sg2_clamp16_fwd(W + L.skip[m->n_rgb - 1], img16, (int64_t)B * m->size * m->size, st);
}

extern "C" int p2l_sg2_synthesis_bwd(const P2LStyleGAN2* m, const float* latent,
                                     const float* noise, int B, void* ws, size_t ws_bytes,
                                     const float* dimg16, float* dlatent, float* dnoise,
                                     void* st) {
  SgLayout L;
  RET_IF(sg_layout(m, B, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !dimg16 || !dlatent) return P2L_EWS;
  float* W = (float*)ws;
  const int D = m->style_dim, lat_ld = m->n_latent * D;
  (void)latent;
  if (hipMemsetAsync(dlatent, 0, (size_t)B * lat_ld * sizeof(float), (hipStream_t)st) != hipSuccess)
    return P2L_ELAUNCH;
  const bool hand = !(m->wfmt & P2L_WFMT_FLAG_NO_AMAX);
  if (hand && hipMemsetAsync(W + L.amax_b, 0, (size_t)2 * m->n_conv * B * P2L_SG2_AMAX_SLOTS * sizeof(float),
                             (hipStream_t)st) != hipSuccess)
    return P2L_ELAUNCH;
  float* gs_cur = W + L.gs_a;
  float* gs_prev = W + L.gs_b;
  RET_IF(p2l_sg2_clamp16_bwd(W + L.skip[m->n_rgb - 1], dimg16, gs_cur,
                             (int64_t)B * m->size * m->size, st));
  float* gy = W + L.g_a;     // dL/dy of the layer being processed
  float* gd = W + L.g_b;     // activation-backward result
  float* gx = W + L.g_c;     // gradient flowing to the previous layer
  float* tmp = W + L.g_d;    // blur transpose / unfused temp
  float* scratch = W + L.scratch;
  // the second stage of every per-(sample, channel) reduction of the pass (26 + 17 launches of a few blocks) is
  // recorded and runs as two launches in front of the style-gradient linears
  struct Defer {
    Defer() { p2l_arb_defer_begin(); p2l_sg2_rows_defer_begin(); }
    ~Defer() { p2l_arb_defer_cancel(); p2l_sg2_rows_defer_cancel(); }     // (no-ops after the flushes)
  } defer;
  bool have_next = false;    // gx holds a gradient for the current layer's output
  int rj = m->n_rgb - 1;
  for (int l = m->n_conv - 1; l >= 0; --l) {
    const P2LSg2Conv& c = m->conv[l];
    const float* x_in = (l == 0) ? W + L.x0 : W + L.y[l - 1];
    const int res_in = c.up ? c.res / 2 : c.res;
    // ---- ToRGB reading y_l -------------------------------------------------
    bool gy_ready = false;
    while (rj >= 0 && m->rgb[rj].after_conv == l) {
      const P2LSg2Rgb& r = m->rgb[rj];
      P2LConv t = mk(B, r.res, 16, r.cin, 1);
      t.algo_flops = 2.0 * B * r.res * r.res * (double)r.cin * 3;
      RET_IF(dgrad_scale(t, gs_cur, r.wt, W + L.y[l], W + L.rs[rj], r.cin,
                         have_next ? gx : nullptr, gy, W + L.rds[rj], tmp, W + L.part_r[rj], scratch, B, r.res,
                         W + L.cws, L.cws_floats, nullptr, st));
      if (rj > 0) {
        RET_IF(p2l_sg2_rgb_up_bwd(gs_cur, gs_prev, B, r.res / 2, r.res / 2, 0, st));
        float* t2 = gs_cur; gs_cur = gs_prev; gs_prev = t2;
      }
      gy_ready = true;
      --rj;
    }
    const float* dy = gy_ready ? gy : gx;      // only the next conv consumed y_l
    if (!gy_ready && !have_next) return P2L_EINVAL;
    // ---- styled conv l -----------------------------------------------------
    const float* nz = noise + (size_t)B * c.noise_off;
    float* dnz = dnoise ? dnoise + (size_t)B * c.noise_off : nullptr;
    // the maxima of the input-gradient conv's input come from the kernel that writes it: the activation
    // backward (plain convs) or the blur transpose (up convs)
    float* am_gd = hand ? W + L.amax_b + (size_t)(2 * l) * B * P2L_SG2_AMAX_SLOTS : nullptr;
    float* am_du = hand ? W + L.amax_b + (size_t)(2 * l + 1) * B * P2L_SG2_AMAX_SLOTS : nullptr;
    RET_IF(p2l_sg2_styled_act_bwd_amax(dy, W + L.y[l], W + L.d[l], nz, c.noise_w, c.act_b, gd, W + L.dd[l],
                                       dnz, W + L.part2_l[l], W + L.strips, B, c.res * c.res, c.cout,
                                       c.up ? nullptr : am_gd, st));
    // gx may alias dy (when !gy_ready): the dgrad below writes gx only after gd was produced
    float* gout = (dy == gx) ? gy : gx;
    if (c.up) {
      RET_IF(p2l_sg2_blur_bwd_amax(gd, tmp, B, c.res, c.res, c.cout, am_du, st));
      P2LConv d = mk(B, c.res, c.cout, c.cin, 9);
      d.ups = 3; d.ext = 1;
      // unfused temp must not alias the conv input (tmp): use gd
      RET_IF(dgrad_scale(d, tmp, c.wt, x_in, W + L.s[l], c.cin, nullptr, gout, W + L.ds[l], gd, W + L.part_l[l],
                         scratch, B, res_in, W + L.cws, L.cws_floats, am_du, st));
    } else {
      P2LConv d = mk(B, c.res, c.cout, c.cin, 9);
      RET_IF(dgrad_scale(d, gd, c.wt, x_in, W + L.s[l], c.cin, nullptr, gout, W + L.ds[l], tmp, W + L.part_l[l],
                         scratch, B, res_in, W + L.cws, L.cws_floats, am_gd, st));
    }
    if (gout != gx) { float* t2 = gy; gy = gx; gx = t2; }   // keep "gx = gradient for layer l-1"
    have_next = true;
  }
  // style gradients of all layers at once: ds += demodulation backward; then
  // dlatent[latent_idx] += ds . mod_w^T (convs, then ToRGBs: each launch touches every
  // latent row at most once, and the two launches are ordered -> deterministic)
  p2lsg2::GLinBwdK gm{}, gc{}, gr{};
  gm.Bn = gc.Bn = gr.Bn = B; gm.mode = 1;
  for (int l = 0; l < m->n_conv; ++l) {
    const P2LSg2Conv& c = m->conv[l];
    p2lsg2::GLinBwdItem& a = gm.g[gm.n++];
    a.W = c.wsq; a.dy = W + L.dd[l]; a.d = W + L.d[l]; a.x = W + L.s[l]; a.dx = W + L.ds[l];
    a.K = c.cin; a.N = c.cout; a.dx_ld = c.cin; a.accumulate = 1;
    p2lsg2::GLinBwdItem& e = gc.g[gc.n++];
    e.W = c.mod_w; e.dy = W + L.ds[l]; e.dx = dlatent + (size_t)c.latent_idx * D;
    e.K = D; e.N = c.cin; e.dx_ld = lat_ld; e.accumulate = 1;
  }
  for (int j = 0; j < m->n_rgb; ++j) {
    const P2LSg2Rgb& r = m->rgb[j];
    p2lsg2::GLinBwdItem& e = gr.g[gr.n++];
    e.W = r.mod_w; e.dy = W + L.rds[j]; e.dx = dlatent + (size_t)r.latent_idx * D;
    e.K = D; e.N = r.cin; e.dx_ld = lat_ld; e.accumulate = 1;
  }
  RET_IF(p2l_arb_defer_flush(st));
  RET_IF(p2l_sg2_rows_defer_flush(st));
  RET_IF(p2lsg2::grouped_linear_bwd(gm, st));
  RET_IF(p2lsg2::grouped_linear_bwd(gc, st));
  RET_IF(p2lsg2::grouped_linear_bwd(gr, st));
  return P2L_OK;
}
