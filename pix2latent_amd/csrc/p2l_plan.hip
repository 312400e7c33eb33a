// Native runtime of the hot path: sequences the HIP kernels of libp2l_hip into
//   * BigGAN-deep generator forward / input-gradient backward
//     (replaces pix2latent/model/biggan.py:58 -> HF Generator.forward and the
//      autograd walk started by pix2latent/optimizer/closure.py:58), and
//   * ProjectionLoss = weighted L1 + beta * weighted LPIPS-VGG16 forward/backward
//     (replaces pix2latent/loss_functions.py:97-100,117-124,140-148).
// Only input gradients are computed (the reference also computes and discards
// weight gradients); target features are cached (F8).
// No allocation, no synchronisation: one workspace arena owned by the caller,
// every launch on the caller's stream, so a whole step can be graph-captured.
#include "p2l_common.h"

namespace {

struct Arena {
  size_t off = 0;  // in floats
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

// ---------------------------------------------------------------------------
// small kernels local to the plans
// ---------------------------------------------------------------------------
__global__ void concat2_kernel(const float* z, const float* c, float* cond, int Bn,
                               int nz, int nc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = nz + nc;
  if (i >= Bn * n) return;
  const int b = i / n, j = i - b * n;
  cond[i] = (j < nz) ? z[b * nz + j] : c[b * nc + (j - nz)];
}
__global__ void split2_kernel(const float* dcond, float* dz, float* dc, int Bn, int nz,
                              int nc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = nz + nc;
  if (i >= Bn * n) return;
  const int b = i / n, j = i - b * n;
  if (j < nz) dz[b * nz + j] = dcond[i];
  else dc[b * nc + (j - nz)] = dcond[i];
}
__global__ void vec_scale_div_kernel(const float* a, const float* div, float* out, int n,
                                     float scale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = div ? a[i] * scale / div[i] : a[i] * scale;
}

// ---------------------------------------------------------------------------
// conv helper
// ---------------------------------------------------------------------------
struct ConvCall {
  P2LConv d{};
  const float* x = nullptr; const float* w = nullptr; const float* bias = nullptr;
  const float* ps = nullptr; const float* pt = nullptr; const float* res = nullptr;
  const float* mask = nullptr; float* y = nullptr; float* yp = nullptr;
  bool want_amax = false;   // the next reader of y / yp is a 3x3 conv: leave its maxima (P2LAmax)
  // ... and the affine that reader fuses in front of its ReLU, when the plan knows it: the maxima of y
  // are then recorded as max|y*s + t| (P2LAmax.next_s) and the reader needs no bound
  const float* next_ps = nullptr; const float* next_pt = nullptr; int next_bstride = 0;
};
// weight format of the 3x3 convs of the model whose plan is running on this thread
// (set at every extern "C" entry from the model struct's wfmt)
thread_local int g_plan_wfmt = P2L_WFMT_F32;
// P2LConv.form bits every conv of the running plan gets (A/B builds: -DP2L_AB_BWD_BF3 / _BG / _PL keep the
// backward passes -- both / generator / loss network -- in the bf16 x 3 arithmetic)
thread_local int g_plan_form = 0;

ConvCall mk_conv(int B, int H, int W, int Cin, int Cout, int taps) {
  ConvCall c;
  c.d.wfmt = (taps == 9) ? (g_plan_wfmt & 0xF)
                         : ((g_plan_wfmt & P2L_WFMT_FLAG_PW) ? P2L_WFMT_PW : P2L_WFMT_F32);
  c.d.ext = 0;
  c.d.form = g_plan_form;
  c.d.B = B; c.d.H = H; c.d.W = W; c.d.Cin = Cin; c.d.Cout = Cout; c.d.taps = taps;
  c.d.ups = 0; c.d.x_ld = Cin; c.d.pro = P2L_PRO_NONE; c.d.pro_bstride = 0;
  c.d.alpha = 1.f; c.d.act = P2L_ACT_NONE; c.d.pool = P2L_POOL_NONE;
  c.d.y_ld = Cout; c.d.yp_ld = Cout; c.d.n_store = Cout; c.d.res_ld = 0;
  c.d.res_ups = 0; c.d.mask_ld = 0; c.d.splitk = 1; c.d.algo_flops = 0.0;
  return c;
}
size_t conv_ws_floats(ConvCall& c) {
  c.d.splitk = p2l_conv_suggest_splitk(&c.d);
  return p2l_conv_workspace_bytes(&c.d) / sizeof(float);
}

// ---------------------------------------------------------------------------
// Maxima handed from the conv that writes a tensor to the conv that reads it (P2LAmax): the
// bookkeeping of ONE plan run, on the stack of the entry point (thread_local pointer: a plan runs
// on the calling thread).  A tensor is known by its address and extents; its entry dies when
// another conv writes the buffer (run_conv), when a non-conv kernel writes it (amax_drop at that
// call site) or when its slot set comes round again in the ring.
// ---------------------------------------------------------------------------
// Round 6: many partials are compacted once.  Every block of a reader reduces ALL partial maxima of its image in
// its prologue; a producer at 512^2 / 1024^2 (VGG behind StyleGAN2) leaves 8 192 ... 32 768 of them per image, and
// thousands of reader blocks each pulled 32 - 128 KB through L2 before their first multiply (512^2 128 -> 128 at
// 3 candidates ran at 243 TFLOP/s where the same kernel does 340 on 128^2 x 18).  The first reader of such an
// entry now runs one tiny kernel that folds the partials to 256 per image (max is exact: same scales, same
// bits), and every reader gets those.
// grid (B, 8): block (b, g) folds the g-th eighth of image b's partials into out[b][32 g .. 32 g + 31] (one block
// per image took 14 us for 32 768 partials, 21 times per FFHQ-1024 step)
__global__ void amax_compact_kernel(const float* __restrict__ in, int n, float* __restrict__ out) {
  __shared__ float red[256];
  const int seg = (n + 7) >> 3, lo = blockIdx.y * seg, hi = min(n, lo + seg);
  const float* p = in + (size_t)blockIdx.x * n;
  float m = 0.f;                                       // (partials are maxima of |.|: >= 0)
  for (int i = lo + threadIdx.x; i < hi; i += 256) m = fmaxf(m, p[i]);
  red[threadIdx.x] = m;
  __syncthreads();
  if (threadIdx.x < 32) {
#pragma unroll
    for (int j = 1; j < 8; ++j) m = fmaxf(m, red[threadIdx.x + 32 * j]);
    out[(size_t)blockIdx.x * 256 + blockIdx.y * 32 + threadIdx.x] = m;
  }
}
struct AmaxReg {
  static constexpr int NSETS = 6, NENT = 8, COMPACT_FROM = 4096, COMPACT_TO = 256;
  // a set = [B][set_floats / B] partials of one tensor + [B][COMPACT_TO] floats for their compacted form
  float* ring = nullptr; size_t set_floats = 0, set_stride = 0; int next = 0;
  struct Ent { const float* t = nullptr; int B = 0, H = 0, W = 0, C = 0; const float* slots = nullptr; int n = 0, set = -1;
               const float* ps = nullptr; int pbs = 0;      // ps: recorded as max|t*ps + pt| (the reader's prologue)
               bool compacted = false; };
  Ent e[NENT];
  float* take(int* set) {
    *set = next % NSETS; ++next;
    for (Ent& x : e) if (x.set == *set) x = Ent();
    return ring + (size_t)*set * set_stride;
  }
  void drop(const float* t) { if (t) for (Ent& x : e) if (x.t == t) x = Ent(); }
  void put(const float* t, int B, int H, int W, int C, const float* slots, int n, int set,
           const float* ps = nullptr, int pbs = 0) {
    drop(t);
    Ent* f = &e[0];
    for (Ent& x : e) if (!x.t) { f = &x; break; }
    f->t = t; f->B = B; f->H = H; f->W = W; f->C = C; f->slots = slots; f->n = n; f->set = set;
    f->ps = ps; f->pbs = pbs; f->compacted = false;
  }
  // ps / pbs = the prologue vector the READER fuses (NULL: none).  Maxima recorded with an affine
  // applied serve that reader only (*applied = 1); raw maxima serve every reader (bound in the kernel).
  // st: the stream the reader will be launched on (the compaction kernel goes in front of it); NULL = host
  // logic only (self-test): nothing is launched
  bool get(const float* t, int B, int H, int W, int C, const float** slots, int32_t* n,
           const float* ps = nullptr, int pbs = 0, int32_t* applied = nullptr, void* st = nullptr,
           bool launch = false) {
    for (Ent& x : e)
      if (x.t == t && t && x.B == B && x.H == H && x.W == W && x.C == C) {
        if (x.ps && (x.ps != ps || x.pbs != pbs || !applied)) return false;
        if (launch && !x.compacted && x.n >= COMPACT_FROM && set_stride >= set_floats + (size_t)B * COMPACT_TO) {
          float* dst = ring + (size_t)x.set * set_stride + set_floats;
          hipLaunchKernelGGL(amax_compact_kernel, dim3(B, 8), dim3(256), 0, (hipStream_t)st, x.slots, x.n, dst);
          x.slots = dst; x.n = COMPACT_TO; x.compacted = true;
        }
        *slots = x.slots; *n = x.n;
        if (applied) *applied = x.ps ? 1 : 0;
        return true;
      }
    return false;
  }
};
thread_local AmaxReg* g_amax = nullptr;
struct AmaxScope {                                     // one per plan entry point
  AmaxReg reg;
  AmaxScope(float* ring, size_t set_floats, size_t set_stride) {
    reg.ring = ring; reg.set_floats = set_floats; reg.set_stride = set_stride;
#ifdef P2L_NO_AMAX_HANDOVER                            // (A/B build: tools/ab_build.sh p2l_plan -DP2L_NO_AMAX_HANDOVER)
    set_floats = 0;
#endif
    if (g_plan_wfmt & P2L_WFMT_FLAG_NO_AMAX) set_floats = 0;   // (model descriptor flag)
    g_amax = set_floats ? &reg : nullptr;
  }
  ~AmaxScope() { g_amax = nullptr; }
};
inline void amax_drop(const float* t) { if (g_amax) g_amax->drop(t); }
// slots per image a plan has to provide for this conv's maxima (ring sizing)
int conv_amax_slots(ConvCall& c) {
  c.d.splitk = p2l_conv_suggest_splitk(&c.d);
  return p2l_conv_amax_slots(&c.d);
}

}  // namespace
// (library-internal, p2l_common.h: the StyleGAN2 plan folds the partials of its high-resolution producers too)
int p2l_amax_compact(const float* in, int B, int n, float* out, void* st) {
  hipLaunchKernelGGL(amax_compact_kernel, dim3(B, 8), dim3(256), 0, (hipStream_t)st, in, n, out);
  return p2l_check_launch();
}
// test hook (host logic only, no GPU): the bookkeeping rules of AmaxReg; 0 = all hold, else the
// number of the first rule that failed (tests/test_abi.py)
extern "C" int p2l_selftest_amaxreg(void) {
  static float ring[AmaxReg::NSETS * 8];
  AmaxReg R;
  R.ring = ring; R.set_floats = 8; R.set_stride = 8;
  const float *slots = nullptr; int32_t n = 0;
  float t1[1], t2[1], t3[1];
  int set = -1;
  float* s1 = R.take(&set);
  R.put(t1, 2, 32, 32, 64, s1, 4, set);
  if (!R.get(t1, 2, 32, 32, 64, &slots, &n) || slots != s1 || n != 4) return 1;   // found, same slots
  if (R.get(t1, 2, 32, 32, 128, &slots, &n)) return 2;                            // other extents: a different tensor
  if (R.get(t2, 2, 32, 32, 64, &slots, &n)) return 3;                             // other address
  R.drop(t1);
  if (R.get(t1, 2, 32, 32, 64, &slots, &n)) return 4;                             // dropped (someone else wrote it)
  float* s2 = R.take(&set);
  R.put(t2, 1, 16, 16, 64, s2, 2, set);
  float* s3 = R.take(&set);
  R.put(t2, 1, 16, 16, 64, s3, 2, set);                                           // re-written: one entry, new slots
  if (!R.get(t2, 1, 16, 16, 64, &slots, &n) || slots != s3) return 5;
  int live = 0;
  for (const auto& e : R.e) live += e.t == t2;
  if (live != 1) return 6;
  for (int i = 0; i < AmaxReg::NSETS; ++i) (void)R.take(&set);                    // the ring comes round
  if (R.get(t2, 1, 16, 16, 64, &slots, &n)) return 7;                             // its set was handed out again
  for (int i = 0; i < AmaxReg::NENT + 3; ++i) {                                   // more tensors than entries
    float* s = R.take(&set);
    R.put(t3 + 0, i + 1, 16, 16, 64, s, 1, set);
  }
  if (!R.get(t3, AmaxReg::NENT + 3, 16, 16, 64, &slots, &n)) return 8;            // the latest is there
  live = 0;
  for (const auto& e : R.e) live += e.t != nullptr;
  if (live > AmaxReg::NSETS) return 9;                                            // never more live entries than sets
  // maxima recorded with the reader's affine applied (P2LAmax.next_s) serve that reader only
  float s1v[1], s2v[1];
  int32_t applied = -1;
  float* s4 = R.take(&set);
  R.put(t1, 2, 8, 8, 64, s4, 3, set, s1v, 64);
  if (!R.get(t1, 2, 8, 8, 64, &slots, &n, s1v, 64, &applied) || applied != 1 || slots != s4) return 10;
  if (R.get(t1, 2, 8, 8, 64, &slots, &n, s2v, 64, &applied)) return 11;            // another prologue
  if (R.get(t1, 2, 8, 8, 64, &slots, &n, s1v, 0, &applied)) return 12;             // another image stride
  if (R.get(t1, 2, 8, 8, 64, &slots, &n, nullptr, 0, &applied)) return 13;         // a reader without prologue
  if (R.get(t1, 2, 8, 8, 64, &slots, &n)) return 14;                               // a reader that cannot say
  float* s5 = R.take(&set);
  R.put(t1, 2, 8, 8, 64, s5, 3, set);                                              // raw maxima: every reader
  applied = -1;
  if (!R.get(t1, 2, 8, 8, 64, &slots, &n, s2v, 64, &applied) || applied != 0) return 15;
  return 0;
}
namespace {

int run_conv(ConvCall& c, float* skws, size_t skws_floats, void* st) {
  c.d.splitk = p2l_conv_suggest_splitk(&c.d);
  AmaxReg* R = g_amax;
  P2LConvExtra ex{};
  float *so = nullptr, *sop = nullptr;
  int ns = 0, set_o = -1, set_p = -1;
  if (R) {
    // (the input of a sub-pixel forward launch is the LOW-resolution tensor, that of its
    //  input-gradient form the high-resolution one)
    const float* rps = c.d.pro != P2L_PRO_NONE ? c.ps : nullptr;
    if ((c.d.ups == 0 || c.d.ups == 3) && c.d.x_ld == c.d.Cin)
      R->get(c.x, c.d.B, c.d.H, c.d.W, c.d.Cin, &ex.amax.in, &ex.amax.in_n, rps, c.d.pro_bstride, &ex.amax.in_applied, st, true);
    else if (c.d.ups == 2 && !c.d.ext && c.d.x_ld == c.d.Cin)
      R->get(c.x, c.d.B, c.d.H / 2, c.d.W / 2, c.d.Cin, &ex.amax.in, &ex.amax.in_n, rps, c.d.pro_bstride, &ex.amax.in_applied, st, true);
    R->drop(c.y); R->drop(c.yp);                       // this launch overwrites them
    ns = c.want_amax ? p2l_conv_amax_slots(&c.d) : 0;
    if (ns > 0 && (size_t)ns * c.d.B <= R->set_floats && c.d.n_store == c.d.Cout) {
      if (c.y && c.d.y_ld == c.d.Cout) so = R->take(&set_o);
      if (c.yp && c.d.yp_ld == c.d.Cout) sop = R->take(&set_p);
      ex.amax.out = so; ex.amax.outp = sop;
#ifndef P2L_AB_NO_NEXT_AFFINE                          // (A/B build: raw maxima + the reader's bound, as until round 4)
      if (so && c.next_ps && c.next_pt) { ex.amax.next_s = c.next_ps; ex.amax.next_t = c.next_pt; ex.amax.next_bstride = c.next_bstride; }
#endif
    }
  }
  const int rc = p2l_conv_fwd_ex(&c.d, &ex, c.x, c.w, c.bias, c.ps, c.pt, c.res, c.mask, c.y, c.yp, skws,
                                 skws_floats * sizeof(float), st);
  if (rc == P2L_OK && R) {
    if (so) R->put(c.y, c.d.B, c.d.H, c.d.W, c.d.Cout, so, ns, set_o, ex.amax.next_s, ex.amax.next_bstride);
    if (sop) R->put(c.yp, c.d.B, c.d.H / 2, c.d.W / 2, c.d.Cout, sop, ns, set_p);
  }
  return rc;
}

// input-gradient conv + backward of the consumer's affine+ReLU; fused in the conv
// epilogue when the layer allows it, otherwise conv -> tmp -> p2l_affine_relu_bwd.
struct ArbArgs {
  const float* x; int x_ld; const float* s; const float* t; int st_bstride;
  const float* skip; int skip_ld, skip_C, skip_ups;
  float* ds; float* dt; int dsdt_bstride;
};
int run_dgrad_arb(ConvCall& c, const ArbArgs& a, float* dx, float* tmp, float* part,
                  float* skws, size_t skws_floats, void* st) {
  c.d.splitk = p2l_conv_suggest_splitk(&c.d);
  const bool pooled = c.d.pool == P2L_POOL_SUM;
  const bool half = pooled || c.d.ups == 3;          // result lives at half resolution
  const int Ho = half ? c.d.H / 2 : c.d.H, Wo = half ? c.d.W / 2 : c.d.W;
  const bool split_fused = c.d.splitk > 1 && p2l_conv_arb_split_fusable(&c.d) &&
                           p2l_conv_workspace_bytes(&c.d) <= skws_floats * sizeof(float);
  if (split_fused || (c.d.splitk == 1 && p2l_conv_arb_fusable(&c.d))) {
    P2LArb arb{};
    arb.x = a.x; arb.x_ld = a.x_ld; arb.s = a.s; arb.t = a.t; arb.st_bstride = a.st_bstride;
    arb.skip = a.skip; arb.skip_ld = a.skip_ld; arb.skip_C = a.skip_C; arb.skip_ups = a.skip_ups;
    arb.ds = a.ds; arb.dt = a.dt; arb.dsdt_bstride = a.dsdt_bstride; arb.partial = part;
    AmaxReg* R = g_amax;
    float* so = nullptr;
    int ns = 0, set_o = -1;
    if (R) {
      if ((c.d.ups == 0 || (c.d.ups == 3 && !c.d.ext)) && c.d.x_ld == c.d.Cin)
        R->get(c.x, c.d.B, c.d.H, c.d.W, c.d.Cin, &arb.amax.in, &arb.amax.in_n, nullptr, 0, nullptr, st, true);
      R->drop(dx);
      ns = c.want_amax ? p2l_conv_amax_slots(&c.d) : 0;
      if (ns > 0 && (size_t)ns * c.d.B <= R->set_floats && (c.d.ups == 0 || (c.d.ups == 3 && !c.d.ext))) {
        so = R->take(&set_o);
        if (pooled) arb.amax.outp = so; else arb.amax.out = so;
      }
    }
    const int rc = p2l_conv_dgrad_arb_ws(&c.d, &arb, c.x, c.w, dx, skws, skws_floats * sizeof(float), st);
    if (rc == P2L_OK && so) R->put(dx, c.d.B, Ho, Wo, c.d.Cout, so, ns, set_o);
    return rc;
  }
  if (pooled) { c.yp = tmp; c.y = nullptr; } else { c.y = tmp; c.yp = nullptr; }
  RET_IF(run_conv(c, skws, skws_floats, st));
  amax_drop(dx);
  return p2l_affine_relu_bwd(tmp, c.d.Cout, a.x, a.x_ld, a.s, a.t, a.st_bstride, a.skip,
                             a.skip_ld, a.skip_C, a.skip_ups, dx, c.d.Cout, a.ds, a.dt,
                             a.dsdt_bstride, part, c.d.B, Ho, Wo, c.d.Cout, st);
}

// ---------------------------------------------------------------------------
// BigGAN workspace layout
// ---------------------------------------------------------------------------
struct BGBlockOff {
  size_t h1, h2, h3, y;
  size_t pf;  // floats of one activation-backward partial buffer of this block
  int H, Ho;  // input / output resolution
};
struct BGLayout {
  size_t cond, raw, s, t, ds, dt, draw, dcond;
  size_t x0;  // gen_z output
  BGBlockOff blk[P2L_MAX_BLOCKS];
  // attention
  size_t att_theta, att_phi, att_phi_p, att_g, att_g_p, att_P, att_ag, att_y;
  size_t att_lse, att_ws, att_ws_floats;   // fused attention (p2l_attn.hip): row statistic, images
  int att_fused;
  int att_H;
  // backward temporaries
  size_t g_a, g_b, g_c, g_d;  // gradient scratch buffers (max activation size)
  size_t arb_partial, tail_pf;
  size_t d_theta, d_phi_p, d_phi, d_g_p, d_g, d_ag;
  size_t skws; size_t skws_floats;
  size_t amax_ring, amax_set_floats, amax_set_stride;   // AmaxReg: NSETS sets of [B][max slots per image] (+ compacted)
  size_t total;
  int out_res;
};

int bg_layout(const P2LBigGAN* m, int B, BGLayout& L) {
  if (!m || m->n_blocks < 1 || m->n_blocks > P2L_MAX_BLOCKS || B < 1) return P2L_EINVAL;
  // kernel choice (hence split-K workspace and activation-backward partial rows) follows the
  // weight format: the size query and the run must lay the arena out for the same one
  g_plan_wfmt = m->wfmt;
  Arena a;
  const int cond = m->z_dim + m->c_dim;
  L.cond = a.take((size_t)B * cond);
  L.raw = a.take((size_t)B * 2 * m->cbn_total);
  L.s = a.take((size_t)B * m->cbn_total);
  L.t = a.take((size_t)B * m->cbn_total);
  L.ds = a.take((size_t)B * m->cbn_total);
  L.dt = a.take((size_t)B * m->cbn_total);
  L.draw = a.take((size_t)B * 2 * m->cbn_total);
  L.dcond = a.take((size_t)B * cond);
  int H = 4;
  L.x0 = a.take((size_t)B * H * H * 16 * m->ch);
  size_t max_act = (size_t)B * H * H * 16 * m->ch;
  // every activation-backward of the backward pass gets its own partial-sum buffer: their
  // second reduction stage is deferred and runs as one launch (p2l_arb_defer_*)
  size_t sum_partial = 0, max_sk = 0;
  int max_slots = 0;
  auto upd_sk = [&](int Hc, int Cin, int Cout, int taps) {
    ConvCall c = mk_conv(B, Hc, Hc, Cin, Cout, taps);
    const size_t f = conv_ws_floats(c);
    if (f > max_sk) max_sk = f;
    const int ns = conv_amax_slots(c);
    if (ns > max_slots) max_slots = ns;
  };
  for (int i = 0; i < m->n_blocks; ++i) {
    if (i == m->attn_before) {
      const int C = m->attn_ch;
      const size_t P = (size_t)H * H;
      L.att_H = H;
      L.att_theta = a.take(B * P * (C / 8));
      L.att_phi = a.take(B * P * (C / 8));
      L.att_phi_p = a.take(B * (P / 4) * (C / 8));
      L.att_g = a.take(B * P * (C / 2));
      L.att_g_p = a.take(B * (P / 4) * (C / 2));
      {
        // fused form (p2l_attn.hip): the P x P/4 matrix is never stored; shapes it does not
        // take keep the GEMM + softmax sequence and its matrix
        P2LAttn ad{B, (int)P, (int)(P / 4), C / 8, C / 2};
        // (a model built for the exact-fp32 MFMA keeps the exact-fp32 GEMMs here too)
        L.att_fused = !(m->wfmt & P2L_WFMT_FLAG_ATTN_GEMM) && (m->wfmt & 0xF) != P2L_WFMT_F32 &&
                      p2l_attn_supported(&ad) && (P % 256 == 0);
        if (L.att_fused) {
          size_t wsb = p2l_attn_fwd_ws_bytes(&ad);
          if (p2l_attn_bwd_dv_ws_bytes(&ad) > wsb) wsb = p2l_attn_bwd_dv_ws_bytes(&ad);
          if (p2l_attn_bwd_qk_ws_bytes(&ad) > wsb) wsb = p2l_attn_bwd_qk_ws_bytes(&ad);
          L.att_ws_floats = (wsb + 3) / 4;
          L.att_ws = a.take(L.att_ws_floats);
          L.att_lse = a.take(B * P);
          L.att_P = 0;
        } else {
          L.att_P = a.take(B * P * (P / 4));
        }
      }
      L.att_ag = a.take(B * P * (C / 2));
      L.att_y = a.take(B * P * C);
      L.d_theta = a.take(B * P * (C / 8));
      L.d_phi_p = a.take(B * (P / 4) * (C / 8));
      L.d_phi = a.take(B * P * (C / 8));
      L.d_g_p = a.take(B * (P / 4) * (C / 2));
      L.d_g = a.take(B * P * (C / 2));
      L.d_ag = a.take(B * P * (C / 2));
      if (B * P * (P / 4) > max_act) max_act = B * P * (P / 4);  // dP temp
      {
        // the two K = P products of the backward pass (d g_p, d phi_p) split K when B is small
        P2LGemm q{};
        q.batch = B; q.M = (int)P / 4; q.K = (int)P;
        q.N = C / 2;
        size_t f = p2l_gemm_ws_bytes(&q) / sizeof(float);
        if (f > max_sk) max_sk = f;
        q.N = C / 8;
        f = p2l_gemm_ws_bytes(&q) / sizeof(float);
        if (f > max_sk) max_sk = f;
      }
    }
    const P2LGenBlock& g = m->blocks[i];
    const int mid = g.cin / 4;
    const int Ho = g.up ? 2 * H : H;
    BGBlockOff& o = L.blk[i];
    o.H = H; o.Ho = Ho;
    o.h1 = a.take((size_t)B * H * H * mid);
    o.h2 = a.take((size_t)B * Ho * Ho * mid);
    o.h3 = a.take((size_t)B * Ho * Ho * mid);
    o.y = a.take((size_t)B * Ho * Ho * g.cout);
    const size_t acts[] = {(size_t)B * H * H * g.cin, (size_t)B * Ho * Ho * mid,
                           (size_t)B * Ho * Ho * g.cout};
    for (size_t v : acts) if (v > max_act) max_act = v;
    {
      // partial sums per (sample, tile | quad, channel) of the four activation-backwards of
      // this block: one per 128-pixel tile in the conv epilogue, one per 2x2 quad when the
      // input-gradient conv splits K (small batches) and its finish kernel does the work
      auto pf_of = [&](int Hc, int Cin, int Cout, int taps, int pool) {
        ConvCall c = mk_conv(B, Hc, Hc, Cin, Cout, taps);
        c.d.pool = pool;
        c.d.splitk = p2l_conv_suggest_splitk(&c.d);
        int nb = p2l_conv_arb_nblk_ws(&c.d);
        const int tiles = cdiv(Hc * Hc, 128);
        if (tiles > nb) nb = tiles;
        return 2 * (size_t)B * nb * Cout;
      };
      o.pf = pf_of(Ho, g.cout, mid, 1, P2L_POOL_NONE);
      size_t f = pf_of(Ho, mid, mid, 9, P2L_POOL_NONE);
      if (f > o.pf) o.pf = f;
      f = pf_of(Ho, mid, mid, 9, g.up ? P2L_POOL_SUM : P2L_POOL_NONE);
      if (f > o.pf) o.pf = f;
      f = pf_of(H, mid, g.cin, 1, P2L_POOL_NONE);
      if (f > o.pf) o.pf = f;
    }
    sum_partial += 4 * o.pf;
    // split-K workspace for fwd and dgrad convs of this block
    upd_sk(H, g.cin, mid, 1);  upd_sk(Ho, mid, mid, 9);  upd_sk(Ho, mid, g.cout, 1);
    upd_sk(Ho, g.cout, mid, 1);  upd_sk(H, mid, g.cin, 1);
    H = Ho;
  }
  L.out_res = H;
  {
    const int nblk = cdiv(H * H, 128);
    L.tail_pf = 2 * (size_t)B * nblk * m->ch;
    sum_partial += L.tail_pf;
    if ((size_t)B * H * H * m->ch > max_act) max_act = (size_t)B * H * H * m->ch;
  }
  L.g_a = a.take(max_act);
  L.g_b = a.take(max_act);
  L.g_c = a.take(max_act);
  L.g_d = a.take(max_act);
  L.arb_partial = a.take(sum_partial);
  L.skws_floats = max_sk;
  L.skws = a.take(max_sk ? max_sk : 64);
  L.amax_set_floats = (size_t)B * max_slots;
  L.amax_set_stride = L.amax_set_floats + (size_t)B * AmaxReg::COMPACT_TO;
  L.amax_ring = a.take(AmaxReg::NSETS * L.amax_set_stride + 64);
  L.total = a.off;
  return P2L_OK;
}

}  // namespace

extern "C" int p2l_vec_scale_div(const float* a, const float* div, float* out, int n,
                                 float scale, void* stream) {
  hipLaunchKernelGGL(vec_scale_div_kernel, dim3(cdiv(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, a, div, out, n, scale);
  return p2l_check_launch();
}
extern "C" int p2l_concat2(const float* z, const float* c, float* cond, int Bn, int nz,
                           int nc, void* stream) {
  hipLaunchKernelGGL(concat2_kernel, dim3(cdiv(Bn * (nz + nc), 256)), dim3(256), 0,
                     (hipStream_t)stream, z, c, cond, Bn, nz, nc);
  return p2l_check_launch();
}
extern "C" int p2l_split2(const float* dcond, float* dz, float* dc, int Bn, int nz,
                          int nc, void* stream) {
  hipLaunchKernelGGL(split2_kernel, dim3(cdiv(Bn * (nz + nc), 256)), dim3(256), 0,
                     (hipStream_t)stream, dcond, dz, dc, Bn, nz, nc);
  return p2l_check_launch();
}

extern "C" size_t p2l_biggan_ws_bytes(const P2LBigGAN* m, int Bn) {
  BGLayout L;
  if (bg_layout(m, Bn, L)) return 0;
  return L.total * sizeof(float);
}

extern "C" int p2l_biggan_ws_lookup(const P2LBigGAN* m, int Bn, int what, int Lidx,
                                    size_t* float_off, int32_t shape[4]) {
  BGLayout L;
  RET_IF(bg_layout(m, Bn, L));
  if (what == 1) {
    *float_off = L.x0; shape[0] = Bn; shape[1] = 4; shape[2] = 4; shape[3] = 16 * m->ch;
    return P2L_OK;
  }
  if (what >= 2 && what <= 5) {
    *float_off = what == 2 ? L.s : what == 3 ? L.t : what == 4 ? L.ds : L.dt;
    shape[0] = Bn; shape[1] = 1; shape[2] = 1; shape[3] = m->cbn_total;
    return P2L_OK;
  }
  if (what == 6) {  // after p2l_biggan_bwd: d loss / d (CBN gains | CBN biases), [B][2*cbn_total]
    *float_off = L.draw; shape[0] = Bn; shape[1] = 1; shape[2] = 1; shape[3] = 2 * m->cbn_total;
    return P2L_OK;
  }
  if (what == 7) {  // inner activations of GenBlock Lidx / 3: input of bn_1 | bn_2 | bn_3
    const int bi = Lidx / 3, kk = Lidx % 3;
    if (bi < 0 || bi >= m->n_blocks) return P2L_EINVAL;
    const BGBlockOff& o = L.blk[bi];
    *float_off = kk == 0 ? o.h1 : kk == 1 ? o.h2 : o.h3;
    const int r = kk == 0 ? o.H : o.Ho;
    shape[0] = Bn; shape[1] = r; shape[2] = r; shape[3] = m->blocks[bi].cin / 4;
    return P2L_OK;
  }
  if (what == 8 || what == 9) {  // un-pooled phi / g of the self-attention
    if (m->attn_before < 0 || m->attn_before >= m->n_blocks) return P2L_EINVAL;
    *float_off = what == 8 ? L.att_phi : L.att_g;
    shape[0] = Bn; shape[1] = L.att_H; shape[2] = L.att_H;
    shape[3] = what == 8 ? m->attn_ch / 8 : m->attn_ch / 2;
    return P2L_OK;
  }
  if (what == 10) {  // the shared split-K workspace (whatever launch used it last)
    *float_off = L.skws; shape[0] = 1; shape[1] = 1; shape[2] = 1; shape[3] = (int32_t)L.skws_floats;
    return P2L_OK;
  }
  if (what == 11) {  // the maxima ring: NSETS sets of [B][slots per image]
    *float_off = L.amax_ring; shape[0] = AmaxReg::NSETS; shape[1] = 1; shape[2] = 1;
    shape[3] = (int32_t)L.amax_set_stride;
    return P2L_OK;
  }
  if (what != 0) return P2L_EINVAL;
  // ModuleList index -> block index (SelfAttn occupies index attn_before)
  int idx = 0;
  for (int i = 0; i < m->n_blocks; ++i) {
    if (i == m->attn_before) {
      if (idx == Lidx) {
        *float_off = L.att_y; shape[0] = Bn; shape[1] = L.att_H; shape[2] = L.att_H;
        shape[3] = m->attn_ch;
        return P2L_OK;
      }
      ++idx;
    }
    if (idx == Lidx) {
      *float_off = L.blk[i].y; shape[0] = Bn; shape[1] = L.blk[i].Ho;
      shape[2] = L.blk[i].Ho; shape[3] = m->blocks[i].cout;
      return P2L_OK;
    }
    ++idx;
  }
  return P2L_EINVAL;
}

extern "C" int p2l_biggan_fwd(const P2LBigGAN* m, const float* z, const float* c, int B,
                              void* ws, size_t ws_bytes, float* img16, void* st) {
  g_plan_wfmt = m ? m->wfmt : P2L_WFMT_F32;
  BGLayout L;
  RET_IF(bg_layout(m, B, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !z || !c || !img16) return P2L_EWS;
  float* W = (float*)ws;
  const int cond = m->z_dim + m->c_dim, CT = m->cbn_total;
  float* skws = W + L.skws;
  AmaxScope amax_scope(W + L.amax_ring, L.amax_set_floats, L.amax_set_stride);

  RET_IF(p2l_concat2(z, c, W + L.cond, B, m->z_dim, m->c_dim, st));
  RET_IF(p2l_linear_fwd(W + L.cond, m->cbn_w, nullptr, W + L.raw, B, cond, 2 * CT, st));
  RET_IF(p2l_cbn_fold_fwd(W + L.raw, W + L.raw + CT, m->cbn_mean, m->cbn_rstd, W + L.s,
                          W + L.t, B, CT, 2 * CT, st));
  RET_IF(p2l_linear_fwd(W + L.cond, m->genz_w, m->genz_b, W + L.x0, B, cond,
                        16 * 16 * m->ch, st));

  const float* x = W + L.x0;
  int Cx = 16 * m->ch;
  for (int i = 0; i < m->n_blocks; ++i) {
    if (i == m->attn_before) {
      const int C = m->attn_ch, H = L.att_H, P = H * H;
      if (C != Cx) return P2L_EINVAL;
      ConvCall th = mk_conv(B, H, H, C, C / 8, 1);
      th.x = x; th.w = m->att_w[0]; th.y = W + L.att_theta;
      RET_IF(run_conv(th, skws, L.skws_floats, st));
      ConvCall ph = mk_conv(B, H, H, C, C / 8, 1);
      ph.x = x; ph.w = m->att_w[1]; ph.y = W + L.att_phi; ph.yp = W + L.att_phi_p;
      ph.d.pool = P2L_POOL_MAX;
      RET_IF(run_conv(ph, skws, L.skws_floats, st));
      ConvCall gg = mk_conv(B, H, H, C, C / 2, 1);
      gg.x = x; gg.w = m->att_w[2]; gg.y = W + L.att_g; gg.yp = W + L.att_g_p;
      gg.d.pool = P2L_POOL_MAX;
      RET_IF(run_conv(gg, skws, L.skws_floats, st));
      if (L.att_fused) {
        // attn_g = softmax(theta phi_p^T) g_p, fused; lse = row statistic for the backward pass
        P2LAttn ad{B, P, P / 4, C / 8, C / 2};
        RET_IF(p2l_attn_fwd(&ad, W + L.att_theta, W + L.att_phi_p, W + L.att_g_p, W + L.att_ag,
                            W + L.att_lse, W + L.att_ws, L.att_ws_floats * sizeof(float), st));
      } else {
      // S = theta^T phi  -> [B, P, P/4]
        P2LGemm g1{};
        g1.batch = B; g1.M = P; g1.N = P / 4; g1.K = C / 8;
        g1.lda = C / 8; g1.ldb = C / 8; g1.ldc = P / 4;
        g1.stride_a = (int64_t)P * (C / 8); g1.stride_b = (int64_t)(P / 4) * (C / 8);
        g1.stride_c = (int64_t)P * (P / 4);
        g1.a_kmajor = 0; g1.b_kmajor = 0; g1.alpha = 1.f; g1.accumulate = 0;
        RET_IF(p2l_gemm_ws(&g1, W + L.att_theta, W + L.att_phi_p, W + L.att_P, skws, L.skws_floats * sizeof(float), st));
        RET_IF(p2l_softmax_fwd(W + L.att_P, W + L.att_P, (int64_t)B * P, P / 4, st));
        // attn_g[p, :] = sum_k P[p,k] g[k, :]
        P2LGemm g2{};
        g2.batch = B; g2.M = P; g2.N = C / 2; g2.K = P / 4;
        g2.lda = P / 4; g2.ldb = C / 2; g2.ldc = C / 2;
        g2.stride_a = (int64_t)P * (P / 4); g2.stride_b = (int64_t)(P / 4) * (C / 2);
        g2.stride_c = (int64_t)P * (C / 2);
        g2.a_kmajor = 0; g2.b_kmajor = 1; g2.alpha = 1.f; g2.accumulate = 0;
        RET_IF(p2l_gemm_ws(&g2, W + L.att_P, W + L.att_g_p, W + L.att_ag, skws, L.skws_floats * sizeof(float), st));
      }
      ConvCall oc = mk_conv(B, H, H, C / 2, C, 1);
      oc.x = W + L.att_ag; oc.w = m->att_w[3]; oc.y = W + L.att_y;
      oc.d.alpha = m->gamma; oc.res = x; oc.d.res_ld = C;
      oc.want_amax = true;                             // (read by this block's conv_0 through cbn_0)
      oc.next_ps = W + L.s + m->blocks[i].cbn_off[0]; oc.next_pt = W + L.t + m->blocks[i].cbn_off[0]; oc.next_bstride = CT;
      RET_IF(run_conv(oc, skws, L.skws_floats, st));
      x = W + L.att_y;
    }
    const P2LGenBlock& g = m->blocks[i];
    const BGBlockOff& o = L.blk[i];
    if (g.cin != Cx) return P2L_EINVAL;
    const int mid = g.cin / 4;
    // conv_0 : relu(cbn_0(x)) 1x1 cin -> mid
    ConvCall c0 = mk_conv(B, o.H, o.H, g.cin, mid, 1);
    c0.x = x; c0.w = g.w[0]; c0.bias = g.b[0]; c0.y = W + o.h1;
    c0.d.pro = P2L_PRO_AFFINE_RELU; c0.d.pro_bstride = CT;
    c0.ps = W + L.s + g.cbn_off[0]; c0.pt = W + L.t + g.cbn_off[0];
    c0.want_amax = true;                               // (conv_1: Winograd / sub-pixel / direct, all fp16 x 2)
    c0.next_ps = W + L.s + g.cbn_off[1]; c0.next_pt = W + L.t + g.cbn_off[1]; c0.next_bstride = CT;
    RET_IF(run_conv(c0, skws, L.skws_floats, st));
    // conv_1 : relu(cbn_1) -> (nearest x2) -> 3x3
    ConvCall c1 = mk_conv(B, o.Ho, o.Ho, mid, mid, 9);
    c1.x = W + o.h1; c1.w = g.w[1]; c1.bias = g.b[1]; c1.y = W + o.h2; c1.d.ups = g.up;
    // sub-pixel form from 16^2 inputs up (8^2 / 4^2 inputs measured: no gain at 18 candidates, -1 ... 2 % at 2-3)
    if (g.up && g.w1_sp && o.H >= 16) { c1.d.ups = 2; c1.w = g.w1_sp; }
    c1.d.pro = P2L_PRO_AFFINE_RELU; c1.d.pro_bstride = CT;
    c1.ps = W + L.s + g.cbn_off[1]; c1.pt = W + L.t + g.cbn_off[1];
    c1.want_amax = true;
    c1.next_ps = W + L.s + g.cbn_off[2]; c1.next_pt = W + L.t + g.cbn_off[2]; c1.next_bstride = CT;
    RET_IF(run_conv(c1, skws, L.skws_floats, st));
    ConvCall c2 = mk_conv(B, o.Ho, o.Ho, mid, mid, 9);
    c2.x = W + o.h2; c2.w = g.w[2]; c2.bias = g.b[2]; c2.y = W + o.h3;
    c2.d.pro = P2L_PRO_AFFINE_RELU; c2.d.pro_bstride = CT;
    c2.ps = W + L.s + g.cbn_off[2]; c2.pt = W + L.t + g.cbn_off[2];
    c2.want_amax = true;                               // (conv_3 is a pointwise conv: fp16 x 2 with the maxima)
    c2.next_ps = W + L.s + g.cbn_off[3]; c2.next_pt = W + L.t + g.cbn_off[3]; c2.next_bstride = CT;
    RET_IF(run_conv(c2, skws, L.skws_floats, st));
    // conv_3 : 1x1 mid -> cout, + shortcut (channel-truncated, nearest x2)
    ConvCall c3 = mk_conv(B, o.Ho, o.Ho, mid, g.cout, 1);
    c3.x = W + o.h3; c3.w = g.w[3]; c3.bias = g.b[3]; c3.y = W + o.y;
    c3.d.pro = P2L_PRO_AFFINE_RELU; c3.d.pro_bstride = CT;
    c3.ps = W + L.s + g.cbn_off[3]; c3.pt = W + L.t + g.cbn_off[3];
    c3.res = x; c3.d.res_ld = g.cin; c3.d.res_ups = g.up;
    c3.want_amax = true;                               // (the next block's conv_0; the attention convs read it raw)
    if (i + 1 < m->n_blocks && i + 1 != m->attn_before) {
      c3.next_ps = W + L.s + m->blocks[i + 1].cbn_off[0]; c3.next_pt = W + L.t + m->blocks[i + 1].cbn_off[0];
      c3.next_bstride = CT;
    }
    RET_IF(run_conv(c3, skws, L.skws_floats, st));
    x = W + o.y;
    Cx = g.cout;
  }
  if (Cx != m->ch) return P2L_EINVAL;
  // tail: relu(bn(x)) -> conv_to_rgb (3 of ch outputs) -> tanh, NHWC16
  ConvCall rc = mk_conv(B, L.out_res, L.out_res, m->ch, 32, 9);
  rc.x = x; rc.w = m->rgb_w; rc.bias = m->rgb_b; rc.y = img16;
  rc.d.pro = P2L_PRO_AFFINE_RELU; rc.d.pro_bstride = 0; rc.ps = m->tail_s; rc.pt = m->tail_t;
  rc.d.act = P2L_ACT_TANH; rc.d.y_ld = 16; rc.d.n_store = 16;
  rc.d.algo_flops = 2.0 * B * L.out_res * L.out_res * (double)m->ch * 3 * 9;
  if (g_plan_wfmt & P2L_WFMT_FLAG_THIN) rc.d.wfmt = P2L_WFMT_BF16X3T;
  RET_IF(run_conv(rc, skws, L.skws_floats, st));
  return P2L_OK;
}

extern "C" int p2l_biggan_bwd(const P2LBigGAN* m, int B, void* ws, size_t ws_bytes,
                              const float* img16, float* dimg16, float* dz, float* dc,
                              void* st) {
  g_plan_wfmt = m ? m->wfmt : P2L_WFMT_F32;
#if defined(P2L_AB_BWD_BF3) || defined(P2L_AB_BWD_BF3_BG)
  struct FormScope { FormScope() { g_plan_form = P2L_FORM_WINO_BF3; } ~FormScope() { g_plan_form = 0; } } form_scope;
#endif
  BGLayout L;
  RET_IF(bg_layout(m, B, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !img16 || !dimg16 || !dz || !dc)
    return P2L_EWS;
  float* W = (float*)ws;
  const int cond = m->z_dim + m->c_dim, CT = m->cbn_total;
  float* skws = W + L.skws;
  float* part = W + L.arb_partial;
  AmaxScope amax_scope(W + L.amax_ring, L.amax_set_floats, L.amax_set_stride);
  // ds / dt of the ~50 activation-backwards are only needed by the conditioning gradient at
  // the very end: record their second reduction stage and run it as ONE launch
  struct ArbDefer {
    ArbDefer() { p2l_arb_defer_begin(); }
    ~ArbDefer() { p2l_arb_defer_cancel(); }      // no-op after a flush; cleans up error paths
  } arb_defer;
  float* ga = W + L.g_a;   // gradient w.r.t. the current layer's output
  float* gb = W + L.g_b;   // scratch
  float* gc = W + L.g_c;   // scratch
  float* gd = W + L.g_d;   // scratch

  const int R = L.out_res;
  RET_IF(p2l_tanh_bwd16(img16, dimg16, (int64_t)B * R * R, st));
  {
    // conv_to_rgb input-gradient fused with the unconditional BN+ReLU backward.
    ConvCall c = mk_conv(B, R, R, 16, m->ch, 9);
    c.x = dimg16; c.w = m->rgb_wt;
    c.d.algo_flops = 2.0 * B * R * R * (double)m->ch * 3 * 9;
    if (g_plan_wfmt & P2L_WFMT_FLAG_THIN) c.d.wfmt = P2L_WFMT_BF16X3T;
    const float* xlast = W + L.blk[m->n_blocks - 1].y;
    // ds/dt of the tail BN feed nothing (no conditioning): park them in `draw`.
    ArbArgs a{xlast, m->ch, m->tail_s, m->tail_t, 0, nullptr, 0, 0, 0, W + L.draw,
              W + L.draw + (size_t)B * m->ch, m->ch};
    c.want_amax = true;                                  // (conv_3's input gradient of the last block reads ga)
    RET_IF(run_dgrad_arb(c, a, ga, gd, part, skws, L.skws_floats, st));
    part += L.tail_pf;
  }
  for (int i = m->n_blocks - 1; i >= 0; --i) {
    const P2LGenBlock& g = m->blocks[i];
    const BGBlockOff& o = L.blk[i];
    const int mid = g.cin / 4;
    const float* xin;
    if (i == m->attn_before) xin = W + L.att_y;
    else if (i == 0) xin = W + L.x0;
    else xin = W + L.blk[i - 1].y;

    // ga = dy (kept for the shortcut), gb / gc alternate, gd = temp of the unfused path
    // conv_3 input-gradient: dy[cout] -> [mid], + relu(cbn_3) backward
    ConvCall d3 = mk_conv(B, o.Ho, o.Ho, g.cout, mid, 1);
    d3.x = ga; d3.w = g.wt[3];
    ArbArgs a3{W + o.h3, mid, W + L.s + g.cbn_off[3], W + L.t + g.cbn_off[3], CT, nullptr, 0,
               0, 0, W + L.ds + g.cbn_off[3], W + L.dt + g.cbn_off[3], CT};
    d3.want_amax = true;
    RET_IF(run_dgrad_arb(d3, a3, gb, gd, part, skws, L.skws_floats, st));
    part += o.pf;
    // conv_2
    ConvCall d2 = mk_conv(B, o.Ho, o.Ho, mid, mid, 9);
    d2.x = gb; d2.w = g.wt[2];
    ArbArgs a2{W + o.h2, mid, W + L.s + g.cbn_off[2], W + L.t + g.cbn_off[2], CT, nullptr, 0,
               0, 0, W + L.ds + g.cbn_off[2], W + L.dt + g.cbn_off[2], CT};
    d2.want_amax = true;
    RET_IF(run_dgrad_arb(d2, a2, gc, gd, part, skws, L.skws_floats, st));
    part += o.pf;
    // conv_1 (+ nearest-x2 backward = 2x2 sum pool fused in the epilogue)
    ConvCall d1 = mk_conv(B, o.Ho, o.Ho, mid, mid, 9);
    d1.x = gc; d1.w = g.wt[1];
    if (g.up && g.wt1_sp && o.H >= 32) { d1.d.ups = 3; d1.w = g.wt1_sp; }   // sub-pixel form
    else if (g.up) d1.d.pool = P2L_POOL_SUM;
    ArbArgs a1{W + o.h1, mid, W + L.s + g.cbn_off[1], W + L.t + g.cbn_off[1], CT, nullptr, 0,
               0, 0, W + L.ds + g.cbn_off[1], W + L.dt + g.cbn_off[1], CT};
    d1.want_amax = true;
    RET_IF(run_dgrad_arb(d1, a1, gb, gd, part, skws, L.skws_floats, st));
    part += o.pf;
    // conv_0, + relu(cbn_0) backward + shortcut gradient from dy (= ga)
    ConvCall d0 = mk_conv(B, o.H, o.H, mid, g.cin, 1);
    d0.x = gb; d0.w = g.wt[0];
    const int skipC = (g.cin != g.cout) ? g.cin / 2 : g.cin;
    ArbArgs a0{xin, g.cin, W + L.s + g.cbn_off[0], W + L.t + g.cbn_off[0], CT, ga, g.cout,
               skipC, g.up, W + L.ds + g.cbn_off[0], W + L.dt + g.cbn_off[0], CT};
    d0.want_amax = true;
    RET_IF(run_dgrad_arb(d0, a0, gc, gd, part, skws, L.skws_floats, st));
    part += o.pf;
    { float* tmp = ga; ga = gc; gc = tmp; }

    if (i == m->attn_before) {
      // ---- SelfAttn backward; ga = d out [B,H,H,C] ------------------------
      // (gb becomes a scratch matrix of the attention kernels: ITS maxima die.  Until round 5 every entry was
      //  cleared here, the incoming gradient's too, and the o_conv gradient below ran in bf16 x 3 for want of them)
      amax_drop(gb);
      const int C = m->attn_ch, H = L.att_H, P = H * H;
      const float* x = (i == 0) ? W + L.x0 : W + L.blk[i - 1].y;
      (void)x;
      // d attn_g = gamma * dgrad(o_conv)(dy)
      ConvCall dob = mk_conv(B, H, H, C, C / 2, 1);
      dob.x = ga; dob.w = m->att_wt[3]; dob.y = W + L.d_ag; dob.d.alpha = m->gamma;
      RET_IF(run_conv(dob, skws, L.skws_floats, st));
      if (L.att_fused) {
        // d g_p = P^T d_ag with P recomputed from lse; then d theta, d phi_p: P and dP are
        // recomputed tile by tile, dS^T goes once through the scratch buffer gb
        P2LAttn ad{B, P, P / 4, C / 8, C / 2};
        RET_IF(p2l_attn_bwd_dv(&ad, W + L.att_theta, W + L.att_phi_p, W + L.d_ag, W + L.att_lse,
                               W + L.d_g_p, W + L.att_ws, L.att_ws_floats * sizeof(float), st));
        RET_IF(p2l_attn_bwd_qk(&ad, W + L.att_theta, W + L.att_phi_p, W + L.att_g_p, W + L.att_ag,
                               W + L.d_ag, W + L.att_lse, gb, W + L.d_theta, W + L.d_phi_p,
                               W + L.att_ws, L.att_ws_floats * sizeof(float), st));
      } else {
      // dP[p,k] = sum_c d_ag[p,c] g_p[k,c]
        P2LGemm q1{};
        q1.batch = B; q1.M = P; q1.N = P / 4; q1.K = C / 2;
        q1.lda = C / 2; q1.ldb = C / 2; q1.ldc = P / 4;
        q1.stride_a = (int64_t)P * (C / 2); q1.stride_b = (int64_t)(P / 4) * (C / 2);
        q1.stride_c = (int64_t)P * (P / 4);
        q1.alpha = 1.f;
        RET_IF(p2l_gemm_ws(&q1, W + L.d_ag, W + L.att_g_p, gb, skws, L.skws_floats * sizeof(float), st));
        // d g_p[k,c] = sum_p P[p,k] d_ag[p,c]
        P2LGemm q2{};
        q2.batch = B; q2.M = P / 4; q2.N = C / 2; q2.K = P;
        q2.lda = P / 4; q2.ldb = C / 2; q2.ldc = C / 2;
        q2.stride_a = (int64_t)P * (P / 4); q2.stride_b = (int64_t)P * (C / 2);
        q2.stride_c = (int64_t)(P / 4) * (C / 2);
        q2.a_kmajor = 1; q2.b_kmajor = 1; q2.alpha = 1.f;
        RET_IF(p2l_gemm_ws(&q2, W + L.att_P, W + L.d_ag, W + L.d_g_p, skws, L.skws_floats * sizeof(float), st));
        // dS = softmax backward (in place in gb)
        RET_IF(p2l_softmax_bwd(W + L.att_P, gb, gb, (int64_t)B * P, P / 4, st));
        // d theta[p,d] = sum_k dS[p,k] phi_p[k,d]
        P2LGemm q3{};
        q3.batch = B; q3.M = P; q3.N = C / 8; q3.K = P / 4;
        q3.lda = P / 4; q3.ldb = C / 8; q3.ldc = C / 8;
        q3.stride_a = (int64_t)P * (P / 4); q3.stride_b = (int64_t)(P / 4) * (C / 8);
        q3.stride_c = (int64_t)P * (C / 8);
        q3.b_kmajor = 1; q3.alpha = 1.f;
        RET_IF(p2l_gemm_ws(&q3, gb, W + L.att_phi_p, W + L.d_theta, skws, L.skws_floats * sizeof(float), st));
        // d phi_p[k,d] = sum_p dS[p,k] theta[p,d]
        P2LGemm q4{};
        q4.batch = B; q4.M = P / 4; q4.N = C / 8; q4.K = P;
        q4.lda = P / 4; q4.ldb = C / 8; q4.ldc = C / 8;
        q4.stride_a = (int64_t)P * (P / 4); q4.stride_b = (int64_t)P * (C / 8);
        q4.stride_c = (int64_t)(P / 4) * (C / 8);
        q4.a_kmajor = 1; q4.b_kmajor = 1; q4.alpha = 1.f;
        RET_IF(p2l_gemm_ws(&q4, gb, W + L.att_theta, W + L.d_phi_p, skws, L.skws_floats * sizeof(float), st));
      }
      // max-pool backward for phi and g
      // (the max-pool backward leaves the maxima of what it writes for the 1x1 convs t2 / t3 below,
      //  which take the fp16 x 2 form with them)
      auto pool_bwd = [&](const float* y, const float* dyp, float* dy, int Cc) -> int {
        float* so = nullptr;
        int set_o = -1;
        const int ns = p2l_maxpool2_bwd_amax_slots(H, H, Cc);
        amax_drop(dy);
        if (g_amax && ns > 0 && (size_t)ns * B <= g_amax->set_floats) so = g_amax->take(&set_o);
        RET_IF(p2l_maxpool2_bwd_amax(y, Cc, dyp, Cc, nullptr, 0, dy, Cc, B, H, H, Cc, 0, so, st));
        if (so) g_amax->put(dy, B, H, H, Cc, so, ns, set_o);
        return P2L_OK;
      };
      RET_IF(pool_bwd(W + L.att_phi, W + L.d_phi_p, W + L.d_phi, C / 8));
      RET_IF(pool_bwd(W + L.att_g, W + L.d_g_p, W + L.d_g, C / 2));
      // dx = dy + theta^T-grad + phi-grad + g-grad (chained residual, in place)
      ConvCall t1 = mk_conv(B, H, H, C / 8, C, 1);
      t1.x = W + L.d_theta; t1.w = m->att_wt[0]; t1.y = ga; t1.res = ga; t1.d.res_ld = C;
      RET_IF(run_conv(t1, skws, L.skws_floats, st));
      ConvCall t2 = mk_conv(B, H, H, C / 8, C, 1);
      t2.x = W + L.d_phi; t2.w = m->att_wt[1]; t2.y = ga; t2.res = ga; t2.d.res_ld = C;
      RET_IF(run_conv(t2, skws, L.skws_floats, st));
      ConvCall t3 = mk_conv(B, H, H, C / 2, C, 1);
      t3.x = W + L.d_g; t3.w = m->att_wt[2]; t3.y = ga; t3.res = ga; t3.d.res_ld = C;
      t3.want_amax = true;                               // (ga feeds the next block's conv_3 input gradient)
      RET_IF(run_conv(t3, skws, L.skws_floats, st));
    }
  }
  // ga = d gen_z output [B, 16*16*ch]; conditioning gradients
  RET_IF(p2l_linear_bwd(ga, m->genz_w, W + L.dcond, B, cond, 16 * 16 * m->ch, 0, st));
  RET_IF(p2l_arb_defer_flush(st));
  RET_IF(p2l_cbn_fold_bwd(W + L.ds, W + L.dt, m->cbn_mean, m->cbn_rstd, W + L.draw,
                          W + L.draw + CT, B, CT, 2 * CT, st));
  RET_IF(p2l_linear_bwd(W + L.draw, m->cbn_w, W + L.dcond, B, cond, 2 * CT, 1, st));
  RET_IF(p2l_split2(W + L.dcond, dz, dc, B, m->z_dim, m->c_dim, st));
  return P2L_OK;
}

// ===========================================================================
// ProjectionLoss = weighted L1 + beta * weighted LPIPS(VGG16)
// ===========================================================================
namespace {

const int kVggCin[13] = {16, 64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512};
const int kVggCout[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
// resolution divisor of each conv's output, pool after convs 1,3,6,9
const int kVggDiv[13] = {1, 1, 2, 2, 4, 4, 4, 8, 8, 8, 16, 16, 16};
const int kVggTapConv[5] = {1, 3, 6, 9, 12};
inline bool vgg_pool_after(int i) { return i == 1 || i == 3 || i == 6 || i == 9; }
inline int vgg_tap_of(int i) {
  for (int k = 0; k < 5; ++k) if (kVggTapConv[k] == i) return k;
  return -1;
}

struct PLLayout {
  size_t y[13];        // post-ReLU conv outputs
  size_t yp[4];        // pooled outputs after convs 1,3,6,9
  size_t tgt16;        // prepare: target in NHWC16
  size_t wsrc;         // prepare: per-pixel weight map
  size_t part;         // loss partial sums
  size_t lp, l1;       // per-sample partial losses
  size_t gs;           // per-sample gradient scale
  size_t ga, gb, gtap; // backward scratch
  size_t skws, skws_floats;
  size_t amax_ring, amax_set_floats, amax_set_stride;   // AmaxReg
  size_t total;
};

int pl_layout(int B, int H, int W, PLLayout& L) {
  if (B < 1 || H < 32 || W < 32 || !is_pow2(H) || !is_pow2(W)) return P2L_EINVAL;
  Arena a;
  size_t max_act = 0, max_sk = 0;
  int pi = 0, max_slots = 0;
  for (int i = 0; i < 13; ++i) {
    const int h = H / kVggDiv[i], w = W / kVggDiv[i];
    const size_t n = (size_t)B * h * w * kVggCout[i];
    L.y[i] = a.take(n);
    if (n > max_act) max_act = n;
    if (vgg_pool_after(i)) L.yp[pi++] = a.take(n / 4);
    ConvCall f = mk_conv(B, h, w, kVggCin[i], kVggCout[i], 9);
    size_t s1 = conv_ws_floats(f);
    ConvCall d = mk_conv(B, h, w, kVggCout[i], i == 0 ? 32 : kVggCin[i], 9);
    size_t s2 = conv_ws_floats(d);
    if (s1 > max_sk) max_sk = s1;
    if (s2 > max_sk) max_sk = s2;
    const int n1 = conv_amax_slots(f), n2 = conv_amax_slots(d);
    if (n1 > max_slots) max_slots = n1;
    if (n2 > max_slots) max_slots = n2;
    if (vgg_pool_after(i) && i >= 1) {
      const int n3 = p2l_maxpool2_bwd_amax_slots(h, w, kVggCout[i]);
      if (n3 > max_slots) max_slots = n3;
      const int n4 = p2l_lpips_tap_nblk(h * w, kVggCout[i]);      // (the fused tap + pool backward)
      if (n4 > max_slots) max_slots = n4;
    }
  }
  L.tgt16 = a.take((size_t)B * H * W * 16);
  L.wsrc = a.take((size_t)B * H * W);
  size_t maxpart = (size_t)p2l_l1_loss_nblk(H, W);
  for (int k = 0; k < 5; ++k) {
    const int d = kVggDiv[kVggTapConv[k]];
    const size_t nb = (size_t)p2l_lpips_tap_nblk((H / d) * (W / d), kVggCout[kVggTapConv[k]]);
    if (nb > maxpart) maxpart = nb;
  }
  L.part = a.take((size_t)B * maxpart);
  L.lp = a.take(B);
  L.l1 = a.take(B);
  L.gs = a.take(B);
  L.ga = a.take(max_act);
  L.gb = a.take(max_act);
  L.gtap = a.take(max_act);
  L.skws_floats = max_sk;
  L.skws = a.take(max_sk ? max_sk : 64);
  L.amax_set_floats = (size_t)B * max_slots;
  L.amax_set_stride = L.amax_set_floats + (size_t)B * AmaxReg::COMPACT_TO;
  L.amax_ring = a.take(AmaxReg::NSETS * L.amax_set_stride + 64);
  L.total = a.off;
  return P2L_OK;
}

// VGG16 features forward on an NHWC16 image; y[i] = relu(conv_i), pooled copies.
int vgg_forward(const P2LVggLpips* v, const float* img16, int B, int H, int W, float* Wk,
                const PLLayout& L, void* st) {
  AmaxScope amax_scope(Wk + L.amax_ring, L.amax_set_floats, L.amax_set_stride);
  const float* x = img16;
  int pi = 0;
  for (int i = 0; i < 13; ++i) {
    const int h = H / kVggDiv[i], w = W / kVggDiv[i];
    ConvCall c = mk_conv(B, h, w, kVggCin[i], kVggCout[i], 9);
    c.x = x; c.w = v->w[i]; c.bias = v->b[i]; c.y = Wk + L.y[i];
    c.d.act = P2L_ACT_RELU;
    if (i == 0) {
      c.d.pro = P2L_PRO_AFFINE; c.d.pro_bstride = 0; c.ps = v->in_s; c.pt = v->in_t;
      c.d.algo_flops = 2.0 * B * h * w * 3.0 * kVggCout[0] * 9;
      if (g_plan_wfmt & P2L_WFMT_FLAG_THIN) c.d.wfmt = P2L_WFMT_BF16X3T;
    }
    if (vgg_pool_after(i)) {
      c.d.pool = P2L_POOL_MAX; c.yp = Wk + L.yp[pi];
    }
    c.want_amax = i < 12;
    RET_IF(run_conv(c, Wk + L.skws, L.skws_floats, st));
    if (vgg_pool_after(i)) { x = Wk + L.yp[pi]; ++pi; }
    else x = Wk + L.y[i];
  }
  return P2L_OK;
}

}  // namespace

extern "C" size_t p2l_loss_cache_floats(int B, int H, int W, size_t nft_off[5],
                                        size_t wt_off[5], size_t* wsum_off) {
  Arena a;
  for (int k = 0; k < 5; ++k) {
    const int d = kVggDiv[kVggTapConv[k]];
    const size_t P = (size_t)(H / d) * (W / d);
    nft_off[k] = a.take((size_t)B * P * kVggCout[kVggTapConv[k]]);
    wt_off[k] = a.take((size_t)B * P);
  }
  *wsum_off = a.take(B);
  return a.off;
}

// debug/test hook: float offset + shape of the post-ReLU output of VGG conv `idx` (0..12) in ws
extern "C" int p2l_projloss_ws_lookup(int B, int H, int W, int idx, size_t* float_off,
                                      int32_t shape[4]) {
  if (idx < 0 || idx >= 13 || !float_off || !shape) return P2L_EINVAL;
  PLLayout L;
  RET_IF(pl_layout(B, H, W, L));
  *float_off = L.y[idx];
  shape[0] = B; shape[1] = H / kVggDiv[idx]; shape[2] = W / kVggDiv[idx]; shape[3] = kVggCout[idx];
  return P2L_OK;
}

extern "C" size_t p2l_projloss_ws_bytes(int B, int H, int W) {
  // no descriptor here: size for the format with the largest split-K workspace (the last
  // region of the arena; every other offset is format-independent): the exact-fp32 kernels split
  // K by the grid size, the Winograd format slices its small-grid layers by their shape
  size_t total = 0;
  for (int fmt : {(int)P2L_WFMT_F32, (int)P2L_WFMT_BF16X3W, (int)P2L_WFMT_BF16X3}) {
    g_plan_wfmt = fmt;
    PLLayout L;
    if (pl_layout(B, H, W, L)) return 0;
    if (L.total > total) total = L.total;
  }
  g_plan_wfmt = P2L_WFMT_F32;
  return total * sizeof(float);
}

extern "C" int p2l_projloss_prepare(const P2LVggLpips* v, const float* target,
                                    const float* weight, const float* loss_mask, int B,
                                    int H, int W, const P2LLossCache* cache, void* ws,
                                    size_t ws_bytes, void* st) {
  g_plan_wfmt = v ? v->wfmt : P2L_WFMT_F32;
  PLLayout L;
  RET_IF(pl_layout(B, H, W, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !cache || !target) return P2L_EWS;
  float* Wk = (float*)ws;
  if (weight) {
    RET_IF(p2l_weight_sum(weight, loss_mask, cache->wsum, B, 3 * H * W, st));
    RET_IF(p2l_weight_map(weight, loss_mask, Wk + L.wsrc, B, H, W, st));
    for (int k = 0; k < 5; ++k) {
      const int d = kVggDiv[kVggTapConv[k]];
      RET_IF(p2l_bilinear_adjoint(Wk + L.wsrc, cache->wt[k], B, H, W, H / d, W / d, st));
    }
  }
  if (v) {
    RET_IF(p2l_nchw3_to_nhwc16(target, Wk + L.tgt16, B, H, W, st));
    RET_IF(vgg_forward(v, Wk + L.tgt16, B, H, W, Wk, L, st));
    for (int k = 0; k < 5; ++k) {
      const int ci = kVggTapConv[k], d = kVggDiv[ci];
      RET_IF(p2l_lpips_normalize(Wk + L.y[ci], cache->nft[k],
                                 (int64_t)B * (H / d) * (W / d), kVggCout[ci], st));
    }
  }
  return P2L_OK;
}

extern "C" int p2l_projloss_fwd(const P2LVggLpips* v, const float* img16,
                                const float* target, const float* weight,
                                const float* loss_mask, const P2LLossCache* cache,
                                float beta, int use_lpips, int B, int H, int W, void* ws,
                                size_t ws_bytes, float* loss, float* loss_l1,
                                float* loss_lpips, void* st) {
  g_plan_wfmt = v ? v->wfmt : P2L_WFMT_F32;
  PLLayout L;
  RET_IF(pl_layout(B, H, W, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !cache || !img16 || !loss) return P2L_EWS;
  float* Wk = (float*)ws;
  float* l1 = loss_l1 ? loss_l1 : Wk + L.l1;
  float* lp = loss_lpips ? loss_lpips : Wk + L.lp;
  RET_IF(p2l_l1_loss_fwd(img16, target, weight, loss_mask, cache->wsum, l1, Wk + L.part, B,
                         H, W, st));
  RET_IF(p2l_vec_scale_div(l1, nullptr, loss, B, 1.f, st));
  if (use_lpips) {
    if (!v) return P2L_EINVAL;
    RET_IF(vgg_forward(v, img16, B, H, W, Wk, L, st));
    for (int k = 0; k < 5; ++k) {
      const int ci = kVggTapConv[k], d = kVggDiv[ci];
      const int P = (H / d) * (W / d), C = kVggCout[ci];
      const int nblk = p2l_lpips_tap_nblk(P, C);
      RET_IF(p2l_lpips_tap_fwd(Wk + L.y[ci], cache->nft[k], (int64_t)P * C, v->lin[k],
                               cache->wt[k], P, Wk + L.part, B, P, C, st));
      RET_IF(p2l_reduce_rows(Wk + L.part, lp, B, nblk, 1.f, cache->wsum, k > 0, st));
    }
    RET_IF(p2l_reduce_rows(lp, loss, B, 1, beta, nullptr, 1, st));
  }
  return P2L_OK;
}

extern "C" int p2l_projloss_bwd(const P2LVggLpips* v, const float* img16,
                                const float* target, const float* weight,
                                const float* loss_mask, const P2LLossCache* cache,
                                float beta, int use_lpips, const float* gloss, int B, int H,
                                int W, void* ws, size_t ws_bytes, float* dimg16, void* st) {
  g_plan_wfmt = v ? v->wfmt : P2L_WFMT_F32;
#if defined(P2L_AB_BWD_BF3) || defined(P2L_AB_BWD_BF3_PL)
  struct FormScope { FormScope() { g_plan_form = P2L_FORM_WINO_BF3; } ~FormScope() { g_plan_form = 0; } } form_scope;
#endif
  PLLayout L;
  RET_IF(pl_layout(B, H, W, L));
  if (!ws || ws_bytes < L.total * sizeof(float) || !cache || !img16 || !gloss || !dimg16)
    return P2L_EWS;
  float* Wk = (float*)ws;
  if (!use_lpips) {
    return p2l_l1_loss_bwd(img16, target, weight, loss_mask, cache->wsum, gloss, dimg16, B,
                           H, W, 0, st);
  }
  if (!v) return P2L_EINVAL;
  // gs[b] = gloss[b] * beta / wsum[b]
  RET_IF(p2l_vec_scale_div(gloss, cache->wsum, Wk + L.gs, B, beta, st));
  AmaxScope amax_scope(Wk + L.amax_ring, L.amax_set_floats, L.amax_set_stride);
  float* ga = Wk + L.ga;   // gradient w.r.t. the PRE-ReLU output of conv i (masked)
  float* gb = Wk + L.gb;
  float* gtap = Wk + L.gtap;
  // top: relu5_3 is only consumed by the LPIPS tap
  {
    const int ci = 12, d = kVggDiv[ci], P = (H / d) * (W / d), C = kVggCout[ci];
    RET_IF(p2l_lpips_tap_bwd(Wk + L.y[ci], cache->nft[4], (int64_t)P * C, v->lin[4],
                             cache->wt[4], P, Wk + L.gs, gtap, B, P, C, st));
    RET_IF(p2l_relu_mask(Wk + L.y[ci], C, gtap, C, ga, C, (int64_t)B * P, C, st));
  }
  int pi = 3;
  for (int i = 12; i >= 1; --i) {
    const int h = H / kVggDiv[i], w = W / kVggDiv[i];
    // input-gradient of conv i: ga [h,w,Cout_i] -> [h,w,Cin_i]
    ConvCall c = mk_conv(B, h, w, kVggCout[i], kVggCin[i], 9);
    c.x = ga; c.w = v->wt[i]; c.y = gb;
    const int prev = i - 1;
    if (!vgg_pool_after(prev)) {
      // input is relu(conv_{i-1}) directly: fuse its ReLU mask in the epilogue
      c.mask = Wk + L.y[prev]; c.d.mask_ld = kVggCout[prev];
      c.want_amax = i > 1;
      RET_IF(run_conv(c, Wk + L.skws, L.skws_floats, st));
    } else {
      // input is maxpool(relu(conv_{i-1})); conv_{i-1} is also an LPIPS tap
      RET_IF(run_conv(c, Wk + L.skws, L.skws_floats, st));
      const int k = vgg_tap_of(prev);
      const int hp = H / kVggDiv[prev], wp = W / kVggDiv[prev];
      const int P = hp * wp, C = kVggCout[prev];
      // (written by a non-conv kernel, which leaves its own maxima for the dgrad that reads ga next)
      amax_drop(ga);
      float* so = nullptr;
      int set_o = -1;
#ifdef P2L_AB_TAP_POOL_2PASS      // A/B builds: the two-pass form of rounds 1-4
      RET_IF(p2l_lpips_tap_bwd(Wk + L.y[prev], cache->nft[k], (int64_t)P * C, v->lin[k],
                               cache->wt[k], P, Wk + L.gs, gtap, B, P, C, st));
      const int ns = p2l_maxpool2_bwd_amax_slots(hp, wp, C);
      if (g_amax && prev >= 1 && ns > 0 && (size_t)ns * B <= g_amax->set_floats) so = g_amax->take(&set_o);
      RET_IF(p2l_maxpool2_bwd_amax(Wk + L.y[prev], C, gb, C, gtap, C, ga, C, B, hp, wp, C, 1, so, st));
#else
      // tap gradient + pool backward of gb + ReLU mask in one pass over conv_{i-1}'s output
      const int ns = p2l_lpips_tap_nblk(P, C);
      if (g_amax && prev >= 1 && ns > 0 && (size_t)ns * B <= g_amax->set_floats) so = g_amax->take(&set_o);
      RET_IF(p2l_lpips_tap_pool_bwd(Wk + L.y[prev], cache->nft[k], (int64_t)P * C, v->lin[k], cache->wt[k], P,
                                    Wk + L.gs, gb, ga, so, B, hp, wp, C, st));
#endif
      if (so) g_amax->put(ga, B, hp, wp, C, so, ns, set_o);
      --pi;
      continue;  // ga already holds the masked gradient of conv_{i-1}
    }
    { float* t = ga; ga = gb; gb = t; }
  }
  (void)pi;
  // conv 0 input-gradient -> d img16 (1/scale folded into the packed weights)
  {
    ConvCall c = mk_conv(B, H, W, 64, 32, 9);
    c.x = ga; c.w = v->wt[0]; c.y = dimg16; c.d.y_ld = 16; c.d.n_store = 16;
    c.d.algo_flops = 2.0 * B * H * W * 3.0 * 64 * 9;
    if (g_plan_wfmt & P2L_WFMT_FLAG_THIN) c.d.wfmt = P2L_WFMT_BF16X3T;
    RET_IF(run_conv(c, Wk + L.skws, L.skws_floats, st));
  }
  if (use_lpips == 2) return P2L_OK;   // PerceptualLoss on its own: no L1 term
  RET_IF(p2l_l1_loss_bwd(img16, target, weight, loss_mask, cache->wsum, gloss, dimg16, B, H,
                         W, 1, st));
  return P2L_OK;
}
