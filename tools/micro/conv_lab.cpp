// Standalone timing / agreement harness for the 3x3 conv kernels of libp2l_hip.so through the
// C ABI (include/p2l.h) -- no Python, no torch: a gpurun call spends its time on kernels.
//   conv_lab [B]        runs every (layer, kernel form) pair below, prints ms, TFLOP/s and the
//                       largest difference from the direct bf16x3 kernel of the same layer
// build: hipcc --offload-arch=gfx950 -O2 -I include tools/micro/conv_lab.cpp -L pix2latent_amd
//        -lp2l_hip -Wl,-rpath,'$ORIGIN/../../pix2latent_amd' -o tools/micro/conv_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "p2l.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define PK(x) do { int r_ = (x); if (r_ != P2L_OK) { printf("p2l error %d at %s:%d\n", r_, __FILE__, __LINE__); exit(1); } } while (0)

static unsigned long long rng = 88172645463325252ull;
static float urand() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 11) * (1.0 / 9007199254740992.0)); }
static float nrand() { float u = urand() + 1e-12f, v = urand(); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); }

struct Form { const char* name; int wfmt; int form; };

#ifdef P2L_LAB
extern "C" int p2l_lab_set(int abl, void* trace);
// ablations of the hand-scheduled 16x16 Winograd kernel + a phase trace of one block
static int lab_main(int B) {
  const int H = 64, W = 64, Cin = 256, Cout = 256;
  const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * H * W * Cout, nw = (size_t)Cout * Cin * 9;
  std::vector<float> hx(nx), hw(nw);
  for (auto& v : hx) { v = nrand(); if (v < 0.f) v = 0.f; }
  for (auto& v : hw) v = nrand() / sqrtf(9.f * Cin);
  float *dx, *dwo, *dy, *dwp; unsigned long long* dtr;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dwo, nw * 4)); CK(hipMalloc(&dy, ny * 4));
  CK(hipMalloc(&dtr, 8 * 64 * 8 * 8));
  CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dwo, hw.data(), nw * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dwp, p2l_packed_weight_floats(9, Cout, Cin, P2L_WFMT_BF16X3W) * 4));
  PK(p2l_pack_conv_weight_bf3w(dwo, Cout, Cin, 9, Cout, Cin, 0, dwp, st));
  P2LConv d; memset(&d, 0, sizeof d);
  d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.taps = 9; d.x_ld = Cin; d.pro = P2L_PRO_NONE;
  d.pro_bstride = Cin; d.alpha = 1.f; d.y_ld = Cout; d.n_store = Cout; d.splitk = 1; d.wfmt = P2L_WFMT_BF16X3W; d.form = P2L_FORM_WINO_ANY;
  void* dws; CK(hipMalloc(&dws, (size_t)B * 64 * 4));
  const size_t wsb = (size_t)B * 64 * 4;
  for (int arith = 0; arith < 3; ++arith) {
  d.form = P2L_FORM_WINO_ANY | (arith == 0 ? P2L_FORM_WINO_BF3 : arith == 1 ? P2L_FORM_WINO_H2_16X16 : P2L_FORM_WINO_H2_8X16);
  printf("---- %s\n", arith == 0 ? "bf16 x 3" : arith == 1 ? "fp16 x 2, 16x16-pixel blocks of 8 waves (times include the max-|x| pass)"
                                                              : "fp16 x 2, 8x16-pixel blocks of 4 waves (round 5)");
  const struct { int abl; const char* what; } A[] = {
      {0, "full"}, {1, "weights once"}, {2, "no transform"}, {4, "no barriers"}, {8, "no m (/l) pieces"},
      {16, "no MFMAs"}, {64, "no patch traffic"}, {3, "weights once, no transform"},
      {10, "no transform, no m/l"}, {67, "no weights/transform/patch"},
      {75, "no weights/transform/patch/ml"}, {79, "... and no barriers"}, {111, "MFMAs + epilogue only"},
      {128, "plain stores instead of the shared epilogue item (fp16 x 2 only)"}, {256, "no output stores (fp16 x 2 only)"},
      {0, "full (again)"}};
  // the clocks of an idle GPU take tens of milliseconds to settle: warm up, then two passes
  PK(p2l_lab_set(0, nullptr));
  for (int i = 0; i < 400; ++i)
    PK(p2l_conv_fwd(&d, dx, dwp, nullptr, nullptr, nullptr, nullptr, nullptr, dy, nullptr, dws, wsb, st));
  CK(hipStreamSynchronize(st));
  for (int pass = 0; pass < 2; ++pass)
  for (auto a : A) {
    if (arith == 0 && a.abl >= 128) continue;
    if (arith == 2 && !(a.abl == 0 || a.abl == 1 || a.abl == 2 || a.abl == 4 || a.abl == 8 || a.abl == 16 ||
                        a.abl == 64 || a.abl == 3 || a.abl == 67 || a.abl == 79)) continue;
    PK(p2l_lab_set(a.abl, nullptr));
    for (int i = 0; i < 3; ++i)
      PK(p2l_conv_fwd(&d, dx, dwp, nullptr, nullptr, nullptr, nullptr, nullptr, dy, nullptr, dws, wsb, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 20; ++i)
      PK(p2l_conv_fwd(&d, dx, dwp, nullptr, nullptr, nullptr, nullptr, nullptr, dy, nullptr, dws, wsb, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("pass %d abl %4d  %.4f ms   %s\n", pass, a.abl, ms / 20, a.what);
  }
  if (arith == 2) continue;               // (the trace layout is the 8-wave block's)
  // phase trace of the full kernel
  CK(hipMemset(dtr, 0, 8 * 64 * 8 * 8));
  PK(p2l_lab_set(0, dtr));
  PK(p2l_conv_fwd(&d, dx, dwp, nullptr, nullptr, nullptr, nullptr, nullptr, dy, nullptr, dws, wsb, st));
  CK(hipStreamSynchronize(st));
  PK(p2l_lab_set(0, nullptr));
  std::vector<unsigned long long> t(8 * 64 * 8);
  CK(hipMemcpy(t.data(), dtr, t.size() * 8, hipMemcpyDeviceToHost));
  const int nch = Cin / 16;
  printf("trace (s_memtime ticks = 100 MHz? see ratio): per chunk, waves 0,1,4,5:  step0 step1 step2 step3 barrier | period\n");
  for (int c : {1, 2, 7, 12}) for (int w : {0, 1, 4, 5}) {
    const unsigned long long* a = &t[(w * 64 + c) * 8];
    const unsigned long long nx0 = t[(w * 64 + c + 1) * 8];
    printf("c%2d w%d  %6lld %6lld %6lld %6lld %6lld | %6lld\n", c, w, (long long)(a[1] - a[0]), (long long)(a[2] - a[1]),
           (long long)(a[3] - a[2]), (long long)(a[4] - a[3]), (long long)(a[5] - a[4]), (long long)(nx0 - a[0]));
  }
  double per = 0; int n = 0;
  for (int w = 0; w < 8; ++w) for (int c = 1; c + 2 < nch; ++c) { per += (double)(t[(w * 64 + c + 1) * 8] - t[(w * 64 + c) * 8]); ++n; }
  printf("mean chunk period %.0f ticks\n", per / n);
  for (int w : {0, 4}) {
    const unsigned long long* a = &t[(w * 64 + 63) * 8];
    printf("block phases w%d (ticks): prologue %lld | %d chunks %lld | epilogue pass 0 %lld | pass 1 %lld\n", w,
           (long long)(a[1] - a[0]), nch, (long long)(a[2] - a[1]), (long long)(a[3] - a[2]), (long long)(a[4] - a[3]));
    const unsigned long long* e = &t[(w * 64 + 62) * 8];
    printf("   pass 0: dump writes %lld | barrier %lld | reads + output transform %lld | epilogue item + stores %lld | to the end of the pass %lld\n",
           (long long)(e[1] - e[0]), (long long)(e[2] - e[1]), (long long)(e[3] - e[2]), (long long)(e[4] - e[3]), (long long)(a[3] - e[4]));
  }
  }
  return 0;
}
#endif


// 1x1 layers of the bench step: exact-fp32 kernel | bf16x3 kernel
static int pw_main(int B, int pro) {
  struct { int H, Cin, Cout; } layers[] = {{64, 256, 512}, {64, 512, 256}, {128, 128, 256}, {128, 64, 256},
                                          {256, 64, 128}, {256, 128, 64}, {64, 128, 512}, {64, 64, 512},
                                          {32, 256, 1024}, {32, 1024, 256}, {64, 512, 128}, {128, 256, 64}};
  struct { const char* name; int wfmt; int form; } forms[] = {
      {"fp32", P2L_WFMT_F32, 0}, {"pw-lds", P2L_WFMT_PW, 0}};
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto L : layers) {
    const int H = L.H, W = L.H, Cin = L.Cin, Cout = L.Cout;
    const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * H * W * Cout, nw = (size_t)Cout * Cin;
    std::vector<float> hx(nx), hw(nw), hs((size_t)B * Cin), ht((size_t)B * Cin);
    for (auto& v : hx) { v = nrand(); if (pro == P2L_PRO_NONE && v < 0.f) v = 0.f; }
    for (auto& v : hw) v = nrand() / sqrtf((float)Cin);
    for (auto& v : hs) v = 0.5f + urand();
    for (auto& v : ht) v = 0.3f * nrand();
    float *dx, *dwo, *dy, *dsv, *dtv;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dwo, nw * 4)); CK(hipMalloc(&dy, ny * 4));
    CK(hipMalloc(&dsv, hs.size() * 4)); CK(hipMalloc(&dtv, ht.size() * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwo, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsv, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtv, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> ref(ny), out(ny);
    for (int pass = 0; pass < 2; ++pass)
    for (auto& f : forms) {
      float* dwp;
      if (f.wfmt == P2L_WFMT_PW) {
        CK(hipMalloc(&dwp, ((size_t)Cout * Cin * 5 / 2 + 64) * 4));
        PK(p2l_pack_conv_weight_pw(dwo, Cout, Cin, Cout, Cin, 0, dwp, st));
      } else {
        CK(hipMalloc(&dwp, (size_t)Cout * Cin * 4));
        PK(p2l_pack_conv_weight(dwo, Cout, Cin, 1, Cout, Cin, 0, dwp, st));
      }
      P2LConv d; memset(&d, 0, sizeof d);
      d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.taps = 1; d.x_ld = Cin; d.pro = pro;
      d.pro_bstride = Cin; d.alpha = 1.f; d.y_ld = Cout; d.n_store = Cout; d.splitk = 1; d.wfmt = f.wfmt;
      for (int i = 0; i < (pass == 0 && &f == &forms[0] ? 300 : 5); ++i)
        PK(p2l_conv_fwd(&d, dx, dwp, nullptr, dsv, dtv, nullptr, nullptr, dy, nullptr, nullptr, 0, st));
      CK(hipStreamSynchronize(st));
      const int reps = 30;
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i)
        PK(p2l_conv_fwd(&d, dx, dwp, nullptr, dsv, dtv, nullptr, nullptr, dy, nullptr, nullptr, 0, st));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      CK(hipMemcpy(out.data(), dy, ny * 4, hipMemcpyDeviceToHost));
      if (&f == &forms[0] && pass == 0) ref = out;
      double md = 0, mx = 0; size_t nbad = 0;
      for (size_t i = 0; i < ny; ++i) {
        if (!(out[i] == out[i])) { ++nbad; continue; }
        md = fmax(md, fabs((double)out[i] - ref[i])); mx = fmax(mx, fabs((double)ref[i]));
      }
      const double fl = 2.0 * B * H * W * (double)Cin * Cout, by = 4.0 * (nx + ny);
      if (pass == 1)
        printf("%2dx%3d^2 %4d->%4d pro%d %-9s %.4f ms %6.1f TFLOP/s %5.2f TB/s  max|d|/max|ref| %.2e nan %zu\n", B, H, Cin,
               Cout, pro, f.name, ms, fl / ms / 1e9, by / ms / 1e9, md / (mx + 1e-30), nbad);
      fflush(stdout);
      CK(hipFree(dwp));
    }
    CK(hipFree(dx)); CK(hipFree(dwo)); CK(hipFree(dy)); CK(hipFree(dsv)); CK(hipFree(dtv));
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 3 && atoi(argv[3]) == 1) return pw_main(atoi(argv[1]), atoi(argv[2]));
#ifdef P2L_LAB
  return lab_main(argc > 1 ? atoi(argv[1]) : 18);
#endif
  const int B = argc > 1 ? atoi(argv[1]) : 18;
  const int pro = argc > 2 ? atoi(argv[2]) : P2L_PRO_NONE;
  struct { int H, Cin, Cout; } layers[] = {{64, 256, 256}, {32, 512, 512}, {128, 128, 128}, {256, 64, 64},
                                          {32, 256, 256}, {16, 512, 512}};
  const Form forms[] = {{"direct", P2L_WFMT_BF16X3, 0}, {"wino8", P2L_WFMT_BF16X3W, P2L_FORM_WINO_ANY | P2L_FORM_WINO_8X16},
                        {"wino16", P2L_WFMT_BF16X3W, P2L_FORM_WINO_ANY | P2L_FORM_WINO_BF3},
                        {"wino16 f16x2", P2L_WFMT_BF16X3W, P2L_FORM_WINO_ANY}};
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto L : layers) {
    const int H = L.H, W = L.H, Cin = L.Cin, Cout = L.Cout;
    const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * H * W * Cout, nw = (size_t)Cout * Cin * 9;
    std::vector<float> hx(nx), hw(nw), hs((size_t)B * Cin), ht((size_t)B * Cin);
    // ReLU-like input statistics (half the values zero) when there is no prologue
    for (auto& v : hx) { v = nrand(); if (pro == P2L_PRO_NONE && v < 0.f) v = 0.f; }
    const float ws = 1.f / sqrtf(9.f * Cin);
    for (auto& v : hw) v = nrand() * ws;
    for (auto& v : hs) v = 0.5f + urand();
    for (auto& v : ht) v = 0.3f * nrand();
    float *dx, *dwo, *dy, *dref, *dsv, *dtv;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dwo, nw * 4)); CK(hipMalloc(&dy, ny * 4)); CK(hipMalloc(&dref, ny * 4));
    CK(hipMalloc(&dsv, hs.size() * 4)); CK(hipMalloc(&dtv, ht.size() * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwo, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsv, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtv, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> ref(ny), out(ny);
    for (int pass = 0; pass < 2; ++pass)
    for (const Form& f : forms) {
      const size_t wf = p2l_packed_weight_floats(9, Cout, Cin, f.wfmt);
      float* dwp; CK(hipMalloc(&dwp, wf * 4));
      if (f.wfmt == P2L_WFMT_BF16X3W) PK(p2l_pack_conv_weight_bf3w(dwo, Cout, Cin, 9, Cout, Cin, 0, dwp, st));
      else PK(p2l_pack_conv_weight_bf3(dwo, Cout, Cin, 9, Cout, Cin, 0, dwp, st));
      P2LConv d; memset(&d, 0, sizeof d);
      d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.taps = 9; d.x_ld = Cin; d.pro = pro;
      d.pro_bstride = Cin; d.alpha = 1.f; d.act = P2L_ACT_NONE; d.pool = P2L_POOL_NONE; d.y_ld = Cout;
      d.n_store = Cout; d.splitk = 1; d.wfmt = f.wfmt; d.form = f.form;
      const size_t wsb = p2l_conv_workspace_bytes(&d);
      void* dws = nullptr; if (wsb) CK(hipMalloc(&dws, wsb));
      CK(hipMemsetAsync(dy, 0xff, ny * 4, st));
      for (int i = 0; i < (pass == 0 && &f == &forms[0] ? 300 : 5); ++i)   // (idle clocks settle first)
        PK(p2l_conv_fwd(&d, dx, dwp, nullptr, dsv, dtv, nullptr, nullptr, dy, nullptr, dws, wsb, st));
      CK(hipStreamSynchronize(st));
      const int reps = 30;
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i)
        PK(p2l_conv_fwd(&d, dx, dwp, nullptr, dsv, dtv, nullptr, nullptr, dy, nullptr, dws, wsb, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      CK(hipMemcpy(out.data(), dy, ny * 4, hipMemcpyDeviceToHost));
      double md = 0, mx = 0; size_t nbad = 0;
      if (&f == &forms[0] && pass == 0) ref = out;
      for (size_t i = 0; i < ny; ++i) {
        if (!(out[i] == out[i])) { ++nbad; continue; }
        md = fmax(md, fabs((double)out[i] - ref[i])); mx = fmax(mx, fabs((double)ref[i]));
      }
      const double fl = 2.0 * B * H * W * (double)Cin * Cout * 9;
      printf("p%d %2dx%3d^2 %3d->%3d pro%d %-14s %.4f ms %6.1f TFLOP/s  max|d|/max|ref| %.2e  nan %zu\n", pass, B, H, Cin, Cout, pro,
             f.name, ms, fl / ms / 1e9, md / (mx + 1e-30), nbad);
      fflush(stdout);
      CK(hipFree(dwp)); if (dws) CK(hipFree(dws));
    }
    CK(hipFree(dx)); CK(hipFree(dwo)); CK(hipFree(dy)); CK(hipFree(dref)); CK(hipFree(dsv)); CK(hipFree(dtv));
  }
  return 0;
}
