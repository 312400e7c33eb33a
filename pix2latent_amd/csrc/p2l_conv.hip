// 3x3 / 1x1 convolution as an implicit GEMM on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Replaces nn.Conv2d forward and input-gradient on the pix2latent hot path
// (BigGAN-deep GenBlock convs, SelfAttn 1x1s, VGG16 features; reached from
// pix2latent/model/biggan.py:58 and pix2latent/loss_functions.py:142 in the
// reference).  Numerics: exact fp32 FMA chains (the f32-input MFMA is bitwise a
// k-ordered fmaf chain), so results differ from the CPU oracle only by
// summation order.
//
// Design (MI355X-first, not a translation of a warp-32 tiling):
//   * NHWC activations; GEMM M = output pixels, N = Cout, K = taps*Cin.
//   * Block tile 128 pixels x BN(64|32) channels, 4 waves; each wave owns a
//     32-pixel x BN strip = BN/32 accumulators of 32x32 (16 VGPR each).
//   * The 128 pixels are a TB x TH x TW spatial patch enumerated in 2x2-quad
//     order: M index i -> quad i>>2, sub-pixel i&3.  In the MFMA C layout
//     (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) the four pixels of a quad then
//     sit in 4 consecutive registers of ONE lane, so 2x2 max/sum pooling in the
//     epilogue is register-local (VGG max-pool, attention max-pool, and the
//     nearest-x2-upsample backward are all fused that way).
//   * Per K-chunk (KC = 16 channels for 3x3, 32 for 1x1) the input patch + halo
//     is staged ONCE in LDS and reused by all 9 taps; weights for all taps of
//     the chunk are staged next to it.  LDS rows are KC+4 floats so that the
//     16-byte fragment reads (ds_read_b128, one per 4 MFMAs) are conflict-free.
//   * K order inside a chunk is permuted identically for A and B (lane half
//     picks k 0-3 vs 4-7 of each 8), which lets a single b128 feed 4 MFMAs.
//   * The CBN/BN affine + ReLU of the producer and the nearest x2 upsample are
//     applied while staging (prologue), bias/residual/activation/mask/pool in
//     the epilogue: no standalone elementwise pass over the activations.
//   * Global->LDS goes through registers (prologue math + padded rows), loads
//     for chunk c+1 are issued before the MFMAs of chunk c; two blocks per CU
//     (<= 69 KB LDS each) overlap one block's barrier with the other's MFMAs.
//   * Small-M layers (4x4..16x16) split K over blockIdx.y and finish with a
//     deterministic reduce+epilogue kernel (no float atomics: CMA ranking must
//     be reproducible).
#include "p2l_common.h"

#include <vector>

namespace {

// Optional per-launch timing of the conv kernel (bench.py's roofline leg):
// hipEvents from a pre-created pool are recorded on the launch stream around
// every conv (incl. its split-K finish); nothing is allocated while enabled.
struct ConvProf {
  bool on = false;
  int n = 0;
  std::vector<hipEvent_t> ev;
  std::vector<double> flops;
  std::vector<int> kind;
} g_prof;

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

template <int TAPS, int BN, int KC, int A_ITERS, int PRO, bool UPS>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvK k) {
  constexpr int PITCH = KC + 4;      // floats per LDS row
  constexpr int VPR = KC / 4;        // float4 per row
  constexpr int NT = BN / 32;        // accumulators per wave
  constexpr int B_ITEMS = TAPS * BN * VPR;
  constexpr int B_ITERS = (B_ITEMS + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int TW = 1 << k.tw_log, TH = 1 << k.th_log, TB = 1 << k.tb_log;
  const int HW_ = TW + 2, HH_ = TH + 2;
  const int a_rows = (TAPS == 9) ? TB * HH_ * HW_ : 128;
  float* As = smem;
  float* Bs = smem + a_rows * PITCH;

  // ---- which tile -------------------------------------------------------
  const int swz = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = swz / k.n_ntiles, nt = swz - mt * k.n_ntiles;
  const int tx = mt & ((1 << k.tiles_x_log) - 1);
  const int ty = (mt >> k.tiles_x_log) & ((1 << k.tiles_y_log) - 1);
  const int bt = mt >> (k.tiles_x_log + k.tiles_y_log);
  const int n0 = nt * BN;
  const int y0 = ty << k.th_log, x0 = tx << k.tw_log, b0 = bt << k.tb_log;

  const int z = blockIdx.y;
  const int c_begin = z * k.chunks_per_split;
  const int c_end = min(k.nchunks, c_begin + k.chunks_per_split);

  // ---- per-thread staging descriptors (fixed across chunks) --------------
  // Loads are issued unconditionally from a clamped (always valid) address and
  // zeroed at LDS-write time: a divergent "load or zero" makes hipcc branch
  // around every load and drain vmcnt per element.
  int a_goff[A_ITERS];   // float offset of the source pixel's chunk-0 vector
  int a_soff[A_ITERS];   // float offset into pro_s / pro_t
  int a_loff[A_ITERS];   // LDS float offset, -1 = no item
  unsigned a_valid = 0;  // bit it: source pixel exists (else zero padding)
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int j = tid + 256 * it;
    const int p = j / VPR, v = j - p * VPR;
    a_goff[it] = 0;
    a_soff[it] = 0;
    a_loff[it] = (p < a_rows) ? p * PITCH + v * 4 : -1;
    if (p < a_rows) {
      int tb, iy, ix;
      if (TAPS == 9) {
        tb = p / (HH_ * HW_);
        const int rem = p - tb * (HH_ * HW_);
        const int hy = rem / HW_, hx = rem - hy * HW_;
        iy = y0 + hy - 1;
        ix = x0 + hx - 1;
      } else {
        const int Q = p >> 2, s = p & 3;
        const int qx = Q & ((TW >> 1) - 1);
        const int qy = (Q >> (k.tw_log - 1)) & ((TH >> 1) - 1);
        tb = Q >> (k.tw_log + k.th_log - 2);
        iy = y0 + 2 * qy + (s >> 1);
        ix = x0 + 2 * qx + (s & 1);
      }
      const int b = b0 + tb;
      if (b < k.B && iy >= 0 && iy < k.H && ix >= 0 && ix < k.W) {
        int pix;
        if (UPS)
          pix = (b * (k.H >> 1) + (iy >> 1)) * (k.W >> 1) + (ix >> 1);
        else
          pix = (b * k.H + iy) * k.W + ix;
        a_goff[it] = pix * k.x_ld + v * 4;
        a_soff[it] = b * k.pro_bstride + v * 4;
        a_valid |= 1u << it;
      }
    }
  }

  f32x4 xr[A_ITERS], sr[A_ITERS], tr[A_ITERS], wr[B_ITERS];

  auto load_regs = [&](int c) {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      xr[it] = *reinterpret_cast<const f32x4*>(k.x + (size_t)a_goff[it] + c * KC);
      if (PRO != P2L_PRO_NONE) {
        sr[it] = *reinterpret_cast<const f32x4*>(k.pro_s + a_soff[it] + c * KC);
        tr[it] = *reinterpret_cast<const f32x4*>(k.pro_t + a_soff[it] + c * KC);
      }
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int j = tid + 256 * it;
      if (j < B_ITEMS) {
        const int tap = j / (BN * VPR);
        const int rem = j - tap * (BN * VPR);  // row*VPR + v
        const size_t off =
            (((size_t)tap * k.nchunks + c) * k.Cout + n0) * KC + rem * 4;
        wr[it] = *reinterpret_cast<const f32x4*>(k.w + off);
      }
    }
  };

  auto write_lds = [&]() {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      if (a_loff[it] >= 0) {
        f32x4 v = xr[it];
        if (PRO != P2L_PRO_NONE) {
          v = v * sr[it] + tr[it];
          if (PRO == P2L_PRO_AFFINE_RELU) {
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
          }
        }
        if (!((a_valid >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(As + a_loff[it]) = v;
      }
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int j = tid + 256 * it;
      if (j < B_ITEMS) {
        const int row = j / VPR, v = j - row * VPR;  // row = tap*BN + n
        *reinterpret_cast<f32x4*>(Bs + row * PITCH + v * 4) = wr[it];
      }
    }
  };

  // ---- fragment addressing ----------------------------------------------
  int a_row0;
  {
    const int i = wave * 32 + l31;
    if (TAPS == 9) {
      const int Q = i >> 2, s = i & 3;
      const int qx = Q & ((TW >> 1) - 1);
      const int qy = (Q >> (k.tw_log - 1)) & ((TH >> 1) - 1);
      const int tb = Q >> (k.tw_log + k.th_log - 2);
      a_row0 = (tb * HH_ + 2 * qy + (s >> 1)) * HW_ + 2 * qx + (s & 1);
    } else {
      a_row0 = i;
    }
  }
  const float* a_frag = As + a_row0 * PITCH + lhi * 4;
  const float* b_frag = Bs + l31 * PITCH + lhi * 4;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  if (c_begin < c_end) {
    load_regs(c_begin);
    write_lds();
  }
  __syncthreads();

  for (int c = c_begin; c < c_end; ++c) {
    const bool more = (c + 1 < c_end);
    if (more) load_regs(c + 1);

#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
      const float* ap = a_frag + ((TAPS == 9) ? (dy * HW_ + dx) * PITCH : 0);
#pragma unroll
      for (int kk = 0; kk < KC / 8; ++kk) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ap + kk * 8);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4 bq = *reinterpret_cast<const f32x4*>(
              b_frag + (tap * BN + j * 32) * PITCH + kk * 8);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq.z, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq.w, acc[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) write_lds();
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------
  // lane owns column n of 4 quads (g): Q = wave*8 + 2g + lhi, 4 sub-pixels each
  int q_pix0[4], q_b[4], q_oy[4], q_ox[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int Q = wave * 8 + 2 * g + lhi;
    const int qx = Q & ((TW >> 1) - 1);
    const int qy = (Q >> (k.tw_log - 1)) & ((TH >> 1) - 1);
    const int tb = Q >> (k.tw_log + k.th_log - 2);
    q_b[g] = b0 + tb;
    q_oy[g] = y0 + 2 * qy;
    q_ox[g] = x0 + 2 * qx;
    q_pix0[g] = (q_b[g] * k.H + q_oy[g]) * k.W + q_ox[g];
  }
  const bool simple = (k.splitk == 1) && !k.res && !k.mask && !k.pool && k.y;
  if (k.splitk > 1) {
    const size_t mtot = (size_t)k.B * k.H * k.W;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + j * 32 + l31;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (q_b[g] >= k.B) continue;
        float* wp = k.ws + ((size_t)z * mtot + (size_t)q_pix0[g]) * k.Cout + n;
        wp[0] = acc[j][g * 4 + 0];
        wp[k.Cout] = acc[j][g * 4 + 1];
        wp[(size_t)k.W * k.Cout] = acc[j][g * 4 + 2];
        wp[(size_t)(k.W + 1) * k.Cout] = acc[j][g * 4 + 3];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + j * 32 + l31;
      if (n >= k.n_store) continue;
      const float bias_n = k.bias ? k.bias[n] : 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (q_b[g] >= k.B) continue;
        const float a[4] = {acc[j][g * 4 + 0], acc[j][g * 4 + 1], acc[j][g * 4 + 2],
                            acc[j][g * 4 + 3]};
        if (simple)
          epilogue_quad<true>(k, a, q_pix0[g], q_b[g], q_oy[g], q_ox[g], n, bias_n);
        else
          epilogue_quad<false>(k, a, q_pix0[g], q_b[g], q_oy[g], q_ox[g], n, bias_n);
      }
    }
  }
}

// Deterministic split-K finish: one thread per (quad, channel).
__global__ __launch_bounds__(256) void conv_splitk_finish(const ConvK k) {
  const int Hh = k.H >> 1, Wh = k.W >> 1;
  const size_t total = (size_t)k.B * Hh * Wh * k.n_store;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx % k.n_store);
  size_t q = idx / k.n_store;
  const int qx = (int)(q % Wh);
  q /= Wh;
  const int qy = (int)(q % Hh);
  const int b = (int)(q / Hh);
  const size_t mtot = (size_t)k.B * k.H * k.W;
  float a[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const size_t pix = ((size_t)b * k.H + 2 * qy + (s >> 1)) * k.W + 2 * qx + (s & 1);
    float acc = 0.f;
    for (int zz = 0; zz < k.splitk; ++zz)
      acc += k.ws[((size_t)zz * mtot + pix) * k.Cout + n];
    a[s] = acc;
  }
  epilogue_quad<false>(k, a, (b * k.H + 2 * qy) * k.W + 2 * qx, b, 2 * qy, 2 * qx, n,
                       k.bias ? k.bias[n] : 0.f);
}

// src is OIHW [O][I][taps].  flip=0 packs the conv I->O (K=I, N=O); flip=1 packs
// its input-gradient conv O->I (K=O, N=I, taps mirrored).
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(
    const float* __restrict__ src, float* __restrict__ dst, int O, int I,
    int taps, int N_pad, int K_pad, int kc, int flip) {
  const size_t total = (size_t)taps * K_pad * N_pad;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int kk = (int)(idx % kc);
  size_t r = idx / kc;
  const int n = (int)(r % N_pad);
  r /= N_pad;
  const int nch = K_pad / kc;
  const int q = (int)(r % nch);
  const int tap = (int)(r / nch);
  const int c = q * kc + kk;
  float v = 0.f;
  if (!flip) {
    if (n < O && c < I) v = src[((size_t)n * I + c) * taps + tap];
  } else {
    if (n < I && c < O) v = src[((size_t)c * I + n) * taps + (taps - 1 - tap)];
  }
  dst[idx] = v;
}

int choose_tile(const P2LConv* d, ConvK& k) {
  if (!is_pow2(d->H) || !is_pow2(d->W) || d->H < 4 || d->W < 4) return P2L_EINVAL;
  int TW = d->W < 16 ? d->W : 16;
  int TH = 128 / TW;
  if (TH > d->H) TH = d->H;
  int TB = 128 / (TW * TH);
  k.tw_log = ilog2(TW);
  k.th_log = ilog2(TH);
  k.tb_log = ilog2(TB);
  k.tiles_x_log = ilog2(d->W / TW);
  k.tiles_y_log = ilog2(d->H / TH);
  k.n_mtiles = (d->W / TW) * (d->H / TH) * cdiv(d->B, TB);
  return P2L_OK;
}

// Output-channel tile: 64 unless the grid then leaves CUs idle in its last
// round.  All blocks of a launch do the same MFMA work and co-resident blocks
// share a CU's matrix pipes, so time ~ ceil(blocks / 256 CUs) * work-per-block.
int choose_bn(const P2LConv* d, int n_mtiles) {
  if (d->Cout % 64) return 32;
  const int n64 = n_mtiles * (d->Cout / 64), n32 = n_mtiles * (d->Cout / 32);
  if (n64 < 256) return 64;               // split-K regime: keep the fatter tile
  const double t64 = (double)cdiv(n64, 256) * 64.0;
  const double t32 = (double)cdiv(n32, 256) * 32.0 * 1.06;   // thinner tile: less reuse
  return (t32 < 0.93 * t64) ? 32 : 64;
}

template <int TAPS, int BN, int KC, int A_ITERS>
int launch_conv(const ConvK& k, int pro, int ups, size_t lds, hipStream_t st) {
  dim3 grid(k.n_mtiles * k.n_ntiles, k.splitk), block(256);
#define P2L_LAUNCH(PRO, UPS)                                                     \
  do {                                                                           \
    auto kfn = conv_mfma_kernel<TAPS, BN, KC, A_ITERS, PRO, UPS>;                \
    static bool attr_set = false;                                                \
    if (!attr_set) {                                                             \
      (void)hipFuncSetAttribute((const void*)kfn,                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                160 * 1024);                                     \
      attr_set = true;                                                           \
    }                                                                            \
    hipLaunchKernelGGL(kfn, grid, block, lds, st, k);                            \
  } while (0)
  if (ups) {
    if (pro == P2L_PRO_NONE) P2L_LAUNCH(P2L_PRO_NONE, true);
    else if (pro == P2L_PRO_AFFINE_RELU) P2L_LAUNCH(P2L_PRO_AFFINE_RELU, true);
    else P2L_LAUNCH(P2L_PRO_AFFINE, true);
  } else {
    if (pro == P2L_PRO_NONE) P2L_LAUNCH(P2L_PRO_NONE, false);
    else if (pro == P2L_PRO_AFFINE_RELU) P2L_LAUNCH(P2L_PRO_AFFINE_RELU, false);
    else P2L_LAUNCH(P2L_PRO_AFFINE, false);
  }
#undef P2L_LAUNCH
  return p2l_check_launch();
}

}  // namespace

extern "C" int p2l_conv_suggest_splitk(const P2LConv* d) {
  ConvK k{};
  if (choose_tile(d, k) != P2L_OK) return 1;
  const int bn = choose_bn(d, k.n_mtiles);
  const int kc = (d->taps == 9) ? 16 : 32;
  const int nblk = k.n_mtiles * (d->Cout / bn);
  const int nchunks = d->Cin / kc;
  if (nblk >= 192) return 1;
  int s = cdiv(512, nblk);
  // keep >= 2 chunks of work per split for 3x3 (18 tap-chunks), >= 4 for 1x1
  const int min_chunks = (d->taps == 9) ? 2 : 4;
  if (s > nchunks / min_chunks) s = nchunks / min_chunks;
  if (s > 32) s = 32;
  if (s < 1) s = 1;
  return s;
}

extern "C" size_t p2l_conv_workspace_bytes(const P2LConv* d) {
  if (d->splitk <= 1) return 0;
  return (size_t)d->splitk * d->B * d->H * d->W * d->Cout * sizeof(float);
}

extern "C" int p2l_conv_fwd(const P2LConv* d, const float* x, const float* w,
                            const float* bias, const float* pro_s,
                            const float* pro_t, const float* res,
                            const float* mask, float* y, float* yp,
                            void* workspace, size_t ws_bytes, void* stream) {
  if (!d || !x || !w) return P2L_EINVAL;
  if (d->taps != 1 && d->taps != 9) return P2L_EINVAL;
  const int kc = (d->taps == 9) ? 16 : 32;
  if (d->Cin % kc || d->Cout % 32 || d->B < 1) return P2L_EINVAL;
  if (d->x_ld % 4 || d->x_ld < d->Cin) return P2L_EINVAL;
  if (d->pro != P2L_PRO_NONE && (!pro_s || !pro_t || d->pro_bstride % 4)) return P2L_EINVAL;
  if (d->pool != P2L_POOL_NONE && !yp) return P2L_EINVAL;
  if (!y && !yp) return P2L_EINVAL;
  if (d->ups && d->taps != 9) return P2L_EUNSUP;
  if (d->n_store < 1 || d->n_store > d->Cout) return P2L_EINVAL;
  {
    // element offsets are computed in 32 bits inside the kernel
    const int64_t px = (int64_t)d->B * d->H * d->W;
    int64_t ldmax = d->x_ld;
    if (d->y_ld > ldmax) ldmax = d->y_ld;
    if (d->res_ld > ldmax) ldmax = d->res_ld;
    if (d->mask_ld > ldmax) ldmax = d->mask_ld;
    if (px * ldmax >= ((int64_t)1 << 31)) return P2L_EUNSUP;
  }

  ConvK k{};
  k.x = x; k.w = w; k.bias = bias; k.pro_s = pro_s; k.pro_t = pro_t;
  k.res = res; k.mask = mask; k.y = y; k.yp = yp; k.ws = (float*)workspace;
  k.B = d->B; k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.Cout = d->Cout;
  k.x_ld = d->x_ld; k.y_ld = d->y_ld; k.yp_ld = d->yp_ld; k.res_ld = d->res_ld;
  k.mask_ld = d->mask_ld; k.n_store = d->n_store; k.pro_bstride = d->pro_bstride;
  k.alpha = d->alpha; k.act = d->act; k.pool = d->pool; k.res_ups = d->res_ups;
  k.ups = d->ups;
  int rc = choose_tile(d, k);
  if (rc) return rc;
  const int bn = choose_bn(d, k.n_mtiles);
  k.n_ntiles = d->Cout / bn;
  k.nchunks = d->Cin / kc;
  k.splitk = d->splitk < 1 ? 1 : d->splitk;
  if (k.splitk > k.nchunks) k.splitk = k.nchunks;
  k.chunks_per_split = cdiv(k.nchunks, k.splitk);
  k.splitk = cdiv(k.nchunks, k.chunks_per_split);
  if (k.splitk > 1) {
    const size_t need = (size_t)k.splitk * d->B * d->H * d->W * d->Cout * sizeof(float);
    if (!workspace || ws_bytes < need) return P2L_EWS;
  }
  hipStream_t st = (hipStream_t)stream;
  const int TW = 1 << k.tw_log, TH = 1 << k.th_log, TB = 1 << k.tb_log;
  const int a_rows = (d->taps == 9) ? TB * (TH + 2) * (TW + 2) : 128;
  const size_t lds = (size_t)(a_rows + d->taps * bn) * (kc + 4) * sizeof(float);

  int prof_slot = -1;
  if (g_prof.on && g_prof.n < (int)g_prof.flops.size()) {
    prof_slot = g_prof.n++;
    g_prof.flops[prof_slot] = d->algo_flops > 0.0
        ? d->algo_flops
        : 2.0 * d->B * d->H * d->W * (double)d->Cin * d->Cout * d->taps;
    g_prof.kind[prof_slot] = d->taps == 9 ? 0 : 1;
    (void)hipEventRecord(g_prof.ev[2 * prof_slot], st);
  }
  if (d->taps == 9) {
    const bool small = (a_rows * 4 <= 3 * 256);
    if (bn == 64) rc = small ? launch_conv<9, 64, 16, 3>(k, d->pro, d->ups, lds, st)
                             : launch_conv<9, 64, 16, 5>(k, d->pro, d->ups, lds, st);
    else          rc = small ? launch_conv<9, 32, 16, 3>(k, d->pro, d->ups, lds, st)
                             : launch_conv<9, 32, 16, 5>(k, d->pro, d->ups, lds, st);
  } else {
    if (bn == 64) rc = launch_conv<1, 64, 32, 4>(k, d->pro, 0, lds, st);
    else          rc = launch_conv<1, 32, 32, 4>(k, d->pro, 0, lds, st);
  }
  if (rc) return rc;
  if (k.splitk > 1) {
    const size_t total = (size_t)k.B * (k.H >> 1) * (k.W >> 1) * k.n_store;
    hipLaunchKernelGGL(conv_splitk_finish, dim3(cdiv(total, 256)), dim3(256), 0,
                       st, k);
    rc = p2l_check_launch();
  }
  if (prof_slot >= 0) (void)hipEventRecord(g_prof.ev[2 * prof_slot + 1], st);
  return rc;
}

extern "C" int p2l_prof_begin(int max_launches) {
  if (max_launches < 1) return P2L_EINVAL;
  while ((int)g_prof.ev.size() < 2 * max_launches) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return P2L_ELAUNCH;
    g_prof.ev.push_back(e);
  }
  g_prof.flops.assign(max_launches, 0.0);
  g_prof.kind.assign(max_launches, 0);
  g_prof.n = 0;
  g_prof.on = true;
  return P2L_OK;
}

extern "C" int p2l_prof_end(double flops[2], double ms[2], int32_t count[2]) {
  g_prof.on = false;
  flops[0] = flops[1] = ms[0] = ms[1] = 0.0;
  count[0] = count[1] = 0;
  for (int i = 0; i < g_prof.n; ++i) {
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return P2L_ELAUNCH;
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess)
      return P2L_ELAUNCH;
    const int k = g_prof.kind[i];
    flops[k] += g_prof.flops[i];
    ms[k] += t;
    count[k] += 1;
  }
  g_prof.n = 0;
  return P2L_OK;
}

extern "C" int p2l_pack_conv_weight(const float* w_oihw, int O, int I, int taps,
                                    int N_pad, int K_pad, int transpose_flip,
                                    float* w_packed, void* stream) {
  if (!w_oihw || !w_packed || (taps != 1 && taps != 9)) return P2L_EINVAL;
  const int kc = (taps == 9) ? 16 : 32;
  const int N = transpose_flip ? I : O, K = transpose_flip ? O : I;
  if (K_pad % kc || N_pad % 32 || N_pad < N || K_pad < K) return P2L_EINVAL;
  const size_t total = (size_t)taps * K_pad * N_pad;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(cdiv(total, 256)), dim3(256),
                     0, (hipStream_t)stream, w_oihw, w_packed, O, I, taps, N_pad,
                     K_pad, kc, transpose_flip);
  return p2l_check_launch();
}
