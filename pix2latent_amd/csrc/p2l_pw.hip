// 1x1 convolution (pointwise GEMM) in the fp32-equivalent 3-way bf16 split arithmetic.
//
// Same role as conv_mfma_kernel<TAPS=1> (p2l_conv.hip): the BigGAN-deep GenBlock conv_0 /
// conv_3 and the SelfAttn 1x1s, forward and input-gradient (reached from
// pix2latent/model/biggan.py:58 in the reference).  That kernel multiplies on the exact-fp32
// MFMA (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate); this one splits both operands into
// three bf16 pieces and accumulates the six significant cross products in fp32
// (include/p2l.h P2L_WFMT_BF16X3): 6 x 32 matrix-pipe cycles per 16 channels instead of
// 8 x 64.
//
// Why this works for ONE tap where the 3x3 kernel's per-chunk structure did not (round 1
// measured 0.62-0.93x with it): the split costs ~40 VALU cycles per activation value, and a
// 16-channel chunk of a 1x1 conv has only 12 MFMAs per wave to hide it behind plus two
// barriers.  Here a stage is 64 channels: every thread splits 32 values (8 float4) while the
// previous stage's 48 MFMAs per wave run, one barrier pair per 64 channels, two blocks per CU
// (72 KB of LDS each) overlap one block's split with the other's multiply.
//
//   * block = the 128-pixel quad-ordered tile of the direct kernel (same epilogue: residual,
//     nearest-x2 shortcut, pooling, fused activation backward) x 64 output channels;
//     wave w owns pixels 32w .. 32w+31: 2 accumulators of 32x32;
//   * A (activations): global fp32 -> registers -> prologue affine / ReLU -> 3-way split ->
//     LDS rows of 96 B ([x1 k0-7 | x1 k8-15 | x2 .. | x3 ..], chunk index XOR bit 3 of the row);
//   * B (weights): pre-split at pack time into the same row format (p2l_pack_conv_weight_pw),
//     copied global -> registers -> LDS as an image.
#include "p2l_conv_k.h"

#include <cstdlib>

using namespace p2lconv;

namespace {

// KS = channels per stage: 64 (72 KB of LDS, two blocks per CU) or 32 (36 KB: three -- a block
// is one load -> split -> multiply -> store latency chain, and layers with 256 input channels
// have only 4 / 8 stages to amortise it over: more chains in flight beat fewer barriers)
template <int KS> struct PwCfg {
  static constexpr int SUB = KS / 16;                   // 16-channel sub-chunks per stage
  static constexpr int A_FLOATS = SUB * 128 * 24;       // [sub][128 rows][96 B]
  static constexpr int B_FLOATS = SUB * 64 * 24;        // [sub][2 N-tiles x 32 rows][96 B]
  static constexpr size_t LDS_BYTES = (size_t)(A_FLOATS + B_FLOATS) * sizeof(float);
};

__device__ __forceinline__ void pw_store_split(float* As, int row, int q4, const f32x4 x) {
  bf16x4 ph, pm, pl;
  split3(x, ph, pm, pl);
  char* rb = reinterpret_cast<char*>(As) + row * 96 + (q4 & 1) * 8;
  char* rq = rb + bf3_chunk(q4 >> 1, row) * 16;         // pieces at +0 / +32 / +64 bytes
  *reinterpret_cast<bf16x4*>(rq) = ph;
  *reinterpret_cast<bf16x4*>(rq + 32) = pm;
  *reinterpret_cast<bf16x4*>(rq + 64) = pl;
}

template <int PRO, int KS>
__global__ __launch_bounds__(256, KS == 64 ? 2 : 4) void pw_bf3_kernel(const ConvK k) {
  constexpr int PW_KS = KS, PW_SUB = PwCfg<KS>::SUB, PW_A_FLOATS = PwCfg<KS>::A_FLOATS;
  constexpr int VPP = KS / 4;                            // float4 per pixel and stage
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + PW_A_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int TW = 1 << k.tw_log, TH = 1 << k.th_log;
  const int swz = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = swz / k.n_ntiles, nt = swz - mt * k.n_ntiles;
  const int tiles_per_image = k.tiles_x * k.tiles_y;
  const int b0 = mt / tiles_per_image;                   // one image per tile (launcher)
  const int tile_in_image = mt - b0 * tiles_per_image;
  const int ty = tile_in_image / k.tiles_x, tx = tile_in_image - ty * k.tiles_x;
  const int n0 = nt * 64;
  const int y0 = ty << k.th_log, x0 = tx << k.tw_log;

  // ---- A staging: 128 pixels x 16 float4 per stage = 8 items per thread; item j = tid + 256*it
  // covers pixel j >> 4, float4 (tid & 15) of the stage: sub-chunk (tid & 15) >> 2, quarter & 3
  constexpr int A_ITERS = 128 * VPP / 256;
  const int av = tid & (VPP - 1);
  int a_goff[A_ITERS];
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int p = (tid + 256 * it) / VPP;
    const int Q = p >> 2, s = p & 3;
    const int qx = Q & ((TW >> 1) - 1);
    const int qy = (Q >> (k.tw_log - 1)) & ((TH >> 1) - 1);
    const int iy = y0 + 2 * qy + (s >> 1), ix = x0 + 2 * qx + (s & 1);
    a_goff[it] = ((b0 * k.H + iy) * k.W + ix) * k.x_ld + av * 4;
  }
  const int s_off = b0 * k.pro_bstride + av * 4;
  // ---- B staging: [sub][64 rows][6 x 16 B] = 1536 items per stage, 6 per thread; the packed
  // image is [chunk][32-channel tile][32 rows][96 B]: the 2 tiles of this block are one 6 KB run
  constexpr int B_ITERS = PW_SUB * 384 / 256;
  int b_goff[B_ITERS];
#pragma unroll
  for (int it = 0; it < B_ITERS; ++it) {
    const int j = tid + 256 * it;
    const int sub = j / 384, within = j - sub * 384;
    b_goff[it] = (sub * (k.Cout >> 5) + (n0 >> 5)) * 32 * 24 + within * 4;
  }
  const int b_stage = PW_SUB * (k.Cout >> 5) * 32 * 24;  // floats per stage of the image

  f32x4 xr[A_ITERS], wr[B_ITERS], sr, tr;
  auto load_regs = [&](int st) {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it)
      xr[it] = *reinterpret_cast<const f32x4*>(k.x + (size_t)a_goff[it] + st * PW_KS);
    if (PRO != P2L_PRO_NONE) {
      sr = *reinterpret_cast<const f32x4*>(k.pro_s + s_off + st * PW_KS);
      tr = *reinterpret_cast<const f32x4*>(k.pro_t + s_off + st * PW_KS);
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it)
      wr[it] = *reinterpret_cast<const f32x4*>(k.w + (size_t)st * b_stage + b_goff[it]);
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      f32x4 v = xr[it];
      if (PRO != P2L_PRO_NONE) {
        v = v * sr + tr;
        if (PRO == P2L_PRO_AFFINE_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
          v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
      }
      const int p = (tid + 256 * it) / VPP;
      pw_store_split(As + (av >> 2) * (128 * 24), p, av & 3, v);
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it)
      *reinterpret_cast<f32x4*>(Bs + (tid + 256 * it) * 4) = wr[it];     // image copy
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int a_row = wave * 32 + l31;
  const int a_c = bf3_chunk(lhi, a_row) * 4, b_c = bf3_chunk(lhi, l31) * 4;
  const int nstages = k.Cin / KS;
  load_regs(0);
  write_lds();
  __syncthreads();
  for (int st = 0; st < nstages; ++st) {
    const bool more = st + 1 < nstages;
    if (more) load_regs(st + 1);
#pragma unroll
    for (int sub = 0; sub < PW_SUB; ++sub) {
      const float* ar = As + (sub * 128 + a_row) * 24 + a_c;
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ar);
      const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(ar + 8);
      const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(ar + 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float* br = Bs + (sub * 64 + j * 32 + l31) * 24 + b_c;
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(br);
        const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(br + 8);
        const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(br + 16);
        f32x16 t = acc[j];                               // smallest terms first
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, t, 0, 0, 0);
        acc[j] = t;
      }
    }
    __syncthreads();
    if (more) write_lds();
    __syncthreads();
  }
  epilogue_vec<2>(k, acc, smem, wave, lane, b0, y0, x0, n0, tile_in_image, 0, 0, 0);
}


// ---- streaming form for 64 / 128 input channels --------------------------------------------
// The layers that EXPAND channels (BigGAN-deep conv_3 forward, conv_0 input-gradient:
// 64->256 at 128^2, 128->512 at 64^2, 64->128 at 256^2 ...) move 4-9x more output than input
// bytes and are HBM-bound: 75-980 MB per launch against 10 GFLOP.  What matters is HOW the
// output rows (1-2 KB per pixel) reach DRAM.  A first version of this kernel gave a wave 32
// pixels and let it walk over the channels 128 B at a time: every pixel row was visited 8-16
// times, tens of microseconds apart, and the in-step time was 0.5-0.9x the exact-fp32 kernel
// although an isolated (Infinity-Cache resident) benchmark showed 1.3-1.5x.  So:
//
//   * block = 32 pixels (2 image rows x 16), split ONCE into bf16x3 and staged in LDS (12 /
//     24 KB); every wave pulls all A fragments into registers (48 / 96 VGPRs);
//   * the 4 waves take DIFFERENT 32-channel output tiles (wave w: tiles w, w+4, ...): one
//     sweep of the block writes 512 B contiguous per pixel, the whole row within 2-4 sweeps;
//   * weights come straight from the packed image in L2 into B-fragment registers (the image
//     rows ARE fragment rows), 4 sub-chunks ahead of their use; no barrier after the staging;
//   * wave-private LDS transpose -> the shared epilogue item (16 B per lane).
// Forward epilogues only (bias / residual / activation / mask / pooling).
// Same products in the same order as pw_bf3_kernel: bit-identical results.
constexpr int PWS_EP = 36;                                // dump pitch of one 32-column tile

template <int PRO, int KSUB>
__global__ __launch_bounds__(256, 2) void pws_bf3_kernel(const ConvK k) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                       // [KSUB][32 rows][96 B]
  float* dump = smem + KSUB * 768;                        // [4 waves][32][PWS_EP]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int swz = xcd_remap(blockIdx.x, gridDim.x);
  const int tx_n = k.W >> 4, per_image = tx_n * (k.H >> 1);
  const int b0 = swz / per_image, t32 = swz - b0 * per_image;
  const int ty = t32 / tx_n, tx = t32 - ty * tx_n;
  const int y0 = ty * 2, x0 = tx * 16;

  // ---- A: 32 pixels x Cin, quad order (row p = 4 * quad + sub-pixel), split once
  {
    constexpr int V4 = KSUB * 4, A_IT = 32 * V4 / 256;    // float4 per pixel; items per thread
    f32x4 xr[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int j = tid + 256 * it;
      const int p = j / V4, v4 = j - p * V4;
      const int iy = y0 + ((p >> 1) & 1), ix = x0 + 2 * (p >> 2) + (p & 1);
      xr[it] = *reinterpret_cast<const f32x4*>(
          k.x + (size_t)((b0 * k.H + iy) * k.W + ix) * k.x_ld + v4 * 4);
    }
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int j = tid + 256 * it;
      const int p = j / V4, v4 = j - p * V4;
      f32x4 v = xr[it];
      if (PRO != P2L_PRO_NONE) {
        const int so = b0 * k.pro_bstride + v4 * 4;
        v = v * *reinterpret_cast<const f32x4*>(k.pro_s + so) +
            *reinterpret_cast<const f32x4*>(k.pro_t + so);
        if (PRO == P2L_PRO_AFFINE_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
          v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
      }
      pw_store_split(As + (v4 >> 2) * 768, p, v4 & 3, v);
    }
  }
  __syncthreads();
  bf16x8 a[KSUB][3];
#pragma unroll
  for (int sub = 0; sub < KSUB; ++sub) {
    const float* ar = As + (sub * 32 + l31) * 24 + bf3_chunk(lhi, l31) * 4;
    a[sub][0] = *reinterpret_cast<const bf16x8*>(ar);
    a[sub][1] = *reinterpret_cast<const bf16x8*>(ar + 8);
    a[sub][2] = *reinterpret_cast<const bf16x8*>(ar + 16);
  }

  // ---- B fragments from the packed image [sub][Cout/32][32 rows][96 B], ring of 4 sub-chunks
  const int ntiles = k.Cout >> 5;
  const float* wl = k.w + l31 * 24 + bf3_chunk(lhi, l31) * 4;
  bf16x8 bq[4][3];
  auto ldb = [&](int nt, int sub, bf16x8 (&b)[3]) {
    const float* bp = wl + (size_t)(sub * ntiles + nt) * 768;
    b[0] = *reinterpret_cast<const bf16x8*>(bp);
    b[1] = *reinterpret_cast<const bf16x8*>(bp + 8);
    b[2] = *reinterpret_cast<const bf16x8*>(bp + 16);
  };
  float* tb = dump + wave * 32 * PWS_EP;
  const int q = lane >> 3, c4 = lane & 7;                  // epilogue item: quad q, channels 4*c4..
  const int ox0 = x0 + 2 * q;

  int nt = wave;
  if (nt < ntiles) {
#pragma unroll
    for (int s = 0; s < 4; ++s) ldb(nt, s, bq[s]);
  }
  for (; nt < ntiles; nt += 4) {
    const bool more = nt + 4 < ntiles;
    const int n0 = nt * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int sub = 0; sub < KSUB; ++sub) {
      const bf16x8 (&b)[3] = bq[sub & 3];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sub][2], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sub][0], b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sub][1], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sub][1], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sub][0], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sub][0], b[0], acc, 0, 0, 0);
      if (sub + 4 < KSUB) ldb(nt, sub + 4, bq[sub & 3]);
      else if (more) ldb(nt + 4, sub + 4 - KSUB, bq[sub & 3]);
    }
    // wave-private transpose: C layout (lane = channel) -> lane = 2x2 quad x 4 channels
#pragma unroll
    for (int r = 0; r < 16; ++r)
      tb[((r & 3) + 8 * (r >> 2) + 4 * lhi) * PWS_EP + l31] = acc[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    EpiSums S;
    const int n = n0 + c4 * 4;
    if (n < k.n_store) {
      f32x4 v[4];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        v[s] = *reinterpret_cast<const f32x4*>(tb + (4 * q + s) * PWS_EP + c4 * 4) * k.alpha;
      epi_item<0>(k, v, b0, y0, ox0, n, 0, 0, 0, S);
    }
    __builtin_amdgcn_wave_barrier();          // (the next tile's dump follows the reads above)
  }
}

}  // namespace

int p2l_pw_launch(const ConvK& k, int pro, hipStream_t st) {
  dim3 grid(k.n_mtiles * k.n_ntiles), block(256);
  static int ks_env = -1;                                // $P2L_PW_KS = 64 | 32 (default 32)
  if (ks_env < 0) { const char* e = getenv("P2L_PW_KS"); ks_env = e ? atoi(e) : 32; }
#define P2L_PW(PRO, KSV)                                                                     \
  do {                                                                                       \
    static bool attr_set = false;                                                            \
    if (!attr_set) {                                                                         \
      (void)hipFuncSetAttribute((const void*)pw_bf3_kernel<PRO, KSV>,                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
      attr_set = true;                                                                       \
    }                                                                                        \
    hipLaunchKernelGGL((pw_bf3_kernel<PRO, KSV>), grid, block, PwCfg<KSV>::LDS_BYTES, st, k); \
  } while (0)
#define P2L_PW_K(PRO) do { if (ks_env == 64) P2L_PW(PRO, 64); else P2L_PW(PRO, 32); } while (0)
  if (pro == P2L_PRO_NONE) P2L_PW_K(P2L_PRO_NONE);
  else if (pro == P2L_PRO_AFFINE_RELU) P2L_PW_K(P2L_PRO_AFFINE_RELU);
  else P2L_PW_K(P2L_PRO_AFFINE);
#undef P2L_PW_K
#undef P2L_PW
  return p2l_check_launch();
}

// streaming form: one block per 32 pixels (2 rows x 16)
int p2l_pws_launch(const ConvK& k, int pro, hipStream_t st) {
  dim3 grid(k.B * (k.H >> 1) * (k.W >> 4)), block(256);
  const int ksub = k.Cin / 16;
  const size_t lds = (size_t)(ksub * 768 + 4 * 32 * PWS_EP) * sizeof(float);
#define P2L_PWS(PRO, KSUB) hipLaunchKernelGGL((pws_bf3_kernel<PRO, KSUB>), grid, block, lds, st, k)
#define P2L_PWS_K(PRO) do { if (ksub == 4) P2L_PWS(PRO, 4); else P2L_PWS(PRO, 8); } while (0)
  if (k.arb_x) return P2L_EUNSUP;
  if (ksub != 4 && ksub != 8) return P2L_EUNSUP;
  if (pro == P2L_PRO_NONE) P2L_PWS_K(P2L_PRO_NONE);
  else if (pro == P2L_PRO_AFFINE_RELU) P2L_PWS_K(P2L_PRO_AFFINE_RELU);
  else P2L_PWS_K(P2L_PRO_AFFINE);
#undef P2L_PWS_K
#undef P2L_PWS
  return p2l_check_launch();
}
