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
// measured 0.62-0.93x with it): a 16-channel chunk of a 1x1 conv has only 12 MFMAs per wave to
// hide the operand split behind plus two barriers.  Here a stage is several chunks: every thread
// splits its values while the previous stage's MFMAs run, one barrier pair per stage.  A block
// is ONE load -> split -> multiply -> store latency chain and these layers have few stages, so
// what counts is how many chains a CU has in flight: 32-channel stages (36 KB of LDS, <= 118
// VGPR: four blocks per CU) beat the first 64-channel form (72 KB, two blocks) by 1.14-1.40x
// and now also win on the 64 / 128-input-channel layers (a streaming kernel written for those
// -- 32-pixel blocks, A fragments in registers, waves sweeping different output tiles -- was
// 1.05-1.13x the fp32 kernel and is gone: this one is 1.2-1.4x).  Double-buffered stages (one
// barrier per stage instead of two) were measured too: 32-channel stages x 2 buffers = two
// blocks per CU again, 0.83x; 16-channel stages x 2 buffers (four blocks), 0.96x.
//
//   * block = the 128-pixel quad-ordered tile of the direct kernel (same epilogue: residual,
//     nearest-x2 shortcut, pooling, fused activation backward) x 64 output channels;
//     wave w owns pixels 32w .. 32w+31: 2 accumulators of 32x32;
//   * A (activations): global fp32 -> registers -> prologue affine / ReLU -> 3-way split ->
//     LDS rows of 96 B ([x1 k0-7 | x1 k8-15 | x2 .. | x3 ..], chunk index XOR bit 3 of the row);
//   * B (weights): pre-split at pack time into the same row format (p2l_pack_conv_weight_pw),
//     copied global -> registers -> LDS as an image.
#include "p2l_conv_k.h"
#include <type_traits>

#include <atomic>

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


// ---- fp16 x 2 form (include/p2l.h, P2L_WFMT_PW): the launch that wrote x handed its per-image
// maxima over (P2LAmax), so the image's power of two is known without a pass over the input.
// Same block, same stages; rows of 64 B ([h k0-7 | h k8-15 | m k0-7 | m k8-15], 16-byte chunk
// index XOR bits 2-3 of the row: 16 consecutive rows hit 16 different bank groups), the split of
// a value is cvt_pk | fma_mix | cvt_pk, three MFMAs per 32x32 tile and 16 channels instead of six.
constexpr int PWH_SUB = 2;                                // 32-channel stages
constexpr int PWH_A_FLOATS = PWH_SUB * 128 * 16, PWH_B_FLOATS = PWH_SUB * 64 * 16;
constexpr int PWH_EPI_FLOATS = 4 * 32 * (64 + 4);         // the vector epilogue's tile dumps (epilogue_vec<2>)
constexpr size_t PWH_LDS_BYTES =
    (size_t)(PWH_A_FLOATS + PWH_B_FLOATS > PWH_EPI_FLOATS ? PWH_A_FLOATS + PWH_B_FLOATS : PWH_EPI_FLOATS) * sizeof(float);
__device__ __forceinline__ int h2_chunk(int c, int row) { return c ^ ((row >> 2) & 3); }
// activation rows: the h / m halves of the ODD 16-channel sub-chunk are swapped on top of that (`sub`),
// so that the 16 lanes of a staging ds_write_b64 -- 2 pixels x 2 sub-chunks x 4 quarters of one piece --
// cover all 32 banks instead of 16 of them twice (a lane group of the fragment READS sees one sub-chunk:
// a constant term there)
__device__ __forceinline__ int h2_chunk_a(int c, int row, int sub) { return h2_chunk(c, row) ^ ((sub & 1) << 1); }
__device__ __forceinline__ void pw_store_split_h2(float* As, int row, int q4, int sub, const f32x4 x) {
  const h16x4 h = __builtin_convertvector(x, h16x4);
  const f32x4 w = __builtin_convertvector(h, f32x4);
  const h16x4 m = __builtin_convertvector(x - w, h16x4);
  char* rb = reinterpret_cast<char*>(As) + row * 64 + (q4 & 1) * 8;
  *reinterpret_cast<h16x4*>(rb + h2_chunk_a(q4 >> 1, row, sub) * 16) = h;
  *reinterpret_cast<h16x4*>(rb + h2_chunk_a(2 + (q4 >> 1), row, sub) * 16) = m;
}

// SM = the small-grid form (4^2 ... 16^2 layers: 16 ... 256 pixels per image, 512 ... 2048 channels;
// on the exact-fp32 MFMA until round 4, at 1/16 of the matrix rate and 3-9 x their floor): a 128-pixel
// tile spans several images, each with its OWN power of two (input rows scaled while they are staged,
// accumulator rows un-scaled at the end); blockIdx.y is a split-K slice of the stages whose un-scaled
// partial outputs go to k.ws for the deterministic finish kernel of the direct path (p2l_conv.hip);
// the maxima come from the producer (P2LAmax) or from the 64 partials per image of the pass in front.
// (A/B builds, tools/ab_build.sh p2l_pw -DP2L_PW_ABL=n: timing ablations of the stage loop -- results are
//  wrong when set: 1 no split / LDS writes (activations and weights; the loads then go too), 2 no global
//  loads of the stages, 8 no MFMAs, 16 no barriers; all after the first stage; 32 no epilogue)
#ifndef P2L_PW_ABL
#define P2L_PW_ABL 0
#endif
template <int PRO, bool SM = false>
__global__ __launch_bounds__(256, SM ? 3 : 4) void pw_h2_kernel(const ConvK k) {   // (SM: per-item prologue operands, 130-160 VGPR)
  constexpr int KS = 32, VPP = KS / 4;
  constexpr int A_ITERS = 128 * VPP / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + PWH_A_FLOATS;
  float* scl = smem + PWH_LDS_BYTES / sizeof(float);     // SM: [TB] x {x scale, output un-scale}

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int TW = 1 << k.tw_log, TH = 1 << k.th_log;
  const int swz = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = swz / k.n_ntiles, nt = swz - mt * k.n_ntiles;
  const int tiles_per_image = k.tiles_x * k.tiles_y;
  const int bt = mt / tiles_per_image;
  const int b0 = SM ? (bt << k.tb_log) : bt;             // (!SM: one image per tile, launcher)
  const int tile_in_image = mt - bt * tiles_per_image;
  const int ty = tile_in_image / k.tiles_x, tx = tile_in_image - ty * k.tiles_x;
  const int n0 = nt * 64;
  const int y0 = ty << k.th_log, x0 = tx << k.tw_log;

  // ---- the images' powers of two (bound max|s| max|x| + max|t| for a fused prologue on handed-over maxima) ----
  float x_scale = 1.f, out_scale = 1.f;
  {
    float sw, inv_w;
    h2_scales(__builtin_amdgcn_readfirstlane(k.w_tail[0]), sw, inv_w);
    if (!SM) {
      float a = 0.f, ms = 0.f, mt_ = 0.f;
      for (int i = tid; i < k.amax_in_n; i += 256) a = fmaxf(a, k.amax_in[(size_t)b0 * k.amax_in_n + i]);
      const bool bound = PRO != P2L_PRO_NONE && !k.amax_in_applied;   // (applied: the maxima ARE those of x*s+t)
      if (bound) {
        const float* ps = k.pro_s + (size_t)b0 * k.pro_bstride;
        const float* pt = k.pro_t + (size_t)b0 * k.pro_bstride;
        for (int c = tid; c < k.Cin; c += 256) { ms = fmaxf(ms, fabsf(ps[c])); mt_ = fmaxf(mt_, fabsf(pt[c])); }
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        a = fmaxf(a, __shfl_xor(a, o, 64));
        if (PRO != P2L_PRO_NONE) { ms = fmaxf(ms, __shfl_xor(ms, o, 64)); mt_ = fmaxf(mt_, __shfl_xor(mt_, o, 64)); }
      }
      if (lane == 0) { smem[wave * 4] = a; smem[wave * 4 + 1] = ms; smem[wave * 4 + 2] = mt_; }
      __syncthreads();
      a = fmaxf(fmaxf(smem[0], smem[4]), fmaxf(smem[8], smem[12]));
      ms = fmaxf(fmaxf(smem[1], smem[5]), fmaxf(smem[9], smem[13]));
      mt_ = fmaxf(fmaxf(smem[2], smem[6]), fmaxf(smem[10], smem[14]));
      __syncthreads();                                   // (the first stage is staged there next)
      if (PRO != P2L_PRO_NONE) a = (bound ? ms * a + mt_ : a) * 1.001f;
      float inv_x;
      h2_scales(__builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a)), x_scale, inv_x);
      out_scale = inv_x * inv_w;
    } else {
      // one WAVE per image (images w, w + 4 of the tile): shuffles only, one barrier for the tile
      const int TBn = 1 << k.tb_log;
      for (int t = wave; t < TBn; t += 4) {
        const int b = b0 + t;
        float a = 0.f, ms = 0.f, mt_ = 0.f;
        if (b < k.B) {
          if (k.amax_in != nullptr) {
            for (int i = lane; i < k.amax_in_n; i += 64) a = fmaxf(a, k.amax_in[(size_t)b * k.amax_in_n + i]);
            if (PRO != P2L_PRO_NONE && !k.amax_in_applied) {
              const f32x4* ps = reinterpret_cast<const f32x4*>(k.pro_s + (size_t)b * k.pro_bstride);
              const f32x4* pt = reinterpret_cast<const f32x4*>(k.pro_t + (size_t)b * k.pro_bstride);
              for (int c = lane; c < (k.Cin >> 2); c += 64) {
                const f32x4 s4 = ps[c], t4 = pt[c];
                ms = fmaxf(fmaxf(ms, fmaxf(fabsf(s4.x), fabsf(s4.y))), fmaxf(fabsf(s4.z), fabsf(s4.w)));
                mt_ = fmaxf(fmaxf(mt_, fmaxf(fabsf(t4.x), fabsf(t4.y))), fmaxf(fabsf(t4.z), fabsf(t4.w)));
              }
            }
          } else {
            a = k.amax[b * 64 + lane];                   // (the prologue was applied by the pass)
          }
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          a = fmaxf(a, __shfl_xor(a, o, 64));
          if (PRO != P2L_PRO_NONE) { ms = fmaxf(ms, __shfl_xor(ms, o, 64)); mt_ = fmaxf(mt_, __shfl_xor(mt_, o, 64)); }
        }
        if (PRO != P2L_PRO_NONE && k.amax_in != nullptr) a = (k.amax_in_applied ? a : ms * a + mt_) * 1.001f;
        float xs, inv_x;
        h2_scales(__builtin_bit_cast(unsigned, a), xs, inv_x);
        if (lane == 0) { scl[2 * t] = xs; scl[2 * t + 1] = inv_x * inv_w; }
      }
      __syncthreads();
    }
  }

  // ---- A staging: 128 pixels x 8 float4 per stage = 4 items per thread ----------------------
  const int av = tid & (VPP - 1);
  int a_goff[A_ITERS], a_soff[SM ? A_ITERS : 1];
  float a_xs[SM ? A_ITERS : 1];
  unsigned a_valid = 0;
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int p = (tid + 256 * it) / VPP;
    const int Q = p >> 2, s = p & 3;
    const int qx = Q & ((TW >> 1) - 1);
    const int qy = (Q >> (k.tw_log - 1)) & ((TH >> 1) - 1);
    const int iy = y0 + 2 * qy + (s >> 1), ix = x0 + 2 * qx + (s & 1);
    if (SM) {
      const int tb = Q >> (k.tw_log + k.th_log - 2);
      const int b = b0 + tb;
      a_goff[it] = 0; a_soff[it] = 0; a_xs[it] = scl[2 * tb];
      if (b < k.B) {                                     // (the grid is a whole number of tiles per image)
        a_goff[it] = ((b * k.H + iy) * k.W + ix) * k.x_ld + av * 4;
        a_soff[it] = b * k.pro_bstride + av * 4;
        a_valid |= 1u << it;
      }
    } else {
      a_goff[it] = ((b0 * k.H + iy) * k.W + ix) * k.x_ld + av * 4;
    }
  }
  const int s_off = b0 * k.pro_bstride + av * 4;
  // ---- B staging: [sub][64 rows][4 x 16 B] = 512 items per stage, 2 per thread; the packed image
  // is [chunk][32-channel tile][32 rows][64 B]: the 2 tiles of this block are one 4 KB run
  constexpr int B_ITERS = PWH_SUB * 256 / 256;
  int b_goff[B_ITERS];
#pragma unroll
  for (int it = 0; it < B_ITERS; ++it) {
    const int j = tid + 256 * it;
    const int sub = j / 256, within = j - sub * 256;
    b_goff[it] = (sub * (k.Cout >> 5) + (n0 >> 5)) * 32 * 16 + within * 4;
  }
  const int b_stage = PWH_SUB * (k.Cout >> 5) * 32 * 16;   // floats per stage of the image

  // Full-tile form (DEEP): the activations of stage st + 2 are requested while stage st is multiplied
  // -- two register sets for A, one for the weights (L2 hits, one stage ahead) and the prologue vector.
  // Round 5: a launch with fewer blocks than the chip has slots (32^2 1024->256: 576 blocks on 1 024
  // slots; 64^2 512->256: 2.25 rounds) took stages x (load -> split -> multiply), 1.8 us per stage of
  // 0.2 us of MFMA work, the one-stage-ahead request exposed in front of every split
  // (profiles/round4_conv1x1_roofline.txt: 2.0-3.9 x floor on those layers).  The loads are issued weights
  // first, so the wait in front of write_lds leaves the youngest A_ITERS requests in flight.
#ifdef P2L_AB_PW_SHALLOW              // (A/B build: the round-4 pipeline, one stage ahead)
  constexpr bool DEEP = false;
#else
  constexpr bool DEEP = !SM;
#endif
  constexpr int NSET = DEEP ? 2 : 1;
  f32x4 xr[NSET][A_ITERS], wr[B_ITERS], sr[SM ? A_ITERS : 1], tr[SM ? A_ITERS : 1];
  // DEEP: image b0 and the weight image as buffer resources -- 32-bit lane offsets that never change and
  // the stage as the SCALAR offset: no address arithmetic in the loop (with 64-bit pointers hipcc builds
  // each address in the destination registers of the load and waits for whatever was last loaded there)
  __amdgpu_buffer_rsrc_t x_rs, w_rs;
  unsigned a_boff[A_ITERS], b_boff[B_ITERS];
  if (DEEP) {
    auto sgpr = [](const void* p) {
      const unsigned long long v = reinterpret_cast<unsigned long long>(p);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
      const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return reinterpret_cast<char*>(((unsigned long long)hi << 32) | lo);
    };
    const size_t img = (size_t)k.H * k.W * k.x_ld;
    x_rs = __builtin_amdgcn_make_buffer_rsrc(sgpr(k.x + (size_t)b0 * img), 0,
                                             __builtin_amdgcn_readfirstlane((int)(img * 4)), 0x00020000);
    w_rs = __builtin_amdgcn_make_buffer_rsrc(sgpr(k.w), 0,
                                             __builtin_amdgcn_readfirstlane(k.Cin * k.Cout * 4), 0x00020000);
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) a_boff[it] = (unsigned)(a_goff[it] - b0 * (int)img) * 4u;
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) b_boff[it] = (unsigned)b_goff[it] * 4u;
  }
  auto load_a = [&](int st, auto SET_) {
    constexpr int set = decltype(SET_)::value;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      if (DEEP) {
        xr[set][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, a_boff[it], st * (KS * 4), 0));
        continue;
      }
      xr[set][it] = *reinterpret_cast<const f32x4*>(k.x + (size_t)a_goff[it] + st * KS);
      if (SM && PRO != P2L_PRO_NONE) {
        sr[it] = *reinterpret_cast<const f32x4*>(k.pro_s + a_soff[it] + st * KS);
        tr[it] = *reinterpret_cast<const f32x4*>(k.pro_t + a_soff[it] + st * KS);
      }
    }
  };
  // DEEP: the image's prologue vectors sit in LDS (copied once, below): no registers held for them
  // across the multiply phase (with them the two A sets spilled: 128 VGPR + 40 B of scratch)
  float* st_lds = smem + PWH_LDS_BYTES / sizeof(float) + 32;           // [2][Cin]
  auto load_b = [&](int st) {
    if (!SM && !DEEP && PRO != P2L_PRO_NONE) {
      sr[0] = *reinterpret_cast<const f32x4*>(k.pro_s + s_off + st * KS);
      tr[0] = *reinterpret_cast<const f32x4*>(k.pro_t + s_off + st * KS);
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      if (DEEP) wr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, b_boff[it], st * (b_stage * 4), 0));
      else wr[it] = *reinterpret_cast<const f32x4*>(k.w + (size_t)st * b_stage + b_goff[it]);
    }
  };
  auto write_lds = [&](auto SET_, int st) {
    constexpr int set = decltype(SET_)::value;
    if (!SM && DEEP && PRO != P2L_PRO_NONE) {
      sr[0] = *reinterpret_cast<const f32x4*>(st_lds + st * KS + av * 4);
      tr[0] = *reinterpret_cast<const f32x4*>(st_lds + k.Cin + st * KS + av * 4);
    }
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      f32x4 v = xr[set][it];
      if (PRO != P2L_PRO_NONE) {
        v = v * sr[SM ? it : 0] + tr[SM ? it : 0];
        if (PRO == P2L_PRO_AFFINE_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
          v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
      }
      if (SM && !((a_valid >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      const int p = (tid + 256 * it) / VPP;
      pw_store_split_h2(As + (av >> 2) * (128 * 16), p, av & 3, av >> 2, v * (SM ? a_xs[it] : x_scale));
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
  const int a_h = h2_chunk(lhi, a_row) * 4, a_m = h2_chunk(2 + lhi, a_row) * 4;     // (even sub-chunks; odd: swapped)
  const int b_h = h2_chunk(lhi, l31) * 4, b_m = h2_chunk(2 + lhi, l31) * 4;
  // SM: blockIdx.y = split-K slice of the stages (k.chunks_per_split stages each)
  const int nstages = k.Cin / KS;
  const int st_lo = SM ? (int)blockIdx.y * k.chunks_per_split : 0;
  const int st_hi = SM ? min(nstages, st_lo + k.chunks_per_split) : nstages;
  using S0_ = std::integral_constant<int, 0>;
  using S1_ = std::integral_constant<int, NSET - 1>;
  // one stage: LDS holds stage st; P = the A set that is FREE (stage st came from it)
  // (FULL_: a steady-state stage -- stages st + 1 and st + 2 exist, nothing is conditional: with the
  //  requests under `if (more)` hipcc's wait insertion has to cover the path on which none was issued
  //  and drains the A set it has just requested in front of every split, i.e. no deeper than before)
  auto stage = [&](int st, auto P_, auto Q_, auto FULL_) {
    constexpr bool full = decltype(FULL_)::value;
    const bool more = full || st + 1 < st_hi;
    if (DEEP) {
      if (more && !(P2L_PW_ABL & 2)) load_b(st + 1);
      if ((full || st + 2 < st_hi) && !(P2L_PW_ABL & 2)) load_a(st + 2, P_);
    } else if (more && !(P2L_PW_ABL & 2)) {
      load_a(st + 1, P_);
      load_b(st + 1);
    }
#pragma unroll
    for (int sub = 0; sub < PWH_SUB; ++sub) {
      const float* ar = As + (sub * 128 + a_row) * 16;
      const h16x8 ah = *reinterpret_cast<const h16x8*>(ar + ((sub & 1) ? a_m : a_h));
      const h16x8 am = *reinterpret_cast<const h16x8*>(ar + ((sub & 1) ? a_h : a_m));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float* br = Bs + (sub * 64 + j * 32 + l31) * 16;
        const h16x8 bh = *reinterpret_cast<const h16x8*>(br + b_h);
        const h16x8 bm = *reinterpret_cast<const h16x8*>(br + b_m);
        f32x16 t = acc[j];                               // smallest terms first
        if (!(P2L_PW_ABL & 8)) {
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, t, 0, 0, 0);
        } else { t[0] += (float)am[0] * (float)bh[0] + (float)ah[1] * (float)bm[1]; }
        acc[j] = t;
      }
    }
    if (!(P2L_PW_ABL & 16)) __syncthreads();
    if (more && !(P2L_PW_ABL & 1)) write_lds(Q_, st + 1);   // stage st + 1: set Q (DEEP: the other one)
    if (!(P2L_PW_ABL & 16)) __syncthreads();
  };
  load_a(st_lo, S0_{});
  load_b(st_lo);
  if (!SM && DEEP && PRO != P2L_PRO_NONE) {
    const float* ps = k.pro_s + (size_t)b0 * k.pro_bstride;
    const float* pt = k.pro_t + (size_t)b0 * k.pro_bstride;
    for (int c = tid * 4; c < k.Cin; c += 1024) {
      *reinterpret_cast<f32x4*>(st_lds + c) = *reinterpret_cast<const f32x4*>(ps + c);
      *reinterpret_cast<f32x4*>(st_lds + k.Cin + c) = *reinterpret_cast<const f32x4*>(pt + c);
    }
    __syncthreads();
  }
  write_lds(S0_{}, st_lo);
  if (DEEP && st_lo + 1 < st_hi) load_a(st_lo + 1, S1_{});
  __syncthreads();
  if (DEEP) {
    int st = st_lo;
    for (; st + 3 < st_hi; st += 2) {                    // steady state, two stages per trip
      stage(st, S0_{}, S1_{}, std::true_type{});
      stage(st + 1, S1_{}, S0_{}, std::true_type{});
    }
    for (; st < st_hi; st += 2) {                        // the last <= 3 stages
      stage(st, S0_{}, S1_{}, std::false_type{});
      if (st + 1 < st_hi) stage(st + 1, S1_{}, S0_{}, std::false_type{});
    }
  } else {
    for (int st = st_lo; st < st_hi; ++st) stage(st, S0_{}, S0_{}, std::false_type{});
  }
  if (!SM) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] *= out_scale;  // (exact: a power of two)
    if (!(P2L_PW_ABL & 32) || acc[0][0] == 12345.678f)
    epilogue_vec<2>(k, acc, smem, wave, lane, b0, y0, x0, n0, tile_in_image, 0, 0, 0);
    return;
  }
  // small-grid form: registers 4g .. 4g+3 of a lane are quad Q = wave*8 + 2g + lhi (one image each)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int Q = wave * 8 + 2 * g + lhi;
    const float os = scl[2 * (Q >> (k.tw_log + k.th_log - 2)) + 1];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[j][4 * g + s] *= os;
  }
  if (k.splitk > 1) {
    const size_t mtot = (size_t)k.B * k.H * k.W;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int Q = wave * 8 + 2 * g + lhi;
      const int qx = Q & ((TW >> 1) - 1);
      const int qy = (Q >> (k.tw_log - 1)) & ((TH >> 1) - 1);
      const int b = b0 + (Q >> (k.tw_log + k.th_log - 2));
      if (b >= k.B) continue;
      const size_t pix0 = ((size_t)b * k.H + y0 + 2 * qy) * k.W + x0 + 2 * qx;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* wp = k.ws + ((size_t)blockIdx.y * mtot + pix0) * k.Cout + n0 + j * 32 + l31;
        wp[0] = acc[j][g * 4 + 0];
        wp[k.Cout] = acc[j][g * 4 + 1];
        wp[(size_t)k.W * k.Cout] = acc[j][g * 4 + 2];
        wp[(size_t)(k.W + 1) * k.Cout] = acc[j][g * 4 + 3];
      }
    }
    return;
  }
  __syncthreads();                                       // (scl read by every wave before the dumps start)
  epilogue_vec<2>(k, acc, smem, wave, lane, b0, y0, x0, n0, tile_in_image, 0, 0, 0);
}

// fp16 x 2 image of a 1x1 weight: [K_pad/16][N_pad/32][32 rows][64 B], scaled by the layer's power
// of two; tail[0] = bits of max |w|
__global__ void pw_wmax_kernel(const float* w, size_t n, unsigned* tail) {
  float mx = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    mx = fmaxf(mx, fabsf(w[i]));
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(tail, __builtin_bit_cast(unsigned, mx));
}
__global__ void pw_pack_h2_kernel(const float* w, float* dst, int O, int I, int N_pad, int K_pad,
                                  int flip, const unsigned* tail) {
  // one thread per (chunk, 32-channel tile, row, k half): 8 values -> two 16-byte pieces
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)(K_pad >> 4) * N_pad * 2;
  if (idx >= total) return;
  float scale, inv;
  h2_scales(tail[0], scale, inv);
  const int kh = (int)(idx & 1);
  size_t q = idx >> 1;
  const int n = (int)(q % N_pad);
  const int cc = (int)(q / N_pad);
  const int N = flip ? I : O, K = flip ? O : I;
  h16x8 ph, pm;
  for (int e = 0; e < 8; ++e) {
    const int kk = cc * 16 + kh * 8 + e;
    float v = 0.f;
    if (n < N && kk < K) v = flip ? w[(size_t)kk * I + n] : w[(size_t)n * I + kk];
    const float x = v * scale;
    const _Float16 h = (_Float16)x;
    ph[e] = h; pm[e] = (_Float16)(x - (float)h);
  }
  const int row = n & 31;
  char* rb = reinterpret_cast<char*>(dst) + (((size_t)cc * (N_pad >> 5) + (n >> 5)) * 32 + row) * 64;
  *reinterpret_cast<h16x8*>(rb + h2_chunk(kh, row) * 16) = ph;
  *reinterpret_cast<h16x8*>(rb + h2_chunk(2 + kh, row) * 16) = pm;
}

}  // namespace

// floats of the fp16 x 2 image behind the bf16 x 3 image of a P2L_WFMT_PW buffer (+ 4 tail floats)
extern "C" size_t p2l_pw_h2_weight_floats(int N_pad, int K_pad) { return (size_t)N_pad * K_pad + 4; }
int p2l_pw_pack_h2(const float* w_oihw, int O, int I, int N_pad, int K_pad, int flip, float* dst,
                   hipStream_t st) {
  unsigned* tail = reinterpret_cast<unsigned*>(dst + (size_t)N_pad * K_pad);
  if (hipMemsetAsync(tail, 0, 16, st) != hipSuccess) return P2L_ELAUNCH;
  const size_t nw = (size_t)O * I;
  hipLaunchKernelGGL(pw_wmax_kernel, dim3((unsigned)(cdiv(nw, 256) < 256 ? cdiv(nw, 256) : 256)), dim3(256), 0, st,
                     w_oihw, nw, tail);
  const size_t total = (size_t)(K_pad >> 4) * N_pad * 2;
  hipLaunchKernelGGL(pw_pack_h2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w_oihw, dst, O, I, N_pad,
                     K_pad, flip, tail);
  return p2l_check_launch();
}

int p2l_pw_launch(const ConvK& k, int pro, hipStream_t st) {
  dim3 grid(k.n_mtiles * k.n_ntiles), block(256);
#define P2L_PW(PRO)                                                                          \
  do {                                                                                       \
    static std::atomic<bool> attr_set{false};                                                \
    if (!attr_set) {                                                                         \
      (void)hipFuncSetAttribute((const void*)pw_bf3_kernel<PRO, 32>,                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
      attr_set = true;                                                                       \
    }                                                                                        \
    hipLaunchKernelGGL((pw_bf3_kernel<PRO, 32>), grid, block, PwCfg<32>::LDS_BYTES, st, k);  \
  } while (0)
#define P2L_PWH(PRO, SMV)                                                                    \
  do {                                                                                       \
    static std::atomic<bool> attr_set{false};                                                \
    if (!attr_set) {                                                                         \
      (void)hipFuncSetAttribute((const void*)pw_h2_kernel<PRO, SMV>,                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
      attr_set = true;                                                                       \
    }                                                                                        \
    hipLaunchKernelGGL((pw_h2_kernel<PRO, SMV>), gridh, block,                               \
                       PWH_LDS_BYTES + 128 + ((!SMV && PRO != P2L_PRO_NONE) ? (size_t)k.Cin * 8 : 0), st, k); \
  } while (0)
  if (k.amax_in != nullptr || k.amax != nullptr) {     // fp16 x 2 (conv_launch_impl decides)
    // k.amax set: the small-grid form (multi-image tiles, split-K slices in blockIdx.y)
    const bool sm = k.amax != nullptr;
    dim3 gridh(k.n_mtiles * k.n_ntiles, sm ? k.splitk : 1);
    if (sm) {
      if (pro == P2L_PRO_NONE) P2L_PWH(P2L_PRO_NONE, true);
      else if (pro == P2L_PRO_AFFINE_RELU) P2L_PWH(P2L_PRO_AFFINE_RELU, true);
      else P2L_PWH(P2L_PRO_AFFINE, true);
      return p2l_check_launch();
    }
    if (pro == P2L_PRO_NONE) P2L_PWH(P2L_PRO_NONE, false);
    else if (pro == P2L_PRO_AFFINE_RELU) P2L_PWH(P2L_PRO_AFFINE_RELU, false);
    else P2L_PWH(P2L_PRO_AFFINE, false);
    return p2l_check_launch();
  }
#undef P2L_PWH
  if (pro == P2L_PRO_NONE) P2L_PW(P2L_PRO_NONE);
  else if (pro == P2L_PRO_AFFINE_RELU) P2L_PW(P2L_PRO_AFFINE_RELU);
  else P2L_PW(P2L_PRO_AFFINE);
#undef P2L_PW
  return p2l_check_launch();
}
