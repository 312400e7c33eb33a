// 3x3 convolutions with THREE real channels on one side -- the image ends of the pipeline:
//   conv_to_rgb of the generator (128 -> 3, tanh) and its input gradient (3 -> 128, fused
//   activation backward), the first VGG16 / LPIPS conv (3 -> 64) and its input gradient (64 -> 3)
// (reached from pix2latent/model/biggan.py:58 and pix2latent/loss_functions.py:142 in the
// reference).  The generic kernel pads the thin side to 16 / 32 channels and multiplies zeros:
// 5.6 % of the BasinCMA step at 15-20 useful TFLOP/s.  Both forms here are memory-bound instead:
//
//   thin INPUT  (3 -> N): the 27 (tap, channel) products of a pixel are ONE K dimension
//     (padded to 32): two K-steps of the 32x32x16 MFMA per 32 output channels instead of
//     nine; the A fragments are gathered from an fp32 patch in LDS and split once per block,
//     every wave then walks over the output channels in tiles of 32 (weights from L2 in
//     fragment order) through the shared epilogue item;
//   thin OUTPUT (K -> 3): out[p][c] = sum_tap Z[p + tap][3 tap + c] with Z = X Wz a POINTWISE
//     product onto 27 (-> 32) columns: the MFMA work does not grow with the 9 taps; the
//     block computes Z for its 8x16 pixels + halo (6 MFMA row tiles, 3 waves) and every
//     output value is a sum of 9 LDS reads.
// Arithmetic: bf16x3 operand split, fp32 accumulate, as everywhere (include/p2l.h).
#include "p2l_conv_k.h"

using namespace p2lconv;

namespace {

constexpr int TH_PATCH = 10 * 18;                        // 8x16 outputs + halo

__device__ __forceinline__ void thin_store_split(float* As, int row, int q4, const f32x4 x) {
  bf16x4 ph, pm, pl;
  split3(x, ph, pm, pl);
  char* rb = reinterpret_cast<char*>(As) + row * 96 + (q4 & 1) * 8;
  char* rq = rb + bf3_chunk(q4 >> 1, row) * 16;           // pieces at +0 / +32 / +64 bytes
  *reinterpret_cast<bf16x4*>(rq) = ph;
  *reinterpret_cast<bf16x4*>(rq + 32) = pm;
  *reinterpret_cast<bf16x4*>(rq + 64) = pl;
}

// ------------------------------------------------------------------------------------------
// thin input: block = 8x16 pixels (wave w: image rows 2w, 2w+1 = 8 quads), all output channels
// ------------------------------------------------------------------------------------------
template <int PRO, bool ARB>
__global__ __launch_bounds__(256, ARB ? 3 : 4) void conv_thinin_kernel(const ConvK k) {
  __shared__ __attribute__((aligned(16))) float patch[TH_PATCH * 4];      // [pixel][3 + pad]
  __shared__ __attribute__((aligned(16))) float dump[4 * 32 * 36];
  __shared__ float red[2 * 4 * 32];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int swz = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_per_image = k.tiles_x * k.tiles_y;
  const int b = swz / tiles_per_image, tile_in_image = swz - b * tiles_per_image;
  const int ty = tile_in_image / k.tiles_x, tx = tile_in_image - ty * k.tiles_x;
  const int y0 = ty * 8, x0 = tx * 16;

  // ---- patch: 180 pixels, prologue per channel, zero padding after it ----------------------
  if (tid < TH_PATCH) {
    const int hy = tid / 18, hx = tid - hy * 18;
    const int iy = y0 + hy - 1, ix = x0 + hx - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < k.H && ix >= 0 && ix < k.W) {
      v = *reinterpret_cast<const f32x4*>(k.x + (size_t)((b * k.H + iy) * k.W + ix) * k.x_ld);
      if (PRO != P2L_PRO_NONE) {
        const f32x4 s = *reinterpret_cast<const f32x4*>(k.pro_s + b * k.pro_bstride);
        const f32x4 t = *reinterpret_cast<const f32x4*>(k.pro_t + b * k.pro_bstride);
        v = v * s + t;
        if (PRO == P2L_PRO_AFFINE_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = 0.f;
        }
      }
    }
    *reinterpret_cast<f32x4*>(patch + tid * 4) = v;
  }
  __syncthreads();

  // ---- A fragments: row l31 = quad (l31 >> 2), sub-pixel (l31 & 3) of this wave's two rows --
  bf16x8 a[2][3];
  {
    const int Q = l31 >> 2, s = l31 & 3;
    const int py = 2 * wave + (s >> 1), px = 2 * Q + (s & 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kk = ks * 16 + lhi * 8 + e;             // (tap, channel) = divmod(kk, 3)
        const int tap = kk / 3, c = kk - tap * 3;
        const int dy = tap / 3, dx = tap - dy * 3;
        v[e] = (kk < 27) ? patch[((py + dy) * 18 + px + dx) * 4 + c] : 0.f;
      }
      bf16x4 h[2], m[2], l[2];
      split3(f32x4{v[0], v[1], v[2], v[3]}, h[0], m[0], l[0]);
      split3(f32x4{v[4], v[5], v[6], v[7]}, h[1], m[1], l[1]);
      a[ks][0] = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
      a[ks][1] = __builtin_shufflevector(m[0], m[1], 0, 1, 2, 3, 4, 5, 6, 7);
      a[ks][2] = __builtin_shufflevector(l[0], l[1], 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }

  // ---- output channels in tiles of 32: weights [tile][k-step][piece][lane] x 16 B ------------
  const f32x4* wq = reinterpret_cast<const f32x4*>(k.w) + lane;
  const int ntiles = k.Cout >> 5;
  float* tb = dump + wave * 32 * 36;
  const int q = lane >> 3, c4 = lane & 7;                  // epilogue item: quad q, channels 4*c4..
  constexpr bool arb = ARB;
  const size_t arb_slot = (size_t)b * k.arb_nblk + tile_in_image;
  f32x4 bw[2][3];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int p = 0; p < 3; ++p) bw[ks][p] = wq[(ks * 3 + p) * 64];
  float blk_amax = 0.f, blk_amaxp = 0.f;                   // (P2LAmax: maxima of what this block stores)
  for (int nt = 0; nt < ntiles; ++nt) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 b1 = __builtin_bit_cast(bf16x8, bw[ks][0]);
      const bf16x8 b2 = __builtin_bit_cast(bf16x8, bw[ks][1]);
      const bf16x8 b3 = __builtin_bit_cast(bf16x8, bw[ks][2]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][2], b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], b3, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], b2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], b2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], b1, acc, 0, 0, 0);
    }
    if (nt + 1 < ntiles) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p) bw[ks][p] = wq[((size_t)(nt + 1) * 6 + ks * 3 + p) * 64];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      tb[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 36 + l31] = acc[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    EpiSums S;
    S.amax = blk_amax; S.amaxp = blk_amaxp;
    const int n = nt * 32 + c4 * 4;
    if (n < k.n_store) {
      f32x4 v[4];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        v[s] = *reinterpret_cast<const f32x4*>(tb + (4 * q + s) * 36 + c4 * 4) * k.alpha;
      epi_item<ARB ? 1 : 0>(k, v, b, y0 + 2 * wave, x0 + 2 * q, n, 0, 0, 0, S);
    }
    blk_amax = S.amax; blk_amaxp = S.amaxp;
    __builtin_amdgcn_wave_barrier();
    if (arb) {                                            // (block-uniform)
      f32x4 sgx = S.sgx, sg = S.sg;
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        sgx.x += __shfl_xor(sgx.x, o, 64); sgx.y += __shfl_xor(sgx.y, o, 64);
        sgx.z += __shfl_xor(sgx.z, o, 64); sgx.w += __shfl_xor(sgx.w, o, 64);
        sg.x += __shfl_xor(sg.x, o, 64); sg.y += __shfl_xor(sg.y, o, 64);
        sg.z += __shfl_xor(sg.z, o, 64); sg.w += __shfl_xor(sg.w, o, 64);
      }
      if (lane < 8) {
        *reinterpret_cast<f32x4*>(red + wave * 32 + lane * 4) = sgx;
        *reinterpret_cast<f32x4*>(red + 128 + wave * 32 + lane * 4) = sg;
      }
      __syncthreads();
      if (tid < 32 && nt * 32 + tid < k.n_store) {
        const float s0 = (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
        const float s1 = (red[128 + tid] + red[160 + tid]) + (red[192 + tid] + red[224 + tid]);
        const size_t o = arb_slot * k.Cout + nt * 32 + tid;
        k.arb_partial[o] = s0;
        k.arb_partial[(size_t)k.B * k.arb_nblk * k.Cout + o] = s1;
      }
      __syncthreads();
    }
  }
  // this block's partial maxima of what it stored (one per wave), for the conv that reads the tensor
  // next (P2LAmax)
  if (k.amax_out != nullptr || k.amax_outp != nullptr) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      blk_amax = fmaxf(blk_amax, __shfl_xor(blk_amax, o, 64));
      blk_amaxp = fmaxf(blk_amaxp, __shfl_xor(blk_amaxp, o, 64));
    }
    if (lane == 0) {
      const size_t slot = (size_t)b * k.amax_out_n + (size_t)tile_in_image * 4 + wave;
      if (k.amax_out != nullptr) k.amax_out[slot] = blk_amax;
      if (k.amax_outp != nullptr) k.amax_outp[slot] = blk_amaxp;
    }
  }
}

// ------------------------------------------------------------------------------------------
// thin output: block = 8x16 pixels, 3 waves x 2 MFMA row tiles = the 10x18 patch (192 rows)
// ------------------------------------------------------------------------------------------
constexpr int TO_ROWS = 192;
constexpr int TO_STAGE_FLOATS = TO_ROWS * 24;
constexpr int TO_ZP = 33;                                 // pitch of Z in LDS

template <int PRO>
__global__ __launch_bounds__(192, 4) void conv_thinout_kernel(const ConvK k) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 A stages, then Z
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int swz = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_per_image = k.tiles_x * k.tiles_y;
  const int b = swz / tiles_per_image, tile_in_image = swz - b * tiles_per_image;
  const int ty = tile_in_image / k.tiles_x, tx = tile_in_image - ty * k.tiles_x;
  const int y0 = ty * 8, x0 = tx * 16;

  // staging items: 192 rows x 4 channel quads over 192 threads = 4 per thread
  const int sv = tid & 3;
  int a_goff[4], a_row[4];
  unsigned a_valid = 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int p = (tid + 192 * it) >> 2;
    a_row[it] = p;
    a_goff[it] = 0;
    if (p < TH_PATCH) {
      const int hy = p / 18, hx = p - hy * 18;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      if (iy >= 0 && iy < k.H && ix >= 0 && ix < k.W) {
        a_goff[it] = ((b * k.H + iy) * k.W + ix) * k.x_ld + sv * 4;
        a_valid |= 1u << it;
      }
    }
  }
  const int s_off = b * k.pro_bstride + sv * 4;
  f32x4 xr[4], sr, tr, wr[3];
  const f32x4* wq = reinterpret_cast<const f32x4*>(k.w) + lane;           // [chunk][piece][lane]
  auto load_regs = [&](int c) {
#pragma unroll
    for (int it = 0; it < 4; ++it)
      xr[it] = *reinterpret_cast<const f32x4*>(k.x + (size_t)a_goff[it] + c * 16);
    if (PRO != P2L_PRO_NONE) {
      sr = *reinterpret_cast<const f32x4*>(k.pro_s + s_off + c * 16);
      tr = *reinterpret_cast<const f32x4*>(k.pro_t + s_off + c * 16);
    }
  };
  auto write_stage = [&](int buf) {
    float* As = smem + buf * TO_STAGE_FLOATS;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      f32x4 v = xr[it];
      if (PRO != P2L_PRO_NONE) {
        v = v * sr + tr;
        if (PRO == P2L_PRO_AFFINE_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
          v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
      }
      if (!((a_valid >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      thin_store_split(As, a_row[it], sv, v);
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  const int nchunks = k.nchunks;
  load_regs(0);
#pragma unroll
  for (int p = 0; p < 3; ++p) wr[p] = wq[p * 64];
  write_stage(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    const bf16x8 b1 = __builtin_bit_cast(bf16x8, wr[0]);
    const bf16x8 b2 = __builtin_bit_cast(bf16x8, wr[1]);
    const bf16x8 b3 = __builtin_bit_cast(bf16x8, wr[2]);
    if (more) {
      load_regs(c + 1);
#pragma unroll
      for (int p = 0; p < 3; ++p) wr[p] = wq[((size_t)(c + 1) * 3 + p) * 64];
    }
    const float* As = smem + (c & 1) * TO_STAGE_FLOATS;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int row = (2 * wave + m) * 32 + l31;
      const float* aq = As + row * 24 + bf3_chunk(lhi, row) * 4;
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(aq);
      const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(aq + 8);
      const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(aq + 16);
      f32x16 t = acc[m];
      t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, t, 0, 0, 0);
      acc[m] = t;
    }
    if (more) write_stage((c + 1) & 1);
    __syncthreads();
  }

  // ---- Z[patch pixel][27] -> LDS, then the 9-tap gather-sum ----------------------------------
  float* Z = smem;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      Z[((2 * wave + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * TO_ZP + l31] = acc[m][r];
  __syncthreads();
  const int nq = k.n_store >> 2;                          // float4 groups per pixel
  for (int it = tid; it < 128 * nq; it += 192) {
    const int px = it / nq, g = it - px * nq;
    const int py = px >> 4, pxx = px & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (g == 0) {
      float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float* z = Z + ((py + tap / 3) * 18 + pxx + tap % 3) * TO_ZP + tap * 3;
        s[0] += z[0]; s[1] += z[1]; s[2] += z[2];
      }
      v = f32x4{s[0], s[1], s[2], 0.f} * k.alpha;
    }
    if (k.bias) v += ld4(k.bias, (unsigned)(g * 4));
    v = act4(v, k.act);
    st4(k.y, (unsigned)(((b * k.H + y0 + py) * k.W + x0 + pxx) * k.y_ld + g * 4), v);
  }
}

// ---- weights ---------------------------------------------------------------------------------
// mode 0 (thin output, N <= 3 real): [chunk K/16][piece][lane]: lane (column 3 tap + c, half lhi),
//        e -> input channel 16 chunk + 8 lhi + e
// mode 1 (thin input, K <= 3 real):  [32-channel tile][k-step 0..1][piece][lane]: lane (channel,
//        half lhi), e -> kk = 16 ks + 8 lhi + e = 3 tap + c
__global__ __launch_bounds__(256) void thin_pack_kernel(const float* __restrict__ w,
                                                        f32x4* __restrict__ dst, int O, int I,
                                                        int N_pad, int K_pad, int flip, int mode,
                                                        int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;          // (group, lane)
  if (idx >= total) return;
  const int lane = idx & 63, grp = idx >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  // conv weight W[n][kch][tap]; forward: w[n][kch], input-gradient: w[kch][n] with mirrored taps
  auto wv = [&](int n, int kch, int tap) -> float {
    const int N = flip ? I : O, K = flip ? O : I;
    if (n >= N || kch >= K) return 0.f;
    return flip ? w[((size_t)kch * I + n) * 9 + (8 - tap)] : w[((size_t)n * I + kch) * 9 + tap];
  };
  float v[8];
  if (mode == 0) {
    const int tap = l31 / 3, c = l31 - tap * 3;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (l31 < 27) ? wv(c, grp * 16 + lhi * 8 + e, tap) : 0.f;
  } else {
    const int nt = grp >> 1, ks = grp & 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kk = ks * 16 + lhi * 8 + e;
      const int tap = kk / 3, c = kk - tap * 3;
      v[e] = (kk < 27) ? wv(nt * 32 + l31, c, tap) : 0.f;
    }
  }
  bf16x4 h[2], m[2], l[2];
  split3(f32x4{v[0], v[1], v[2], v[3]}, h[0], m[0], l[0]);
  split3(f32x4{v[4], v[5], v[6], v[7]}, h[1], m[1], l[1]);
  f32x4* o = dst + (size_t)grp * 192 + lane;
  o[0] = __builtin_bit_cast(f32x4, __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7));
  o[64] = __builtin_bit_cast(f32x4, __builtin_shufflevector(m[0], m[1], 0, 1, 2, 3, 4, 5, 6, 7));
  o[128] = __builtin_bit_cast(f32x4, __builtin_shufflevector(l[0], l[1], 0, 1, 2, 3, 4, 5, 6, 7));
}

}  // namespace

// floats of the thin image behind the direct bf16x3 image of a P2L_WFMT_BF16X3T weight
size_t p2l_thin_weight_floats(int N_pad, int K_pad) { return (size_t)(N_pad + K_pad) * 48; }

// which thin form a padded N_pad x K_pad 3x3 conv takes: 0 thin output (3 real outputs padded to
// 32), 1 thin input (3 real inputs padded to 16), -1 none
int p2l_thin_mode(int N_pad, int K_pad) {
  if (N_pad == 32 && K_pad > 16) return 0;
  if (K_pad == 16 && N_pad > 32) return 1;
  return -1;
}

int p2l_thin_pack(const float* w_oihw, int O, int I, int N_pad, int K_pad, int flip, float* dst,
                  hipStream_t st) {
  const int N = flip ? I : O, K = flip ? O : I;
  const int mode = p2l_thin_mode(N_pad, K_pad);
  if (mode < 0 || (mode == 0 && N > 3) || (mode == 1 && K > 3)) return P2L_EINVAL;
  const int groups = mode == 0 ? K_pad / 16 : (N_pad / 32) * 2;
  const int total = groups * 64;
  hipLaunchKernelGGL(thin_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w_oihw, (f32x4*)dst, O, I,
                     N_pad, K_pad, flip, mode, total);
  return p2l_check_launch();
}

int p2l_thinin_launch(const ConvK& k, int pro, hipStream_t st) {
  dim3 grid(k.n_mtiles), block(256);
#define P2L_TI(PRO)                                                                          \
  do {                                                                                       \
    if (k.arb_x) hipLaunchKernelGGL((conv_thinin_kernel<PRO, true>), grid, block, 0, st, k); \
    else hipLaunchKernelGGL((conv_thinin_kernel<PRO, false>), grid, block, 0, st, k);        \
  } while (0)
  if (pro == P2L_PRO_NONE) P2L_TI(P2L_PRO_NONE);
  else if (pro == P2L_PRO_AFFINE_RELU) P2L_TI(P2L_PRO_AFFINE_RELU);
  else P2L_TI(P2L_PRO_AFFINE);
#undef P2L_TI
  return p2l_check_launch();
}

int p2l_thinout_launch(const ConvK& k, int pro, hipStream_t st) {
  dim3 grid(k.n_mtiles), block(192);
  const size_t lds = (size_t)2 * TO_STAGE_FLOATS * sizeof(float);          // 36,864 B >= Z (25,344 B)
  if (pro == P2L_PRO_NONE) hipLaunchKernelGGL(conv_thinout_kernel<P2L_PRO_NONE>, grid, block, lds, st, k);
  else if (pro == P2L_PRO_AFFINE_RELU) hipLaunchKernelGGL(conv_thinout_kernel<P2L_PRO_AFFINE_RELU>, grid, block, lds, st, k);
  else hipLaunchKernelGGL(conv_thinout_kernel<P2L_PRO_AFFINE>, grid, block, lds, st, k);
  return p2l_check_launch();
}
