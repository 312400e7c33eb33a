// Fused self-attention of the BigGAN-deep generator in the fp32-grade fp16 x 2 arithmetic (round 4;
// bf16 x 3 until then: with three instead of six MFMAs per product the core kernels went 0.35 -> 0.25 ms).
//
// Role: SelfAttn.forward of the generator (reached from pix2latent/model/biggan.py:58 in the
// reference -> pytorch_pretrained_biggan SelfAttn): beta = softmax(theta^T phi) over the 1024
// pooled key positions, o = beta g, for 4096 query positions, 64 / 256 channels.  The
// 4096 x 1024 matrix is NEVER written: a block keeps 128 query rows, streams the keys / values
// in tiles of 32 through LDS and carries the running row maximum and row sum (online softmax).
// The same kernel, with the roles of the two position sets swapped and the row statistics
// given (MODE 1), recomputes the probabilities for the backward pass:
//     MODE 0:  O[r][:]  = sum_t softmax_t(S[r][t]) W[t][:]        R = theta, T = phi, W = g
//     MODE 1:  O[r][:]  = sum_t exp(S[t][r] - lse[t]) W[t][:]     R = phi,  T = theta, W = d(o)
//              (= d g: the gradient of the values)
//
// Work layout (MI355X: two blocks of 4 waves per CU, 74 KB of LDS each):
//   * wave = 32 private rows r: their 64 channels sit in registers as the B operand (split
//     once); the products are formed TRANSPOSED, S^T[t][r] = T R^T, so that a lane owns ONE
//     private row: row statistics are per-lane scalars and the probabilities leave the MFMA
//     already in the register layout the next MFMA wants as its B operand -- the 16 values a
//     lane holds are 16 of the 32 tile rows t; the value fragments are stored in that same
//     (permuted) row order, so P never moves between lanes or through LDS;
//   * O^T[channel][r] accumulates in 4 x 16 registers per wave for HALF of the 256 value
//     channels (blockIdx selects the half; S is recomputed by both halves: 24 of 72 MFMAs);
//   * the streamed operands are split into fp16 pieces ONCE per call by attn_prep_kernel and
//     stored in MFMA A-fragment order, a tile is a flat 25 KB copy done by the LDS DMA
//     (global_load_lds_dwordx4, double buffered, one barrier per tile);
//   * every product is the 3-term fp16 x 2 form of the conv kernels (include/p2l.h
//     P2L_WFMT_BF16X3W): each operand scaled by a power of two taken from ITS image's maximum
//     (attn_amax_kernel: 32 partial maxima per image and tensor in front of the split; the
//     probabilities, <= 1, by 2^14; d S^T by the maxima its own kernel leaves), two
//     round-to-nearest fp16 pieces, h*h + h*m + m*h, fp32 accumulate, exact un-scaling;
//     exp() is v_exp_f32 on (s - max) * log2(e).
#include "p2l_conv_k.h"
#include <atomic>

using namespace p2lconv;

namespace {

constexpr int AT_D = 64;                       // channels of theta / phi
constexpr int AT_DV = 256;                     // channels of g
constexpr int AT_NP = 2;                       // fp16 pieces per value
constexpr int AT_KU = 4 * AT_NP * 64;          // 16-byte units of the row-matrix part of a tile (512)
constexpr int AT_WU = 8 * AT_NP * 64;          // units of one value-channel half (1024)
constexpr int AT_LU = 64;                      // units reserved for the tile's 32 row statistics
constexpr int AT_TILE_U = AT_KU + 2 * AT_WU + AT_LU;          // global image: 2624 units / tile
constexpr int AT_BUF_U = AT_KU + AT_WU + AT_LU;                // LDS buffer: 1600 units
constexpr size_t AT_LDS_BYTES = (size_t)2 * AT_BUF_U * 16;     // 51,200 B
constexpr int AT_NAM = 32;                     // partial maxima per image and tensor
constexpr float AT_PSCALE = 16384.f, AT_PINV = 1.f / 16384.f;  // probabilities (<= 1) as fp16 x 2
constexpr float AT_LOG2E = 1.4426950408889634f;

struct AttnK {
  const float* r;          // private rows [B][NR][64]
  const f32x4* img;        // streamed tiles [B][NT][AT_TILE_U]
  const float *am_r, *am_x, *am_w;   // partial maxima [B][AT_NAM] of the private rows | streamed rows | values
  float* out;              // MODE 0: [B][NR][256]; MODE 1: partial [tsplit][B][NR][256]
  float* lse;              // MODE 0: written, [B][NR]
  int B, NR, NT, tsplit;
};

// ---- per-image maxima -> powers of two --------------------------------------------------------
// partial maxima of |x| of up to four tensors in ONE launch: grid (AT_NAM, B, tensors), n = floats per
// image (a multiple of 4); out[t][b * AT_NAM + blockIdx.x].  No atomics, nothing to clear.
struct AmaxIn {
  const float* x[4];
  float* out[4];
  size_t n[4];
};
__global__ __launch_bounds__(256) void attn_amax_kernel(const AmaxIn p) {
  const int t = blockIdx.z, b = blockIdx.y;
  const float* x = t == 0 ? p.x[0] : (t == 1 ? p.x[1] : (t == 2 ? p.x[2] : p.x[3]));
  float* out = t == 0 ? p.out[0] : (t == 1 ? p.out[1] : (t == 2 ? p.out[2] : p.out[3]));
  const size_t n4 = (t == 0 ? p.n[0] : (t == 1 ? p.n[1] : (t == 2 ? p.n[2] : p.n[3]))) >> 2;
  const f32x4* xb = reinterpret_cast<const f32x4*>(x) + (size_t)b * n4;
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)AT_NAM * 256) {
    const f32x4 v = xb[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[b * AT_NAM + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// power of two of image b from its n partial maxima (every lane reads the same few floats)
__device__ __forceinline__ void img_scale(const float* part, int b, int n, float& scale, float& inv) {
  float m = 0.f;
  for (int i = 0; i < n; ++i) m = fmaxf(m, part[(size_t)b * n + i]);
  h2_scales(__builtin_bit_cast(unsigned, m), scale, inv);
}
// two round-to-nearest fp16 pieces of eight (scaled) values
__device__ __forceinline__ void split2(const f32x4 a, const f32x4 b, h16x8 (&p)[2]) {
  const h16x4 h0 = __builtin_convertvector(a, h16x4), h1 = __builtin_convertvector(b, h16x4);
  const h16x4 m0 = __builtin_convertvector(a - __builtin_convertvector(h0, f32x4), h16x4);
  const h16x4 m1 = __builtin_convertvector(b - __builtin_convertvector(h1, f32x4), h16x4);
  p[0] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
  p[1] = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- streamed operands -> fp16 x 2 fragment images ------------------------------------------
// x [B][N][64] (rows of the S product), w [B][N][256] (values), stat [B][N] or null.
// Tile of 32 rows:  K part [t 0..3][piece h | m][lane] : lane (row l31, half lhi) = x[row][16t + 8lhi + e]
//                   W part [half][j 0..3][u 0..1][piece][lane] : lane (channel 128 half + 32j + l31,
//                   lhi) = w[row(u, lhi, e)][channel], row(u,lhi,e) = 16u + 8(e>>2) + 4lhi + (e&3)
//                   -- the order in which the S^T accumulator registers of a lane hold the rows.
__global__ __launch_bounds__(256) void attn_prep_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ stat,
                                                        const float* __restrict__ am_x,
                                                        const float* __restrict__ am_w,
                                                        f32x4* __restrict__ img, int N, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;      // (b, tile, fragment 0..20, lane)
  if (idx >= total) return;
  const int lane = idx & 63;
  int q = idx >> 6;
  const int frag = q % 21; q /= 21;                     // 0..3 K | 4..19 W | 20 statistics
  const int NT = N >> 5;
  const int tile = q % NT, b = q / NT;
  const int l31 = lane & 31, lhi = lane >> 5;
  f32x4* dst = img + ((size_t)b * NT + tile) * AT_TILE_U;
  if (frag == 20) {
    if (lane < 8) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (stat) v = *reinterpret_cast<const f32x4*>(stat + (size_t)b * N + tile * 32 + lane * 4);
      dst[AT_KU + 2 * AT_WU + lane] = v;
    }
    return;
  }
  float v[8];
  int unit;
  if (frag < 4) {
    const float* src = x + ((size_t)b * N + tile * 32 + l31) * AT_D + frag * 16 + lhi * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[e];
    unit = frag * AT_NP * 64;
  } else {
    const int f = frag - 4;                              // half*8 + j*2 + u
    const int u = f & 1, ch = (f >> 1) * 32 + l31;       // channel 0..255 = half*128 + j*32 + l31
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = 16 * u + 8 * (e >> 2) + 4 * lhi + (e & 3);
      v[e] = w[((size_t)b * N + tile * 32 + row) * AT_DV + ch];
    }
    unit = AT_KU + f * AT_NP * 64;
  }
  float sc, inv;
  img_scale(frag < 4 ? am_x : am_w, b, AT_NAM, sc, inv);
  h16x8 pc[2];
  split2(f32x4{v[0], v[1], v[2], v[3]} * sc, f32x4{v[4], v[5], v[6], v[7]} * sc, pc);
  dst[unit + lane] = __builtin_bit_cast(f32x4, pc[0]);
  dst[unit + 64 + lane] = __builtin_bit_cast(f32x4, pc[1]);
}

// one fp32-grade product step: m*h + h*m + h*h (smallest terms first), fp32 accumulate
__device__ __forceinline__ f32x16 mfma3(const h16x8 (&a)[2], const h16x8 (&b)[2], f32x16 t) {
  t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], t, 0, 0, 0);
  t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], t, 0, 0, 0);
  t = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], t, 0, 0, 0);
  return t;
}

template <int MODE>
__global__ __launch_bounds__(256, 3) void attn_core_kernel(const AttnK a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  // block = (sample, 128-row block, value-channel half, slice of the tile range)
  int id = blockIdx.x;
  const int h = id & 1; id >>= 1;
  const int ts = id % a.tsplit; id /= a.tsplit;
  const int nrb = a.NR >> 7;
  const int rb = id % nrb, b = id / nrb;
  const int t_per = a.NT / a.tsplit, t0 = ts * t_per;
  const int row = rb * 128 + wave * 32 + l31;             // this lane's private row

  // ---- the image's powers of two: S = acc / (s_r s_x), O = acc / (2^14 s_w) -------------------
  float s_r, inv_s, inv_o;
  {
    float inv_r, s_x, inv_x, s_w, inv_w;
    img_scale(a.am_r, b, AT_NAM, s_r, inv_r);
    img_scale(a.am_x, b, AT_NAM, s_x, inv_x);
    img_scale(a.am_w, b, AT_NAM, s_w, inv_w);
    inv_s = inv_r * inv_x;
    inv_o = AT_PINV * inv_w;
  }
  // ---- private rows -> B fragments (split once) ------------------------------------------
  h16x8 rf[4][2];
  {
    const float* rp = a.r + ((size_t)b * a.NR + row) * AT_D + lhi * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(rp + t * 16);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(rp + t * 16 + 4);
      split2(x0 * s_r, x1 * s_r, rf[t]);
    }
  }

  // ---- tile DMA: K part | W half | statistics -------------------------------------------------
  const f32x4* img_b = a.img + (size_t)b * a.NT * AT_TILE_U;
  auto dma_tile = [&](int tile, int buf) {
    const f32x4* src = img_b + (size_t)tile * AT_TILE_U;
    constexpr int NI = (AT_KU + AT_WU) / 64 + ((MODE == 1) ? 1 : 0);   // 24 | 25 wave-instructions of 1 KB
#pragma unroll
    for (int i = 0; i < (NI + 3) / 4; ++i) {
      const int ins = wave + 4 * i;                       // wave-instruction index
      if (ins >= NI) continue;
      // source unit of the first lane: K part as is, W part of this half, statistics
      const int du = ins * 64;                            // destination unit in the buffer
      const int su = (du < AT_KU) ? du : (du < AT_KU + AT_WU ? du + h * AT_WU : du + AT_WU);
      const unsigned lds_base = __builtin_amdgcn_readfirstlane(
          (unsigned)(size_t)(__attribute__((address_space(3))) float*)(smem + ((size_t)buf * AT_BUF_U + du) * 4));
      const f32x4* g = src + su + lane;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                   :: "v"(g), "s"(lds_base) : "memory");
    }
  };

  f32x16 O[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[j][r] = 0.f;
  float m_run = -__builtin_inff(), l_run = 0.f;

  dma_tile(t0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int it = 0; it < t_per; ++it) {
    const int buf = it & 1;
    if (it + 1 < t_per) dma_tile(t0 + it + 1, buf ^ 1);
    const f32x4* Kb = reinterpret_cast<const f32x4*>(smem) + (size_t)buf * AT_BUF_U;
    const f32x4* Wb = Kb + AT_KU;

    // ---- S^T[t][r] = T R^T : 4 k-steps of 16 channels ------------------------------------
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      h16x8 ka[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) ka[p] = __builtin_bit_cast(h16x8, Kb[(t * 2 + p) * 64 + lane]);
      S = mfma3(ka, rf[t], S);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] *= inv_s;              // (exact: a power of two)

    // ---- probabilities ---------------------------------------------------------------------
    if (MODE == 0) {
      float mx = S[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * AT_LOG2E);   // exp(-inf) = 0 at start
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        S[r] = __builtin_amdgcn_exp2f((S[r] - m_new) * AT_LOG2E);
        rs += S[r];
      }
      rs += __shfl_xor(rs, 32, 64);
      l_run = l_run * alpha + rs;
      m_run = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {      // (rare after the first tiles)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[j][r] *= alpha;
      }
    } else {
      const float* st = reinterpret_cast<const float*>(Wb + AT_WU);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 ls = *reinterpret_cast<const f32x4*>(st + 8 * g + 4 * lhi);
        S[4 * g + 0] = __builtin_amdgcn_exp2f((S[4 * g + 0] - ls.x) * AT_LOG2E);
        S[4 * g + 1] = __builtin_amdgcn_exp2f((S[4 * g + 1] - ls.y) * AT_LOG2E);
        S[4 * g + 2] = __builtin_amdgcn_exp2f((S[4 * g + 2] - ls.z) * AT_LOG2E);
        S[4 * g + 3] = __builtin_amdgcn_exp2f((S[4 * g + 3] - ls.w) * AT_LOG2E);
      }
    }
    // B fragments of the second product: K-step u = accumulator registers 8u .. 8u+7
    h16x8 pf[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
      split2(f32x4{S[8 * u + 0], S[8 * u + 1], S[8 * u + 2], S[8 * u + 3]} * AT_PSCALE,
             f32x4{S[8 * u + 4], S[8 * u + 5], S[8 * u + 6], S[8 * u + 7]} * AT_PSCALE, pf[u]);

    // ---- O^T[channel][r] += W^T P^T --------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        h16x8 wa[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          wa[p] = __builtin_bit_cast(h16x8, Wb[((j * 2 + u) * 2 + p) * 64 + lane]);
        O[j] = mfma3(wa, pf[u], O[j]);
      }

    // next tile landed (this wave's DMAs), everyone done with this buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- output: O^T -> rows of 128 channels --------------------------------------------------
  {
    const float inv = (MODE == 0) ? inv_o / l_run : inv_o;   // (un-scaling; MODE 0: and the row sum)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[j][r] *= inv;
    if (MODE == 0 && h == 0 && lhi == 0) a.lse[(size_t)b * a.NR + row] = m_run + logf(l_run);
  }
  // (two passes of 64 channels: the dumps of the four waves fit the tile buffers, 33 of 51 KB)
  constexpr int DP = 65;                                   // odd pitch: conflict-free 4-byte writes
  float* dump = smem + wave * 32 * DP;
  float* ob = a.out + (((size_t)ts * a.B + b) * a.NR + rb * 128 + wave * 32) * AT_DV + h * 128;
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dump[l31 * DP + jj * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi] = O[2 * jp + jj][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
    for (int rr = 0; rr < 32; ++rr) ob[(size_t)rr * AT_DV + jp * 64 + lane] = dump[rr * DP + lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ==========================================================================================
// Backward: d S^T[k][q] = P[q][k] (dP[q][k] - D[q]),  P = exp(q.k - lse[q]),  dP = d(o) g^T,
// D[q] = d(o)[q] . o[q].  The 256-channel contraction of dP pairs every query row with every
// key row, so neither side can live in a wave's registers (32 x 256 x 2 pieces = 128 VGPR):
// this part is a tiled GEMM with BOTH operands streamed through LDS, two products into two
// accumulator sets (S over 4 chunks of 16 channels, dP over 16) and the elementwise epilogue
// above.  P and dP are never stored; d S^T is, once (it is the operand of the two remaining,
// memory-bound products d theta = dS phi and d phi = dS^T theta: attn_apply_kernel).
//   * block = 128 keys x 256 queries, 8 waves as 2 x 4, wave tile 64 x 64: 2 x (2 x 2 x 16)
//     accumulator registers; transposed (keys = MFMA rows) so that a lane owns one query and
//     lse / D are per-lane scalars;
//   * operands are fp16 x 2 "row images" [32-row tile][16-channel chunk][piece][lane] made once
//     per call (attn_rows_prep_kernel); a stage is two chunks = 48 KB (4 + 8 row tiles each),
//     LDS DMA, double buffered, one barrier per stage (24 MFMAs per wave).
constexpr int AX_FU = AT_NP * 64;                        // units of one (tile, chunk) fragment: h | m
constexpr int AX_STAGE_U = 2 * 12 * AX_FU;               // 2 chunks x (4 key tiles + 8 query tiles), 2 KB each
constexpr size_t AX_LDS_BYTES = (size_t)2 * AX_STAGE_U * 16;

struct AttnX {
  const f32x4 *kimg, *qimg, *vimg, *doimg;               // row images: 4 | 4 | 16 | 16 chunks
  const float *lse, *dsum;                               // [B][Nq]
  const float *am_k, *am_q, *am_v, *am_do;               // partial maxima [B][AT_NAM] of the four tensors
  float* dst;                                            // d S^T [B][Nk][Nq]
  float* am_dst;                                         // its partial maxima [B][blocks per image], written here
  int B, Nq, Nk;
};

// x [B][N][C] -> [b][tile][chunk][piece][lane]: lane (row l31, half lhi) = x[row][16 chunk + 8 lhi + e]
// (up to four matrices per launch: the backward pass needs phi, theta, g and d(o))
struct RowsPrep {
  const float* x[4];
  const float* am[4];      // partial maxima [B][AT_NAM]
  int N[4];                // rows per image
  f32x4* img[4];
  int C[4];                // channels
  int end[4];              // exclusive prefix of the item counts (b, tile, chunk, lane)
};

__global__ __launch_bounds__(256) void attn_rows_prep_kernel(const RowsPrep p) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  int m = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (idx >= p.end[m]) ++m;
  if (idx >= p.end[m]) return;
  if (m > 0) idx -= p.end[m - 1];
  const float* x = m == 0 ? p.x[0] : (m == 1 ? p.x[1] : (m == 2 ? p.x[2] : p.x[3]));
  f32x4* img = m == 0 ? p.img[0] : (m == 1 ? p.img[1] : (m == 2 ? p.img[2] : p.img[3]));
  const int C = m == 0 ? p.C[0] : (m == 1 ? p.C[1] : (m == 2 ? p.C[2] : p.C[3]));
  const float* am = m == 0 ? p.am[0] : (m == 1 ? p.am[1] : (m == 2 ? p.am[2] : p.am[3]));
  const int Nrows = m == 0 ? p.N[0] : (m == 1 ? p.N[1] : (m == 2 ? p.N[2] : p.N[3]));
  const int lane = idx & 63;
  int q = idx >> 6;
  const int nch = C >> 4;
  const int chunk = q % nch; q /= nch;                   // q = b * NT + tile
  const int l31 = lane & 31, lhi = lane >> 5;
  const float* src = x + ((size_t)q * 32 + l31) * C + chunk * 16 + lhi * 8;
  float sc, inv;
  img_scale(am, q / (Nrows >> 5), AT_NAM, sc, inv);
  h16x8 pc[2];
  split2(*reinterpret_cast<const f32x4*>(src) * sc, *reinterpret_cast<const f32x4*>(src + 4) * sc, pc);
  f32x4* dst = img + ((size_t)q * nch + chunk) * AX_FU + lane;
  dst[0] = __builtin_bit_cast(f32x4, pc[0]);
  dst[64] = __builtin_bit_cast(f32x4, pc[1]);
}

// D[b][q] = sum_c d(o)[q][c] o[q][c]; one wave per row
__global__ __launch_bounds__(256) void attn_rowdot_kernel(const float* __restrict__ a,
                                                          const float* __restrict__ b,
                                                          float* __restrict__ out, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const f32x4 x = reinterpret_cast<const f32x4*>(a + (size_t)row * AT_DV)[lane];
  const f32x4 y = reinterpret_cast<const f32x4*>(b + (size_t)row * AT_DV)[lane];
  const float s = wave_sum((x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w));
  if (lane == 0) out[row] = s;
}

__global__ __launch_bounds__(512, 1) void attn_ds_kernel(const AttnX a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;               // key half | query quarter of the block tile

  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int nkb = a.Nk >> 7, nqb = a.Nq >> 8;
  const int kb = id % nkb; id /= nkb;
  const int qb = id % nqb, b = id / nqb;
  const int ktile0 = b * (a.Nk >> 5) + kb * 4, qtile0 = b * (a.Nq >> 5) + qb * 8;

  // stage s of 10 = two 16-channel chunks: 0..1 = S (phi | theta images, 4 chunks per tile),
  // 2..9 = dP (g | d(o), 16 chunks per tile)
  auto dma_stage = [&](int sidx, int buf) {
    const bool sp = sidx < 2;
    const f32x4* ai = sp ? a.kimg : a.vimg;
    const f32x4* bi = sp ? a.qimg : a.doimg;
    const int nch = sp ? 4 : 16, ch0 = sp ? 2 * sidx : 2 * (sidx - 2);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int ins = wave + 8 * i;                       // 48 wave-instructions of 1 KB
      const int half = ins / 24, in2 = ins - half * 24;
      const int slot = in2 >> 1, piece = in2 & 1;         // slots 0..3 key tiles, 4..11 query tiles
      const f32x4* g = (slot < 4 ? ai + ((size_t)(ktile0 + slot) * nch + ch0 + half) * AX_FU
                                 : bi + ((size_t)(qtile0 + slot - 4) * nch + ch0 + half) * AX_FU) + piece * 64 + lane;
      const unsigned lds_base = __builtin_amdgcn_readfirstlane(
          (unsigned)(size_t)(__attribute__((address_space(3))) float*)(smem + ((size_t)buf * AX_STAGE_U + ins * 64) * 4));
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                   :: "v"(g), "s"(lds_base) : "memory");
    }
  };

  f32x16 accS[2][2], accP[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accS[mi][ni][r] = 0.f; accP[mi][ni][r] = 0.f; }

  auto compute = [&](int buf, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const f32x4* st = reinterpret_cast<const f32x4*>(smem) + (size_t)buf * AX_STAGE_U + half * 12 * AX_FU;
      h16x8 fa[2][2], fb[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          fa[i][p] = __builtin_bit_cast(h16x8, st[((2 * wm + i) * 2 + p) * 64 + lane]);
          fb[i][p] = __builtin_bit_cast(h16x8, st[((4 + 2 * wn + i) * 2 + p) * 64 + lane]);
        }
      // the three terms of a product stay in their order per accumulator (m h, h m, h h); consecutive
      // MFMAs go to different accumulators (no back-to-back dependence on the matrix pipe)
      constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mi][TA[t3]], fb[ni][TB[t3]],
                                                                 acc[mi][ni], 0, 0, 0);
    }
  };

  dma_stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int c = 0; c < 10; ++c) {
    const int buf = c & 1;
    if (c + 1 < 10) dma_stage(c + 1, buf ^ 1);
    if (c < 2) compute(buf, accS); else compute(buf, accP);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- d S^T = exp(S - lse) (dP - D), rows = keys, lanes = queries ---------------------------
  float inv_s, inv_p;
  {
    float s0, i0, s1, i1;
    img_scale(a.am_k, b, AT_NAM, s0, i0);
    img_scale(a.am_q, b, AT_NAM, s1, i1);
    inv_s = i0 * i1;
    img_scale(a.am_v, b, AT_NAM, s0, i0);
    img_scale(a.am_do, b, AT_NAM, s1, i1);
    inv_p = i0 * i1;
  }
  float mx = 0.f;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int q = qb * 256 + wn * 64 + ni * 32 + l31;
    const float ls = a.lse[(size_t)b * a.Nq + q], dd = a.dsum[(size_t)b * a.Nq + q];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      float* o = a.dst + ((size_t)b * a.Nk + kb * 128 + wm * 64 + mi * 32 + 4 * lhi) * a.Nq + q;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f((accS[mi][ni][r] * inv_s - ls) * AT_LOG2E);
        const float v = p * (accP[mi][ni][r] * inv_p - dd);
        o[(size_t)((r & 3) + 8 * (r >> 2)) * a.Nq] = v;
        mx = fmaxf(mx, fabsf(v));
      }
    }
  }
  // this block's maximum of |d S^T| for the two products that read it next (attn_apply_kernel)
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __syncthreads();                                       // (everybody is done with the stage buffers)
  if (lane == 0) smem[wave] = mx;
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int w = 1; w < 8; ++w) mx = fmaxf(mx, smem[w]);
    a.am_dst[(size_t)b * (nkb * nqb) + qb * nkb + kb] = mx;
  }
}

// O[r][0..63] = sum_t P'[t][r] W[t][0..63] for a STORED weight matrix M (the d S^T above):
//   TMAJ = true : P'[t][r] = M[t][r]   (rows of M = streamed index: lanes read consecutive r)
//   TMAJ = false: P'[t][r] = M[r][t]   (rows of M = private index: 16-byte reads along t)
// Same register layout trick as attn_core_kernel: the 32 x 32 tile of M is loaded straight into
// the MFMA C layout, split, and used as the B operand; W (64 channels) comes as the permuted
// fragment image [tile][j 0..1][u 0..1][piece][lane] (attn_wprep_kernel), 12 KB per tile, DMA.
constexpr int AP_TILE_U = 2 * 2 * AX_FU;                 // 512 units = 8 KB

struct AttnP {
  const float* m;          // stored weights, row pitch ldm, sample stride sm
  const f32x4* wimg;       // [B][NT][AP_TILE_U]
  const float* am_m; int n_am_m;     // partial maxima of m [B][n_am_m] (left by attn_ds_kernel)
  const float* am_w;                 // ... of the streamed rows [B][AT_NAM]
  float* out;              // [tsplit][B][NR][64]
  int B, NR, NT, tsplit, ldm;
  size_t sm;
};

__global__ __launch_bounds__(256) void attn_wprep_kernel(const float* __restrict__ w,
                                                         const float* __restrict__ am_w,
                                                         f32x4* __restrict__ img, int N, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;        // (b, tile, f = j*2+u, lane)
  if (idx >= total) return;
  const int lane = idx & 63;
  int q = idx >> 6;
  const int f = q & 3; q >>= 2;                          // q = b * NT + tile
  const int l31 = lane & 31, lhi = lane >> 5;
  const int u = f & 1, ch = (f >> 1) * 32 + l31;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int row = 16 * u + 8 * (e >> 2) + 4 * lhi + (e & 3);
    v[e] = w[((size_t)q * 32 + row) * AT_D + ch];
  }
  float sc, inv;
  img_scale(am_w, q / (N >> 5), AT_NAM, sc, inv);
  h16x8 pc[2];
  split2(f32x4{v[0], v[1], v[2], v[3]} * sc, f32x4{v[4], v[5], v[6], v[7]} * sc, pc);
  f32x4* dst = img + (size_t)q * AP_TILE_U + f * AX_FU + lane;
  dst[0] = __builtin_bit_cast(f32x4, pc[0]);
  dst[64] = __builtin_bit_cast(f32x4, pc[1]);
}

template <bool TMAJ>
__global__ __launch_bounds__(256, 4) void attn_apply_kernel(const AttnP a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  int id = blockIdx.x;
  const int ts = id % a.tsplit; id /= a.tsplit;
  const int nrb = a.NR >> 7;
  const int rb = id % nrb, b = id / nrb;
  const int t_per = a.NT / a.tsplit, t0 = ts * t_per;
  const int r0 = rb * 128 + wave * 32;
  const float* mb = a.m + (size_t)b * a.sm;

  const f32x4* img_b = a.wimg + (size_t)b * a.NT * AP_TILE_U;
  auto dma_tile = [&](int tile, int buf) {
#pragma unroll
    for (int i = 0; i < AP_TILE_U / 256; ++i) {
      const int ins = wave + 4 * i;                       // 8 wave-instructions
      const unsigned lds_base = __builtin_amdgcn_readfirstlane(
          (unsigned)(size_t)(__attribute__((address_space(3))) float*)(smem + ((size_t)buf * AP_TILE_U + ins * 64) * 4));
      const f32x4* g = img_b + (size_t)tile * AP_TILE_U + ins * 64 + lane;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                   :: "v"(g), "s"(lds_base) : "memory");
    }
  };
  // the 32 x 32 tile of M for tile t, in C-layout order: register 4g+i <-> streamed row 8g+4lhi+i
  f32x16 pm;
  auto load_m = [&](int tile) {
    if (TMAJ) {
      const float* p = mb + (size_t)(tile * 32 + 4 * lhi) * a.ldm + r0 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) pm[r] = p[(size_t)((r & 3) + 8 * (r >> 2)) * a.ldm];
    } else {
      const float* p = mb + (size_t)(r0 + l31) * a.ldm + tile * 32 + 4 * lhi;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 8 * g);
        pm[4 * g] = v.x; pm[4 * g + 1] = v.y; pm[4 * g + 2] = v.z; pm[4 * g + 3] = v.w;
      }
    }
  };

  f32x16 O[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[j][r] = 0.f;
  // powers of two of the stored matrix and of the streamed rows of this image
  float s_m, inv_o;
  {
    float inv_m, s_w, inv_w;
    img_scale(a.am_m, b, a.n_am_m, s_m, inv_m);
    img_scale(a.am_w, b, AT_NAM, s_w, inv_w);
    inv_o = inv_m * inv_w;
  }

  load_m(t0);
  dma_tile(t0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int it = 0; it < t_per; ++it) {
    const int buf = it & 1;
    h16x8 pf[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
      split2(f32x4{pm[8 * u + 0], pm[8 * u + 1], pm[8 * u + 2], pm[8 * u + 3]} * s_m,
             f32x4{pm[8 * u + 4], pm[8 * u + 5], pm[8 * u + 6], pm[8 * u + 7]} * s_m, pf[u]);
    if (it + 1 < t_per) { load_m(t0 + it + 1); dma_tile(t0 + it + 1, buf ^ 1); }
    const f32x4* Wb = reinterpret_cast<const f32x4*>(smem) + (size_t)buf * AP_TILE_U;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        h16x8 wa[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          wa[p] = __builtin_bit_cast(h16x8, Wb[((j * 2 + u) * 2 + p) * 64 + lane]);
        O[j] = mfma3(wa, pf[u], O[j]);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[j][r] *= inv_o;         // (exact: powers of two)
  // O^T[channel][r] -> out[r][64]
  constexpr int DP = 65;
  float* dump = smem + wave * 32 * DP;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      dump[l31 * DP + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi] = O[j][r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float* ob = a.out + (((size_t)ts * a.B + b) * a.NR + r0) * AT_D;
#pragma unroll 4
  for (int rr = 0; rr < 32; ++rr) ob[(size_t)rr * AT_D + lane] = dump[rr * DP + lane];
}

// out[i] = sum_s part[s][i] in slice order (deterministic)
__global__ __launch_bounds__(256) void attn_reduce_kernel(const float* __restrict__ part,
                                                          float* __restrict__ out, size_t n4,
                                                          int nsplit) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 s = reinterpret_cast<const f32x4*>(part)[i];
  for (int k = 1; k < nsplit; ++k) s += reinterpret_cast<const f32x4*>(part)[(size_t)k * n4 + i];
  reinterpret_cast<f32x4*>(out)[i] = s;
}

int attn_shape_ok(const P2LAttn* d) {
  return d && d->B >= 1 && d->d == AT_D && d->dv == AT_DV && d->Nq >= 128 && d->Nq % 128 == 0 &&
         d->Nk >= 128 && d->Nk % 128 == 0;
}

size_t img_bytes(int B, int N) { return (size_t)B * (N >> 5) * AT_TILE_U * 16; }

int run_prep(const float* x, const float* w, const float* stat, const float* am_x, const float* am_w,
             f32x4* img, int B, int N, hipStream_t st) {
  const int total = B * (N >> 5) * 21 * 64;
  hipLaunchKernelGGL(attn_prep_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, x, w, stat, am_x, am_w,
                     img, N, total);
  return p2l_check_launch();
}

template <int MODE>
int run_core(const AttnK& a, hipStream_t st) {
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_core_kernel<MODE>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid(a.B * (a.NR >> 7) * a.tsplit * 2), block(256);
  hipLaunchKernelGGL(attn_core_kernel<MODE>, grid, block, AT_LDS_BYTES, st, a);
  return p2l_check_launch();
}

// slices of the query range in the d(values) form: 8 row blocks x 2 halves per sample alone
// leave most of the chip idle
// (counted for a REFERENCE batch, not d->B: the slices are summed in a fixed order that follows their
//  count, and an image's gradient must not depend on how many images share the launch -- round 5;
//  until then 18 / 9 / 2 candidates ran 4 / 8 / 8 slices here and 2 / 4 / 8 in the d(queries) form.
//  Reference batch 9 for both: 8 slices here, 4 / 8 in the d(queries) / d(keys) forms -- what every local
//  batch <= 9 ran before; 18 candidates pay ~110 MB more partial-sum traffic per step (0.2 %); with 18 as
//  the reference here the d(values) kernel of a 2-candidate rank went 31 -> 51 us,
//  profiles/round5_small_batch_kernel_stats.csv)
constexpr int AT_REF_B_DV = 9, AT_REF_B_APPLY = 9;
int dv_tsplit(const P2LAttn* d) {
  int s = 1;
  while (s < 8 && (long)AT_REF_B_DV * (d->Nk >> 7) * 2 * s < 1024 && (d->Nq >> 5) % (2 * s) == 0) s *= 2;
  return s;
}

// slices of the streamed range when 128-row blocks alone do not fill the chip
int apply_tsplit(int /*B*/, int NR, int NT) {
  int s = 1;
  while (s < 8 && (long)AT_REF_B_APPLY * (NR >> 7) * s < 1024 && NT % (2 * s) == 0) s *= 2;
  return s;
}

// O[r][64] = sum_t P'[t][r] W[t][64] with the stored matrix m (see attn_apply_kernel)
int run_apply(bool tmaj, const float* m, int ldm, size_t sm, const float* am_m, int n_am_m, const float* w,
              const float* am_w, float* out, int B, int NR, int Nt, f32x4* wimg, float* part, hipStream_t st) {
  const int NT = Nt >> 5;
  const int total = B * NT * 4 * 64;
  hipLaunchKernelGGL(attn_wprep_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, am_w, wimg, Nt, total);
  int rc = p2l_check_launch();
  if (rc) return rc;
  const int s = apply_tsplit(B, NR, NT);
  AttnP a{};
  a.m = m; a.wimg = wimg; a.out = s > 1 ? part : out;
  a.am_m = am_m; a.n_am_m = n_am_m; a.am_w = am_w;
  a.B = B; a.NR = NR; a.NT = NT; a.tsplit = s; a.ldm = ldm; a.sm = sm;
  dim3 grid(B * (NR >> 7) * s), block(256);
  const size_t lds = (size_t)2 * AP_TILE_U * 16 > (size_t)4 * 32 * 65 * 4 ? (size_t)2 * AP_TILE_U * 16
                                                                          : (size_t)4 * 32 * 65 * 4;
  if (tmaj) hipLaunchKernelGGL(attn_apply_kernel<true>, grid, block, lds, st, a);
  else hipLaunchKernelGGL(attn_apply_kernel<false>, grid, block, lds, st, a);
  rc = p2l_check_launch();
  if (rc || s == 1) return rc;
  const size_t n4 = (size_t)B * NR * AT_D / 4;
  hipLaunchKernelGGL(attn_reduce_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, part, out, n4, s);
  return p2l_check_launch();
}

size_t apply_part_bytes(const P2LAttn* d) {
  const size_t a = (size_t)apply_tsplit(d->B, d->Nq, d->Nk >> 5) * d->B * d->Nq;
  const size_t b = (size_t)apply_tsplit(d->B, d->Nk, d->Nq >> 5) * d->B * d->Nk;
  return ((a > b ? a : b) * AT_D * sizeof(float) + 255) & ~(size_t)255;
}
size_t rows_img_bytes(int B, int N, int C) { return (size_t)B * (N >> 5) * (C >> 4) * AX_FU * 16; }
size_t amax_bytes(int B, int ntens) { return ((size_t)ntens * B * AT_NAM * sizeof(float) + 255) & ~(size_t)255; }
// partial maxima of up to four tensors ([B][n[i]] floats each) -> out + i * B * AT_NAM
int run_amax(int ntens, const float* const* x, const size_t* n, float* out, int B, hipStream_t st) {
  AmaxIn p{};
  for (int i = 0; i < ntens; ++i) { p.x[i] = x[i]; p.n[i] = n[i]; p.out[i] = out + (size_t)i * B * AT_NAM; }
  hipLaunchKernelGGL(attn_amax_kernel, dim3(AT_NAM, B, ntens), dim3(256), 0, st, p);
  return p2l_check_launch();
}
size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" int p2l_attn_supported(const P2LAttn* d) { return attn_shape_ok(d); }

extern "C" size_t p2l_attn_fwd_ws_bytes(const P2LAttn* d) {
  return attn_shape_ok(d) ? align256(img_bytes(d->B, d->Nk)) + amax_bytes(d->B, 3) : 0;
}

extern "C" int p2l_attn_fwd(const P2LAttn* d, const float* q, const float* k, const float* v,
                            float* out, float* lse, void* ws, size_t ws_bytes, void* stream) {
  if (!attn_shape_ok(d)) return P2L_EUNSUP;
  if (!q || !k || !v || !out || !lse) return P2L_EINVAL;
  if (!ws || ws_bytes < p2l_attn_fwd_ws_bytes(d)) return P2L_EWS;
  hipStream_t st = (hipStream_t)stream;
  float* am = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + align256(img_bytes(d->B, d->Nk)));
  const float* xs[3] = {q, k, v};
  const size_t ns[3] = {(size_t)d->Nq * AT_D, (size_t)d->Nk * AT_D, (size_t)d->Nk * AT_DV};
  int rc = run_amax(3, xs, ns, am, d->B, st);
  if (rc) return rc;
  const float *am_q = am, *am_k = am + (size_t)d->B * AT_NAM, *am_v = am + (size_t)2 * d->B * AT_NAM;
  rc = run_prep(k, v, nullptr, am_k, am_v, (f32x4*)ws, d->B, d->Nk, st);
  if (rc) return rc;
  AttnK a{};
  a.r = q; a.img = (const f32x4*)ws; a.out = out; a.lse = lse;
  a.am_r = am_q; a.am_x = am_k; a.am_w = am_v;
  a.B = d->B; a.NR = d->Nq; a.NT = d->Nk >> 5; a.tsplit = 1;
  return run_core<0>(a, st);
}

extern "C" size_t p2l_attn_bwd_dv_ws_bytes(const P2LAttn* d) {
  if (!attn_shape_ok(d)) return 0;
  const int s = dv_tsplit(d);
  return align256(img_bytes(d->B, d->Nq)) + amax_bytes(d->B, 3) +
         (s > 1 ? (size_t)s * d->B * d->Nk * AT_DV * sizeof(float) : 0);
}

extern "C" int p2l_attn_bwd_dv(const P2LAttn* d, const float* q, const float* k,
                               const float* dout, const float* lse, float* dv, void* ws,
                               size_t ws_bytes, void* stream) {
  if (!attn_shape_ok(d)) return P2L_EUNSUP;
  if (!q || !k || !dout || !lse || !dv) return P2L_EINVAL;
  if (!ws || ws_bytes < p2l_attn_bwd_dv_ws_bytes(d)) return P2L_EWS;
  hipStream_t st = (hipStream_t)stream;
  float* am = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + align256(img_bytes(d->B, d->Nq)));
  const float* xs[3] = {k, q, dout};
  const size_t ns[3] = {(size_t)d->Nk * AT_D, (size_t)d->Nq * AT_D, (size_t)d->Nq * AT_DV};
  int rc = run_amax(3, xs, ns, am, d->B, st);
  if (rc) return rc;
  const float *am_k = am, *am_q = am + (size_t)d->B * AT_NAM, *am_do = am + (size_t)2 * d->B * AT_NAM;
  rc = run_prep(q, dout, lse, am_q, am_do, (f32x4*)ws, d->B, d->Nq, st);
  if (rc) return rc;
  const int s = dv_tsplit(d);
  float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + align256(img_bytes(d->B, d->Nq)) +
                                         amax_bytes(d->B, 3));
  AttnK a{};
  a.r = k; a.img = (const f32x4*)ws; a.out = s > 1 ? part : dv; a.lse = nullptr;
  a.am_r = am_k; a.am_x = am_q; a.am_w = am_do;
  a.B = d->B; a.NR = d->Nk; a.NT = d->Nq >> 5; a.tsplit = s;
  rc = run_core<1>(a, st);
  if (rc || s == 1) return rc;
  const size_t n4 = (size_t)d->B * d->Nk * AT_DV / 4;
  hipLaunchKernelGGL(attn_reduce_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, part, dv, n4, s);
  return p2l_check_launch();
}

// ---- d theta, d phi --------------------------------------------------------------------------
// workspace: row images of phi, theta, g, d(o) | D | value images / partial sums of the two
// apply passes (sized for the larger) ; `dst` (B * Nk * Nq floats) receives d S^T
extern "C" size_t p2l_attn_bwd_qk_ws_bytes(const P2LAttn* d) {
  if (!attn_shape_ok(d)) return 0;
  size_t n = align256(rows_img_bytes(d->B, d->Nk, AT_D)) + align256(rows_img_bytes(d->B, d->Nq, AT_D)) +
             align256(rows_img_bytes(d->B, d->Nk, AT_DV)) + align256(rows_img_bytes(d->B, d->Nq, AT_DV)) +
             align256((size_t)d->B * d->Nq * sizeof(float)) + amax_bytes(d->B, 4) +
             align256((size_t)d->B * (d->Nk >> 7) * (d->Nq >> 8) * sizeof(float));
  const size_t wimg = align256((size_t)d->B * ((d->Nq > d->Nk ? d->Nq : d->Nk) >> 5) * AP_TILE_U * 16);
  return n + wimg + apply_part_bytes(d);
}

extern "C" int p2l_attn_bwd_qk(const P2LAttn* d, const float* q, const float* k, const float* v,
                               const float* out, const float* dout, const float* lse, float* dst,
                               float* dq, float* dk, void* ws, size_t ws_bytes, void* stream) {
  if (!attn_shape_ok(d) || d->Nq % 256) return P2L_EUNSUP;
  if (!q || !k || !v || !out || !dout || !lse || !dst || !dq || !dk) return P2L_EINVAL;
  if (!ws || ws_bytes < p2l_attn_bwd_qk_ws_bytes(d)) return P2L_EWS;
  hipStream_t st = (hipStream_t)stream;
  char* p = reinterpret_cast<char*>(ws);
  f32x4* kimg = (f32x4*)p; p += align256(rows_img_bytes(d->B, d->Nk, AT_D));
  f32x4* qimg = (f32x4*)p; p += align256(rows_img_bytes(d->B, d->Nq, AT_D));
  f32x4* vimg = (f32x4*)p; p += align256(rows_img_bytes(d->B, d->Nk, AT_DV));
  f32x4* doimg = (f32x4*)p; p += align256(rows_img_bytes(d->B, d->Nq, AT_DV));
  float* dsum = (float*)p; p += align256((size_t)d->B * d->Nq * sizeof(float));
  float* am = (float*)p; p += amax_bytes(d->B, 4);       // partial maxima of k, q, v, d(o)
  const int n_am_ds = (d->Nk >> 7) * (d->Nq >> 8);       // ... and of d S^T, one per block of attn_ds_kernel
  float* am_ds = (float*)p; p += align256((size_t)d->B * n_am_ds * sizeof(float));
  f32x4* wimg = (f32x4*)p; p += align256((size_t)d->B * ((d->Nq > d->Nk ? d->Nq : d->Nk) >> 5) * AP_TILE_U * 16);
  float* part = (float*)p;
  int rc;
  const float *am_k = am, *am_q = am + (size_t)d->B * AT_NAM, *am_v = am + (size_t)2 * d->B * AT_NAM,
              *am_do = am + (size_t)3 * d->B * AT_NAM;
  {
    RowsPrep rp{};
    const float* xs[4] = {k, q, v, dout};
    f32x4* imgs[4] = {kimg, qimg, vimg, doimg};
    const int Ns[4] = {d->Nk, d->Nq, d->Nk, d->Nq}, Cs[4] = {AT_D, AT_D, AT_DV, AT_DV};
    const size_t ns[4] = {(size_t)d->Nk * AT_D, (size_t)d->Nq * AT_D, (size_t)d->Nk * AT_DV, (size_t)d->Nq * AT_DV};
    if ((rc = run_amax(4, xs, ns, am, d->B, st))) return rc;
    int end = 0;
    for (int i = 0; i < 4; ++i) {
      rp.x[i] = xs[i]; rp.img[i] = imgs[i]; rp.C[i] = Cs[i];
      rp.am[i] = am + (size_t)i * d->B * AT_NAM; rp.N[i] = Ns[i];
      end += d->B * (Ns[i] >> 5) * (Cs[i] >> 4) * 64;
      rp.end[i] = end;
    }
    hipLaunchKernelGGL(attn_rows_prep_kernel, dim3(cdiv(end, 256)), dim3(256), 0, st, rp);
    if ((rc = p2l_check_launch())) return rc;
  }
  hipLaunchKernelGGL(attn_rowdot_kernel, dim3(cdiv((size_t)d->B * d->Nq, 4)), dim3(256), 0, st, dout,
                     out, dsum, d->B * d->Nq);
  if ((rc = p2l_check_launch())) return rc;
  {
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)attn_ds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
      attr_set = true;
    }
    AttnX a{};
    a.kimg = kimg; a.qimg = qimg; a.vimg = vimg; a.doimg = doimg; a.lse = lse; a.dsum = dsum;
    a.am_k = am_k; a.am_q = am_q; a.am_v = am_v; a.am_do = am_do; a.am_dst = am_ds;
    a.dst = dst; a.B = d->B; a.Nq = d->Nq; a.Nk = d->Nk;
    hipLaunchKernelGGL(attn_ds_kernel, dim3(d->B * (d->Nk >> 7) * (d->Nq >> 8)), dim3(512),
                       AX_LDS_BYTES, st, a);
    if ((rc = p2l_check_launch())) return rc;
  }
  const size_t sm = (size_t)d->Nk * d->Nq;
  // d theta[q] = sum_k dS^T[k][q] phi[k]   (private rows = queries, M rows = streamed keys)
  if ((rc = run_apply(true, dst, d->Nq, sm, am_ds, n_am_ds, k, am_k, dq, d->B, d->Nq, d->Nk, wimg, part, st))) return rc;
  // d phi[k] = sum_q dS^T[k][q] theta[q]   (private rows = keys = rows of M)
  return run_apply(false, dst, d->Nq, sm, am_ds, n_am_ds, q, am_q, dk, d->B, d->Nk, d->Nq, wimg, part, st);
}
