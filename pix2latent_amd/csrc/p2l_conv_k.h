// Shared by the conv kernels (p2l_conv.hip): kernel argument block and
// the fused epilogue.
#pragma once
#include "p2l_common.h"

namespace p2lconv {

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
  // general (non power-of-two) M grid: k.H x k.W grid points, tiles_x x tiles_y tiles per
  // image; input coordinates are valid in [0,iH) x [0,iW) and live in an ibH x ibW pixel
  // buffer; outputs go to an obH x obW pixel buffer.  partial != 0 when the grid is not a
  // multiple of the tile (border sub-pixels are skipped in the epilogue).
  int tiles_x, tiles_y, iH, iW, ibH, ibW, obH, obW, partial;
  // StyleGAN2 epilogue terms: v = acc * oscale[b][n] + noise_w * noise[b][pixel] + bias[n]
  const float* oscale; const float* noise; float noise_w; int oscale_bstride;
  // fused backward of a = max(x*s+t, 0) applied to the conv result (dgrad epilogue)
  const float* arb_x; const float* arb_s; const float* arb_t; const float* arb_skip;
  float* arb_partial;
  int arb_x_ld, arb_bstride, arb_skip_ld, arb_skip_C, arb_skip_ups, arb_nblk;
  int arb_nomask;   // 1: plain scale backward (g = da), StyleGAN2 modulation
  // sub-pixel mode of the TAPS=4 kernel (nearest-x2 upsample folded into the weights):
  //   1 = forward: low-res input, 4 output phases (blockIdx.y), output stride 2
  //   2 = input-gradient: the 4 phase planes of the high-res dY are 4 K-slices
  int sp_mode, sp_ncc;   // sp_ncc = channel chunks per phase plane (mode 2)
  int sp_skip;           // 1: the weights are those of a stride-2 TRANSPOSED conv (ext = 1): 7 of the 16 phase taps are zero
  // LDS pitch (in rows) of one line of the staged input patch; >= tile width + 2.  bf16x3:
  // 24 for 16-wide tiles, which puts the two pixel rows a wave's ds_read_b128 lane group
  // touches on disjoint banks (with tile width + 2 = 18 every activation-fragment read was a
  // 2-way bank conflict: rows 16 apart share a bank window for the 96-byte row).
  int hp;
  int form;  // P2LConv.form of this launch (kernel-form choice of the launchers)
  // fp16 x 2 arithmetic of the 16x16 Winograd kernel (p2l_wino.hip): [B][64] partial maxima of
  // |input| per image (written by the pass in front of the launch), bits of max |weight|
  float* amax;
  const unsigned* w_tail;
  // maxima handed between launches (P2LAmax): partial maxima of the RAW input tensor from the launch
  // that wrote it [B][amax_in_n]; this launch's own partial maxima of what it stores to y / yp, one
  // per block [B][amax_out_n]
  const float* amax_in; int amax_in_n;
  float* amax_out; float* amax_outp; int amax_out_n;
  // the affine the reader of y will fuse (P2LAmax.next_s / next_t): amax_out then holds the maxima of
  // |y*s + t|; amax_in_applied: amax_in was recorded that way with THIS launch's prologue
  const float* amax_ps; const float* amax_pt; int amax_pbstride;
  int amax_in_applied;
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// fp16 x 2 arithmetic (16x16 Winograd kernel, pointwise kernel: include/p2l.h P2L_WFMT_BF16X3W / _PW)
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
// power-of-two scale that puts max |x| (bits `mx`; 4x headroom for the Winograd input transform)
// below 2^15, and its inverse
__device__ __forceinline__ void h2_scales(unsigned mx, float& scale, float& inv) {
  int E = (int)((mx >> 23) & 0xffu);
  E = E < 40 ? 40 : (E > 254 ? 254 : E);
  scale = __builtin_bit_cast(float, (unsigned)(266 - E) << 23);      // 2^(139 - E)
  inv = __builtin_bit_cast(float, (unsigned)(E - 12) << 23);         // 2^(E - 139)
}

__device__ __forceinline__ f32x4 act4(f32x4 v, int act) {
  if (act == P2L_ACT_RELU) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  } else if (act == P2L_ACT_TANH) {
    v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
  } else if (act == P2L_ACT_LRELU_SQRT2) {   // FusedLeakyReLU: lrelu(0.2) * sqrt(2)
    const float a = 1.41421356237f, c = 0.2f * 1.41421356237f;
    v.x *= v.x > 0.f ? a : c; v.y *= v.y > 0.f ? a : c;
    v.z *= v.z > 0.f ? a : c; v.w *= v.w > 0.f ? a : c;
  }
  return v;
}
__device__ __forceinline__ f32x4 ld4(const float* p, unsigned off) {
  return *reinterpret_cast<const f32x4*>(p + (size_t)off);
}
__device__ __forceinline__ void st4(float* p, unsigned off, f32x4 v) {
  *reinterpret_cast<f32x4*>(p + (size_t)off) = v;
}

// position of logical 16-byte chunk c (0..5) inside LDS row `row`: lowest bit XOR-ed with
// bit 3 of the row.  Rows 8 or 24 apart start on the same bank (96-byte pitch = 24 dwords)
// and get distinct 16-byte windows this way; a window may only move by +-4 dwords (row
// bases are multiples of 8 dwords), so rows 16 apart - which the 2x2-quad pixel order does
// put into one ds_read_b128 lane group - still collide unless the patch lines are 24 rows
// apart in LDS (ConvK::hp; a rotation over all 6 chunks was tried: 46 % conflicts).
__device__ __forceinline__ int bf3_chunk(int c, int row) { return c ^ ((row >> 3) & 1); }

// x = h + m + l with three round-to-nearest bf16 pieces.  Written on PAIRS: v_cvt_pk_bf16_f32
// (half rate), shift / mask back to fp32, two subtractions: 5.5 VALU per value.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 widen2(const bf16x2 p) {
  const unsigned u = __builtin_bit_cast(unsigned, p);
  return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}
// a - b on two values.  Two forms:
//  * P2L_SCALAR_SPLIT (kernels that interleave the split with their own MFMAs, p2l_wino.hip): two
//    v_sub_f32.  Measured on gfx950 (tools/micro/issue_rate.hip): a packed fp32 add next to
//    v_mfma_f32_32x32x16_bf16 holds the matrix pipe for ~10 cycles (2 per MFMA: +50 % time) while
//    up to 4 plain VALU instructions per MFMA and wave are free; such a TU is also built with
//    -packed-fp32-ops so that hipcc does not pack additions itself.
//  * otherwise one v_pk_add_f32 with neg modifiers (hipcc turns every vector subtraction, and
//    fma(b, -1, a), into scalar v_sub_f32; only additions get packed): kernels whose split runs
//    in its own phase are issue-bound there and measured 3-7 % slower with the scalar form.
__device__ __forceinline__ f32x2 pk_sub(const f32x2 a, const f32x2 b) {
#ifdef P2L_SCALAR_SPLIT
  return f32x2{a.x - b.x, a.y - b.y};
#else
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
#endif
}
__device__ __forceinline__ f32x4 sub4(const f32x4 a, const f32x4 b) {
  const f32x2 lo = pk_sub(f32x2{a.x, a.y}, f32x2{b.x, b.y}), hi = pk_sub(f32x2{a.z, a.w}, f32x2{b.z, b.w});
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
// v - (the two bf16 values of p): shift / mask back to fp32 + two subtractions.
// (Tried: gfx950's v_dot2c_f32_bf16 with the selectors (-1, 0) / (0, -1) takes one half of a packed
// pair off an fp32 value in ONE instruction -- 3 instead of 5 VALU per split stage, bit-identical
// residuals (tools/micro/split_exact.hip).  But a DOT instruction holds the matrix pipe like
// v_pk_add_f32 does (tools/micro/issue_rate.hip rows "dot2c": +12 cycles each next to an MFMA): the
// 16x16 Winograd kernel got 8 % slower, the kernels with a separate split phase did not change, the
// step went 922 -> 898 evals/s.  Also: hipcc emits the selector (-1, 0) = 0x0000BF80 as the inline
// constant "-1.0", which the hardware reads as 0xBF800000 = (0, -1) in this instruction.)
__device__ __forceinline__ f32x2 resid2(const f32x2 v, const bf16x2 p) { return pk_sub(v, widen2(p)); }
__device__ __forceinline__ void split3_pair(const f32x2 v, bf16x2& h, bf16x2& m, bf16x2& l) {
  h = __builtin_convertvector(v, bf16x2);
  const f32x2 r1 = resid2(v, h);
  m = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = resid2(r1, m);
  l = __builtin_convertvector(r2, bf16x2);
}
__device__ __forceinline__ void split3(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
  bf16x2 h0, m0, l0, h1, m1, l1;
  split3_pair(f32x2{v.x, v.y}, h0, m0, l0);
  split3_pair(f32x2{v.z, v.w}, h1, m1, l1);
  h = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
  m = __builtin_shufflevector(m0, m1, 0, 1, 2, 3);
  l = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
}

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
                                              float bias_n, float* amax = nullptr,
                                              float* amaxp = nullptr) {
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
    if (amax) *amax = fmaxf(*amax, fabsf(k.amax_ps ? t * k.amax_ps[(size_t)b * k.amax_pbstride + n] +
                                                     k.amax_pt[(size_t)b * k.amax_pbstride + n] : t));
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
    if (amaxp) *amaxp = fmaxf(*amaxp, fabsf(p));
  }
}


// ---- one epilogue item ------------------------------------------------------------------
// quad = 2x2 output pixels with top-left (oy0, ox0) of image b, times FOUR consecutive output
// channels n..n+3.  v[s] = alpha-scaled accumulators of the quad's pixels in the order
// (0,0) (0,1) (1,0) (1,1).  Handles every epilogue mode of the conv kernels:
//   v = v*oscale + bias + noise + residual ; act ; mask ; store ; 2x2 max/sum pool, or
//   ARB (input-gradient convs): g = (x*s+t>0) ? da : 0 ; dx = g*s + shortcut ;
//   running sums of g*x and g per channel (S)  [da 2x2-summed first if pool == SUM].
// osh = 1: sub-pixel forward, the block writes phase (ph_y, ph_x) of the output buffer.
struct EpiSums {
  f32x4 sgx = {0, 0, 0, 0}, sg = {0, 0, 0, 0};
  float amax = 0.f, amaxp = 0.f;   // max |value stored to y| / |... to yp| (k.amax_out)
};
__device__ __forceinline__ float absmax4(float m, const f32x4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}

// ARBM: -1 = decided at run time (k.arb_x), 0 / 1 = known at compile time
template <int ARBM = -1>
__device__ __forceinline__ void epi_item(const ConvK& k, f32x4 (&v)[4], int b, int oy0, int ox0,
                                         int n, int osh, int ph_y, int ph_x, EpiSums& S) {
  const bool arb = ARBM < 0 ? (k.arb_x != nullptr) : (ARBM == 1);
  const bool pool_sum = k.pool == P2L_POOL_SUM;
  // osh = 1: sub-pixel forward, this block writes phase (ph_y, ph_x) of the output buffer
  const unsigned OW = (unsigned)k.obW;
  const unsigned pix0 = ((unsigned)(b * k.obH) + ((unsigned)oy0 << osh) + (unsigned)ph_y) * OW +
                        ((unsigned)ox0 << osh) + (unsigned)ph_x;
  const unsigned sub[4] = {0u, 1u << osh, OW << osh, (OW << osh) + (1u << osh)};
  // border quads of a grid that is not a multiple of the tile
  bool ok[4] = {true, true, true, true};
  if (k.partial) {
#pragma unroll
    for (int s = 0; s < 4; ++s) ok[s] = (oy0 + (s >> 1) < k.H) && (ox0 + (s & 1) < k.W);
    if (!ok[0]) return;
  }
  const unsigned ppix = (unsigned)((b * (k.H >> 1) + (oy0 >> 1)) * (k.W >> 1) + (ox0 >> 1));

  if (!arb) {
    f32x4 bias4 = {0, 0, 0, 0};
    if (k.bias) bias4 = ld4(k.bias, (unsigned)n);
    f32x4 osc = {1.f, 1.f, 1.f, 1.f};
    if (k.oscale) osc = ld4(k.oscale, (unsigned)(b * k.oscale_bstride + n));
    f32x4 ns4 = {1.f, 1.f, 1.f, 1.f}, nt4 = {0, 0, 0, 0};     // the reader's prologue (maxima only)
    if (k.amax_ps) { ns4 = ld4(k.amax_ps, (unsigned)(b * k.amax_pbstride + n)); nt4 = ld4(k.amax_pt, (unsigned)(b * k.amax_pbstride + n)); }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (!ok[s]) continue;
      const unsigned pix = pix0 + sub[s];
      f32x4 t = v[s] * osc + bias4;
      if (k.noise) t += k.noise_w * k.noise[(size_t)pix];
      if (k.res) t += ld4(k.res, (k.res_ups ? ppix : pix) * (unsigned)k.res_ld + (unsigned)n);
      t = act4(t, k.act);
      if (k.mask) {
        const f32x4 m = ld4(k.mask, pix * (unsigned)k.mask_ld + (unsigned)n);
        t.x = m.x > 0.f ? t.x : 0.f; t.y = m.y > 0.f ? t.y : 0.f;
        t.z = m.z > 0.f ? t.z : 0.f; t.w = m.w > 0.f ? t.w : 0.f;
      }
      if (k.y) st4(k.y, pix * (unsigned)k.y_ld + (unsigned)n, t);
      if (k.amax_out) S.amax = absmax4(S.amax, k.amax_ps ? t * ns4 + nt4 : t);
      v[s] = t;
    }
    if (k.pool) {
      f32x4 p;
      if (k.pool == P2L_POOL_MAX) {
        p.x = fmaxf(fmaxf(v[0].x, v[1].x), fmaxf(v[2].x, v[3].x));
        p.y = fmaxf(fmaxf(v[0].y, v[1].y), fmaxf(v[2].y, v[3].y));
        p.z = fmaxf(fmaxf(v[0].z, v[1].z), fmaxf(v[2].z, v[3].z));
        p.w = fmaxf(fmaxf(v[0].w, v[1].w), fmaxf(v[2].w, v[3].w));
      } else {
        p = (v[0] + v[1]) + (v[2] + v[3]);
      }
      st4(k.yp, ppix * (unsigned)k.yp_ld + (unsigned)n, p);
      if (k.amax_outp) S.amaxp = absmax4(S.amaxp, p);
    }
  } else {
    const f32x4 s4 = ld4(k.arb_s, (unsigned)(b * k.arb_bstride + n));
    const f32x4 t4 = ld4(k.arb_t, (unsigned)(b * k.arb_bstride + n));
    const bool has_skip = k.arb_skip && n < k.arb_skip_C;
    const int nv = pool_sum ? 1 : 4;
    if (pool_sum) v[0] = (v[0] + v[1]) + (v[2] + v[3]);
    float* dst = pool_sum ? k.yp : k.y;
    const unsigned dld = (unsigned)(pool_sum ? k.yp_ld : k.y_ld);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < nv) {
        const unsigned pix = pool_sum ? ppix : pix0 + sub[s];
        const f32x4 xv = ld4(k.arb_x, pix * (unsigned)k.arb_x_ld + (unsigned)n);
        const f32x4 pre = xv * s4 + t4;
        f32x4 g = v[s];
        if (!k.arb_nomask) {
          g.x = pre.x > 0.f ? g.x : 0.f; g.y = pre.y > 0.f ? g.y : 0.f;
          g.z = pre.z > 0.f ? g.z : 0.f; g.w = pre.w > 0.f ? g.w : 0.f;
        }
        f32x4 o = g * s4;
        if (has_skip) {
          const unsigned ld = (unsigned)k.arb_skip_ld;
          if (k.arb_skip_ups) {
            // this output pixel's 2x2 children in the [B,2Ho,2Wo,*] gradient
            const int Wo = pool_sum ? (k.W >> 1) : k.W, Ho = pool_sum ? (k.H >> 1) : k.H;
            const int yy = pool_sum ? (oy0 >> 1) : oy0 + (s >> 1);
            const int xx = pool_sum ? (ox0 >> 1) : ox0 + (s & 1);
            const unsigned W2 = 2u * (unsigned)Wo;
            const unsigned cq = ((unsigned)(b * 2 * Ho + 2 * yy)) * W2 + 2u * (unsigned)xx;
            o += (ld4(k.arb_skip, cq * ld + n) + ld4(k.arb_skip, (cq + 1) * ld + n)) +
                 (ld4(k.arb_skip, (cq + W2) * ld + n) + ld4(k.arb_skip, (cq + W2 + 1) * ld + n));
          } else {
            o += ld4(k.arb_skip, pix * ld + (unsigned)n);
          }
        }
        st4(dst, pix * dld + (unsigned)n, o);
        if (k.amax_out || k.amax_outp) { if (pool_sum) S.amaxp = absmax4(S.amaxp, o); else S.amax = absmax4(S.amax, o); }
        S.sgx += g * xv;
        S.sg += g;
      }
    }
  }
}

// ARB: per-(block, channel) partial sums of g*x and g.  Lanes with equal (lane % C4) hold the
// same 4 channels for different quads; NW = 4 waves took part.  `slot` = row of arb_partial
// ((image * arb_nblk + tile) of the 128-pixel tiling the caller sized the buffer for).
template <int COLS, int C4>
__device__ __forceinline__ void epi_arb_reduce(const ConvK& k, EpiSums& S, float* smem, int wave,
                                               int lane, int tid, size_t slot, int n0,
                                               const EpiSums* S2 = nullptr) {
  // The 8 quads of a wave meet in ONE tree whatever the channel width of the tile: q ^ 1, then q ^ 2,
  // then q ^ 4.  A 32-channel tile (C4 = 8) holds one quad per lane: shuffles 8, 16, 32.  A 64-channel
  // tile (C4 = 16) holds quads q and q + 4 in one lane (S and *S2, kept apart by epilogue_vec): each is
  // reduced over q ^ 1, q ^ 2 (shuffles 16, 32) and the two meet last.  Until round 5 the second quad
  // was added onto the first pixel by pixel -- another order, an ulp apart -- and 32 vs 64 channels is
  // chosen from the GRID, i.e. from the batch: d s / d t of the 64-channel layers and of the sub-pixel
  // gradients depended on who shared the launch.
  f32x4 sgx = S.sgx, sg = S.sg;
  f32x4 sgx2 = {0, 0, 0, 0}, sg2 = {0, 0, 0, 0};
  if (C4 == 16 && S2 != nullptr) { sgx2 = S2->sgx; sg2 = S2->sg; }
#pragma unroll
  for (int o = C4; o < 64; o <<= 1) {
    sgx.x += __shfl_xor(sgx.x, o, 64); sgx.y += __shfl_xor(sgx.y, o, 64);
    sgx.z += __shfl_xor(sgx.z, o, 64); sgx.w += __shfl_xor(sgx.w, o, 64);
    sg.x += __shfl_xor(sg.x, o, 64); sg.y += __shfl_xor(sg.y, o, 64);
    sg.z += __shfl_xor(sg.z, o, 64); sg.w += __shfl_xor(sg.w, o, 64);
    if (C4 == 16 && S2 != nullptr) {
      sgx2.x += __shfl_xor(sgx2.x, o, 64); sgx2.y += __shfl_xor(sgx2.y, o, 64);
      sgx2.z += __shfl_xor(sgx2.z, o, 64); sgx2.w += __shfl_xor(sgx2.w, o, 64);
      sg2.x += __shfl_xor(sg2.x, o, 64); sg2.y += __shfl_xor(sg2.y, o, 64);
      sg2.z += __shfl_xor(sg2.z, o, 64); sg2.w += __shfl_xor(sg2.w, o, 64);
    }
  }
  if (C4 == 16 && S2 != nullptr) { sgx = sgx + sgx2; sg = sg + sg2; }
  __syncthreads();                      // everyone is done reading the tile dumps
  float* red = smem;                    // [2][4 waves][COLS]
  if (lane < C4 && wave < 4) {
    *reinterpret_cast<f32x4*>(red + wave * COLS + lane * 4) = sgx;
    *reinterpret_cast<f32x4*>(red + (4 + wave) * COLS + lane * 4) = sg;
  }
  __syncthreads();
  if (tid < COLS && n0 + tid < k.n_store) {
    const float a = (red[tid] + red[COLS + tid]) + (red[2 * COLS + tid] + red[3 * COLS + tid]);
    const float t = (red[4 * COLS + tid] + red[5 * COLS + tid]) +
                    (red[6 * COLS + tid] + red[7 * COLS + tid]);
    const size_t o = slot * k.Cout + n0 + tid;
    k.arb_partial[o] = a;
    k.arb_partial[(size_t)k.B * k.arb_nblk * k.Cout + o] = t;
  }
}

// Vectorised epilogue.  The MFMA C layout gives a lane one channel of 16 pixels:
// storing that directly is 4-byte accesses, 128 B per pixel and instruction.  Instead
// every wave dumps its 32 x (NT*32) accumulator tile into LDS (free after the K loop)
// and re-reads it so that a lane owns ONE 2x2 pixel quad x FOUR consecutive channels:
// all global accesses are 16 B per lane and NT*128 B contiguous per pixel, 2x2 pooling
// stays lane-local, and the per-channel sums of the fused activation backward reduce
// with 2-3 shuffles.  Handles every epilogue mode of the conv:
//   v = alpha*acc + bias + residual ; act ; mask ; store ; 2x2 max/sum pool, or
//   ARB (input-gradient convs): g = (x*s+t>0) ? da : 0 ; dx = g*s + shortcut ;
//   partial sums of g*x and g per (tile, channel)  [da 2x2-summed first if pool==SUM].
// The second half of epilogue_vec: the dumps [4 pixel groups of 32][COLS + 4] are in `smem` and visible; wave w
// works through pixel group w.  (Separate so that a kernel with another wave -> accumulator mapping --
// p2l_h2r.hip: a wave owns 64 pixels x 32 channels -- can fill the same dumps and share everything behind.)
template <int NT>
__device__ __forceinline__ void epilogue_vec_items(const ConvK& k, float* smem, int wave, int lane, int b0,
                                                   int y0, int x0, int n0, int tile_in_image, int osh,
                                                   int ph_y, int ph_x);
template <int NT>
__device__ __forceinline__ void epilogue_vec(const ConvK& k, const f32x16 (&acc)[NT],
                                             float* smem, int wave, int lane, int b0, int y0,
                                             int x0, int n0, int tile_in_image, int osh,
                                             int ph_y, int ph_x) {
  constexpr int COLS = NT * 32, EP = COLS + 4;
  const int l31 = lane & 31, lhi = lane >> 5;
  float* tb = smem + wave * 32 * EP;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      tb[((r & 3) + 8 * (r >> 2) + 4 * lhi) * EP + j * 32 + l31] = acc[j][r];
  __syncthreads();
  epilogue_vec_items<NT>(k, smem, wave, lane, b0, y0, x0, n0, tile_in_image, osh, ph_y, ph_x);
}
template <int NT>
__device__ __forceinline__ void epilogue_vec_items(const ConvK& k, float* smem, int wave, int lane, int b0,
                                                   int y0, int x0, int n0, int tile_in_image, int osh,
                                                   int ph_y, int ph_x) {
  constexpr int COLS = NT * 32, EP = COLS + 4, C4 = COLS / 4, ITEMS = 8 * C4;
  float* tb = smem + wave * 32 * EP;

  const int TWh = (1 << k.tw_log) >> 1, THh = (1 << k.th_log) >> 1;
  EpiSums S, S2;                           // S2: the second quad of a lane (64-channel tiles), see epi_arb_reduce
#pragma unroll
  for (int it0 = 0; it0 < ITEMS; it0 += 64) {
    EpiSums& Sx = (it0 == 0) ? S : S2;
    const int it = it0 + lane;
    const int q = it / C4, c4 = it - q * C4;
    const int n = n0 + c4 * 4;
    const int Q = wave * 8 + q;
    const int qx = Q & (TWh - 1), qy = (Q >> (k.tw_log - 1)) & (THh - 1);
    const int b = b0 + (Q >> (k.tw_log + k.th_log - 2));
    if (b >= k.B || n >= k.n_store) continue;
    const int oy0 = y0 + 2 * qy, ox0 = x0 + 2 * qx;
    f32x4 v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
      v[s] = *reinterpret_cast<const f32x4*>(tb + (4 * q + s) * EP + c4 * 4) * k.alpha;
    epi_item(k, v, b, oy0, ox0, n, osh, ph_y, ph_x, Sx);
  }
  static_assert(ITEMS == 64 || ITEMS == 128, "one or two items per lane");
  if (k.arb_x != nullptr)
    epi_arb_reduce<COLS, C4>(k, S, smem, wave, lane, threadIdx.x,
                             (size_t)b0 * k.arb_nblk + tile_in_image, n0, &S2);
  S.amax = fmaxf(S.amax, S2.amax); S.amaxp = fmaxf(S.amaxp, S2.amaxp);
  // this block's partial maxima of what it stored (one per wave), for the launch that reads the
  // tensor next (P2LAmax; the launcher only sets the pointers for tiles inside one image)
  if (k.amax_out != nullptr || k.amax_outp != nullptr) {
    float m = S.amax, mp = S.amaxp;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); mp = fmaxf(mp, __shfl_xor(mp, o, 64)); }
    if (lane == 0) {                                   // one partial per WAVE: no block reduction
      const int nph = osh ? 4 : 1;
      const size_t slot = (size_t)b0 * k.amax_out_n +
                          (((size_t)tile_in_image * k.n_ntiles + n0 / COLS) * nph + (osh ? ph_y * 2 + ph_x : 0)) * 4 + wave;
      if (k.amax_out != nullptr) k.amax_out[slot] = m;
      if (k.amax_outp != nullptr) k.amax_outp[slot] = mp;
    }
  }
}


}  // namespace p2lconv

// bf16x3 form of the 1x1 conv (p2l_pw.hip)
int p2l_pw_launch(const p2lconv::ConvK& k, int pro, hipStream_t st);

// 3-channel image ends of the pipeline (p2l_thin.hip)
size_t p2l_thin_weight_floats(int N_pad, int K_pad);
int p2l_thin_mode(int N_pad, int K_pad);
int p2l_thin_pack(const float* w_oihw, int O, int I, int N_pad, int K_pad, int flip, float* dst,
                  hipStream_t st);
int p2l_thinin_launch(const p2lconv::ConvK& k, int pro, hipStream_t st);
int p2l_thinout_launch(const p2lconv::ConvK& k, int pro, hipStream_t st);

// Winograd F(2x2,3x3) form of the bf16x3 3x3 conv (p2l_wino.hip)
extern "C" size_t p2l_wino_weight_floats(int N_pad, int K_pad);
extern "C" int p2l_wino_weight_ok(int N_pad, int K_pad);
extern "C" int p2l_wino_split_factor(int H, int W, int Cin, int Cout);
extern "C" size_t p2l_wino_h2_weight_floats(int N_pad, int K_pad);
extern "C" size_t p2l_pw_h2_weight_floats(int N_pad, int K_pad);
int p2l_pw_pack_h2(const float* w_oihw, int O, int I, int N_pad, int K_pad, int flip, float* dst, hipStream_t st);
int p2l_wino_pack(const float* w_oihw, int O, int I, int N_pad, int K_pad, int transpose_flip,
                  float* dst, hipStream_t st);
int p2l_wino_launch(const p2lconv::ConvK& k, int pro, hipStream_t st);
// max |x| of every image (the fused prologue applied): 64 partial maxima each -> k.amax[B][64];
// the extents of the INPUT tensor in k.H, k.W, k.Cin, k.x_ld (p2l_wino.hip)
int p2l_amax_launch(const p2lconv::ConvK& k, int pro, hipStream_t st);

// fp16 x 2 form of the direct 3x3 / sub-pixel kernel (p2l_h2.hip)
size_t p2l_h2_weight_floats(int N_pad, int K_pad, int subpix);
int p2l_h2_pack(const float* w_oihw, int O, int I, int N_pad, int K_pad, int flip, int mode, float* dst,
                hipStream_t st);
int p2l_h2_launch(const p2lconv::ConvK& k, int pro, int taps, int bn, bool small, hipStream_t st);

// ... with the weights of a 64 -> 64 channel layer RESIDENT IN REGISTERS (p2l_h2r.hip): persistent blocks,
// one per CU, that walk over 128-pixel tiles; bit-identical to conv_h2_kernel<9, ..>
bool p2l_h2r_shape(int taps, int ups, int H, int W, int Cin, int Cout, int x_ld);
bool p2l_h2r_takes(const p2lconv::ConvK& k);
int p2l_h2r_launch(const p2lconv::ConvK& k, int pro, hipStream_t st);
