// Direct 3x3 / sub-pixel convolution in the fp16 x 2 arithmetic (include/p2l.h): the kernel
// of every 3x3 launch the Winograd form does not take -- the nearest-x2 up-convs of the BigGAN-deep
// GenBlocks and the stride-2 transposed convs of StyleGAN2 (sub-pixel forward, 4 output phases, and
// their input gradient), the 4^2 ... 16^2 layers in split-K slices, shapes whose H / W are not
// multiples of 16 (reached from pix2latent/model/biggan.py:58, pix2latent/model/stylegan2.py:116-125
// and pix2latent/loss_functions.py:142 in the reference).
//
// Same role, same tiles, same epilogue as conv_mfma_kernel<TAPS = 9 | 4, .., BF3> (p2l_conv.hip);
// what changes is the operand representation.  bf16 x 3 splits an fp32 value into three 8-bit
// pieces and needs 6 MFMAs per 16-channel product; fp16 has 11 significant bits, so TWO
// round-to-nearest pieces carry the value to 1 ulp of fp32 once it is scaled by a power of two into
// fp16's exponent range -- the weights per LAYER at pack time, the activations per IMAGE (never per
// batch: a candidate's bits must not depend on who shares its launch) from the maxima the producer
// of the tensor handed over (P2LAmax) or from a max-|x| pass in front of the launch:
//   * 3 x v_mfma_f32_32x32x16_f16 per (tap, 32x32 tile, 16 channels): h h, h m, m h (the dropped
//     m m is <= 2^-22 of the product);
//   * the split of four values is cvt_pk | cvt back | sub | cvt_pk instead of 22 VALU;
//   * LDS rows of 64 B: [h k0-7 | h k8-15 | m k0-7 | m k8-15], the 16-byte chunk index XOR bits 2-3
//     of the row.  A ds_read_b128 lane group ({0-3,12-15,20-27}, ...) is conflict free when its 16
//     rows differ mod 16: true for the weight rows (consecutive) and, in the 2x2-quad pixel order,
//     for patch lines 24 rows apart (16-wide tiles: the bf16 x 3 pitch already) -- tools/h2_banks.py;
//   * the weight tile (an image of the LDS tile, 4 bytes per weight like fp32) arrives by LDS-direct
//     DMA in two halves, exactly as in the bf16 x 3 kernel: 48 KB instead of 72 KB per tile.
// Tiles may span several images at 4^2 / 8^2 (128 pixels = 2 ... 8 images): every image keeps its
// own scale (input rows scaled while they are staged, accumulator rows un-scaled before the epilogue).
#include "p2l_conv_k.h"

#include <atomic>
#include <type_traits>

using namespace p2lconv;

namespace {

// (A/B builds, tools/ab_build.sh p2l_h2 -DP2L_H2_ABL=n: timing ablations of the main loop -- results are
//  wrong when set: 1 no split / LDS write of the activation tile, 2 no activation loads, 4 no weight DMA,
//  8 no MFMAs, 16 no barriers; all after the first chunk; 32 no epilogue)
#ifndef P2L_H2_ABL
#define P2L_H2_ABL 0
#endif

// position of logical 16-byte chunk c (0, 1 = h; 2, 3 = m) inside 64-byte LDS row `row`.  Bits 2-3 of the
// row make the 16 rows of a ds_read_b128 lane group hit 16 different bank windows; bit 1 swaps the h / m
// halves of rows 2, 3 (mod 4) so that the 16 lanes of a ds_write_b64 (4 consecutive rows x 4 channel
// quarters of ONE piece) cover all 32 banks instead of 16 twice (PMC: 6.4 % of LDS cycles were conflicts,
// all of them these writes)
__device__ __forceinline__ int h2c(int c, int row) { return c ^ ((row >> 2) & 3) ^ (((row >> 1) & 1) << 1); }

// s_waitcnt immediate (gfx9 encoding): vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14
#define P2L_WAIT(VM, LGKM) __builtin_amdgcn_s_waitcnt(((VM) & 15) | (7 << 4) | ((LGKM) << 8) | (((VM) >> 4) << 14))

// SKIP (TAPS = 4 only): the transposed-conv form that leaves out its structurally zero taps -- an instantiation of
// its own: as a run-time flag the test sat in every unit of BigGAN's sub-pixel launches too (+ 5 % on them)
// TAPS = 8: the sub-pixel FORWARD form with TWO output phases per block ((ph_y, 0) and (ph_y, 1): 2 x 4 window taps on
// one staged patch, two sets of accumulators, the epilogue once per phase; blockIdx.y = ph_y).  With a phase per block
// (TAPS = 4) four blocks split and wrote the same patch: these launches are bound by staging and the global -> LDS
// path, not by the matrix pipe.  Same products in the same order per output: bit-identical (P2L_FORM_NO_SP_PAIR).
template <int TAPS, int BN, int A_ITERS, int PRO, bool SKIP = false>
__global__ __launch_bounds__(256, 2) void conv_h2_kernel(const ConvK k) {
  constexpr bool SUBPIX = (TAPS == 4 || TAPS == 8);
  constexpr int NP = (TAPS == 8) ? 2 : 1;            // output phases per block
  constexpr int NT = BN / 32;                        // accumulators per wave and phase
  constexpr int B_ITEMS = TAPS * BN * 4;             // 16-byte items of the weight tile
  constexpr int B_ITERS = (B_ITEMS + 255) / 256;
  constexpr int SL = SUBPIX ? 16 : TAPS;             // slabs per (chunk, 32-channel tile) of the image
  constexpr int T0 = (TAPS == 4) ? 2 : 4;            // taps in half 0 of the weight tile (TAPS = 8: phase 0)
  constexpr int NU = TAPS * NT, U0 = T0 * NT;        // MFMA units (tap-major), units in half 0
  constexpr int H0 = U0 * 128;                       // 16-byte items in half 0
  static_assert(H0 % 256 == 0, "half 0 must be whole DMA instructions");
  constexpr int NH1_MIN = (B_ITEMS - H0) / 256;      // half-1 DMA instructions every wave issues
  constexpr bool S_UNI = (A_ITERS == 3);             // one image per tile (launcher)
  constexpr int S_ITERS = S_UNI ? 1 : A_ITERS;
  constexpr int NA_LD = A_ITERS + ((PRO != P2L_PRO_NONE) ? 2 * S_ITERS : 0);

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int TW = 1 << k.tw_log, TH = 1 << k.th_log, TB = 1 << k.tb_log;
  const int HW_ = TW + 2, HH_ = TH + 2;
  const int HP = k.hp;                                       // LDS pitch of a patch line (rows)
  const int a_rows = TB * HH_ * HW_;                         // staged pixels
  const int a_rows_lds = TB * HH_ * HP;                      // LDS rows they occupy
  char* As = reinterpret_cast<char*>(smem);
  char* Bs = As + (size_t)a_rows_lds * 64;
  // per-image scales behind everything else the block keeps in LDS (tile or epilogue dumps)
  const int main_floats = max(a_rows_lds * 16 + TAPS * BN * 16, 4 * 32 * (BN + 4));
  float* scl = smem + main_floats;                           // [TB] x {x scale, output un-scale}

  // ---- which tile -------------------------------------------------------
  const int swz = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = swz / k.n_ntiles, nt = swz - mt * k.n_ntiles;
  const int tiles_per_image = k.tiles_x * k.tiles_y;
  const int bt = mt / tiles_per_image;
  const int tile_in_image = mt - bt * tiles_per_image;
  const int ty = tile_in_image / k.tiles_x, tx = tile_in_image - ty * k.tiles_x;
  const int n0 = nt * BN;
  const int y0 = ty << k.th_log, x0 = tx << k.tw_log, b0 = bt << k.tb_log;

  // blockIdx.y: split-K slice, or (sub-pixel forward) the output phase
  const bool sp_fwd = SUBPIX && (TAPS == 8 || k.sp_mode == 1);
  const bool sp_bwd = (TAPS == 4) && k.sp_mode == 2;
  const int z = sp_fwd ? 0 : blockIdx.y;
  const int ph_y = sp_fwd ? (TAPS == 8 ? (int)blockIdx.y : (int)(blockIdx.y >> 1)) : 0;
  const int ph_x = (sp_fwd && TAPS == 4) ? (int)(blockIdx.y & 1) : 0;          // (TAPS = 8: both, tap >> 2)
  const int c_begin = z * k.chunks_per_split;
  const int c_end = min(k.nchunks, c_begin + k.chunks_per_split);

  // ---- per-thread staging descriptors (fixed across chunks) --------------
  // item j = tid + 256 it: pixel j >> 2, channel quarter tid & 3.  Loads are issued unconditionally
  // from a clamped (always valid) address and zeroed at LDS-write time.
  const int av = tid & 3;
  int a_goff[A_ITERS];   // float offset of the source pixel's chunk-0 vector
  int a_soff[A_ITERS];   // float offset into pro_s / pro_t
  int a_row[A_ITERS];    // LDS row, -1 = no item
  float a_xs[S_ITERS];   // the item's image scale (read once the scales exist, below)
  int a_tb[S_ITERS];     // ... and which image of the tile the item belongs to
  unsigned a_valid = 0;  // bit it: source pixel exists (else zero padding)
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int p = (tid + 256 * it) >> 2;
    a_goff[it] = 0;
    a_soff[it] = 0;
    a_row[it] = -1;
    if (it < S_ITERS) { a_xs[it] = 1.f; a_tb[it] = 0; }
    if (p < a_rows) {
      const int tb = p / (HH_ * HW_);
      const int rem = p - tb * (HH_ * HW_);
      const int hy = rem / HW_, hx = rem - hy * HW_;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      a_row[it] = (tb * HH_ + hy) * HP + hx;
      const int b = b0 + tb;
      if (!S_UNI) a_tb[it] = tb;
      if (b < k.B && iy >= 0 && iy < k.iH && ix >= 0 && ix < k.iW) {
        int pix;
        if (sp_bwd)            // phase plane (0,0) of the high-res gradient buffer
          pix = (b * k.ibH + 2 * iy) * k.ibW + 2 * ix;
        else
          pix = (b * k.ibH + iy) * k.ibW + ix;
        a_goff[it] = pix * k.x_ld + av * 4;
        a_soff[it] = b * k.pro_bstride + av * 4;
        a_valid |= 1u << it;
      }
    }
  }
  const int s_uni = b0 * k.pro_bstride + av * 4;
  f32x4 xr[A_ITERS], sr[S_ITERS], tr[S_ITERS];

  // chunk c -> (channel chunk cc, first weight slab, extra input offset): the sub-pixel
  // input-gradient walks over (phase plane cls, channel chunk cc)
  auto geom = [&](int c, int& cc, int& wslab, int& a_extra) {
    cc = c; wslab = 0; a_extra = 0;
    if (SUBPIX) {
      if (sp_bwd) {
        const int cls = c / k.sp_ncc;
        cc = c - cls * k.sp_ncc;
        wslab = cls * 4;
        a_extra = ((cls >> 1) * k.ibW + (cls & 1)) * k.x_ld;
      } else {
        wslab = (ph_y * 2 + ph_x) * 4;               // (TAPS = 8: ph_x = 0, the slabs of both phases follow)
      }
    }
  };
  // LDS order of the weight tile: [tap][n-tile j][32 rows] (= MFMA unit order)
  int b_goff[B_ITERS];
#pragma unroll
  for (int it = 0; it < B_ITERS; ++it) {
    const int jj = tid + 256 * it;
    const int u = jj >> 7, within = jj & 127;
    const int tap = u / NT, j = u - tap * NT;
    b_goff[it] = ((j * SL + tap) * 32) * 16 + within * 4;
  }
  auto dma_b = [&](int c, auto half_c) {
    constexpr int half = decltype(half_c)::value;
    int cc, wslab, a_extra;
    geom(c, cc, wslab, a_extra);
    const float* base = k.w + ((((size_t)cc * (k.Cout >> 5) + (n0 >> 5)) * SL + wslab) * 32) * 16;
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      if ((half == 0 && 256 * it >= H0) || (half == 1 && 256 * (it + 1) <= H0)) continue;
      const int jj = tid + 256 * it;
      if (jj < B_ITEMS) {
        // (inline asm: through the builtin hipcc drains vmcnt in front of the next LDS access;
        //  the waits are placed by hand below, see p2l_conv.hip)
        const float* src = base + b_goff[it];
        if (SUBPIX && SKIP) {
          // a dead tap's slab is all zeros and its products are skipped: the DMA instruction still issues (the
          // waits count instructions) but every lane fetches the SAME 16 bytes -- one L2 request per wave
          // instead of 1 KB: 7 / 16 of the weight traffic of a transposed conv gone
          unsigned lv;
          if (TAPS == 8) { const unsigned my = ph_y ? 0x3u : 0xFu; lv = my | ((my & 0x5u) << 4); }
          else if (sp_fwd) lv = (ph_y ? 0x3u : 0xFu) & (ph_x ? 0x5u : 0xFu);
          else { const int cls = c / k.sp_ncc; lv = ((cls >> 1) ? 0xCu : 0xFu) & ((cls & 1) ? 0xAu : 0xFu); }
          const int tap = (jj >> 7) / NT;                       // (wave-uniform: 64 consecutive items)
          if (!((lv >> tap) & 1u)) src = base;
        }
        const unsigned lds_wave_base = __builtin_amdgcn_readfirstlane(
            (unsigned)(size_t)(__attribute__((address_space(3))) char*)(Bs + (size_t)(jj - lane) * 16));
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(src), "s"(lds_wave_base) : "memory");
      }
    }
  };
  auto load_a = [&](int c) {
    int cc, wslab, a_extra;
    geom(c, cc, wslab, a_extra);
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      xr[it] = *reinterpret_cast<const f32x4*>(k.x + (size_t)(a_goff[it] + a_extra) + cc * 16);
      if (PRO != P2L_PRO_NONE && it < S_ITERS) {
        const int so = S_UNI ? s_uni : a_soff[it];
        sr[it] = *reinterpret_cast<const f32x4*>(k.pro_s + so + cc * 16);
        tr[it] = *reinterpret_cast<const f32x4*>(k.pro_t + so + cc * 16);
      }
    }
  };
  auto write_a = [&]() {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      if (a_row[it] >= 0) {
        f32x4 v = xr[it];
        if (PRO != P2L_PRO_NONE) {
          v = v * sr[S_UNI ? 0 : it] + tr[S_UNI ? 0 : it];
          if (PRO == P2L_PRO_AFFINE_RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          }
        }
        if (!((a_valid >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        v = v * a_xs[S_UNI ? 0 : it];
        const h16x4 h = __builtin_convertvector(v, h16x4);
        const f32x4 w = __builtin_convertvector(h, f32x4);
        const h16x4 m = __builtin_convertvector(v - w, h16x4);
        const int row = a_row[it];
        char* rb = As + row * 64 + (av & 1) * 8;
        *reinterpret_cast<h16x4*>(rb + h2c(av >> 1, row) * 16) = h;
        *reinterpret_cast<h16x4*>(rb + h2c(2 + (av >> 1), row) * 16) = m;
      }
    }
  };

  // ---- fragment addressing ----------------------------------------------
  int a_row0;
  {
    const int i = wave * 32 + l31;
    const int Q = i >> 2, s = i & 3;
    const int qx = Q & ((TW >> 1) - 1);
    const int qy = (Q >> (k.tw_log - 1)) & ((TH >> 1) - 1);
    const int tb = Q >> (k.tw_log + k.th_log - 2);
    a_row0 = (tb * HH_ + 2 * qy + (s >> 1)) * HP + 2 * qx + (s & 1);
  }
  // weight rows are u * 32 + l31: the swizzle bits come from l31 alone
  const int b_h = l31 * 64 + h2c(lhi, l31) * 16, b_m = l31 * 64 + h2c(2 + lhi, l31) * 16;

  f32x16 acc[NP * NT];
#pragma unroll
  for (int j = 0; j < NP * NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  using H0c = std::integral_constant<int, 0>;
  using H1c = std::integral_constant<int, 1>;
  // Pipeline of the bf16 x 3 kernel: half 0 of the weight tile of chunk c+1 is fetched while the
  // taps of half 1 of chunk c are multiplied, half 1 while the activation tile is split and
  // written; the activation loads of chunk c+2 are issued right after write_a() has consumed
  // those of chunk c+1.
  if (c_begin < c_end) {
    dma_b(c_begin, H0c{});
    dma_b(c_begin, H1c{});
    load_a(c_begin);
  }
  // (the scales are only needed when the first tile is WRITTEN: the first weight tile and patch are
  //  in flight while the maxima are reduced)
  // max |x| of image b: the partial maxima the producer of the tensor left (P2LAmax; a fused
  // prologue x*s+t is then bounded by max|s| max|x| + max|t|), or the 64 partials of the pass in
  // front of this launch (the prologue applied there)
  {
    float sw, inv_w;
    h2_scales(__builtin_amdgcn_readfirstlane(k.w_tail[0]), sw, inv_w);
    // one image per tile: the whole block reduces; several: one WAVE per image (shuffles only)
    const int step = (TB == 1) ? 256 : 64, me = (TB == 1) ? tid : lane;
    for (int t = (TB == 1) ? 0 : wave; t < TB; t += 4) {
      const int b = b0 + t;
      float a = 0.f, ms = 0.f, mtt = 0.f;
      if (b < k.B) {
        if (k.amax_in != nullptr) {
          for (int i = me; i < k.amax_in_n; i += step) a = fmaxf(a, k.amax_in[(size_t)b * k.amax_in_n + i]);
          if (PRO != P2L_PRO_NONE && !k.amax_in_applied) {   // (applied: the maxima ARE those of x*s+t)
            const f32x4* ps = reinterpret_cast<const f32x4*>(k.pro_s + (size_t)b * k.pro_bstride);
            const f32x4* pt = reinterpret_cast<const f32x4*>(k.pro_t + (size_t)b * k.pro_bstride);
            for (int c = me; c < (k.Cin >> 2); c += step) {
              const f32x4 s4 = ps[c], t4 = pt[c];
              ms = fmaxf(fmaxf(ms, fmaxf(fabsf(s4.x), fabsf(s4.y))), fmaxf(fabsf(s4.z), fabsf(s4.w)));
              mtt = fmaxf(fmaxf(mtt, fmaxf(fabsf(t4.x), fabsf(t4.y))), fmaxf(fabsf(t4.z), fabsf(t4.w)));
            }
          }
        } else if (me < 64) {
          a = k.amax[b * 64 + me];
        }
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        a = fmaxf(a, __shfl_xor(a, o, 64));
        if (PRO != P2L_PRO_NONE) { ms = fmaxf(ms, __shfl_xor(ms, o, 64)); mtt = fmaxf(mtt, __shfl_xor(mtt, o, 64)); }
      }
      if (TB == 1) {
        if (lane == 0) { smem[wave * 4] = a; smem[wave * 4 + 1] = ms; smem[wave * 4 + 2] = mtt; }
        __syncthreads();
        a = fmaxf(fmaxf(smem[0], smem[4]), fmaxf(smem[8], smem[12]));
        ms = fmaxf(fmaxf(smem[1], smem[5]), fmaxf(smem[9], smem[13]));
        mtt = fmaxf(fmaxf(smem[2], smem[6]), fmaxf(smem[10], smem[14]));
      }
      if (PRO != P2L_PRO_NONE && k.amax_in != nullptr) a = (k.amax_in_applied ? a : ms * a + mtt) * 1.001f;
      float xs, inv_x;
      h2_scales(__builtin_bit_cast(unsigned, a), xs, inv_x);
      if ((TB == 1) ? (tid == 0) : (lane == 0)) { scl[2 * t] = xs; scl[2 * t + 1] = inv_x * inv_w; }
      if (TB == 1) break;
    }
    __syncthreads();                                   // (scales visible; the scratch becomes the tile)
  }

  if (S_UNI) {
    a_xs[0] = scl[0];
  } else {
#pragma unroll
    for (int it = 0; it < S_ITERS; ++it) a_xs[it] = scl[2 * a_tb[it]];
  }
  if (c_begin < c_end) {
    write_a();
    __builtin_amdgcn_sched_barrier(0);
    if (c_begin + 1 < c_end) load_a(c_begin + 1);
  }
  __builtin_amdgcn_sched_barrier(0);
  P2L_WAIT(NA_LD, 0);                   // everything but the loads just issued
  __builtin_amdgcn_s_barrier();

  for (int c = c_begin; c < c_end; ++c) {
    const bool more = (c + 1 < c_end);
    int win_row = 0;
    if (SUBPIX) {
      int oy = ph_y, ox = ph_x;
      if (sp_bwd) {
        const int cls = c / k.sp_ncc;
        oy = 1 - (cls >> 1);
        ox = 1 - (cls & 1);
      }
      win_row = oy * HP + ox;
    }
    // Stride-2 transposed conv (StyleGAN2's up convs, k.sp_skip): per dimension an output of phase 1 has ONE source
    // pixel, an input-gradient plane 1 one too -- pack_subpix_kernel (mode 1) leaves the other window tap's slab
    // zero: 9 live taps of the 16 of a 2x2 quad of phases.  Their products are skipped (adding exact zeros changes
    // no bit); the zero slabs still travel with the weight tile (the vmcnt bookkeeping counts DMA instructions).
    // Bit t of `live`: tap t = (ty, tx) of this block's phase (forward) / of this chunk's phase plane (gradient).
    unsigned live = 0xFFu;
    if (SUBPIX && SKIP) {
      if (TAPS == 8) {
        const unsigned my = ph_y ? 0x3u : 0xFu;
        live = my | ((my & 0x5u) << 4);                              // taps 4 .. 7: phase (ph_y, 1)
      } else if (sp_fwd) {
        live = (ph_y ? 0x3u : 0xFu) & (ph_x ? 0x5u : 0xFu);          // phase 1: window tap 0 only
      } else {
        const int cls = c / k.sp_ncc;
        live = ((cls >> 1) ? 0xCu : 0xFu) & ((cls & 1) ? 0xAu : 0xFu);   // plane 1: window tap 1 only
      }
    }
    h16x8 af[2][2], bq[2][2];
    auto lda = [&](int tap, h16x8 (&a)[2]) {
      const int dy = (TAPS == 9) ? tap / 3 : ((tap & 3) >> 1);
      const int dx = (TAPS == 9) ? tap - dy * 3 : (tap & 1);
      const int arow = a_row0 + win_row + dy * HP + dx + ((TAPS == 8) ? (tap >> 2) : 0);   // (+ the phase's window column)
      const char* ar = As + arow * 64;
      a[0] = *reinterpret_cast<const h16x8*>(ar + h2c(lhi, arow) * 16);
      a[1] = *reinterpret_cast<const h16x8*>(ar + h2c(2 + lhi, arow) * 16);
    };
    auto ldb = [&](int u, h16x8 (&b)[2]) {
      const char* br = Bs + u * (32 * 64);
      b[0] = *reinterpret_cast<const h16x8*>(br + b_h);
      b[1] = *reinterpret_cast<const h16x8*>(br + b_m);
    };
    lda(0, af[0]);
    ldb(0, bq[0]);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int tap = u / NT, j = u - tap * NT;
      if (u == U0) {
        // half 1 of THIS chunk (DMA'd at the end of the previous one, the youngest VMEM op) must
        // have landed in every wave, and every wave must be done reading half 0 before it is
        // refilled
        // (the activation loads of chunk c+1 were issued AFTER that DMA and stay in flight: with
        //  three MFMAs per unit half a chunk is ~770 cycles, less than a global round trip)
        if (more) P2L_WAIT(NA_LD, 0); else P2L_WAIT(0, 0);
        if (!(P2L_H2_ABL & 16)) __builtin_amdgcn_s_barrier();
        if (more && !(P2L_H2_ABL & 4)) dma_b(c + 1, H0c{});
        if (j == 0) lda(tap, af[tap & 1]);
        ldb(u, bq[u & 1]);
      }
      if (u + 1 < NU && u + 1 != U0) {
        const int tn = (u + 1) / NT, jn = (u + 1) - tn * NT;
        if (!(SUBPIX && SKIP) || ((live >> tn) & 1u)) {
          if (jn == 0) lda(tn, af[tn & 1]);
          ldb(u + 1, bq[(u + 1) & 1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      const h16x8 (&a)[2] = af[tap & 1];
      const h16x8 (&b)[2] = bq[u & 1];
      if (SUBPIX && SKIP && !((live >> tap) & 1u)) continue;        // (uniform: a scalar branch over the three MFMAs)
      const int ja = (TAPS == 8) ? (tap >> 2) * NT + j : j;            // (compile-time: the loop is unrolled)
      if (!(P2L_H2_ABL & 8)) {
      acc[ja] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc[ja], 0, 0, 0);   // smallest terms first
      acc[ja] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc[ja], 0, 0, 0);
      acc[ja] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc[ja], 0, 0, 0);
      } else { acc[ja][0] += (float)a[0][0] * (float)b[0][0]; }
      __builtin_amdgcn_sched_barrier(0);
    }
    P2L_WAIT(63, 0);                      // my LDS reads are done
    if (!(P2L_H2_ABL & 16)) __builtin_amdgcn_s_barrier();         // everybody is done with half 1 and the A tile
    if (more) {
      if (!(P2L_H2_ABL & 1)) write_a();
      __builtin_amdgcn_sched_barrier(0);
      const bool more2 = (c + 2 < c_end);
      if (!(P2L_H2_ABL & 4)) dma_b(c + 1, H1c{});                // (BEFORE the activation loads: the mid-chunk wait for it
      __builtin_amdgcn_sched_barrier(0);  //  then leaves those loads in flight)
      if (more2 && !(P2L_H2_ABL & 2)) load_a(c + 2);
      __builtin_amdgcn_sched_barrier(0);
      // A tile written (lgkmcnt 0) and half 0 of c+1 landed; the ops issued after it - the half-1
      // DMAs and the activation loads of c+2 (if any) - may still be in flight
      if (more2) P2L_WAIT(NA_LD + NH1_MIN, 0); else P2L_WAIT(NH1_MIN, 0);
    } else {
      P2L_WAIT(63, 0);
    }
    if (!(P2L_H2_ABL & 16)) __builtin_amdgcn_s_barrier();
  }

  // ---- un-scale: accumulator row r of a lane belongs to pixel wave*32 + (r&3) + 8(r>>2) + 4 lhi ----
  if (S_UNI) {
    const float os = scl[1];
#pragma unroll
    for (int j = 0; j < NP * NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] *= os;          // (exact: a power of two)
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {                             // registers 4g .. 4g+3 = one quad
      const int Q = wave * 8 + 2 * g + lhi;
      const float os = scl[2 * (Q >> (k.tw_log + k.th_log - 2)) + 1];
#pragma unroll
      for (int j = 0; j < NP * NT; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[j][4 * g + s] *= os;
    }
  }

  // ---- epilogue -----------------------------------------------------------
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
      for (int j = 0; j < NT; ++j) {
        float* wp = k.ws + ((size_t)z * mtot + pix0) * k.Cout + n0 + j * 32 + l31;
        wp[0] = acc[j][g * 4 + 0];
        wp[k.Cout] = acc[j][g * 4 + 1];
        wp[(size_t)k.W * k.Cout] = acc[j][g * 4 + 2];
        wp[(size_t)(k.W + 1) * k.Cout] = acc[j][g * 4 + 3];
      }
    }
  } else {
    __syncthreads();                      // (scl read above by every wave before the dumps start)
    if (!(P2L_H2_ABL & 32) || acc[0][0] == 12345.678f) {
      if (TAPS == 8) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          if (p) __syncthreads();             // (the dumps of the first phase have been read)
          epilogue_vec<NT>(k, reinterpret_cast<f32x16 (&)[NT]>(acc[p * NT]), smem, wave, lane, b0, y0, x0, n0,
                           tile_in_image, 1, ph_y, p);
        }
      } else {
        epilogue_vec<NT>(k, reinterpret_cast<f32x16 (&)[NT]>(acc[0]), smem, wave, lane, b0, y0, x0, n0, tile_in_image,
                         sp_fwd ? 1 : 0, ph_y, ph_x);
      }
    }
  }
}
#undef P2L_WAIT

// ---- weights: the fp16 x 2 image of the LDS weight tile ------------------------------------
// [16-channel chunk q][32-channel tile n/32][slab (tap, or phase*4+tap)][row n%32][64 bytes], row =
// [h k0-7 | h k8-15 | m k0-7 | m k8-15] with the 16-byte chunk index already swizzled (h2c); scaled
// by the layer's power of two; tail[0] = bits of the bound on |slab value| the scale was taken from
__global__ void h2_wmax_kernel(const float* w, size_t n, unsigned* tail) {
  float mx = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    mx = fmaxf(mx, fabsf(w[i]));
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(tail, __builtin_bit_cast(unsigned, mx));   // (>= 0: bit order = value order)
}
// sub-pixel slabs are sums of up to 4 taps: the bound is 4 max |w| (an exact power-of-two factor)
__global__ void h2_tail_x4_kernel(unsigned* tail) {
  const float m = __builtin_bit_cast(float, tail[0]);
  tail[0] = __builtin_bit_cast(unsigned, m > 0.f && m < 1e37f ? 4.f * m : m);
}

// which 3x3 taps a sub-pixel slab sums (p2l_conv.hip sp_tapset; mode 0 = conv on a nearest-x2
// upsampled input, 1 = stride-2 transposed conv)
__device__ __forceinline__ unsigned h2_sp_tapset(int mode, int flip, int p, int i) {
  if (mode == 0) {
    if (!flip) {
      if (p == 0) return i == 0 ? 1u : 6u;
      return i == 0 ? 3u : 4u;
    }
    const int u = 2 * i - p;
    return u == 2 ? 1u : u == 0 ? 6u : u == 1 ? 3u : 4u;
  }
  if (!flip) {
    if (p == 0) return i == 0 ? 4u : 1u;
    return i == 0 ? 2u : 0u;
  }
  if (p == 0) return i == 0 ? 1u : 4u;
  return i == 0 ? 0u : 2u;
}

// mode < 0: the 9 taps of a plain 3x3 conv (flip = input-gradient form: taps mirrored, channels
// swapped); mode 0 / 1: the 16 phase-tap slabs of the sub-pixel forms
__global__ __launch_bounds__(256) void h2_pack_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                      int O, int I, int N_pad, int K_pad, int flip,
                                                      int mode, const unsigned* tail) {
  const int n_slabs = mode < 0 ? 9 : 16;
  // one thread per (chunk q, tile, slab, row, k half): 8 values -> two 16-byte pieces
  const size_t total = (size_t)(K_pad >> 4) * (N_pad >> 5) * n_slabs * 32 * 2;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  float scale, inv;
  h2_scales(tail[0], scale, inv);
  const int kh = (int)(idx & 1);
  size_t r = idx >> 1;
  const int row = (int)(r & 31); r >>= 5;
  const int slab = (int)(r % n_slabs); r /= n_slabs;
  const int tile = (int)(r % (N_pad >> 5));
  const int q = (int)(r / (N_pad >> 5));
  const int n = tile * 32 + row;
  const bool n_ok = flip ? (n < I) : (n < O);
  h16x8 ph, pm;
  for (int e = 0; e < 8; ++e) {
    const int c = q * 16 + kh * 8 + e;
    const bool ok = n_ok && (flip ? (c < O) : (c < I));
    float v = 0.f;
    if (ok) {
      if (mode < 0) {
        v = flip ? src[((size_t)c * I + n) * 9 + (8 - slab)] : src[((size_t)n * I + c) * 9 + slab];
      } else {
        const int ph_ = slab >> 2, tap = slab & 3;
        const unsigned my = h2_sp_tapset(mode, flip, ph_ >> 1, tap >> 1);
        const unsigned mx = h2_sp_tapset(mode, flip, ph_ & 1, tap & 1);
        for (int dy = 0; dy < 3; ++dy)
          for (int dx = 0; dx < 3; ++dx)
            if (((my >> dy) & 1u) && ((mx >> dx) & 1u))
              v += flip ? src[((size_t)c * I + n) * 9 + dy * 3 + dx]
                        : src[((size_t)n * I + c) * 9 + dy * 3 + dx];
      }
    }
    const float x = v * scale;
    const _Float16 h = (_Float16)x;
    ph[e] = h; pm[e] = (_Float16)(x - (float)h);
  }
  char* rb = reinterpret_cast<char*>(dst) +
             ((((size_t)q * (N_pad >> 5) + tile) * n_slabs + slab) * 32 + row) * 64;
  *reinterpret_cast<h16x8*>(rb + h2c(kh, row) * 16) = ph;
  *reinterpret_cast<h16x8*>(rb + h2c(2 + kh, row) * 16) = pm;
}

}  // namespace

// floats of the fp16 x 2 direct image of an N_pad x K_pad 3x3 conv (9 slabs) / of its sub-pixel
// form (16 slabs): 4 bytes per weight, then 4 tail floats
size_t p2l_h2_weight_floats(int N_pad, int K_pad, int subpix) {
  return (size_t)N_pad * K_pad * (subpix ? 16 : 9) + 4;
}
int p2l_h2_pack(const float* w_oihw, int O, int I, int N_pad, int K_pad, int flip, int mode, float* dst,
                hipStream_t st) {
  const int n_slabs = mode < 0 ? 9 : 16;
  unsigned* tail = reinterpret_cast<unsigned*>(dst + (size_t)N_pad * K_pad * n_slabs);
  if (hipMemsetAsync(tail, 0, 16, st) != hipSuccess) return P2L_ELAUNCH;
  const size_t nw = (size_t)O * I * 9;
  hipLaunchKernelGGL(h2_wmax_kernel, dim3((unsigned)(cdiv(nw, 256) < 256 ? cdiv(nw, 256) : 256)), dim3(256), 0, st,
                     w_oihw, nw, tail);
  if (mode >= 0) hipLaunchKernelGGL(h2_tail_x4_kernel, dim3(1), dim3(1), 0, st, tail);
  const size_t total = (size_t)(K_pad >> 4) * (N_pad >> 5) * n_slabs * 32 * 2;
  hipLaunchKernelGGL(h2_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w_oihw, dst, O, I, N_pad, K_pad,
                     flip, mode, tail);
  return p2l_check_launch();
}

// LDS bytes of a launch: patch + weight tile (or the vector epilogue's dumps) + the per-image scales
size_t p2l_h2_lds_bytes(const ConvK& k, int taps, int bn) {
  const int TB = 1 << k.tb_log, TH = 1 << k.th_log;
  const int a_rows_lds = TB * (TH + 2) * k.hp;
  int main_floats = a_rows_lds * 16 + taps * bn * 16;
  const int epi = 4 * 32 * (bn + 4);
  if (epi > main_floats) main_floats = epi;
  return (size_t)(main_floats + 2 * TB + 16) * sizeof(float);
}

// k: the ConvK conv_launch_impl built for the direct / sub-pixel kernel (tile geometry, k.hp, sp_mode,
// nchunks, splitk ...) with k.w = the fp16 x 2 image, k.w_tail, k.amax | k.amax_in set.  taps = 9 | 4.
int p2l_h2_launch(const ConvK& k, int pro, int taps, int bn, bool small, hipStream_t st) {
  dim3 grid(k.n_mtiles * k.n_ntiles, taps == 8 ? 2 : (taps == 4 && k.sp_mode == 1) ? 4 : k.splitk), block(256);
  const size_t lds = p2l_h2_lds_bytes(k, taps, bn);
#define P2L_H2L(TAPS, BNV, AIT, PROV)                                                         \
  do {                                                                                        \
    if ((TAPS == 4 || TAPS == 8) && k.sp_skip) P2L_H2K(TAPS, BNV, AIT, PROV, (TAPS == 4 || TAPS == 8)); \
    else P2L_H2K(TAPS, BNV, AIT, PROV, false);                                                \
  } while (0)
#define P2L_H2K(TAPS, BNV, AIT, PROV, SKIPV)                                                  \
  do {                                                                                        \
    auto kfn = conv_h2_kernel<TAPS, BNV, AIT, PROV, SKIPV>;                                   \
    static std::atomic<bool> attr_set{false};                                                 \
    if (!attr_set) {                                                                          \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                160 * 1024);                                                  \
      attr_set = true;                                                                        \
    }                                                                                         \
    hipLaunchKernelGGL(kfn, grid, block, lds, st, k);                                         \
  } while (0)
#define P2L_H2P(TAPS, BNV, AIT)                                                               \
  do {                                                                                        \
    if (pro == P2L_PRO_NONE) P2L_H2L(TAPS, BNV, AIT, P2L_PRO_NONE);                           \
    else if (pro == P2L_PRO_AFFINE_RELU) P2L_H2L(TAPS, BNV, AIT, P2L_PRO_AFFINE_RELU);        \
    else P2L_H2L(TAPS, BNV, AIT, P2L_PRO_AFFINE);                                             \
  } while (0)
  if (taps == 9) {
    if (bn == 64) { if (small) P2L_H2P(9, 64, 3); else P2L_H2P(9, 64, 5); }
    else          { if (small) P2L_H2P(9, 32, 3); else P2L_H2P(9, 32, 5); }
  } else if (taps == 8) {
    if (bn == 64) { if (small) P2L_H2P(8, 64, 3); else P2L_H2P(8, 64, 5); }
    else          { if (small) P2L_H2P(8, 32, 3); else P2L_H2P(8, 32, 5); }
  } else {
    if (bn == 64) { if (small) P2L_H2P(4, 64, 3); else P2L_H2P(4, 64, 5); }
    else          { if (small) P2L_H2P(4, 32, 3); else P2L_H2P(4, 32, 5); }
  }
#undef P2L_H2P
#undef P2L_H2L
#undef P2L_H2K
  return p2l_check_launch();
}
