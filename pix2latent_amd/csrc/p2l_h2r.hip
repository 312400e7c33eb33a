// Direct 3x3 convolution, 64 -> 64 channels, fp16 x 2 arithmetic, WEIGHTS RESIDENT IN REGISTERS (round 6).
//
// The 64-channel 3x3 layers of the path -- VGG conv1_2 and the last BigGAN GenBlock at 256^2, their input
// gradients, the 128^2 64 -> 64 layers, VGG at 512^2 / 1024^2 behind StyleGAN2 (reached from
// pix2latent/loss_functions.py:142 and pix2latent/model/biggan.py:58 in the reference) -- ran in
// conv_h2_kernel<9, ..> (p2l_h2.hip) at ~3 x both of their floors: with K = 576 a block of
// that kernel is four 16-channel chunks, each with its own 36 KB weight tile through LDS (147 KB of weights per
// 32 KB of output), three barriers and one LDS fragment read per MFMA group.
//
// Here the whole layer's weights never leave the register file.  One 4-wave block per CU (one wave per SIMD,
// 512 registers each) is PERSISTENT: it loads its B fragments once -- wave (mh, nt) keeps the 36 (chunk, tap)
// fragments x 2 fp16 pieces of output channels nt*32 .. nt*32+31: 288 VGPRs -- and then walks over a contiguous
// range of 8x16-pixel tiles:
//   * the tile's 10x18-pixel patch is staged ONCE for all 64 channels (16 lanes = one pixel's 256 bytes: fully
//     coalesced; the chunked kernel fetched every line in four 64-byte passes), split into the two fp16 pieces
//     and written to the other half of a double-buffered LDS patch while the current tile is multiplied;
//   * a wave multiplies 64 pixels x 32 channels: per (chunk, tap) 4 ds_read_b128 of activations feed 6 MFMAs
//     (the chunked kernel: 6 reads, 2 of them weights), no weight DMA, no barrier inside a tile;
//   * the accumulators are dumped in the layout of epilogue_vec; the forward-style epilogues (bias + ReLU [+ mask]
//     [+ 2x2 max pool] + partial maxima: template parameter EPI) are then written out one pixel per step UNDER the
//     multiply stream of the next tile, everything else goes through the item shared with the other conv kernels
//     (p2l_conv_k.h epilogue_vec_items) between two tiles -- and, being slower that way than the chunked kernel,
//     is left to that kernel by the launcher (p2l_h2r_takes).
// The per-output summation order is the chunked kernel's -- chunk, tap, (m h, h m, h h) -- on the same operand
// pieces, so the results are BIT-IDENTICAL to conv_h2_kernel<9, 64 | 32, ..> (tests/test_kernels_gpu.py
// test_register_resident_64ch_kernel_bit_identical); which of the two runs is a function of the layer shape only.
#include "p2l_conv_k.h"

#include <atomic>
#include <type_traits>

using namespace p2lconv;

namespace {

// (A/B builds, tools/ab_build.sh p2l_h2r -DP2L_H2R_ABL=n -mllvm -pragma-unroll-threshold=400000: timing ablations,
//  results are wrong when set: 1 no staging of the next tile (loads, split, LDS writes), 2 no MFMAs, 4 no
//  epilogue items (loads / stores behind the dump), 8 no accumulator dump, 16 no fragment reads, 32 no pipelined
//  epilogue pieces inside the stream)
#ifndef P2L_H2R_ABL
#define P2L_H2R_ABL 0
#endif

__device__ __forceinline__ int h2c(int c, int row) { return c ^ ((row >> 2) & 3) ^ (((row >> 1) & 1) << 1); }

constexpr int HP = 24, HH = 10, HW = 18;              // patch: 10 lines of 18 pixels, 24 LDS rows per line
constexpr int ROWS = HH * HP;                          // 240 rows of 64 B per 16-channel plane
// (+ 32: the four planes of a pixel -- written by the 16 lanes of one ds_write_b64 lane group -- start 8 banks
//  apart; at a multiple of 128 bytes they met in the same banks: 28 % of the LDS cycles were conflicts)
constexpr int PLANE = ROWS * 64 + 32;                  // 15 392 B
constexpr int BUF = 4 * PLANE;                         // one patch, 64 channels: 61 568 B
constexpr int EP = 68;                                 // dump row pitch (epilogue_vec<2>: COLS + 4)
constexpr int DUMP_FLOATS = 4 * 32 * EP;
constexpr int ETAB_FLOATS = 4 * 16 * 12;               // per wave, per channel quad: bias4 | next_s4 | next_t4
constexpr int PTAB_FLOATS = 16 * 8;                    // per channel quad: pro_s4 | pro_t4 of the staged image
constexpr size_t LDS_BYTES = 2 * (size_t)BUF + (DUMP_FLOATS + 32 + ETAB_FLOATS + PTAB_FLOATS) * sizeof(float);

// EPI: 0 = the shared epilogue item behind the dump (every mode: residual, StyleGAN2 terms, fused activation
// backward ...), run between two tiles; 1 / 2 / 3 = the forward-style epilogue  y = act(acc + bias) [mask] ->
// store [-> 2x2 max pool -> store] + partial maxima  (1 plain, 2 with a mask, 3 with the pool) of tile i written
// out UNDER the multiply stream of tile i + 1: a lone wave per SIMD cannot hide its own stores behind another
// wave, and the 32 KB a tile stores are issue-bound (~2.3 k cycles per CU) -- measured sequentially the
// epilogue cost as much as the 216 MFMAs (profiles/round6_h2r_ablation.txt).  Same operations in the same
// order as epi_item: the bits do not change.
template <int PRO, int EPI>
__global__ __launch_bounds__(256, 1) void conv_h2r_kernel(const ConvK k, const int n_tiles) {
  constexpr bool FAST = EPI != 0, MASK = EPI == 2, POOL = EPI == 3;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* As = reinterpret_cast<char*>(smem);
  float* dump = smem + 2 * BUF / 4;
  float* red = dump + DUMP_FLOATS;                     // 32 floats: block reductions of the scales
  float* etab = red + 32;                              // per-wave constants of the pipelined epilogue
  float* ptab = etab + ETAB_FLOATS;                    // the fused prologue's affine of the image being staged

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int mh = wave >> 1, nt = wave & 1;

  // ---- weights: the fp16 x 2 image [chunk][32-channel tile][tap][32 rows][64 B] (p2l_h2.hip h2_pack_kernel),
  //      this wave's 36 fragments of both pieces, once
  h16x8 bh[36], bm[36];
  {
    const char* wb = reinterpret_cast<const char*>(k.w) + l31 * 64;
    const int oh = h2c(lhi, l31) * 16, om = h2c(2 + lhi, l31) * 16;
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      const int c = s / 9, tap = s - c * 9;
      const char* slab = wb + (size_t)(((c * 2 + nt) * 9 + tap) * 32) * 64;
      bh[s] = *reinterpret_cast<const h16x8*>(slab + oh);
      bm[s] = *reinterpret_cast<const h16x8*>(slab + om);
    }
  }

  // ---- my tiles: a contiguous range; the blocks of one XCD (dispatch order: block b -> XCD b % 8) take
  //      neighbouring ranges, so that the halos two tiles share are met in one L2
  const int G = gridDim.x;
  const int v = xcd_remap(blockIdx.x, G);
  const int t_begin = (int)(((long)v * n_tiles) / G), t_end = (int)(((long)(v + 1) * n_tiles) / G);
  const int tiles_per_image = k.tiles_x * k.tiles_y;

  // ---- staging descriptors (the same for every tile).  Items 0 .. 9: patch line `it`, pixel hx = tid >> 4
  //      (0 .. 15) of it, channel quad cq = tid & 15 (16 lanes = one pixel's 256 bytes, 256 threads = 4 KB of one
  //      image row); items 10, 11: the two right-hand halo pixels (hx = 16, 17) of every line.  LDS: plane
  //      cq >> 2, row = line * 24 + hx of 64 bytes [h k0-7 | h k8-15 | m k0-7 | m k8-15], the 16-byte chunk
  //      swizzled by the row (h2c).  24 lines apart the swizzle toggles bit 1 of the chunk: offset ^ 32, which is
  //      also where the m piece of a value sits -- odd lines swap the two pieces' places.
  const int cq = tid & 15, p0 = tid >> 4;
  const int goff0 = (-k.W + (p0 - 1)) * k.x_ld + cq * 4;                 // line 0 (image row y0 - 1)
  const int lplane = (cq >> 2) * PLANE;                 // (added AFTER the ^ 32: a plane offset has bit 5 set)
  const int loff0 = p0 * 64 + (cq & 1) * 8 + h2c((cq >> 1) & 1, p0) * 16;
  int e_line[2], e_hx[2], e_goff[2], e_loff[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int idx = tid + 256 * e;
    e_line[e] = idx < 320 ? (idx >> 5) : -1;
    e_hx[e] = 16 + ((idx >> 4) & 1);
    const int row = (idx >> 5) * HP + e_hx[e];
    e_goff[e] = (((idx >> 5) - 1) * k.W + (e_hx[e] - 1)) * k.x_ld + cq * 4;
    e_loff[e] = row * 64 + (cq & 1) * 8 + h2c((cq >> 1) & 1, row) * 16;
  }
  const int line_step = k.W * k.x_ld;

  // ---- fragment addressing: wave (mh, nt), pixel sub-tile ms: MFMA row l31 = pixel (mh*2+ms)*32 + l31 of the
  //      tile in 2x2-quad order (epilogue_vec's order).  Tap (dy, dx): row + 24 dy + dx -- per dx one offset,
  //      dy adds 1536 bytes and (odd dy) swaps the pieces as above
  int aoff[2][3];
#pragma unroll
  for (int ms = 0; ms < 2; ++ms) {
    const int i = (mh * 2 + ms) * 32 + l31;
    const int Q = i >> 2, s = i & 3;
    const int row0 = (2 * ((Q >> 3) & 3) + (s >> 1)) * HP + 2 * (Q & 7) + (s & 1);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) aoff[ms][dx] = (row0 + dx) * 64 + h2c(lhi, row0 + dx) * 16;
  }

  float sw, inv_w;
  h2_scales(__builtin_amdgcn_readfirstlane(k.w_tail[0]), sw, inv_w);

  // scales of image b: x scale (power of two from max |x| of the image -- the maxima the producer of the tensor
  // handed over, or the 64 partials of the pass in front of the launch) and the exact un-scale of the products
  auto image_scales = [&](int b, float& xs, float& os) {
    float a = 0.f, ms_ = 0.f, mt_ = 0.f;
    if (k.amax_in != nullptr) {
      for (int i = tid; i < k.amax_in_n; i += 256) a = fmaxf(a, k.amax_in[(size_t)b * k.amax_in_n + i]);
      if (PRO != P2L_PRO_NONE && !k.amax_in_applied) {
        const f32x4* ps = reinterpret_cast<const f32x4*>(k.pro_s + (size_t)b * k.pro_bstride);
        const f32x4* pt = reinterpret_cast<const f32x4*>(k.pro_t + (size_t)b * k.pro_bstride);
        for (int c = tid; c < (k.Cin >> 2); c += 256) {
          const f32x4 s4 = ps[c], t4 = pt[c];
          ms_ = fmaxf(fmaxf(ms_, fmaxf(fabsf(s4.x), fabsf(s4.y))), fmaxf(fabsf(s4.z), fabsf(s4.w)));
          mt_ = fmaxf(fmaxf(mt_, fmaxf(fabsf(t4.x), fabsf(t4.y))), fmaxf(fabsf(t4.z), fabsf(t4.w)));
        }
      }
    } else if (tid < 64) {
      a = k.amax[b * 64 + tid];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      a = fmaxf(a, __shfl_xor(a, o, 64));
      if (PRO != P2L_PRO_NONE) { ms_ = fmaxf(ms_, __shfl_xor(ms_, o, 64)); mt_ = fmaxf(mt_, __shfl_xor(mt_, o, 64)); }
    }
    __syncthreads();                                   // (the scratch of the previous call has been read)
    if (lane == 0) { red[wave * 4] = a; red[wave * 4 + 1] = ms_; red[wave * 4 + 2] = mt_; }
    if (PRO != P2L_PRO_NONE && tid < 16) {             // (the tiles staged from here on belong to image b)
      *reinterpret_cast<f32x4*>(ptab + tid * 8) = *reinterpret_cast<const f32x4*>(k.pro_s + (size_t)b * k.pro_bstride + tid * 4);
      *reinterpret_cast<f32x4*>(ptab + tid * 8 + 4) = *reinterpret_cast<const f32x4*>(k.pro_t + (size_t)b * k.pro_bstride + tid * 4);
    }
    __syncthreads();
    a = fmaxf(fmaxf(red[0], red[4]), fmaxf(red[8], red[12]));
    ms_ = fmaxf(fmaxf(red[1], red[5]), fmaxf(red[9], red[13]));
    mt_ = fmaxf(fmaxf(red[2], red[6]), fmaxf(red[10], red[14]));
    if (PRO != P2L_PRO_NONE && k.amax_in != nullptr) a = (k.amax_in_applied ? a : ms_ * a + mt_) * 1.001f;
    float inv_x;
    h2_scales(__builtin_bit_cast(unsigned, a), xs, inv_x);
    os = inv_x * inv_w;
  };

  // ---- staging of tile t into patch buffer `buf` in two halves (lines 0-5 | lines 6-9 + the edge items):
  //      issue<half>() starts the loads, land(it) converts and writes one item
  f32x4 xr[6];
  float st_xs = 1.f;
  char* st_dst = As;
  const float* st_base = k.x;
  bool st_top = false, st_bot = false, st_left = false, st_right = false;
  auto tile_geom = [&](int t, int& b, int& y0, int& x0, int& til) {
    b = t / tiles_per_image;
    til = t - b * tiles_per_image;
    const int ty = til / k.tiles_x;
    y0 = ty << 3; x0 = (til - ty * k.tiles_x) << 4;
  };
  // does item `it` of the staged tile exist in the image (else: zero padding)
  auto item_ok = [&](int it) -> bool {
    if (it < 10) return !((st_top && it == 0) || (st_bot && it == 9) || (st_left && p0 == 0));
    const int e = it - 10;
    return e_line[e] >= 0 && !((st_top && e_line[e] == 0) || (st_bot && e_line[e] == 9) || (st_right && e_hx[e] == 17));
  };
  auto begin_tile = [&](int t, int buf, float xs) {
    int b, y0, x0, til;
    tile_geom(t, b, y0, x0, til);
    st_top = (y0 == 0); st_bot = (y0 + 8 == k.H); st_left = (x0 == 0); st_right = (x0 + 16 == k.W);
    st_xs = xs;
    st_dst = As + buf * BUF;
    st_base = k.x + (size_t)((b * k.H + y0) * k.W + x0) * k.x_ld;
  };
  auto issue = [&](auto half_c) {                       // (unconditional loads from an always valid address)
    constexpr int half = decltype(half_c)::value;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int it = half * 6 + j;
      const int go = it < 10 ? goff0 + it * line_step : e_goff[it - 10];
      xr[j] = *reinterpret_cast<const f32x4*>(st_base + (item_ok(it) ? go : cq * 4));
    }
  };
  auto land = [&](int it) {
    if (it >= 10 && e_line[it - 10] < 0) return;
    f32x4 v4 = xr[it % 6];
    if (PRO != P2L_PRO_NONE) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(ptab + cq * 8), t4 = *reinterpret_cast<const f32x4*>(ptab + cq * 8 + 4);
      v4 = v4 * s4 + t4;
      if (PRO == P2L_PRO_AFFINE_RELU) {
        v4.x = fmaxf(v4.x, 0.f); v4.y = fmaxf(v4.y, 0.f); v4.z = fmaxf(v4.z, 0.f); v4.w = fmaxf(v4.w, 0.f);
      }
    }
    v4 = v4 * (item_ok(it) ? st_xs : 0.f);                          // (x 0: zero padding, after the prologue)
    const h16x4 h = __builtin_convertvector(v4, h16x4);
    const f32x4 w = __builtin_convertvector(h, f32x4);
    const h16x4 m = __builtin_convertvector(v4 - w, h16x4);
    const int lo = it < 10 ? ((loff0 + it * (HP * 64)) ^ ((it & 1) ? 32 : 0)) : e_loff[it - 10];
    *reinterpret_cast<h16x4*>(st_dst + lplane + lo) = h;
    *reinterpret_cast<h16x4*>(st_dst + lplane + (lo ^ 32)) = m;
  };
  using Half0 = std::integral_constant<int, 0>;
  using Half1 = std::integral_constant<int, 1>;

  // ---- the pipelined epilogue (FAST): a lane owns quads q = lane >> 4 and 4 + (lane >> 4) of its wave's 32-pixel
  //      group and channels n = 4 (lane & 15) .. + 3 -- epilogue_vec_items' assignment --------------------------
  const int e_c4 = lane & 15, e_n = e_c4 * 4;
  // bias and the reader's affine (maxima) of a lane's four channels: constants per image, parked in LDS (one
  // table per wave: written and read by the same wave, no barrier) -- 12 registers the stream cannot spare
  float* my_etab = etab + (wave * 16 + e_c4) * 12;
  f32x4 e_bias, e_ns, e_nt;
  if (FAST && lane < 16) {
    *reinterpret_cast<f32x4*>(my_etab) = k.bias ? ld4(k.bias, (unsigned)e_n) : f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(my_etab + 4) = f32x4{1.f, 1.f, 1.f, 1.f};
    *reinterpret_cast<f32x4*>(my_etab + 8) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int e_tab_b = -1;                                     // image whose next_s / next_t the table holds
  const float e_lo = (k.act == P2L_ACT_RELU) ? 0.f : -__builtin_inff();     // act: max(t, lo)
  int pb = 0, py0 = 0, px0 = 0, ptil = 0;                                      // the tile whose dump is in LDS
  float e_amax = 0.f, e_amaxp = 0.f;
  f32x4 ev, ep, em[4];                                 // (pixel in flight, running 2x2 max, the quad's mask)
  unsigned e_pix0 = 0, e_ppix = 0;
  auto epi_tile = [&](int b, int y0_, int x0_, int til_) {
    pb = b; py0 = y0_; px0 = x0_; ptil = til_;
    e_amax = 0.f; e_amaxp = 0.f;
    if (k.amax_ps && b != e_tab_b) {
      if (lane < 16) {
        *reinterpret_cast<f32x4*>(my_etab + 4) = ld4(k.amax_ps, (unsigned)(b * k.amax_pbstride + e_n));
        *reinterpret_cast<f32x4*>(my_etab + 8) = ld4(k.amax_pt, (unsigned)(b * k.amax_pbstride + e_n));
      }
      e_tab_b = b;
    }
  };
  auto epi_addr = [&](int j) {
    const int Q = wave * 8 + j * 4 + (lane >> 4);
    const int oy0 = py0 + 2 * ((Q >> 3) & 3), ox0 = px0 + 2 * (Q & 7);
    e_pix0 = (unsigned)((pb * k.H + oy0) * k.W + ox0);
    e_ppix = (unsigned)((pb * (k.H >> 1) + (oy0 >> 1)) * (k.W >> 1) + (ox0 >> 1));
  };
  auto epi_mask = [&](int j) {                          // (requested a few steps ahead of its use)
    epi_addr(j);
    if (MASK) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        em[s] = ld4(k.mask, (e_pix0 + (unsigned)((s >> 1) * k.W + (s & 1))) * (unsigned)k.mask_ld + (unsigned)e_n);
    }
  };
  auto epi_read = [&](int j, int s) {                   // (one step ahead of epi_px(s))
    ev = *reinterpret_cast<const f32x4*>(dump + wave * 32 * EP + (4 * (j * 4 + (lane >> 4)) + s) * EP + e_c4 * 4);
    if (s == 0) {
      e_bias = *reinterpret_cast<const f32x4*>(my_etab);
      e_ns = *reinterpret_cast<const f32x4*>(my_etab + 4);
      e_nt = *reinterpret_cast<const f32x4*>(my_etab + 8);
    }
  };
  auto epi_px = [&](int s) {
    f32x4 t = ev + e_bias;
    t.x = fmaxf(t.x, e_lo); t.y = fmaxf(t.y, e_lo); t.z = fmaxf(t.z, e_lo); t.w = fmaxf(t.w, e_lo);
    if (MASK) {
      t.x = em[s].x > 0.f ? t.x : 0.f; t.y = em[s].y > 0.f ? t.y : 0.f;
      t.z = em[s].z > 0.f ? t.z : 0.f; t.w = em[s].w > 0.f ? t.w : 0.f;
    }
    st4(k.y, (e_pix0 + (unsigned)((s >> 1) * k.W + (s & 1))) * (unsigned)k.y_ld + (unsigned)e_n, t);
    e_amax = absmax4(e_amax, t * e_ns + e_nt);
    if (POOL) {                                         // (max is exact: any order of the four gives epi_item's bits)
      if (s == 0) ep = t;
      else { ep.x = fmaxf(ep.x, t.x); ep.y = fmaxf(ep.y, t.y); ep.z = fmaxf(ep.z, t.z); ep.w = fmaxf(ep.w, t.w); }
    }
  };
  auto epi_pool = [&]() {
    if (POOL) {
      st4(k.yp, e_ppix * (unsigned)k.yp_ld + (unsigned)e_n, ep);
      e_amaxp = absmax4(e_amaxp, ep);
    }
  };
  auto epi_maxima = [&]() {                             // one partial per wave and tile (epilogue_vec_items' slots)
    if (k.amax_out != nullptr || k.amax_outp != nullptr) {
      float m = e_amax, mp = e_amaxp;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); mp = fmaxf(mp, __shfl_xor(mp, o, 64)); }
      if (lane == 0) {
        const size_t slot = (size_t)pb * k.amax_out_n + (size_t)ptil * 4 + wave;
        if (k.amax_out != nullptr) k.amax_out[slot] = m;
        if (k.amax_outp != nullptr) k.amax_outp[slot] = mp;
      }
    }
  };
  // where the pieces sit in the 36 steps of the stream (the patch of the next tile lands at 10-15 and 28-33)
  auto epi_hook = [&](int s) {
    if (s == 0) epi_mask(0);
    if (s >= 5 && s < 9) epi_px(s - 5);                 // (consumes the pixel read one step earlier ...)
    if (s >= 4 && s < 8) epi_read(0, s - 4);            // (... before the next one replaces it)
    if (s == 9) epi_pool();
    if (s == 16) epi_mask(1);
    if (s >= 21 && s < 25) epi_px(s - 21);
    if (s >= 20 && s < 24) epi_read(1, s - 20);
    if (s == 25) epi_pool();
  };

  if (t_begin >= t_end) return;

  // ---- first tile ----
  int b_cur, y0, x0, til;
  tile_geom(t_begin, b_cur, y0, x0, til);
  float xs_cur, os_cur;
  image_scales(b_cur, xs_cur, os_cur);
  begin_tile(t_begin, 0, xs_cur);
  issue(Half0{});
#pragma unroll
  for (int it = 0; it < 6; ++it) land(it);
  issue(Half1{});
#pragma unroll
  for (int it = 6; it < 12; ++it) land(it);
  __syncthreads();

  auto run_tile = [&](int t, auto with_epi_c) {
    constexpr bool WITH_EPI = decltype(with_epi_c)::value;       // (FAST: the previous tile's dump is written out)
    const int cur = (t - t_begin) & 1;
    tile_geom(t, b_cur, y0, x0, til);
    // the NEXT tile (the last iteration re-stages its own tile: no branch in the stream below); its image may
    // be another one, with another scale
    const int tn = (t + 1 < t_end) ? t + 1 : t;
    const int b_next = tn / tiles_per_image;
    float xs_next = xs_cur, os_next = os_cur;
    if (b_next != b_cur) image_scales(b_next, xs_next, os_next);
    begin_tile(tn, cur ^ 1, xs_next);
    if (!(P2L_H2R_ABL & 1)) issue(Half0{});

    const char* Ab = As + cur * BUF;
    f32x16 acc[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][r] = 0.f;

    h16x8 af[2][2][2];                                  // [double buffer][ms][h | m]
    auto lda = [&](int s, h16x8 (&a)[2][2]) {
      const int c = s / 9, tap = s - c * 9;
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) {
        // (the m piece: 16-byte chunk index ^ 2 = offset ^ 32; an odd dy swaps the pieces' places)
        const int oh = aoff[ms][dx] ^ ((dy & 1) ? 32 : 0);
        a[ms][0] = *reinterpret_cast<const h16x8*>(Ab + c * PLANE + dy * (HP * 64) + oh);
        a[ms][1] = *reinterpret_cast<const h16x8*>(Ab + c * PLANE + dy * (HP * 64) + (oh ^ 32));
      }
    };
    lda(0, af[0]);
#pragma unroll
    for (int s = 0; s < 36; ++s) {
      if (s + 1 < 36 && !(P2L_H2R_ABL & 16)) lda(s + 1, af[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const h16x8 (&a)[2][2] = af[s & 1];
      // per accumulator: m h, h m, h h (smallest terms first) -- the order of conv_h2_kernel; the two accumulators
      // alternate.  (Measured against one back-to-back chain per accumulator -- acc0 x 3, acc1 x 3 --: MFMA-busy
      // 54 % vs 45 %, waves parked 15 % vs 27 %: the chain form needs both pieces of a sub-tile's fragment at
      // once and waits for the LDS more often.)
      if (!(P2L_H2R_ABL & 2)) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][1], bh[s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][1], bh[s], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0], bm[s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0], bm[s], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0], bh[s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0], bh[s], acc[1], 0, 0, 0);
      } else { acc[0][s & 15] += (float)a[0][0][0] * (float)bh[s][0] + (float)a[1][1][1] * (float)bm[s][1]; }
      // the next tile's patch lands under the stream: lines 0-5 (requested in front of it) at steps 10 .. 15, the
      // second half is requested behind them and lands at steps 28 .. 33
      if (!(P2L_H2R_ABL & 1)) {
        if (s >= 10 && s < 16) land(s - 10);
        if (s == 15) issue(Half1{});
        if (s >= 28 && s < 34) land(s - 28 + 6);
      }
      if (WITH_EPI && !(P2L_H2R_ABL & 32)) epi_hook(s);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (WITH_EPI) epi_maxima();

    // ---- un-scale (exact: a power of two), dump in epilogue_vec's layout: pixel group mh*2+ms, columns nt*32.. ----
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][r] *= os_cur;
    __syncthreads();                                    // the items of the previous tile have been read
    if (!(P2L_H2R_ABL & 8) || acc[0][0] == 12345.678f)
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      float* tb = dump + (mh * 2 + ms) * 32 * EP + nt * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2) + 4 * lhi) * EP] = acc[ms][r];
    }
    __syncthreads();                                    // dumps visible; the next patch is complete
    if (FAST) {
      epi_tile(b_cur, y0, x0, til);
    } else {
      if (!(P2L_H2R_ABL & 4) || acc[0][0] == 12345.678f)
      epilogue_vec_items<2>(k, dump, wave, lane, b_cur, y0, x0, 0, til, 0, 0, 0);
    }
    xs_cur = xs_next; os_cur = os_next;
  };
  if (FAST) {
    run_tile(t_begin, std::false_type{});
    for (int t = t_begin + 1; t < t_end; ++t) run_tile(t, std::true_type{});
    // the last tile's dump, with nothing to hide behind
#pragma unroll
    for (int s = 0; s < 36; ++s) epi_hook(s);
    epi_maxima();
  } else {
    for (int t = t_begin; t < t_end; ++t) run_tile(t, std::false_type{});
  }
}

}  // namespace

// which launches: plain stride-1 3x3, 64 -> 64 channels, whole 8x16-pixel tiles of one image
bool p2l_h2r_shape(int taps, int ups, int H, int W, int Cin, int Cout, int x_ld) {
#ifdef P2L_AB_NO_H2R               // (A/B builds: the chunked kernel for these layers)
  return false;
#endif
  return taps == 9 && ups == 0 && Cin == 64 && Cout == 64 && H % 8 == 0 && W % 16 == 0 && x_ld % 4 == 0;
}

// which epilogue a launch gets: 1 / 2 / 3 = forward-style, pipelined under the next tile (plain | mask | 2x2 max
// pool); 0 = the shared item between two tiles (everything else)
static int h2r_epi(const ConvK& k) {
  if (!k.arb_x && !k.res && !k.oscale && !k.noise && k.alpha == 1.0f && k.y != nullptr && k.n_store == 64 &&
      (k.act == P2L_ACT_NONE || k.act == P2L_ACT_RELU) && !(k.form & P2L_FORM_H2R_SEQ_EPI)) {
    if (k.pool == P2L_POOL_NONE) return k.mask ? 2 : 1;
    if (k.pool == P2L_POOL_MAX && !k.mask && k.yp != nullptr) return 3;
  }
  return 0;
}
// Does this launch of an h2r-shaped layer run here?  With the pipelined epilogue, yes; with the shared item
// between two tiles (fused activation backward, residual, StyleGAN2 terms) the lone wave per SIMD has nothing to
// hide its stores behind and the chunked kernel -- three blocks per CU -- is the faster one (in the bench step:
// 256^2 fused activation backward 0.455 vs 0.374 ms): only on request (P2L_FORM_H2R_SEQ_EPI: tests).  Both give
// the same bits and the same maxima slots, so the choice may follow the epilogue.
bool p2l_h2r_takes(const ConvK& k) { return h2r_epi(k) != 0 || (k.form & P2L_FORM_H2R_SEQ_EPI); }

// k: the ConvK of the direct fp16 x 2 launch (k.w = the fp16 x 2 direct image, k.w_tail, k.amax | k.amax_in).
int p2l_h2r_launch(const ConvK& k_in, int pro, hipStream_t st) {
  ConvK k = k_in;
  k.n_ntiles = 1;                                       // a block writes all 64 channels (maxima slots, sums)
  const int n_tiles = k.B * k.tiles_x * k.tiles_y;
  static std::atomic<int> n_cu{0};
  if (n_cu == 0) {
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    n_cu = cus;
  }
  const int grid = n_tiles < n_cu ? n_tiles : (int)n_cu;
  // the pipelined forward-style epilogue where the launch asks for nothing else (the compiled-in order of
  // operations is epi_item's: same bits), the shared item otherwise
  const int epi = h2r_epi(k);
#define P2L_H2RL(PROV)                                                                        \
  do {                                                                                        \
    if (epi == 1) P2L_H2RE(PROV, 1); else if (epi == 2) P2L_H2RE(PROV, 2);                    \
    else if (epi == 3) P2L_H2RE(PROV, 3); else P2L_H2RE(PROV, 0);                             \
  } while (0)
#define P2L_H2RE(PROV, EPIV)                                                                  \
  do {                                                                                        \
    auto kfn = conv_h2r_kernel<PROV, EPIV>;                                                   \
    static std::atomic<bool> attr_set{false};                                                 \
    if (!attr_set) {                                                                          \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                160 * 1024);                                                  \
      attr_set = true;                                                                        \
    }                                                                                         \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), LDS_BYTES, st, k, n_tiles);                \
  } while (0)
  if (pro == P2L_PRO_NONE) P2L_H2RL(P2L_PRO_NONE);
  else if (pro == P2L_PRO_AFFINE_RELU) P2L_H2RL(P2L_PRO_AFFINE_RELU);
  else P2L_H2RL(P2L_PRO_AFFINE);
#undef P2L_H2RL
#undef P2L_H2RE
  return p2l_check_launch();
}
