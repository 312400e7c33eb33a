// conv v2: persistent, software-pipelined 3x3 implicit GEMM on v_mfma_f32_32x32x2_f32.
//
// STATUS: experimental variant, selected with p2l_set_conv_variant(0|1); parity-tested
// (tests/test_kernels_gpu.py) but NOT the default: on MI355X it measures within +-3 % of
// v1 on the 128^2/256^2 layers and loses on mid-size layers to tile quantisation.  An
// ablation (P2L_ABL bits: skip loads / LDS writes / barrier / stores / LDS reads) shows
// the MFMA-only skeleton of this loop tops out at ~120 TFLOP/s on random data (sustained
// clock ~2.14 GHz under fp32-MFMA load), i.e. v1 at 100-117 TFLOP/s is already within
// 85-95 % of what the matrix pipes deliver here.
//
// Why a second kernel was tried: PMC counters on the v1 kernel (profiles/round1_*) show the
// matrix pipes busy only ~74 % of the time although two blocks share every CU.
// Co-resident blocks start together, contend for the same pipes symmetrically and
// therefore stay phase-locked: both reach their "barrier -> refill LDS -> barrier"
// phase at the same time and the pipes idle.  v2 removes the phase instead of
// hoping another block covers it:
//   * LDS is double-buffered inside ONE 8-wave workgroup (2 x 72 KB of the CU's
//     160 KB): while the MFMAs of K-chunk g read stage g&1, the same waves write
//     chunk g+1 (prefetched into registers at the start of chunk g) into stage
//     (g+1)&1; ONE barrier per chunk, nothing else between chunks;
//   * the workgroup is persistent: 256 workgroups (one per CU) walk the tile list,
//     and the first chunk of the next tile is prefetched during the last chunk of
//     the current one, so the per-tile prologue latency (index math + first HBM
//     round trip) is paid once per launch instead of once per tile;
//   * tile = 256 pixels (16x16 patch) x 64 channels (8 waves x 32px x 64ch) or
//     128 pixels (8x16) x 64 channels (8 waves as 4(M) x 2(N), 32px x 32ch each),
//     picked per launch for the better balance over 256 CUs;
//   * same quad-ordered pixel enumeration, fused prologue (affine / affine+ReLU /
//     nearest-x2) and epilogue (bias, residual, activation, mask, 2x2 pool) as v1.
#include "p2l_conv_k.h"

using namespace p2lconv;

namespace {

template <int TAPS, int PRO, bool UPS, int CFG>
__global__ __launch_bounds__(512, 1) void conv2_kernel(const ConvK k) {
  constexpr int KC = 16, PITCH = KC + 4, VPR = KC / 4, BN = 64;
  constexpr int TH = (CFG == 0) ? 16 : 8, TW = 16;
  constexpr int HALO_W = (TAPS == 9) ? TW + 2 : TW;
  constexpr int HALO_H = (TAPS == 9) ? TH + 2 : TH;
  constexpr int A_ROWS = HALO_H * HALO_W;
  constexpr int B_ROWS = TAPS * BN;
  constexpr int STAGE = (A_ROWS + B_ROWS) * PITCH;  // floats per LDS stage
  constexpr int A_ITEMS = A_ROWS * VPR;
  constexpr int A_ITERS = (A_ITEMS + 511) / 512;
  constexpr int B_ITEMS = B_ROWS * VPR;
  constexpr int B_ITERS = (B_ITEMS + 511) / 512;
  constexpr int NT = (CFG == 0) ? 2 : 1;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int wm = (CFG == 0) ? wave : (wave & 3);   // which 32-pixel strip
  const int wn = (CFG == 0) ? 0 : (wave >> 2);     // which 32-channel half (CFG 1)

  const int tiles_x_log = k.tiles_x_log, tiles_y_log = k.tiles_y_log;
  const int n_tiles = k.n_mtiles * k.n_ntiles;
  const int nchunks = k.nchunks;

  // tile order: give each XCD (block b -> XCD b%8) a contiguous run of 32 tiles per
  // round so that the N-tiles of one pixel tile and neighbouring pixel tiles share an L2.
  const int nblk = gridDim.x;
  auto tile_of = [&](int it) {
    if ((nblk & 7) == 0) {
      const int per = nblk >> 3;
      return it * nblk + (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    }
    return it * nblk + (int)blockIdx.x;
  };

  // ---- per-tile state ------------------------------------------------------
  int a_goff[A_ITERS], a_soff[A_ITERS];
  unsigned a_valid = 0;
  int t_n0 = 0, t_y0 = 0, t_x0 = 0, t_b = 0;

  auto setup_tile = [&](int tile) {
    const int mt = tile / k.n_ntiles, nt = tile - mt * k.n_ntiles;
    const int tx = mt & ((1 << tiles_x_log) - 1);
    const int ty = (mt >> tiles_x_log) & ((1 << tiles_y_log) - 1);
    t_b = mt >> (tiles_x_log + tiles_y_log);
    t_n0 = nt * BN;
    t_y0 = ty * TH;
    t_x0 = tx * TW;
    a_valid = 0;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const int j = tid + 512 * it;
      const int p = j / VPR, v = j - p * VPR;
      a_goff[it] = 0;
      a_soff[it] = 0;
      if (p < A_ROWS) {
        int iy, ix;
        if (TAPS == 9) {
          const int hy = p / HALO_W, hx = p - hy * HALO_W;
          iy = t_y0 + hy - 1;
          ix = t_x0 + hx - 1;
        } else {
          const int Q = p >> 2, s = p & 3;
          iy = t_y0 + 2 * (Q >> 3) + (s >> 1);
          ix = t_x0 + 2 * (Q & 7) + (s & 1);
        }
        if (iy >= 0 && iy < k.H && ix >= 0 && ix < k.W) {
          int pix;
          if (UPS)
            pix = (t_b * (k.H >> 1) + (iy >> 1)) * (k.W >> 1) + (ix >> 1);
          else
            pix = (t_b * k.H + iy) * k.W + ix;
          a_goff[it] = pix * k.x_ld + v * 4;
          a_soff[it] = t_b * k.pro_bstride + v * 4;
          a_valid |= 1u << it;
        }
      }
    }
  };

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
      const int j = tid + 512 * it;
      if (j < B_ITEMS) {
        const int tap = j / (BN * VPR);
        const int rem = j - tap * (BN * VPR);
        const size_t off = (((size_t)tap * nchunks + c) * k.Cout + t_n0) * KC + rem * 4;
        wr[it] = *reinterpret_cast<const f32x4*>(k.w + off);
      }
    }
  };

  auto write_lds = [&](float* As, float* Bs) {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const int j = tid + 512 * it;
      const int p = j / VPR, v = j - p * VPR;
      if (p < A_ROWS) {
        f32x4 val = xr[it];
        if (PRO != P2L_PRO_NONE) {
          val = val * sr[it] + tr[it];
          if (PRO == P2L_PRO_AFFINE_RELU) {
            val.x = fmaxf(val.x, 0.f);
            val.y = fmaxf(val.y, 0.f);
            val.z = fmaxf(val.z, 0.f);
            val.w = fmaxf(val.w, 0.f);
          }
        }
        if (!((a_valid >> it) & 1u)) val = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(As + p * PITCH + v * 4) = val;
      }
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int j = tid + 512 * it;
      if (j < B_ITEMS) {
        const int row = j / VPR, v = j - row * VPR;
        *reinterpret_cast<f32x4*>(Bs + row * PITCH + v * 4) = wr[it];
      }
    }
  };

  // ---- fragment addressing (stage-relative) ---------------------------------
  int a_row0;
  {
    const int i = wm * 32 + l31;
    const int Q = i >> 2, s = i & 3;
    if (TAPS == 9)
      a_row0 = (2 * (Q >> 3) + (s >> 1)) * HALO_W + 2 * (Q & 7) + (s & 1);
    else
      a_row0 = i;
  }
  const int a_frag_off = a_row0 * PITCH + lhi * 4;
  const int b_frag_off = A_ROWS * PITCH + (wn * 32 + l31) * PITCH + lhi * 4;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  int it_idx = 0;
  int tile = tile_of(0);
  if (tile >= n_tiles) return;
  setup_tile(tile);
  load_regs(0);
  write_lds(smem, smem + A_ROWS * PITCH);
  __syncthreads();
  int stage = 0;

  while (tile < n_tiles) {
    // epilogue coordinates of THIS tile (setup_tile(next) overwrites the t_* state)
    const int e_n0 = t_n0, e_y0 = t_y0, e_x0 = t_x0, e_b = t_b;
    const int next_tile_id = tile_of(it_idx + 1);
    for (int c = 0; c < nchunks; ++c) {
      const bool last = (c == nchunks - 1);
      const bool have_next = !last || (next_tile_id < n_tiles);
      if (have_next) {
        if (last) setup_tile(next_tile_id);
        if (!(k.abl & 1)) load_regs(last ? 0 : c + 1);
      }
      const float* st = smem + stage * STAGE;
      const float* a_frag = st + a_frag_off;
      const float* b_frag = st + b_frag_off;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
        const float* ap = a_frag + ((TAPS == 9) ? (dy * HALO_W + dx) * PITCH : 0);
#pragma unroll
        for (int kk = 0; kk < KC / 8; ++kk) {
          const f32x4 a = *reinterpret_cast<const f32x4*>((k.abl & 16) ? a_frag : ap + kk * 8);
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            // 4 MFMAs back to back on one accumulator: measured faster than
            // interleaving the two accumulators (100 vs 96 TFLOP/s)
            const f32x4 bq = *reinterpret_cast<const f32x4*>(
                (k.abl & 16) ? b_frag : b_frag + (tap * BN + j * 32) * PITCH + kk * 8);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq.z, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq.w, acc[j], 0, 0, 0);
          }
        }
      }
      if (have_next && !(k.abl & 2)) {
        float* nst = smem + (stage ^ 1) * STAGE;
        write_lds(nst, nst + A_ROWS * PITCH);
      }
      if (!(k.abl & 4)) __syncthreads();
      stage ^= 1;
    }

    // ---- epilogue of the finished tile ---------------------------------------
    {
      const bool simple = !k.res && !k.mask && !k.pool && k.y;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = e_n0 + (wn + j) * 32 + l31;
        if (n < k.n_store && !(k.abl & 8)) {
          const float bias_n = k.bias ? k.bias[n] : 0.f;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int Q = wm * 8 + 2 * g + lhi;
            const int oy0 = e_y0 + 2 * (Q >> 3), ox0 = e_x0 + 2 * (Q & 7);
            const int pix0 = (e_b * k.H + oy0) * k.W + ox0;
            const float a[4] = {acc[j][g * 4 + 0], acc[j][g * 4 + 1], acc[j][g * 4 + 2],
                                acc[j][g * 4 + 3]};
            if (simple)
              epilogue_quad<true>(k, a, pix0, e_b, oy0, ox0, n, bias_n);
            else
              epilogue_quad<false>(k, a, pix0, e_b, oy0, ox0, n, bias_n);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      }
    }
    ++it_idx;
    tile = next_tile_id;
  }
}

template <int TAPS, int CFG>
int launch2(const ConvK& k, int pro, int ups, hipStream_t st) {
  constexpr int TH = (CFG == 0) ? 16 : 8;
  constexpr int A_ROWS = (TAPS == 9) ? (TH + 2) * 18 : TH * 16;
  const size_t lds = 2 * (size_t)(A_ROWS + TAPS * 64) * 20 * sizeof(float);
  const int n_tiles = k.n_mtiles * k.n_ntiles;
  const int grid = n_tiles < 256 ? n_tiles : 256;
#define P2L_LAUNCH2(PRO, UPS)                                                        \
  do {                                                                               \
    auto kfn = conv2_kernel<TAPS, PRO, UPS, CFG>;                                    \
    static bool attr_set = false;                                                    \
    if (!attr_set) {                                                                 \
      (void)hipFuncSetAttribute((const void*)kfn,                                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize,          \
                                160 * 1024);                                         \
      attr_set = true;                                                               \
    }                                                                                \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, st, k);                      \
  } while (0)
  if (ups) {
    if (pro == P2L_PRO_NONE) P2L_LAUNCH2(P2L_PRO_NONE, true);
    else if (pro == P2L_PRO_AFFINE_RELU) P2L_LAUNCH2(P2L_PRO_AFFINE_RELU, true);
    else P2L_LAUNCH2(P2L_PRO_AFFINE, true);
  } else {
    if (pro == P2L_PRO_NONE) P2L_LAUNCH2(P2L_PRO_NONE, false);
    else if (pro == P2L_PRO_AFFINE_RELU) P2L_LAUNCH2(P2L_PRO_AFFINE_RELU, false);
    else P2L_LAUNCH2(P2L_PRO_AFFINE, false);
  }
#undef P2L_LAUNCH2
  return p2l_check_launch();
}

}  // namespace

// k must already carry n_mtiles / n_ntiles / tiles_*_log for the chosen cfg
// (cfg 0: 16x16-pixel tiles, cfg 1: 8x16), BN = 64, splitk = 1.
int p2l_launch_conv2(const ConvK& k, int pro, int ups, int cfg, hipStream_t st) {
  return cfg == 0 ? launch2<9, 0>(k, pro, ups, st) : launch2<9, 1>(k, pro, ups, st);
}
