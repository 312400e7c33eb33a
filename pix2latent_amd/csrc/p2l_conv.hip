// 3x3 / 1x1 convolution as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces nn.Conv2d forward and input-gradient on the pix2latent hot path
// (BigGAN-deep GenBlock convs, SelfAttn 1x1s, VGG16 features, StyleGAN2 modulated
// convs; reached from pix2latent/model/biggan.py:58, model/stylegan2.py:116-125 and
// pix2latent/loss_functions.py:142 in the reference).  Two arithmetic formats:
//   * P2L_WFMT_F32: v_mfma_f32_32x32x2_f32, exact fp32 FMA chains (bitwise a k-ordered
//     fmaf chain): results differ from the CPU oracle only by summation order.  Used by
//     all 1x1 convs and, on request, the 3x3 convs.
//   * P2L_WFMT_BF16X3 (default for 3x3 / sub-pixel): every fp32 operand split into three
//     bf16 pieces, six v_mfma_f32_32x32x16_bf16 cross products accumulated in fp32 -
//     fp32-level accuracy at 2.67x fewer matrix cycles (template flag BF3 below; weights
//     pre-split and streamed global -> LDS by global_load_lds_dwordx4).
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
//   * fp32 format: global->LDS goes through registers (prologue math + padded rows), loads
//     for chunk c+1 are issued before the MFMAs of chunk c.  bf16x3 format: only the
//     activation tile goes through registers (prologue + split); the weight tile is an
//     LDS-direct DMA in two tap halves.  Two blocks per CU (<= 73 KB LDS each) overlap one
//     block's barriers with the other's MFMAs.
//   * Small-M layers (4x4..16x16) split K over blockIdx.y and finish with a
//     deterministic reduce+epilogue kernel (no float atomics: CMA ranking must
//     be reproducible).
#include "p2l_conv_k.h"

#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <array>
#include <atomic>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

using namespace p2lconv;

namespace {

// Optional per-launch timing of the conv kernel (bench.py's roofline leg):
// hipEvents from a pre-created pool are recorded on the launch stream around
// every conv (incl. its split-K finish); nothing is allocated while enabled.
struct ConvProf {
  std::atomic<bool> on{false};
  int n = 0;
  std::vector<hipEvent_t> ev;
  std::vector<double> flops;    // algorithmic (direct convolution on the real channels)
  std::vector<double> xflops;   // executed on the matrix pipe (padded channels, 4 phase-taps)
  std::vector<int> nprod;       // 16-bit MFMA products per fp32 product: 6 bf16x3 | 3 fp16x2 | 16 = fp32 MFMA
  std::vector<double> bytes;    // algorithmic bytes: every operand once + the packed weights
  std::vector<double> wbytes;   // ... of which written (y, yp)
  std::vector<int> kind;
  std::vector<int> fam;         // P2L_PROF_FAM_*: which kernel took the launch
  std::vector<std::array<int, 10>> shape;   // taps B H W Cin Cout ups pro arb splitk
  std::string dump_path;                    // p2l_prof_dump
  int seq = 0, period = 1, phase = 0;       // sampling (p2l_prof_step)
  std::mutex mu;                            // launches of any host thread may claim a slot
};
// The one piece of process-wide state of the library: the OPT-IN launch profiler (off unless
// p2l_prof_begin was called).  Process-wide on purpose -- the backward pass of a torch program
// runs on the autograd engine's thread, and bench.py times those launches too -- and guarded:
// slot claims and begin / end take the mutex, a disabled profiler costs one relaxed load.
ConvProf& prof() {
  static ConvProf p;
  return p;
}
#define g_prof prof()

// BF3 = fp32-equivalent arithmetic on the bf16 matrix pipe (16x the fp32 MFMA rate):
// every fp32 operand is split into three bf16 pieces x = x1 + x2 + x3 (round-to-nearest
// residuals, |x - (x1+x2+x3)| <= 2^-24 |x|) and a.b is accumulated in fp32 from the six
// products a1b1 a1b2 a2b1 a2b2 a1b3 a3b1 (the dropped a2b3 a3b2 a3b3 are <= 2^-23 |ab|),
// i.e. 6 v_mfma_f32_32x32x16_bf16 (32 cycles each) instead of 8 v_mfma_f32_32x32x2_f32
// (64 cycles each) per 16 channels: 2.67x fewer matrix cycles at fp32-level accuracy.
// Activations are split while they are staged into LDS; weights are pre-split by
// p2l_pack_conv_weight_bf3.  LDS row = [x1 k0-7 | x1 k8-15 | x2 .. | x3 ..] = 96 bytes with
// the 16-byte chunk index XOR-ed with bit 3 of the row (16 consecutive rows then cover all
// 64 banks exactly once per ds_read_b128, no padding).  The C layout is the same as the
// fp32 instruction's, so the epilogues are shared.

constexpr bool kDmaWeights = true;   // bf16x3 3x3: weight tile by LDS-direct DMA (see the kernel)

template <int TAPS, int BN, int KC, int A_ITERS, int PRO, bool UPS, bool BF3 = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvK k) {
  static_assert(!BF3 || KC == 16, "bf16x3 path works on 16-channel chunks");
  constexpr int PITCH = BF3 ? 24 : KC + 4;   // floats per LDS row (BF3: 96 B, swizzled)
  constexpr int VPR = KC / 4;        // float4 per row
  constexpr int NT = BN / 32;        // accumulators per wave
  constexpr int B_ITEMS = BF3 ? TAPS * BN * 6 : TAPS * BN * VPR;   // 16-byte items
  constexpr int B_ITERS = (B_ITEMS + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int TW = 1 << k.tw_log, TH = 1 << k.th_log, TB = 1 << k.tb_log;
  const int HW_ = TW + 2, HH_ = TH + 2;
  const int HP = (TAPS != 1) ? k.hp : HW_;                  // LDS pitch of a patch line (rows)
  const int a_rows = (TAPS != 1) ? TB * HH_ * HW_ : 128;    // staged pixels
  const int a_rows_lds = (TAPS != 1) ? TB * HH_ * HP : 128; // LDS rows they occupy
  float* As = smem;
  float* Bs = smem + a_rows_lds * PITCH;

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
  const bool sp_fwd = (TAPS == 4) && k.sp_mode == 1;
  const bool sp_bwd = (TAPS == 4) && k.sp_mode == 2;
  const int z = sp_fwd ? 0 : blockIdx.y;
  const int ph_y = sp_fwd ? (int)(blockIdx.y >> 1) : 0, ph_x = sp_fwd ? (int)(blockIdx.y & 1) : 0;
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
      if (TAPS != 1) {
        tb = p / (HH_ * HW_);
        const int rem = p - tb * (HH_ * HW_);
        const int hy = rem / HW_, hx = rem - hy * HW_;
        iy = y0 + hy - 1;
        ix = x0 + hx - 1;
        a_loff[it] = ((tb * HH_ + hy) * HP + hx) * PITCH + v * 4;
      } else {
        const int Q = p >> 2, s = p & 3;
        const int qx = Q & ((TW >> 1) - 1);
        const int qy = (Q >> (k.tw_log - 1)) & ((TH >> 1) - 1);
        tb = Q >> (k.tw_log + k.th_log - 2);
        iy = y0 + 2 * qy + (s >> 1);
        ix = x0 + 2 * qx + (s & 1);
      }
      const int b = b0 + tb;
      if (b < k.B && iy >= 0 && iy < k.iH && ix >= 0 && ix < k.iW) {
        int pix;
        if (UPS)
          pix = (b * (k.H >> 1) + (iy >> 1)) * (k.W >> 1) + (ix >> 1);
        else if (sp_bwd)      // phase plane (0,0) of the high-res gradient buffer
          pix = (b * k.ibH + 2 * iy) * k.ibW + 2 * ix;
        else
          pix = (b * k.ibH + iy) * k.ibW + ix;
        a_goff[it] = pix * k.x_ld + v * 4;
        a_soff[it] = b * k.pro_bstride + v * 4;
        a_valid |= 1u << it;
      }
    }
  }

  // Prologue scale/shift: with one image per tile (always the case for the A_ITERS == 3
  // instantiations, enforced by the launcher) every item of a thread has the same
  // (sample, channel quarter), so ONE s/t pair per thread serves all its items.
  constexpr bool S_UNI = (A_ITERS == 3) && (TAPS != 1);
  constexpr int S_ITERS = S_UNI ? 1 : A_ITERS;
  const int s_uni = b0 * k.pro_bstride + (tid & (VPR - 1)) * 4;
  f32x4 xr[A_ITERS], sr[S_ITERS], tr[S_ITERS], wr[B_ITERS];

  // sub-pixel input-gradient: chunk c = (phase plane cls, channel chunk cc)
  auto load_regs = [&](int c) {
    int cc = c, wslab = 0, a_extra = 0;
    if (TAPS == 4) {
      if (sp_bwd) {
        const int cls = c / k.sp_ncc;
        cc = c - cls * k.sp_ncc;
        wslab = cls * 4;
        a_extra = ((cls >> 1) * k.ibW + (cls & 1)) * k.x_ld;
      } else {
        wslab = (ph_y * 2 + ph_x) * 4;
      }
    }
    const int ncc = (TAPS == 4 && sp_bwd) ? k.sp_ncc : k.nchunks;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      xr[it] = *reinterpret_cast<const f32x4*>(k.x + (size_t)(a_goff[it] + a_extra) + cc * KC);
      if (PRO != P2L_PRO_NONE && it < S_ITERS) {
        const int so = S_UNI ? s_uni : a_soff[it];
        sr[it] = *reinterpret_cast<const f32x4*>(k.pro_s + so + cc * KC);
        tr[it] = *reinterpret_cast<const f32x4*>(k.pro_t + so + cc * KC);
      }
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int j = tid + 256 * it;
      if (j < B_ITEMS) {
        if (BF3) {
          // tile-image layout: span t of this block = TAPS*32 rows of 32-channel tile
          // n0/32 + t, chunk cc; item j copies 16 bytes, LDS offset = global span offset
          constexpr int SPAN = TAPS * 32 * 6;                 // 16-byte items per span
          constexpr int SL = (TAPS == 4) ? 16 : TAPS;         // slabs per (chunk, tile)
          const int t = (NT > 1 && j >= SPAN) ? 1 : 0;
          const size_t off =
              ((((size_t)cc * (k.Cout >> 5) + (n0 >> 5) + t) * SL + wslab) * 32) * 24 +
              (size_t)(j - t * SPAN) * 4;
          wr[it] = *reinterpret_cast<const f32x4*>(k.w + off);
        } else {
          const int tap = j / (BN * VPR);
          const int rem = j - tap * (BN * VPR);  // row*VPR + v
          const size_t off =
              (((size_t)(wslab + tap) * ncc + cc) * k.Cout + n0) * KC + rem * 4;
          wr[it] = *reinterpret_cast<const f32x4*>(k.w + off);
        }
      }
    }
  };

  auto write_lds = [&]() {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      if (a_loff[it] >= 0) {
        f32x4 v = xr[it];
        if (PRO != P2L_PRO_NONE) {
          v = v * sr[S_UNI ? 0 : it] + tr[S_UNI ? 0 : it];
          if (PRO == P2L_PRO_AFFINE_RELU) {
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
          }
        }
        if (!((a_valid >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (BF3) {
          // a_loff = row*24 + v*4 (floats): row base + which quarter of the 16 channels
          const int row = a_loff[it] / 24, q4 = (a_loff[it] - row * 24) >> 2;
          bf16x4 ph, pm, pl;
          split3(v, ph, pm, pl);
          char* rb = reinterpret_cast<char*>(As) + row * 96 + (q4 & 1) * 8;
          // (c ^ s) with c = {0,2,4} + (q4 >> 1) and s in {0,1}: only the low bit moves,
          // so the three pieces are one address + 0 / 32 / 64 bytes
          char* rq = rb + bf3_chunk(q4 >> 1, row) * 16;
          *reinterpret_cast<bf16x4*>(rq) = ph;
          *reinterpret_cast<bf16x4*>(rq + 32) = pm;
          *reinterpret_cast<bf16x4*>(rq + 64) = pl;
        } else {
          *reinterpret_cast<f32x4*>(As + a_loff[it]) = v;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int j = tid + 256 * it;
      if (j < B_ITEMS) {
        if (BF3) {
          *reinterpret_cast<f32x4*>(Bs + j * 4) = wr[it];     // already in LDS order
        } else {
          const int row = j / VPR, v = j - row * VPR;  // row = tap*BN + n
          *reinterpret_cast<f32x4*>(Bs + row * PITCH + v * 4) = wr[it];
        }
      }
    }
  };

  // ---- fragment addressing ----------------------------------------------
  int a_row0;
  {
    const int i = wave * 32 + l31;
    if (TAPS != 1) {
      const int Q = i >> 2, s = i & 3;
      const int qx = Q & ((TW >> 1) - 1);
      const int qy = (Q >> (k.tw_log - 1)) & ((TH >> 1) - 1);
      const int tb = Q >> (k.tw_log + k.th_log - 2);
      a_row0 = (tb * HH_ + 2 * qy + (s >> 1)) * HP + 2 * qx + (s & 1);
    } else {
      a_row0 = i;
    }
  }
  const float* a_frag = As + a_row0 * PITCH + lhi * 4;
  const float* b_frag = Bs + l31 * PITCH + lhi * 4;
  // BF3: B rows are (j*TAPS + tap)*32 + l31 (multiples of 32 + l31): swizzle bit from l31 only
  const int b_c1 = bf3_chunk(lhi, l31) * 4, b_c2 = b_c1 + 8, b_c3 = b_c1 + 16;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- bf16x3 3x3 / sub-pixel: weights by LDS-direct DMA -----------------------------
  // The packed weights are an image of the LDS tile, so a lane's 16 bytes go global -> LDS
  // with global_load_lds_dwordx4: no staging registers, no ds_write, no index math.  The
  // tile is split by taps into two halves whose item counts are multiples of 256 (every wave
  // issues the same number of DMA instructions, so fixed vmcnt immediates are valid): half 0
  // of chunk c+1 is fetched while the taps of half 1 of chunk c are still being multiplied,
  // half 1 at the end of the chunk while the activation tile is split and written; each DMA
  // has at least half a chunk of MFMAs to land.  Only the activation tile still goes through
  // registers (it has to be split into bf16 pieces).
  constexpr bool DMA_B = BF3 && kDmaWeights && (TAPS == 9 || (TAPS == 4 && NT == 2));
  if constexpr (DMA_B) {
    constexpr int T0 = (TAPS == 9) ? 4 : 2;            // taps in half 0
    constexpr int NU = TAPS * NT, U0 = T0 * NT;        // MFMA units (tap-major), units in half 0
    constexpr int SL = (TAPS == 4) ? 16 : TAPS;        // slabs per (chunk, 32-channel tile)
    constexpr int H0 = U0 * 192;                       // 16-byte items in half 0
    static_assert(H0 % 256 == 0, "half 0 must be whole DMA instructions");
    constexpr int NH1_MIN = (B_ITEMS - H0) / 256;      // half-1 DMA instructions every wave issues
    constexpr int NA_LD = A_ITERS + ((PRO != P2L_PRO_NONE) ? 2 * S_ITERS : 0);
    // LDS order of the weight tile: [tap][n-tile j][32 rows] (= MFMA unit order)
    int b_goff[B_ITERS];
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int jj = tid + 256 * it;
      const int u = jj / 192, within = jj - u * 192;
      const int tap = u / NT, j = u - tap * NT;
      b_goff[it] = ((j * SL + tap) * 32) * 24 + within * 4;
    }
    auto geom = [&](int c, int& cc, int& wslab, int& a_extra) {
      cc = c; wslab = 0; a_extra = 0;
      if (TAPS == 4) {
        if (sp_bwd) {
          const int cls = c / k.sp_ncc;
          cc = c - cls * k.sp_ncc;
          wslab = cls * 4;
          a_extra = ((cls >> 1) * k.ibW + (cls & 1)) * k.x_ld;
        } else {
          wslab = (ph_y * 2 + ph_x) * 4;
        }
      }
    };
    auto dma_b = [&](int c, auto half_c) {
      constexpr int half = decltype(half_c)::value;
      int cc, wslab, a_extra;
      geom(c, cc, wslab, a_extra);
      const float* base = k.w + ((((size_t)cc * (k.Cout >> 5) + (n0 >> 5)) * SL + wslab) * 32) * 24;
#pragma unroll
      for (int it = 0; it < B_ITERS; ++it) {
        if ((half == 0 && 256 * it >= H0) || (half == 1 && 256 * (it + 1) <= H0)) continue;
        const int jj = tid + 256 * it;
        if (jj < B_ITEMS) {
          // inline asm on purpose: through the builtin hipcc treats the DMA as an LDS write
          // that may alias everything and puts s_waitcnt vmcnt(0) in front of the next
          // ds_read / ds_write, i.e. right after the issue.  The waits are placed by hand
          // below (P2L_WAIT); the compiler's own vmcnt arithmetic for the activation loads
          // ignores these instructions and therefore only ever over-waits.
          const float* src = base + b_goff[it];
          const unsigned lds_wave_base = __builtin_amdgcn_readfirstlane(
              (unsigned)(size_t)(__attribute__((address_space(3))) float*)(Bs + (size_t)(jj - lane) * 4));
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
        xr[it] = *reinterpret_cast<const f32x4*>(k.x + (size_t)(a_goff[it] + a_extra) + cc * KC);
        if (PRO != P2L_PRO_NONE && it < S_ITERS) {
          const int so = S_UNI ? s_uni : a_soff[it];
          sr[it] = *reinterpret_cast<const f32x4*>(k.pro_s + so + cc * KC);
          tr[it] = *reinterpret_cast<const f32x4*>(k.pro_t + so + cc * KC);
        }
      }
    };
    auto write_a = [&]() {
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        if (a_loff[it] >= 0) {
          f32x4 v = xr[it];
          if (PRO != P2L_PRO_NONE) {
            v = v * sr[S_UNI ? 0 : it] + tr[S_UNI ? 0 : it];
            if (PRO == P2L_PRO_AFFINE_RELU) {
              v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
              v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
          }
          if (!((a_valid >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
          const int row = a_loff[it] / 24, q4 = (a_loff[it] - row * 24) >> 2;
          bf16x4 ph, pm, pl;
          split3(v, ph, pm, pl);
          char* rb = reinterpret_cast<char*>(As) + row * 96 + (q4 & 1) * 8;
          // (c ^ s) with c = {0,2,4} + (q4 >> 1) and s in {0,1}: only the low bit moves,
          // so the three pieces are one address + 0 / 32 / 64 bytes
          char* rq = rb + bf3_chunk(q4 >> 1, row) * 16;
          *reinterpret_cast<bf16x4*>(rq) = ph;
          *reinterpret_cast<bf16x4*>(rq + 32) = pm;
          *reinterpret_cast<bf16x4*>(rq + 64) = pl;
        }
      }
    };
    // s_waitcnt immediate (gfx9 encoding): vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14
#define P2L_WAIT(VM, LGKM) __builtin_amdgcn_s_waitcnt(((VM) & 15) | (7 << 4) | ((LGKM) << 8) | (((VM) >> 4) << 14))
    using H0c = std::integral_constant<int, 0>;
    using H1c = std::integral_constant<int, 1>;
    // The activation loads of chunk c+2 are issued right after write_a() has consumed those
    // of chunk c+1 (NOT at the loop top: hipcc re-uses the destination registers for the
    // addresses and would put a vmcnt(0) there, which also waits for the DMAs in flight).
    if (c_begin < c_end) {
      dma_b(c_begin, H0c{});
      dma_b(c_begin, H1c{});
      load_a(c_begin);
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
      if (TAPS == 4) {
        int oy = ph_y, ox = ph_x;
        if (sp_bwd) {
          const int cls = c / k.sp_ncc;
          oy = 1 - (cls >> 1);
          ox = 1 - (cls & 1);
        }
        win_row = oy * HP + ox;
      }
      bf16x8 af[2][3], bq[2][3];
      auto lda = [&](int tap, bf16x8 (&a)[3]) {
        const int dy = (TAPS == 9) ? tap / 3 : (tap >> 1);
        const int dx = (TAPS == 9) ? tap - dy * 3 : (tap & 1);
        const int arow = a_row0 + win_row + dy * HP + dx;
        const float* ar = As + arow * 24;
        const float* aq = ar + bf3_chunk(lhi, arow) * 4;      // pieces at +0 / +32 / +64 bytes
        a[0] = *reinterpret_cast<const bf16x8*>(aq);
        a[1] = *reinterpret_cast<const bf16x8*>(aq + 8);
        a[2] = *reinterpret_cast<const bf16x8*>(aq + 16);
      };
      auto ldb = [&](int u, bf16x8 (&b)[3]) {
        const float* br = Bs + (u * 32 + l31) * 24;
        b[0] = *reinterpret_cast<const bf16x8*>(br + b_c1);
        b[1] = *reinterpret_cast<const bf16x8*>(br + b_c2);
        b[2] = *reinterpret_cast<const bf16x8*>(br + b_c3);
      };
      lda(0, af[0]);
      ldb(0, bq[0]);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int tap = u / NT, j = u - tap * NT;
        if (u == U0) {
          // half 1 of THIS chunk (DMA'd at the end of the previous one, the youngest VMEM op)
          // must have landed in every wave, and every wave must be done reading half 0
          // before it is refilled
          P2L_WAIT(0, 0);
          __builtin_amdgcn_s_barrier();
          if (more) dma_b(c + 1, H0c{});
          if (j == 0) lda(tap, af[tap & 1]);
          ldb(u, bq[u & 1]);
        }
        if (u + 1 < NU && u + 1 != U0) {
          const int tn = (u + 1) / NT, jn = (u + 1) - tn * NT;
          if (jn == 0) lda(tn, af[tn & 1]);
          ldb(u + 1, bq[(u + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 (&a)[3] = af[tap & 1];
        const bf16x8 (&b)[3] = bq[u & 1];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      P2L_WAIT(63, 0);                      // my LDS reads are done
      __builtin_amdgcn_s_barrier();         // everybody is done with half 1 and the A tile
      if (more) {
        write_a();
        __builtin_amdgcn_sched_barrier(0);
        const bool more2 = (c + 2 < c_end);
        if (more2) load_a(c + 2);
        __builtin_amdgcn_sched_barrier(0);
        dma_b(c + 1, H1c{});
        __builtin_amdgcn_sched_barrier(0);
        // A tile written (lgkmcnt 0) and half 0 of c+1 landed; the ops issued after it - the
        // activation loads of c+2 (if any) and the half-1 DMAs - may still be in flight
        if (more2) P2L_WAIT(NA_LD + NH1_MIN, 0); else P2L_WAIT(NH1_MIN, 0);
      } else {
        P2L_WAIT(63, 0);
      }
      __builtin_amdgcn_s_barrier();
    }
#undef P2L_WAIT
  } else {
  if (c_begin < c_end) {
      load_regs(c_begin);
      write_lds();
    }
    __syncthreads();
  
    for (int c = c_begin; c < c_end; ++c) {
      const bool more = (c + 1 < c_end);
      if (more) load_regs(c + 1);
  
      // window origin inside the halo: 3x3 -> (0,0); sub-pixel forward -> the output
      // phase; sub-pixel input-gradient -> (1 - plane parity)
      int win0 = 0;
      if (TAPS == 4) {
        int oy = ph_y, ox = ph_x;
        if (sp_bwd) {
          const int cls = c / k.sp_ncc;
          oy = 1 - (cls >> 1);
          ox = 1 - (cls & 1);
        }
        win0 = (oy * HP + ox) * PITCH;
      }
      if constexpr (BF3) {
        // Fragment software pipeline.  Unit u = (tap, n-tile j): 6 MFMAs (192 issue cycles)
        // on acc[j].  While unit u issues, the 3 weight fragments of unit u+1 (and, at a tap
        // boundary, the 3 activation fragments of the next tap) are already in flight into the
        // other register set, so the only LDS wait per unit has a whole unit of MFMAs in front
        // of it.  (hipcc's own schedule of the straightforward loop reads each fragment right
        // before its first use: 5-6 exposed LDS round trips per tap, MFMA pipe 47 % busy.)
        constexpr int NU = TAPS * NT;
        bf16x8 af[2][3], bq[2][3];
        auto lda = [&](int tap, bf16x8 (&a)[3]) {
          const int dy = (TAPS == 9) ? tap / 3 : (tap >> 1);
          const int dx = (TAPS == 9) ? tap - dy * 3 : (tap & 1);
          const int arow = a_row0 + ((TAPS != 1) ? win0 / PITCH + dy * HP + dx : 0);
          const float* ar = As + arow * 24;
          const float* aq = ar + bf3_chunk(lhi, arow) * 4;
          a[0] = *reinterpret_cast<const bf16x8*>(aq);
          a[1] = *reinterpret_cast<const bf16x8*>(aq + 8);
          a[2] = *reinterpret_cast<const bf16x8*>(aq + 16);
        };
        auto ldb = [&](int tap, int j, bf16x8 (&b)[3]) {
          const float* br = Bs + ((j * TAPS + tap) * 32 + l31) * 24;     // [n-tile][tap][row]
          b[0] = *reinterpret_cast<const bf16x8*>(br + b_c1);
          b[1] = *reinterpret_cast<const bf16x8*>(br + b_c2);
          b[2] = *reinterpret_cast<const bf16x8*>(br + b_c3);
        };
        lda(0, af[0]);
        ldb(0, 0, bq[0]);
  #pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int tap = u / NT, j = u - tap * NT;
          if (u + 1 < NU) {
            const int tn = (u + 1) / NT, jn = (u + 1) - tn * NT;
            if (jn == 0) lda(tn, af[tn & 1]);
            ldb(tn, jn, bq[(u + 1) & 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 (&a)[3] = af[tap & 1];
          const bf16x8 (&b)[3] = bq[u & 1];
          // smallest terms first
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
  #pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int dy = (TAPS == 9) ? tap / 3 : (tap >> 1);
        const int dx = (TAPS == 9) ? tap - dy * 3 : (tap & 1);
        if (BF3) {
          const int arow = a_row0 + ((TAPS != 1) ? win0 / PITCH + dy * HP + dx : 0);
          const float* ar = As + arow * 24;
          const float* aq = ar + bf3_chunk(lhi, arow) * 4;
          const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(aq);
          const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(aq + 8);
          const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(aq + 16);
          bf16x8 b1[NT], b2[NT], b3[NT];
  #pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float* br = Bs + (tap * BN + j * 32 + l31) * 24;
            b1[j] = *reinterpret_cast<const bf16x8*>(br + b_c1);
            b2[j] = *reinterpret_cast<const bf16x8*>(br + b_c2);
            b3[j] = *reinterpret_cast<const bf16x8*>(br + b_c3);
          }
          // smallest terms first; the NT accumulators alternate so that consecutive MFMAs
          // never depend on each other
  #pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1[j], acc[j], 0, 0, 0);
  #pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3[j], acc[j], 0, 0, 0);
  #pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2[j], acc[j], 0, 0, 0);
  #pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1[j], acc[j], 0, 0, 0);
  #pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2[j], acc[j], 0, 0, 0);
  #pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[j], acc[j], 0, 0, 0);
          continue;
        }
        const float* ap = a_frag + ((TAPS != 1) ? win0 + (dy * HP + dx) * PITCH : 0);
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
      }  // !BF3
      __syncthreads();
      if (more) write_lds();
      __syncthreads();
    }
  
  }   // !DMA_B

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
    epilogue_vec<NT>(k, acc, smem, wave, lane, b0, y0, x0, n0, tile_in_image,
                     sp_fwd ? 1 : 0, ph_y, ph_x);
  }
}

// Deterministic split-K finish.  Item = (quad, channel); ZP = 4 lanes share an item, lane z
// sums slabs z, z+4, ... (independent loads in flight instead of one serial chain of
// splitk x 4 dependent adds), the four partial sums are combined in a FIXED shuffle order
// ((z0 + z1) + (z2 + z3)) and the z = 0 lane runs the epilogue.  Lanes 0-15 of a wave are 16
// consecutive items (channels: coalesced), the four lanes of an item are l, l^16, l^32, l^48.
__global__ __launch_bounds__(256) void conv_splitk_finish(const ConvK k) {
  const int Hh = k.H >> 1, Wh = k.W >> 1;
  const size_t total = (size_t)k.B * Hh * Wh * k.n_store;
  // wave = 16 items x 4 z-parts; item index of this lane
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int zp = lane >> 4;
  const size_t idx = ((size_t)blockIdx.x * 4 + wave) * 16 + (lane & 15);
  const bool live = idx < total;
  const size_t id2 = live ? idx : 0;
  const int n = (int)(id2 % k.n_store);
  size_t q = id2 / k.n_store;
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
    for (int zz = zp; zz < k.splitk; zz += 4)
      acc += k.ws[((size_t)zz * mtot + pix) * k.Cout + n];
    // fixed-order combine of the 4 z-parts (lanes l, l^16 | l^32, l^48)
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    a[s] = acc;
  }
  // the lanes that own an item (zp == 0) run the epilogue; every lane stays for the maxima
  float mx = 0.f, mxp = 0.f;          // max |value stored to y| / |... to yp| of this lane's item
  if (live && zp == 0) {
  if (k.arb_x == nullptr) {
    epilogue_quad<false>(k, a, (b * k.H + 2 * qy) * k.W + 2 * qx, b, 2 * qy, 2 * qx, n,
                         k.bias ? k.bias[n] : 0.f, &mx, &mxp);
  } else {
  // fused backward of relu(x*s+t) on the finished input gradient (same arithmetic as the
  // ARB branch of epilogue_vec, one channel per lane); the per-(sample, channel) sums are
  // written per QUAD (arb_nblk = quads per image) and reduced by p2l_arb_finish
  const float sc = k.arb_s[(size_t)b * k.arb_bstride + n];
  const float tc = k.arb_t[(size_t)b * k.arb_bstride + n];
  const bool pool_sum = k.pool == P2L_POOL_SUM;
  const bool has_skip = k.arb_skip && n < k.arb_skip_C;
  float v[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) v[s] = k.alpha * a[s];
  if (pool_sum) v[0] = (v[0] + v[1]) + (v[2] + v[3]);
  float* dst = pool_sum ? k.yp : k.y;
  const unsigned dld = (unsigned)(pool_sum ? k.yp_ld : k.y_ld);
  const int Wo = pool_sum ? Wh : k.W, Ho = pool_sum ? Hh : k.H;
  float sgx = 0.f, sg = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s < (pool_sum ? 1 : 4)) {
      const int yy = pool_sum ? qy : 2 * qy + (s >> 1), xx = pool_sum ? qx : 2 * qx + (s & 1);
      const unsigned pix = (unsigned)((b * Ho + yy) * Wo + xx);
      const float xv = k.arb_x[(size_t)pix * k.arb_x_ld + n];
      const float pre = xv * sc + tc;
      const float g = (k.arb_nomask || pre > 0.f) ? v[s] : 0.f;
      float o = g * sc;
      if (has_skip) {
        const unsigned ld = (unsigned)k.arb_skip_ld;
        if (k.arb_skip_ups) {
          const unsigned W2 = 2u * (unsigned)Wo;
          const unsigned cq = ((unsigned)(b * 2 * Ho + 2 * yy)) * W2 + 2u * (unsigned)xx;
          o += (k.arb_skip[(size_t)cq * ld + n] + k.arb_skip[(size_t)(cq + 1) * ld + n]) +
               (k.arb_skip[(size_t)(cq + W2) * ld + n] + k.arb_skip[(size_t)(cq + W2 + 1) * ld + n]);
        } else {
          o += k.arb_skip[(size_t)pix * ld + n];
        }
      }
      dst[(size_t)pix * dld + n] = o;
      if (pool_sum) mxp = fmaxf(mxp, fabsf(o)); else mx = fmaxf(mx, fabsf(o));
      sgx += g * xv;
      sg += g;
    }
  }
  const size_t po = ((size_t)b * k.arb_nblk + (size_t)qy * Wh + qx) * k.Cout + n;
  k.arb_partial[po] = sgx;
  k.arb_partial[(size_t)k.B * k.arb_nblk * k.Cout + po] = sg;
  }
  }
  // one partial maximum per BLOCK (64 items) for the launch that reads the tensor next (P2LAmax; the
  // launcher sets the pointers only when a block's 64 items lie in one image)
  if (k.amax_out != nullptr || k.amax_outp != nullptr) {
    __shared__ float red[8];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mxp = fmaxf(mxp, __shfl_xor(mxp, o, 64)); }
    if (lane == 0) { red[wave] = mx; red[4 + wave] = mxp; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const size_t first = (size_t)blockIdx.x * 64;                // first item of the block
      const size_t per_image = (size_t)Hh * Wh * k.n_store;
      if (first < total) {
        const size_t bb = first / per_image;
        const size_t slot = bb * k.amax_out_n + (first - bb * per_image) / 64;
        if (k.amax_out != nullptr) k.amax_out[slot] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (k.amax_outp != nullptr) k.amax_outp[slot] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
      }
    }
  }
}

// Finish of a K-sliced Winograd launch (2 or 4 slices of [B,H,W,Cout] partial outputs): item =
// one 2x2 quad x FOUR consecutive channels, every access a float4 (the scalar kernel above is
// made for the 4^2 ... 16^2 split-K layers: 4-byte accesses, 0.8 TB/s on a 9 MB output).  Slices
// are added in the same fixed order, ((z0 + z1) + (z2 + z3)), then the shared epilogue item runs:
// bias / residual / activation / mask / pooling, or the fused activation backward with one
// partial sum per quad (arb_nblk = quads per image, as the scalar finish writes them).
__global__ __launch_bounds__(256) void conv_splitk_finish4(const ConvK k) {
  const int Hh = k.H >> 1, Wh = k.W >> 1, C4 = k.n_store >> 2;
  const size_t total = (size_t)k.B * Hh * Wh * C4;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx % C4) * 4;
  size_t q = idx / C4;
  const int qx = (int)(q % Wh);
  q /= Wh;
  const int qy = (int)(q % Hh);
  const int b = (int)(q / Hh);
  const size_t mtot = (size_t)k.B * k.H * k.W;
  f32x4 v[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const size_t pix = ((size_t)b * k.H + 2 * qy + (s >> 1)) * k.W + 2 * qx + (s & 1);
    f32x4 z[4];
#pragma unroll
    for (int zz = 0; zz < 4; ++zz)
      z[zz] = zz < k.splitk ? *reinterpret_cast<const f32x4*>(k.ws + ((size_t)zz * mtot + pix) * k.Cout + n)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
    v[s] = ((z[0] + z[1]) + (z[2] + z[3])) * k.alpha;
  }
  EpiSums S;
  epi_item(k, v, b, 2 * qy, 2 * qx, n, 0, 0, 0, S);
  if (k.arb_x != nullptr) {
    const size_t po = ((size_t)b * k.arb_nblk + (size_t)qy * Wh + qx) * k.Cout + n;
    *reinterpret_cast<f32x4*>(k.arb_partial + po) = S.sgx;
    *reinterpret_cast<f32x4*>(k.arb_partial + (size_t)k.B * k.arb_nblk * k.Cout + po) = S.sg;
  }
  // one partial maximum per wave for the launch that reads the tensor next (P2LAmax; the launcher
  // sets the pointers only when a wave's 64 items lie in one image: whole waves, no early exit)
  if (k.amax_out != nullptr || k.amax_outp != nullptr) {
    float m = S.amax, mp = S.amaxp;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); mp = fmaxf(mp, __shfl_xor(mp, o, 64)); }
    if ((threadIdx.x & 63) == 0) {
      const size_t per_image = (size_t)Hh * Wh * C4;
      const size_t slot = (size_t)b * k.amax_out_n + (idx - (size_t)b * per_image) / 64;
      if (k.amax_out != nullptr) k.amax_out[slot] = m;
      if (k.amax_outp != nullptr) k.amax_outp[slot] = mp;
    }
  }
}

// The split-K finish of the direct / pointwise kernels with 16-byte accesses (round 4; the scalar kernel
// above moves 64-byte segments: 2 TB/s on tensors of a few MB).  Item = one 2x2 quad x FOUR consecutive
// channels; wave = 16 items x 4 z-parts, lane z sums slabs z, z+4, ... and the four partial sums meet in
// the SAME fixed order ((z0 + z1) + (z2 + z3)): bit-identical sums.  The epilogue is the shared item
// (epi_item), the maxima one partial per block of 64 items.
__global__ __launch_bounds__(256) void conv_splitk_finish_v4(const ConvK k) {
  const int Hh = k.H >> 1, Wh = k.W >> 1, C4 = k.n_store >> 2;
  const size_t total = (size_t)k.B * Hh * Wh * C4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int zp = lane >> 4;
  const size_t idx = ((size_t)blockIdx.x * 4 + wave) * 16 + (lane & 15);
  const bool live = idx < total;
  const size_t id2 = live ? idx : 0;
  const int n = (int)(id2 % C4) * 4;
  size_t q = id2 / C4;
  const int qx = (int)(q % Wh);
  q /= Wh;
  const int qy = (int)(q % Hh);
  const int b = (int)(q / Hh);
  const size_t mtot = (size_t)k.B * k.H * k.W;
  f32x4 v[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const size_t pix = ((size_t)b * k.H + 2 * qy + (s >> 1)) * k.W + 2 * qx + (s & 1);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int zz = zp; zz < k.splitk; zz += 4)
      acc += *reinterpret_cast<const f32x4*>(k.ws + ((size_t)zz * mtot + pix) * k.Cout + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = acc[e];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      acc[e] = t;
    }
    v[s] = acc * k.alpha;
  }
  EpiSums S;
  if (live && zp == 0) {
    epi_item(k, v, b, 2 * qy, 2 * qx, n, 0, 0, 0, S);
    if (k.arb_x != nullptr) {
      const size_t po = ((size_t)b * k.arb_nblk + (size_t)qy * Wh + qx) * k.Cout + n;
      *reinterpret_cast<f32x4*>(k.arb_partial + po) = S.sgx;
      *reinterpret_cast<f32x4*>(k.arb_partial + (size_t)k.B * k.arb_nblk * k.Cout + po) = S.sg;
    }
  }
  if (k.amax_out != nullptr || k.amax_outp != nullptr) {
    __shared__ float red[8];
    float mx = S.amax, mxp = S.amaxp;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mxp = fmaxf(mxp, __shfl_xor(mxp, o, 64)); }
    if (lane == 0) { red[wave] = mx; red[4 + wave] = mxp; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const size_t first = (size_t)blockIdx.x * 64;                // first item of the block
      const size_t per_image = (size_t)Hh * Wh * C4;
      if (first < total) {
        const size_t bb = first / per_image;
        const size_t slot = bb * k.amax_out_n + (first - bb * per_image) / 64;
        if (k.amax_out != nullptr) k.amax_out[slot] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (k.amax_outp != nullptr) k.amax_outp[slot] = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
      }
    }
  }
}

// src is OIHW [O][I][taps].  flip=0 packs the conv I->O (K=I, N=O); flip=1 packs
// its input-gradient conv O->I (K=O, N=I, taps mirrored).
// bf16x3 packed element: row (idx / 16) holds [x1 k0-15 | x2 k0-15 | x3 k0-15] (96 bytes)
// bf16x3 packed weights are an IMAGE of the kernel's LDS weight tile, so staging them is a
// flat copy (no per-item index arithmetic in the K loop):
//   [16-channel chunk q][32-channel tile n/32][slab (tap, or phase*4+tap)][row n%32][96 bytes]
// with the 96-byte row = [x1 k0-7 | x1 k8-15 | x2 .. | x3 ..] and the 16-byte chunk index
// already XOR-swizzled by bit 3 of the row (bf3_chunk).  A block reads BN/32 contiguous spans
// of TAPS*32 rows.
__device__ __forceinline__ void store_bf3(float* dst, int n_slabs, int N_pad, int slab, int q, int n,
                                          int kk, float v) {
  const __bf16 h = (__bf16)v;
  const float r1 = v - (float)h;
  const __bf16 m = (__bf16)r1;
  const __bf16 l = (__bf16)(r1 - (float)m);
  const int r = n & 31;
  const size_t row = (((size_t)q * (N_pad >> 5) + (n >> 5)) * n_slabs + slab) * 32 + r;
  __bf16* rp = reinterpret_cast<__bf16*>(dst) + row * 48;
  const int hi = kk >> 3, lo = kk & 7;
  rp[bf3_chunk(0 + hi, r) * 8 + lo] = h;
  rp[bf3_chunk(2 + hi, r) * 8 + lo] = m;
  rp[bf3_chunk(4 + hi, r) * 8 + lo] = l;
}

__global__ __launch_bounds__(256) void pack_conv_weight_kernel(
    const float* __restrict__ src, float* __restrict__ dst, int O, int I,
    int taps, int N_pad, int K_pad, int kc, int flip, int bf3) {
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
  if (bf3) store_bf3(dst, taps, N_pad, tap, q, n, kk, v);
  else dst[idx] = v;
}

// Sub-pixel weights.  dst layout [16 slabs = phase*4 + tap][K_pad/16][N_pad][16]; each
// slab is a sum of original 3x3 taps selected by sp_tapset (a 3-bit mask over dy / dx).
// mode 0: 3x3 conv on a nearest-x2 upsampled input
//   forward: S(0,0)={0} S(0,1)={1,2} S(1,0)={0,1} S(1,1)={2};
//   input-gradient (flip): plane row offset u = 2i - pu: D(2)={0} D(0)={1,2} D(1)={0,1} D(-1)={2}
// mode 1: stride-2 transposed 3x3 conv (StyleGAN2 up-conv: out[2p+k] += x[p] W[k])
//   forward: phase 0 window {p-1,p} <- taps {2},{0}; phase 1 window {p,p+1} <- {1},{}
//   input-gradient: plane 0 window {p,p+1} <- {0},{2}; plane 1 window {p-1,p} <- {},{1}
__device__ __forceinline__ unsigned sp_tapset(int mode, int flip, int p, int i) {
  if (mode == 0) {
    if (!flip) {
      if (p == 0) return i == 0 ? 1u : 6u;
      return i == 0 ? 3u : 4u;
    }
    const int u = 2 * i - p;      // p = plane parity
    return u == 2 ? 1u : u == 0 ? 6u : u == 1 ? 3u : 4u;
  }
  if (!flip) {
    if (p == 0) return i == 0 ? 4u : 1u;
    return i == 0 ? 2u : 0u;
  }
  if (p == 0) return i == 0 ? 1u : 4u;
  return i == 0 ? 0u : 2u;
}
__global__ __launch_bounds__(256) void pack_subpix_kernel(const float* __restrict__ src,
                                                          float* __restrict__ dst, int O, int I,
                                                          int N_pad, int K_pad, int flip,
                                                          int mode, int bf3) {
  const size_t per = (size_t)K_pad * N_pad;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= 16 * per) return;
  const int slab = (int)(idx / per);
  size_t r = idx - (size_t)slab * per;
  const int kk = (int)(r % 16);
  r /= 16;
  const int n = (int)(r % N_pad);
  const int q = (int)(r / N_pad);
  const int c = q * 16 + kk;
  const int ph = slab >> 2, tap = slab & 3;
  const unsigned my = sp_tapset(mode, flip, ph >> 1, tap >> 1);
  const unsigned mx = sp_tapset(mode, flip, ph & 1, tap & 1);
  float v = 0.f;
  const bool ok = flip ? (n < I && c < O) : (n < O && c < I);
  if (ok) {
    for (int dy = 0; dy < 3; ++dy)
      for (int dx = 0; dx < 3; ++dx)
        if (((my >> dy) & 1u) && ((mx >> dx) & 1u))
          v += flip ? src[((size_t)c * I + n) * 9 + dy * 3 + dx]
                    : src[((size_t)n * I + c) * 9 + dy * 3 + dx];
  }
  if (bf3) store_bf3(dst, 16, N_pad, slab, q, n, kk, v);
  else dst[idx] = v;
}

// The GEMM M grid: output pixels, except in the sub-pixel modes (ups 2 / 3) where it is
// the LOW-RES pixel grid (d->H, d->W are always the high-res dims there); ext = 1 adds
// the extra row/column of a stride-2 transposed conv (ups 2: grid H/2+1).
static inline int npow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
void grid_dims(const P2LConv* d, int& gH, int& gW) {
  const int sh = (d->ups >= 2) ? 1 : 0;
  gH = d->H >> sh; gW = d->W >> sh;
  if (d->ups == 2 && d->ext) { gH += 1; gW += 1; }
}
int choose_tile(const P2LConv* d, ConvK& k) {
  int gH, gW;
  grid_dims(d, gH, gW);
  if (d->H < 4 || d->W < 4 || gH < 2 || gW < 2) return P2L_EINVAL;
  if (d->ups < 2 && ((d->H & 1) || (d->W & 1))) return P2L_EINVAL;   // quads need even dims
  int TW = npow2(gW) < 16 ? npow2(gW) : 16;
  int TH = 128 / TW;
  if (TH > npow2(gH)) TH = npow2(gH);
  int TB = 128 / (TW * TH);
  k.tw_log = ilog2(TW);
  k.th_log = ilog2(TH);
  k.tb_log = ilog2(TB);
  k.tiles_x = cdiv(gW, TW);
  k.tiles_y = cdiv(gH, TH);
  k.tiles_x_log = ilog2(k.tiles_x);      // only meaningful for power-of-two grids (v2 kernel)
  k.tiles_y_log = ilog2(k.tiles_y);
  k.n_mtiles = k.tiles_x * k.tiles_y * cdiv(d->B, TB);
  k.partial = (gW % TW != 0) || (gH % TH != 0);
  return P2L_OK;
}

// LDS pitch (rows) of one line of the staged input patch (ConvK::hp).  Brute-forced over the
// ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27}, ...) for the 2x2-quad pixel order and
// the 96-byte swizzled row: 16-wide tiles -> 24 (1.0 LDS cycle per lane group instead of 2.0
// at 18), 8-wide -> 12 (2.0 instead of 3.0); 23 KB of patch instead of 17 KB still leaves
// two blocks per CU (78.3 KB each).
int halo_pitch(int TW, bool bf3) {
  if (!bf3) return TW + 2;
  return TW == 16 ? 24 : (TW == 8 ? 12 : TW + 2);
}
// ... of the fp16 x 2 kernel's 64-byte rows (tools/h2_banks.py: 16- and 8-wide tiles are conflict
// free at 24 rows per line, 2.0 LDS cycles per lane group at best for the 4-wide tiles of the 4^2 layers)
int halo_pitch_h2(int TW) { return TW >= 8 ? 24 : (TW == 4 ? 8 : TW + 2); }

// Winograd form (p2l_wino.hip).  Which launches take it is a function of the LAYER SHAPE only,
// never of the batch: a candidate's result must not depend on how many others share its chunk
// (tests/test_fullsize_gpu.py), so the same layer runs the same kernel for every B - and for
// these shapes split-K is never suggested or honoured.  Conditions: weights carry the
// transform-domain image (P2L_WFMT_BF16X3W), plain stride-1 3x3 (no upsample fusion), whole
// 8x16-pixel x 64-channel blocks, and at least 64 such blocks PER IMAGE: measured per layer
// inside the bench step (profiles/round2_layers_*.txt) the Winograd kernel is 1.05-1.29x the
// direct one on those, and 0.92-1.10x on the 16^2 / 32^2 layers with fewer blocks (its longer
// prologue / epilogue against rounds of the chip that are mostly empty).
// P2LConv.form: P2L_FORM_NO_WINO keeps the direct kernel, P2L_FORM_WINO_ANY takes every eligible
// shape (tests: small grids too).
// K slices of a small-grid Winograd layer (shape only; 1 = none)
static int wino_split(const P2LConv* d) {
  if (d->wfmt != P2L_WFMT_BF16X3W || d->taps != 9 || d->ups != 0 || (d->form & P2L_FORM_NO_WINO)) return 1;
  if (d->x_ld % 4 || !p2l_wino_weight_ok(d->Cout, d->Cin) || (d->form & P2L_FORM_WINO_8X16)) return 1;
  return p2l_wino_split_factor(d->H, d->W, d->Cin, d->Cout);
}
static bool wino_shape(const P2LConv* d) {
  if (d->wfmt != P2L_WFMT_BF16X3W || d->taps != 9 || d->ups != 0 || (d->form & P2L_FORM_NO_WINO)) return false;
  if (d->H % 8 || d->W % 16 || d->x_ld % 4 || !p2l_wino_weight_ok(d->Cout, d->Cin)) return false;
  // 64 input channels = 4 chunks per block: the 16x16 kernel's prologue + epilogue (16.5 k cycles, one
  // block per CU) are more than its loop (14 k), and the direct fp16 x 2 kernel -- three blocks per CU
  // that overlap each other's epilogues -- is as fast per launch (tools/bench_h2.py: 256^2 0.387 vs
  // 0.393 ms, 128^2 0.093 vs 0.100) and +0.7 ... 1.7 % in the step at 18 / 9 candidates, neutral
  // below (same-box A/B, round 4); 128-channel layers stay (1.22x for Winograd)
  if (d->Cin <= 64 && !(d->form & P2L_FORM_WINO_ANY)) return false;
  const int per_image = (d->H / 8) * (d->W / 16) * (d->Cout / 64);
  if ((d->form & P2L_FORM_WINO_ANY) || per_image >= 64) return true;
  // small-grid layers: in the K-sliced form only, i.e. when the caller passes the slice count
  // p2l_conv_suggest_splitk gives for the shape (with it a 48-block launch at 2-3 candidates per
  // GPU becomes 96-192 short blocks; unsliced, a threshold of 32 measured +1 % at 18 candidates
  // and -2...-4 % at 2-3)
  const int s = wino_split(d);
  return s > 1 && d->splitk == s;
}

// full-tile form of the 1x1 conv on the 16-bit pipe (p2l_pw.hip: bf16 x 3, or fp16 x 2 when the
// launch gets maxima): weights carry the pre-split images (P2L_WFMT_PW), whole 128-pixel tiles of one
// image, 64-channel tiles.  Like the Winograd form a function of the layer shape only; the 4^2 ..
// 16^2 layers take the small-grid fp16 x 2 form of the same kernel (pw_small_h2 below: multi-image
// tiles, split-K slices).  P2L_FORM_NO_PW keeps the exact-fp32 kernel.
static bool pw_shape(const P2LConv* d) {
  if ((d->form & P2L_FORM_NO_PW) || d->wfmt != P2L_WFMT_PW || d->taps != 1 || d->ups != 0) return false;
  if (d->Cin % 64 || d->Cout % 64 || d->x_ld % 4 || d->H % 8 || d->W % 16) return false;
  // measured per layer inside the bench step: 1.2-1.9x the exact-fp32 kernel from 256 input
  // channels up, 1.07-1.39x with 64 / 128 (32-channel stages, four blocks per CU)
  return d->H * d->W >= 1024;
}

// 3-channel image convs (p2l_thin.hip): 0 thin output, 1 thin input, -1 the generic kernel.
// Shape + format only (the epilogue conditions are checked at the launch).
static int thin_shape(const P2LConv* d) {
  if ((d->form & P2L_FORM_NO_THIN) || d->wfmt != P2L_WFMT_BF16X3T || d->taps != 9 || d->ups != 0) return -1;
  if (d->H % 8 || d->W % 16 || d->x_ld % 4 || d->Cin % 16) return -1;
  return p2l_thin_mode(d->Cout, d->Cin);
}
// either bf16x3 pointwise kernel: no split-K
static bool pw_any(const P2LConv* d) { return pw_shape(d) || thin_shape(d) >= 0; }

// fp16 x 2 form of the direct 3x3 / sub-pixel kernel (p2l_h2.hip): every P2L_WFMT_BF16X3W launch that
// the Winograd kernel does not take (sub-pixel up-convs, the 4^2 ... 16^2 layers in split-K slices,
// H / W not multiples of 16) -- a function of the layer shape and the format; P2L_FORM_WINO_BF3
// ("bf16 x 3 instead of fp16 x 2") keeps the bf16 x 3 kernel.  The launch also needs the 256 B per
// image of workspace p2l_conv_workspace_bytes asks for.
static bool direct_h2(const P2LConv* d) {
#ifdef P2L_AB_NO_DIRECT_H2      // A/B builds (tools/ab_build.sh): bisecting an arithmetic difference by kernel family
  return false;
#endif
  if (d->wfmt != P2L_WFMT_BF16X3W || d->taps != 9 || (d->form & P2L_FORM_WINO_BF3)) return false;
  if (d->ups == 1 || d->x_ld % 4 || d->Cin % 16 || d->Cout % 32) return false;
  return !wino_shape(d);
}

// ... and of the 1x1 kernel's small-grid form (p2l_pw.hip, pw_h2_kernel<.., SM>): the 4^2 ... 16^2
// layers of P2L_WFMT_PW weights that the full-tile pointwise kernels do not take (multi-image tiles,
// split-K slices).  Shape and format only.
static bool pw_small_h2(const P2LConv* d) {
#ifdef P2L_AB_NO_PWSMALL_H2
  return false;
#endif
  if (d->wfmt != P2L_WFMT_PW || d->taps != 1 || d->ups != 0 || pw_shape(d)) return false;
  if (d->form & (P2L_FORM_NO_PW | P2L_FORM_WINO_BF3)) return false;
  if (d->Cin % 32 || d->Cout % 64 || d->x_ld % 4) return false;
  ConvK k{};
  return choose_tile(d, k) == P2L_OK && !k.partial;
}

// Output-channel tile: 64 unless the grid then leaves CUs idle in its last
// round.  All blocks of a launch do the same MFMA work and co-resident blocks
// share a CU's matrix pipes, so time ~ ceil(blocks / 256 CUs) * work-per-block.
int choose_bn(const P2LConv* d, int n_mtiles) {
  if (d->Cout % 64) return 32;
  const int ph = (d->ups == 2) ? 4 : 1;      // sub-pixel forward: 4 phases per tile
  const int n64 = n_mtiles * (d->Cout / 64) * ph, n32 = n_mtiles * (d->Cout / 32) * ph;
  if (n64 < 256 && d->ups >= 2) return 32;   // sub-pixel forms never split K: more blocks
  if (n64 < 256) return 64;               // split-K regime: keep the fatter tile
  const double t64 = (double)cdiv(n64, 256) * 64.0;
  const double t32 = (double)cdiv(n32, 256) * 32.0 * 1.06;   // thinner tile: less reuse
  return (t32 < 0.93 * t64) ? 32 : 64;
}

// (A tap-row pipelined, double-buffered variant of the bf16x3 kernel - one barrier per 3 taps
// instead of two per chunk - measured slower, 149-166 vs 158-188 TFLOP/s, and was removed;
// see DESIGN.md 4.1 and the repository history.)
template <int TAPS, int BN, int KC, int A_ITERS, int PRO, bool UPS, bool BF3>
constexpr auto pick_kernel() {
  return conv_mfma_kernel<TAPS, BN, KC, A_ITERS, PRO, UPS, BF3>;
}

template <int TAPS, int BN, int KC, int A_ITERS, bool BF3 = false>
int launch_conv(const ConvK& k, int pro, int ups, size_t lds, hipStream_t st) {
  dim3 grid(k.n_mtiles * k.n_ntiles, k.splitk), block(256);
#define P2L_LAUNCH(PRO, UPS)                                                     \
  do {                                                                           \
    auto kfn = pick_kernel<TAPS, BN, KC, A_ITERS, PRO, UPS, BF3>();             \
    static std::atomic<bool> attr_set{false};                                                \
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

// K slices of a launch of the direct / pointwise kernels: a function of the LAYER SHAPE only (round 5).
// Until round 4 the factor followed the grid (cdiv(512, blocks), blocks ~ batch), so the fp32 summation
// order of the 4^2 ... 16^2 layers -- and with it a candidate's low bits -- depended on how many others
// shared its launch: an 8-GPU run (2-3 local candidates) did not reproduce a 1-GPU run's bits.  Measured
// per shape and local batch (tools/splitk_sweep.py -> profiles/round5_splitk_sweep.txt: conv + finish
// kernel, 13 shapes x batches 2 ... 18 x every slice count) ONE count per shape is within 0-4 % of the
// best count of every batch and better than the old grid rule from 3 candidates up (sum over the 13
// shapes at 2 / 3 / 5 / 9 / 18 candidates: 256 / 269 / 306 / 377 / 519 us, old rule 255 / 275 / 322 /
// 378 / 519): a slice must be worth its share of the finish launch, and more slices than ~64 blocks per
// image only add partial-sum traffic.
//   work term:  Cin * taps / 128   (>= 4 pointwise stages, >= 1.8 3x3 chunks per slice)
//   grid term:  64 blocks per image:  64 / (H W / 128 * Cout / 64)
//   at most 8 slices, 4 where an image has several tiles (H W >= 256); 2 slices never pay: 1
static int p2f(long v) { int p = 1; while (2L * p <= v) p <<= 1; return p; }
static int direct_split(const P2LConv* d) {
  const int kc = (d->taps == 9) ? 16 : (d->Cin % 32 == 0 ? 32 : 16);
  const int nchunks = d->Cin / kc;
  const long hw = (long)d->H * d->W;
  int s = p2f((long)d->Cin * d->taps / 128);
  const int g = p2f(524288L / (hw * d->Cout > 0 ? hw * d->Cout : 1));
  if (g < s) s = g;
  const int cap = hw >= 256 ? 4 : 8;
  if (s > cap) s = cap;
  // keep >= 2 chunks of work per slice for 3x3 (18 tap-chunks), >= 4 for 1x1
  const int min_chunks = (d->taps == 9) ? 2 : 4;
  while (s > 1 && nchunks / s < min_chunks) s >>= 1;
  return s <= 2 ? 1 : s;
}

// ... with the weights resident in registers (p2l_h2r.hip: persistent blocks, bit-identical results): the
// 64 -> 64 channel layers of whole 8x16-pixel tiles that never split K.  Shape and format only.
static bool direct_h2r(const P2LConv* d) {
  if ((d->form & P2L_FORM_NO_H2R) || !direct_h2(d)) return false;
  if (!p2l_h2r_shape(d->taps, d->ups, d->H, d->W, d->Cin, d->Cout, d->x_ld)) return false;
  return d->H * d->W >= 128 && direct_split(d) <= 1;
}


// K slices of the sub-pixel INPUT-GRADIENT form of a stride-2 transposed conv (ups 3, ext 1: StyleGAN2's up
// convs) in the fp16 x 2 kernel, whose chunk loop walks (phase plane, channel chunk) and takes a slice like the
// stride-1 kernel does: the 8^2 ... 32^2 layers of a few candidates were 8 - 32 blocks running 128 chunks each
// (0.20 - 0.23 ms for 5 - 50 us of work, profiles/round6_sg2_1024_layers.txt).  Shape and format only.
static int subpix_bwd_split(const P2LConv* d) {
  if (d->ups != 3 || !d->ext || !direct_h2(d)) return 1;
  ConvK k{};
  if (choose_tile(d, k) != P2L_OK || k.partial) return 1;
  const long hw = (long)(d->H >> 1) * (d->W >> 1);
  int s = p2f((long)d->Cin * 16 / 128);
  const int g = p2f(524288L / (hw * d->Cout > 0 ? hw * d->Cout : 1));
  if (g < s) s = g;
  const int cap = hw >= 256 ? 4 : 8;
  if (s > cap) s = cap;
  const int nchunks = 4 * (d->Cin / 16);
  while (s > 1 && nchunks / s < 8) s >>= 1;
  return s <= 2 ? 1 : s;
}

extern "C" int p2l_conv_suggest_splitk(const P2LConv* d) {
  ConvK k{};
  if (d && d->ups == 3 && d->ext) return subpix_bwd_split(d);
  if (!d || d->ups >= 2) return 1;       // the other sub-pixel modes never split K
  if (choose_tile(d, k) == P2L_OK && wino_split(d) > 1 && !(d->form & P2L_FORM_WINO_ANY) &&
      (d->H / 8) * (d->W / 16) * (d->Cout / 64) < 64)
    return wino_split(d);                // small-grid Winograd layer: a function of the shape only
  if (choose_tile(d, k) != P2L_OK || k.partial || wino_shape(d) || pw_any(d)) return 1;
  return direct_split(d);
}

// the 16x16 Winograd kernel in the fp16 x 2 arithmetic: 64 partial maxima per image in front of
// the split-K slices
static bool wino_h2(const P2LConv* d) {
  return wino_shape(d) && d->H % 16 == 0 && d->W % 16 == 0 &&
         !(d->form & (P2L_FORM_WINO_8X16 | P2L_FORM_WINO_BF3));
}
// P2LAmax producers: launches whose blocks each cover one tile of ONE image and end in the shared
// epilogue -- the unsplit 16x16 Winograd kernel (16x16 pixels x 64 channels) and the kernels that
// go through epilogue_vec (direct 3x3, sub-pixel forward: 4 phases, 1x1 in both arithmetics:
// 128 pixels x 32 / 64 channels); every WAVE of a block writes its own partial.  Split-K launches leave
// theirs through the finish kernel (one per 64 items).  The 8x16 Winograd kernel, the thin-output
// image kernel and unsplit tiles that span images write none.
static int effective_splitk(const P2LConv* d);
// the 16-byte finish kernel of a split-K launch: every pitch it touches a multiple of four floats
// (descriptor only: p2l_conv_amax_slots and the launch must agree)
static bool finish_v4_ok(const P2LConv* d) {
#ifdef P2L_AB_SCALAR_FINISH           // (A/B build: the 4-byte finish kernel for every split-K launch)
  return false;
#endif
  return d->n_store % 4 == 0 && d->Cout % 4 == 0 && d->y_ld % 4 == 0 && d->yp_ld % 4 == 0 &&
         d->res_ld % 4 == 0 && d->mask_ld % 4 == 0;
}
extern "C" int p2l_conv_amax_slots(const P2LConv* d) {
  if (!d) return 0;
  if (wino_shape(d)) {
    if (d->H % 16 || d->W % 16 || (d->form & P2L_FORM_WINO_8X16)) return 0;
    if (d->splitk > 1 && wino_split(d) == d->splitk) {
      // K-sliced launch: the float4 finish kernel writes the tensor, one partial per wave of 64
      // (quad, 4 channels) items -- when a wave never straddles two images
      const int per_image = (d->H / 2) * (d->W / 2) * (d->n_store / 4);
      return (per_image % 64 == 0 && d->n_store % 4 == 0) ? per_image / 64 : 0;
    }
    return (d->H / 16) * (d->W / 16) * (d->Cout / 64) * 8;     // (one partial per wave)
  }
  if (d->ups == 3 && effective_splitk(d) > 1) return 0;   // (the finish kernel runs on the low-res grid: no slots)
  if (effective_splitk(d) > 1) {
    // split-K launch of the direct kernels: a finish kernel writes the tensor, one partial per block of
    // 64 items -- (quad, 4 channels) for the 16-byte kernel, (quad, channel) for the scalar one -- when
    // a block never straddles two images.  (The hand-over thereby follows the LAYER, not the split-K
    // choice a batch size brings with it.)
    const int per_image = (d->H / 2) * (d->W / 2) * (finish_v4_ok(d) ? d->n_store / 4 : d->n_store);
    return (per_image % 64 == 0) ? per_image / 64 : 0;
  }
  ConvK k{};
  if (choose_tile(d, k) != P2L_OK || k.tb_log != 0 || k.partial) return 0;
  const int tm = thin_shape(d);
  if (tm >= 0) {                                       // three-channel image convs: the thin-input kernel only
    const bool geom = k.tw_log == 4 && k.th_log == 3;
    return (tm == 1 && geom) ? k.tiles_x * k.tiles_y * 4 : 0;
  }
  // (both pointwise forms tile 64 output channels whatever choose_bn says for the grid.  Until round 5
  //  the small-grid form was counted with choose_bn's 32 where the grid made it say so -- 16^2 256->1024 at
  //  9 and 18 candidates: twice the slots its blocks write, the reader took the maximum over stale ring
  //  contents as well and an image's power of two followed what had used the ring set before.)
#ifdef P2L_AB_PW_SLOTS_R4             // (A/B build: the round-4 count, for the test that has to fail on it)
  const int nnt = pw_shape(d) ? d->Cout / 64 : d->Cout / choose_bn(d, k.n_mtiles);
#else
  // (the register-resident 64-channel kernel writes all 64 channels of a tile from one block)
  const int nnt = (pw_shape(d) || pw_small_h2(d)) ? d->Cout / 64 : direct_h2r(d) ? 1 : d->Cout / choose_bn(d, k.n_mtiles);
#endif
  return k.tiles_x * k.tiles_y * nnt * (d->ups == 2 ? 4 : 1) * 4;
}
extern "C" size_t p2l_conv_workspace_bytes(const P2LConv* d) {
  const size_t h2 = (wino_h2(d) || direct_h2(d) || pw_small_h2(d)) ? (size_t)d->B * 64 * sizeof(float) : 0;
  if (d->splitk <= 1) return h2;
  return h2 + (size_t)d->splitk * d->B * d->H * d->W * d->Cout * sizeof(float);
}

// the split-K factor conv_launch_impl ends up with for d->splitk
static int effective_splitk(const P2LConv* d) {
  if (d->splitk > 1 && d->ups == 3 && d->ext && subpix_bwd_split(d) > 1) {
    const int nchunks = 4 * (d->Cin / 16);
    const int sk = d->splitk > nchunks ? nchunks : d->splitk;
    return cdiv(nchunks, cdiv(nchunks, sk));
  }
  if (d->splitk > 1 && d->ups == 0 && wino_split(d) == d->splitk && wino_shape(d)) return d->splitk;
  if (d->splitk <= 1 || d->ups >= 2 || wino_shape(d) || pw_any(d)) return 1;
  const int kc = (d->taps == 9) ? 16 : (d->Cin % 32 == 0 ? 32 : 16);
  const int nchunks = d->Cin / kc;
  int sk = d->splitk > nchunks ? nchunks : d->splitk;
  const int per = cdiv(nchunks, sk);
  return cdiv(nchunks, per);
}

static int conv_launch_impl(const P2LConv* d, const P2LArb* arb, const P2LConvExtra* ex,
                            const float* x, const float* w,
                            const float* bias, const float* pro_s,
                            const float* pro_t, const float* res,
                            const float* mask, float* y, float* yp,
                            void* workspace, size_t ws_bytes, void* stream) {
  if (!d || !x || !w) return P2L_EINVAL;
  if (d->taps != 1 && d->taps != 9) return P2L_EINVAL;
  const int kc = (d->taps == 9) ? 16 : (d->Cin % 32 == 0 ? 32 : 16);
  if (d->Cin % kc || d->Cout % 32 || d->B < 1) return P2L_EINVAL;
  if (d->x_ld % 4 || d->x_ld < d->Cin) return P2L_EINVAL;
  if (d->pro != P2L_PRO_NONE && (!pro_s || !pro_t || d->pro_bstride % 4)) return P2L_EINVAL;
  if (d->pool != P2L_POOL_NONE && !yp) return P2L_EINVAL;
  if (!y && !yp) return P2L_EINVAL;
  if (d->ups && d->taps != 9) return P2L_EUNSUP;
  if (d->taps == 9 && d->wfmt != P2L_WFMT_F32 && d->wfmt != P2L_WFMT_BF16X3 &&
      d->wfmt != P2L_WFMT_BF16X3W && d->wfmt != P2L_WFMT_BF16X3T)
    return P2L_EUNSUP;
  if (d->taps == 1 && d->wfmt != P2L_WFMT_F32 && d->wfmt != P2L_WFMT_PW) return P2L_EUNSUP;
  if (d->ups < 0 || d->ups > 3) return P2L_EINVAL;
  if (d->w_floats != 0) {
    // the packed formats carry no tag: the images a launch of (taps, ups, Cin, Cout, wfmt) may read sit
    // at fixed offsets of a buffer of exactly this size; a shorter one was packed for something else
    const size_t need = d->ups >= 2 ? p2l_packed_subpix_weight_floats(d->Cout, d->Cin, d->wfmt)
                                    : p2l_packed_weight_floats(d->taps, d->Cout, d->Cin, d->wfmt);
    if (d->w_floats < 0 || (size_t)d->w_floats < need) return P2L_EINVAL;
  }
  if (d->n_store < 1 || d->n_store > d->Cout || d->n_store % 4) return P2L_EINVAL;
  if ((y && d->y_ld % 4) || (yp && d->yp_ld % 4) || (res && d->res_ld % 4) ||
      (mask && d->mask_ld % 4))
    return P2L_EINVAL;
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
  k.form = d->form;
  int rc = choose_tile(d, k);
  if (rc) return rc;
  k.iH = d->H; k.iW = d->W; k.ibH = d->H; k.ibW = d->W; k.obH = d->H; k.obW = d->W;
  k.hp = halo_pitch(1 << k.tw_log, d->wfmt != P2L_WFMT_F32 && d->taps == 9);
  const bool bf3_3x3 = d->taps == 9 && d->wfmt != P2L_WFMT_F32;
  if (ex) {
    k.oscale = ex->oscale; k.oscale_bstride = ex->oscale_bstride;
    k.noise = ex->noise; k.noise_w = ex->noise_w;
    if (k.noise && d->y_ld != d->n_store && false) return P2L_EINVAL;
  }
  if (k.partial && (arb || d->pool != P2L_POOL_NONE || d->splitk > 1)) return P2L_EUNSUP;
  if (arb) {
    // one image per tile for the in-kernel form; with split-K the finish kernel does it
    if ((k.tb_log != 0 && effective_splitk(d) <= 1) || res || mask || d->act != P2L_ACT_NONE ||
        d->pool == P2L_POOL_MAX || !arb->x || !arb->s || !arb->t || !arb->partial)
      return P2L_EUNSUP;
    k.arb_x = arb->x; k.arb_s = arb->s; k.arb_t = arb->t; k.arb_skip = arb->skip;
    k.arb_partial = arb->partial; k.arb_x_ld = arb->x_ld; k.arb_bstride = arb->st_bstride;
    k.arb_skip_ld = arb->skip_ld; k.arb_skip_C = arb->skip_C; k.arb_skip_ups = arb->skip_ups;
    k.arb_nblk = k.n_mtiles / d->B;
    k.arb_nomask = arb->nomask;
  }
  // (an h2r-shaped layer keeps 64-channel tiles in the chunked kernel too: its launches may take either kernel,
  //  by their epilogue, and both must fill the same maxima slots)
  const int bn = direct_h2r(d) ? 64 : choose_bn(d, k.n_mtiles);
  k.n_ntiles = d->Cout / bn;
  k.nchunks = d->Cin / kc;
  const bool wino_sliced = wino_shape(d) && d->splitk > 1 && wino_split(d) == d->splitk;
  k.splitk = (d->splitk < 1 || (wino_shape(d) && !wino_sliced) || pw_any(d)) ? 1 : d->splitk;
  if (k.splitk > k.nchunks) k.splitk = k.nchunks;
  k.chunks_per_split = cdiv(k.nchunks, k.splitk);
  k.splitk = cdiv(k.nchunks, k.chunks_per_split);
  // fp16 x 2 Winograd: the partial maxima sit at the head of the workspace; a caller that gives
  // none gets the bf16 x 3 arithmetic
  const bool h2_direct = direct_h2(d), h2_pw_small = pw_small_h2(d);
  const size_t h2_bytes = (wino_h2(d) || h2_direct || h2_pw_small) ? (size_t)d->B * 64 * sizeof(float) : 0;
  const bool use_h2 = h2_bytes && workspace &&
                      ws_bytes >= h2_bytes + (k.splitk > 1 ? (size_t)k.splitk * d->B * d->H * d->W * d->Cout * sizeof(float) : 0);
  if (use_h2) {
    k.amax = (float*)workspace;
    k.ws = (float*)workspace + (size_t)d->B * 64;
  }
  {
    const P2LAmax* am = ex ? &ex->amax : (arb ? &arb->amax : nullptr);
    // consumers: the fp16 x 2 Winograd launch (instead of its own max-|x| pass) and the pointwise
    // kernel (which takes the fp16 x 2 form only with handed-over maxima)
    // (maxima recorded with a prologue applied say nothing about the raw tensor: not usable without one)
    if (am && am->in && am->in_n > 0 && !(am->in_applied && d->pro == P2L_PRO_NONE) &&
        (use_h2 || (pw_shape(d) && !(d->form & P2L_FORM_WINO_BF3)))) {
      k.amax_in = am->in; k.amax_in_n = am->in_n;
      k.amax_in_applied = am->in_applied ? 1 : 0;
    }
    int nslots = (am && (am->out || am->outp)) ? p2l_conv_amax_slots(d) : 0;
    if (thin_shape(d) >= 0 && ex && (ex->oscale || ex->noise)) nslots = 0;   // (generic kernel then)
    // (the slot count was promised for the fp16 x 2 small-grid pointwise kernel: 64-channel tiles)
    if (nslots > 0 && h2_pw_small && !use_h2 && effective_splitk(d) <= 1) return P2L_EWS;
    if (nslots > 0 && direct_h2r(d) && !use_h2) return P2L_EWS;   // (... for the register-resident kernel: one block per tile)
    if (nslots > 0) { k.amax_out = am->out; k.amax_outp = am->outp; k.amax_out_n = nslots; }
    if (nslots > 0 && am->out && am->next_s && am->next_t && !arb) {
      k.amax_ps = am->next_s; k.amax_pt = am->next_t; k.amax_pbstride = am->next_bstride;
    }
  }
  if (k.splitk > 1) {
    const size_t need = (use_h2 ? h2_bytes : 0) + (size_t)k.splitk * d->B * d->H * d->W * d->Cout * sizeof(float);
    if (!workspace || ws_bytes < need) return P2L_EWS;
    if (arb) k.arb_nblk = p2l_conv_arb_nblk_ws(d);      // finish kernel: one partial per quad
  }
  hipStream_t st = (hipStream_t)stream;

  int prof_slot = -1;
  bool prof_ok = g_prof.on;
  if (prof_ok) {
    // an event pair recorded during stream capture becomes graph nodes and can never be read
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) prof_ok = false;
  }
  if (prof_ok) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (g_prof.on && g_prof.n < (int)g_prof.flops.size() && (g_prof.seq++ % g_prof.period) == g_prof.phase)
      prof_slot = g_prof.n++;
  }
  if (prof_slot >= 0) {
    g_prof.flops[prof_slot] = d->algo_flops > 0.0
        ? d->algo_flops
        : 2.0 * d->B * d->H * d->W * (double)d->Cin * d->Cout * d->taps;
    // what the MFMAs actually multiply: padded channel counts, and for the sub-pixel forms 4
    // phase-taps per high-resolution output pixel instead of 9
    g_prof.xflops[prof_slot] =
        2.0 * d->B * d->H * d->W * (double)d->Cin * d->Cout * (d->ups >= 2 ? 4 : d->taps);
    g_prof.kind[prof_slot] = d->taps == 9 ? 0 : 1;
    g_prof.fam[prof_slot] = P2L_PROF_FAM_OTHER;
    g_prof.nprod[prof_slot] = (d->wfmt == P2L_WFMT_F32 || (d->taps == 1 && !pw_shape(d))) ? 16 : 6;
    g_prof.shape[prof_slot] = {d->taps, d->B, d->H, d->W, d->Cin, d->Cout, d->ups, d->pro,
                               arb ? (arb->skip ? 2 : 1) : 0, d->splitk};
    {
      // algorithmic bytes: every operand tensor once (input at ITS resolution, outputs,
      // residual / mask, the fused activation-backward operands) + the packed weights
      const double opx = (double)d->B * d->H * d->W;
      const double ipx = d->ups == 1 || d->ups == 2 ? opx / 4 : (d->ups == 3 ? opx * 4 : opx);
      double by = ipx * d->Cin + (double)d->taps * d->Cin * d->Cout;
      if (y) by += opx * d->n_store;
      if (yp) by += opx / 4 * d->n_store;
      if (res) by += (d->res_ups ? opx / 4 : opx) * d->n_store;
      if (mask) by += opx * d->n_store;
      if (arb) by += opx * d->n_store * (arb->skip ? 2.0 : 1.0);
      g_prof.bytes[prof_slot] = 4.0 * by;
      g_prof.wbytes[prof_slot] = 4.0 * ((y ? opx * d->n_store : 0.0) + (yp ? opx / 4 * d->n_store : 0.0));
    }
    (void)hipEventRecord(g_prof.ev[2 * prof_slot], st);
  }
  // ---- Winograd F(2x2,3x3) form (p2l_wino.hip): stride-1 3x3 layers whose grid fills the
  //      chip with 8x16-pixel x 64-channel blocks; everything else stays on the direct kernel
  if (wino_shape(d)) {
    ConvK kw = k;                    // (arb_nblk keeps the 128-pixel tiling of the caller's buffer)
    kw.w = w + (size_t)9 * d->Cout * d->Cin * 3 / 2;
    if (use_h2) {
      kw.w += p2l_wino_weight_floats(d->Cout, d->Cin);
      kw.w_tail = reinterpret_cast<const unsigned*>(kw.w + (size_t)d->Cout * d->Cin * 16);
    }
    kw.tiles_x = d->W / 16; kw.tiles_y = d->H / 8;
    kw.n_mtiles = d->B * kw.tiles_x * kw.tiles_y;
    kw.n_ntiles = d->Cout / 64;
    kw.nchunks = d->Cin / 16;
    if (wino_sliced) {
      kw.splitk = d->splitk;
      kw.chunks_per_split = kw.nchunks / kw.splitk;     // (the factor divides the chunk count)
    } else {
      kw.splitk = 1;
    }
    rc = p2l_wino_launch(kw, d->pro, st);
    if (rc == P2L_OK && wino_sliced) {
      // the slices meet in the deterministic finish kernel of the direct path: fixed-order sum,
      // then the whole epilogue (bias / residual / activation / fused activation backward)
      const size_t total = (size_t)k.B * (k.H >> 1) * (k.W >> 1) * (k.n_store >> 2);
      hipLaunchKernelGGL(conv_splitk_finish4, dim3(cdiv(total, 256)), dim3(256), 0, st, k);
      rc = p2l_check_launch();
    }
    if (prof_slot >= 0) {
      g_prof.xflops[prof_slot] = 2.0 * d->B * d->H * d->W * (double)d->Cin * d->Cout * 4;   // 16 per quad
      if (use_h2) g_prof.nprod[prof_slot] = 3;
      g_prof.fam[prof_slot] = use_h2 ? P2L_PROF_FAM_WINO_H2 : P2L_PROF_FAM_OTHER;
      (void)hipEventRecord(g_prof.ev[2 * prof_slot + 1], st);
    }
    return rc;
  }
  // ---- 1x1 conv in the bf16x3 arithmetic (p2l_pw.hip): the image follows the fp32 weights ----
  if (pw_shape(d)) {
    ConvK kp = k;
    kp.w = w + (size_t)d->Cout * d->Cin;
    if (k.amax_in != nullptr) {                          // fp16 x 2 image behind the bf16 x 3 one
      kp.w += (size_t)d->Cout * d->Cin * 3 / 2;
      kp.w_tail = reinterpret_cast<const unsigned*>(kp.w + (size_t)d->Cout * d->Cin);
    }
    kp.n_ntiles = d->Cout / 64;
    kp.nchunks = d->Cin / 64;
    kp.splitk = 1;
    rc = p2l_pw_launch(kp, d->pro, st);
    if (prof_slot >= 0) {
      if (k.amax_in != nullptr) g_prof.nprod[prof_slot] = 3;
      g_prof.fam[prof_slot] = P2L_PROF_FAM_PW;
      (void)hipEventRecord(g_prof.ev[2 * prof_slot + 1], st);
    }
    return rc;
  }
  // ---- 3-channel image convs ----
  {
    const int tm = thin_shape(d);
    const bool geom = k.tw_log == 4 && k.th_log == 3 && k.tb_log == 0 && !k.partial && !(ex && (ex->oscale || ex->noise));
    if (tm == 1 && geom) {
      ConvK kt = k;
      kt.w = w + (size_t)9 * d->Cout * d->Cin * 3 / 2;
      kt.splitk = 1;
      rc = p2l_thinin_launch(kt, d->pro, st);
      if (prof_slot >= 0) {
        g_prof.xflops[prof_slot] = 2.0 * d->B * d->H * d->W * 32.0 * d->Cout;    // K = 27 -> 32
        g_prof.fam[prof_slot] = P2L_PROF_FAM_THIN;
        (void)hipEventRecord(g_prof.ev[2 * prof_slot + 1], st);
      }
      return rc;
    }
    if (tm == 0 && geom && !arb && !res && !mask && d->pool == P2L_POOL_NONE && y) {
      ConvK kt = k;
      kt.w = w + (size_t)9 * d->Cout * d->Cin * 3 / 2;
      kt.nchunks = d->Cin / 16;
      kt.splitk = 1;
      rc = p2l_thinout_launch(kt, d->pro, st);
      if (prof_slot >= 0) {
        // pointwise product onto 32 columns for the 8x16 pixels + halo (192 rows per 128)
        g_prof.xflops[prof_slot] = 2.0 * d->B * d->H * d->W * 1.5 * d->Cin * 32.0;
        g_prof.fam[prof_slot] = P2L_PROF_FAM_THIN;
        (void)hipEventRecord(g_prof.ev[2 * prof_slot + 1], st);
      }
      return rc;
    }
  }
  // ---- sub-pixel modes (ups 2 = forward, 3 = input-gradient of an upsampled conv) ----
  if (d->ups >= 2) {
    // K slices: the input-gradient form of a transposed conv in the fp16 x 2 kernel only (subpix_bwd_split);
    // the fused modulation / activation backward of a sliced launch runs in the finish kernel, per quad,
    // so multi-image tiles are no obstacle there
    const int sk3 = (d->taps == 9 && d->splitk > 1) ? ((d->ups == 3 && d->ext) ? effective_splitk(d) : 1) : 1;
    if (sk3 > 1 && !(use_h2 && direct_h2(d))) return P2L_EWS;
    if (d->taps != 9 || (k.splitk != 1 && sk3 == 1) || d->Cin % 16 || d->pool != P2L_POOL_NONE ||
        (arb && k.tb_log != 0 && sk3 == 1) || (d->ups == 2 && (res || mask)))
      return P2L_EUNSUP;
    {
      int gH, gW;
      grid_dims(d, gH, gW);
      const int e = d->ext ? 1 : 0, e2 = d->ext ? 2 : 0;
      k.H = gH; k.W = gW;                    // the kernel tiles the low-res grid
      if (d->ups == 2) {                     // low-res input -> phases of the high-res buffer
        k.iH = d->H >> 1; k.iW = d->W >> 1; k.ibH = k.iH; k.ibW = k.iW;
        k.obH = d->H + e2; k.obW = d->W + e2;
      } else {                               // phase planes of the high-res buffer -> low-res
        k.iH = (d->H >> 1) + e; k.iW = (d->W >> 1) + e;
        k.ibH = d->H + e2; k.ibW = d->W + e2;
        k.obH = gH; k.obW = gW;
      }
    }
    k.sp_mode = d->ups - 1;
    k.sp_skip = (d->ext && !(d->form & P2L_FORM_NO_SP_SKIP)) ? 1 : 0;
    k.sp_ncc = d->Cin / 16;
    k.nchunks = (d->ups == 3) ? 4 * k.sp_ncc : k.sp_ncc;
    k.chunks_per_split = cdiv(k.nchunks, sk3);
    k.splitk = sk3;
    k.ups = 0;
    const int a_rows_sp = (1 << k.tb_log) * ((1 << k.th_log) + 2) * ((1 << k.tw_log) + 2);
    const bool small_sp = (a_rows_sp * 4 <= 3 * 256) && k.tb_log == 0;
    const bool bf3 = bf3_3x3;
    if (use_h2 && h2_direct) {
      // fp16 x 2 form (p2l_h2.hip): its image follows the bf16 x 3 one
      ConvK kh = k;
      kh.hp = halo_pitch_h2(1 << k.tw_log);
      kh.w = w + (size_t)16 * d->Cout * d->Cin * 3 / 2;
      kh.w_tail = reinterpret_cast<const unsigned*>(kh.w + (size_t)16 * d->Cout * d->Cin);
      if (kh.amax_in == nullptr) {
        ConvK ka = kh;                      // the INPUT tensor: low-res (forward) / high-res frame (gradient)
        ka.H = k.ibH; ka.W = k.ibW;
        rc = p2l_amax_launch(ka, d->pro, st);
        if (rc) return rc;
      }
      // Forward: two output phases per block on one staged patch -- for the transposed convs with 64-channel
      // tiles (tools/bench_subpix.py: 1.10 - 1.18 x per launch; BigGAN's nearest-upsample convs, whose 16 taps are
      // all live and whose launches are 60 - 100 us, lose 5 - 15 % to the halved grid; 32-channel tiles 3 %), or
      // wherever P2L_FORM_SP_PAIR asks for it (tests); P2L_FORM_NO_SP_PAIR: never.  Shape and form only.
#ifdef P2L_AB_NO_SP_PAIR
      const bool pair = d->ups == 2 && !(d->form & P2L_FORM_NO_SP_PAIR) && (d->form & P2L_FORM_SP_PAIR);
#else
      const bool pair = d->ups == 2 && !(d->form & P2L_FORM_NO_SP_PAIR) &&
                        ((d->ext && bn == 64) || (d->form & P2L_FORM_SP_PAIR));
#endif
      rc = p2l_h2_launch(kh, d->pro, pair ? 8 : 4, bn, small_sp, st);
      if (rc == P2L_OK && k.splitk > 1) {
        // the slices meet in the finish kernel of the direct path, on the low-res grid this launch tiled
        const bool arb_al = !arb || (k.arb_x_ld % 4 == 0 && (!k.arb_skip || k.arb_skip_ld % 4 == 0));
        if (finish_v4_ok(d) && arb_al) {
          const size_t total = (size_t)k.B * (k.H >> 1) * (k.W >> 1) * (k.n_store >> 2);
          hipLaunchKernelGGL(conv_splitk_finish_v4, dim3(cdiv(total, 64)), dim3(256), 0, st, k);
        } else {
          const size_t total = (size_t)k.B * (k.H >> 1) * (k.W >> 1) * k.n_store;
          hipLaunchKernelGGL(conv_splitk_finish, dim3(cdiv(total, 64)), dim3(256), 0, st, k);
        }
        rc = p2l_check_launch();
      }
      if (prof_slot >= 0) {
        g_prof.nprod[prof_slot] = 3;
        g_prof.fam[prof_slot] = P2L_PROF_FAM_SUBPIX_H2;
        (void)hipEventRecord(g_prof.ev[2 * prof_slot + 1], st);
      }
      return rc;
    }
    const int a_lds_sp = (1 << k.tb_log) * ((1 << k.th_log) + 2) * k.hp;
    size_t lds_sp = (size_t)(a_lds_sp + 4 * bn) * (bf3 ? 24 : 20) * sizeof(float);
    const size_t lds_epi = (size_t)4 * 32 * (bn + 4) * sizeof(float);
    if (lds_epi > lds_sp) lds_sp = lds_epi;
    dim3 grid(k.n_mtiles * k.n_ntiles, d->ups == 2 ? 4 : 1), block(256);
#define P2L_LAUNCH_SP(BNV, AIT, PROV)                                                     \
    do {                                                                                  \
      if (bf3) { P2L_LAUNCH_SP2(BNV, AIT, PROV, true); } else { P2L_LAUNCH_SP2(BNV, AIT, PROV, false); } \
    } while (0)
#define P2L_LAUNCH_SP2(BNV, AIT, PROV, BF3V)                                              \
    do {                                                                                  \
      auto kfn = pick_kernel<4, BNV, 16, AIT, PROV, false, BF3V>();                       \
      static std::atomic<bool> attr_set{false};                                                       \
      if (!attr_set) {                                                                    \
        (void)hipFuncSetAttribute((const void*)kfn,                                       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        attr_set = true;                                                                  \
      }                                                                                   \
      hipLaunchKernelGGL(kfn, grid, block, lds_sp, st, k);                                \
    } while (0)
#define P2L_LAUNCH_SP_PRO(BNV, AIT)                                                       \
    do {                                                                                  \
      if (d->pro == P2L_PRO_NONE) P2L_LAUNCH_SP(BNV, AIT, P2L_PRO_NONE);                  \
      else if (d->pro == P2L_PRO_AFFINE_RELU) P2L_LAUNCH_SP(BNV, AIT, P2L_PRO_AFFINE_RELU); \
      else P2L_LAUNCH_SP(BNV, AIT, P2L_PRO_AFFINE);                                       \
    } while (0)
    if (bn == 64) {
      if (small_sp) P2L_LAUNCH_SP_PRO(64, 3); else P2L_LAUNCH_SP_PRO(64, 5);
    } else {
      if (small_sp) P2L_LAUNCH_SP_PRO(32, 3); else P2L_LAUNCH_SP_PRO(32, 5);
    }
#undef P2L_LAUNCH_SP_PRO
#undef P2L_LAUNCH_SP
#undef P2L_LAUNCH_SP2
    rc = p2l_check_launch();
    if (prof_slot >= 0) (void)hipEventRecord(g_prof.ev[2 * prof_slot + 1], st);
    return rc;
  }
  const int TW = 1 << k.tw_log, TH = 1 << k.th_log, TB = 1 << k.tb_log;
  const int a_rows = (d->taps == 9) ? TB * (TH + 2) * (TW + 2) : 128;
  const bool bf3 = bf3_3x3;
  const int a_rows_lds = (d->taps == 9) ? TB * (TH + 2) * k.hp : 128;
  size_t lds = (size_t)(a_rows_lds + d->taps * bn) * (bf3 ? 24 : kc + 4) * sizeof(float);
  {
    // the vectorised epilogue re-uses the staging LDS for 4 wave tiles of 32 x (bn+4)
    const size_t lds_epi = (size_t)4 * 32 * (bn + 4) * sizeof(float);
    if (lds_epi > lds) lds = lds_epi;
  }

  if (d->taps == 9 && use_h2 && h2_direct) {
    ConvK kh = k;
    kh.hp = halo_pitch_h2(TW);
    kh.w = w + (size_t)9 * d->Cout * d->Cin * 3 / 2;
    if (p2l_wino_weight_ok(d->Cout, d->Cin))
      kh.w += p2l_wino_weight_floats(d->Cout, d->Cin) + p2l_wino_h2_weight_floats(d->Cout, d->Cin);
    kh.w_tail = reinterpret_cast<const unsigned*>(kh.w + (size_t)9 * d->Cout * d->Cin);
    if (kh.amax_in == nullptr) {
      rc = p2l_amax_launch(kh, d->pro, st);
      if (rc) return rc;
    }
    const bool small = (a_rows * 4 <= 3 * 256) && TB == 1;
    const bool h2r = direct_h2r(d) && k.splitk == 1 && !k.partial && TB == 1 && p2l_h2r_takes(kh);
    rc = h2r ? p2l_h2r_launch(kh, d->pro, st) : p2l_h2_launch(kh, d->pro, 9, bn, small, st);
    if (prof_slot >= 0) { g_prof.nprod[prof_slot] = 3; g_prof.fam[prof_slot] = h2r ? P2L_PROF_FAM_DIRECT_H2R : P2L_PROF_FAM_DIRECT_H2; }
  } else
  if (d->taps == 9) {
    const bool small = (a_rows * 4 <= 3 * 256) && TB == 1;
    if (bf3) {
      if (bn == 64) rc = small ? launch_conv<9, 64, 16, 3, true>(k, d->pro, d->ups, lds, st)
                               : launch_conv<9, 64, 16, 5, true>(k, d->pro, d->ups, lds, st);
      else          rc = small ? launch_conv<9, 32, 16, 3, true>(k, d->pro, d->ups, lds, st)
                               : launch_conv<9, 32, 16, 5, true>(k, d->pro, d->ups, lds, st);
    } else
    if (bn == 64) rc = small ? launch_conv<9, 64, 16, 3>(k, d->pro, d->ups, lds, st)
                             : launch_conv<9, 64, 16, 5>(k, d->pro, d->ups, lds, st);
    else          rc = small ? launch_conv<9, 32, 16, 3>(k, d->pro, d->ups, lds, st)
                             : launch_conv<9, 32, 16, 5>(k, d->pro, d->ups, lds, st);
  } else if (d->taps == 1 && use_h2 && h2_pw_small) {
    // 1x1 layers of 4^2 ... 16^2 in the fp16 x 2 arithmetic (multi-image tiles, split-K in
    // blockIdx.y; the slices meet in conv_splitk_finish below)
    ConvK kp = k;
    kp.w = w + (size_t)d->Cout * d->Cin + (size_t)d->Cout * d->Cin * 3 / 2;
    kp.w_tail = reinterpret_cast<const unsigned*>(kp.w + (size_t)d->Cout * d->Cin);
    kp.n_ntiles = d->Cout / 64;
    if (kp.amax_in == nullptr) {
      rc = p2l_amax_launch(kp, d->pro, st);
      if (rc) return rc;
    }
    rc = p2l_pw_launch(kp, d->pro, st);
    if (prof_slot >= 0) { g_prof.nprod[prof_slot] = 3; g_prof.fam[prof_slot] = P2L_PROF_FAM_PW; }
  } else if (kc == 32) {
    if (bn == 64) rc = launch_conv<1, 64, 32, 4>(k, d->pro, 0, lds, st);
    else          rc = launch_conv<1, 32, 32, 4>(k, d->pro, 0, lds, st);
  } else {          // 1x1 with Cin a multiple of 16 only (RGB image gradients)
    if (bn == 64) rc = launch_conv<1, 64, 16, 2>(k, d->pro, 0, lds, st);
    else          rc = launch_conv<1, 32, 16, 2>(k, d->pro, 0, lds, st);
  }
  if (rc) return rc;
  if (k.splitk > 1) {
    const bool arb_al = !arb || (k.arb_x_ld % 4 == 0 && (!k.arb_skip || k.arb_skip_ld % 4 == 0));
    if (finish_v4_ok(d) && !arb_al) return P2L_EINVAL;   // (the maxima slots were sized for the 16-byte kernel)
    if (finish_v4_ok(d)) {
      const size_t total = (size_t)k.B * (k.H >> 1) * (k.W >> 1) * (k.n_store >> 2);
      hipLaunchKernelGGL(conv_splitk_finish_v4, dim3(cdiv(total, 64)), dim3(256), 0, st, k);
    } else {
      const size_t total = (size_t)k.B * (k.H >> 1) * (k.W >> 1) * k.n_store;
      hipLaunchKernelGGL(conv_splitk_finish, dim3(cdiv(total, 64)), dim3(256), 0, st, k);
    }
    rc = p2l_check_launch();
  }
  if (prof_slot >= 0) (void)hipEventRecord(g_prof.ev[2 * prof_slot + 1], st);
  return rc;
}

extern "C" int p2l_conv_fwd(const P2LConv* d, const float* x, const float* w,
                            const float* bias, const float* pro_s,
                            const float* pro_t, const float* res,
                            const float* mask, float* y, float* yp,
                            void* workspace, size_t ws_bytes, void* stream) {
  return conv_launch_impl(d, nullptr, nullptr, x, w, bias, pro_s, pro_t, res, mask, y, yp,
                          workspace, ws_bytes, stream);
}

extern "C" int p2l_conv_fwd_ex(const P2LConv* d, const P2LConvExtra* ex, const float* x,
                               const float* w, const float* bias, const float* pro_s,
                               const float* pro_t, const float* res, const float* mask,
                               float* y, float* yp, void* workspace, size_t ws_bytes,
                               void* stream) {
  return conv_launch_impl(d, nullptr, ex, x, w, bias, pro_s, pro_t, res, mask, y, yp, workspace,
                          ws_bytes, stream);
}

extern "C" int p2l_conv_arb_fusable(const P2LConv* d) {
  ConvK k{};
  if (!d || choose_tile(d, k) != P2L_OK) return 0;
  return k.tb_log == 0;     // one image per tile; the caller decides about split-K
}

extern "C" int p2l_conv_arb_nblk(const P2LConv* d) {
  ConvK k{};
  if (!d || choose_tile(d, k) != P2L_OK) return 0;
  return k.n_mtiles / d->B;
}

extern "C" int p2l_conv_arb_split_fusable(const P2LConv* d) {
  ConvK k{};
  if (!d || choose_tile(d, k) != P2L_OK || k.partial) return 0;
  return effective_splitk(d) > 1 && d->pool != P2L_POOL_MAX;
}

extern "C" int p2l_conv_arb_nblk_ws(const P2LConv* d) {
  if (!d) return 0;
  const int sh = d->ups == 3 ? 2 : 1;       // (quads of the grid the finish kernel walks: low-res for ups 3)
  return effective_splitk(d) > 1 ? (d->H >> sh) * (d->W >> sh) : p2l_conv_arb_nblk(d);
}

static int dgrad_arb_unsplit(const P2LConv* d, const P2LArb* arb, const float* dy, const float* w,
                             float* dx, void* workspace, size_t ws_bytes, void* stream);
extern "C" int p2l_conv_dgrad_arb_ws(const P2LConv* d, const P2LArb* arb, const float* dy,
                                     const float* w, float* dx, void* workspace, size_t ws_bytes,
                                     void* stream) {
  if (!d || !arb || !dx) return P2L_EINVAL;
  P2LConv dd = *d;
  if (dd.ups == 3) dd.pool = P2L_POOL_NONE;
  if (effective_splitk(&dd) <= 1) return dgrad_arb_unsplit(d, arb, dy, w, dx, workspace, ws_bytes, stream);
  const bool pooled = dd.pool == P2L_POOL_SUM;
  int rc = conv_launch_impl(&dd, arb, nullptr, dy, w, nullptr, nullptr, nullptr, nullptr, nullptr,
                            pooled ? nullptr : dx, pooled ? dx : nullptr, workspace, ws_bytes,
                            stream);
  if (rc) return rc;
  return p2l_arb_finish(arb->partial, arb->ds, arb->dt, dd.B, p2l_conv_arb_nblk_ws(&dd), dd.Cout,
                        arb->dsdt_bstride, stream);
}

// (the workspace of an unsplit launch only serves the fp16 x 2 Winograd form: wino_h2)
static int dgrad_arb_unsplit(const P2LConv* d, const P2LArb* arb, const float* dy, const float* w,
                             float* dx, void* workspace, size_t ws_bytes, void* stream) {
  if (!d || !arb || !dx) return P2L_EINVAL;
  P2LConv dd = *d;
  dd.splitk = 1;
  if (dd.ups == 3) dd.pool = P2L_POOL_NONE;
  const bool pooled = dd.pool == P2L_POOL_SUM;
  int rc = conv_launch_impl(&dd, arb, nullptr, dy, w, nullptr, nullptr, nullptr, nullptr, nullptr,
                            pooled ? nullptr : dx, pooled ? dx : nullptr, workspace, ws_bytes, stream);
  if (rc) return rc;
  const int nblk = p2l_conv_arb_nblk(&dd);
  return p2l_arb_finish(arb->partial, arb->ds, arb->dt, dd.B, nblk, dd.Cout,
                        arb->dsdt_bstride, stream);
}
extern "C" int p2l_conv_dgrad_arb(const P2LConv* d, const P2LArb* arb, const float* dy,
                                  const float* w, float* dx, void* stream) {
  return dgrad_arb_unsplit(d, arb, dy, w, dx, nullptr, 0, stream);
}

extern "C" int p2l_pack_conv_weight_subpix(const float* w_oihw, int O, int I, int N_pad,
                                           int K_pad, int transpose_flip, int mode,
                                           float* w_packed, void* stream) {
  if (!w_oihw || !w_packed || mode < 0 || mode > 1) return P2L_EINVAL;
  const int N = transpose_flip ? I : O, K = transpose_flip ? O : I;
  if (K_pad % 16 || N_pad % 32 || N_pad < N || K_pad < K) return P2L_EINVAL;
  const size_t total = (size_t)16 * K_pad * N_pad;
  hipLaunchKernelGGL(pack_subpix_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, w_packed, O, I, N_pad, K_pad, transpose_flip,
                     mode, 0);
  return p2l_check_launch();
}

extern "C" int p2l_pack_conv_weight_subpix_bf3(const float* w_oihw, int O, int I, int N_pad,
                                               int K_pad, int transpose_flip, int mode,
                                               float* w_packed, void* stream) {
  if (!w_oihw || !w_packed || mode < 0 || mode > 1) return P2L_EINVAL;
  const int N = transpose_flip ? I : O, K = transpose_flip ? O : I;
  if (K_pad % 16 || N_pad % 32 || N_pad < N || K_pad < K) return P2L_EINVAL;
  const size_t total = (size_t)16 * K_pad * N_pad;
  hipLaunchKernelGGL(pack_subpix_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, w_packed, O, I, N_pad, K_pad, transpose_flip,
                     mode, 1);
  return p2l_check_launch();
}


extern "C" int p2l_prof_begin(int max_launches) {
  if (max_launches < 1) return P2L_EINVAL;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  while ((int)g_prof.ev.size() < 2 * max_launches) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return P2L_ELAUNCH;
    g_prof.ev.push_back(e);
  }
  g_prof.flops.assign(max_launches, 0.0);
  g_prof.xflops.assign(max_launches, 0.0);
  g_prof.nprod.assign(max_launches, 6);
  g_prof.bytes.assign(max_launches, 0.0);
  g_prof.wbytes.assign(max_launches, 0.0);
  g_prof.kind.assign(max_launches, 0);
  g_prof.fam.assign(max_launches, P2L_PROF_FAM_OTHER);
  g_prof.shape.assign(max_launches, {});
  g_prof.n = 0;
  g_prof.seq = 0; g_prof.period = 1; g_prof.phase = 0;
  g_prof.on = true;
  return P2L_OK;
}

extern "C" int p2l_prof_dump(const char* path) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.dump_path = path ? path : "";
  return P2L_OK;
}

extern "C" int p2l_prof_step(int step, int period) {
  if (step < 0 || period < 1) return P2L_EINVAL;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.seq = 0;
  g_prof.period = period;
  g_prof.phase = step % period;
  return P2L_OK;
}

extern "C" int p2l_prof_totals(P2LProfTotals* out) {
  // the caller's struct may be older (shorter) than this library's: totals are gathered in a full one
  // and the leading out->size bytes are copied back
  if (!out || out->size < offsetof(P2LProfTotals, flops)) return P2L_EINVAL;
  P2LProfTotals T{};
  double *flops = T.flops, *ms = T.ms, *bytes = T.bytes, *exec_flops = T.exec_flops,
         *mfma_flops = T.mfma_flops, *write_bytes = T.write_bytes;
  int32_t* count = T.count;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.on = false;
  flops[0] = flops[1] = ms[0] = ms[1] = 0.0;
  count[0] = count[1] = 0;
  if (bytes) bytes[0] = bytes[1] = 0.0;
  // p2l_prof_dump(<file>): one line per launch (tools/prof_layers.py reads it)
  FILE* dump = g_prof.dump_path.empty() ? nullptr : fopen(g_prof.dump_path.c_str(), "w");
  for (int i = 0; i < g_prof.n; ++i) {
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return P2L_ELAUNCH;
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess)
      return P2L_ELAUNCH;
    const int k = g_prof.kind[i];
    flops[k] += g_prof.flops[i];
    if (exec_flops) exec_flops[k] += g_prof.xflops[i];
    if (mfma_flops) mfma_flops[k] += g_prof.xflops[i] * g_prof.nprod[i];
    if (bytes) bytes[k] += g_prof.bytes[i];
    if (write_bytes) write_bytes[k] += g_prof.wbytes[i];
    ms[k] += t;
    count[k] += 1;
    {
      const int fm = g_prof.fam[i];
      T.fam_count[fm] += 1;
      T.fam_ms[fm] += t;
      T.fam_flops[fm] += g_prof.flops[i];
      T.fam_mfma_flops[fm] += g_prof.xflops[i] * g_prof.nprod[i];
      T.fam_bytes[fm] += g_prof.bytes[i];
    }
    if (dump) {
      const auto& sh = g_prof.shape[i];
      fprintf(dump, "%d %d %d %d %d %d %d %d %d %d %.6e %.6e %.6f %d\n", sh[0], sh[1], sh[2], sh[3],
              sh[4], sh[5], sh[6], sh[7], sh[8], sh[9], g_prof.flops[i], g_prof.bytes[i], t, g_prof.nprod[i]);
    }
  }
  if (dump) fclose(dump);
  g_prof.n = 0;
  const uint32_t n = out->size < sizeof(T) ? out->size : (uint32_t)sizeof(T);
  T.size = n;
  memcpy(out, &T, n);
  return P2L_OK;
}

extern "C" int p2l_pack_conv_weight(const float* w_oihw, int O, int I, int taps,
                                    int N_pad, int K_pad, int transpose_flip,
                                    float* w_packed, void* stream) {
  // taps = KH*KW: 1 and 9 feed conv_mfma_kernel, anything else (25, 121) p2l_gconv_fwd
  if (!w_oihw || !w_packed || taps < 1) return P2L_EINVAL;
  const int kc = (taps != 1) ? 16 : (K_pad % 32 == 0 ? 32 : 16);
  const int N = transpose_flip ? I : O, K = transpose_flip ? O : I;
  if (K_pad % kc || N_pad % 32 || N_pad < N || K_pad < K) return P2L_EINVAL;
  const size_t total = (size_t)taps * K_pad * N_pad;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(cdiv(total, 256)), dim3(256),
                     0, (hipStream_t)stream, w_oihw, w_packed, O, I, taps, N_pad,
                     K_pad, kc, transpose_flip, 0);
  return p2l_check_launch();
}

// weights of the generic gather conv (p2l_gconv_fwd, p2l_alex.hip): [tap][K_pad/16][N_pad][16] for EVERY kernel
// size -- p2l_pack_conv_weight gives 1x1 convs 32-channel chunks where K_pad allows it (conv_mfma_kernel<1, ..>)
extern "C" int p2l_pack_gconv_weight(const float* w_oihw, int O, int I, int taps, int N_pad, int K_pad,
                                     int transpose_flip, float* w_packed, void* stream) {
  if (!w_oihw || !w_packed || taps < 1) return P2L_EINVAL;
  const int N = transpose_flip ? I : O, K = transpose_flip ? O : I;
  if (K_pad % 16 || N_pad % 32 || N_pad < N || K_pad < K) return P2L_EINVAL;
  const size_t total = (size_t)taps * K_pad * N_pad;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     w_packed, O, I, taps, N_pad, K_pad, 16, transpose_flip, 0);
  return p2l_check_launch();
}

// bf16x3 pre-split weights for the BF3 conv kernels: 1.5 x taps*K_pad*N_pad floats
// P2L_WFMT_BF16X3W: the direct bf16x3 tile image followed -- for shapes the Winograd kernel
// takes (p2l_wino_weight_ok) -- by the transform-domain image of the same weights
extern "C" size_t p2l_packed_weight_floats(int taps, int N_pad, int K_pad, int wfmt) {
  const size_t direct = (size_t)taps * N_pad * K_pad;
  if (wfmt == P2L_WFMT_F32) return direct;
  if (wfmt == P2L_WFMT_PW)                                      // fp32 layout + bf16x3 image + fp16x2 image
    return direct + direct * 3 / 2 + p2l_pw_h2_weight_floats(N_pad, K_pad);
  size_t n = direct * 3 / 2;
  if (wfmt == P2L_WFMT_BF16X3W && taps == 9 && p2l_wino_weight_ok(N_pad, K_pad))
    n += p2l_wino_weight_floats(N_pad, K_pad) + p2l_wino_h2_weight_floats(N_pad, K_pad);
  if (wfmt == P2L_WFMT_BF16X3W && taps == 9) n += p2l_h2_weight_floats(N_pad, K_pad, 0);   // fp16 x 2 direct image
  if (wfmt == P2L_WFMT_BF16X3T && taps == 9) n += p2l_thin_weight_floats(N_pad, K_pad);
  return n;
}

// P2L_WFMT_PW (1x1 convs): the fp32 layout of p2l_pack_conv_weight followed by the bf16x3 image
// [K_pad/16][N_pad/32][32 rows][96 B]; the launcher picks per layer shape
extern "C" int p2l_pack_conv_weight_pw(const float* w_oihw, int O, int I, int N_pad, int K_pad,
                                       int transpose_flip, float* w_packed, void* stream) {
  int rc = p2l_pack_conv_weight(w_oihw, O, I, 1, N_pad, K_pad, transpose_flip, w_packed, stream);
  if (rc) return rc;
  const size_t total = (size_t)K_pad * N_pad;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, w_packed + total, O, I, 1, N_pad, K_pad, 16,
                     transpose_flip, 1);
  rc = p2l_check_launch();
  if (rc) return rc;
  return p2l_pw_pack_h2(w_oihw, O, I, N_pad, K_pad, transpose_flip, w_packed + total + total * 3 / 2,
                        (hipStream_t)stream);
}

// P2L_WFMT_BF16X3T: the direct bf16x3 image followed by the thin image of a conv with three real
// channels on one side (p2l_thin.hip); N_pad == 32 (3 outputs) or K_pad == 16 (3 inputs)
extern "C" int p2l_pack_conv_weight_bf3t(const float* w_oihw, int O, int I, int N_pad, int K_pad,
                                         int transpose_flip, float* w_packed, void* stream) {
  int rc = p2l_pack_conv_weight_bf3(w_oihw, O, I, 9, N_pad, K_pad, transpose_flip, w_packed, stream);
  if (rc) return rc;
  return p2l_thin_pack(w_oihw, O, I, N_pad, K_pad, transpose_flip,
                       w_packed + (size_t)9 * N_pad * K_pad * 3 / 2, (hipStream_t)stream);
}

extern "C" int p2l_pack_conv_weight_bf3w(const float* w_oihw, int O, int I, int taps, int N_pad,
                                         int K_pad, int transpose_flip, float* w_packed,
                                         void* stream) {
  int rc = p2l_pack_conv_weight_bf3(w_oihw, O, I, taps, N_pad, K_pad, transpose_flip, w_packed,
                                    stream);
  if (rc) return rc;
  float* next = w_packed + (size_t)taps * N_pad * K_pad * 3 / 2;
  if (p2l_wino_weight_ok(N_pad, K_pad)) {
    rc = p2l_wino_pack(w_oihw, O, I, N_pad, K_pad, transpose_flip, next, (hipStream_t)stream);
    if (rc) return rc;
    next += p2l_wino_weight_floats(N_pad, K_pad) + p2l_wino_h2_weight_floats(N_pad, K_pad);
  }
  // the fp16 x 2 image of the direct kernel's weight tile (p2l_h2.hip) behind everything else
  return p2l_h2_pack(w_oihw, O, I, N_pad, K_pad, transpose_flip, -1, next, (hipStream_t)stream);
}

// sub-pixel weights of a P2L_WFMT_BF16X3W model: the bf16 x 3 image of p2l_pack_conv_weight_subpix_bf3
// followed by the fp16 x 2 image of the same 16 phase-tap slabs
extern "C" size_t p2l_packed_subpix_weight_floats(int N_pad, int K_pad, int wfmt) {
  const size_t n = (size_t)16 * N_pad * K_pad;
  if (wfmt == P2L_WFMT_F32) return n;
  return n * 3 / 2 + (wfmt == P2L_WFMT_BF16X3W ? p2l_h2_weight_floats(N_pad, K_pad, 1) : 0);
}
extern "C" int p2l_pack_conv_weight_subpix_h2(const float* w_oihw, int O, int I, int N_pad, int K_pad,
                                              int transpose_flip, int mode, float* w_packed,
                                              void* stream) {
  int rc = p2l_pack_conv_weight_subpix_bf3(w_oihw, O, I, N_pad, K_pad, transpose_flip, mode, w_packed, stream);
  if (rc) return rc;
  return p2l_h2_pack(w_oihw, O, I, N_pad, K_pad, transpose_flip, mode,
                     w_packed + (size_t)16 * N_pad * K_pad * 3 / 2, (hipStream_t)stream);
}

extern "C" int p2l_pack_conv_weight_bf3(const float* w_oihw, int O, int I, int taps, int N_pad,
                                        int K_pad, int transpose_flip, float* w_packed,
                                        void* stream) {
  if (!w_oihw || !w_packed || taps != 9) return P2L_EINVAL;
  const int N = transpose_flip ? I : O, K = transpose_flip ? O : I;
  if (K_pad % 16 || N_pad % 32 || N_pad < N || K_pad < K) return P2L_EINVAL;
  const size_t total = (size_t)taps * K_pad * N_pad;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(cdiv(total, 256)), dim3(256),
                     0, (hipStream_t)stream, w_oihw, w_packed, O, I, taps, N_pad,
                     K_pad, 16, transpose_flip, 1);
  return p2l_check_launch();
}
