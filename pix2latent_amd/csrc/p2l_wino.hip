// Winograd F(2x2, 3x3) form of the stride-1 3x3 convolution on the bf16 matrix pipe, in the
// fp32-equivalent 3-way-split arithmetic of P2L_WFMT_BF16X3 (p2l_conv.hip).
//
// Same role as conv_mfma_kernel<TAPS=9,BF3> -- nn.Conv2d forward and input-gradient of the
// BigGAN-deep GenBlock 3x3 convs and the VGG16 features (reached from
// pix2latent/model/biggan.py:58 and pix2latent/loss_functions.py:142 in the reference) -- with
// 2.25x fewer matrix-pipe products: a 2x2 output quad needs 16 multiplies per (cin, cout)
// pair in the transform domain instead of 36.
//
//     Y = A^T [ (G g G^T)  (.)  (B^T d B) ] A        d: 4x4 input patch, g: 3x3 kernel
//
// Layout of the work (MI355X-first; two blocks of 4 waves per CU):
//   * block = 8x16 output pixels of one image = 32 quads ("tiles", GEMM M; the same patch and
//     the same 2x2-quad notion as the direct kernel), 64 output channels (GEMM N), K-chunks
//     of 16 input channels;
//   * wave w owns the transform-domain frequencies 4w .. 4w+3 for all 32 tiles and 64 channels:
//     4 freq x 2 N-tiles accumulators of 32x32 (128 VGPR).  Frequencies never mix before the
//     output transform, so a weight fragment is needed by exactly one wave: the
//     pre-transformed, pre-split weights are laid out in MFMA B-fragment order and go
//     global -> registers (1 KB coalesced per wave-load, one frequency ahead), never
//     through LDS;
//   * per chunk the 10x18x16 input patch is staged once in LDS as fp32 (prologue affine /
//     ReLU applied while staging), then every thread transforms one (tile, 4 channels,
//     half of the frequencies) -- 16 ds_read_b128, 32 adds -- splits the 8 results into three
//     bf16 pieces and writes them into the [frequency][tile] operand image (49 KB, the
//     96-byte swizzled row format of the direct kernel: A-fragment reads are conflict-free);
//     66 KB of LDS per block: while one block transforms, the CU's other block multiplies;
//   * 6 v_mfma_f32_32x32x16_bf16 per (frequency, N-tile, chunk), fp32 accumulate;
//   * epilogue: the accumulators of the 4 waves meet in LDS (2 passes of 32 channels), every
//     thread inverse-transforms one (tile, 4 channels) and runs the shared epilogue item
//     (p2l_conv_k.h: bias / residual / activation / mask / pooling / fused activation backward).
// Numerics: transforms are exact-rounded fp32 additions (entries of B, A are 0, +-1), G g G^T
// is formed in fp64 at pack time; results match the direct kernel to fp32 rounding
// (tests/test_kernels_gpu.py runs both against the same tolerance).
#include "p2l_conv_k.h"

#include <atomic>
#include <type_traits>

using namespace p2lconv;

namespace {

constexpr int WN_THREADS = 256;
constexpr int WN_RAW_PITCH = 24;                    // floats per staged pixel (16 channels + pad):
                                                    // conflict-free ds_read_b128 in the transform
constexpr int WN_RAW_ROWS = 10 * 18;                // 8x16 outputs + halo
constexpr int WN_RAW_FLOATS = WN_RAW_ROWS * WN_RAW_PITCH;
constexpr int WN_V_ROWS = 16 * 32;                  // [frequency][tile], 96 B each
constexpr int WN_DUMP_PITCH = 32;                   // floats per (frequency, tile) in the epilogue
constexpr size_t WN_LDS_BYTES = (size_t)(WN_RAW_FLOATS + WN_V_ROWS * 24) * sizeof(float);
static_assert(16 * 32 * WN_DUMP_PITCH <= WN_RAW_FLOATS + WN_V_ROWS * 24, "epilogue dump fits");

__device__ __forceinline__ void store_split(float* Vs, int row, int v, const f32x4 x) {
  bf16x4 ph, pm, pl;
  split3(x, ph, pm, pl);
  char* rb = reinterpret_cast<char*>(Vs) + row * 96 + (v & 1) * 8;
  char* rq = rb + bf3_chunk(v >> 1, row) * 16;      // pieces at +0 / +32 / +64 bytes
  *reinterpret_cast<bf16x4*>(rq) = ph;
  *reinterpret_cast<bf16x4*>(rq + 32) = pm;
  *reinterpret_cast<bf16x4*>(rq + 64) = pl;
}

template <int PRO>
__global__ __launch_bounds__(WN_THREADS, 2) void wino_conv_kernel(const ConvK k) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw = smem;
  float* Vs = smem + WN_RAW_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  // ---- which block --------------------------------------------------------------------
  const int swz = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = swz / k.n_ntiles, nt = swz - mt * k.n_ntiles;
  const int tiles_per_image = k.tiles_x * k.tiles_y;
  const int b = mt / tiles_per_image;
  const int tile_in_image = mt - b * tiles_per_image;
  const int by = tile_in_image / k.tiles_x, bx = tile_in_image - by * k.tiles_x;
  const int y0 = by * 8, x0 = bx * 16, n0 = nt * 64;

  // ---- staging descriptors: 180 pixels x 4 channel quads over 256 threads -------------
  constexpr int A_ITERS = 3;
  const int sv = tid & 3;                            // channel quad: the same for every item
  int a_goff[A_ITERS], a_loff[A_ITERS];
  unsigned a_valid = 0;
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int p = (tid + WN_THREADS * it) >> 2;
    a_goff[it] = 0;
    a_loff[it] = (p < WN_RAW_ROWS) ? p * WN_RAW_PITCH + sv * 4 : -1;
    if (p < WN_RAW_ROWS) {
      const int hy = p / 18, hx = p - hy * 18;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      if (iy >= 0 && iy < k.H && ix >= 0 && ix < k.W) {
        a_goff[it] = ((b * k.H + iy) * k.W + ix) * k.x_ld + sv * 4;
        a_valid |= 1u << it;
      }
    }
  }
  const int s_off = b * k.pro_bstride + sv * 4;
  f32x4 xr[A_ITERS], sr, tr;
  auto load_raw = [&](int c) {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it)
      xr[it] = *reinterpret_cast<const f32x4*>(k.x + (size_t)a_goff[it] + c * 16);
    if (PRO != P2L_PRO_NONE) {
      sr = *reinterpret_cast<const f32x4*>(k.pro_s + s_off + c * 16);
      tr = *reinterpret_cast<const f32x4*>(k.pro_t + s_off + c * 16);
    }
  };
  auto write_raw = [&]() {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      if (a_loff[it] < 0) continue;
      f32x4 v = xr[it];
      if (PRO != P2L_PRO_NONE) {
        v = v * sr + tr;
        if (PRO == P2L_PRO_AFFINE_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
          v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
      }
      if (!((a_valid >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};   // zero padding AFTER the prologue
      *reinterpret_cast<f32x4*>(raw + a_loff[it]) = v;
    }
  };

  // ---- input transform item of this thread: (half h, tile tt, channel quad tv) ----------
  const int th = tid >> 7, tt = (tid >> 2) & 31, tv = tid & 3;
  const int tty = tt >> 3, ttx = tt & 7;
  // frequency row i of B^T d is one signed sum of two patch rows:
  //   row 0 = d0 - d2 ,  row 1 = d1 + d2 ,  row 2 = d2 - d1 ,  row 3 = d1 - d3
  // half h of the block's threads produces rows 2h, 2h+1 (h is wave-uniform)
  const float* t_src = raw + (2 * tty * 18 + 2 * ttx) * WN_RAW_PITCH + tv * 4;
  auto transform = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int fr = 2 * th + i;
      const int ra = (fr == 0) ? 0 : (fr == 2 ? 2 : 1);
      const int rb = (fr == 2) ? 1 : (fr == 3 ? 3 : 2);
      const float sb = (fr == 1) ? 1.f : -1.f;
      f32x4 R[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(t_src + (ra * 18 + c) * WN_RAW_PITCH);
        const f32x4 bq = *reinterpret_cast<const f32x4*>(t_src + (rb * 18 + c) * WN_RAW_PITCH);
        R[c] = a + sb * bq;                            // +-1: exact
      }
      const int f = fr * 4;
      store_split(Vs, (f + 0) * 32 + tt, tv, R[0] - R[2]);
      store_split(Vs, (f + 1) * 32 + tt, tv, R[1] + R[2]);
      store_split(Vs, (f + 2) * 32 + tt, tv, R[2] - R[1]);
      store_split(Vs, (f + 3) * 32 + tt, tv, R[1] - R[3]);
    }
  };

  // ---- weight fragments: global -> registers --------------------------------------------
  // image: [chunk][frequency][32-channel tile of Cout][piece][lane] x 16 B
  const int n_t32 = k.Cout >> 5;
  const f32x4* wq = reinterpret_cast<const f32x4*>(k.w);
  f32x4 bw[2][2][3];                                   // [set][N-tile][piece]
  auto load_b = [&](int c, int fi, int set) {
    const size_t base = (((size_t)c * 16 + (4 * wave + fi)) * n_t32 + (n0 >> 5)) * 3 * 64 + lane;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) bw[set][j][p] = wq[base + (size_t)(j * 3 + p) * 64];
  };

  f32x16 acc[4][2];                                    // [freq][N-tile]
#pragma unroll
  for (int fi = 0; fi < 4; ++fi)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fi][j][r] = 0.f;

  const int nchunks = k.nchunks;
  // weight fragments are fetched ONE frequency ahead into two alternating register sets
  // (48 VGPR); only the set of frequency 0 is live across the staging / transform phases
  load_raw(0);
  load_b(0, 0, 0);
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    write_raw();
    __syncthreads();
    if (more) load_raw(c + 1);
    transform();
    load_b(c, 1, 1);                 // frequency 1's fragments: in flight across the barrier
    __syncthreads();
    bf16x8 af[2][3];
    auto lda = [&](int fi, bf16x8 (&a)[3]) {
      const int row = (4 * wave + fi) * 32 + l31;
      const float* aq = Vs + row * 24 + bf3_chunk(lhi, row) * 4;
      a[0] = *reinterpret_cast<const bf16x8*>(aq);
      a[1] = *reinterpret_cast<const bf16x8*>(aq + 8);
      a[2] = *reinterpret_cast<const bf16x8*>(aq + 16);
    };
    lda(0, af[0]);
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) {
      if (fi >= 1 && fi + 1 < 4) load_b(c, fi + 1, (fi + 1) & 1);
      else if (fi == 3 && more) load_b(c + 1, 0, 0);
      if (fi + 1 < 4) lda(fi + 1, af[(fi + 1) & 1]);     // next frequency's A fragments in flight
      const bf16x8 a1 = af[fi & 1][0], a2 = af[fi & 1][1], a3 = af[fi & 1][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, bw[fi & 1][j][0]);
        const bf16x8 b2 = __builtin_bit_cast(bf16x8, bw[fi & 1][j][1]);
        const bf16x8 b3 = __builtin_bit_cast(bf16x8, bw[fi & 1][j][2]);
        f32x16 t = acc[fi][j];                         // (the order of wino16s_conv_kernel: bit-identical results)
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, t, 0, 0, 0);
        acc[fi][j] = t;
      }
    }
    // the next write_raw() touches only `raw` (free since the transform barrier); the barrier
    // after it orders every wave's MFMA reads of Vs before the next transform's writes
  }
  __syncthreads();

  // ---- epilogue: 2 passes of 32 output channels --------------------------------------------
  // dump[f][tile][32 channels] <- C layout: lane = channel column, register r = tile row
  // (r&3) + 8*(r>>2) + 4*lhi.  Writes are lane-contiguous, item reads 128 B per tile contiguous.
  float* dump = smem;
  EpiSums S;
  const int e_t = tid >> 3, e_c4 = tid & 7;             // item: (tile, 4 channels)
  const int ety = e_t >> 3, etx = e_t & 7;
#pragma unroll
  for (int j = 0; j < 2; ++j) {           // (compile-time: the accumulators must stay in registers)
#pragma unroll
    for (int fi = 0; fi < 4; ++fi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tile = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        dump[((4 * wave + fi) * 32 + tile) * WN_DUMP_PITCH + l31] = acc[fi][j][r];
      }
    __syncthreads();
    const int nb = n0 + j * 32;
    if (nb + e_c4 * 4 < k.n_store) {
      f32x4 T[2][4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {                  // A^T M, column jj
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(dump + ((0 + jj) * 32 + e_t) * WN_DUMP_PITCH + e_c4 * 4);
        const f32x4 m1 = *reinterpret_cast<const f32x4*>(dump + ((4 + jj) * 32 + e_t) * WN_DUMP_PITCH + e_c4 * 4);
        const f32x4 m2 = *reinterpret_cast<const f32x4*>(dump + ((8 + jj) * 32 + e_t) * WN_DUMP_PITCH + e_c4 * 4);
        const f32x4 m3 = *reinterpret_cast<const f32x4*>(dump + ((12 + jj) * 32 + e_t) * WN_DUMP_PITCH + e_c4 * 4);
        T[0][jj] = (m0 + m1) + m2;
        T[1][jj] = (m1 - m2) - m3;
      }
      f32x4 v[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {                     // (A^T M) A
        v[2 * i + 0] = ((T[i][0] + T[i][1]) + T[i][2]) * k.alpha;
        v[2 * i + 1] = ((T[i][1] - T[i][2]) - T[i][3]) * k.alpha;
      }
      epi_item(k, v, b, y0 + 2 * ety, x0 + 2 * etx, nb + e_c4 * 4, 0, 0, 0, S);
    }
    if (k.arb_x != nullptr) {
      // the block IS tile (by, bx) of the 128-pixel tiling the caller's partial sums use
      epi_arb_reduce<32, 8>(k, S, smem, wave, lane, tid,
                            (size_t)b * k.arb_nblk + tile_in_image, nb);
      S = EpiSums();
    }
    __syncthreads();                                   // dump is rewritten by the next pass
  }
}

// ---- 16x16-pixel blocks, 8 waves -----------------------------------------------------------
// The 8x16-pixel kernel above fetches the chunk's whole transform-domain weight slice (96 KB
// for 64 output channels) once per 32 tiles: with two blocks per CU that alone is ~64 B/clk,
// the L2 -> L1 rate of a CU, and its phases (weight wait | transform + split | MFMA) run one
// after the other: MFMA pipe 26-34 % busy.  The 16x16-pixel form:
//   * block = 16x16 output pixels = 64 tiles (2 M-tiles) x 64 output channels, 8 waves; wave w
//     owns frequencies 2w, 2w+1: 2 freq x 2 M x 2 N accumulators (128 VGPR).  A weight
//     fragment serves two M-tiles: half the L2 traffic per product;
//   * the transformed input V stays FP32 in LDS ([frequency][tile][16 channels], 64 KB, double
//     buffered, 16-byte slots XOR-swizzled by the tile index) and is split into bf16 pieces by
//     its CONSUMER: each (frequency, M-tile) fragment is read by exactly one wave;
//   * two barriers per chunk (V buffers swap | the fp32 patch is rewritten).
// Same additions and products in the same order as the 8x16 kernel: bit-identical results.
// a wave-uniform pointer pinned to scalar registers: hipcc then addresses `p + lane offset` as
// global_load v_off, s[base] instead of carrying a 64-bit address per lane and load
__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

constexpr int W16_THREADS = 512;
constexpr int W16_RAW_ROWS = 18 * 18;
constexpr int W16_RAW_FLOATS = W16_RAW_ROWS * WN_RAW_PITCH;            // 31,104 B
constexpr int W16_V_FLOATS = 16 * 64 * 16;                            // one buffer: 64 KB
constexpr size_t W16_LDS_BYTES = (size_t)(W16_RAW_FLOATS + 2 * W16_V_FLOATS) * sizeof(float);
static_assert(W16_LDS_BYTES <= 160 * 1024, "one block per CU");
static_assert(16 * 64 * WN_DUMP_PITCH <= 2 * W16_V_FLOATS, "epilogue dump fits");
// the 4-wave block of the fp16 x 2 form (MT = 1): 10x18 patch rows, 32 tiles
constexpr size_t W8_LDS_BYTES = (size_t)(10 * 18 * WN_RAW_PITCH + 2 * 16 * 32 * 16) * sizeof(float);

// ---- 16x16-pixel blocks, hand-scheduled multiply phase ------------------------------------
// Left to hipcc, a multiply step comes out as [read fragment | 36 VALU of the 3-way split | 12
// MFMAs back to back] (round 2: MFMA pipe 34 % busy, 43 % of wave cycles waiting to issue): the
// two waves of a SIMD leave every barrier in phase, split at the same time and then queue their
// MFMAs at the same time.  What the SIMD can overlap was measured (tools/micro/issue_rate.hip):
// up to 4 plain VALU instructions per MFMA and wave are free, v_cvt_pk_bf16_f32 is half rate,
// and a packed fp32 add holds the matrix pipe for ~10 cycles.  So the order is fixed by hand
// (sched_barrier between every MFMA and the few other instructions that follow it), the TU is
// built without packed fp32 instructions, and:
//   * the products of a fragment are taken LARGEST PIECE FIRST: h b1, h b2, h b3, m b1, m b2,
//     l b1 -- the h pieces are four v_cvt_pk away from the fp32 values, so the first MFMA of a
//     step issues almost at once and the m / l pieces are computed in the gaps behind the MFMAs
//     that do not need them yet (one pair of values = 4 VALU per gap);
//   * the h pieces of the NEXT fragment, the input transform of the next chunk, the patch
//     write and every load sit in the remaining gaps of the step; nothing but the fragment
//     read + 4 conversions at the top of a chunk is outside an MFMA shadow.
// Chunk period 6 500 -> 4 950 cycles (3 072 of MFMA work per SIMD); tools/micro/conv_lab.cpp.
// Transform micro-operation of MFMA gap G = 12 * step + gap of a chunk: part * 8 + op, or -1.
// Usable gaps: 0-8 and 10 of every step, from gap 8 of step 0 (the patch barrier) on.  Per part:
// request | (wait) | rows + request | (wait) | rows | columns (1 output for even parts, 3 for odd).
struct TxTable {
  int slot[48];
  constexpr TxTable() : slot{} {
    constexpr int seq[28] = {0 * 8 + 0, -1, 0 * 8 + 1, -1, 0 * 8 + 2, 0 * 8 + 3,
                             1 * 8 + 0, -1, 1 * 8 + 1, -1, 1 * 8 + 2, 1 * 8 + 3, 1 * 8 + 4, 1 * 8 + 5,
                             2 * 8 + 0, -1, 2 * 8 + 1, -1, 2 * 8 + 2, 2 * 8 + 3,
                             3 * 8 + 0, -1, 3 * 8 + 1, -1, 3 * 8 + 2, 3 * 8 + 3, 3 * 8 + 4, 3 * 8 + 5};
    int n = 0;
    for (int g = 0; g < 48; ++g) {
      const int q = g % 12;
      slot[g] = -1;
      if (g < 8 || q == 9 || q == 11) continue;
      if (n < 28) slot[g] = seq[n];
      ++n;
    }
  }
};
constexpr TxTable kTx{};

// ABL (lab builds only, -DP2L_LAB): timing ablations -- results are wrong when set.
//   1 weights loaded once | 2 no input transform | 4 no barriers in the loop | 8 no m / l pieces
//   16 no MFMAs | 32 fragment values read once | 64 no patch loads / writes
// H2: the fp16 x 2 arithmetic (include/p2l.h, P2L_WFMT_BF16X3W): the staged patch is scaled by the
// image's power of two, a fragment is split into two fp16 pieces (cvt_pk | 2 v_fma_mix | cvt_pk per
// pair: 4 instructions instead of 11) and multiplied by three MFMAs per N-tile instead of six.
// Transform micro-operations of the H2 form: MFMA gap G = 6 * step + gap; every gap behind the
// patch barrier (gap 3 of step 0) carries one of the 20.
struct TxTableH {
  int slot[24];
  constexpr TxTableH() : slot{} {
    constexpr int seq[20] = {0 * 8 + 0, 0 * 8 + 1, 0 * 8 + 2, 0 * 8 + 3,
                             1 * 8 + 0, 1 * 8 + 1, 1 * 8 + 2, 1 * 8 + 3, 1 * 8 + 4, 1 * 8 + 5,
                             2 * 8 + 0, 2 * 8 + 1, 2 * 8 + 2, 2 * 8 + 3,
                             3 * 8 + 0, 3 * 8 + 1, 3 * 8 + 2, 3 * 8 + 3, 3 * 8 + 4, 3 * 8 + 5};
    for (int g = 0; g < 24; ++g) slot[g] = g < 4 ? -1 : seq[g - 4];
  }
};
constexpr TxTableH kTxH{};
// MT = M-tiles of 32 quads per block: 2 = 16x16 pixels, 8 waves (wave w: frequencies 2w, 2w+1, both
// M-tiles); 1 = 8x16 pixels, 4 waves (wave w: frequencies 4w .. 4w+3) -- round 5, the small-batch form
// of the fp16 x 2 arithmetic: at 2-3 candidates per GPU a 32^2 / 64^2 layer is 48-128 blocks of 16x16 on
// 256 CUs and the launch lasts one block's latency; halved blocks are twice as many and, with ONE wave per
// SIMD instead of two sharing its issue slots, about half as long.  The launcher picks it from the grid
// size.  Re-tiling along M changes no output's summation order: same transform, same products, same
// chunk order, same partial sums per 8x16-pixel tile, same maxima slots -- bit-identical to the 16x16 form
// (tests/test_kernels_gpu.py).  A step of the multiply phase is fragment (fi, m) = (s / MT, s % MT).
template <int PRO, int ABL = 0, bool H2 = false, int MT = 2>
__global__ __launch_bounds__(256 * MT, 1) void wino16s_conv_kernel(const ConvK k) {
  static_assert(MT == 2 || (MT == 1 && H2), "the 4-wave block exists in the fp16 x 2 arithmetic only");
  constexpr int THREADS = 256 * MT, TILES = 32 * MT, NW = 4 * MT, NF = 4 / MT;
  constexpr int RAW_ROWS = (8 * MT + 2) * 18, RAW_FLOATS = RAW_ROWS * WN_RAW_PITCH, V_FLOATS = 16 * TILES * 16;
  constexpr int FV = TILES * 16;                         // floats of one frequency of V
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* raw = smem;
  float* Vs = smem + RAW_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;

  const int swz = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = swz / k.n_ntiles, nt = swz - mt * k.n_ntiles;
  const int tiles_per_image = k.tiles_x * k.tiles_y;                  // (8 MT)x16-pixel tiles
  const int b = __builtin_amdgcn_readfirstlane(mt / tiles_per_image);   // (scalar address bases)
  const int tile_in_image = mt - b * tiles_per_image;
  const int by = tile_in_image / k.tiles_x, bx = tile_in_image - by * k.tiles_x;
  const int y0 = by * (8 * MT), x0 = bx * 16, n0 = __builtin_amdgcn_readfirstlane(nt * 64);

  // split-K (blockIdx.y = slice z of the input channels): chunks c_lo .. c_lo + nchunks - 1 of the
  // layer; the un-scaled partial outputs go to k.ws[z] and conv_splitk_finish adds the slices in
  // fixed order and runs the epilogue (p2l_conv.hip).  The slice count is a function of the LAYER
  // SHAPE only (p2l_wino_split_factor), like the choice of the Winograd form itself.
  const int z = blockIdx.y;
  const int c_lo = z * k.chunks_per_split;
  const int nchunks = min(k.chunks_per_split, k.nchunks - c_lo);

  float x_scale = 1.f, out_scale = 1.f;   // fp16 x 2: the image's power of two and its inverse (set below, H2)

  // ---- staging: 324 pixels x 4 channel quads over 512 threads ---------------------------
  constexpr int A_ITERS = 3;
  const int sv = tid & 3;
  // item `it` of a thread is pixel (tid >> 2) + 128 * it: LDS offsets differ by a constant
  unsigned a_goff[A_ITERS];                              // (scalar base + 32-bit lane offset)
  const int a_loff0 = (tid >> 2) * WN_RAW_PITCH + sv * 4;
  const bool a_third = tid < 4 * (RAW_ROWS - THREADS / 2);      // items 0, 1 always exist
  unsigned a_valid = 0;
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int p = (tid + THREADS * it) >> 2;
    a_goff[it] = 0x80000000u;
    if (p < RAW_ROWS) {
      const int hy = p / 18, hx = p - hy * 18;
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      if (iy >= 0 && iy < k.H && ix >= 0 && ix < k.W) {
        a_goff[it] = ((iy * k.W + ix) * k.x_ld + sv * 4) * 4;     // bytes inside image b (< 2^32)
        a_valid |= 1u << it;
      }
    }
  }
  const char* ps_img = reinterpret_cast<const char*>(k.pro_s + (size_t)b * k.pro_bstride);
  const char* pt_img = reinterpret_cast<const char*>(k.pro_t + (size_t)b * k.pro_bstride);
  const unsigned s_off = sv * 16;
  f32x4 xr[A_ITERS], sr, tr;
  // image b as a buffer resource: 32-bit lane offsets + the chunk as the scalar offset; lanes
  // of the zero padding carry an offset past the end and read 0.f
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(sgpr_ptr(reinterpret_cast<const char*>(k.x + (size_t)b * k.H * k.W * k.x_ld))), 0,
      __builtin_amdgcn_readfirstlane(k.H * k.W * k.x_ld * 4), 0x00020000);
  auto load_raw = [&](int c) {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it)
      xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, a_goff[it], (c_lo + c) * 64, 0));
    if (PRO != P2L_PRO_NONE) {
      sr = *reinterpret_cast<const f32x4*>(ps_img + (c_lo + c) * 64 + s_off);
      tr = *reinterpret_cast<const f32x4*>(pt_img + (c_lo + c) * 64 + s_off);
    }
  };
  auto write_raw1 = [&](int it) {
    if (it == 2 && !a_third) return;
    f32x4 v = xr[it];
    if (PRO != P2L_PRO_NONE) {
      v = v * sr + tr;
      if (PRO == P2L_PRO_AFFINE_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
    }
    if (PRO != P2L_PRO_NONE && !((a_valid >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};   // padding is 0 AFTER the prologue
    if (H2) v = v * x_scale;
    *reinterpret_cast<f32x4*>(raw + a_loff0 + it * (THREADS / 4) * WN_RAW_PITCH) = v;
  };

  // ---- input transform item: (half h, tile tt of 64, channel quad tv) --------------------
  const int th = __builtin_amdgcn_readfirstlane(tid / (THREADS / 2));
  const int tt = (tid >> 2) & (TILES - 1), tv = tid & 3;
  const int tty = tt >> 3, ttx = tt & 7;
  const float* t_src = raw + (2 * tty * 18 + 2 * ttx) * WN_RAW_PITCH + tv * 4;
  const int t_dst = tt * 16 + ((tv ^ ((tt >> 2) & 3)) << 2);
  // A part = one frequency row x two patch columns (c0, c0 + 2): t_load(part, h) requests the two
  // patch rows of column c0 + 2h, t_rows(part, h) combines them into Ra (h = 0) / Rb (h = 1)
  f32x4 tq[2], tRa, tRb;
  f32x4 tR2;                                            // R[2] of the current frequency row
  auto t_load = [&](int part, int h) {
    const int fr = 2 * th + (part >> 1);
    const int ra = (fr == 0) ? 0 : (fr == 2 ? 2 : 1);
    const int rb = (fr == 2) ? 1 : (fr == 3 ? 3 : 2);
    const int c0 = (part & 1) + 2 * h;
    tq[0] = *reinterpret_cast<const f32x4*>(t_src + (ra * 18 + c0) * WN_RAW_PITCH);
    tq[1] = *reinterpret_cast<const f32x4*>(t_src + (rb * 18 + c0) * WN_RAW_PITCH);
  };
  // (a + sign * b as one fma per value: the sign is a wave-uniform float, no select)
  const float t_sgn[2] = {(2 * th + 0 == 1) ? 1.f : -1.f, (2 * th + 1 == 1) ? 1.f : -1.f};
  auto t_rows = [&](int part, int h) {
    const float sg = t_sgn[part >> 1];
    const f32x4 r = {__builtin_fmaf(tq[1].x, sg, tq[0].x), __builtin_fmaf(tq[1].y, sg, tq[0].y),
                     __builtin_fmaf(tq[1].z, sg, tq[0].z), __builtin_fmaf(tq[1].w, sg, tq[0].w)};
    if (h == 0) tRa = r; else tRb = r;
  };
  // column combinations + stores: which = 0 the only output of an even part / R[1] of an odd one,
  // 1, 2 = R[2], R[3] of an odd part
  auto t_cols = [&](int part, int which, float* Vn) {
    const int fr = 2 * th + (part >> 1);
    float* d = Vn + fr * 4 * FV + t_dst;
    const f32x4 Ra = tRa, Rb = tRb;
    if ((part & 1) == 0) {                              // R[0]; Rb is R[2]'s second operand
      *reinterpret_cast<f32x4*>(d) = Ra - Rb;
      tR2 = Rb;
    } else if (which == 0) {
      *reinterpret_cast<f32x4*>(d + FV) = Ra + tR2;
    } else if (which == 1) {
      *reinterpret_cast<f32x4*>(d + 2 * FV) = tR2 - Ra;
    } else {
      *reinterpret_cast<f32x4*>(d + 3 * FV) = Ra - Rb;
    }
  };

  // ---- weight fragments: global -> registers, one frequency ahead -------------------------
  const int n_t32 = k.Cout >> 5;
  // (buffer resource over the whole image: lane offset in a register, everything else scalar)
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(sgpr_ptr(reinterpret_cast<const char*>(k.w))), 0,
      __builtin_amdgcn_readfirstlane(k.Cout * k.nchunks * (16 * (H2 ? 64 : 96))), 0x00020000);
  const int w_lane = lane * 16;
  constexpr int NP = H2 ? 2 : 3;                       // pieces of a weight
  // (a second bank for MT = 1 -- the next chunk's fragments requested a whole chunk ahead -- put 128 more
  //  registers under the loop: spills; instead the 4-wave block re-fills a frequency's registers right
  //  behind the last MFMA that reads them, see stepH)
  constexpr int NBK = 1;
  f32x4 bw[NBK][NF][2][NP];                            // [bank][frequency of the wave][N-tile][piece]
  auto load_b = [&](int c, int fi, int bank) {
    const int base = (((c_lo + c) * 16 + (NF * wave + fi)) * n_t32 + (n0 >> 5)) * (NP * 64 * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p)
        bw[bank][fi][j][p] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, w_lane + p * 1024, base + j * (NP * 1024), 0));
  };

  f32x16 acc[NF][MT][2];                               // [freq][M-tile][N-tile]
#pragma unroll
  for (int fi = 0; fi < NF; ++fi)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fi][m][j][r] = 0.f;

  // ---- fragment pipeline registers ---------------------------------------------------------
  const int a_sw = (l31 >> 2) & 3;
  const int a_off0 = l31 * 16 + (((lhi * 2) ^ a_sw) << 2), a_off1 = l31 * 16 + (((lhi * 2 + 1) ^ a_sw) << 2);
  // (one set: the values of fragment s+1 are requested once the residuals of fragment s are
  //  dead, its h pieces are formed once the last h product of fragment s has issued)
  f32x2 rr[4];                                         // fp32 values -> running residuals, [pair]
  bf16x2 hh[4];                                        // h pieces
  bf16x2 mm[4], ll[4];                                 // m, l pieces of the current fragment
  auto lda = [&](const float* Vc, int s) {
    const float* rowp = Vc + ((NF * wave + s / MT) * TILES + (s % MT) * 32) * 16;
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(rowp + a_off0);
    const f32x4 q1 = *reinterpret_cast<const f32x4*>(rowp + a_off1);
    rr[0] = f32x2{q0.x, q0.y}; rr[1] = f32x2{q0.z, q0.w};
    rr[2] = f32x2{q1.x, q1.y}; rr[3] = f32x2{q1.z, q1.w};
  };
  auto resid = [&](const f32x2 v, const bf16x2 piece) { return resid2(v, piece); };
  auto hstage = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      hh[p] = __builtin_convertvector(rr[p], bf16x2);
      asm volatile("" : "+v"(hh[p]));                  // (stays in its gap: see P2L_PIN)
    }
  };
  auto cat8 = [](const bf16x2 (&q)[4]) {
    const bf16x4 lo = __builtin_shufflevector(q[0], q[1], 0, 1, 2, 3);
    const bf16x4 hi = __builtin_shufflevector(q[2], q[3], 0, 1, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
#define P2L_SB() __builtin_amdgcn_sched_barrier(0)
// (hipcc sinks a conversion to its first use, four gaps later, in a clump: pin it to its gap)
#define P2L_PIN(X) asm volatile("" : "+v"(X))
#define P2L_MF(A, FI, J, P, M)                                                                \
  if (!(ABL & 16))                                                                            \
    acc[FI][M][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                  \
        A, __builtin_bit_cast(bf16x8, bw[0][FI][J][P]), acc[FI][M][J], 0, 0, 0);              \
  P2L_SB()

#ifdef P2L_LAB
  // phase timestamps of one block: [wave][chunk][8] s_memtime ticks (slot 0 chunk top, 1-4 after
  // steps 0-3, 5 after the closing barrier)
  unsigned long long* trace =
      (k.ws != nullptr && k.splitk <= 1 && swz == (int)(gridDim.x / 2)) ? reinterpret_cast<unsigned long long*>(k.ws) : nullptr;
#define P2L_TR(SLOT, C)                                                                      \
  if (trace != nullptr && lane == 0 && (C) < 64) trace[(wave * 64 + (C)) * 8 + (SLOT)] = __builtin_amdgcn_s_memtime()
#else
#define P2L_TR(SLOT, C)
#endif
  // The input transform of the NEXT chunk rides along as micro-operations of <= 4 VALU each, one
  // per MFMA gap (gaps 9 and 11 of a step carry the loads and the next h pieces), from the
  // barrier of step 0 on; a request and its first use are two gaps apart.
  //   op 0: request column pair 0 | 1: combine its rows, request pair 1 | 2: combine | 3..5: columns
  auto tx = [&](int part, int op, float* Vn) {
    if (op == 0) t_load(part, 0);
    else if (op == 1) { t_rows(part, 0); t_load(part, 1); }
    else if (op == 2) t_rows(part, 1);
    else t_cols(part, op - 3, Vn);
  };
#define P2L_TX(G)                                                                             \
  if (more && !(ABL & 2) && kTx.slot[G] >= 0) tx(kTx.slot[G] >> 3, kTx.slot[G] & 7, Vn)
  // one multiply step: fragment (fi, m) = (s >> 1, s & 1) of chunk c
  // (MORE_: not the last chunk -- a compile-time flag: the last chunk is its own copy of the
  //  four steps, so the steady-state loop carries no conditional branches)
  auto step = [&](auto S_, auto MORE_, const float* Vc, float* Vn, int c) {
    constexpr int s = decltype(S_)::value;
    constexpr bool more = decltype(MORE_)::value;
    constexpr int fi = s >> 1, m = s & 1;
    const bf16x8 a1 = cat8(hh);
    P2L_MF(a1, fi, 0, 0, m);
    // gaps 0-3: m pieces pair by pair; (step 0) patch write
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (!(ABL & 8)) {
        rr[p] = resid(rr[p], hh[p]);
        mm[p] = __builtin_convertvector(rr[p], bf16x2);
        P2L_PIN(mm[p]);
      } else mm[p] = hh[p];
      if (s == 0 && p >= 1 && more && !(ABL & 64)) write_raw1(p - 1);
      P2L_TX(12 * s + p);
      P2L_SB();
      if (p == 0) { P2L_MF(a1, fi, 1, 0, m); }
      if (p == 1) { P2L_MF(a1, fi, 0, 1, m); }
      if (p == 2) { P2L_MF(a1, fi, 1, 1, m); }
      if (p == 3) { P2L_MF(a1, fi, 0, 2, m); }
    }
    const bf16x8 a2 = cat8(mm);
    // gaps 4-7: l pieces
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (!(ABL & 8)) {
        rr[p] = resid(rr[p], mm[p]);
        ll[p] = __builtin_convertvector(rr[p], bf16x2);
        P2L_PIN(ll[p]);
      } else ll[p] = hh[p];
      P2L_TX(12 * s + 4 + p);
      P2L_SB();
      if (p == 0) { P2L_MF(a1, fi, 1, 2, m); }
      if (p == 1) { P2L_MF(a2, fi, 0, 0, m); }
      if (p == 2) { P2L_MF(a2, fi, 1, 0, m); }
      if (p == 3) { P2L_MF(a2, fi, 0, 1, m); }
    }
    const bf16x8 a3 = cat8(ll);
    // gap 8: the residuals are dead: next fragment's values requested
    if (s == 0) {
      // the patch of chunk c+1 (written in gaps 1-3 by every wave) becomes visible; the MFMAs
      // issued so far keep the pipe busy while the barrier fills
      if (!(ABL & 4)) __syncthreads();
    }
    if (s + 1 < 4 && !(ABL & 32)) lda(Vc, s + 1);
    P2L_TX(12 * s + 8);
    P2L_SB();
    P2L_MF(a2, fi, 1, 1, m);
    // gap 9: weight fragments / next patch (address arithmetic + loads)
    if (s == 0 && !(ABL & 1)) load_b(c, 1, 0);
    else if (s == 2 && more && !(ABL & 1)) load_b(c + 1, 0, 0);
    else if (s == 1 && more && !(ABL & 64)) load_raw(c + 2 < nchunks ? c + 2 : c + 1);   // (last one: a harmless re-read)
    P2L_SB();
    P2L_MF(a3, fi, 0, 0, m);
    // gap 10
    P2L_TX(12 * s + 10);
    P2L_SB();
    P2L_MF(a3, fi, 1, 0, m);
    // gap 11: h pieces of the next fragment
    if (s + 1 < 4 && !(ABL & 32)) hstage();
    P2L_SB();
    P2L_TR(1 + s, c);
  };

  // ---- fp16 x 2 form of the step: 6 MFMAs per fragment ------------------------------------
  //   M0 h b1 (N-tile 0) | gaps 0-3: the m pieces pair by pair (2 v_fma_mix + 1 cvt_pk each), the
  //   patch write of step 0 | M1 h b1 (1) | M2 h b2 (0) | M3 h b2 (1) | [step 0: patch barrier]
  //   next fragment requested | M4 m b1 (0) | gap 4: loads | M5 m b1 (1) | gap 5: next h pieces.
  //   The transform of the next chunk rides in every gap behind the patch barrier (kTxH).
  h16x2 hhH[4], mmH[4];
  auto hstageH = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      hhH[p] = __builtin_convertvector(rr[p], h16x2);
      asm volatile("" : "+v"(hhH[p]));
    }
  };
  auto cat8H = [](const h16x2 (&q)[4]) {
    const h16x4 lo = __builtin_shufflevector(q[0], q[1], 0, 1, 2, 3);
    const h16x4 hi = __builtin_shufflevector(q[2], q[3], 0, 1, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  // v - piece as one v_fma_mix_f32 per value (fp16 source widened inside the instruction; it
  // shares the MFMA shadow like a plain VALU instruction: tools/micro/issue_rate.hip "fma_mix")
  // (written as fma(float(piece), -1, v) hipcc emits cvt + cvt_sdwa + 2 sub instead)
  auto residH = [&](const f32x2 v, const h16x2 piece) {
    f32x2 r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r.x) : "v"(piece), "s"(-1.f), "v"(v.x));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.y) : "v"(piece), "s"(-1.f), "v"(v.y));
    return r;
  };
#define P2L_MFH(A, FI, J, P, M)                                                               \
  if (!(ABL & 16))                                                                            \
    acc[FI][M][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(                                   \
        A, __builtin_bit_cast(h16x8, bw[BK][FI][J][P]), acc[FI][M][J], 0, 0, 0);              \
  P2L_SB()
#define P2L_TXH(G)                                                                            \
  if (more && !(ABL & 2) && kTxH.slot[G] >= 0) tx(kTxH.slot[G] >> 3, kTxH.slot[G] & 7, Vn)
  auto stepH = [&](auto S_, auto MORE_, auto BK_, const float* Vc, float* Vn, int c) {
    constexpr int s = decltype(S_)::value;
    constexpr bool more = decltype(MORE_)::value;
    constexpr int BK = decltype(BK_)::value;           // bank of THIS chunk's weight fragments
    constexpr int fi = s / MT, m = s % MT;
    const h16x8 a1 = cat8H(hhH);
    P2L_MFH(a1, fi, 0, 0, m);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (!(ABL & 8)) {
        rr[p] = residH(rr[p], hhH[p]);
        mmH[p] = __builtin_convertvector(rr[p], h16x2);
        P2L_PIN(mmH[p]);
      } else mmH[p] = hhH[p];
      if (s == 0 && p < 3 && more && !(ABL & 64)) write_raw1(p);
      P2L_TXH(6 * s + p);
      P2L_SB();
      if (p == 0) { P2L_MFH(a1, fi, 1, 0, m); }
      if (p == 1) { P2L_MFH(a1, fi, 0, 1, m); }
      if (p == 2) { P2L_MFH(a1, fi, 1, 1, m); }
    }
    const h16x8 a2 = cat8H(mmH);
    if (s == 0 && !(ABL & 4)) __syncthreads();   // the patch of chunk c+1 becomes visible
    if (s + 1 < 4 && !(ABL & 32)) lda(Vc, s + 1);   // (the residuals are dead)
    P2L_SB();
    P2L_MFH(a2, fi, 0, 0, m);
    // gap 4: weight fragments (16x16 blocks: two steps ahead of their first use) / next patch
    if (MT == 2 && s % 2 == 0 && !(ABL & 1)) {
      if (s == 0) load_b(c, 1, 0);
      else if (more) load_b(c + 1, 0, 0);
    }
    if (s == 1 && more && !(ABL & 64)) load_raw(c + 2 < nchunks ? c + 2 : c + 1);
    P2L_TXH(6 * s + 4);
    P2L_SB();
    P2L_MFH(a2, fi, 1, 0, m);
    // gap 5: h pieces of the next fragment.  8x16 blocks (one wave per SIMD: a step is ~400 cycles and a
    // fragment requested two steps ahead arrives late -- lab build, 64^2 256->256 at 2 candidates: weights
    // loaded once 30.8 us against 37.3): frequency s of the NEXT chunk goes into the registers the last
    // MFMA of this step has just read, four steps ahead of its use
    if (MT == 1 && more && !(ABL & 1)) load_b(c + 1, s, 0);
    if (s + 1 < 4 && !(ABL & 32)) hstageH();
    P2L_TXH(6 * s + 5);
    P2L_SB();
    P2L_TR(1 + s, c);
  };
#undef P2L_TXH
#undef P2L_MFH

#ifdef P2L_AB_WINO_DEPHASE              // (A/B build; not in the product: see the last sentence)
  // Round 5: the blocks of a launch run in lockstep -- one block per CU, equal work -- and reach their
  // epilogues together: 256 CUs x 32 KB per pass is an HBM write burst during which nobody multiplies (lab
  // trace: 3.5-5 k cycles to ISSUE four stores per thread).  Launches of >= 8 rounds delay the first round's
  // blocks by 0 / 1/4 / 1/2 / 3/4 of a block, so that the CUs stay out of phase for the rest of the launch:
  // 128^2 128->128 at 18 candidates 256 -> 235 us; below ~8 rounds the start-up delay (3/8 of a block on
  // average) costs what it gains (profiles/round5_wino_dephase.txt).  Timing only: no result depends on it.
  // On a second box of the pool the un-delayed launch already ran at the de-phased time (233 us): not adopted.
  if (MT == 2 && gridDim.y == 1 && gridDim.x >= 8 * 256 && blockIdx.x < 256) {
    const int q = (int)(blockIdx.x >> 3) & 3;          // (8 consecutive ids = one per XCD)
    for (int i = 0; i < q * nchunks; ++i) __builtin_amdgcn_s_sleep(19);     // (19 x 64 cycles ~ a quarter chunk-share of a block)
  }
#endif
  P2L_TR(0, 63);                                       // (lab) block phases: start | loop | epilogue | pass 1 | end
  load_raw(0);
  load_b(0, 0, 0);
  if (MT == 1) { load_b(0, 1, 0); load_b(0, 2, 0); load_b(0, 3, 0); }   // (the whole first chunk)
  // (round 4: the scale is only needed when the patch is WRITTEN, so the first patch and weight
  //  requests are in flight while the partial maxima are reduced: ~1 k cycles of every block)
  // fp16 x 2: the image's scale from the 64 partial maxima of the pass in front of the launch
  if (H2) {
    float a;
    if (k.amax_in != nullptr) {
      // maxima handed over by the launch that wrote x (P2LAmax): its per-block partials of this
      // image; a fused prologue x*s+t (ReLU or not) is bounded by max|s| max|x| + max|t|
      a = 0.f;
      for (int i = tid; i < k.amax_in_n; i += THREADS) a = fmaxf(a, k.amax_in[(size_t)b * k.amax_in_n + i]);
      float ms = 0.f, mt = 0.f;
      const bool bound = PRO != P2L_PRO_NONE && !k.amax_in_applied;   // (applied: the maxima ARE those of x*s+t)
      if (bound) {
        const float* ps = k.pro_s + (size_t)b * k.pro_bstride;
        const float* pt = k.pro_t + (size_t)b * k.pro_bstride;
        for (int c = tid; c < k.Cin; c += THREADS) { ms = fmaxf(ms, fabsf(ps[c])); mt = fmaxf(mt, fabsf(pt[c])); }
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        a = fmaxf(a, __shfl_xor(a, o, 64));
        if (PRO != P2L_PRO_NONE) { ms = fmaxf(ms, __shfl_xor(ms, o, 64)); mt = fmaxf(mt, __shfl_xor(mt, o, 64)); }
      }
      if (lane == 0) { raw[wave * 4 + 0] = a; raw[wave * 4 + 1] = ms; raw[wave * 4 + 2] = mt; }
      __syncthreads();
      a = 0.f; ms = 0.f; mt = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { a = fmaxf(a, raw[w * 4]); ms = fmaxf(ms, raw[w * 4 + 1]); mt = fmaxf(mt, raw[w * 4 + 2]); }
      __syncthreads();                                   // (the patch is staged there next)
      if (PRO != P2L_PRO_NONE) a = (bound ? ms * a + mt : a) * 1.001f;
    } else {
      a = k.amax[b * 64 + lane];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
    }
    float inv_x, sw, inv_w;
    h2_scales(__builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a)), x_scale, inv_x);
    h2_scales(__builtin_amdgcn_readfirstlane(k.w_tail[0]), sw, inv_w);
    out_scale = inv_x * inv_w;
  }

#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) write_raw1(it);
  __syncthreads();
  if (nchunks > 1) load_raw(1);
#pragma unroll
  for (int part = 0; part < 4; ++part) {
    t_load(part, 0); t_rows(part, 0); t_load(part, 1); t_rows(part, 1);
    t_cols(part, 0, Vs);
    if (part & 1) { t_cols(part, 1, Vs); t_cols(part, 2, Vs); }
  }
  __syncthreads();
  using T_ = std::true_type; using F_ = std::false_type;
  if ((ABL & 1) && MT == 2) load_b(0, 1, 0);
  P2L_TR(1, 63);
  for (int c = 0; c + 1 < nchunks; ++c) {
    float* Vc = Vs + (c & 1) * V_FLOATS;
    float* Vn = Vs + ((c + 1) & 1) * V_FLOATS;
    P2L_TR(0, c);
    if constexpr (H2) {
      if (!(ABL & 32) || c == 0) { lda(Vc, 0); hstageH(); }
      P2L_SB();
      using B0_ = std::integral_constant<int, 0>;
      stepH(std::integral_constant<int, 0>{}, T_{}, B0_{}, Vc, Vn, c);
      stepH(std::integral_constant<int, 1>{}, T_{}, B0_{}, Vc, Vn, c);
      stepH(std::integral_constant<int, 2>{}, T_{}, B0_{}, Vc, Vn, c);
      stepH(std::integral_constant<int, 3>{}, T_{}, B0_{}, Vc, Vn, c);
    } else {
      if (!(ABL & 32) || c == 0) { lda(Vc, 0); hstage(); }
      P2L_SB();
      step(std::integral_constant<int, 0>{}, T_{}, Vc, Vn, c);
      step(std::integral_constant<int, 1>{}, T_{}, Vc, Vn, c);
      step(std::integral_constant<int, 2>{}, T_{}, Vc, Vn, c);
      step(std::integral_constant<int, 3>{}, T_{}, Vc, Vn, c);
    }
    if (!(ABL & 4)) __syncthreads();    // V(c+1) complete; every read of V(c) and of the patch done
    P2L_TR(5, c);
  }
  {
    const int c = nchunks - 1;
    float* Vc = Vs + (c & 1) * V_FLOATS;
    lda(Vc, 0);
    if constexpr (H2) {
      hstageH();
      P2L_SB();
      using B0_ = std::integral_constant<int, 0>;
      stepH(std::integral_constant<int, 0>{}, F_{}, B0_{}, Vc, Vc, c);
      stepH(std::integral_constant<int, 1>{}, F_{}, B0_{}, Vc, Vc, c);
      stepH(std::integral_constant<int, 2>{}, F_{}, B0_{}, Vc, Vc, c);
      stepH(std::integral_constant<int, 3>{}, F_{}, B0_{}, Vc, Vc, c);
    } else {
      hstage();
      P2L_SB();
      step(std::integral_constant<int, 0>{}, F_{}, Vc, Vc, c);
      step(std::integral_constant<int, 1>{}, F_{}, Vc, Vc, c);
      step(std::integral_constant<int, 2>{}, F_{}, Vc, Vc, c);
      step(std::integral_constant<int, 3>{}, F_{}, Vc, Vc, c);
    }
    __syncthreads();
  }
  P2L_TR(2, 63); P2L_TR(0, 62);
#undef P2L_TX
#undef P2L_PIN
#undef P2L_MF
#undef P2L_SB

  // ---- epilogue: 2 passes of 32 output channels; dump[f][tile 0..TILES-1][32 channels] -----
  float* dump = Vs;
  const int e_t = tid >> 3, e_c4 = tid & 7;             // item: (tile, 4 channels)
  const int ety = e_t >> 3, etx = e_t & 7;
  float* red = raw;                                      // [2 kinds][NW waves][32]
  const bool split = k.splitk > 1;
  const float alpha = (split ? 1.f : k.alpha) * out_scale;   // (the finish kernel scales the sum)
  float blk_amax = 0.f, blk_amaxp = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int fi = 0; fi < NF; ++fi)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tile = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          dump[((NF * wave + fi) * TILES + tile) * WN_DUMP_PITCH + l31] = acc[fi][m][j][r];
        }
    if (j == 0) { P2L_TR(1, 62); }
    __syncthreads();
    if (j == 0) { P2L_TR(2, 62); }
    const int nb = n0 + j * 32;
    EpiSums S;
    S.amax = blk_amax; S.amaxp = blk_amaxp;
    if (nb + e_c4 * 4 < k.n_store) {
      f32x4 T[2][4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {                  // A^T M, column jj
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(dump + ((0 + jj) * TILES + e_t) * WN_DUMP_PITCH + e_c4 * 4);
        const f32x4 m1 = *reinterpret_cast<const f32x4*>(dump + ((4 + jj) * TILES + e_t) * WN_DUMP_PITCH + e_c4 * 4);
        const f32x4 m2 = *reinterpret_cast<const f32x4*>(dump + ((8 + jj) * TILES + e_t) * WN_DUMP_PITCH + e_c4 * 4);
        const f32x4 m3 = *reinterpret_cast<const f32x4*>(dump + ((12 + jj) * TILES + e_t) * WN_DUMP_PITCH + e_c4 * 4);
        T[0][jj] = (m0 + m1) + m2;
        T[1][jj] = (m1 - m2) - m3;
      }
      f32x4 v[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) {                     // (A^T M) A
        v[2 * i + 0] = ((T[i][0] + T[i][1]) + T[i][2]) * alpha;
        v[2 * i + 1] = ((T[i][1] - T[i][2]) - T[i][3]) * alpha;
      }
      if (j == 0) { P2L_TR(3, 62); }
      if (split) {
        const size_t pix = ((size_t)b * k.H + y0 + 2 * ety) * k.W + x0 + 2 * etx;
        float* wp = k.ws + (((size_t)z * k.B * k.H * k.W + pix) * k.Cout + nb + e_c4 * 4);
        *reinterpret_cast<f32x4*>(wp) = v[0];
        *reinterpret_cast<f32x4*>(wp + k.Cout) = v[1];
        *reinterpret_cast<f32x4*>(wp + (size_t)k.W * k.Cout) = v[2];
        *reinterpret_cast<f32x4*>(wp + (size_t)(k.W + 1) * k.Cout) = v[3];
      } else if (ABL & 128) {                            // (lab) the result without the shared epilogue item: plain stores
        float* yp0 = k.y + (((size_t)b * k.H + y0 + 2 * ety) * k.W + x0 + 2 * etx) * k.y_ld + nb + e_c4 * 4;
        *reinterpret_cast<f32x4*>(yp0) = v[0];
        *reinterpret_cast<f32x4*>(yp0 + k.y_ld) = v[1];
        *reinterpret_cast<f32x4*>(yp0 + (size_t)k.W * k.y_ld) = v[2];
        *reinterpret_cast<f32x4*>(yp0 + (size_t)(k.W + 1) * k.y_ld) = v[3];
      } else if (ABL & 256) {                            // (lab) no stores at all
        if (v[0].x == 123.456f) k.y[0] = v[1].x + v[2].x + v[3].x;
      } else {
        epi_item(k, v, b, y0 + 2 * ety, x0 + 2 * etx, nb + e_c4 * 4, 0, 0, 0, S);
      }
    }
    if (j == 0) { P2L_TR(4, 62); }
    if (k.arb_x != nullptr && !split) {
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
        *reinterpret_cast<f32x4*>(red + NW * 32 + wave * 32 + lane * 4) = sg;
      }
      __syncthreads();
      if (tid < 32 * MT && nb + (tid & 31) < k.n_store) {
        // one partial sum per 8x16-pixel tile of the caller's buffer (4 waves each), whatever the block
        const int g = tid >> 5, col = tid & 31;
        const float* r0 = red + g * 128 + col;
        const float s0 = (r0[0] + r0[32]) + (r0[64] + r0[96]);
        const float s1 = (r0[NW * 32] + r0[NW * 32 + 32]) + (r0[NW * 32 + 64] + r0[NW * 32 + 96]);
        const size_t slot = (size_t)b * k.arb_nblk + (size_t)(MT * by + g) * k.tiles_x + bx;
        const size_t o = slot * k.Cout + nb + col;
        k.arb_partial[o] = s0;
        k.arb_partial[(size_t)k.B * k.arb_nblk * k.Cout + o] = s1;
      }
    }
    blk_amax = S.amax; blk_amaxp = S.amaxp;
    __syncthreads();
    P2L_TR(3 + j, 63);
  }
#undef P2L_TR
  // this block's partial maxima of what it stored (one per wave), for the launch that reads the tensor
  // next (P2LAmax)
  if ((k.amax_out != nullptr || k.amax_outp != nullptr) && !split) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      blk_amax = fmaxf(blk_amax, __shfl_xor(blk_amax, o, 64));
      blk_amaxp = fmaxf(blk_amaxp, __shfl_xor(blk_amaxp, o, 64));
    }
    if (lane == 0) {
      // 8 partials per 16x16-pixel tile and 64 channels in both block shapes (the reader must not see
      // which one ran): an 8x16 block is the upper / lower half (by & 1) of its 16x16 tile
      const size_t t16 = MT == 2 ? (size_t)tile_in_image : (size_t)(by >> 1) * k.tiles_x + bx;
      const size_t slot = (size_t)b * k.amax_out_n + (t16 * k.n_ntiles + (n0 >> 6)) * 8 + (MT == 2 ? 0 : (by & 1) * 4) + wave;
      if (k.amax_out != nullptr) k.amax_out[slot] = blk_amax;
      if (k.amax_outp != nullptr) k.amax_outp[slot] = blk_amaxp;
    }
  }
}

// ---- weights: U = G g G^T per (cout, cin), split into 3 bf16 pieces, fragment order --------
__global__ void wino_pack_kernel(const float* w, float* dst, int O, int I, int N_pad, int K_pad,
                                 int transpose_flip) {
  // one thread per (chunk cc, frequency f, 32-channel tile jn, lane): 8 input channels of one
  // output channel -> three 16-byte pieces
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int n_t32 = N_pad >> 5;
  const size_t total = (size_t)(K_pad >> 4) * 16 * n_t32 * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  size_t q = idx >> 6;
  const int jn = (int)(q % n_t32); q /= n_t32;
  const int f = (int)(q & 15);
  const int cc = (int)(q >> 4);
  const int n = jn * 32 + (lane & 31);
  const int fi = f >> 2, fj = f & 3;
  const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  const int N = transpose_flip ? I : O, K = transpose_flip ? O : I;
  bf16x8 p1, p2, p3;
  for (int e = 0; e < 8; ++e) {
    const int kk = cc * 16 + (lane >> 5) * 8 + e;
    double u = 0.0;
    if (n < N && kk < K) {
      for (int a = 0; a < 3; ++a)
        for (int bb = 0; bb < 3; ++bb) {
          // conv weight [n][kk][a][bb]; input-gradient form: w[kk][n] mirrored
          const float g = transpose_flip ? w[(((size_t)kk * I + n) * 3 + (2 - a)) * 3 + (2 - bb)]
                                         : w[(((size_t)n * I + kk) * 3 + a) * 3 + bb];
          u += G[fi][a] * (double)g * G[fj][bb];
        }
    }
    const float x = (float)u;
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    p1[e] = h; p2[e] = m; p3[e] = (__bf16)r2;
  }
  bf16x8* out = reinterpret_cast<bf16x8*>(dst) +
                ((((size_t)cc * 16 + f) * n_t32 + jn) * 3) * 64 + lane;
  out[0] = p1; out[64] = p2; out[128] = p3;
}

// ---- fp16 x 2 image of the same U: scaled by the layer's power of two, two fp16 pieces -------
// tail[0] = bits of max |w| over the layer (|U| <= 2.25 max |w|: the scale leaves that headroom)
__global__ void wino_wmax_kernel(const float* w, size_t n, unsigned* tail) {
  float mx = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    mx = fmaxf(mx, fabsf(w[i]));
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(tail, __builtin_bit_cast(unsigned, mx));   // (>= 0: bit order = value order)
}
__global__ void wino_pack_h2_kernel(const float* w, float* dst, int O, int I, int N_pad, int K_pad,
                                    int transpose_flip, const unsigned* tail) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int n_t32 = N_pad >> 5;
  const size_t total = (size_t)(K_pad >> 4) * 16 * n_t32 * 64;
  if (idx >= total) return;
  float scale, inv;
  h2_scales(tail[0], scale, inv);
  const int lane = (int)(idx & 63);
  size_t q = idx >> 6;
  const int jn = (int)(q % n_t32); q /= n_t32;
  const int f = (int)(q & 15);
  const int cc = (int)(q >> 4);
  const int n = jn * 32 + (lane & 31);
  const int fi = f >> 2, fj = f & 3;
  const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  const int N = transpose_flip ? I : O, K = transpose_flip ? O : I;
  h16x8 p1, p2;
  for (int e = 0; e < 8; ++e) {
    const int kk = cc * 16 + (lane >> 5) * 8 + e;
    double u = 0.0;
    if (n < N && kk < K) {
      for (int a = 0; a < 3; ++a)
        for (int bb = 0; bb < 3; ++bb) {
          const float g = transpose_flip ? w[(((size_t)kk * I + n) * 3 + (2 - a)) * 3 + (2 - bb)]
                                         : w[(((size_t)n * I + kk) * 3 + a) * 3 + bb];
          u += G[fi][a] * (double)g * G[fj][bb];
        }
    }
    const float x = (float)u * scale;                  // (rounded to fp32 first, like the bf16 x 3 image)
    const _Float16 h = (_Float16)x;
    p1[e] = h; p2[e] = (_Float16)(x - (float)h);
  }
  h16x8* out = reinterpret_cast<h16x8*>(dst) + ((((size_t)cc * 16 + f) * n_t32 + jn) * 2) * 64 + lane;
  out[0] = p1; out[64] = p2;
}

// ---- fp16 x 2: max |x| of every image, 64 partial maxima each (the prologue applied) ---------
template <int PRO>
__global__ __launch_bounds__(256) void wino_amax_kernel(const ConvK k) {
  const int b = blockIdx.y;
  const int q4 = k.Cin >> 2;
  const unsigned total = (unsigned)k.H * k.W * q4;
  const float* xi = k.x + (size_t)b * k.H * k.W * k.x_ld;
  float mx = 0.f;
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += 64 * 256) {
    const unsigned pix = i / q4, c4 = i - pix * q4;
    f32x4 v = *reinterpret_cast<const f32x4*>(xi + (size_t)pix * k.x_ld + c4 * 4);
    if (PRO != P2L_PRO_NONE) {
      const f32x4 sr = *reinterpret_cast<const f32x4*>(k.pro_s + (size_t)b * k.pro_bstride + c4 * 4);
      const f32x4 tr = *reinterpret_cast<const f32x4*>(k.pro_t + (size_t)b * k.pro_bstride + c4 * 4);
      v = v * sr + tr;                                 // (|.| below covers the ReLU form too)
      if (PRO == P2L_PRO_AFFINE_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
    }
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) k.amax[b * 64 + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

}  // namespace

// floats of the Winograd image for an N_pad x K_pad 3x3 conv (16 frequencies x 6 bytes)
extern "C" size_t p2l_wino_weight_floats(int N_pad, int K_pad) {
  return (size_t)N_pad * K_pad * 24;
}
// ... and of the fp16 x 2 image behind it (16 frequencies x 4 bytes, then 4 tail floats)
extern "C" size_t p2l_wino_h2_weight_floats(int N_pad, int K_pad) {
  return (size_t)N_pad * K_pad * 16 + 4;
}

// Small-grid layers (16 ... 63 blocks of 8x16x64 per image, H and W multiples of 16): the 16x16
// kernel with the input channels cut into this many slices -- a function of the layer shape only,
// never of the batch (1 = not such a layer).  At least 8 chunks per slice.  Measured with 18
// candidates (tools/micro/conv_lab.cpp): 32^2 256->256 and 16^2 512->512 run 1.5-1.7x faster than
// on the direct kernel; without the slices the same layers would be 24-48 blocks at 2-3 candidates
// per GPU.
extern "C" int p2l_wino_split_factor(int H, int W, int Cin, int Cout) {
#ifdef P2L_NO_WINO_SLICES
  return 1;                                            // (A/B build: tools/ab_build.sh)
#endif
  if (H % 16 || W % 16 || Cout % 64 || Cin % 16) return 1;
  const int per_image = (H / 8) * (W / 16) * (Cout / 64);
  if (per_image >= 64 || per_image < 16) return 1;
  int s = 64 / per_image;                              // 2 or 4
  const int nchunks = Cin / 16;
  while (s > 1 && nchunks / s < 8) s >>= 1;
  return s;
}

// shapes the Winograd kernel takes (the launcher adds the per-launch conditions)
extern "C" int p2l_wino_weight_ok(int N_pad, int K_pad) {
  return N_pad % 64 == 0 && K_pad % 16 == 0;
}

int p2l_wino_pack(const float* w_oihw, int O, int I, int N_pad, int K_pad, int transpose_flip,
                  float* dst, hipStream_t st) {
  const size_t total = (size_t)(K_pad >> 4) * 16 * (N_pad >> 5) * 64;
  hipLaunchKernelGGL(wino_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w_oihw, dst, O,
                     I, N_pad, K_pad, transpose_flip);
  // the fp16 x 2 image behind it
  float* h2 = dst + p2l_wino_weight_floats(N_pad, K_pad);
  unsigned* tail = reinterpret_cast<unsigned*>(h2 + (size_t)N_pad * K_pad * 16);
  if (hipMemsetAsync(tail, 0, 16, st) != hipSuccess) return P2L_ELAUNCH;
  const size_t nw = (size_t)O * I * 9;
  hipLaunchKernelGGL(wino_wmax_kernel, dim3((unsigned)(cdiv(nw, 256) < 256 ? cdiv(nw, 256) : 256)), dim3(256),
                     0, st, w_oihw, nw, tail);
  hipLaunchKernelGGL(wino_pack_h2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w_oihw, h2, O,
                     I, N_pad, K_pad, transpose_flip, tail);
  return p2l_check_launch();
}

int p2l_amax_launch(const ConvK& k, int pro, hipStream_t st) {
  if (pro == P2L_PRO_NONE) hipLaunchKernelGGL(wino_amax_kernel<P2L_PRO_NONE>, dim3(64, k.B), dim3(256), 0, st, k);
  else if (pro == P2L_PRO_AFFINE_RELU) hipLaunchKernelGGL(wino_amax_kernel<P2L_PRO_AFFINE_RELU>, dim3(64, k.B), dim3(256), 0, st, k);
  else hipLaunchKernelGGL(wino_amax_kernel<P2L_PRO_AFFINE>, dim3(64, k.B), dim3(256), 0, st, k);
  return p2l_check_launch();
}

// k arrives with the 8x16-pixel tiling (tiles_x = W/16, tiles_y = H/8, n_mtiles, n_ntiles =
// Cout/64).  16x16-pixel blocks whenever H and W allow it (measured in the bench step at 18, 9,
// 5, 3, 2 candidates per GPU: tools/policy_probe.py), the 8x16-pixel kernel otherwise or on
// P2L_FORM_WINO_8X16; both give bit-identical results.
#ifdef P2L_LAB
// lab build (tools/micro/build_lab.sh): timing ablations and the phase trace of one block
static int g_lab_abl = 0;
static void* g_lab_trace = nullptr;
extern "C" int p2l_lab_set(int abl, void* trace) { g_lab_abl = abl; g_lab_trace = trace; return P2L_OK; }
#endif

int p2l_wino_launch(const ConvK& k_in, int pro, hipStream_t st) {
  ConvK k = k_in;
  if (k.H % 16 == 0 && k.W % 16 == 0 && !(k.form & P2L_FORM_WINO_8X16)) {
    k.tiles_y = k.H / 16;
    k.n_mtiles = k.B * k.tiles_x * k.tiles_y;
    if (k.splitk <= 1) { k.ws = nullptr; k.splitk = 1; k.chunks_per_split = k.nchunks; }
    dim3 grid(k.n_mtiles * k.n_ntiles, k.splitk), block(W16_THREADS);
#ifdef P2L_LAB
    if (k.splitk <= 1) k.ws = (float*)g_lab_trace;
#define P2L_W16L(ABL)                                                                        \
  case ABL: {                                                                                \
    (void)hipFuncSetAttribute((const void*)wino16s_conv_kernel<P2L_PRO_NONE, ABL>,           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);       \
    hipLaunchKernelGGL((wino16s_conv_kernel<P2L_PRO_NONE, ABL>), grid, block, W16_LDS_BYTES, st, k); \
    return p2l_check_launch();                                                               \
  }
    if (pro == P2L_PRO_NONE && k.amax != nullptr && (k.form & P2L_FORM_WINO_H2_8X16)) {
      // the 4-wave block (MT = 1) under the same ablations
      ConvK k8 = k;
      k8.tiles_y = k.H / 8;
      k8.n_mtiles = k.B * k.tiles_x * k8.tiles_y;
      dim3 grid8(k8.n_mtiles * k8.n_ntiles, k8.splitk), block8(256);
#define P2L_W8LH(ABL)                                                                        \
  case ABL: {                                                                                \
    (void)hipFuncSetAttribute((const void*)wino16s_conv_kernel<P2L_PRO_NONE, ABL, true, 1>,  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);       \
    hipLaunchKernelGGL(wino_amax_kernel<P2L_PRO_NONE>, dim3(64, k.B), dim3(256), 0, st, k8); \
    hipLaunchKernelGGL((wino16s_conv_kernel<P2L_PRO_NONE, ABL, true, 1>), grid8, block8, W8_LDS_BYTES, st, k8); \
    return p2l_check_launch();                                                               \
  }
      switch (g_lab_abl) {
        P2L_W8LH(0) P2L_W8LH(1) P2L_W8LH(2) P2L_W8LH(4) P2L_W8LH(8) P2L_W8LH(16) P2L_W8LH(64) P2L_W8LH(3) P2L_W8LH(67) P2L_W8LH(79)
        default: return P2L_EINVAL;
      }
#undef P2L_W8LH
    }
    if (pro == P2L_PRO_NONE && k.amax != nullptr) {
#define P2L_W16LH(ABL)                                                                       \
  case ABL: {                                                                                \
    (void)hipFuncSetAttribute((const void*)wino16s_conv_kernel<P2L_PRO_NONE, ABL, true>,     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);       \
    hipLaunchKernelGGL(wino_amax_kernel<P2L_PRO_NONE>, dim3(64, k.B), dim3(256), 0, st, k);  \
    hipLaunchKernelGGL((wino16s_conv_kernel<P2L_PRO_NONE, ABL, true>), grid, block, W16_LDS_BYTES, st, k); \
    return p2l_check_launch();                                                               \
  }
      switch (g_lab_abl) {
        P2L_W16LH(0) P2L_W16LH(1) P2L_W16LH(2) P2L_W16LH(4) P2L_W16LH(8) P2L_W16LH(16) P2L_W16LH(64)
        P2L_W16LH(3) P2L_W16LH(10) P2L_W16LH(67) P2L_W16LH(75) P2L_W16LH(79) P2L_W16LH(111) P2L_W16LH(128) P2L_W16LH(256)
        default: return P2L_EINVAL;
      }
#undef P2L_W16LH
    }
    if (pro == P2L_PRO_NONE && k.amax == nullptr) {
      switch (g_lab_abl) {
        P2L_W16L(0) P2L_W16L(1) P2L_W16L(2) P2L_W16L(4) P2L_W16L(8) P2L_W16L(16) P2L_W16L(64)
        P2L_W16L(3) P2L_W16L(10) P2L_W16L(67) P2L_W16L(75) P2L_W16L(79) P2L_W16L(111)
        default: return P2L_EINVAL;
      }
    }
#undef P2L_W16L
#endif
#define P2L_W16S(PRO)                                                                        \
  do {                                                                                       \
    static std::atomic<bool> attr_set{false};                                                \
    if (!attr_set) {                                                                         \
      (void)hipFuncSetAttribute((const void*)wino16s_conv_kernel<PRO>,                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
      attr_set = true;                                                                       \
    }                                                                                        \
    hipLaunchKernelGGL(wino16s_conv_kernel<PRO>, grid, block, W16_LDS_BYTES, st, k);         \
  } while (0)
#define P2L_W16H(PRO, MTV, LDSB)                                                             \
  do {                                                                                       \
    static std::atomic<bool> attr_set{false};                                                \
    if (!attr_set) {                                                                         \
      (void)hipFuncSetAttribute((const void*)wino16s_conv_kernel<PRO, 0, true, MTV>,         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
      attr_set = true;                                                                       \
    }                                                                                        \
    if (k.amax_in == nullptr)                                                                \
      hipLaunchKernelGGL(wino_amax_kernel<PRO>, dim3(64, k.B), dim3(256), 0, st, k);         \
    hipLaunchKernelGGL((wino16s_conv_kernel<PRO, 0, true, MTV>), grid, block, LDSB, st, k);  \
  } while (0)
    if (k.amax != nullptr) {                           // fp16 x 2 arithmetic (conv_launch_impl decides)
      // Block shape from the GRID (round 5): while the 16x16 blocks of the launch leave half the CUs
      // without one, 8x16-pixel blocks of 4 waves -- twice as many, one wave per SIMD, about half as long.
      // Bit-identical results and maxima slots (MT in the kernel), so the batch may decide.
      const long blocks16 = (long)k.n_mtiles * k.n_ntiles * k.splitk;
      const bool half = (k.form & P2L_FORM_WINO_H2_8X16) ||
                        (!(k.form & P2L_FORM_WINO_H2_16X16) && blocks16 <= 128);
      if (half) {
        k.tiles_y = k.H / 8;
        k.n_mtiles = k.B * k.tiles_x * k.tiles_y;
        grid = dim3(k.n_mtiles * k.n_ntiles, k.splitk);
        block = dim3(256);
        if (pro == P2L_PRO_NONE) P2L_W16H(P2L_PRO_NONE, 1, W8_LDS_BYTES);
        else if (pro == P2L_PRO_AFFINE_RELU) P2L_W16H(P2L_PRO_AFFINE_RELU, 1, W8_LDS_BYTES);
        else P2L_W16H(P2L_PRO_AFFINE, 1, W8_LDS_BYTES);
        return p2l_check_launch();
      }
      if (pro == P2L_PRO_NONE) P2L_W16H(P2L_PRO_NONE, 2, W16_LDS_BYTES);
      else if (pro == P2L_PRO_AFFINE_RELU) P2L_W16H(P2L_PRO_AFFINE_RELU, 2, W16_LDS_BYTES);
      else P2L_W16H(P2L_PRO_AFFINE, 2, W16_LDS_BYTES);
      return p2l_check_launch();
    }
#undef P2L_W16H
    if (pro == P2L_PRO_NONE) P2L_W16S(P2L_PRO_NONE);
    else if (pro == P2L_PRO_AFFINE_RELU) P2L_W16S(P2L_PRO_AFFINE_RELU);
    else P2L_W16S(P2L_PRO_AFFINE);
#undef P2L_W16S
    return p2l_check_launch();
  }
  dim3 grid(k.n_mtiles * k.n_ntiles), block(WN_THREADS);
#define P2L_WN(PRO)                                                                          \
  do {                                                                                       \
    static std::atomic<bool> attr_set{false};                                                \
    if (!attr_set) {                                                                         \
      (void)hipFuncSetAttribute((const void*)wino_conv_kernel<PRO>,                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
      attr_set = true;                                                                       \
    }                                                                                        \
    hipLaunchKernelGGL(wino_conv_kernel<PRO>, grid, block, WN_LDS_BYTES, st, k);             \
  } while (0)
  if (pro == P2L_PRO_NONE) P2L_WN(P2L_PRO_NONE);
  else if (pro == P2L_PRO_AFFINE_RELU) P2L_WN(P2L_PRO_AFFINE_RELU);
  else P2L_WN(P2L_PRO_AFFINE);
#undef P2L_WN
  return p2l_check_launch();
}
