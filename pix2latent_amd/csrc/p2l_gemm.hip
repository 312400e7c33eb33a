// Batched fp32 GEMM on v_mfma_f32_32x32x2_f32 for the SelfAttn bmm's of
// BigGAN-deep (theta^T phi, attn * g) and their five gradient products.
// [3P-recall of pytorch_pretrained_biggan SelfAttn.forward, reached from
//  pix2latent/model/biggan.py:58]
//
// Block tile 128 x BN (64|32), K chunk 16, 4 waves each owning a 32 x BN strip.
// Operands may be K-contiguous ([rows][K], fragment = one ds_read_b128 per 4
// MFMAs, rows padded to 20 floats) or K-major ([K][rows], fragment = 4
// conflict-free ds_read_b32), so no operand ever needs a transposed copy in HBM.
#include "p2l_common.h"

namespace {

struct GemmK {
  const float* A;
  const float* B;
  float* C;
  int M, N, K, lda, ldb, ldc;
  long long sa, sb, sc;
  float alpha;
  int accumulate;
  int n_ntiles;
  // split-K (deep-K products with few output tiles): slice z = blockIdx.z multiplies
  // K range [z*kper, (z+1)*kper) and dumps its raw accumulators to ws[z][batch][M][N];
  // gemm_splitk_finish adds the slices in a fixed order
  float* ws;
  int ksplit, kper;
};

template <int BN, bool AKM, bool BKM>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(const GemmK g) {
  constexpr int KC = 16;
  constexpr int NT = BN / 32;
  constexpr int PA = AKM ? 132 : 20;  // floats per LDS row of A
  constexpr int PB = BKM ? (BN + 4) : 20;
  constexpr int A_FLOATS = AKM ? KC * PA : 128 * PA;
  constexpr int B_FLOATS = BKM ? KC * PB : BN * PB;
  __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + B_FLOATS];
  float* As = smem;
  float* Bs = smem + A_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int swz = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = swz / g.n_ntiles, nt = swz - mt * g.n_ntiles;
  const int m0 = mt * 128, n0 = nt * BN;
  const int batch = blockIdx.y;
  const float* A = g.A + (size_t)batch * g.sa;
  const float* B = g.B + (size_t)batch * g.sb;
  float* C = g.C + (size_t)batch * g.sc;

  // staging: A tile = 128x16 floats = 512 float4 (2/thread); B = BN x 16.
  constexpr int A_ITERS = 2;
  constexpr int B_ITEMS = BN * 4;
  constexpr int B_ITERS = (B_ITEMS + 255) / 256;
  f32x4 ar[A_ITERS], br[B_ITERS];

  auto load_regs = [&](int kc) {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const int j = tid + 256 * it;
      if (!AKM) {
        const int r = j >> 2, v = j & 3;
        ar[it] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + r) * g.lda + kc + v * 4);
      } else {
        const int r = j >> 5, v = j & 31;  // r = k row, v = float4 along M
        ar[it] = *reinterpret_cast<const f32x4*>(A + (size_t)(kc + r) * g.lda + m0 + v * 4);
      }
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int j = tid + 256 * it;
      if (j < B_ITEMS) {
        if (!BKM) {
          const int r = j >> 2, v = j & 3;
          br[it] = *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + r) * g.ldb + kc + v * 4);
        } else {
          const int r = j / (BN / 4), v = j - r * (BN / 4);
          br[it] = *reinterpret_cast<const f32x4*>(B + (size_t)(kc + r) * g.ldb + n0 + v * 4);
        }
      }
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const int j = tid + 256 * it;
      if (!AKM) {
        const int r = j >> 2, v = j & 3;
        *reinterpret_cast<f32x4*>(As + r * PA + v * 4) = ar[it];
      } else {
        const int r = j >> 5, v = j & 31;
        *reinterpret_cast<f32x4*>(As + r * PA + v * 4) = ar[it];
      }
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int j = tid + 256 * it;
      if (j < B_ITEMS) {
        if (!BKM) {
          const int r = j >> 2, v = j & 3;
          *reinterpret_cast<f32x4*>(Bs + r * PB + v * 4) = br[it];
        } else {
          const int r = j / (BN / 4), v = j - r * (BN / 4);
          *reinterpret_cast<f32x4*>(Bs + r * PB + v * 4) = br[it];
        }
      }
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int k_begin = (int)blockIdx.z * g.kper;
  const int k_end = min(g.K, k_begin + g.kper);
  load_regs(k_begin);
  write_lds();
  __syncthreads();

  for (int kc = k_begin; kc < k_end; kc += KC) {
    const bool more = kc + KC < k_end;
    if (more) load_regs(kc + KC);
#pragma unroll
    for (int kk = 0; kk < KC / 8; ++kk) {
      f32x4 a;
      if (!AKM) {
        a = *reinterpret_cast<const f32x4*>(As + (wave * 32 + l31) * PA + kk * 8 + lhi * 4);
      } else {
        const float* p = As + (kk * 8 + lhi * 4) * PA + wave * 32 + l31;
        a = f32x4{p[0], p[PA], p[2 * PA], p[3 * PA]};
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4 b;
        if (!BKM) {
          b = *reinterpret_cast<const f32x4*>(Bs + (j * 32 + l31) * PB + kk * 8 + lhi * 4);
        } else {
          const float* p = Bs + (kk * 8 + lhi * 4) * PB + j * 32 + l31;
          b = f32x4{p[0], p[PB], p[2 * PB], p[3 * PB]};
        }
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) write_lds();
    __syncthreads();
  }

  if (g.ksplit > 1) {
    float* S = g.ws + ((size_t)blockIdx.z * gridDim.y + batch) * ((size_t)g.M * g.N);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        S[(size_t)m * g.N + n] = acc[j][r];
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      float* cp = C + (size_t)m * g.ldc + n;
      float v = g.alpha * acc[j][r];
      if (g.accumulate) v += *cp;
      *cp = v;
    }
  }
}

// C = alpha * (ws[0] + ws[1] + ... in this order) (+ C): one float4 of C per thread
__global__ __launch_bounds__(256) void gemm_splitk_finish(const GemmK g, int batch) {
  const size_t mn4 = (size_t)g.M * g.N / 4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= mn4 * batch) return;
  const size_t b = i / mn4, r = i - b * mn4;
  const size_t m = r / (g.N / 4), n = (r - m * (g.N / 4)) * 4;
  const size_t slab = (size_t)batch * g.M * g.N;
  const float* p = g.ws + b * ((size_t)g.M * g.N) + m * g.N + n;
  f32x4 v = *reinterpret_cast<const f32x4*>(p);
  for (int z = 1; z < g.ksplit; ++z) v += *reinterpret_cast<const f32x4*>(p + z * slab);
  float* cp = g.C + b * g.sc + m * g.ldc + n;
  v = v * g.alpha;
  if (g.accumulate) { cp[0] += v.x; cp[1] += v.y; cp[2] += v.z; cp[3] += v.w; }
  else { cp[0] = v.x; cp[1] = v.y; cp[2] = v.z; cp[3] = v.w; }
}

template <int BN>
int launch_gemm(const GemmK& g, int akm, int bkm, int batch, hipStream_t st) {
  dim3 grid((g.M / 128) * g.n_ntiles, batch, g.ksplit), block(256);
  if (!akm && !bkm) hipLaunchKernelGGL((gemm_mfma_kernel<BN, false, false>), grid, block, 0, st, g);
  else if (!akm && bkm) hipLaunchKernelGGL((gemm_mfma_kernel<BN, false, true>), grid, block, 0, st, g);
  else if (akm && !bkm) hipLaunchKernelGGL((gemm_mfma_kernel<BN, true, false>), grid, block, 0, st, g);
  else hipLaunchKernelGGL((gemm_mfma_kernel<BN, true, true>), grid, block, 0, st, g);
  return p2l_check_launch();
}

}  // namespace

// split-K only pays for deep-K products that leave most CUs without a tile
static int gemm_suggest_split(const P2LGemm* d) {
  if (!d || d->M % 128 || d->N % 32 || d->K < 1024) return 1;
  const int bn = (d->N % 64 == 0) ? 64 : 32;
  // (counted for a reference batch of 4, NOT d->batch: the slice count -- hence the fp32 summation order
  //  of an image's product -- must not depend on how many images share the launch; round 5)
  const int blocks = (d->M / 128) * (d->N / bn) * 4;
  if (blocks >= 192) return 1;
  int s = cdiv(512, blocks);
  if (s > d->K / 256) s = d->K / 256;      // >= 16 chunks per slice
  if (s > 8) s = 8;
  return s < 1 ? 1 : s;
}

extern "C" size_t p2l_gemm_ws_bytes(const P2LGemm* d) {
  const int s = gemm_suggest_split(d);
  return s > 1 ? (size_t)s * d->batch * d->M * d->N * sizeof(float) : 0;
}

static int gemm_impl(const P2LGemm* d, const float* A, const float* B, float* C, void* ws,
                     size_t ws_bytes, void* stream);

extern "C" int p2l_gemm(const P2LGemm* d, const float* A, const float* B,
                        float* C, void* stream) {
  return gemm_impl(d, A, B, C, nullptr, 0, stream);
}

extern "C" int p2l_gemm_ws(const P2LGemm* d, const float* A, const float* B, float* C,
                           void* ws, size_t ws_bytes, void* stream) {
  return gemm_impl(d, A, B, C, ws, ws_bytes, stream);
}

static int gemm_impl(const P2LGemm* d, const float* A, const float* B, float* C, void* ws,
                     size_t ws_bytes, void* stream) {
  if (!d || !A || !B || !C) return P2L_EINVAL;
  if (d->M % 128 || d->N % 32 || d->K % 16 || d->batch < 1) return P2L_EINVAL;
  if (d->lda % 4 || d->ldb % 4) return P2L_EINVAL;
  GemmK g{};
  g.A = A; g.B = B; g.C = C;
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc;
  g.sa = d->stride_a; g.sb = d->stride_b; g.sc = d->stride_c;
  g.alpha = d->alpha; g.accumulate = d->accumulate;
  const int bn = (d->N % 64 == 0) ? 64 : 32;
  g.n_ntiles = d->N / bn;
  hipStream_t st = (hipStream_t)stream;
  g.ksplit = 1; g.kper = d->K; g.ws = nullptr;
  const int s = gemm_suggest_split(d);
  if (s > 1 && ws && ws_bytes >= p2l_gemm_ws_bytes(d) && d->ldc % 4 == 0) {
    g.ksplit = s;
    g.kper = cdiv(cdiv(d->K, s), 16) * 16;
    g.ksplit = cdiv(d->K, g.kper);
    g.ws = (float*)ws;
  }
  const int rc = bn == 64 ? launch_gemm<64>(g, d->a_kmajor, d->b_kmajor, d->batch, st)
                          : launch_gemm<32>(g, d->a_kmajor, d->b_kmajor, d->batch, st);
  if (rc || g.ksplit == 1) return rc;
  const size_t items = (size_t)d->batch * d->M * d->N / 4;
  hipLaunchKernelGGL(gemm_splitk_finish, dim3((unsigned)cdiv(items, (size_t)256)), dim3(256), 0, st,
                     g, d->batch);
  return p2l_check_launch();
}
