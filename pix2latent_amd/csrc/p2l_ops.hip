// HBM-bound pieces of the pix2latent hot path: conditioning linears, CBN fold,
// activation backward + per-(sample,channel) reductions, softmax, pooling
// backward, image layout helpers, the L1 / LPIPS loss tails and fused Adam.
// Every reduction is a fixed-order tree (wave shuffles -> LDS -> second stage):
// CMA-ES ranks candidates by these numbers, so no float atomics anywhere.
#include "p2l_common.h"

namespace {

// ---------------------------------------------------------------------------
// block-level fixed-order sum (256 threads), result valid in every thread
// ---------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red /*>=4*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------
// linear fwd: y[b][n] = sum_k x[b][k] W[k][n] + bias[n]
// block = 64 output columns x 4 K-groups; x staged in LDS (broadcast reads)
// ---------------------------------------------------------------------------
template <int BG>
__global__ __launch_bounds__(256) void linear_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ W,
    const float* __restrict__ bias, float* __restrict__ y, int Bn, int b_begin,
    int K, int N, int x_ld) {
  extern __shared__ float sm[];  // [BG][K] x slab, then [4][BG][64] reduce
  float* xs = sm;
  float* red = sm + BG * K;
  const int tid = threadIdx.x, nl = tid & 63, kg = tid >> 6;
  const int n = blockIdx.x * 64 + nl;
  const int nb = min(BG, Bn - b_begin);
  for (int i = tid; i < BG * K; i += 256) {
    const int b = i / K, kk = i - b * K;
    xs[i] = (b < nb) ? x[(size_t)(b_begin + b) * x_ld + kk] : 0.f;
  }
  __syncthreads();
  float acc[BG];
#pragma unroll
  for (int b = 0; b < BG; ++b) acc[b] = 0.f;
  const int kper = K >> 2;
  const int k0 = kg * kper;
  if (n < N) {
#pragma unroll 8
    for (int kk = k0; kk < k0 + kper; ++kk) {
      const float w = W[(size_t)kk * N + n];
#pragma unroll
      for (int b = 0; b < BG; ++b) acc[b] = fmaf(xs[b * K + kk], w, acc[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < BG; ++b) red[(kg * BG + b) * 64 + nl] = acc[b];
  __syncthreads();
  // 256 threads finish BG*64 outputs in fixed k-group order
  for (int i = tid; i < BG * 64; i += 256) {
    const int b = i >> 6, c = i & 63;
    const int nn = blockIdx.x * 64 + c;
    if (b < nb && nn < N) {
      float v = (red[(0 * BG + b) * 64 + c] + red[(1 * BG + b) * 64 + c]) +
                (red[(2 * BG + b) * 64 + c] + red[(3 * BG + b) * 64 + c]);
      if (bias) v += bias[nn];
      y[(size_t)(b_begin + b) * N + nn] = v;
    }
  }
}

// linear bwd: dx[b][k] = sum_n dy[b][n] W[k][n]; one 1024-thread block per k
// (16 waves per CU keep enough loads in flight to stream W and the L2-resident
// dy rows), float4 loads, fixed-order wave -> LDS reduction.
template <int BG>
__global__ __launch_bounds__(1024) void linear_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ W,
    float* __restrict__ dx, int Bn, int b_begin, int K, int N, int accumulate, int dx_ld) {
  __shared__ float red[BG][16];
  const int k = blockIdx.x, tid = threadIdx.x;
  const int nb = min(BG, Bn - b_begin);
  float acc[BG];
#pragma unroll
  for (int b = 0; b < BG; ++b) acc[b] = 0.f;
  const float* wrow = W + (size_t)k * N;
  const int N4 = N >> 2;
  for (int i = tid; i < N4; i += 1024) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(wrow + 4 * i);
#pragma unroll
    for (int b = 0; b < BG; ++b) {
      if (b < nb) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + (size_t)(b_begin + b) * N + 4 * i);
        acc[b] += (d.x * w.x + d.y * w.y) + (d.z * w.z + d.w * w.w);
      }
    }
  }
  for (int n = 4 * N4 + tid; n < N; n += 1024) {   // tail when N % 4 != 0
    const float w = wrow[n];
#pragma unroll
    for (int b = 0; b < BG; ++b)
      if (b < nb) acc[b] = fmaf(dy[(size_t)(b_begin + b) * N + n], w, acc[b]);
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int b = 0; b < BG; ++b) {
    const float v = wave_sum(acc[b]);
    if (lane == 0) red[b][wave] = v;
  }
  __syncthreads();
  if (tid < nb) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[tid][w];
    float* p = dx + (size_t)(b_begin + tid) * dx_ld + k;
    *p = accumulate ? (*p + s) : s;
  }
}

__global__ void cbn_fold_fwd_kernel(const float* g_raw, const float* b_raw,
                                    const float* mean, const float* rstd,
                                    float* s, float* t, int Bn, int C, int raw_ld) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Bn * C) return;
  const int b = i / C, c = i - b * C;
  const float gain = 1.f + g_raw[(size_t)b * raw_ld + c];
  const float bias = b_raw[(size_t)b * raw_ld + c];
  const float sv = gain * rstd[c];
  s[i] = sv;
  t[i] = bias - mean[c] * sv;
}

__global__ void cbn_fold_bwd_kernel(const float* ds, const float* dt,
                                    const float* mean, const float* rstd,
                                    float* dg_raw, float* db_raw, int Bn, int C,
                                    int raw_ld) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Bn * C) return;
  const int b = i / C, c = i - b * C;
  // s = (1+g)*rstd ; t = bias - mean*s  =>  dL/dg = (ds - dt*mean)*rstd ; dL/dbias = dt
  dg_raw[(size_t)b * raw_ld + c] = (ds[i] - dt[i] * mean[c]) * rstd[c];
  db_raw[(size_t)b * raw_ld + c] = dt[i];
}

// ---------------------------------------------------------------------------
// backward of a = max(x*s+t, 0):  g = mask*da ; dx = g*s + skip ; partial sums
// grid (slab, C/64, B); block = 16 channel-float4 lanes x 16 pixel lanes
// ---------------------------------------------------------------------------
struct ArbK {
  const float* da; const float* x; const float* s; const float* t;
  const float* skip; float* dx; float* partial;
  int da_ld, x_ld, dx_ld, skip_ld, skip_C, skip_ups, st_bstride;
  int Bn, P, C, H, W, nblk, nomask;
};
constexpr int ARB_SLAB = 256;

__global__ __launch_bounds__(256) void affine_relu_bwd_kernel(const ArbK k) {
  __shared__ f32x4 red_s[256], red_t[256];
  const int tid = threadIdx.x, cl = tid & 15, pl = tid >> 4;
  const int slab = blockIdx.x, c = blockIdx.y * 64 + cl * 4, b = blockIdx.z;
  const bool live = c < k.C;          // C % 64 == 32: the upper half of the last strip idles
  f32x4 s4 = {0, 0, 0, 0}, t4 = {0, 0, 0, 0};
  if (live) {
    s4 = *reinterpret_cast<const f32x4*>(k.s + (size_t)b * k.st_bstride + c);
    t4 = *reinterpret_cast<const f32x4*>(k.t + (size_t)b * k.st_bstride + c);
  }
  f32x4 as = {0, 0, 0, 0}, at = {0, 0, 0, 0};
  const int p_end = live ? min(k.P, (slab + 1) * ARB_SLAB) : 0;
  for (int p = slab * ARB_SLAB + pl; p < p_end; p += 16) {
    const size_t pix = (size_t)b * k.P + p;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(k.x + pix * k.x_ld + c);
    const f32x4 dv = *reinterpret_cast<const f32x4*>(k.da + pix * k.da_ld + c);
    f32x4 g;
    g.x = (k.nomask || xv.x * s4.x + t4.x > 0.f) ? dv.x : 0.f;
    g.y = (k.nomask || xv.y * s4.y + t4.y > 0.f) ? dv.y : 0.f;
    g.z = (k.nomask || xv.z * s4.z + t4.z > 0.f) ? dv.z : 0.f;
    g.w = (k.nomask || xv.w * s4.w + t4.w > 0.f) ? dv.w : 0.f;
    as += g * xv;
    at += g;
    f32x4 o = g * s4;
    if (k.skip && c < k.skip_C) {
      if (k.skip_ups) {
        const int yy = p / k.W, xx = p - yy * k.W;
        const int W2 = 2 * k.W;
        const size_t q = ((size_t)b * (2 * k.H) + 2 * yy) * W2 + 2 * xx;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(k.skip + q * k.skip_ld + c);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(k.skip + (q + 1) * k.skip_ld + c);
        const f32x4 c2 = *reinterpret_cast<const f32x4*>(k.skip + (q + W2) * k.skip_ld + c);
        const f32x4 c3 = *reinterpret_cast<const f32x4*>(k.skip + (q + W2 + 1) * k.skip_ld + c);
        o += (c0 + c1) + (c2 + c3);
      } else {
        o += *reinterpret_cast<const f32x4*>(k.skip + pix * k.skip_ld + c);
      }
    }
    *reinterpret_cast<f32x4*>(k.dx + pix * k.dx_ld + c) = o;
  }
  red_s[tid] = as;
  red_t[tid] = at;
  __syncthreads();
  if (pl == 0 && live) {
    f32x4 a = red_s[cl], bsum = red_t[cl];
#pragma unroll
    for (int j = 1; j < 16; ++j) {
      a += red_s[j * 16 + cl];
      bsum += red_t[j * 16 + cl];
    }
    const size_t o = ((size_t)b * k.nblk + slab) * k.C + c;
    *reinterpret_cast<f32x4*>(k.partial + o) = a;
    *reinterpret_cast<f32x4*>(k.partial + (size_t)k.Bn * k.nblk * k.C + o) = bsum;
  }
}

// second stage of the per-(sample, channel) sums: partial[2][B][nblk][C] -> ds, dt.
// grid (C/64, B); 16 segments x 16 channel-float4 lanes per block; a thread adds its segment's rows
// (seg, seg + 16, ...) in FOUR independent chains (row index / 16 mod 4), combined ((0 + 1) + (2 + 3)), then
// the segments in a fixed order -- an order set by the constants, the same for the per-layer and the
// grouped launch and for every batch.  (Round 5: four chains instead of one -- the 256^2 layers hold 2048
// partial rows per image and 36 blocks work on them, a 128-deep chain of dependent loads; 64 segments of
// 1024 threads were tried first and lost more to dispatching the group's 27 k mostly empty blocks.)
constexpr int ARB_SEGS = 16, ARB_CHAINS = 4;
__device__ __forceinline__ void arb_finish_body(const float* partial, float* ds, float* dt, int Bn,
                                                int nblk, int C, int out_bstride, int cgroup, int b,
                                                f32x4* red_s, f32x4* red_t) {
  const int tid = threadIdx.x, cl = tid & 15, seg = tid >> 4;
  const int c = cgroup * 64 + cl * 4;
  const size_t half = (size_t)Bn * nblk * C;
  f32x4 a[ARB_CHAINS], t[ARB_CHAINS];
#pragma unroll
  for (int q = 0; q < ARB_CHAINS; ++q) { a[q] = f32x4{0, 0, 0, 0}; t[q] = f32x4{0, 0, 0, 0}; }
  const bool live = c < C;
  if (live) {
    int j = seg;
    for (; j + (ARB_CHAINS - 1) * ARB_SEGS < nblk; j += ARB_CHAINS * ARB_SEGS) {
#pragma unroll
      for (int q = 0; q < ARB_CHAINS; ++q) {
        const size_t o = ((size_t)b * nblk + j + q * ARB_SEGS) * C + c;
        a[q] += *reinterpret_cast<const f32x4*>(partial + o);
        t[q] += *reinterpret_cast<const f32x4*>(partial + half + o);
      }
    }
#pragma unroll
    for (int q = 0; q < ARB_CHAINS - 1; ++q) {
      if (j + q * ARB_SEGS < nblk) {
        const size_t o = ((size_t)b * nblk + j + q * ARB_SEGS) * C + c;
        a[q] += *reinterpret_cast<const f32x4*>(partial + o);
        t[q] += *reinterpret_cast<const f32x4*>(partial + half + o);
      }
    }
    a[0] = (a[0] + a[1]) + (a[2] + a[3]);
    t[0] = (t[0] + t[1]) + (t[2] + t[3]);
  }
  red_s[tid] = a[0];
  red_t[tid] = t[0];
  __syncthreads();
  if (seg == 0 && live) {
    f32x4 sa = a[0], st = t[0];
#pragma unroll
    for (int k = 1; k < ARB_SEGS; ++k) {
      sa += red_s[k * 16 + cl];
      st += red_t[k * 16 + cl];
    }
    *reinterpret_cast<f32x4*>(ds + (size_t)b * out_bstride + c) = sa;
    *reinterpret_cast<f32x4*>(dt + (size_t)b * out_bstride + c) = st;
  }
}
__global__ __launch_bounds__(ARB_SEGS * 16) void arb_finish_kernel(const float* partial, float* ds,
                                                                   float* dt, int Bn, int nblk, int C,
                                                                   int out_bstride) {
  __shared__ f32x4 red_s[ARB_SEGS * 16], red_t[ARB_SEGS * 16];
  arb_finish_body(partial, ds, dt, Bn, nblk, C, out_bstride, blockIdx.x, blockIdx.y, red_s, red_t);
}

// the same reduction for up to ARB_GROUP_MAX recorded layers in one launch: grid (all channel groups of
// all layers, B); e[i].first = the first block of layer i
constexpr int ARB_GROUP_MAX = 56;
struct ArbFin { const float* partial; float* ds; float* dt; int nblk, C, bstride, first; };
struct ArbFinGroup { ArbFin e[ARB_GROUP_MAX]; int n, Bn; };
__global__ __launch_bounds__(ARB_SEGS * 16) void arb_finish_group_kernel(const ArbFinGroup g) {
  __shared__ f32x4 red_s[ARB_SEGS * 16], red_t[ARB_SEGS * 16];
  int l = 0;
  while (l + 1 < g.n && (int)blockIdx.x >= g.e[l + 1].first) ++l;      // (uniform: scalar loads)
  const ArbFin& e = g.e[l];
  arb_finish_body(e.partial, e.ds, e.dt, g.Bn, e.nblk, e.C, e.bstride, (int)blockIdx.x - e.first, blockIdx.y,
                  red_s, red_t);
}
thread_local bool g_arb_defer = false;
thread_local ArbFinGroup g_arb_group;

// ---------------------------------------------------------------------------
// softmax over rows of `cols` (multiple of 256, <= 2048): one wave per row
// ---------------------------------------------------------------------------
template <int VPL>  // float4 per lane
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* S, float* P,
                                                          long long rows, int cols) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* sp = S + row * cols;
  float* pp = P + row * cols;
  f32x4 v[VPL];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] = *reinterpret_cast<const f32x4*>(sp + (i * 64 + lane) * 4);
    m = fmaxf(m, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
  }
  m = wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i].x = expf(v[i].x - m);
    v[i].y = expf(v[i].y - m);
    v[i].z = expf(v[i].z - m);
    v[i].w = expf(v[i].w - m);
    sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int i = 0; i < VPL; ++i)
    *reinterpret_cast<f32x4*>(pp + (i * 64 + lane) * 4) = v[i] * inv;
}

template <int VPL>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* P, const float* dP,
                                                          float* dS, long long rows,
                                                          int cols) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  f32x4 p[VPL], d[VPL];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    p[i] = *reinterpret_cast<const f32x4*>(P + row * cols + (i * 64 + lane) * 4);
    d[i] = *reinterpret_cast<const f32x4*>(dP + row * cols + (i * 64 + lane) * 4);
    dot += (p[i].x * d[i].x + p[i].y * d[i].y) + (p[i].z * d[i].z + p[i].w * d[i].w);
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < VPL; ++i)
    *reinterpret_cast<f32x4*>(dS + row * cols + (i * 64 + lane) * 4) = p[i] * (d[i] - dot);
}

// ---------------------------------------------------------------------------
// pooling / relu backward
// ---------------------------------------------------------------------------
// (grid.y = image: a block stays inside one image, so that it can leave ONE partial maximum of
// what it wrote per wave for the conv that reads dy next -- P2LAmax, amax_out[b][gridDim.x * 4])
__global__ void maxpool2_bwd_kernel(const float* y, int y_ld, const float* dyp,
                                    int dyp_ld, const float* add, int add_ld,
                                    float* dy, int dy_ld, int Bn, int H, int W,
                                    int C, int relu_mask, float* amax_out) {
  // one thread per (quad, float4 channel)
  const int C4 = C >> 2, Hh = H >> 1, Wh = W >> 1;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)Hh * Wh * C4;
  const int b = blockIdx.y;
  float mx = 0.f;
  if (idx < total) {
  const int c = (int)(idx % C4) * 4;
  size_t q = idx / C4;
  const int qx = (int)(q % Wh);
  const int qy = (int)(q / Wh);
  const size_t pq = ((size_t)b * Hh + qy) * Wh + qx;
  const f32x4 g = *reinterpret_cast<const f32x4*>(dyp + pq * dyp_ld + c);
  size_t pix[4];
  f32x4 v[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    pix[s] = ((size_t)b * H + 2 * qy + (s >> 1)) * W + 2 * qx + (s & 1);
    v[s] = *reinterpret_cast<const f32x4*>(y + pix[s] * y_ld + c);
  }
  f32x4 o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    // first maximum in window scan order (row-major), as ATen max_pool2d does
    int arg = 0;
    float m = v[0][e];
#pragma unroll
    for (int s = 1; s < 4; ++s)
      if (v[s][e] > m) { m = v[s][e]; arg = s; }
#pragma unroll
    for (int s = 0; s < 4; ++s) o[s][e] = (s == arg) ? g[e] : 0.f;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    f32x4 r = o[s];
    if (add) r += *reinterpret_cast<const f32x4*>(add + pix[s] * add_ld + c);
    if (relu_mask) {
      r.x = v[s].x > 0.f ? r.x : 0.f;
      r.y = v[s].y > 0.f ? r.y : 0.f;
      r.z = v[s].z > 0.f ? r.z : 0.f;
      r.w = v[s].w > 0.f ? r.w : 0.f;
    }
    *reinterpret_cast<f32x4*>(dy + pix[s] * dy_ld + c) = r;
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
  }
  }
  if (amax_out != nullptr) {                           // one partial per wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0)
      amax_out[((size_t)b * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = mx;
  }
}

__global__ void relu_mask_kernel(const float* y, int y_ld, const float* g, int g_ld,
                                 float* dy, int dy_ld, long long P, int C) {
  const int C4 = C >> 2;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)P * C4) return;
  const int c = (int)(idx % C4) * 4;
  const size_t p = idx / C4;
  const f32x4 yv = *reinterpret_cast<const f32x4*>(y + p * y_ld + c);
  f32x4 gv = *reinterpret_cast<const f32x4*>(g + p * g_ld + c);
  gv.x = yv.x > 0.f ? gv.x : 0.f;
  gv.y = yv.y > 0.f ? gv.y : 0.f;
  gv.z = yv.z > 0.f ? gv.z : 0.f;
  gv.w = yv.w > 0.f ? gv.w : 0.f;
  *reinterpret_cast<f32x4*>(dy + p * dy_ld + c) = gv;
}

// ---------------------------------------------------------------------------
// image layout helpers
// ---------------------------------------------------------------------------
__global__ void nchw3_to_nhwc16_kernel(const float* src, float* dst, int Bn, int HW) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)Bn * HW) return;
  const size_t b = idx / HW, p = idx - b * HW;
  const float* s = src + b * 3 * HW + p;
  f32x4 z = {0, 0, 0, 0};
  f32x4 v = {s[0], s[HW], s[2 * (size_t)HW], 0.f};
  f32x4* d = reinterpret_cast<f32x4*>(dst + idx * 16);
  d[0] = v; d[1] = z; d[2] = z; d[3] = z;
}
__global__ void nhwc16_to_nchw3_kernel(const float* src, float* dst, int Bn, int HW) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)Bn * HW) return;
  const size_t b = idx / HW, p = idx - b * HW;
  const f32x4 v = *reinterpret_cast<const f32x4*>(src + idx * 16);
  float* d = dst + b * 3 * HW + p;
  d[0] = v.x; d[HW] = v.y; d[2 * (size_t)HW] = v.z;
}
__global__ void tanh_bwd16_kernel(const float* img, float* dimg, long long P) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)P) return;
  const f32x4 o = *reinterpret_cast<const f32x4*>(img + idx * 16);
  f32x4* dp = reinterpret_cast<f32x4*>(dimg + idx * 16);
  f32x4 d = dp[0];
  d.x *= (1.f - o.x * o.x);
  d.y *= (1.f - o.y * o.y);
  d.z *= (1.f - o.z * o.z);
  d.w = 0.f;
  dp[0] = d;
}

// ---------------------------------------------------------------------------
// losses
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void weight_sum_kernel(const float* weight,
                                                          const float* mask,
                                                          float* wsum, int n) {
  __shared__ float red[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* w = weight + (size_t)b * n;
  const float* m = mask ? mask + (size_t)b * n : nullptr;
  float acc = 0.f;
  for (int i = tid; i < n; i += 1024) acc += m ? w[i] * m[i] : w[i];
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += red[i];
    wsum[b] = s;
  }
}

__global__ __launch_bounds__(256) void l1_fwd_kernel(const float* img16,
                                                     const float* target,
                                                     const float* weight,
                                                     const float* mask,
                                                     float* partial, int HW) {
  __shared__ float red[4];
  const int b = blockIdx.y, nblk = gridDim.x;
  const int p = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  if (p < HW) {
    const f32x4 o = *reinterpret_cast<const f32x4*>(img16 + ((size_t)b * HW + p) * 16);
    const size_t base = (size_t)b * 3 * HW + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float w = weight[base + (size_t)c * HW];
      if (mask) w *= mask[base + (size_t)c * HW];
      acc += fabsf(target[base + (size_t)c * HW] - o[c]) * w;
    }
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[(size_t)b * nblk + blockIdx.x] = s;
}

__global__ void l1_bwd_kernel(const float* img16, const float* target,
                              const float* weight, const float* mask,
                              const float* wsum, const float* gscale,
                              float* dimg16, int HW, int accumulate) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const size_t pix = (size_t)b * HW + p;
  const f32x4 o = *reinterpret_cast<const f32x4*>(img16 + pix * 16);
  const size_t base = (size_t)b * 3 * HW + p;
  const float gs = gscale[b] / wsum[b];
  f32x4 d = {0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float w = weight[base + (size_t)c * HW];
    if (mask) w *= mask[base + (size_t)c * HW];
    const float diff = o[c] - target[base + (size_t)c * HW];
    const float sg = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
    d[c] = gs * sg * w;
  }
  f32x4* dp = reinterpret_cast<f32x4*>(dimg16 + pix * 16);
  if (accumulate) {
    dp[0] = dp[0] + d;
  } else {
    const f32x4 z = {0, 0, 0, 0};
    dp[0] = d; dp[1] = z; dp[2] = z; dp[3] = z;
  }
}

// LPIPS: LP lanes cooperate on one pixel, each holding C/(4*LP) float4.
// LP = largest power of two dividing C/4, capped at a wave (C = 192 -> 16 lanes x 3 float4)
constexpr int lpips_lp(int c4) {
  int lp = 1;
  while (lp < 64 && c4 % (lp * 2) == 0) lp *= 2;
  return lp;
}
template <int C>
struct LpipsCfg {
  static constexpr int LP = lpips_lp(C / 4);
  static constexpr int VPL = C / (4 * LP);
  static constexpr int PPW = 64 / LP;  // pixels per wave
};
template <int LP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int C>
__global__ __launch_bounds__(256) void lpips_normalize_kernel(const float* f, float* nf,
                                                              long long P) {
  using Cfg = LpipsCfg<C>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / Cfg::LP, ll = lane % Cfg::LP;
  const long long p = ((long long)blockIdx.x * 4 + wave) * Cfg::PPW + sub;
  if (p >= P) return;
  f32x4 v[Cfg::VPL];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < Cfg::VPL; ++i) {
    v[i] = *reinterpret_cast<const f32x4*>(f + p * C + (i * Cfg::LP + ll) * 4);
    ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  ss = group_sum<Cfg::LP>(ss);
  const float inv = 1.f / (sqrtf(ss) + 1e-10f);
#pragma unroll
  for (int i = 0; i < Cfg::VPL; ++i)
    *reinterpret_cast<f32x4*>(nf + p * C + (i * Cfg::LP + ll) * 4) = v[i] * inv;
}

struct LpipsK {
  const float* f; const float* nft; const float* lin; const float* wt;
  const float* gscale; float* out;  // partial (fwd) or df (bwd)
  long long nft_bstride, wt_bstride;
  int Bn, P, nblk;
  // backward of a tap that a 2x2 max pool follows (p2l_lpips_tap_pool_bwd): the pooled gradient, the
  // tap's width in pixels, one partial maximum of |out| per block
  const float* dyp; float* amax; int W;
};

// POOL (backward only): the tap is also the input of relu -> 2x2 max pool; the kernel adds the pool's
// backward of k.dyp and applies the ReLU mask, i.e. writes the whole gradient of the conv output in one
// pass (until round 5: tap gradient written, then read again by maxpool2_bwd_kernel with the tensor itself
// -- 1.9 GB instead of 1.0 GB at the 256^2 tap of 18 candidates).  A block covers 2 rows x 2 PPW columns,
// whole quads, so the three partner pixels of a quad come from the block's own loads (L1 hits).
template <int C, bool BWD, bool POOL = false>
__global__ __launch_bounds__(256) void lpips_tap_kernel(const LpipsK k) {
  using Cfg = LpipsCfg<C>;
  __shared__ float red[4];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / Cfg::LP, ll = lane % Cfg::LP;
  int p = (blockIdx.x * 4 + wave) * Cfg::PPW + sub;
  int py = 0, px = 0;
  if (POOL) {
    constexpr int HALF = 2 * Cfg::PPW;                   // columns of a block
    const int j = wave * Cfg::PPW + sub, bpr = k.W / HALF;
    py = 2 * (blockIdx.x / bpr) + j / HALF;
    px = (blockIdx.x % bpr) * HALF + j % HALF;
    p = py * k.W + px;
  }
  float contrib = 0.f, mx = 0.f;
  if (p < k.P) {
    const float* fp = k.f + ((size_t)b * k.P + p) * C;
    const float* tp = k.nft + (size_t)b * k.nft_bstride + (size_t)p * C;
    f32x4 v[Cfg::VPL], tv[Cfg::VPL], lw[Cfg::VPL];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < Cfg::VPL; ++i) {
      const int c = (i * Cfg::LP + ll) * 4;
      v[i] = *reinterpret_cast<const f32x4*>(fp + c);
      tv[i] = *reinterpret_cast<const f32x4*>(tp + c);
      lw[i] = *reinterpret_cast<const f32x4*>(k.lin + c);
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    ss = group_sum<Cfg::LP>(ss);
    const float nrm = sqrtf(ss);
    const float inv = 1.f / (nrm + 1e-10f);
    const float w = k.wt[(size_t)b * k.wt_bstride + p];
    if (!BWD) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < Cfg::VPL; ++i) {
        const f32x4 e = v[i] * inv - tv[i];
        const f32x4 q = lw[i] * e * e;
        d += (q.x + q.y) + (q.z + q.w);
      }
      d = group_sum<Cfg::LP>(d);
      contrib = (ll == 0) ? d * w : 0.f;
    } else {
      // u_c = 2 lin_c (nf_c - nt_c);  dd/df_j = u_j*inv - (sum_c u_c f_c) f_j inv^2 / nrm
      f32x4 u[Cfg::VPL];
      float uf = 0.f;
#pragma unroll
      for (int i = 0; i < Cfg::VPL; ++i) {
        u[i] = 2.f * lw[i] * (v[i] * inv - tv[i]);
        uf += (u[i].x * v[i].x + u[i].y * v[i].y) + (u[i].z * v[i].z + u[i].w * v[i].w);
      }
      uf = group_sum<Cfg::LP>(uf);
      const float gsw = k.gscale[b] * w;
      const float c2 = (nrm > 0.f) ? uf * inv * inv / nrm : 0.f;
      float* dp = k.out + ((size_t)b * k.P + p) * C;
#pragma unroll
      for (int i = 0; i < Cfg::VPL; ++i) {
        f32x4 g = gsw * (u[i] * inv - c2 * v[i]);
        if (POOL) {
          const int c = (i * Cfg::LP + ll) * 4, s_own = (py & 1) * 2 + (px & 1);
          const size_t quad = ((size_t)b * (k.P / k.W / 2) + (py >> 1)) * (k.W >> 1) + (px >> 1);
          const f32x4 gp = *reinterpret_cast<const f32x4*>(k.dyp + quad * C + c);
          f32x4 q[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const size_t pix = ((size_t)b * k.P) + (size_t)((py & ~1) + (s >> 1)) * k.W + (px & ~1) + (s & 1);
            q[s] = *reinterpret_cast<const f32x4*>(k.f + pix * C + c);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // first maximum in window scan order (row-major), as ATen max_pool2d does
            int arg = 0;
            float m = q[0][e];
#pragma unroll
            for (int s = 1; s < 4; ++s)
              if (q[s][e] > m) { m = q[s][e]; arg = s; }
            const float r = __fadd_rn((s_own == arg) ? gp[e] : 0.f, g[e]);   // (no contraction: = the two-pass form)
            g[e] = v[i][e] > 0.f ? r : 0.f;
          }
          mx = fmaxf(fmaxf(mx, fmaxf(fabsf(g.x), fabsf(g.y))), fmaxf(fabsf(g.z), fabsf(g.w)));
        }
        *reinterpret_cast<f32x4*>(dp + (i * Cfg::LP + ll) * 4) = g;
      }
    }
  }
  if (!BWD) {
    const float s = block_sum_256(contrib, red);
    if (threadIdx.x == 0) k.out[(size_t)b * k.nblk + blockIdx.x] = s;
  }
  if (POOL && k.amax != nullptr) {                       // one partial maximum per block (P2LAmax.in of the next dgrad)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0)
      k.amax[(size_t)b * k.nblk + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
}

// adjoint of bilinear upsampling (align_corners=False) h x w -> H x W
__global__ void bilinear_adjoint_kernel(const float* wsrc, float* wt, int H, int W,
                                        int h, int w) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= h * w) return;
  const int qy = q / w, qx = q - qy * w;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  // conservative source range (any ratio, e.g. 63 -> 256): fy in (qy-1, qy+1); the exact
  // membership test is inside the loop
  const int py0 = qy == 0 ? 0 : max(0, (int)floorf((qy - 0.5f) / sy - 0.5f) - 1);
  const int py1 = qy == h - 1 ? H : min(H, (int)ceilf((qy + 1.5f) / sy - 0.5f) + 2);
  const int px0 = qx == 0 ? 0 : max(0, (int)floorf((qx - 0.5f) / sx - 0.5f) - 1);
  const int px1 = qx == w - 1 ? W : min(W, (int)ceilf((qx + 1.5f) / sx - 0.5f) + 2);
  float acc = 0.f;
  for (int py = py0; py < py1; ++py) {
    const float fy = fmaxf(sy * (py + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly = fy - y0;
    float wy = 0.f;
    if (y0 == qy) wy += 1.f - ly;
    if (y1 == qy) wy += ly;
    if (wy == 0.f) continue;
    float row = 0.f;
    for (int px = px0; px < px1; ++px) {
      const float fx = fmaxf(sx * (px + 0.5f) - 0.5f, 0.f);
      const int x0 = (int)fx;
      const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
      const float lx = fx - x0;
      float wx = 0.f;
      if (x0 == qx) wx += 1.f - lx;
      if (x1 == qx) wx += lx;
      if (wx != 0.f) row += wx * wsrc[((size_t)b * H + py) * W + px];
    }
    acc += wy * row;
  }
  wt[(size_t)b * h * w + q] = acc;
}

// wsrc[b][p] = sum_c weight[b][c][p] * mask[b][c][p]
__global__ void weight_map_kernel(const float* weight, const float* mask, float* wsrc,
                                  int HW) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const size_t base = (size_t)b * 3 * HW + p;
  float a = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float w = weight[base + (size_t)c * HW];
    if (mask) w *= mask[base + (size_t)c * HW];
    a += w;
  }
  wsrc[(size_t)b * HW + p] = a;
}

__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* partial,
                                                          float* out, int n, float scale,
                                                          const float* div,
                                                          int accumulate) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[(size_t)b * n + i];
  float s = block_sum_256(acc, red) * scale;
  if (threadIdx.x == 0) {
    if (div) s /= div[b];
    out[b] = accumulate ? out[b] + s : s;
  }
}

// fused F.affine_grid + F.grid_sample (bilinear, zero padding, align_corners=False) for
// NCHW images; theta[b] = [a00 a01 a02 a10 a11 a12] (reference
// pix2latent/transform/spatial_transform.py:69-104)
__global__ void affine_grid_sample_kernel(const float* src, const float* theta, float* dst,
                                          int Bn, int C, int H, int W) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)Bn * H * W) return;
  const int x = (int)(idx % W);
  const int y = (int)((idx / W) % H);
  const int b = (int)(idx / ((size_t)W * H));
  const float* th = theta + b * 6;
  const float gx0 = (2.f * x + 1.f) / W - 1.f, gy0 = (2.f * y + 1.f) / H - 1.f;
  const float gx = th[0] * gx0 + th[1] * gy0 + th[2];
  const float gy = th[3] * gx0 + th[4] * gy0 + th[5];
  const float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
  const bool vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
  for (int c = 0; c < C; ++c) {
    const float* s = src + ((size_t)b * C + c) * H * W;
    float v = 0.f;
    if (vy0 && vx0) v += s[(size_t)y0 * W + x0] * (wy0 * wx0);
    if (vy0 && vx1) v += s[(size_t)y0 * W + x1] * (wy0 * wx1);
    if (vy1 && vx0) v += s[(size_t)y1 * W + x0] * (wy1 * wx0);
    if (vy1 && vx1) v += s[(size_t)y1 * W + x1] * (wy1 * wx1);
    dst[((size_t)b * C + c) * H * W + (size_t)y * W + x] = v;
  }
}

// ---- backward of the fused warp (differentiable uses of SpatialTransform on the device:
// invertibility_loss, gradient-based search of t).  Both in fixed summation order.
// d src: GATHER form of the adjoint.  Source pixel (sx, sy) receives from every output pixel whose
// sampling position (ix, iy) lies in (sx-1, sx+1) x (sy-1, sy+1); that set is the pre-image of a
// 2x2 box under the affine map, a parallelogram whose bounding box follows from the inverse of
// theta's 2x2 part (the whole image when the map is singular).  Visited row-major: no atomics.
// Cost: sum over source pixels of their box = H W C max(4, 4 / |det|) products -- fine for the scales a
// transform search visits (|det| ~ 0.25 ... 4), quadratic in the image when a search drives the scale
// towards 0 (the price of a fixed summation order; the reference's scatter-add is not reproducible).
__global__ void affine_grid_sample_bwd_src_kernel(const float* dout, const float* theta, float* dsrc,
                                                  int Bn, int C, int H, int W) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)Bn * H * W) return;
  const int sx = (int)(idx % W);
  const int sy = (int)((idx / W) % H);
  const int b = (int)(idx / ((size_t)W * H));
  const float* th = theta + b * 6;
  // ix = ax*x + bx*y + cx, iy = ay*x + by*y + cy in PIXEL coordinates of the output grid
  const float ax = th[0], bx = th[1] * (float)W / (float)H, ay = th[3] * (float)H / (float)W, by = th[4];
  const float cx = 0.5f * ((th[0] * (1.f / W - 1.f) + th[1] * (1.f / H - 1.f) + th[2] + 1.f) * W - 1.f);
  const float cy = 0.5f * ((th[3] * (1.f / W - 1.f) + th[4] * (1.f / H - 1.f) + th[5] + 1.f) * H - 1.f);
  int x_lo = 0, x_hi = W - 1, y_lo = 0, y_hi = H - 1;
  const float det = ax * by - bx * ay;
  if (fabsf(det) > 1e-12f) {
    // corners of the box (sx +- 1, sy +- 1) mapped back to the output grid
    float mnx = 3.4e38f, mxx = -3.4e38f, mny = 3.4e38f, mxy = -3.4e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float u = (float)sx + ((k & 1) ? 1.f : -1.f) - cx, v = (float)sy + ((k & 2) ? 1.f : -1.f) - cy;
      const float ox = (by * u - bx * v) / det, oy = (-ay * u + ax * v) / det;
      mnx = fminf(mnx, ox); mxx = fmaxf(mxx, ox); mny = fminf(mny, oy); mxy = fmaxf(mxy, oy);
    }
    // (one pixel of slack against rounding of the inverse; every candidate is re-tested below)
    // (clamped on BOTH sides before the int cast: a near-singular theta sends the corners to 1e30 or NaN,
    //  and converting those is undefined; fminf / fmaxf drop a NaN operand)
    const float Wf = (float)W + 2.f, Hf = (float)H + 2.f;
    x_lo = max(0, (int)floorf(fminf(fmaxf(mnx, -2.f), Wf)) - 1); x_hi = min(W - 1, (int)ceilf(fmaxf(fminf(mxx, Wf), -2.f)) + 1);
    y_lo = max(0, (int)floorf(fminf(fmaxf(mny, -2.f), Hf)) - 1); y_hi = min(H - 1, (int)ceilf(fmaxf(fminf(mxy, Hf), -2.f)) + 1);
  }
  for (int c = 0; c < C; ++c) {
    const float* g = dout + ((size_t)b * C + c) * H * W;
    float acc = 0.f;
    for (int y = y_lo; y <= y_hi; ++y)
      for (int x = x_lo; x <= x_hi; ++x) {
        // EXACTLY the forward kernel's arithmetic for this output pixel
        const float gx0 = (2.f * x + 1.f) / W - 1.f, gy0 = (2.f * y + 1.f) / H - 1.f;
        const float gx = th[0] * gx0 + th[1] * gy0 + th[2];
        const float gy = th[3] * gx0 + th[4] * gy0 + th[5];
        const float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float wx1 = ix - fx, wy1 = iy - fy;
        float wx = 0.f, wy = 0.f;
        if (x0 == sx) wx = 1.f - wx1; else if (x0 + 1 == sx) wx = wx1; else continue;
        if (y0 == sy) wy = 1.f - wy1; else if (y0 + 1 == sy) wy = wy1; else continue;
        acc += g[(size_t)y * W + x] * (wy * wx);
      }
    dsrc[((size_t)b * C + c) * H * W + (size_t)sy * W + sx] = acc;
  }
}
// d theta: per output pixel d L / d (ix, iy) summed over channels, chained through the affine grid;
// block partials [B][nblk][6] in fixed shuffle order, then one thread per (image, entry) adds the
// blocks in order.
__global__ __launch_bounds__(256) void affine_grid_sample_bwd_theta_kernel(
    const float* src, const float* theta, const float* dout, float* partial, int C, int H, int W) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  float v6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p < H * W) {
    const int x = p % W, y = p / W;
    const float* th = theta + b * 6;
    const float gx0 = (2.f * x + 1.f) / W - 1.f, gy0 = (2.f * y + 1.f) / H - 1.f;
    const float gx = th[0] * gx0 + th[1] * gy0 + th[2];
    const float gy = th[3] * gx0 + th[4] * gy0 + th[5];
    const float ix = ((gx + 1.f) * W - 1.f) * 0.5f, iy = ((gy + 1.f) * H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
    const bool vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < C; ++c) {
      const float* s = src + ((size_t)b * C + c) * H * W;
      const float s00 = (vy0 && vx0) ? s[(size_t)y0 * W + x0] : 0.f;
      const float s01 = (vy0 && vx1) ? s[(size_t)y0 * W + x1] : 0.f;
      const float s10 = (vy1 && vx0) ? s[(size_t)y1 * W + x0] : 0.f;
      const float s11 = (vy1 && vx1) ? s[(size_t)y1 * W + x1] : 0.f;
      const float g = dout[((size_t)b * C + c) * H * W + p];
      gix += g * ((s01 - s00) * wy0 + (s11 - s10) * wy1);
      giy += g * ((s10 - s00) * wx0 + (s11 - s01) * wx1);
    }
    const float dgx = gix * (0.5f * W), dgy = giy * (0.5f * H);
    v6[0] = dgx * gx0; v6[1] = dgx * gy0; v6[2] = dgx;
    v6[3] = dgy * gx0; v6[4] = dgy * gy0; v6[5] = dgy;
  }
  __shared__ float red[4][6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float a = v6[k];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = a;
  }
  __syncthreads();
  if (threadIdx.x < 6)
    partial[((size_t)b * gridDim.x + blockIdx.x) * 6 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void affine_grid_sample_bwd_theta_finish(const float* partial, float* dtheta, int Bn, int nblk) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= Bn * 6) return;
  const int b = i / 6, k = i - b * 6;
  float a = 0.f;
  for (int j = 0; j < nblk; ++j) a += partial[((size_t)b * nblk + j) * 6 + k];
  dtheta[i] = a;
}

__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long long n,
                            float step_size, float beta1, float beta2, float eps,
                            float bc2_sqrt) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  // torch.optim.Adam (single tensor): exp_avg.lerp_(grad, 1-beta1);
  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2);
  // denom = sqrt(exp_avg_sq)/sqrt(bc2) + eps ; p += -step_size * exp_avg/denom
  const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
  const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - step_size * (mi / denom);
}

// Adam with the step number in DEVICE memory: nothing about the launch changes from step to
// step, so a captured HIP graph of the inner step can be replayed (optimizer/base_optimizer.py).
// Same arithmetic as adam_kernel; the bias corrections are formed in double like torch does
// on the host.  The counters are advanced by a second, one-block kernel behind it.
__global__ void adam_dev_kernel(float* p, const float* g, float* m, float* v, long long n,
                                float lr, float beta1, float beta2, float eps,
                                const int* __restrict__ steps) {
  // the bias corrections once per block (two double-precision pow per ELEMENT were most of the 110 us this kernel
  // took on StyleGAN2's 8.4 M noise values); same values, same arithmetic per element
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const double step = (double)(steps[0] + 1);
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    bc[0] = (float)((double)lr / bc1);
    bc[1] = (float)sqrt(bc2);
  }
  __syncthreads();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float step_size = bc[0];
  const float bc2_sqrt = bc[1];
  const float gi = g[i];
  const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
  const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - step_size * (mi / denom);
}
__global__ void counters_advance_kernel(int* c, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) c[i] += 1;
}

__global__ void clamp_kernel(float* p, long long n, float lo, float hi) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = fminf(fmaxf(p[i], lo), hi);
}

__global__ void mfma_probe_kernel(const float* A, const float* B, float* C, int K) {
  const int lane = threadIdx.x;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < K; k += 2) {
    const float a = A[(lane & 31) * K + k + (lane >> 5)];
    const float b = B[(k + (lane >> 5)) * 32 + (lane & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    C[row * 32 + (lane & 31)] = acc[r];
  }
}

}  // namespace

#define ST(s) ((hipStream_t)(s))

extern "C" int p2l_linear_fwd_ld(const float* x, int x_ld, const float* W, const float* bias,
                                 float* y, int Bn, int K, int N, void* stream);
extern "C" int p2l_linear_fwd(const float* x, const float* W, const float* bias,
                              float* y, int Bn, int K, int N, void* stream) {
  return p2l_linear_fwd_ld(x, K, W, bias, y, Bn, K, N, stream);
}
extern "C" int p2l_linear_fwd_ld(const float* x, int x_ld, const float* W, const float* bias,
                                 float* y, int Bn, int K, int N, void* stream) {
  if (!x || !W || !y || K % 4 || K > 1024 || Bn < 1) return P2L_EINVAL;
  // every launch streams W once: rows in groups of 16, or of 24 when that saves a launch (the
  // population of 18: one pass over the 33 MB gen_z matrix instead of two); a row's sum does not
  // depend on its group
  constexpr int BG = 16, BGW = 24;
  const size_t lds = (size_t)(BG * K + 4 * BG * 64) * sizeof(float);
  const size_t ldsw = (size_t)(BGW * K + 4 * BGW * 64) * sizeof(float);
  if (cdiv(Bn, BGW) < cdiv(Bn, BG) && ldsw <= 64 * 1024) {
    for (int b0 = 0; b0 < Bn; b0 += BGW)
      hipLaunchKernelGGL(linear_fwd_kernel<BGW>, dim3(cdiv(N, 64)), dim3(256), ldsw,
                         ST(stream), x, W, bias, y, Bn, b0, K, N, x_ld);
    return p2l_check_launch();
  }
  for (int b0 = 0; b0 < Bn; b0 += BG) {
    hipLaunchKernelGGL(linear_fwd_kernel<BG>, dim3(cdiv(N, 64)), dim3(256), lds,
                       ST(stream), x, W, bias, y, Bn, b0, K, N, x_ld);
  }
  return p2l_check_launch();
}

extern "C" int p2l_linear_bwd_ld(const float* dy, const float* W, float* dx, int dx_ld, int Bn,
                                 int K, int N, int accumulate, void* stream);
extern "C" int p2l_linear_bwd(const float* dy, const float* W, float* dx, int Bn,
                              int K, int N, int accumulate, void* stream) {
  return p2l_linear_bwd_ld(dy, W, dx, K, Bn, K, N, accumulate, stream);
}
extern "C" int p2l_linear_bwd_ld(const float* dy, const float* W, float* dx, int dx_ld, int Bn,
                                 int K, int N, int accumulate, void* stream) {
  if (!dy || !W || !dx || Bn < 1 || (N % 4)) return P2L_EINVAL;
  constexpr int BG = 16, BGW = 24;      // (as p2l_linear_fwd_ld: 24 rows per pass over W when that saves one)
  if (cdiv(Bn, BGW) < cdiv(Bn, BG)) {
    for (int b0 = 0; b0 < Bn; b0 += BGW)
      hipLaunchKernelGGL(linear_bwd_kernel<BGW>, dim3(K), dim3(1024), 0, ST(stream), dy,
                         W, dx, Bn, b0, K, N, accumulate, dx_ld);
    return p2l_check_launch();
  }
  for (int b0 = 0; b0 < Bn; b0 += BG)
    hipLaunchKernelGGL(linear_bwd_kernel<BG>, dim3(K), dim3(1024), 0, ST(stream), dy,
                       W, dx, Bn, b0, K, N, accumulate, dx_ld);
  return p2l_check_launch();
}

extern "C" int p2l_cbn_fold_fwd(const float* g_raw, const float* b_raw,
                                const float* mean, const float* rstd, float* s,
                                float* t, int Bn, int C, int raw_ld, void* stream) {
  hipLaunchKernelGGL(cbn_fold_fwd_kernel, dim3(cdiv(Bn * C, 256)), dim3(256), 0,
                     ST(stream), g_raw, b_raw, mean, rstd, s, t, Bn, C, raw_ld);
  return p2l_check_launch();
}
extern "C" int p2l_cbn_fold_bwd(const float* ds, const float* dt, const float* mean,
                                const float* rstd, float* dg_raw, float* db_raw,
                                int Bn, int C, int raw_ld, void* stream) {
  hipLaunchKernelGGL(cbn_fold_bwd_kernel, dim3(cdiv(Bn * C, 256)), dim3(256), 0,
                     ST(stream), ds, dt, mean, rstd, dg_raw, db_raw, Bn, C, raw_ld);
  return p2l_check_launch();
}

extern "C" int p2l_affine_relu_bwd_nblk(int P) { return cdiv(P, ARB_SLAB); }

extern "C" int p2l_affine_relu_bwd(const float* da, int da_ld, const float* x,
                                   int x_ld, const float* s, const float* t,
                                   int st_bstride, const float* skip, int skip_ld,
                                   int skip_C, int skip_ups, float* dx, int dx_ld,
                                   float* ds, float* dt, int dsdt_bstride,
                                   float* partial, int Bn, int H, int W, int C,
                                   void* stream) {
  if (!da || !x || !s || !t || !dx || !ds || !dt || !partial) return P2L_EINVAL;
  if (C % 32 || da_ld % 4 || x_ld % 4 || dx_ld % 4 || st_bstride % 4) return P2L_EINVAL;
  if (skip && (skip_ld % 4 || skip_C % 4)) return P2L_EINVAL;
  ArbK k{};
  k.da = da; k.x = x; k.s = s; k.t = t; k.skip = skip; k.dx = dx; k.partial = partial;
  k.da_ld = da_ld; k.x_ld = x_ld; k.dx_ld = dx_ld; k.skip_ld = skip_ld;
  k.skip_C = skip_C; k.skip_ups = skip_ups; k.st_bstride = st_bstride;
  k.Bn = Bn; k.P = H * W; k.C = C; k.H = H; k.W = W;
  k.nblk = cdiv(k.P, ARB_SLAB);
  hipLaunchKernelGGL(affine_relu_bwd_kernel, dim3(k.nblk, C / 64, Bn), dim3(256), 0,
                     ST(stream), k);
  const int rc = p2l_check_launch();
  if (rc) return rc;
  return p2l_arb_finish(partial, ds, dt, Bn, k.nblk, C, dsdt_bstride, stream);
}

extern "C" void p2l_arb_defer_begin(void) {
  g_arb_defer = true;
  g_arb_group.n = 0;
}
extern "C" void p2l_arb_defer_cancel(void) {
  g_arb_defer = false;
  g_arb_group.n = 0;
}
static int arb_group_launch(void* stream) {
  ArbFinGroup& g = g_arb_group;
  if (g.n == 0) return P2L_OK;
  int total = 0;
  for (int i = 0; i < g.n; ++i) { g.e[i].first = total; total += cdiv(g.e[i].C, 64); }
  hipLaunchKernelGGL(arb_finish_group_kernel, dim3(total, g.Bn), dim3(ARB_SEGS * 16), 0,
                     ST(stream), g);
  g.n = 0;
  return p2l_check_launch();
}
extern "C" int p2l_arb_defer_flush(void* stream) {
  g_arb_defer = false;
  return arb_group_launch(stream);
}

extern "C" int p2l_arb_finish(const float* partial, float* ds, float* dt, int Bn, int nblk,
                              int C, int out_bstride, void* stream) {
  if (C % 32 || out_bstride % 4) return P2L_EINVAL;
  if (g_arb_defer) {
    ArbFinGroup& g = g_arb_group;
    if (g.n == ARB_GROUP_MAX || (g.n > 0 && g.Bn != Bn)) {
      const int rc = arb_group_launch(stream);
      if (rc) return rc;
    }
    g.Bn = Bn;
    g.e[g.n++] = ArbFin{partial, ds, dt, nblk, C, out_bstride, 0};
    return P2L_OK;
  }
  hipLaunchKernelGGL(arb_finish_kernel, dim3(cdiv(C, 64), Bn), dim3(ARB_SEGS * 16), 0, ST(stream),
                     partial, ds, dt, Bn, nblk, C, out_bstride);
  return p2l_check_launch();
}

// plain scale backward (StyleGAN2 modulation x*s):  dx = da*s + skip ; ds[b,c] = sum_p da*x
extern "C" int p2l_scale_bwd(const float* da, int da_ld, const float* x, int x_ld, const float* s,
                             int st_bstride, const float* skip, int skip_ld, int skip_C,
                             float* dx, int dx_ld, float* ds, float* dt_scratch, int dsdt_bstride,
                             float* partial, int Bn, int H, int W, int C, void* stream) {
  if (!da || !x || !s || !dx || !ds || !dt_scratch || !partial) return P2L_EINVAL;
  if (C % 32 || da_ld % 4 || x_ld % 4 || dx_ld % 4 || st_bstride % 4) return P2L_EINVAL;
  ArbK k{};
  k.da = da; k.x = x; k.s = s; k.t = s; k.skip = skip; k.dx = dx; k.partial = partial;
  k.da_ld = da_ld; k.x_ld = x_ld; k.dx_ld = dx_ld; k.skip_ld = skip_ld;
  k.skip_C = skip_C; k.skip_ups = 0; k.st_bstride = st_bstride;
  k.Bn = Bn; k.P = H * W; k.C = C; k.H = H; k.W = W; k.nomask = 1;
  k.nblk = cdiv(k.P, ARB_SLAB);
  hipLaunchKernelGGL(affine_relu_bwd_kernel, dim3(k.nblk, cdiv(C, 64), Bn), dim3(256), 0, ST(stream), k);
  const int rc = p2l_check_launch();
  if (rc) return rc;
  // (through p2l_arb_finish: recorded only while the caller defers the finishes, p2l_arb_defer_begin)
  return p2l_arb_finish(partial, ds, dt_scratch, Bn, k.nblk, C, dsdt_bstride, stream);
}

extern "C" int p2l_softmax_fwd(const float* S, float* P, int64_t rows, int cols,
                               void* stream) {
  const dim3 grid(cdiv(rows, 4)), block(256);
  if (cols == 1024) hipLaunchKernelGGL(softmax_fwd_kernel<4>, grid, block, 0, ST(stream), S, P, (long long)rows, cols);
  else if (cols == 256) hipLaunchKernelGGL(softmax_fwd_kernel<1>, grid, block, 0, ST(stream), S, P, (long long)rows, cols);
  else if (cols == 512) hipLaunchKernelGGL(softmax_fwd_kernel<2>, grid, block, 0, ST(stream), S, P, (long long)rows, cols);
  else if (cols == 2048) hipLaunchKernelGGL(softmax_fwd_kernel<8>, grid, block, 0, ST(stream), S, P, (long long)rows, cols);
  else return P2L_EUNSUP;
  return p2l_check_launch();
}
extern "C" int p2l_softmax_bwd(const float* P, const float* dP, float* dS,
                               int64_t rows, int cols, void* stream) {
  const dim3 grid(cdiv(rows, 4)), block(256);
  if (cols == 1024) hipLaunchKernelGGL(softmax_bwd_kernel<4>, grid, block, 0, ST(stream), P, dP, dS, (long long)rows, cols);
  else if (cols == 256) hipLaunchKernelGGL(softmax_bwd_kernel<1>, grid, block, 0, ST(stream), P, dP, dS, (long long)rows, cols);
  else if (cols == 512) hipLaunchKernelGGL(softmax_bwd_kernel<2>, grid, block, 0, ST(stream), P, dP, dS, (long long)rows, cols);
  else if (cols == 2048) hipLaunchKernelGGL(softmax_bwd_kernel<8>, grid, block, 0, ST(stream), P, dP, dS, (long long)rows, cols);
  else return P2L_EUNSUP;
  return p2l_check_launch();
}

extern "C" int p2l_maxpool2_bwd_amax_slots(int H, int W, int C) {
  if (C % 4 || (H & 1) || (W & 1)) return 0;
  return (int)cdiv((size_t)(H / 2) * (W / 2) * (C / 4), 256) * 4;
}
extern "C" int p2l_maxpool2_bwd_amax(const float* y, int y_ld, const float* dyp,
                                     int dyp_ld, const float* add, int add_ld, float* dy,
                                     int dy_ld, int Bn, int H, int W, int C,
                                     int relu_mask, float* amax_out, void* stream) {
  if (C % 4 || (H & 1) || (W & 1) || Bn < 1 || Bn > 65535) return P2L_EINVAL;
  const size_t per_image = (size_t)(H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(cdiv(per_image, 256), Bn), dim3(256), 0,
                     ST(stream), y, y_ld, dyp, dyp_ld, add, add_ld, dy, dy_ld, Bn, H,
                     W, C, relu_mask, amax_out);
  return p2l_check_launch();
}
extern "C" int p2l_maxpool2_bwd(const float* y, int y_ld, const float* dyp,
                                int dyp_ld, const float* add, int add_ld, float* dy,
                                int dy_ld, int Bn, int H, int W, int C,
                                int relu_mask, void* stream) {
  return p2l_maxpool2_bwd_amax(y, y_ld, dyp, dyp_ld, add, add_ld, dy, dy_ld, Bn, H, W, C, relu_mask,
                               nullptr, stream);
}
extern "C" int p2l_relu_mask(const float* y, int y_ld, const float* g, int g_ld,
                             float* dy, int dy_ld, int64_t P, int C, void* stream) {
  if (C % 4) return P2L_EINVAL;
  hipLaunchKernelGGL(relu_mask_kernel, dim3(cdiv((size_t)P * (C / 4), 256)), dim3(256),
                     0, ST(stream), y, y_ld, g, g_ld, dy, dy_ld, (long long)P, C);
  return p2l_check_launch();
}

extern "C" int p2l_nchw3_to_nhwc16(const float* src, float* dst, int Bn, int H, int W,
                                   void* stream) {
  hipLaunchKernelGGL(nchw3_to_nhwc16_kernel, dim3(cdiv((size_t)Bn * H * W, 256)),
                     dim3(256), 0, ST(stream), src, dst, Bn, H * W);
  return p2l_check_launch();
}
extern "C" int p2l_nhwc16_to_nchw3(const float* src, float* dst, int Bn, int H, int W,
                                   void* stream) {
  hipLaunchKernelGGL(nhwc16_to_nchw3_kernel, dim3(cdiv((size_t)Bn * H * W, 256)),
                     dim3(256), 0, ST(stream), src, dst, Bn, H * W);
  return p2l_check_launch();
}
extern "C" int p2l_tanh_bwd16(const float* img, float* dimg, int64_t P, void* stream) {
  hipLaunchKernelGGL(tanh_bwd16_kernel, dim3(cdiv(P, 256)), dim3(256), 0, ST(stream),
                     img, dimg, (long long)P);
  return p2l_check_launch();
}

extern "C" int p2l_weight_sum(const float* weight, const float* loss_mask, float* wsum,
                              int Bn, int HW3, void* stream) {
  hipLaunchKernelGGL(weight_sum_kernel, dim3(Bn), dim3(1024), 0, ST(stream), weight,
                     loss_mask, wsum, HW3);
  return p2l_check_launch();
}
extern "C" int p2l_weight_map(const float* weight, const float* loss_mask, float* wsrc,
                              int Bn, int H, int W, void* stream) {
  hipLaunchKernelGGL(weight_map_kernel, dim3(cdiv(H * W, 256), Bn), dim3(256), 0,
                     ST(stream), weight, loss_mask, wsrc, H * W);
  return p2l_check_launch();
}
extern "C" int p2l_l1_loss_nblk(int H, int W) { return cdiv(H * W, 256); }
extern "C" int p2l_l1_loss_fwd(const float* img16, const float* target,
                               const float* weight, const float* loss_mask,
                               const float* wsum, float* loss, float* partial, int Bn,
                               int H, int W, void* stream) {
  const int nblk = cdiv(H * W, 256);
  hipLaunchKernelGGL(l1_fwd_kernel, dim3(nblk, Bn), dim3(256), 0, ST(stream), img16,
                     target, weight, loss_mask, partial, H * W);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(Bn), dim3(256), 0, ST(stream), partial,
                     loss, nblk, 1.f, wsum, 0);
  return p2l_check_launch();
}
extern "C" int p2l_l1_loss_bwd(const float* img16, const float* target,
                               const float* weight, const float* loss_mask,
                               const float* wsum, const float* gscale, float* dimg16,
                               int Bn, int H, int W, int accumulate, void* stream) {
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(cdiv(H * W, 256), Bn), dim3(256), 0,
                     ST(stream), img16, target, weight, loss_mask, wsum, gscale,
                     dimg16, H * W, accumulate);
  return p2l_check_launch();
}

#define P2L_LPIPS_DISPATCH(CALL)        \
  switch (C) {                          \
    case 64: CALL(64); break;           \
    case 128: CALL(128); break;         \
    case 192: CALL(192); break;         \
    case 256: CALL(256); break;         \
    case 384: CALL(384); break;         \
    case 512: CALL(512); break;         \
    default: return P2L_EUNSUP;         \
  }

extern "C" int p2l_lpips_normalize(const float* f, float* nf, int64_t P, int C,
                                   void* stream) {
#define CALL(CC)                                                                   \
  hipLaunchKernelGGL(lpips_normalize_kernel<CC>,                                   \
                     dim3(cdiv(P, 4 * LpipsCfg<CC>::PPW)), dim3(256), 0, ST(stream), \
                     f, nf, (long long)P)
  P2L_LPIPS_DISPATCH(CALL)
#undef CALL
  return p2l_check_launch();
}

extern "C" int p2l_lpips_tap_nblk(int P, int C) {
  const int lp = lpips_lp(C / 4);
  return cdiv(P, 4 * (64 / lp));
}

extern "C" int p2l_lpips_tap_fwd(const float* f, const float* nft,
                                 int64_t nft_bstride, const float* lin,
                                 const float* wt, int64_t wt_bstride,
                                 float* loss_partial, int Bn, int P, int C,
                                 void* stream) {
  LpipsK k{};
  k.f = f; k.nft = nft; k.lin = lin; k.wt = wt; k.gscale = nullptr;
  k.out = loss_partial; k.nft_bstride = nft_bstride; k.wt_bstride = wt_bstride;
  k.Bn = Bn; k.P = P; k.nblk = p2l_lpips_tap_nblk(P, C);
#define CALL(CC)                                                                \
  hipLaunchKernelGGL((lpips_tap_kernel<CC, false>), dim3(k.nblk, Bn), dim3(256), 0, \
                     ST(stream), k)
  P2L_LPIPS_DISPATCH(CALL)
#undef CALL
  return p2l_check_launch();
}
extern "C" int p2l_lpips_tap_bwd(const float* f, const float* nft,
                                 int64_t nft_bstride, const float* lin,
                                 const float* wt, int64_t wt_bstride,
                                 const float* gscale, float* df, int Bn, int P, int C,
                                 void* stream) {
  LpipsK k{};
  k.f = f; k.nft = nft; k.lin = lin; k.wt = wt; k.gscale = gscale;
  k.out = df; k.nft_bstride = nft_bstride; k.wt_bstride = wt_bstride;
  k.Bn = Bn; k.P = P; k.nblk = p2l_lpips_tap_nblk(P, C);
#define CALL(CC)                                                               \
  hipLaunchKernelGGL((lpips_tap_kernel<CC, true>), dim3(k.nblk, Bn), dim3(256), 0, \
                     ST(stream), k)
  P2L_LPIPS_DISPATCH(CALL)
#undef CALL
  return p2l_check_launch();
}

// tap backward + 2x2 max-pool backward + ReLU mask in one pass (see lpips_tap_kernel POOL)
extern "C" int p2l_lpips_tap_pool_bwd(const float* f, const float* nft, int64_t nft_bstride,
                                      const float* lin, const float* wt, int64_t wt_bstride,
                                      const float* gscale, const float* dyp, float* df,
                                      float* amax_out, int Bn, int H, int W, int C, void* stream) {
  const int lp = lpips_lp(C / 4), half = 2 * (64 / lp);
  if (!f || !nft || !lin || !wt || !gscale || !dyp || !df) return P2L_EINVAL;
  if (Bn < 1 || H < 2 || (H & 1) || W < half || W % half) return P2L_EINVAL;
  LpipsK k{};
  k.f = f; k.nft = nft; k.lin = lin; k.wt = wt; k.gscale = gscale;
  k.out = df; k.nft_bstride = nft_bstride; k.wt_bstride = wt_bstride;
  k.Bn = Bn; k.P = H * W; k.nblk = p2l_lpips_tap_nblk(H * W, C);
  k.dyp = dyp; k.amax = amax_out; k.W = W;
#define CALL(CC)                                                                     \
  hipLaunchKernelGGL((lpips_tap_kernel<CC, true, true>), dim3(k.nblk, Bn), dim3(256), 0, \
                     ST(stream), k)
  P2L_LPIPS_DISPATCH(CALL)
#undef CALL
  return p2l_check_launch();
}

extern "C" int p2l_bilinear_adjoint(const float* wsrc, float* wt, int Bn, int H, int W,
                                    int h, int w, void* stream) {
  if (h < 1 || w < 1 || h > H || w > W) return P2L_EINVAL;
  hipLaunchKernelGGL(bilinear_adjoint_kernel, dim3(cdiv(h * w, 256), Bn), dim3(256), 0,
                     ST(stream), wsrc, wt, H, W, h, w);
  return p2l_check_launch();
}

extern "C" int p2l_reduce_rows(const float* partial, float* out, int Bn, int n,
                               float scale, const float* div, int accumulate,
                               void* stream) {
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(Bn), dim3(256), 0, ST(stream), partial,
                     out, n, scale, div, accumulate);
  return p2l_check_launch();
}

extern "C" int p2l_affine_grid_sample(const float* src, const float* theta, float* dst, int Bn,
                                      int C, int H, int W, void* stream) {
  if (!src || !theta || !dst || Bn < 1) return P2L_EINVAL;
  hipLaunchKernelGGL(affine_grid_sample_kernel, dim3(cdiv((size_t)Bn * H * W, 256)), dim3(256),
                     0, ST(stream), src, theta, dst, Bn, C, H, W);
  return p2l_check_launch();
}

extern "C" size_t p2l_affine_grid_sample_bwd_ws_bytes(int Bn, int H, int W) {
  return (size_t)Bn * cdiv(H * W, 256) * 6 * sizeof(float);
}
extern "C" int p2l_affine_grid_sample_bwd(const float* src, const float* theta, const float* dout,
                                          float* dsrc, float* dtheta, int Bn, int C, int H, int W,
                                          void* workspace, size_t ws_bytes, void* stream) {
  if (!theta || !dout || Bn < 1 || (!dsrc && !dtheta)) return P2L_EINVAL;
  if (dsrc)
    hipLaunchKernelGGL(affine_grid_sample_bwd_src_kernel, dim3(cdiv((size_t)Bn * H * W, 256)), dim3(256),
                       0, ST(stream), dout, theta, dsrc, Bn, C, H, W);
  if (dtheta) {
    if (!src) return P2L_EINVAL;
    const int nblk = cdiv(H * W, 256);
    if (!workspace || ws_bytes < p2l_affine_grid_sample_bwd_ws_bytes(Bn, H, W)) return P2L_EWS;
    hipLaunchKernelGGL(affine_grid_sample_bwd_theta_kernel, dim3(nblk, Bn), dim3(256), 0, ST(stream),
                       src, theta, dout, (float*)workspace, C, H, W);
    hipLaunchKernelGGL(affine_grid_sample_bwd_theta_finish, dim3(cdiv(Bn * 6, 64)), dim3(64), 0,
                       ST(stream), (const float*)workspace, dtheta, Bn, nblk);
  }
  return p2l_check_launch();
}

extern "C" int p2l_adam_step(float* p, const float* g, float* m, float* v, int64_t n,
                             float lr, float beta1, float beta2, float eps,
                             int step_count, void* stream) {
  if (!p || !g || !m || !v || step_count < 1) return P2L_EINVAL;
  // same host-side double arithmetic as torch.optim.adam._single_tensor_adam
  const double bc1 = 1.0 - pow((double)beta1, (double)step_count);
  const double bc2 = 1.0 - pow((double)beta2, (double)step_count);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), p, g, m,
                     v, (long long)n, step_size, beta1, beta2, eps, bc2_sqrt);
  return p2l_check_launch();
}
extern "C" int p2l_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n,
                                 float lr, float beta1, float beta2, float eps,
                                 int32_t* step_counters, int n_counters, void* stream) {
  if (!p || !g || !m || !v || !step_counters || n_counters < 1) return P2L_EINVAL;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), p, g, m,
                     v, (long long)n, lr, beta1, beta2, eps, (const int*)step_counters);
  hipLaunchKernelGGL(counters_advance_kernel, dim3(cdiv(n_counters, 256)), dim3(256), 0,
                     ST(stream), (int*)step_counters, n_counters);
  return p2l_check_launch();
}
extern "C" int p2l_clamp(float* p, int64_t n, float lo, float hi, void* stream) {
  hipLaunchKernelGGL(clamp_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST(stream), p,
                     (long long)n, lo, hi);
  return p2l_check_launch();
}

extern "C" int p2l_mfma_probe(const float* A, const float* B, float* C, int K,
                              void* stream) {
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, ST(stream), A, B, C, K);
  return p2l_check_launch();
}
