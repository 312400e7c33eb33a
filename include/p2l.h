/*
 * p2l.h — C ABI of libp2l_hip.so: the MI355X (gfx950) native hot path of the
 * pix2latent latent-inversion inner loop.
 *
 * Drop-in boundary.  The reference has no FFI of its own (it is
 * pure Python on top of torch / pytorch_pretrained_biggan / lpips); the entry
 * points below are what a maintainer would bind in place of
 *
 *   model(**input_args)                  pix2latent/optimizer/closure.py:51
 *     -> BigGAN.forward                  pix2latent/model/biggan.py:50-58
 *   loss_fn(out, **target_args)          pix2latent/optimizer/closure.py:55
 *     -> ProjectionLoss.__call__         pix2latent/loss_functions.py:97-100
 *     -> ReconstructionLoss.__call__     pix2latent/loss_functions.py:117-124
 *     -> PerceptualLoss.__call__         pix2latent/loss_functions.py:140-148
 *   loss.mean().backward()               pix2latent/optimizer/closure.py:58
 *   opt.step (torch.optim.Adam)          pix2latent/optimizer/closure.py:65
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - activations are NHWC fp32, channel pitch ("ld") given explicitly;
 *   - nothing here allocates, frees or synchronises; every kernel is enqueued on
 *     the hipStream_t passed in (void* so that C callers need no HIP headers);
 *   - return value: 0 on success, negative P2L_E* on error (p2l_strerror()).
 */
#ifndef P2L_H_
#define P2L_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2L_OK 0
#define P2L_EINVAL (-1)   /* bad shape / alignment / null pointer            */
#define P2L_ELAUNCH (-2)  /* hipLaunchKernel reported an error               */
#define P2L_EWS (-3)      /* workspace too small                             */
#define P2L_EUNSUP (-4)   /* combination not implemented                     */

int p2l_version(void);
const char* p2l_strerror(int rc);
/* last hipError_t seen by a failed launch on this thread (0 if none) */
int p2l_last_hip_error(void);

/* ------------------------------------------------------------------------- */
/* Convolution (3x3 pad 1 / 1x1), implicit GEMM on v_mfma_f32_32x32x2_f32.   */
/* Replaces every nn.Conv2d on the path (HF BigGAN-deep GenBlock convs,      */
/* SelfAttn 1x1s, VGG16 features) and its input-gradient.                    */
/* ------------------------------------------------------------------------- */

enum { P2L_ACT_NONE = 0, P2L_ACT_RELU = 1, P2L_ACT_TANH = 2,
       P2L_ACT_LRELU_SQRT2 = 3 /* leaky_relu(0.2) * sqrt(2), StyleGAN2 FusedLeakyReLU */ };
enum { P2L_POOL_NONE = 0, P2L_POOL_MAX = 1, P2L_POOL_SUM = 2 };
enum { P2L_PRO_NONE = 0, P2L_PRO_AFFINE_RELU = 1, P2L_PRO_AFFINE = 2 };

typedef struct P2LConv {
  /* geometry: B images, OUTPUT H x W; H, W powers of two >= 4                */
  int32_t B, H, W;
  int32_t Cin;      /* multiple of 16 (zero-pad weights + input otherwise)    */
  int32_t Cout;     /* multiple of 32                                         */
  int32_t taps;     /* 1 (1x1) or 9 (3x3, zero pad 1)                         */
  int32_t ups;      /* 1: x is [B,H/2,W/2,*]; nearest x2 applied on the fly   */
                    /* 2: same conv in SUB-PIXEL form (4 phase 2x2 convs on   */
                    /*    the low-res input, weights from                     */
                    /*    p2l_pack_conv_weight_subpix; 2.25x fewer FLOPs)     */
                    /* 3: input-gradient of such a conv: x is the high-res    */
                    /*    dY [B,H,W,Cin], result is low-res [B,H/2,W/2,Cout]  */
  int32_t x_ld;     /* floats per input pixel (>= Cin)                        */
  /* prologue on the input operand (applied before zero padding):             */
  /*   AFFINE_RELU: a = max(x*s[b,c] + t[b,c], 0)   (CBN+ReLU, BN+ReLU)       */
  /*   AFFINE     : a = x*s[b,c] + t[b,c]           (LPIPS scaling layer)     */
  int32_t pro;
  int32_t pro_bstride; /* floats between samples in pro_s/pro_t (0 = shared)  */
  /* epilogue: v = alpha*acc + bias[n] + res ; act ; mask ; store ; pool      */
  float alpha;
  int32_t act;
  int32_t pool;     /* pooled output written to yp ([B,H/2,W/2,*], yp_ld)     */
  int32_t y_ld, yp_ld;
  int32_t n_store;  /* store only channels n < n_store (<= Cout)              */
  int32_t res_ld;   /* residual pixel pitch                                   */
  int32_t res_ups;  /* 1: residual is [B,H/2,W/2,*] read with nearest x2      */
  int32_t mask_ld;  /* mask pixel pitch (mask > 0 keeps the value)            */
  int32_t splitk;   /* >=1; >1 needs workspace of splitk*B*H*W*Cout floats    */
  int32_t ext;      /* ups 2/3 only: 1 = stride-2 TRANSPOSED conv geometry: the   */
                    /* high-res buffer is [B,H+2,W+2,*] (rows/cols 0..H real,     */
                    /* H+1 zero) and the low-res grid of ups=2 has H/2+1 points   */
  int32_t wfmt;     /* P2L_WFMT_*: format of w_packed (3x3 / sub-pixel only)       */
  int32_t form;     /* P2L_FORM_* bits: which kernel form may run THIS launch (0 =  */
                    /* automatic).  Per call, not a library-wide switch: the library  */
                    /* holds no mutable state that a launch reads                     */
  double algo_flops; /* algorithmic FLOPs of this launch for the profiler;    */
                     /* 0 = 2*B*H*W*Cin*Cout*taps (set it when Cin/Cout are   */
                     /* zero-padded, e.g. the 3-channel image convs)          */
  int64_t w_floats;  /* length of the buffer behind `w` in floats (version 101). */
                     /* The packed formats are not tagged on the device: a launch */
                     /* finds the image it reads at a fixed offset that follows   */
                     /* from (taps, ups, Cin, Cout, wfmt), so a buffer packed for */
                     /* another format (a _subpix_bf3 buffer handed to a          */
                     /* P2L_WFMT_BF16X3W launch) would be read past its end.      */
                     /* > 0: the call returns P2L_EINVAL when the buffer is       */
                     /* shorter than p2l_packed_weight_floats /                   */
                     /* p2l_packed_subpix_weight_floats(.., wfmt) of THIS launch; */
                     /* 0 = not stated, nothing is checked.                       */
} P2LConv;

/* Weight layout expected in `w`: [taps][Cin/KC][Cout][KC] fp32 with KC = 16 for
 * 3x3 and KC = 32 for 1x1 (Cin must be a multiple of KC): for tap t, input
 * channel chunk q and output channel n, KC consecutive input channels.
 * p2l_pack_conv_weight builds it (and the input-gradient copy) from OIHW. */
size_t p2l_conv_workspace_bytes(const P2LConv* d);
/* K slices for this layer: a function of the LAYER SHAPE (H, W, Cin, Cout, taps, ups, format) only,
 * never of d->B -- the slices are summed in a fixed order that follows their count, so a count that
 * followed the batch would make a candidate's low bits depend on who shares its launch (it did for
 * the 4^2 ... 16^2 layers until round 5).  With the suggested count a candidate's result is
 * bit-identical for every batch size and position in the batch.
 * Sub-pixel launches: only the input-gradient form of a stride-2 transposed conv (ups = 3, ext = 1: StyleGAN2's
 * up convs) in the fp16 x 2 kernel takes slices (its 8^2 ... 32^2 layers; round 6) -- the finish kernel then runs
 * on the low-resolution grid, p2l_conv_arb_nblk_ws counts ITS quads, and such a launch leaves no maxima
 * (p2l_conv_amax_slots = 0).  ups = 2 and the nearest-upsample gradient (ups = 3, ext = 0) never split.  */
int p2l_conv_suggest_splitk(const P2LConv* d);
int p2l_conv_fwd(const P2LConv* d, const float* x, const float* w,
                 const float* bias, const float* pro_s, const float* pro_t,
                 const float* res, const float* mask, float* y, float* yp,
                 void* workspace, size_t ws_bytes, void* stream);

/* Input-gradient conv fused with the backward of a = max(x*s+t,0) of its consumer
 * (replaces p2l_conv_fwd + p2l_affine_relu_bwd for the GenBlock convs):
 *   da = conv(dy, w)  [2x2-summed when d->pool == P2L_POOL_SUM: nearest-x2 backward]
 *   g = (x*s+t>0) ? da : 0 ; dx = g*s + shortcut ; ds[b,c] = sum g*x ; dt[b,c] = sum g
 * Usable when p2l_conv_arb_fusable(d) (one image per tile, no split-K); `partial`
 * needs 2*B*p2l_conv_arb_nblk(d)*Cout floats. */
/* Per-image maxima of |tensor|, handed from the conv that WRITES a tensor to the conv that reads
 * it, so that an fp16 x 2 Winograd launch (P2L_WFMT_BF16X3W) does not need its own pass over the
 * input.  Producer: `out` (and `outp` for the pooled output) receive one partial maximum per wave of every block,
 * [B][p2l_conv_amax_slots(d)] floats each; a launch that cannot produce them (slots == 0) ignores
 * the fields.  Consumer: `in` = the partial maxima of the tensor it reads as x ([B][in_n], the RAW
 * tensor: a fused prologue x*s+t is bounded by max|s| max|x| + max|t| inside the kernel); NULL =
 * the launch reduces max|x| itself.  The caller guarantees that nothing else wrote the tensor in
 * between.  All NULL / 0 = not used.
 * The bound of a fused prologue is loose when large |s| and large |x| sit in different channels
 * (heavy-tailed synthetic weights: x 50 - 160 on every CBN layer), and every factor of two of it is a bit of the
 * fp16 x 2 operand's range.  A producer that is told the affine its reader will fuse (`next_s`,
 * `next_t`: [B][next_bstride] per-channel vectors, bstride 0 = shared by the images) therefore
 * records the maxima of |y*next_s + next_t| in `out` instead (never in `outp`); the reader is then
 * given them with `in_applied` = 1 and takes max|x*s+t| from them as it stands. */
typedef struct P2LAmax {
  float* out; float* outp;
  const float* in; int32_t in_n;
  const float* next_s; const float* next_t; int32_t next_bstride;
  int32_t in_applied;
} P2LAmax;
typedef struct P2LArb {
  const float* x; int32_t x_ld;              /* pre-activation input of the forward conv */
  const float* s; const float* t; int32_t st_bstride;
  const float* skip; int32_t skip_ld, skip_C, skip_ups;   /* GenBlock shortcut gradient   */
  float* ds; float* dt; int32_t dsdt_bstride;
  float* partial;
  int32_t nomask;                            /* 1: plain scale backward g = da (StyleGAN2) */
  P2LAmax amax;                              /* maxima of dy (in) / of the written dx (out)  */
} P2LArb;
int p2l_conv_arb_fusable(const P2LConv* d);
int p2l_conv_arb_nblk(const P2LConv* d);
int p2l_conv_dgrad_arb(const P2LConv* d, const P2LArb* arb, const float* dy,
                       const float* w, float* dx, void* stream);
/* Split-K form (d->splitk > 1, small launches): the K slices go to `workspace`
 * (p2l_conv_workspace_bytes) and the deterministic finish kernel applies the activation
 * backward; `partial` then needs 2*B*p2l_conv_arb_nblk_ws(d)*Cout floats (one partial per
 * 2x2 quad).  Falls back to p2l_conv_dgrad_arb when d->splitk resolves to 1. */
int p2l_conv_arb_split_fusable(const P2LConv* d);
int p2l_conv_arb_nblk_ws(const P2LConv* d);
int p2l_conv_dgrad_arb_ws(const P2LConv* d, const P2LArb* arb, const float* dy, const float* w,
                          float* dx, void* workspace, size_t ws_bytes, void* stream);
int p2l_arb_finish(const float* partial, float* ds, float* dt, int Bn, int nblk, int C,
                   int out_bstride, void* stream);
/* Deferred form of the finish (per host thread): between _begin and _flush every
 * p2l_arb_finish - including the ones p2l_conv_dgrad_arb / p2l_affine_relu_bwd issue - is
 * only recorded, and _flush reduces all of them in ONE launch (the backward pass of a
 * generator has ~50 of them and needs ds/dt only at its end).  The caller must give every
 * deferred layer its own `partial` buffer and keep it untouched until _flush.  _cancel
 * drops what was recorded (error paths). */
void p2l_arb_defer_begin(void);
int p2l_arb_defer_flush(void* stream);
void p2l_arb_defer_cancel(void);

/* Per-launch timing of the conv kernel with HIP events recorded on the launch
 * stream (bench.py roofline leg).  begin() pre-creates the event pool; p2l_prof_totals()
 * synchronises, ends the session and returns totals per family: [0] = 3x3, [1] = 1x1. */
int p2l_prof_begin(int max_launches);
/* Totals per conv family, index 0 = 3x3 launches, 1 = 1x1 launches.  `size` is set by the CALLER to
 * sizeof(P2LProfTotals) as it was compiled; the library fills the members that fit (later versions of
 * the library only ever append members), so p2l_prof_totals serves every caller built against version
 * >= 101.  (Version 100 exported `p2l_prof_end`, first as (double[2], double[2], int32_t[2]) and then
 * with this struct under the SAME name: a caller of the array form would have had its doubles read as
 * `size`.  The name is retired -- a stale binding now fails to resolve instead of corrupting memory --
 * and p2l_version() says which side of the change a library is on.)                                 */
typedef struct P2LProfTotals {
  uint32_t size;           /* in: sizeof(P2LProfTotals)                                             */
  int32_t count[2];        /* timed launches                                                         */
  int32_t reserved0;
  double flops[2];         /* algorithmic FLOPs (9 taps on the output grid, un-padded channels)       */
  double ms[2];            /* event time of those launches (conv + its split-K finish)                */
  double bytes[2];         /* algorithmic bytes: every operand tensor once + the packed weights       */
  double exec_flops[2];    /* FLOPs the matrix pipe executed (padded channels; sub-pixel forms: 4     *
                            * phase-taps per output pixel; Winograd: 16 products per 2x2 quad)        */
  double mfma_flops[2];    /* 16-bit MFMA FLOPs issued: exec_flops x products per fp32 product (6 =   *
                            * bf16 x 3, 3 = fp16 x 2; a launch on the fp32 MFMA counts 16, its cost   *
                            * in 16-bit-MFMA time): / time / dense 16-bit peak = matrix-pipe busy      */
  double write_bytes[2];   /* the WRITTEN share of `bytes` (y / yp): PMC FETCH_SIZE and WRITE_SIZE     *
                            * can each be set against their own algorithmic count                     */
  /* version 101: the same totals per KERNEL family (P2L_PROF_FAM_*), so that a bench line can name its *
   * dominant kernel and be checked against that kernel's row of a rocprofv3 --kernel-trace --stats table */
  int32_t fam_count[8];
  double fam_ms[8], fam_flops[8], fam_mfma_flops[8], fam_bytes[8];
} P2LProfTotals;
enum { P2L_PROF_FAM_OTHER = 0,       /* exact-fp32 / bf16 x 3 direct kernels, 8x16 Winograd ...            */
       P2L_PROF_FAM_WINO_H2 = 1,     /* wino16s_conv_kernel<.., H2 = true, ..> (+ its split-K finish)        */
       P2L_PROF_FAM_DIRECT_H2 = 2,   /* conv_h2_kernel<9, ..> (+ split-K finish)                            */
       P2L_PROF_FAM_SUBPIX_H2 = 3,   /* conv_h2_kernel<4, ..>: sub-pixel forward / input gradient           */
       P2L_PROF_FAM_THIN = 4,        /* conv_thinin_kernel / conv_thinout_kernel (3-channel image convs)    */
       P2L_PROF_FAM_PW = 5,          /* pw_h2_kernel / pw_conv_kernel (1x1)                                  */
       P2L_PROF_FAM_DIRECT_H2R = 6 };/* conv_h2r_kernel: 64 -> 64 channels, weights resident in registers    */
int p2l_prof_totals(P2LProfTotals* out);
/* Sampling: every hipEventRecord pair costs the stream a ~5 us bubble (500 of them are 5 %
 * of a 26 ms step), so a caller that times a whole step loop can ask for only every
 * `period`-th conv launch to be timed: call p2l_prof_step(i, period) at the top of step i;
 * launch number n of that step is timed iff n % period == i % period, so after `period`
 * steps every launch of the step has been timed once.  Without the call every launch is
 * timed. */
int p2l_prof_step(int step, int period);
/* p2l_prof_totals also writes one line per timed launch to `path` (taps B H W Cin Cout ups pro arb
 * splitk flops bytes ms) when a path was given; NULL switches it off.  The profiler is the one
 * process-wide object of the library (the backward pass of a torch program runs on the autograd
 * engine's thread and must be timed too): opt-in, every access under a mutex. */
int p2l_prof_dump(const char* path);

/* Extra epilogue terms of the StyleGAN2 styled conv (p2l_conv_fwd_ex):
 *   v = acc * oscale[b][n] + noise_w * noise[b][pixel] + bias[n] ; act            */
typedef struct P2LConvExtra {
  const float* oscale;      /* [B][oscale_bstride] demodulation factors, or NULL  */
  int32_t oscale_bstride;
  const float* noise;       /* [B][H*W] per-pixel noise, or NULL                  */
  float noise_w;
  P2LAmax amax;             /* maxima of x (in) / of the written y, yp (out, outp) */
} P2LConvExtra;
/* partial maxima per image a launch of d writes into P2LAmax.out / outp (0: it writes none) */
int p2l_conv_amax_slots(const P2LConv* d);
int p2l_conv_fwd_ex(const P2LConv* d, const P2LConvExtra* ex, const float* x,
                    const float* w, const float* bias, const float* pro_s,
                    const float* pro_t, const float* res, const float* mask, float* y,
                    float* yp, void* workspace, size_t ws_bytes, void* stream);

/* w_oihw is [O][I][kh][kw].  transpose_flip=0 packs the conv I->O (K_pad >= I,
 * N_pad >= O, zero padded); transpose_flip=1 packs the conv that maps dY[O] to
 * dX[I] (taps mirrored, channels swapped; K_pad >= O, N_pad >= I). */
int p2l_pack_conv_weight(const float* w_oihw, int O, int I, int taps, int N_pad,
                         int K_pad, int transpose_flip, float* w_packed,
                         void* stream);

/* Sub-pixel weight packing (P2LConv.ups = 2 forward, 3 input-gradient with
 * transpose_flip=1): [16 slabs = phase*4 + tap][K_pad/16][N_pad][16].
 *   mode 0: 3x3 conv reading a nearest-x2 upsampled input (BigGAN GenBlock conv_1)
 *   mode 1: stride-2 transposed 3x3 conv (StyleGAN2 up-conv; use with P2LConv.ext=1) */
int p2l_pack_conv_weight_subpix(const float* w_oihw, int O, int I, int N_pad,
                                int K_pad, int transpose_flip, int mode,
                                float* w_packed, void* stream);

/* Weight formats of the 3x3 / sub-pixel conv kernels.
 *   P2L_WFMT_F32   : fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32 products).
 *   P2L_WFMT_BF16X3: fp32-EQUIVALENT arithmetic on the bf16 matrix pipe: each fp32 operand
 *     is split into three bf16 pieces (x = x1+x2+x3 to 2^-24 relative) and the product is
 *     accumulated in fp32 from six v_mfma_f32_32x32x16_bf16 cross terms (dropped terms
 *     <= 2^-23 relative).  gfx950 runs bf16 MFMA at 16x the fp32-MFMA rate, so this costs
 *     2.67x fewer matrix cycles.  Weights are pre-split by the *_bf3 pack functions into
 *     buffers of 1.5 x the fp32 packed size; activations are split inside the kernel.
 *     Packed layout (an image of the kernel's LDS weight tile, streamed by LDS-direct DMA):
 *       [K_pad/16 chunks][N_pad/32 tiles][slabs: 9 taps | 16 = phase*4+tap][32 rows][96 B],
 *       row = [x1 k0-7 | x1 k8-15 | x2 k0-7 | x2 k8-15 | x3 k0-7 | x3 k8-15] as bf16, the
 *       16-byte chunk index XOR-ed with bit 3 of the row.  N_pad must equal P2LConv.Cout.
 *   P2L_WFMT_BF16X3W: the same arithmetic and the same packed image, FOLLOWED by the
 *     Winograd F(2x2,3x3) transform-domain image of the weights (G g G^T in fp64, rounded to
 *     fp32, split into the same three bf16 pieces; [K_pad/16][16 frequencies][N_pad/32]
 *     [3 pieces][64 lanes] x 16 B = MFMA B-fragment order).  p2l_conv_fwd / p2l_conv_dgrad_arb
 *     then run stride-1 3x3 layers made of whole 8x16-pixel x 64-channel blocks (at least 64 per
 *     image: a function of the layer shape only, never of the batch) in the
 *     Winograd form (2.25x fewer matrix products, csrc/p2l_wino.hip) and everything else on the
 *     direct kernel.  Results agree with P2L_WFMT_BF16X3 to fp32 rounding.
 *     The buffer ends with a SECOND transform-domain image in the fp16 x 2 arithmetic of the
 *     16x16-pixel Winograd kernel: every operand is scaled by a power of two (weights: per layer,
 *     at pack time; activations: per image, from a max-|x| pass in front of the launch) so that
 *     its largest value sits at 2^13..2^15, and split into TWO round-to-nearest fp16 pieces
 *     (x = h + m to 1 ulp of fp32; elements more than 2^15 below the maximum of their image get a
 *     denormal second piece and lose relative precision -- 2^-37 of the maximum in absolute terms); the product is accumulated in
 *     fp32 from THREE v_mfma_f32_32x32x16_f16 cross terms (h h, h m, m h) and un-scaled exactly in
 *     the epilogue: half the matrix work and a third of the split instructions of bf16 x 3.
 *     [K_pad/16][16][N_pad/32][2 pieces][64 lanes] x 16 B, then 4 floats (the bits of max |w|).
 *     Needs 256 B of workspace per image (p2l_conv_workspace_bytes counts them); without it, or
 *     on P2L_FORM_WINO_BF3, the launch keeps the bf16 x 3 arithmetic.
 *     p2l_packed_weight_floats() gives the buffer size of any format.
 *   P2L_WFMT_PW (1x1 convs only): the fp32 layout of p2l_pack_conv_weight FOLLOWED by the
 *     bf16x3 image [K_pad/16][N_pad/32][32 rows][96 B] (p2l_pack_conv_weight_pw).  Layers of at
 *     least 32x32 pixels with Cin, Cout multiples of 64 then run in the bf16x3 arithmetic
 *     (csrc/p2l_pw.hip), the rest on the exact-fp32 kernel.
 *   Model structs (P2LBigGAN.wfmt ...) hold the 3x3 format in bits 0-3 and P2L_WFMT_FLAG_PW when
 *   their 1x1 weight buffers are P2L_WFMT_PW buffers. */
enum { P2L_WFMT_F32 = 0, P2L_WFMT_BF16X3 = 1, P2L_WFMT_BF16X3W = 2, P2L_WFMT_PW = 3,
       P2L_WFMT_BF16X3T = 4, P2L_WFMT_FLAG_PW = 0x10, P2L_WFMT_FLAG_THIN = 0x20,
       P2L_WFMT_FLAG_ATTN_GEMM = 0x40,/* model descriptors: self-attention as GEMM + softmax   */
                                      /* (the attention matrix is stored) instead of the fused  */
                                      /* kernels of csrc/p2l_attn.hip                          */
       P2L_WFMT_FLAG_NO_AMAX = 0x80   /* model descriptors: no maxima handed between the convs  */
                                      /* of the plan (P2LAmax): every fp16 x 2 Winograd launch  */
                                      /* runs its own max-|x| pass, the 1x1 convs stay bf16 x 3 */ };
/* P2L_WFMT_BF16X3T (3x3 convs with THREE real channels on one side, padded to N_pad == 32 or
 * K_pad == 16: conv_to_rgb, the first VGG conv and their input gradients): the BF16X3 image
 * followed by the image of p2l_thin.hip's kernels (27 tap-channel products as one K dimension
 * | a pointwise product onto 27 columns + 9-tap gather).  P2L_WFMT_FLAG_THIN in a MODEL
 * descriptor's wfmt says its image-end weights were packed this way. */
int p2l_pack_conv_weight_bf3t(const float* w_oihw, int O, int I, int N_pad, int K_pad,
                              int transpose_flip, float* w_packed, void* stream);
int p2l_pack_conv_weight_pw(const float* w_oihw, int O, int I, int N_pad, int K_pad,
                            int transpose_flip, float* w_packed, void* stream);
size_t p2l_packed_weight_floats(int taps, int N_pad, int K_pad, int wfmt);
/* P2LConv.form.  Which P2L_WFMT_BF16X3W launches take the Winograd form by default: those whose
 * layer shape gives at least 64 blocks of 8x16 pixels x 64 channels per image (never a function
 * of the batch: a candidate's result must not depend on who shares its launch).  The Winograd
 * form runs 16x16-pixel blocks (8 waves, hand-scheduled, csrc/p2l_wino.hip) when H and W are
 * multiples of 16 and 8x16-pixel blocks (4 waves) otherwise; both give bit-identical results. */
enum {
  P2L_FORM_AUTO = 0,
  P2L_FORM_NO_WINO = 1,      /* direct kernel even where the Winograd form is eligible         */
  P2L_FORM_WINO_ANY = 2,     /* Winograd form for every eligible shape, small grids too (tests) */
  P2L_FORM_WINO_8X16 = 4,    /* the 8x16-pixel Winograd kernel even where 16x16 fits (tests)   */
  P2L_FORM_NO_PW = 8,        /* P2L_WFMT_PW weights, but the exact-fp32 1x1 kernel             */
  P2L_FORM_NO_THIN = 16,     /* P2L_WFMT_BF16X3T weights, but the generic 3x3 kernel           */
  P2L_FORM_WINO_BF3 = 32,    /* bf16 x 3 arithmetic, not fp16 x 2: 16x16 Winograd and 1x1 kernels */
  /* block shape of the fp16 x 2 Winograd kernel.  Default: 16x16-pixel blocks (8 waves), 8x16-pixel
   * blocks (4 waves) while the launch has <= 128 blocks of 16x16 -- chosen from the grid, i.e. from the
   * batch, which is allowed because the two shapes give bit-identical results and maxima slots.     */
  P2L_FORM_WINO_H2_8X16 = 64,   /* always the 4-wave block (tests)                                   */
  P2L_FORM_WINO_H2_16X16 = 128, /* always the 8-wave block (tests, A/B)                              */
  /* 64 -> 64 channel 3x3 layers of whole 8x16-pixel tiles run with their weights resident in registers  *
   * (csrc/p2l_h2r.hip: persistent blocks, one per CU); the results are bit-identical to the chunked      *
   * direct kernel, which this bit keeps (tests, A/B)                                                     */
  P2L_FORM_NO_H2R = 256,
  P2L_FORM_H2R_SEQ_EPI = 512,   /* ... that kernel with the shared epilogue item between two tiles: instead of the  *
                                 * forward-style epilogue pipelined under the next tile's stream, and for the      *
                                 * launches (fused activation backward, residual ...) that otherwise stay on the   *
                                 * chunked kernel (tests, A/B)                                                     */
  /* P2LConv.ext = 1 declares the weights of a stride-2 TRANSPOSED 3x3 conv (p2l_pack_conv_weight_subpix* mode 1):
   * 7 of the 16 (phase, window tap) slabs are zero by construction, and the fp16 x 2 sub-pixel kernel skips their
   * products (round 6: 16 -> 9 matrix products per 2x2 output quad, bit-identical).  This bit multiplies them
   * anyway (tests, A/B).                                                                                        */
  P2L_FORM_NO_SP_SKIP = 1024,
  /* the fp16 x 2 sub-pixel FORWARD kernel with one output phase per block (four blocks stage the same patch) instead *
   * of two on one staged patch (round 6, bit-identical; tests, A/B)                                              */
  P2L_FORM_NO_SP_PAIR = 2048,
  P2L_FORM_SP_PAIR = 4096       /* ... two phases per block for every sub-pixel forward launch (default: the        *
                                 * transposed convs with 64-channel tiles only)                                   */
};
/* K slices of a small-grid Winograd layer: 3x3 layers with 16..63 blocks of 8x16 pixels x 64
 * channels per image (H, W multiples of 16) run the 16x16 Winograd kernel with the input channels
 * cut into this many slices (2 or 4; >= 8 chunks of 16 channels each) and the deterministic
 * split-K finish -- IF the caller passes exactly this number as P2LConv.splitk (and the
 * workspace p2l_conv_workspace_bytes asks for); p2l_conv_suggest_splitk returns it for such a
 * shape.  A function of the layer shape only, never of the batch.  1 = not such a layer. */
int p2l_wino_split_factor(int H, int W, int Cin, int Cout);
int p2l_pack_conv_weight_bf3w(const float* w_oihw, int O, int I, int taps, int N_pad,
                              int K_pad, int transpose_flip, float* w_packed, void* stream);
int p2l_pack_conv_weight_bf3(const float* w_oihw, int O, int I, int taps, int N_pad,
                             int K_pad, int transpose_flip, float* w_packed, void* stream);
/* Sub-pixel buffer of format P2L_WFMT_BF16X3 ONLY (16 phase-tap slabs, bf16 x 3 image, nothing behind
 * it).  A P2L_WFMT_BF16X3W launch reads the fp16 x 2 image BEHIND the bf16 x 3 one at a fixed offset: its
 * sub-pixel buffers must come from p2l_pack_conv_weight_subpix_h2 (sized by
 * p2l_packed_subpix_weight_floats(.., P2L_WFMT_BF16X3W)); handing a _subpix_bf3 buffer to such a launch
 * reads past its end -- the formats are not tagged on the device, the format field of the call is the
 * contract. */
int p2l_pack_conv_weight_subpix_bf3(const float* w_oihw, int O, int I, int N_pad, int K_pad,
                                    int transpose_flip, int mode, float* w_packed,
                                    void* stream);
/* Sub-pixel weights of a P2L_WFMT_BF16X3W model: the image above FOLLOWED by the fp16 x 2 image of
 * the same 16 phase-tap slabs (4 bytes per weight + 4 tail floats; p2l_h2.hip): every 3x3 / sub-pixel
 * launch of such a model that the Winograd kernel does not take runs in the fp16 x 2 arithmetic of
 * section "fp16 x 2" above (3 MFMA products per fp32 product on per-image / per-layer power-of-two
 * scaled operands) when the caller passes the workspace p2l_conv_workspace_bytes asks for; plain
 * 3x3 buffers (p2l_pack_conv_weight_bf3w) carry their fp16 x 2 direct image behind the Winograd
 * images.  p2l_packed_subpix_weight_floats: floats of a sub-pixel buffer of format wfmt. */
size_t p2l_packed_subpix_weight_floats(int N_pad, int K_pad, int wfmt);
int p2l_pack_conv_weight_subpix_h2(const float* w_oihw, int O, int I, int N_pad, int K_pad,
                                   int transpose_flip, int mode, float* w_packed, void* stream);

/* ------------------------------------------------------------------------- */
/* Batched GEMM fp32 (attention bmm's and their gradients).                  */
/*   C[b] = alpha * op(A[b]) * op(B[b])  (+ C[b] if accumulate)              */
/*   a_kmajor=0: A is [M][lda] with K contiguous; 1: A is [K][lda], M contig */
/*   b_kmajor=0: B is [N][ldb] with K contiguous; 1: B is [K][ldb], N contig */
/*   M % 128 == 0, N % 32 == 0, K % 16 == 0.                                 */
/* ------------------------------------------------------------------------- */
typedef struct P2LGemm {
  int32_t batch, M, N, K;
  int32_t lda, ldb, ldc;
  int64_t stride_a, stride_b, stride_c; /* floats between batches            */
  int32_t a_kmajor, b_kmajor;
  float alpha;
  int32_t accumulate;
} P2LGemm;
int p2l_gemm(const P2LGemm* d, const float* A, const float* B, float* C,
             void* stream);
/* same, splitting K over extra blocks when the product is deep (K >= 1024) and has fewer
 * than 192 output tiles; the K slices are summed in a fixed order (no atomics).  `ws` needs
 * p2l_gemm_ws_bytes(d) bytes (0 = this product is never split); without it the call is
 * p2l_gemm. */
size_t p2l_gemm_ws_bytes(const P2LGemm* d);
int p2l_gemm_ws(const P2LGemm* d, const float* A, const float* B, float* C, void* ws,
                size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Fused self-attention (p2l_attn.hip): out[b][i][:] = sum_j softmax_j(q_i . k_j) v[j][:]   */
/* for q [B][Nq][d], k [B][Nk][d], v [B][Nk][dv]; the Nq x Nk matrix is never stored.       */
/* Replaces SelfAttn.forward of pytorch_pretrained_biggan (reached from                      */
/* pix2latent/model/biggan.py:58); bf16x3 arithmetic, fp32 accumulate.                       */
/* Shapes: d == 64, dv == 256, Nq % 128 == 0, Nk % 128 == 0 (p2l_attn_supported).            */
/*   lse[b][i] = log sum_j exp(q_i . k_j): the row statistic the backward pass recomputes    */
/*   the probabilities from.                                                                 */
/*   p2l_attn_bwd_dv: dv[b][j][:] = sum_i exp(q_i . k_j - lse_i) dout[b][i][:]               */
/* ------------------------------------------------------------------------- */
typedef struct P2LAttn {
  int32_t B, Nq, Nk, d, dv;
} P2LAttn;
int p2l_attn_supported(const P2LAttn* d);
size_t p2l_attn_fwd_ws_bytes(const P2LAttn* d);
int p2l_attn_fwd(const P2LAttn* d, const float* q, const float* k, const float* v, float* out,
                 float* lse, void* ws, size_t ws_bytes, void* stream);
size_t p2l_attn_bwd_dv_ws_bytes(const P2LAttn* d);
int p2l_attn_bwd_dv(const P2LAttn* d, const float* q, const float* k, const float* dout,
                    const float* lse, float* dv, void* ws, size_t ws_bytes, void* stream);
/* dq, dk of the same attention (Nq % 256 == 0).  P and dP = dout v^T are recomputed tile by tile
 * and never stored; dS^T[b][j][i] = P_ij (dP_ij - dout_i . out_i) is written once into `dst`
 * (B*Nk*Nq floats, scratch) and applied to k and q. */
size_t p2l_attn_bwd_qk_ws_bytes(const P2LAttn* d);
int p2l_attn_bwd_qk(const P2LAttn* d, const float* q, const float* k, const float* v,
                    const float* out, const float* dout, const float* lse, float* dst, float* dq,
                    float* dk, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Small dense layers on the conditioning vector (gen_z, CBN gain/bias).     */
/*   y[b][n] = sum_k x[b][k] * W[k][n] + bias[n]      (W is [K][N])          */
/*   dx[b][k] = sum_n dy[b][n] * W[k][n]                                     */
/* ------------------------------------------------------------------------- */
int p2l_linear_fwd(const float* x, const float* W, const float* bias, float* y,
                   int Bn, int K, int N, void* stream);
int p2l_linear_bwd(const float* dy, const float* W, float* dx, int Bn, int K,
                   int N, int accumulate, void* stream);
/* same with an explicit row pitch of x / dx (rows of a [B, n_latent, 512] w+ tensor) */
int p2l_linear_fwd_ld(const float* x, int x_ld, const float* W, const float* bias, float* y,
                      int Bn, int K, int N, void* stream);
int p2l_linear_bwd_ld(const float* dy, const float* W, float* dx, int dx_ld, int Bn, int K,
                      int N, int accumulate, void* stream);

/* ------------------------------------------------------------------------- */
/* Conditional BatchNorm folded to a per-(sample,channel) affine.            */
/*   gain = 1 + g_raw, bias = b_raw  (g_raw/b_raw = rows of the big linear)  */
/*   s = gain * rstd[c],  t = bias - mean[c] * s                             */
/* bwd: dg_raw = ds*rstd - dt*mean*rstd ; db_raw = dt                        */
/* ------------------------------------------------------------------------- */
int p2l_cbn_fold_fwd(const float* g_raw, const float* b_raw, const float* mean,
                     const float* rstd, float* s, float* t, int Bn, int C,
                     int raw_ld, void* stream);
int p2l_cbn_fold_bwd(const float* ds, const float* dt, const float* mean,
                     const float* rstd, float* dg_raw, float* db_raw, int Bn,
                     int C, int raw_ld, void* stream);

/* Backward of a = max(x*s+t,0) given da ([B,H,W,C]):                        */
/*   g = (x*s+t>0) ? da : 0 ; dx = g*s + skip ; ds[b,c] = sum_p g*x ;        */
/*   dt[b,c] = sum_p g   (written at ds[b*dsdt_bstride + c]).                 */
/* skip (optional) is the GenBlock shortcut gradient: the block-output        */
/* gradient dY restricted to channels c < skip_C; with skip_ups=1 dY lives at */
/* [B,2H,2W,*] and its 2x2 children are summed (nearest-x2 backward).         */
/* Deterministic two-stage reduction; `partial` needs 2*B*nblk*C floats with  */
/* nblk = p2l_affine_relu_bwd_nblk(H*W).  C % 64 == 0.                        */
int p2l_affine_relu_bwd_nblk(int P);
int p2l_affine_relu_bwd(const float* da, int da_ld, const float* x, int x_ld,
                        const float* s, const float* t, int st_bstride,
                        const float* skip, int skip_ld, int skip_C, int skip_ups,
                        float* dx, int dx_ld, float* ds, float* dt,
                        int dsdt_bstride, float* partial, int Bn, int H, int W,
                        int C, void* stream);

/* plain scale backward (StyleGAN2 modulation a = x*s): dx = da*s + skip ;              */
/* ds[b,c] = sum_p da*x.  `partial`/nblk as p2l_affine_relu_bwd; dt_scratch: B*C floats. */
int p2l_scale_bwd(const float* da, int da_ld, const float* x, int x_ld, const float* s,
                  int st_bstride, const float* skip, int skip_ld, int skip_C, float* dx,
                  int dx_ld, float* ds, float* dt_scratch, int dsdt_bstride, float* partial,
                  int Bn, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------- */
/* Row softmax for the attention matrix and its backward.                    */
/* ------------------------------------------------------------------------- */
int p2l_softmax_fwd(const float* S, float* P, int64_t rows, int cols,
                    void* stream);
int p2l_softmax_bwd(const float* P, const float* dP, float* dS, int64_t rows,
                    int cols, void* stream);

/* 2x2 max pool backward (argmax = first max in window scan order), fused     */
/* with an optional additive term and a ReLU mask of the un-pooled tensor:    */
/*   dy[p,c] = ((y[p,c]==max of its quad, first hit) ? dyp[q,c] : 0)         */
/*             + (add ? add[p,c] : 0) ;  if relu_mask: dy *= (y>0)            */
int p2l_maxpool2_bwd(const float* y, int y_ld, const float* dyp, int dyp_ld,
                     const float* add, int add_ld, float* dy, int dy_ld, int Bn,
                     int H, int W, int C, int relu_mask, void* stream);
/* same, and one partial maximum of |dy| per wave for the conv that reads dy next (P2LAmax.in):
 * amax_out [Bn][p2l_maxpool2_bwd_amax_slots(H, W, C)] floats, or NULL */
int p2l_maxpool2_bwd_amax_slots(int H, int W, int C);
int p2l_maxpool2_bwd_amax(const float* y, int y_ld, const float* dyp, int dyp_ld,
                          const float* add, int add_ld, float* dy, int dy_ld, int Bn, int H,
                          int W, int C, int relu_mask, float* amax_out, void* stream);
/* dy = (y>0) ? g : 0  (plain ReLU mask, used where no pool follows) */
int p2l_relu_mask(const float* y, int y_ld, const float* g, int g_ld, float* dy,
                  int dy_ld, int64_t P, int C, void* stream);

/* ------------------------------------------------------------------------- */
/* Image <-> NHWC16 helpers (3 channels padded to 16 floats per pixel).      */
/* ------------------------------------------------------------------------- */
int p2l_nchw3_to_nhwc16(const float* src, float* dst, int Bn, int H, int W,
                        void* stream);
int p2l_nhwc16_to_nchw3(const float* src, float* dst, int Bn, int H, int W,
                        void* stream);
/* d(pre-tanh) = d(img) * (1 - img^2), in place on the 16-channel gradient    */
int p2l_tanh_bwd16(const float* img, float* dimg, int64_t P, void* stream);

/* ------------------------------------------------------------------------- */
/* Losses (pix2latent/loss_functions.py:117-124, :140-148).                  */
/* ------------------------------------------------------------------------- */
/* Weighted L1:  loss[b] = sum_{c,p} |t-o|*w / sum_{c,p} w  (w = weight*mask) */
/* target/weight are NCHW3 [B,3,H,W] as the Python API hands them over; img   */
/* and dimg are NHWC16.  dimg (+)= gscale[b] * sign(o-t)*w/sum w.             */
/* wsum[b] is written by p2l_weight_sum.                                      */
int p2l_weight_sum(const float* weight, const float* loss_mask, float* wsum,
                   int Bn, int HW3, void* stream);
/* wsrc[b,p] = sum_c weight[b,c,p]*loss_mask[b,c,p]  (per-pixel LPIPS weight) */
int p2l_weight_map(const float* weight, const float* loss_mask, float* wsrc,
                   int Bn, int H, int W, void* stream);
int p2l_l1_loss_nblk(int H, int W); /* floats per sample needed in `partial` */
int p2l_l1_loss_fwd(const float* img16, const float* target, const float* weight,
                    const float* loss_mask, const float* wsum, float* loss,
                    float* partial, int Bn, int H, int W, void* stream);
int p2l_l1_loss_bwd(const float* img16, const float* target, const float* weight,
                    const float* loss_mask, const float* wsum,
                    const float* gscale, float* dimg16, int Bn, int H, int W,
                    int accumulate, void* stream);

/* LPIPS tap (lpips.LPIPS(spatial=True) restated in oracle/lpips_ref.py):     */
/*   nf = f / (||f||_2 + 1e-10) ; d[p] = sum_c lin[c]*(nf_o - nf_t)^2         */
/*   partial[b][blk] = sum over the block's pixels of d[p] * wt[b,p]          */
/* wt = adjoint-bilinear-resized per-pixel weight map, so that the sum equals */
/* the spatially weighted sum of the bilinearly upsampled distance map.       */
/* nft = cached normalised target features (bstride 0 = one shared target).   */
/* C in {64,128,256,512}.  Finish with p2l_reduce_rows(div = wsum).           */
int p2l_lpips_normalize(const float* f, float* nf, int64_t P, int C,
                        void* stream);
int p2l_lpips_tap_nblk(int P, int C);
int p2l_lpips_tap_fwd(const float* f, const float* nft, int64_t nft_bstride,
                      const float* lin, const float* wt, int64_t wt_bstride,
                      float* loss_partial, int Bn, int P, int C, void* stream);
/* df[b,p,c] = gscale[b] * wt[b,p] * d d[p] / d f[c]  (gscale already holds   */
/* beta / wsum[b] * upstream grad)                                            */
int p2l_lpips_tap_bwd(const float* f, const float* nft, int64_t nft_bstride,
                      const float* lin, const float* wt, int64_t wt_bstride,
                      const float* gscale, float* df, int Bn, int P, int C,
                      void* stream);
/* The same for a tap that relu -> 2x2 max pool follows (VGG16 conv1_2 / 2_2 / 3_3 / 4_3,
 * reference /root/reference/pix2latent/loss_functions.py:131-142 through lpips' vgg16 slices): the whole
 * gradient of the conv output f [Bn][H][W][C] in one pass,
 *   df = (f > 0) ? tap gradient + (f is the first maximum of its quad ? dyp[quad] : 0) : 0
 * = p2l_lpips_tap_bwd followed by p2l_maxpool2_bwd_amax(add = tap gradient, relu_mask = 1), bit for bit.
 * dyp [Bn][H/2][W/2][C]; amax_out [Bn][p2l_lpips_tap_nblk(H * W, C)] partial maxima of |df| (P2LAmax.in of
 * the conv that reads df next) or NULL.  H even, W a multiple of 2 * (pixels per wave of the tap kernel). */
int p2l_lpips_tap_pool_bwd(const float* f, const float* nft, int64_t nft_bstride,
                           const float* lin, const float* wt, int64_t wt_bstride,
                           const float* gscale, const float* dyp, float* df, float* amax_out,
                           int Bn, int H, int W, int C, void* stream);
/* adjoint of F.interpolate(bilinear, align_corners=False) from h x w up to   */
/* H x W: wt[b,q] = sum_p U[p,q] * wsrc[b,p]                                  */
int p2l_bilinear_adjoint(const float* wsrc, float* wt, int Bn, int H, int W,
                         int h, int w, void* stream);
/* deterministic finishing reduction:                                         */
/*   out[b] (+)= scale * sum_i partial[b][i] / (div ? div[b] : 1)             */
int p2l_reduce_rows(const float* partial, float* out, int Bn, int n, float scale,
                    const float* div, int accumulate, void* stream);

/* ------------------------------------------------------------------------- */
/* Fused Adam over a contiguous block of per-sample leaves                    */
/* (torch.optim.Adam defaults, closure.py:65 / variable_manager.py:231-238).  */
/*   p,g,m,v: [n] ; step_count is the 1-based step number                     */
/* ------------------------------------------------------------------------- */
int p2l_adam_step(float* p, const float* g, float* m, float* v, int64_t n,
                  float lr, float beta1, float beta2, float eps, int step_count,
                  void* stream);
/* same update with the step numbers kept on the device: uses step_counters[0] + 1 as the step
 * of all n elements, then advances step_counters[0 .. n_counters) by one (one counter per
 * sample of the chunk).  No launch argument changes between steps: the inner step can be
 * captured in a HIP graph and replayed. */
int p2l_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n,
                      float lr, float beta1, float beta2, float eps,
                      int32_t* step_counters, int n_counters, void* stream);
int p2l_clamp(float* p, int64_t n, float lo, float hi, void* stream);

/* out[b] = a[b] * scale / (div ? div[b] : 1) */
int p2l_vec_scale_div(const float* a, const float* div, float* out, int n,
                      float scale, void* stream);
/* cond[b] = cat(z[b], c[b]) ; and the split of its gradient */
int p2l_concat2(const float* z, const float* c, float* cond, int Bn, int nz,
                int nc, void* stream);
int p2l_split2(const float* dcond, float* dz, float* dc, int Bn, int nz, int nc,
               void* stream);

/* ------------------------------------------------------------------------- */
/* Whole-graph plans: the native runtime that sequences the kernels above.   */
/* Everything is enqueued on `stream`; the caller owns `ws` (workspace) and   */
/* all parameter memory.  Pointers inside the structs are device pointers.    */
/* ------------------------------------------------------------------------- */
#define P2L_MAX_BLOCKS 16

typedef struct P2LGenBlock {
  int32_t cin, cout, up;
  int32_t cbn_off[4];             /* channel offsets of bn_0..bn_3 in the CBN arrays */
  const float* w[4];              /* packed forward weights conv_0..conv_3   */
  const float* b[4];              /* biases                                  */
  const float* wt[4];             /* packed input-gradient weights           */
  const float* w1_sp;             /* up blocks: conv_1 in sub-pixel form (or NULL) */
  const float* wt1_sp;            /* and its input-gradient form (or NULL)   */
} P2LGenBlock;

typedef struct P2LBigGAN {
  int32_t n_blocks;               /* 12 for biggan-deep-256                  */
  int32_t attn_before;            /* SelfAttn runs before block[attn_before] */
  int32_t ch;                     /* channel_width (128)                     */
  int32_t z_dim, c_dim;           /* 128, 128 -> cond = 256                  */
  int32_t cbn_total;              /* sum of all conditional-BN channels      */
  const float* genz_w;            /* [cond][16*16*ch] (K-major)              */
  const float* genz_b;            /* [16*16*ch]  (output is NHWC [4,4,16ch]) */
  const float* cbn_w;             /* [cond][2*cbn_total]: gains then biases  */
  const float* cbn_mean;          /* [cbn_total] running mean (truncation row) */
  const float* cbn_rstd;          /* [cbn_total] 1/sqrt(var+eps)             */
  P2LGenBlock blocks[P2L_MAX_BLOCKS];
  int32_t attn_ch;                /* 512                                     */
  const float* att_w[4];          /* theta, phi, g, o packed forward         */
  const float* att_wt[4];         /* packed input-gradient                   */
  float gamma;
  const float* tail_s;            /* [ch] folded unconditional BN            */
  const float* tail_t;
  const float* rgb_w;             /* packed 3x3 ch -> 32 (3 real)            */
  const float* rgb_b;             /* [32]                                    */
  const float* rgb_wt;            /* packed input-gradient 16 (3 real) -> ch */
  int32_t wfmt;                   /* P2L_WFMT_* of every 3x3 / sub-pixel weight */
  int32_t reserved1;
} P2LBigGAN;

size_t p2l_biggan_ws_bytes(const P2LBigGAN* m, int Bn);
/* z [B,z_dim], c [B,c_dim] -> img16 [B,res,res,16] (tanh output, ch 0..2)   */
int p2l_biggan_fwd(const P2LBigGAN* m, const float* z, const float* c, int Bn,
                   void* ws, size_t ws_bytes, float* img16, void* stream);
/* needs the workspace of the matching fwd; dimg16 is consumed (overwritten) */
int p2l_biggan_bwd(const P2LBigGAN* m, int Bn, void* ws, size_t ws_bytes,
                   const float* img16, float* dimg16, float* dz, float* dc,
                   void* stream);

typedef struct P2LVggLpips {
  const float* w[13];             /* packed forward (conv0: Cin padded to 16) */
  const float* b[13];
  const float* wt[13];            /* packed input-gradient (conv0: N padded to 32,
                                     pre-multiplied by 1/scale)               */
  const float* lin[5];            /* [C_k] LPIPS linear weights               */
  const float* in_s;              /* [16] scaling layer: 1/scale (0 padded)   */
  const float* in_t;              /* [16] -shift/scale                        */
  int32_t wfmt;                   /* P2L_WFMT_* of the 13 conv weights (both copies) */
  int32_t reserved1;
} P2LVggLpips;

/* cached, target-dependent state (caller allocates):                         */
typedef struct P2LLossCache {
  float* nft[5];                  /* normalised target features [B,P_k,C_k]   */
  float* wt[5];                   /* adjoint-resized weight maps [B,P_k]      */
  float* wsum;                    /* [B]                                      */
} P2LLossCache;
size_t p2l_loss_cache_floats(int Bn, int H, int W, size_t nft_off[5],
                             size_t wt_off[5], size_t* wsum_off);
size_t p2l_projloss_ws_bytes(int Bn, int H, int W);
/* target/weight/loss_mask: NCHW3 [B,3,H,W] (loss_mask may be NULL)           */
int p2l_projloss_prepare(const P2LVggLpips* v, const float* target,
                         const float* weight, const float* loss_mask, int Bn,
                         int H, int W, const P2LLossCache* cache, void* ws,
                         size_t ws_bytes, void* stream);
/* loss[b] = L1_weighted + beta * LPIPS_weighted; use_lpips=0 -> L1 only;       *
 * in the backward entry points use_lpips=2 -> gradient of beta * LPIPS_weighted *
 * alone (PerceptualLoss, pix2latent/loss_functions.py:140-148)                 */
int p2l_projloss_fwd(const P2LVggLpips* v, const float* img16,
                     const float* target, const float* weight,
                     const float* loss_mask, const P2LLossCache* cache,
                     float beta, int use_lpips, int Bn, int H, int W, void* ws,
                     size_t ws_bytes, float* loss, float* loss_l1,
                     float* loss_lpips, void* stream);
int p2l_projloss_bwd(const P2LVggLpips* v, const float* img16,
                     const float* target, const float* weight,
                     const float* loss_mask, const P2LLossCache* cache,
                     float beta, int use_lpips, const float* gloss, int Bn,
                     int H, int W, void* ws, size_t ws_bytes, float* dimg16,
                     void* stream);

/* ---- LPIPS-AlexNet variant of the same loss --------------------------------------
 * ProjectionLoss() defaults to lpips_net='alex' (pix2latent/loss_functions.py:87,131);
 * every examples/invert_*.py uses that default.  torchvision alexnet.features:
 * conv(3,64,11,s4,p2) | pool3/2 conv(64,192,5,p2) | pool3/2 conv(192,384,3,p1) |
 * conv(384,256,3,p1) | conv(256,256,3,p1), ReLU after each = the 5 LPIPS taps.        */
typedef struct P2LGConv {         /* generic NHWC conv: any size / stride / padding     */
  int32_t B, Hi, Wi, Cin;         /* Cin % 16 == 0 (image: nhwc16)                      */
  int32_t Cout;                   /* % 64 == 0                                          */
  int32_t KH, KW, stride, pad;
  int32_t x_ld, y_ld, res_ld, mask_ld;
  int32_t n_store;                /* 0 = Cout                                           */
  int32_t relu;
  int32_t reserved0;
} P2LGConv;
/* y = [mask>0] relu?( conv(pro_s*x+pro_t zero-padded) + bias + res );
 * w packed by p2l_pack_conv_weight(taps = KH*KW, K chunk 16).                          */
int p2l_gconv_fwd(const P2LGConv* d, const float* x, const float* w_packed, const float* bias,
                  const float* pro_s, const float* pro_t, const float* res, const float* mask,
                  float* y, void* stream);
/* 3x3 stride-2 max-pool, no padding (Ho = (Hi-3)/2+1)                                  */
int p2l_maxpool3s2_fwd(const float* x, float* y, int Bn, int Hi, int Wi, int C, void* stream);
/* dx = (pool-backward(gpooled) [first maximum in scan order] + gtap) * (x > 0)          */
int p2l_maxpool3s2_bwd(const float* x, const float* gpooled, const float* gtap, float* dx,
                       int Bn, int Hi, int Wi, int C, void* stream);
/* input gradient of a strided KxK conv to the 3 image channels; w_t3: [K*K][3][Co]      */
int p2l_conv1_dgrad(const float* g, const float* w_t3, float* dimg16, int Bn, int H, int W,
                    int Co, int K, int S, int pad, void* stream);

typedef struct P2LAlexLpips {
  const float* w[5];              /* packed forward (conv0: Cin padded to 16)           */
  const float* b[5];
  const float* wt[5];             /* [1..4]: packed input-gradient; [0]: [121][3][64]
                                     pre-multiplied by 1/scale (p2l_conv1_dgrad)        */
  const float* lin[5];            /* [C_k] LPIPS linear weights, C = 64,192,384,256,256 */
  const float* in_s;              /* [16] scaling layer: 1/scale (0 padded)             */
  const float* in_t;              /* [16] -shift/scale                                  */
} P2LAlexLpips;
size_t p2l_alex_cache_floats(int Bn, int H, int W, size_t nft_off[5], size_t wt_off[5],
                             size_t* wsum_off);
size_t p2l_alexloss_ws_bytes(int Bn, int H, int W);
int p2l_alexloss_prepare(const P2LAlexLpips* v, const float* target, const float* weight,
                         const float* loss_mask, int Bn, int H, int W,
                         const P2LLossCache* cache, void* ws, size_t ws_bytes, void* stream);
int p2l_alexloss_fwd(const P2LAlexLpips* v, const float* img16, const float* target,
                     const float* weight, const float* loss_mask, const P2LLossCache* cache,
                     float beta, int use_lpips, int Bn, int H, int W, void* ws,
                     size_t ws_bytes, float* loss, float* loss_l1, float* loss_lpips,
                     void* stream);
int p2l_alexloss_bwd(const P2LAlexLpips* v, const float* img16, const float* target,
                     const float* weight, const float* loss_mask, const P2LLossCache* cache,
                     float beta, int use_lpips, const float* gloss, int Bn, int H, int W,
                     void* ws, size_t ws_bytes, float* dimg16, void* stream);

/* ---- LPIPS-SqueezeNet variant (round 6) ---------------------------------------------
 * The reference hands any lpips net name through (pix2latent/loss_functions.py:128-131).
 * torchvision squeezenet1_1.features as lpips slices it: conv(3,64,k3,s2) | pool fire fire |
 * pool fire fire | pool fire | fire | fire | fire = 7 taps of 64,128,256,384,384,512,512
 * channels; Fire(in, s, e): relu(conv1x1(in,s)) -> cat(relu(conv1x1(s,e)), relu(conv3x3(s,e))).
 * The ceil-mode pools equal the floor-mode ones for the sizes taken (every window whole:
 * powers of two >= 64; other sizes: P2L_EINVAL / ws_bytes 0).  Weights: p2l_pack_gconv_weight
 * (16-channel K chunks for every kernel size); squeeze convs padded to 64 output channels.     */
int p2l_pack_gconv_weight(const float* w_oihw, int O, int I, int taps, int N_pad, int K_pad,
                          int transpose_flip, float* w_packed, void* stream);
typedef struct P2LSqueezeLpips {
  const float* w0; const float* b0;   /* conv0 packed (Cin padded to 16), bias [64]              */
  const float* wt0;                   /* [9][3][64] with 1/scale folded in (p2l_conv1_dgrad)      */
  const float* sq_w[8]; const float* sq_b[8];   /* squeeze 1x1: N_pad 64, bias padded to 64       */
  const float* e1_w[8]; const float* e1_b[8];   /* expand 1x1                                     */
  const float* e3_w[8]; const float* e3_b[8];   /* expand 3x3, pad 1                              */
  const float* sq_wt[8]; const float* e1_wt[8]; const float* e3_wt[8];   /* input-gradient copies  */
  const float* lin[7];                /* [C_k] LPIPS linear weights                               */
  const float* in_s; const float* in_t;   /* [16] scaling layer                                   */
} P2LSqueezeLpips;
typedef struct P2LLossCache7 {
  float* nft[7];                  /* normalised target features [B,P_k,C_k]   */
  float* wt[7];                   /* adjoint-resized weight maps [B,P_k]      */
  float* wsum;                    /* [B]                                      */
} P2LLossCache7;
size_t p2l_sqz_cache_floats(int Bn, int H, int W, size_t nft_off[7], size_t wt_off[7], size_t* wsum_off);
size_t p2l_sqzloss_ws_bytes(int Bn, int H, int W);
int p2l_sqzloss_prepare(const P2LSqueezeLpips* v, const float* target, const float* weight,
                        const float* loss_mask, int Bn, int H, int W, const P2LLossCache7* cache, void* ws,
                        size_t ws_bytes, void* stream);
int p2l_sqzloss_fwd(const P2LSqueezeLpips* v, const float* img16, const float* target, const float* weight,
                    const float* loss_mask, const P2LLossCache7* cache, float beta, int use_lpips, int Bn,
                    int H, int W, void* ws, size_t ws_bytes, float* loss, float* loss_l1, float* loss_lpips,
                    void* stream);
int p2l_sqzloss_bwd(const P2LSqueezeLpips* v, const float* img16, const float* target, const float* weight,
                    const float* loss_mask, const P2LLossCache7* cache, float beta, int use_lpips,
                    const float* gloss, int Bn, int H, int W, void* ws, size_t ws_bytes, float* dimg16,
                    void* stream);

/* Fused F.affine_grid + F.grid_sample (bilinear, zeros, align_corners=False) on NCHW
 * images; theta is [B][6] row-major 2x3.  Replaces the warps of
 * pix2latent/transform/spatial_transform.py:69-104 (SpatialTransform.transform /
 * invert_transform). */
int p2l_affine_grid_sample(const float* src, const float* theta, float* dst, int Bn,
                           int C, int H, int W, void* stream);
/* Its backward, for the differentiable uses of SpatialTransform (reference
 * pix2latent/loss_functions.py:30-38 invertibility_loss; autograd through
 * spatial_transform.py:69-104): d src (gather form of the adjoint, may be NULL) and d theta [B][6]
 * (may be NULL; needs src and p2l_affine_grid_sample_bwd_ws_bytes of workspace).  Fixed summation
 * order, no atomics. */
size_t p2l_affine_grid_sample_bwd_ws_bytes(int Bn, int H, int W);
int p2l_affine_grid_sample_bwd(const float* src, const float* theta, const float* dout,
                               float* dsrc, float* dtheta, int Bn, int C, int H, int W,
                               void* workspace, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* StyleGAN2 (rosinality) pieces; replace fused_bias_act / upfirdn2d and the   */
/* torch ops inside Generator.forward (reference model/stylegan2.py:116-125).  */
/* ------------------------------------------------------------------------- */
int p2l_sg2_pixelnorm_fwd(const float* z, float* y, int Bn, int D, void* stream);
int p2l_sg2_pixelnorm_bwd(const float* z, const float* dy, float* dz, int Bn, int D, void* stream);
/* x = lrelu(x + bias*bias_mul, 0.2) * sqrt(2), in place ; g *= lrelu'(y) */
int p2l_sg2_bias_lrelu_fwd(float* x, const float* bias, float bias_mul, int Bn, int D, void* stream);
int p2l_sg2_lrelu_bwd(const float* y, float* g, int n, void* stream);
/* d[b,o] = rsqrt(sum_i s[b,i]^2 Wsq[i][o] + 1e-8) and its gradient into ds */
int p2l_sg2_demod_fwd(const float* s, const float* Wsq, float* d, int Bn, int Cin, int Cout,
                      void* stream);
int p2l_sg2_demod_bwd(const float* s, const float* Wsq, const float* d, const float* dd, float* ds,
                      int Bn, int Cin, int Cout, int accumulate, void* stream);
/* y[B,H,W,C] = lrelu(blur4x4(u[B,H+2,W+2,C]) * d[b,c] + nw*noise[b,p] + bias[c]) * sqrt2 */
int p2l_sg2_blur_fwd(const float* u, const float* d, const float* noise, float nw,
                     const float* bias, float* y, int Bn, int H, int W, int C, void* stream);
/* activation backward of a styled conv: gd = dy*lrelu'(y)*d ; dd[b,c] = sum_p dy*lrelu'(y)*c
 * with c = (pre - nw*noise - bias)/d recomputed from y ; dnoise[b,p] = nw*sum_c dy*lrelu'(y)
 * (optional).  partial: B*nblk*C floats, strips: (C/64)*B*P floats (if dnoise). */
int p2l_sg2_act_bwd_nblk(int P);
int p2l_sg2_styled_act_bwd(const float* dy, const float* y, const float* d, const float* noise,
                           float nw, const float* bias, float* gd, float* dd, float* dnoise,
                           float* partial, float* strips, int Bn, int P, int C, void* stream);
int p2l_sg2_blur_bwd(const float* g, float* du, int Bn, int H, int W, int C, void* stream);
/* The same three kernels, leaving the per-image maxima of what they write for the fp16 x 2 conv that reads it
 * (P2LAmax.in with in_n = P2L_SG2_AMAX_SLOTS): amax_out [Bn][P2L_SG2_AMAX_SLOTS] floats the CALLER ZEROED on the
 * same stream; partial maxima arrive by atomic max on the bit pattern -- exact, so the reader's scale (and every
 * bit of its output) is that of its own pass.  blur_fwd: the maxima of |y * next_s[b,c]| (next_s [Bn][C] = the
 * style the reader fuses as its prologue -> P2LAmax.in_applied = 1; NULL: of |y|); styled_act_bwd: of |gd|;
 * blur_bwd: of |du| over the whole (H+2)^2 frame.  amax_out NULL = the plain form. */
#define P2L_SG2_AMAX_SLOTS 64
int p2l_sg2_blur_fwd_amax(const float* u, const float* d, const float* noise, float nw, const float* bias,
                          float* y, int Bn, int H, int W, int C, const float* next_s, float* amax_out,
                          void* stream);
int p2l_sg2_styled_act_bwd_amax(const float* dy, const float* y, const float* d, const float* noise,
                                float nw, const float* bias, float* gd, float* dd, float* dnoise,
                                float* partial, float* strips, int Bn, int P, int C, float* amax_out,
                                void* stream);
int p2l_sg2_blur_bwd_amax(const float* g, float* du, int Bn, int H, int W, int C, float* amax_out, void* stream);
/* Deferred form of the second reduction stage of p2l_sg2_styled_act_bwd (partial -> dd), per host thread: between
 * _begin and _flush it is only recorded and _flush runs all of them in ONE launch (same sums, same order).  Every
 * deferred layer needs its own `partial` buffer, untouched until _flush; _cancel drops what was recorded. */
void p2l_sg2_rows_defer_begin(void);
int p2l_sg2_rows_defer_flush(void* stream);
void p2l_sg2_rows_defer_cancel(void);
/* RGB skip upsample (upfirdn2d up=2, [1,3,3,1]) on NHWC16 images and its transpose */
int p2l_sg2_rgb_up_fwd(const float* skip, float* out, int Bn, int h, int w, void* stream);
int p2l_sg2_rgb_up_bwd(const float* dout, float* dskip, int Bn, int h, int w, int accumulate,
                       void* stream);
int p2l_sg2_clamp16_fwd(const float* x, float* y, int64_t P, void* stream);
int p2l_sg2_clamp16_bwd(const float* x, const float* dy, float* dx, int64_t P, void* stream);
int p2l_broadcast_rows(const float* src, float* dst, int64_t n, int Bn, void* stream);
int p2l_add_inplace(float* a, const float* b, int64_t n, void* stream);

#define P2L_SG2_MAX_CONVS 20
#define P2L_SG2_MAX_RGBS 10
typedef struct P2LSg2Conv {
  int32_t cin, cout, up, res;     /* res = output resolution                         */
  const float* w;                 /* packed forward weights, 1/sqrt(cin*9) folded;   */
                                  /* up=1: sub-pixel mode 1 (transposed conv)        */
  const float* wt;                /* packed input-gradient weights                   */
  const float* wsq;               /* [cin][cout]: scale^2 * sum_k w^2 (demodulation) */
  const float* mod_w;             /* [512][cin] modulation weight^T / sqrt(512)      */
  const float* mod_b;             /* [cin]                                           */
  const float* act_b;             /* [cout] FusedLeakyReLU bias                      */
  float noise_w;
  int32_t latent_idx, noise_off;  /* which w+ row; float offset into the flat noise  */
} P2LSg2Conv;
typedef struct P2LSg2Rgb {
  int32_t cin, res, latent_idx, after_conv;   /* reads the output of conv[after_conv] */
  const float* w;                 /* packed 1x1 cin -> 32 (3 real), 1/sqrt(cin) folded */
  const float* wt;                /* packed input-gradient 16 (3 real) -> cin          */
  const float* mod_w; const float* mod_b;
  const float* bias;              /* [32]                                              */
} P2LSg2Rgb;
typedef struct P2LStyleGAN2 {
  int32_t size, n_conv, n_rgb, style_dim, n_latent, noise_total;
  const float* map_w[8];          /* [512][512] K-major, (weight*scale)^T              */
  const float* map_b[8];          /* [512], lr_mul folded                              */
  const float* const_input;       /* [4][4][C0] NHWC                                   */
  P2LSg2Conv conv[P2L_SG2_MAX_CONVS];
  P2LSg2Rgb rgb[P2L_SG2_MAX_RGBS];
  int32_t wfmt;                   /* P2L_WFMT_* of every styled-conv 3x3 weight          */
  int32_t reserved1;
} P2LStyleGAN2;
size_t p2l_sg2_ws_bytes(const P2LStyleGAN2* m, int Bn);
/* The per-layer noise between the reference's per-sample layout [Bn][noise_total] (model/stylegan2.py:128-138
 * `reshape_noise` slices it per layer) and the layer-major one the synthesis entry points take (layer l at
 * Bn*noise_off[l], [Bn][h*w]); to_layer_major = 0: the transpose (gradients on their way back). */
int p2l_sg2_noise_relayout(const P2LStyleGAN2* m, const float* src, float* dst, int Bn, int to_layer_major,
                           void* stream);
/* latent: [B, n_latent, 512] w+ rows (broadcast w for z-mode); noise: [B, noise_total]
 * explicit per-layer noise; img16: [B,size,size,16] clamped image */
int p2l_sg2_synthesis_fwd(const P2LStyleGAN2* m, const float* latent, const float* noise, int Bn,
                          void* ws, size_t ws_bytes, float* img16, void* stream);
/* dlatent [B, n_latent, 512] ; dnoise [B, noise_total] or NULL */
int p2l_sg2_synthesis_bwd(const P2LStyleGAN2* m, const float* latent, const float* noise, int Bn,
                          void* ws, size_t ws_bytes, const float* dimg16, float* dlatent,
                          float* dnoise, void* stream);
/* mapping network: z [B,512] -> w [B,512]; acts: 9*B*512 floats kept for the backward */
int p2l_sg2_mapping_fwd(const P2LStyleGAN2* m, const float* z, float* w, float* acts, int Bn,
                        void* stream);
int p2l_sg2_mapping_bwd(const P2LStyleGAN2* m, const float* z, const float* acts, const float* dw,
                        float* dz, float* scratch /* 2*B*512 */, int Bn, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* P2L_H_ */
