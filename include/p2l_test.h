/* p2l_test.h -- debug / test hooks of libp2l_hip.so.
 *
 * Exported by the same library as include/p2l.h but NOT part of the drop-in boundary: they read back
 * where a plan keeps a saved activation inside the caller's workspace, run host-side self-checks or
 * probe the MFMA layout.  Used by tests/ and tools/ only; a binding of the reference needs none of them.
 */
#ifndef P2L_TEST_H_
#define P2L_TEST_H_

#include "p2l.h"

#ifdef __cplusplus
extern "C" {
#endif

/* test hook, host logic only: the bookkeeping rules of the plans' per-run maxima registry
 * (csrc/p2l_plan.hip AmaxReg); 0 = all hold */
int p2l_selftest_amaxreg(void);

/* debug/test hook: float offset + shape of a saved activation in ws.        */
/* what: 0 = output of layer L (ModuleList index, SelfAttn included),        */
/*       1 = gen_z output, 2 = folded CBN s, 3 = folded CBN t,               */
/*       4 = d s, 5 = d t (after bwd),                                       */
/*       6 = d loss / d (CBN gains | CBN biases) [B][2*cbn_total] (after bwd):*/
/*           the per-layer gradients the parity tests compare with the oracle */
/*       7 = input of bn_1 | bn_2 | bn_3 of GenBlock L / 3 (k = L % 3),         */
/*       8 / 9 = un-pooled phi / g of the self-attention: with 0-3 every       */
/*           discrete decision of a forward pass (ReLU signs, max-pool winners) */
/*           can be read back (tests/test_fixed_mask_grad_gpu.py);             */
/*       10 = the shared split-K workspace, 11 = the maxima ring [sets][B*slots]  */
/*           (tools/ulp_hunt.py)                                               */
int p2l_biggan_ws_lookup(const P2LBigGAN* m, int Bn, int what, int L,
                         size_t* float_off, int32_t shape[4]);

/* debug/test hook: float offset + shape [B,h,w,C] of the post-ReLU output of VGG conv idx
 * (0..12) inside ws after p2l_projloss_fwd */
int p2l_projloss_ws_lookup(int Bn, int H, int W, int idx, size_t* float_off, int32_t shape[4]);

/* debug/test hook: the saved activations of the LPIPS-SqueezeNet pass inside ws after p2l_sqzloss_fwd, [B,h,w,C]:
 * idx 0 = relu(conv0), 1..8 = squeeze outputs of fires 0..7, 9..16 = outputs of fires 0..7 */
int p2l_sqzloss_ws_lookup(int Bn, int H, int W, int idx, size_t* float_off, int32_t shape[4]);

/* debug/test hook: float offset + shape [B,res,res,cout] of the post-activation output of styled conv l
 * (0 .. n_conv-1) inside ws after p2l_sg2_synthesis_fwd: the signs are the run's leaky-ReLU decisions */
int p2l_sg2_ws_lookup(const P2LStyleGAN2* m, int Bn, int l, size_t* float_off, int32_t shape[4]);

/* MFMA layout self-test: C[32x32] = A[32xK] * B[Kx32] via one wave. */
int p2l_mfma_probe(const float* A, const float* B, float* C, int K,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* P2L_TEST_H_ */
