"""ctypes binding of libp2l_hip.so (include/p2l.h).

The product path has NO fallback: if the shared library is missing or a call
returns an error code this module raises.  Build with
`python -c "import __graft_entry__ as g; g.build()"` or `make -C pix2latent_amd/csrc`.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('P2L_LIB_PATH') or os.path.join(_HERE, 'libp2l_hip.so')   # (P2L_LIB_PATH: A/B builds, tools/ab_build.sh)

c_float_p = C.c_void_p  # device pointers travel as integers


class P2LConv(C.Structure):
    _fields_ = [('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('Cin', C.c_int32), ('Cout', C.c_int32), ('taps', C.c_int32),
                ('ups', C.c_int32), ('x_ld', C.c_int32), ('pro', C.c_int32),
                ('pro_bstride', C.c_int32), ('alpha', C.c_float), ('act', C.c_int32),
                ('pool', C.c_int32), ('y_ld', C.c_int32), ('yp_ld', C.c_int32),
                ('n_store', C.c_int32), ('res_ld', C.c_int32), ('res_ups', C.c_int32),
                ('mask_ld', C.c_int32), ('splitk', C.c_int32), ('ext', C.c_int32),
                ('wfmt', C.c_int32), ('form', C.c_int32), ('algo_flops', C.c_double),
                ('w_floats', C.c_int64)]


class P2LAmax(C.Structure):
    _fields_ = [('out', C.c_void_p), ('outp', C.c_void_p), ('in_', C.c_void_p), ('in_n', C.c_int32),
                ('next_s', C.c_void_p), ('next_t', C.c_void_p), ('next_bstride', C.c_int32),
                ('in_applied', C.c_int32)]


class P2LArb(C.Structure):
    _fields_ = [('x', C.c_void_p), ('x_ld', C.c_int32), ('s', C.c_void_p), ('t', C.c_void_p),
                ('st_bstride', C.c_int32), ('skip', C.c_void_p), ('skip_ld', C.c_int32),
                ('skip_C', C.c_int32), ('skip_ups', C.c_int32), ('ds', C.c_void_p),
                ('dt', C.c_void_p), ('dsdt_bstride', C.c_int32), ('partial', C.c_void_p),
                ('nomask', C.c_int32), ('amax', P2LAmax)]


class P2LConvExtra(C.Structure):
    _fields_ = [('oscale', C.c_void_p), ('oscale_bstride', C.c_int32), ('noise', C.c_void_p),
                ('noise_w', C.c_float), ('amax', P2LAmax)]


class P2LGemm(C.Structure):
    _fields_ = [('batch', C.c_int32), ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
                ('lda', C.c_int32), ('ldb', C.c_int32), ('ldc', C.c_int32),
                ('stride_a', C.c_int64), ('stride_b', C.c_int64), ('stride_c', C.c_int64),
                ('a_kmajor', C.c_int32), ('b_kmajor', C.c_int32), ('alpha', C.c_float),
                ('accumulate', C.c_int32)]


class P2LAttn(C.Structure):
    _fields_ = [('B', C.c_int32), ('Nq', C.c_int32), ('Nk', C.c_int32), ('d', C.c_int32),
                ('dv', C.c_int32)]


P2L_MAX_BLOCKS = 16


class P2LGenBlock(C.Structure):
    _fields_ = [('cin', C.c_int32), ('cout', C.c_int32), ('up', C.c_int32),
                ('cbn_off', C.c_int32 * 4),
                ('w', C.c_void_p * 4), ('b', C.c_void_p * 4), ('wt', C.c_void_p * 4),
                ('w1_sp', C.c_void_p), ('wt1_sp', C.c_void_p)]


class P2LBigGAN(C.Structure):
    _fields_ = [('n_blocks', C.c_int32), ('attn_before', C.c_int32), ('ch', C.c_int32),
                ('z_dim', C.c_int32), ('c_dim', C.c_int32), ('cbn_total', C.c_int32),
                ('genz_w', C.c_void_p), ('genz_b', C.c_void_p), ('cbn_w', C.c_void_p),
                ('cbn_mean', C.c_void_p), ('cbn_rstd', C.c_void_p),
                ('blocks', P2LGenBlock * P2L_MAX_BLOCKS),
                ('attn_ch', C.c_int32),
                ('att_w', C.c_void_p * 4), ('att_wt', C.c_void_p * 4),
                ('gamma', C.c_float),
                ('tail_s', C.c_void_p), ('tail_t', C.c_void_p),
                ('rgb_w', C.c_void_p), ('rgb_b', C.c_void_p), ('rgb_wt', C.c_void_p),
                ('wfmt', C.c_int32), ('reserved1', C.c_int32)]


P2L_SG2_MAX_CONVS, P2L_SG2_MAX_RGBS = 20, 10


class P2LSg2Conv(C.Structure):
    _fields_ = [('cin', C.c_int32), ('cout', C.c_int32), ('up', C.c_int32), ('res', C.c_int32),
                ('w', C.c_void_p), ('wt', C.c_void_p), ('wsq', C.c_void_p), ('mod_w', C.c_void_p),
                ('mod_b', C.c_void_p), ('act_b', C.c_void_p), ('noise_w', C.c_float),
                ('latent_idx', C.c_int32), ('noise_off', C.c_int32)]


class P2LSg2Rgb(C.Structure):
    _fields_ = [('cin', C.c_int32), ('res', C.c_int32), ('latent_idx', C.c_int32),
                ('after_conv', C.c_int32), ('w', C.c_void_p), ('wt', C.c_void_p),
                ('mod_w', C.c_void_p), ('mod_b', C.c_void_p), ('bias', C.c_void_p)]


class P2LStyleGAN2(C.Structure):
    _fields_ = [('size', C.c_int32), ('n_conv', C.c_int32), ('n_rgb', C.c_int32),
                ('style_dim', C.c_int32), ('n_latent', C.c_int32), ('noise_total', C.c_int32),
                ('map_w', C.c_void_p * 8), ('map_b', C.c_void_p * 8), ('const_input', C.c_void_p),
                ('conv', P2LSg2Conv * P2L_SG2_MAX_CONVS), ('rgb', P2LSg2Rgb * P2L_SG2_MAX_RGBS),
                ('wfmt', C.c_int32), ('reserved1', C.c_int32)]


class P2LVggLpips(C.Structure):
    _fields_ = [('w', C.c_void_p * 13), ('b', C.c_void_p * 13), ('wt', C.c_void_p * 13),
                ('lin', C.c_void_p * 5), ('in_s', C.c_void_p), ('in_t', C.c_void_p),
                ('wfmt', C.c_int32), ('reserved1', C.c_int32)]


class P2LGConv(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'B', 'Hi', 'Wi', 'Cin', 'Cout', 'KH', 'KW', 'stride', 'pad', 'x_ld', 'y_ld', 'res_ld',
        'mask_ld', 'n_store', 'relu', 'reserved0')]


class P2LAlexLpips(C.Structure):
    _fields_ = [('w', C.c_void_p * 5), ('b', C.c_void_p * 5), ('wt', C.c_void_p * 5),
                ('lin', C.c_void_p * 5), ('in_s', C.c_void_p), ('in_t', C.c_void_p)]


class P2LSqueezeLpips(C.Structure):
    _fields_ = [('w0', C.c_void_p), ('b0', C.c_void_p), ('wt0', C.c_void_p),
                ('sq_w', C.c_void_p * 8), ('sq_b', C.c_void_p * 8), ('e1_w', C.c_void_p * 8), ('e1_b', C.c_void_p * 8),
                ('e3_w', C.c_void_p * 8), ('e3_b', C.c_void_p * 8),
                ('sq_wt', C.c_void_p * 8), ('e1_wt', C.c_void_p * 8), ('e3_wt', C.c_void_p * 8),
                ('lin', C.c_void_p * 7), ('in_s', C.c_void_p), ('in_t', C.c_void_p)]


class P2LLossCache7(C.Structure):
    _fields_ = [('nft', C.c_void_p * 7), ('wt', C.c_void_p * 7), ('wsum', C.c_void_p)]


class P2LLossCache(C.Structure):
    _fields_ = [('nft', C.c_void_p * 5), ('wt', C.c_void_p * 5), ('wsum', C.c_void_p)]


class P2LProfTotals(C.Structure):
    _fields_ = [('size', C.c_uint32), ('count', C.c_int32 * 2), ('reserved0', C.c_int32),
                ('flops', C.c_double * 2), ('ms', C.c_double * 2), ('bytes', C.c_double * 2),
                ('exec_flops', C.c_double * 2), ('mfma_flops', C.c_double * 2),
                ('write_bytes', C.c_double * 2),
                ('fam_count', C.c_int32 * 8), ('fam_ms', C.c_double * 8), ('fam_flops', C.c_double * 8),
                ('fam_mfma_flops', C.c_double * 8), ('fam_bytes', C.c_double * 8)]


# P2LProfTotals.fam_* index -> the kernel a rocprofv3 kernel table lists (include/p2l.h P2L_PROF_FAM_*)
PROF_FAMILIES = {0: 'other (exact-fp32 / bf16x3 direct, 8x16 Winograd)', 1: 'wino16s_conv_kernel<.., H2>',
                 2: 'conv_h2_kernel<9, ..>', 3: 'conv_h2_kernel<4, ..>', 4: 'conv_thinin/thinout_kernel',
                 5: 'pw_h2_kernel / pw_conv_kernel', 6: 'conv_h2r_kernel<..>'}


def prof_totals():
    """p2l_prof_totals: ends the timing session; totals of the timed conv launches per family
    (index 0 = 3x3, 1 = 1x1)"""
    t = P2LProfTotals()
    t.size = C.sizeof(P2LProfTotals)
    check(lib().p2l_prof_totals(C.byref(t)), 'p2l_prof_totals')
    return t


prof_end = prof_totals      # (the helper's name in tools/ and tests/ written before version 101)


ACT_NONE, ACT_RELU, ACT_TANH, ACT_LRELU_SQRT2 = 0, 1, 2, 3
WFMT_F32, WFMT_BF16X3, WFMT_BF16X3W, WFMT_PW, WFMT_BF16X3T = 0, 1, 2, 3, 4
WFMT_FLAG_PW, WFMT_FLAG_THIN, WFMT_FLAG_ATTN_GEMM, WFMT_FLAG_NO_AMAX = 0x10, 0x20, 0x40, 0x80
# P2LConv.form (per launch; the library has no switches of its own)
FORM_AUTO, FORM_NO_WINO, FORM_WINO_ANY, FORM_WINO_8X16, FORM_NO_PW, FORM_NO_THIN, FORM_WINO_BF3 = 0, 1, 2, 4, 8, 16, 32
FORM_WINO_H2_8X16, FORM_WINO_H2_16X16 = 64, 128      # block shape of the fp16 x 2 Winograd kernel (default: from the grid)
FORM_H2R_SEQ_EPI = 512
FORM_NO_SP_SKIP = 1024
FORM_NO_SP_PAIR = 2048
FORM_SP_PAIR = 4096
FORM_NO_H2R = 256                                    # the chunked direct kernel where the register-resident one would run


def default_wfmt():
    """weight format of the 3x3 convs (include/p2l.h).  Default `bf16x3` = P2L_WFMT_BF16X3W: the
    buffer carries the bf16 x 3 direct image, the Winograd-domain images and the fp16 x 2 images, and
    every launch of such a model runs in the fp32-grade fp16 x 2 arithmetic (two fp16 pieces of
    power-of-two scaled operands, three MFMA products: Winograd F(2x2,3x3) from 128 input channels
    up, the direct / sub-pixel kernel otherwise); the bf16 x 3 images (three pieces, six products)
    serve the three-channel image convs, launches without a workspace and P2L_FORM_WINO_BF3.
    P2L_CONV_WFMT=bf16x3-direct keeps every layer on the bf16 x 3 direct kernel, =f32 asks for the
    exact-fp32 MFMA."""
    v = os.environ.get('P2L_CONV_WFMT', 'bf16x3').lower()
    if v in ('f32', 'fp32', '0'):
        return WFMT_F32
    if v in ('bf16x3-direct', 'bf3d', '1'):
        return WFMT_BF16X3
    if v in ('bf16x3', 'bf3', 'bf16x3w', '2'):
        return WFMT_BF16X3W
    raise ValueError('P2L_CONV_WFMT=%r (expected f32, bf16x3 or bf16x3-direct)' % v)


def default_thin():
    """3-channel image convs (conv_to_rgb, first VGG conv and their input gradients) on the
    kernels of csrc/p2l_thin.hip unless P2L_THIN=0 or the exact-fp32 MFMA was asked for"""
    return os.environ.get('P2L_THIN', '1') != '0' and default_wfmt() != WFMT_F32


def default_attn_gemm():
    """P2L_ATTN=0: self-attention of the generator as GEMM + softmax (the attention matrix is
    stored) instead of the fused kernels -- a MODEL descriptor flag, read here once"""
    return os.environ.get('P2L_ATTN', '1') == '0'


def default_no_amax():
    """P2L_AMAX=0: no maxima handed between the convs of a plan (P2LAmax) -- a MODEL descriptor
    flag, read when a model is constructed"""
    return os.environ.get('P2L_AMAX', '1') == '0'


def default_pw():
    """1x1 convs on the 16-bit matrix pipe (csrc/p2l_pw.hip; P2L_WFMT_PW buffers: fp32 | bf16 x 3 |
    fp16 x 2 images) -- fp16 x 2 wherever the producer of the input handed its maxima over or the
    layer is a 4^2 ... 16^2 one, bf16 x 3 otherwise -- unless P2L_PW=0 or the 3x3 convs were asked
    to run on the exact-fp32 MFMA"""
    return os.environ.get('P2L_PW', '1') != '0' and default_wfmt() != WFMT_F32


def pack_conv_weight(src, taps, n_pad, k_pad, flip, wfmt, subpix_mode=None):
    """[O,I,kh,kw] fp32 device tensor -> packed weight buffer of format `wfmt` (1x1 convs are
    always fp32; subpix_mode 0 / 1 = the 16 phase-tap matrices of the sub-pixel forms)"""
    L = lib()
    O, I = src.shape[0], src.shape[1]
    src = src.detach().float().contiguous()
    if taps == 1 and wfmt == WFMT_PW:
        dst = torch.empty(L.p2l_packed_weight_floats(1, n_pad, k_pad, WFMT_PW), device=src.device,
                          dtype=torch.float32)
        check(L.p2l_pack_conv_weight_pw(ptr(src), O, I, n_pad, k_pad, int(flip), ptr(dst), stream()),
              'p2l_pack_conv_weight_pw')
        return dst
    if taps != 9:
        wfmt = WFMT_F32
    if subpix_mode is not None:
        dst = torch.empty(L.p2l_packed_subpix_weight_floats(n_pad, k_pad, wfmt), device=src.device,
                          dtype=torch.float32)
        fn = {WFMT_F32: L.p2l_pack_conv_weight_subpix,
              WFMT_BF16X3W: L.p2l_pack_conv_weight_subpix_h2}.get(wfmt, L.p2l_pack_conv_weight_subpix_bf3)
        check(fn(ptr(src), O, I, n_pad, k_pad, int(flip), int(subpix_mode), ptr(dst), stream()),
              'p2l_pack_conv_weight_subpix')
        return dst
    n = L.p2l_packed_weight_floats(taps, n_pad, k_pad, wfmt)
    dst = torch.empty(n, device=src.device, dtype=torch.float32)
    if wfmt == WFMT_BF16X3T:
        check(L.p2l_pack_conv_weight_bf3t(ptr(src), O, I, n_pad, k_pad, int(flip), ptr(dst), stream()),
              'p2l_pack_conv_weight_bf3t')
        return dst
    fn = {WFMT_F32: L.p2l_pack_conv_weight, WFMT_BF16X3: L.p2l_pack_conv_weight_bf3,
          WFMT_BF16X3W: L.p2l_pack_conv_weight_bf3w}[wfmt]
    check(fn(ptr(src), O, I, taps, n_pad, k_pad, int(flip), ptr(dst), stream()),
          'p2l_pack_conv_weight')
    return dst
def pack_gconv_weight(src, taps, n_pad, k_pad, flip):
    """[O,I,kh,kw] -> the generic gather conv's layout [tap][k_pad/16][n_pad][16] (p2l_pack_gconv_weight)"""
    L = lib()
    src = src.detach().float().contiguous()
    dst = torch.empty(taps * n_pad * k_pad, device=src.device, dtype=torch.float32)
    check(L.p2l_pack_gconv_weight(ptr(src), src.shape[0], src.shape[1], taps, n_pad, k_pad, int(flip), ptr(dst),
                                  stream()), 'p2l_pack_gconv_weight')
    return dst


POOL_NONE, POOL_MAX, POOL_SUM = 0, 1, 2
PRO_NONE, PRO_AFFINE_RELU, PRO_AFFINE = 0, 1, 2

# every symbol include/p2l.h declares (checked by tests/test_abi.py)
EXPORTS = [
    'p2l_version', 'p2l_strerror', 'p2l_last_hip_error',
    'p2l_conv_workspace_bytes', 'p2l_conv_amax_slots', 'p2l_selftest_amaxreg', 'p2l_conv_suggest_splitk', 'p2l_conv_fwd', 'p2l_conv_fwd_ex',
    'p2l_pack_conv_weight', 'p2l_pack_conv_weight_subpix', 'p2l_pack_conv_weight_bf3',
    'p2l_pack_conv_weight_bf3w', 'p2l_packed_weight_floats', 
    'p2l_pack_conv_weight_bf3t',
    'p2l_pack_conv_weight_pw', 'p2l_adam_step_dev',
    'p2l_attn_supported', 'p2l_attn_fwd_ws_bytes', 'p2l_attn_fwd', 'p2l_attn_bwd_dv_ws_bytes',
    'p2l_attn_bwd_dv', 'p2l_attn_bwd_qk_ws_bytes', 'p2l_attn_bwd_qk',
    'p2l_pack_conv_weight_subpix_bf3', 'p2l_pack_conv_weight_subpix_h2', 'p2l_packed_subpix_weight_floats', 'p2l_gemm', 'p2l_gemm_ws_bytes', 'p2l_gemm_ws', 'p2l_linear_fwd', 'p2l_linear_bwd',
    'p2l_cbn_fold_fwd', 'p2l_cbn_fold_bwd', 'p2l_affine_relu_bwd_nblk',
    'p2l_affine_relu_bwd', 'p2l_softmax_fwd', 'p2l_softmax_bwd', 'p2l_maxpool2_bwd', 'p2l_maxpool2_bwd_amax', 'p2l_maxpool2_bwd_amax_slots',
    'p2l_relu_mask', 'p2l_nchw3_to_nhwc16', 'p2l_nhwc16_to_nchw3', 'p2l_tanh_bwd16',
    'p2l_weight_sum', 'p2l_weight_map', 'p2l_l1_loss_nblk', 'p2l_l1_loss_fwd',
    'p2l_l1_loss_bwd', 'p2l_lpips_normalize', 'p2l_lpips_tap_nblk', 'p2l_lpips_tap_fwd',
    'p2l_lpips_tap_bwd', 'p2l_lpips_tap_pool_bwd', 'p2l_bilinear_adjoint', 'p2l_reduce_rows', 'p2l_adam_step',
    'p2l_clamp', 'p2l_affine_grid_sample', 'p2l_affine_grid_sample_bwd', 'p2l_affine_grid_sample_bwd_ws_bytes', 'p2l_vec_scale_div', 'p2l_concat2', 'p2l_split2',
    'p2l_biggan_ws_bytes', 'p2l_biggan_fwd', 'p2l_biggan_bwd', 'p2l_biggan_ws_lookup',
    'p2l_loss_cache_floats', 'p2l_projloss_ws_bytes', 'p2l_projloss_ws_lookup', 'p2l_projloss_prepare',
    'p2l_projloss_fwd', 'p2l_projloss_bwd', 'p2l_mfma_probe', 'p2l_prof_begin', 'p2l_prof_totals', 'p2l_prof_step', 'p2l_prof_dump', 'p2l_wino_split_factor', 'p2l_linear_fwd_ld', 'p2l_linear_bwd_ld', 'p2l_scale_bwd',
    'p2l_sg2_pixelnorm_fwd', 'p2l_sg2_pixelnorm_bwd', 'p2l_sg2_bias_lrelu_fwd', 'p2l_sg2_lrelu_bwd',
    'p2l_sg2_demod_fwd', 'p2l_sg2_demod_bwd', 'p2l_sg2_blur_fwd', 'p2l_sg2_act_bwd_nblk',
    'p2l_sg2_styled_act_bwd', 'p2l_sg2_blur_bwd', 'p2l_sg2_rgb_up_fwd', 'p2l_sg2_rgb_up_bwd',
    'p2l_sg2_clamp16_fwd', 'p2l_sg2_clamp16_bwd', 'p2l_broadcast_rows', 'p2l_add_inplace',
    'p2l_sg2_ws_bytes', 'p2l_sg2_ws_lookup', 'p2l_sg2_synthesis_fwd', 'p2l_sg2_synthesis_bwd', 'p2l_sg2_mapping_fwd',
    'p2l_sg2_mapping_bwd', 'p2l_conv_arb_fusable', 'p2l_conv_arb_nblk',
    'p2l_conv_dgrad_arb', 'p2l_conv_arb_split_fusable', 'p2l_conv_arb_nblk_ws',
    'p2l_conv_dgrad_arb_ws', 'p2l_arb_finish', 'p2l_arb_defer_begin', 'p2l_arb_defer_flush',
    'p2l_arb_defer_cancel',
    'p2l_gconv_fwd', 'p2l_maxpool3s2_fwd', 'p2l_maxpool3s2_bwd', 'p2l_conv1_dgrad',
    'p2l_alex_cache_floats', 'p2l_alexloss_ws_bytes', 'p2l_alexloss_prepare', 'p2l_alexloss_fwd',
    'p2l_alexloss_bwd',
    'p2l_pack_gconv_weight', 'p2l_sqz_cache_floats', 'p2l_sqzloss_ws_bytes', 'p2l_sqzloss_prepare', 'p2l_sqzloss_fwd',
    'p2l_sqzloss_bwd', 'p2l_sqzloss_ws_lookup',
    'p2l_sg2_blur_fwd_amax', 'p2l_sg2_styled_act_bwd_amax', 'p2l_sg2_blur_bwd_amax', 'p2l_sg2_noise_relayout',
    'p2l_sg2_rows_defer_begin', 'p2l_sg2_rows_defer_flush', 'p2l_sg2_rows_defer_cancel',
]

_lib = None
ABI_VERSION = 101


class NativeError(RuntimeError):
    pass


def lib():
    """Load libp2l_hip.so once; fail loudly when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                'libp2l_hip.so not found at %s: the HIP extension is required '
                '(no CPU fallback). Run __graft_entry__.build().' % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        # the structs below mirror include/p2l.h of THIS version (101: P2LConv.w_floats, p2l_prof_totals);
        # an older library would read them at other offsets
        if _lib.p2l_version() < ABI_VERSION:
            v, _lib = _lib.p2l_version(), None
            raise NativeError('libp2l_hip.so at %s is version %d, this package binds version %d: rebuild it '
                              '(__graft_entry__.build())' % (LIB_PATH, v, ABI_VERSION))
        _lib.p2l_strerror.restype = C.c_char_p
        _lib.p2l_arb_defer_begin.restype = None
        _lib.p2l_arb_defer_cancel.restype = None
        for name in ('p2l_conv_workspace_bytes', 'p2l_biggan_ws_bytes',
                     'p2l_projloss_ws_bytes', 'p2l_loss_cache_floats', 'p2l_sg2_ws_bytes',
                     'p2l_alexloss_ws_bytes', 'p2l_alex_cache_floats', 'p2l_sqzloss_ws_bytes', 'p2l_sqz_cache_floats', 'p2l_gemm_ws_bytes',
                     'p2l_packed_weight_floats', 'p2l_packed_subpix_weight_floats', 'p2l_attn_fwd_ws_bytes', 'p2l_affine_grid_sample_bwd_ws_bytes',
                     'p2l_attn_bwd_dv_ws_bytes', 'p2l_attn_bwd_qk_ws_bytes'):
            getattr(_lib, name).restype = C.c_size_t
    return _lib


def check(rc, what=''):
    if rc != 0:
        L = lib()
        raise NativeError('%s failed: %s (rc=%d, hipError=%d)' % (
            what or 'p2l call', L.p2l_strerror(rc).decode(), rc, L.p2l_last_hip_error()))


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    assert t.dtype == torch.float32 and t.is_contiguous(), 'expect contiguous fp32'
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def f32(v):
    return C.c_float(float(v))


def i64(v):
    return C.c_int64(int(v))
