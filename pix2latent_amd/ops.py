"""Thin tensor-level wrappers over the per-kernel C-ABI entry points of
libp2l_hip (include/p2l.h).  The product path drives the whole-graph plans
(`p2l_biggan_*`, `p2l_projloss_*`); these wrappers exist so that every kernel
can be exercised and parity-tested on its own.  All tensors are NHWC fp32 on the
ROCm device.
"""
import ctypes as C

import torch

from . import _native as N


def _lib():
    return N.lib()


def mfma_probe(A, B):
    K = A.shape[1]
    Cm = torch.empty(32, 32, device=A.device)
    N.check(_lib().p2l_mfma_probe(N.ptr(A), N.ptr(B), N.ptr(Cm), K, N.stream()), 'mfma_probe')
    return Cm


def pack_conv_weight(w_oihw, taps, n_pad, k_pad, flip=False, wfmt=N.WFMT_F32):
    if taps == 9 and wfmt == N.WFMT_PW:
        wfmt = N.WFMT_F32
    return N.pack_conv_weight(w_oihw, taps, n_pad, k_pad, flip, wfmt)


def pack_conv_weight_subpix(w_oihw, n_pad, k_pad, flip=False, mode=0, wfmt=N.WFMT_F32):
    return N.pack_conv_weight(w_oihw, 9, n_pad, k_pad, flip, wfmt, subpix_mode=mode)


# P2LConv.form of launches that do not say otherwise (N.FORM_*): a PYTHON-side default for tests
# and tools; the native library takes the form per call and has no switch of its own
DEFAULT_FORM = N.FORM_AUTO


def conv(x, w_packed, B, H, W, Cin, Cout, taps, bias=None, pro=N.PRO_NONE, pro_s=None,
         pro_t=None, pro_bstride=0, ups=False, alpha=1.0, act=N.ACT_NONE, pool=N.POOL_NONE,
         res=None, res_ups=False, mask=None, n_store=None, y_ld=None, want_y=True,
         splitk=None, x_ld=None, ext=0, oscale=None, noise=None, noise_w=0.0, wfmt=N.WFMT_F32,
         form=None, amax_in=None, want_amax=False, amax_next=None, amax_applied=False):
    """amax_in: [B, n] partial maxima of |x| from the launch that wrote x (P2LAmax.in);
    want_amax: also return (amax_y, amax_yp) partial maxima of the outputs, or None when this launch
    writes none (p2l_conv_amax_slots == 0); amax_next = (s, t, bstride): the affine the reader of y
    will fuse -- the maxima of y are recorded as max|y*s + t| (P2LAmax.next_s); amax_applied: amax_in
    was recorded that way with THIS launch's pro_s / pro_t (P2LAmax.in_applied)"""
    d = N.P2LConv()
    d.wfmt = wfmt
    d.form = DEFAULT_FORM if form is None else form
    d.B, d.H, d.W, d.Cin, d.Cout, d.taps = B, H, W, Cin, Cout, taps
    d.ups = int(ups)
    d.x_ld = x_ld if x_ld is not None else Cin
    d.pro, d.pro_bstride = pro, pro_bstride
    d.alpha, d.act, d.pool = alpha, act, pool
    d.n_store = n_store if n_store is not None else Cout
    d.y_ld = y_ld if y_ld is not None else d.n_store if n_store is not None else Cout
    d.yp_ld = d.y_ld
    d.res_ld = res.shape[-1] if res is not None else 0
    d.res_ups = int(res_ups)
    d.mask_ld = mask.shape[-1] if mask is not None else 0
    d.ext = int(ext)
    d.w_floats = w_packed.numel()      # (the library refuses a buffer too short for this format: P2L_EINVAL)
    d.splitk = splitk if splitk is not None else _lib().p2l_conv_suggest_splitk(C.byref(d))
    wsb = _lib().p2l_conv_workspace_bytes(C.byref(d))
    ws = torch.empty(max(wsb // 4, 1), device=x.device)
    if d.ups == 3:      # sub-pixel input-gradient: result at half resolution
        y = torch.empty(B, H // 2, W // 2, d.y_ld, device=x.device)
    elif d.ups == 2 and ext:   # transposed-conv geometry: [B,H+2,W+2,*] buffer
        y = torch.empty(B, H + 2, W + 2, d.y_ld, device=x.device)
    else:
        y = torch.empty(B, H, W, d.y_ld, device=x.device) if want_y else None
    yp = torch.empty(B, H // 2, W // 2, d.yp_ld, device=x.device) if pool else None
    ex = N.P2LConvExtra()
    if oscale is not None:
        ex.oscale, ex.oscale_bstride = oscale.data_ptr(), oscale.shape[-1]
    if noise is not None:
        ex.noise, ex.noise_w = noise.data_ptr(), float(noise_w)
    am_y = am_yp = None
    if amax_in is not None:
        ex.amax.in_, ex.amax.in_n = amax_in.data_ptr(), amax_in.shape[-1]
        ex.amax.in_applied = int(amax_applied)
    if amax_next is not None:
        ex.amax.next_s, ex.amax.next_t, ex.amax.next_bstride = amax_next[0].data_ptr(), amax_next[1].data_ptr(), int(amax_next[2])
    if want_amax:
        ns = _lib().p2l_conv_amax_slots(C.byref(d))
        if ns > 0:
            am_y = torch.full((B, ns), -1.0, device=x.device)
            ex.amax.out = am_y.data_ptr()
            if pool:
                am_yp = torch.full((B, ns), -1.0, device=x.device)
                ex.amax.outp = am_yp.data_ptr()
    N.check(_lib().p2l_conv_fwd_ex(C.byref(d), C.byref(ex), N.ptr(x), N.ptr(w_packed), N.ptr(bias),
                                   N.ptr(pro_s), N.ptr(pro_t), N.ptr(res), N.ptr(mask), N.ptr(y),
                                   N.ptr(yp), N.ptr(ws), C.c_size_t(wsb), N.stream()), 'conv_fwd')
    if want_amax:
        return y, yp, (am_y, am_yp)
    return y, yp


def gemm(A, B, batch, M, Nn, K, a_kmajor=False, b_kmajor=False, alpha=1.0, Cacc=None):
    d = N.P2LGemm()
    d.batch, d.M, d.N, d.K = batch, M, Nn, K
    d.lda = A.shape[-1]
    d.ldb = B.shape[-1]
    d.ldc = Nn
    d.stride_a = A.shape[-2] * A.shape[-1]
    d.stride_b = B.shape[-2] * B.shape[-1]
    d.stride_c = M * Nn
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.alpha = alpha
    d.accumulate = int(Cacc is not None)
    Cm = Cacc if Cacc is not None else torch.empty(batch, M, Nn, device=A.device)
    wsb = _lib().p2l_gemm_ws_bytes(C.byref(d))      # > 0: deep-K product, split over K
    ws = torch.empty(max(wsb // 4, 1), device=A.device)
    N.check(_lib().p2l_gemm_ws(C.byref(d), N.ptr(A), N.ptr(B), N.ptr(Cm), N.ptr(ws),
                               C.c_size_t(wsb), N.stream()), 'gemm')
    return Cm


def _attn_desc(q, k, v):
    d = N.P2LAttn()
    d.B, d.Nq, d.d = q.shape
    d.Nk, d.dv = k.shape[1], v.shape[2]
    return d


def attn_fwd(q, k, v):
    """fused softmax(q k^T) v (p2l_attn_fwd): q [B,Nq,64], k [B,Nk,64], v [B,Nk,256]
    -> out [B,Nq,256], lse [B,Nq]"""
    d = _attn_desc(q, k, v)
    wsb = _lib().p2l_attn_fwd_ws_bytes(C.byref(d))
    if wsb == 0:
        raise N.NativeError('p2l_attn_fwd does not take this shape')
    ws = torch.empty(wsb // 4, device=q.device)
    out = torch.empty(d.B, d.Nq, d.dv, device=q.device)
    lse = torch.empty(d.B, d.Nq, device=q.device)
    N.check(_lib().p2l_attn_fwd(C.byref(d), N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(out), N.ptr(lse),
                                N.ptr(ws), C.c_size_t(wsb), N.stream()), 'attn_fwd')
    return out, lse


def attn_bwd_dv(q, k, dout, lse):
    """d v[j] = sum_i exp(q_i k_j - lse_i) dout[i] (p2l_attn_bwd_dv)"""
    d = _attn_desc(q, k, dout)
    wsb = _lib().p2l_attn_bwd_dv_ws_bytes(C.byref(d))
    ws = torch.empty(wsb // 4, device=q.device)
    dv = torch.empty(d.B, d.Nk, d.dv, device=q.device)
    N.check(_lib().p2l_attn_bwd_dv(C.byref(d), N.ptr(q), N.ptr(k), N.ptr(dout), N.ptr(lse),
                                   N.ptr(dv), N.ptr(ws), C.c_size_t(wsb), N.stream()), 'attn_bwd_dv')
    return dv


def attn_bwd_qk(q, k, v, out, dout, lse):
    """d q, d k of the fused attention (p2l_attn_bwd_qk); also returns the dS^T scratch"""
    d = _attn_desc(q, k, v)
    wsb = _lib().p2l_attn_bwd_qk_ws_bytes(C.byref(d))
    ws = torch.empty(wsb // 4, device=q.device)
    dst = torch.empty(d.B, d.Nk, d.Nq, device=q.device)
    dq, dk = torch.empty_like(q), torch.empty_like(k)
    N.check(_lib().p2l_attn_bwd_qk(C.byref(d), N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(out), N.ptr(dout),
                                   N.ptr(lse), N.ptr(dst), N.ptr(dq), N.ptr(dk), N.ptr(ws),
                                   C.c_size_t(wsb), N.stream()), 'attn_bwd_qk')
    return dq, dk, dst


def linear_fwd(x, W, bias=None):
    Bn, K = x.shape
    Nn = W.shape[1]
    y = torch.empty(Bn, Nn, device=x.device)
    N.check(_lib().p2l_linear_fwd(N.ptr(x), N.ptr(W), N.ptr(bias), N.ptr(y), Bn, K, Nn,
                                  N.stream()), 'linear_fwd')
    return y


def linear_bwd(dy, W, dx=None):
    Bn, Nn = dy.shape
    K = W.shape[0]
    acc = dx is not None
    if dx is None:
        dx = torch.empty(Bn, K, device=dy.device)
    N.check(_lib().p2l_linear_bwd(N.ptr(dy), N.ptr(W), N.ptr(dx), Bn, K, Nn, int(acc),
                                  N.stream()), 'linear_bwd')
    return dx


def affine_relu_bwd(da, x, s, t, st_bstride, skip=None, skip_C=0, skip_ups=False):
    Bn, H, W, Cc = x.shape
    nblk = _lib().p2l_affine_relu_bwd_nblk(H * W)
    part = torch.empty(2 * Bn * nblk * Cc, device=x.device)
    dx = torch.empty_like(x)
    ds = torch.empty(Bn, Cc, device=x.device)
    dt = torch.empty(Bn, Cc, device=x.device)
    N.check(_lib().p2l_affine_relu_bwd(
        N.ptr(da), da.shape[-1], N.ptr(x), Cc, N.ptr(s), N.ptr(t), st_bstride, N.ptr(skip),
        skip.shape[-1] if skip is not None else 0, skip_C, int(skip_ups), N.ptr(dx), Cc,
        N.ptr(ds), N.ptr(dt), Cc, N.ptr(part), Bn, H, W, Cc, N.stream()), 'affine_relu_bwd')
    return dx, ds, dt


def softmax_fwd(S):
    rows, cols = S.numel() // S.shape[-1], S.shape[-1]
    P = torch.empty_like(S)
    N.check(_lib().p2l_softmax_fwd(N.ptr(S), N.ptr(P), N.i64(rows), cols, N.stream()), 'softmax')
    return P


def softmax_bwd(P, dP):
    rows, cols = P.numel() // P.shape[-1], P.shape[-1]
    dS = torch.empty_like(P)
    N.check(_lib().p2l_softmax_bwd(N.ptr(P), N.ptr(dP), N.ptr(dS), N.i64(rows), cols,
                                   N.stream()), 'softmax_bwd')
    return dS


def maxpool2_bwd(y, dyp, add=None, relu_mask=False):
    Bn, H, W, Cc = y.shape
    dy = torch.empty_like(y)
    N.check(_lib().p2l_maxpool2_bwd(N.ptr(y), Cc, N.ptr(dyp), Cc, N.ptr(add), Cc, N.ptr(dy), Cc,
                                    Bn, H, W, Cc, int(relu_mask), N.stream()), 'maxpool2_bwd')
    return dy


def lpips_tap_pool_bwd(f, nft, lin, wt, gscale, dyp, want_amax=False):
    """gradient of a VGG tap that relu -> max pool follows, in one pass (p2l_lpips_tap_pool_bwd)"""
    Bn, H, W, Cc = f.shape
    df = torch.empty_like(f)
    am = torch.full((Bn, _lib().p2l_lpips_tap_nblk(H * W, Cc)), -1.0, device=f.device) if want_amax else None
    N.check(_lib().p2l_lpips_tap_pool_bwd(N.ptr(f), N.ptr(nft), N.i64(H * W * Cc), N.ptr(lin), N.ptr(wt),
                                          N.i64(H * W), N.ptr(gscale), N.ptr(dyp), N.ptr(df), N.ptr(am),
                                          Bn, H, W, Cc, N.stream()), 'lpips_tap_pool_bwd')
    return (df, am) if want_amax else df


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    N.check(_lib().p2l_adam_step(N.ptr(p), N.ptr(g), N.ptr(m), N.ptr(v), N.i64(p.numel()),
                                 N.f32(lr), N.f32(beta1), N.f32(beta2), N.f32(eps), int(step),
                                 N.stream()), 'adam_step')


def clamp_(p, lo, hi):
    N.check(_lib().p2l_clamp(N.ptr(p), N.i64(p.numel()), N.f32(lo), N.f32(hi), N.stream()),
            'clamp')


def bilinear_adjoint(wsrc, h, w):
    Bn, H, W = wsrc.shape
    wt = torch.empty(Bn, h, w, device=wsrc.device)
    N.check(_lib().p2l_bilinear_adjoint(N.ptr(wsrc), N.ptr(wt), Bn, H, W, h, w, N.stream()),
            'bilinear_adjoint')
    return wt


def lpips_normalize(f):
    Cc = f.shape[-1]
    nf = torch.empty_like(f)
    N.check(_lib().p2l_lpips_normalize(N.ptr(f), N.ptr(nf), N.i64(f.numel() // Cc), Cc,
                                       N.stream()), 'lpips_normalize')
    return nf


def lpips_tap_fwd(f, nft, lin, wt, wsum):
    Bn, P, Cc = f.shape[0], f.shape[1] * f.shape[2], f.shape[3]
    nblk = _lib().p2l_lpips_tap_nblk(P, Cc)
    part = torch.empty(Bn, nblk, device=f.device)
    N.check(_lib().p2l_lpips_tap_fwd(N.ptr(f), N.ptr(nft), N.i64(P * Cc), N.ptr(lin), N.ptr(wt),
                                     N.i64(P), N.ptr(part), Bn, P, Cc, N.stream()), 'lpips_tap_fwd')
    out = torch.empty(Bn, device=f.device)
    N.check(_lib().p2l_reduce_rows(N.ptr(part), N.ptr(out), Bn, nblk, N.f32(1.0), N.ptr(wsum), 0,
                                   N.stream()), 'reduce_rows')
    return out


def lpips_tap_bwd(f, nft, lin, wt, gscale):
    Bn, P, Cc = f.shape[0], f.shape[1] * f.shape[2], f.shape[3]
    df = torch.empty_like(f)
    N.check(_lib().p2l_lpips_tap_bwd(N.ptr(f), N.ptr(nft), N.i64(P * Cc), N.ptr(lin), N.ptr(wt),
                                     N.i64(P), N.ptr(gscale), N.ptr(df), Bn, P, Cc, N.stream()),
            'lpips_tap_bwd')
    return df


def conv_dgrad_arb(dy, wt_packed, B, H, W, Cin, Cout, taps, x, s, t, st_bstride, pool_sum=False,
                   skip=None, skip_C=0, skip_ups=False, subpix=False, wfmt=N.WFMT_F32, splitk=1,
                   keep=None, form=None):
    """fused input-gradient conv + backward of relu(x*s+t) (p2l_conv_dgrad_arb);
    H, W = resolution of dy; Cin = channels of dy, Cout = channels of x.
    splitk > 1: the split-K form, activation backward in the finish kernel
    (p2l_conv_dgrad_arb_ws).  keep: list that receives the partial-sum buffer (it must stay
    alive until p2l_arb_defer_flush when the finish is deferred)."""
    d = N.P2LConv()
    d.wfmt = wfmt
    d.form = DEFAULT_FORM if form is None else form
    d.B, d.H, d.W, d.Cin, d.Cout, d.taps = B, H, W, Cin, Cout, taps
    d.x_ld = Cin
    d.alpha = 1.0
    d.pool = N.POOL_SUM if (pool_sum and not subpix) else N.POOL_NONE
    d.ups = 3 if subpix else 0
    d.y_ld = d.yp_ld = d.n_store = Cout
    d.w_floats = wt_packed.numel()
    d.splitk = splitk
    if splitk > 1 and not _lib().p2l_conv_arb_split_fusable(C.byref(d)):
        splitk = d.splitk = 1        # (shapes that run in the Winograd form never split K)
    if splitk > 1:
        nblk = _lib().p2l_conv_arb_nblk_ws(C.byref(d))
    else:
        assert _lib().p2l_conv_arb_fusable(C.byref(d)) == 1
        nblk = _lib().p2l_conv_arb_nblk(C.byref(d))
    part = torch.empty(2 * B * nblk * Cout, device=dy.device)
    if keep is not None:
        keep.append(part)
    Ho, Wo = (H // 2, W // 2) if (pool_sum or subpix) else (H, W)
    dx = torch.empty(B, Ho, Wo, Cout, device=dy.device)
    ds = torch.empty(B, Cout, device=dy.device)
    dt = torch.empty(B, Cout, device=dy.device)
    a = N.P2LArb()
    a.x, a.x_ld = x.data_ptr(), x.shape[-1]
    a.s, a.t, a.st_bstride = s.data_ptr(), t.data_ptr(), st_bstride
    if skip is not None:
        a.skip, a.skip_ld, a.skip_C, a.skip_ups = skip.data_ptr(), skip.shape[-1], skip_C, int(skip_ups)
    a.ds, a.dt, a.dsdt_bstride = ds.data_ptr(), dt.data_ptr(), Cout
    a.partial = part.data_ptr()
    wsb = _lib().p2l_conv_workspace_bytes(C.byref(d))
    if splitk > 1 or wsb:
        ws = torch.empty(max(wsb // 4, 1), device=dy.device)
        N.check(_lib().p2l_conv_dgrad_arb_ws(C.byref(d), C.byref(a), N.ptr(dy), N.ptr(wt_packed),
                                             N.ptr(dx), N.ptr(ws), C.c_size_t(wsb), N.stream()),
                'conv_dgrad_arb_ws')
    else:
        N.check(_lib().p2l_conv_dgrad_arb(C.byref(d), C.byref(a), N.ptr(dy), N.ptr(wt_packed),
                                          N.ptr(dx), N.stream()), 'conv_dgrad_arb')
    return dx, ds, dt


# ---- LPIPS-AlexNet building blocks (csrc/p2l_alex.hip) ---------------------------------
def gconv(x, w_packed, B, Hi, Wi, Cin, Cout, K, stride, pad, bias=None, pro_s=None, pro_t=None,
          res=None, mask=None, relu=False):
    """generic NHWC conv (any size / stride / padding); x: [B,Hi,Wi,Cin] -> [B,Ho,Wo,Cout]"""
    Ho, Wo = (Hi + 2 * pad - K) // stride + 1, (Wi + 2 * pad - K) // stride + 1
    d = N.P2LGConv()
    d.B, d.Hi, d.Wi, d.Cin, d.Cout = B, Hi, Wi, Cin, Cout
    d.KH = d.KW = K
    d.stride, d.pad = stride, pad
    d.x_ld, d.y_ld, d.res_ld, d.mask_ld = x.shape[-1], Cout, Cout, Cout
    d.relu = int(relu)
    y = torch.empty(B, Ho, Wo, Cout, device=x.device)
    N.check(_lib().p2l_gconv_fwd(C.byref(d), N.ptr(x), N.ptr(w_packed), N.ptr(bias), N.ptr(pro_s),
                                 N.ptr(pro_t), N.ptr(res), N.ptr(mask), N.ptr(y), N.stream()),
            'gconv_fwd')
    return y


def maxpool3s2_fwd(x):
    B, Hi, Wi, Cc = x.shape
    y = torch.empty(B, (Hi - 3) // 2 + 1, (Wi - 3) // 2 + 1, Cc, device=x.device)
    N.check(_lib().p2l_maxpool3s2_fwd(N.ptr(x), N.ptr(y), B, Hi, Wi, Cc, N.stream()), 'maxpool3s2_fwd')
    return y


def maxpool3s2_bwd(x, gpooled, gtap=None):
    B, Hi, Wi, Cc = x.shape
    dx = torch.empty_like(x)
    N.check(_lib().p2l_maxpool3s2_bwd(N.ptr(x), N.ptr(gpooled), N.ptr(gtap), N.ptr(dx), B, Hi, Wi, Cc,
                                      N.stream()), 'maxpool3s2_bwd')
    return dx


def conv1_dgrad(g, w_t3, H, W, K, S, pad):
    B, Co = g.shape[0], g.shape[-1]
    d = torch.empty(B, H, W, 16, device=g.device)
    N.check(_lib().p2l_conv1_dgrad(N.ptr(g), N.ptr(w_t3), N.ptr(d), B, H, W, Co, K, S, pad, N.stream()),
            'conv1_dgrad')
    return d
