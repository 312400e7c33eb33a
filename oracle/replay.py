"""ORACLE (test infrastructure): reads the discrete decisions of a NATIVE forward pass back from
the product's workspaces, in the order the CPU oracle takes them (oracle/masks.py DecisionTape):
49 generator ReLUs (input tensor x and the folded CBN affine s, t: sign of x*s + t), the two
attention max-pools, the sign of (out - target) of the L1 term, 13 VGG ReLUs and 4 VGG max-pools.
Uses the test hooks p2l_biggan_ws_lookup / p2l_projloss_ws_lookup.

Only tests/ and __graft_entry__.smoke() may import this module."""
import ctypes as C

import torch


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def native_decisions(model, loss_fn, W, B, dev, out, target, diag=None):
    """tape items (oracle call order: generator, L1 term, VGG of the generated image) read back
    from the workspaces of the LAST native forward of `model` / `loss_fn` on B candidates;
    `out` = the image the native loss saw, `target` = its target (device tensors, NCHW)"""
    from pix2latent_amd import _native as N
    from oracle import biggan_ref as R
    from oracle.masks import winner_mask
    items = []
    d = model._desc
    s_all, t_all = model.saved_activation(2), model.saved_activation(3)      # [B,1,1,cbn_total]
    s_all, t_all = s_all.view(B, -1), t_all.view(B, -1)
    prev = model.saved_activation(1)                                          # gen_z output, NHWC
    bi = 0
    for i, spec in enumerate(R.layer_table()):
        p = 'generator.layers.%d' % i
        if spec[0] == 'attn':
            for what, nm in ((8, '.phi'), (9, '.g')):
                items.append((p + nm, 'pool', winner_mask(_nchw(model.saved_activation(what))).cpu()))
        else:
            for k in range(4):
                x = prev if k == 0 else model.saved_activation(7, 3 * bi + (k - 1))
                c = x.shape[-1]
                off = d.blocks[bi].cbn_off[k]
                sv, tv = s_all[:, off:off + c].view(B, 1, 1, c), t_all[:, off:off + c].view(B, 1, 1, c)
                # the kernels evaluate x*s + t as ONE fused multiply-add (hipcc contracts it in every
                # prologue and in the activation-backward epilogues): its sign is the sign of the exact
                # value.  An unfused fp32 evaluation differs for the one-in-10^7 pre-activation whose
                # product rounds across -t (heavy-tailed weights, seed 7: one element of 37.7 M at the
                # input of block 10, latent gradient 8.5e-5 off when that decision is replayed wrongly)
                pre = torch.addcmul(tv.double(), x.double(), sv.double())
                if diag is not None:
                    pre = torch.addcmul(tv, x, sv)
                if diag is not None:
                    pre_b = (x * sv) + tv
                    pre_c = x.double() * sv.double() + tv.double()
                    amb = (pre_c.abs() < 4e-7 * (x.double() * sv.double()).abs() + 4e-7 * tv.double().abs())
                    diag.append(('%s.bn_%d' % (p, k), [int(v) for v in ((pre > 0) != (pre_b > 0)).flatten(1).sum(1)],
                                 [int(v) for v in ((pre > 0) != (pre_c > 0)).flatten(1).sum(1)],
                                 [int(v) for v in amb.flatten(1).sum(1)]))
                    pre = pre_c
                items.append(('%s.bn_%d' % (p, k), 'relu', _nchw(pre > 0).cpu()))
            bi += 1
        prev = model.saved_activation(0, i)
    # tail: unconditional BN folded as the plan folds it
    # (the NATIVE fold, to the bit: t = b - m * s with the rounded s -- the algebraically equal
    #  b - m * w / sqrt(v + eps) is an ulp away, and 8.4 M pre-activations per image put one or two of
    #  them inside that ulp of zero: a decision the oracle would then replay differently from the run)
    s, t = getattr(model, '_tail_s', None), getattr(model, '_tail_t', None)
    if s is None or t is None:
        mean, var = R._bn_stats(W['generator.bn.running_means'], W['generator.bn.running_vars'], 1.0)
        s = (W['generator.bn.weight'] / torch.sqrt(var + R.BN_EPS))
        t = (W['generator.bn.bias'] - mean * s).to(dev)
        s = s.to(dev)
    tail_pre = torch.addcmul(t.double(), prev.double(), s.double())
    if diag is not None:
        amb = tail_pre.double().abs() < 4e-7 * ((prev.double() * s.double()).abs() + t.double().abs())
        diag.append(('generator.bn', [0] * B, [0] * B, [int(v) for v in amb.flatten(1).sum(1)]))
    items.append(('generator.bn', 'relu', _nchw(tail_pre > 0).cpu()))
    return items + loss_decisions(loss_fn, B, out, target)


def loss_decisions(loss_fn, B, out, target):
    """the decisions of the LAST native ProjectionLoss(VGG) forward on B candidates, in oracle/lpips_ref.py's
    order: the sign of (out - target) of the L1 term, then the 13 VGG ReLUs and 4 max-pools of the pass over
    the generated image (read through p2l_projloss_ws_lookup at the resolution of `out`: 256^2 BigGAN,
    512^2 / 1024^2 StyleGAN2)"""
    from pix2latent_amd import _native as N
    from oracle.masks import winner_mask
    items = [('l1', 'sign', torch.sign(out.detach() - target).cpu())]
    eng = loss_fn._engine
    H, Wd = int(out.shape[2]), int(out.shape[3])
    lib = N.lib()

    def vgg_y(idx):
        off, shape = C.c_size_t(0), (C.c_int32 * 4)()
        N.check(lib.p2l_projloss_ws_lookup(B, H, Wd, idx, C.byref(off), shape), 'p2l_projloss_ws_lookup')
        n = shape[0] * shape[1] * shape[2] * shape[3]
        return eng.ws[off.value:off.value + n].view(*list(shape))
    for ci in range(13):
        if ci in (2, 4, 7, 10):
            items.append(('vgg.pool%d' % ci, 'pool', winner_mask(_nchw(vgg_y(ci - 1))).cpu()))
        items.append(('vgg.conv%d' % ci, 'relu', _nchw(vgg_y(ci) > 0).cpu()))
    return items


def squeeze_decisions(loss_fn, B, out, target, with_l1=True):
    """the decisions of the LAST native ProjectionLoss / PerceptualLoss (SqueezeNet) forward on B candidates in
    oracle/lpips_ref.py squeeze_features' order: conv0's ReLU, then per fire (the pool winners in front of fires
    0, 2, 4, recomputed from the saved post-ReLU tensor with torch's first-maximum rule -- the rule
    p2l_maxpool3s2_bwd applies), the squeeze ReLU and the two expand ReLUs (p2l_sqzloss_ws_lookup)"""
    from pix2latent_amd import _native as N
    from oracle.masks import pool3s2_winners
    from oracle.lpips_ref import SQZ_FIRES, SQZ_POOL_BEFORE_FIRE
    items = [('l1', 'sign', torch.sign(out.detach() - target).cpu())] if with_l1 else []
    eng = loss_fn._engine
    H, Wd = int(out.shape[2]), int(out.shape[3])
    lib = N.lib()

    def act(idx):
        off, shape = C.c_size_t(0), (C.c_int32 * 4)()
        N.check(lib.p2l_sqzloss_ws_lookup(B, H, Wd, idx, C.byref(off), shape), 'p2l_sqzloss_ws_lookup')
        n = shape[0] * shape[1] * shape[2] * shape[3]
        return _nchw(eng.ws[off.value:off.value + n].view(*list(shape))).cpu()
    x = act(0)
    items.append(('squeeze.conv0', 'relu', x > 0))
    for i, (_, _, ex) in enumerate(SQZ_FIRES):
        if i in SQZ_POOL_BEFORE_FIRE:
            items.append(('squeeze.pool%d' % i, 'pool3', pool3s2_winners(x)))
        p = 'squeeze.fire%d.' % i
        items.append((p + 'squeeze', 'relu', act(1 + i) > 0))
        x = act(9 + i)
        items.append((p + 'expand1x1', 'relu', x[:, :ex] > 0))
        items.append((p + 'expand3x3', 'relu', x[:, ex:] > 0))
    return items


def sg2_decisions(model, B, out, with_mapping):
    """StyleGAN2: the leaky-ReLU decisions of the LAST native forward of `model` on B candidates in the
    order oracle/stylegan2_ref.py takes them -- the 8 mapping layers (when the run went through the mapping),
    the styled convs (signs of the saved post-activation outputs, read through p2l_sg2_ws_lookup: the
    activation keeps the sign of its pre-activation) -- and which pixels the final clamp let through
    (`out` = the image the native run returned, NCHW)."""
    items = []
    if with_mapping:
        acts = model._last_acts                      # [9][B][D]: slot i + 1 = output of mapping layer i
        for i in range(8):
            items.append((acts[i + 1] > 0).cpu())
    for l in range(model._desc.n_conv):
        items.append(_nchw(model.saved_activation(l, B) > 0).cpu())
    items.append((out.detach().abs() < 1.0).cpu())
    return items
