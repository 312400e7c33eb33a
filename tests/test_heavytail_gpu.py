"""fp16 x 2 is block floating point: per-image / per-layer powers of two, maxima handed between
launches as BOUNDS.  `utils/synthetic.py` draws well-conditioned weights (max / typical ~ 5), so
the model-level parity tests never showed that arithmetic a trained network's statistics: BN
variances over orders of magnitude, a few channels 10^2 - 10^3 x the median (VERDICT round 3, weak
#3).  Here the same pipeline runs on `heavy_tailed` weights three ways --

  * the default kernels (fp16 x 2 where the layer shape takes them, bf16 x 3 elsewhere),
  * P2L_CONV_WFMT=f32: every conv on the exact-fp32 MFMA,
  * the CPU oracle in fp64 (truth) and fp32 --

and both native arithmetics must meet the north_star's bars (|d pix| < 1e-3, |d loss| < 1e-3)
and, with the native run's discrete decisions replayed in the oracle, the fixed-decision gradient
bound of tests/test_fixed_mask_grad_gpu.py (native <= 1.5 x fp32-oracle + 2e-5 per candidate).
Reference call sites served: pix2latent/model/biggan.py:58, pix2latent/loss_functions.py:142."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inner_tail(model, n):
    """max / median of |x| over the inner activations the 3x3 / 1x1 kernels read (inputs of bn_1,
    bn_2, bn_3 of every GenBlock): how heavy the tails really are"""
    worst = 0.0
    for li in range(12 * 3):
        a = model.saved_activation(7, li).abs()
        worst = max(worst, (a.amax() / a.median().clamp_min(1e-30)).item())
    return worst


# seeds: the two draws on which the replayed oracle once took a decision the run had not (CHANGELOG, round 4)
@pytest.mark.timeout(2400)
@pytest.mark.parametrize('seed', [2, 7])
def test_heavy_tailed_weights_both_arithmetics(dev, monkeypatch, seed):
    from pix2latent_amd import _native as N
    from pix2latent_amd.utils import synthetic as S
    from pix2latent_amd.model.biggan import BigGAN
    import pix2latent_amd.loss_functions as LF
    from oracle import biggan_ref as R, lpips_ref as L
    from oracle.masks import DecisionTape
    from oracle.replay import native_decisions
    W = S.heavy_tailed(S.biggan_weights(0))
    Wv = S.heavy_tailed_vgg(S.lpips_vgg_weights(1))
    n = 2
    g = torch.Generator().manual_seed(seed)
    z = torch.fmod(torch.randn(n, 128, generator=g), 2.0)
    c = (0.05 * torch.randn(1, 128, generator=g)).repeat(n, 1)
    target = S.synthetic_target(256, 1).unsqueeze(0).repeat(n, 1, 1, 1)
    weight = S.synthetic_weight_mask(256).unsqueeze(0).repeat(n, 1, 1, 1)

    def native(wfmt_env):
        if wfmt_env:
            monkeypatch.setenv('P2L_CONV_WFMT', wfmt_env)
        else:
            monkeypatch.delenv('P2L_CONV_WFMT', raising=False)
        model = BigGAN(weights=W, device=dev)
        loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=Wv, device=dev)
        zd, cd = z.to(dev).requires_grad_(True), c.to(dev).requires_grad_(True)
        out = model(z=zd, c=cd)
        loss = loss_fn(out, target.to(dev), weight.to(dev))
        loss.mean().backward()
        torch.cuda.synchronize()
        items = native_decisions(model, loss_fn, W, n, dev, out, target.to(dev), None)
        return dict(out=out.detach().cpu(), loss=loss.detach().cpu(), dz=zd.grad.cpu().double(),
                    dc=cd.grad.cpu().double(), items=items, tail=_inner_tail(model, n),
                    wfmt=model._wfmt)

    def oracle(dtype, tape):
        Wd = {k: v.to(dtype) for k, v in W.items()}
        Wvd = {k: v.to(dtype) for k, v in Wv.items()}
        zr = z.to(dtype).clone().requires_grad_(True)
        cr = c.to(dtype).clone().requires_grad_(True)
        o = R.biggan_forward(Wd, zr, cr, tape=tape)
        l = L.projection_loss(Wvd, o, target.to(dtype), weight.to(dtype), tape=tape)
        l.mean().backward()
        return o.detach().double(), l.detach().double(), zr.grad.double(), cr.grad.double()

    def rel(a, b):
        return (a - b).norm(dim=1) / b.norm(dim=1)

    o64, l64, _, _ = oracle(torch.float64, None)            # free-running truth: pixels and loss
    runs = {'default': native(None), 'f32': native('f32')}
    assert runs['default']['wfmt'] == N.WFMT_BF16X3W and runs['f32']['wfmt'] == N.WFMT_F32
    assert runs['default']['tail'] > 200, 'the inner activations are not heavy-tailed: %g' % runs['default']['tail']
    report = []
    for name, r in runs.items():
        dpix = (r['out'].double() - o64).abs().max().item()
        dloss = (r['loss'].double() - l64).abs().max().item()
        # arithmetic alone: this run's own decisions replayed in the fp64 and the fp32 oracle
        _, _, dz64, dc64 = oracle(torch.float64, DecisionTape(replay=r['items']))
        _, _, dz32, dc32 = oracle(torch.float32, DecisionTape(replay=r['items']))
        nat_z, nat_c = rel(r['dz'], dz64), rel(r['dc'], dc64)
        f32_z, f32_c = rel(dz32, dz64), rel(dc32, dc64)
        report.append('%s: |dpix| %.2e |dloss| %.2e  dz native %s fp32-oracle %s  dc native %s fp32-oracle %s' % (
            name, dpix, dloss, ['%.2e' % v for v in nat_z.tolist()], ['%.2e' % v for v in f32_z.tolist()],
            ['%.2e' % v for v in nat_c.tolist()], ['%.2e' % v for v in f32_c.tolist()]))
        r.update(dpix=dpix, dloss=dloss, nat_z=nat_z, nat_c=nat_c, f32_z=f32_z, f32_c=f32_c)
    print('heavy-tailed weights (inner max/median %.0f):\n  ' % runs['default']['tail'] + '\n  '.join(report))
    for name, r in runs.items():
        assert r['dpix'] < 1e-3 and r['dloss'] < 1e-3, (name, r['dpix'], r['dloss'])
        for nat, f32 in ((r['nat_z'], r['f32_z']), (r['nat_c'], r['f32_c'])):
            assert (nat <= 1.5 * f32 + 2e-5).all(), (name, nat, f32)
    # the two native arithmetics against each other: same image to fp32 rounding of the products
    assert (runs['default']['out'] - runs['f32']['out']).abs().max().item() < 2e-4


def test_outlier_channels_error_on_the_ordinary_outputs(dev):
    """kernel level, measured where it hurts: an image whose input has 2 % outlier CHANNELS (x 1e3,
    met by weights / 1e3: every product is O(1)) -- the error of the fp16 x 2 Winograd and 1x1
    kernels on every output, relative to the condition-aware fp32 bound sum |x||w| of THAT output
    (not to the image's largest output).  An fp32 dot product is within ~K^(1/2) 2^-24 of it."""
    import math
    import torch.nn.functional as F
    from pix2latent_amd import _native as N, ops as O
    g = torch.Generator().manual_seed(11)
    H, Cin, Cout, B = 32, 128, 64, 2
    x = torch.randn(B, Cin, H, H, generator=g).abs()
    out_ch = torch.randperm(Cin, generator=g)[:3]
    x[:, out_ch] *= 1e3
    for taps, wf, form in ((9, 2, N.FORM_WINO_ANY), (1, 3, N.FORM_AUTO)):
        k = 3 if taps == 9 else 1
        w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
        w[:, out_ch] /= 1e3
        ref = F.conv2d(x.double(), w.double(), None, padding=k // 2)
        cond = F.conv2d(x.double().abs(), w.double().abs(), None, padding=k // 2)
        O.DEFAULT_FORM = form
        try:
            wp = O.pack_conv_weight(w.to(dev), taps, Cout, Cin, wfmt=wf)
            xs = x.permute(0, 2, 3, 1).contiguous().to(dev)
            if taps == 1:
                # the 1x1 kernel takes fp16 x 2 only with handed-over maxima: produce x by a conv
                # launch that leaves them (identity 1x1 through the same API)
                eye = torch.eye(Cin).view(Cin, Cin, 1, 1)
                wi = O.pack_conv_weight(eye.to(dev), 1, Cin, Cin, wfmt=3)
                xs, _, (amax, _) = O.conv(xs, wi, B, H, H, Cin, Cin, 1, wfmt=3, want_amax=True)
                y, _ = O.conv(xs, wp, B, H, H, Cin, Cout, 1, wfmt=3, amax_in=amax)
                ref = F.conv2d(xs.permute(0, 3, 1, 2).cpu().double(), w.double())
                cond = F.conv2d(xs.permute(0, 3, 1, 2).cpu().double().abs(), w.double().abs())
            else:
                y, _ = O.conv(xs, wp, B, H, H, Cin, Cout, 9, wfmt=wf)
        finally:
            O.DEFAULT_FORM = N.FORM_AUTO
        y = y.permute(0, 3, 1, 2).cpu().double()
        err = ((y - ref).abs() / cond).max().item()
        assert err < 2e-6, (taps, err)        # ~ 30 x 2^-24: fp32-grade on EVERY output


@pytest.mark.parametrize('H', [32, 8])
def test_maxima_with_the_readers_affine_applied(dev, H):
    """P2LAmax.next_s / next_t / in_applied.  The bound max|s| max|x| + max|t| of a fused prologue is
    loose when the large |s| and the large |x| sit in different channels -- three channels x 1e3 that
    the affine scales by 1e-3, three ordinary ones it scales by 30: x 1e3 here, and the heavy-tailed
    model weights above measure x 50 - 160 on every CBN layer -- and fp16 x 2 has 2^18 of range
    below the scaled maximum before an operand's low piece goes subnormal.  A producer that knows the
    reader's affine records max|y*s + t| itself: exactly that maximum (to the rounding of one FMA),
    from every kernel family that writes maxima, and every fp16 x 2 reader is fp32-grade on EVERY
    output (relative to that output's own sum |x||w|).  With the bound they still are (2^10 of slack
    inside 2^18, here and in the model above alike: tools/heavytail_ab.py with -DP2L_AB_NO_NEXT_AFFINE);
    the applied maxima are margin for statistics this repo has not seen, and two reductions less in
    every reader's prologue."""
    import math
    import torch.nn.functional as F
    from pix2latent_amd import _native as N, ops as O
    g = torch.Generator().manual_seed(21 + H)
    # (8x8: deep enough for K slices -- the finish kernel is the producer there; an unsplit tile that
    #  spans two images writes no maxima.  The slice count follows the layer shape since round 5.)
    B, C0, C1, C2 = 2, (64 if H == 32 else 256), 128, 64
    x0 = torch.randn(B, C0, H, H, generator=g)
    big = torch.randperm(C1, generator=g)[:6]
    w1 = torch.randn(C1, C0, 3, 3, generator=g) / math.sqrt(9 * C0)
    w1[big[:3]] *= 1e3                                     # three outlier channels in y ...
    s = 0.5 + torch.rand(B, C1, generator=g)
    s[:, big[:3]] *= 1e-3                                  # ... that the reader's affine scales back,
    s[:, big[3:]] *= 30.0                                  # and three ordinary ones it blows up
    t = 0.3 * torch.randn(B, C1, generator=g)
    sd, td = s.to(dev), t.to(dev)
    xin = F.relu(F.conv2d(x0.double(), w1.double(), None, padding=1) * s.double().view(B, C1, 1, 1) + t.double().view(B, C1, 1, 1))

    def producers():
        # (name, y, raw maxima, applied maxima) from the kernel families that write maxima
        for name, wf, form in (('winograd', 2, N.FORM_WINO_ANY), ('direct fp16x2', 2, N.FORM_NO_WINO), ('direct bf16x3', 1, N.FORM_AUTO)):
            O.DEFAULT_FORM = form
            try:
                wp = O.pack_conv_weight(w1.to(dev), 9, C1, C0, wfmt=wf)
                y, _, (raw, _) = O.conv(nhwc(x0), wp, B, H, H, C0, C1, 9, wfmt=wf, want_amax=True)
                y2, _, (app, _) = O.conv(nhwc(x0), wp, B, H, H, C0, C1, 9, wfmt=wf, want_amax=True, amax_next=(sd, td, C1))
            finally:
                O.DEFAULT_FORM = N.FORM_AUTO
            if raw is None:
                continue
            assert torch.equal(y, y2)
            yield name, y, raw, app

    def nhwc(v):
        return v.permute(0, 2, 3, 1).contiguous().to(dev)

    seen = []
    for name, y, raw, app in producers():
        seen.append(name)
        assert torch.equal(raw.amax(dim=1), y.abs().amax(dim=(1, 2, 3)))
        want = (y * sd.view(B, 1, 1, C1) + td.view(B, 1, 1, C1)).abs().amax(dim=(1, 2, 3))
        assert ((app.amax(dim=1) - want).abs() <= 1e-6 * want).all(), (name, app.amax(dim=1), want)
        loose = (sd.abs().amax(1) * raw.amax(1) + td.abs().amax(1)) / want
        assert (loose > 100).all(), loose                   # the case this test is about
        yin = F.relu(y.permute(0, 3, 1, 2).cpu().double() * s.double().view(B, C1, 1, 1) + t.double().view(B, C1, 1, 1))
        for rname, taps, wf, form in (('winograd', 9, 2, N.FORM_WINO_ANY), ('direct', 9, 2, N.FORM_NO_WINO), ('pointwise', 1, 3, N.FORM_AUTO)):
            k = 3 if taps == 9 else 1
            w2 = torch.randn(C2, C1, k, k, generator=g) / math.sqrt(C1 * k * k)
            ref = F.conv2d(yin, w2.double(), None, padding=k // 2)
            cond = F.conv2d(yin.abs(), w2.double().abs(), None, padding=k // 2)
            O.DEFAULT_FORM = form
            try:
                wp2 = O.pack_conv_weight(w2.to(dev), taps, C2, C1, wfmt=wf)
                kw = dict(wfmt=wf, pro=N.PRO_AFFINE_RELU, pro_s=sd, pro_t=td, pro_bstride=C1)
                za, _ = O.conv(y, wp2, B, H, H, C1, C2, taps, amax_in=app, amax_applied=True, **kw)
                zb, _ = O.conv(y, wp2, B, H, H, C1, C2, taps, amax_in=raw, **kw)
                # maxima recorded with an affine say nothing about the raw tensor: a reader WITHOUT
                # prologue must not take them (the launch falls back to its own pass / bf16 x 3)
                zn0, _ = O.conv(y, wp2, B, H, H, C1, C2, taps, wfmt=wf)
                zn1, _ = O.conv(y, wp2, B, H, H, C1, C2, taps, wfmt=wf, amax_in=app, amax_applied=True)
            finally:
                O.DEFAULT_FORM = N.FORM_AUTO
            assert torch.equal(zn0, zn1), (name, rname)
            ea = ((za.permute(0, 3, 1, 2).cpu().double() - ref).abs() / cond).max().item()
            eb = ((zb.permute(0, 3, 1, 2).cpu().double() - ref).abs() / cond).max().item()
            print('%dx%d %s -> %s: error / sum|x||w|  applied %.1e  bound %.1e' % (H, H, name, rname, ea, eb))
            assert ea < 2e-6, (name, rname, ea)
            assert ea <= eb * 1.5 + 1e-7, (name, rname, ea, eb)
    assert len(seen) >= 2, seen
