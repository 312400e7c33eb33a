"""Model-level gradient parity with the DISCRETE decisions held fixed.

The gradients of generator + LPIPS are piecewise: ~10^8 ReLU signs and max-pool winners.  Two
arithmetics flip a handful of them differently and every flip is a step in the gradient, which
is why the free-running comparisons (tests/test_biggan_grad64_gpu.py, smoke()) need a noise floor
of ~6e-4.  Here the decisions of the NATIVE run are read back from its workspaces (ReLU inputs and
folded CBN affines of all 49 generator ReLUs, the two attention max-pools, the sign of every
pixel difference of the L1 term, the 13 VGG ReLUs and 4 VGG max-pools) and REPLAYED in the CPU oracle (oracle/masks.py), in fp64 and in fp32.  With the
decisions equal, what separates native from the fp64 oracle is arithmetic alone -- and it has to be
as small as what separates the fp32 oracle from the fp64 one (no clamp, no factor 3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1500)
def test_gradients_with_the_native_decisions_replayed(dev):
    from pix2latent_amd.utils import synthetic as S
    from pix2latent_amd.model.biggan import BigGAN
    import pix2latent_amd.loss_functions as LF
    from oracle import biggan_ref as R, lpips_ref as L
    from oracle.masks import DecisionTape
    W, Wv = S.biggan_weights(0), S.lpips_vgg_weights(1)
    model = BigGAN(weights=W, device=dev)
    loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=Wv, device=dev)
    n = 3
    g = torch.Generator().manual_seed(2)                   # (the smoke() probe: its candidate 0 sits
    z = torch.fmod(torch.randn(n, 128, generator=g), 2.0)  #  next to a mask boundary)
    c = (0.05 * torch.randn(1, 128, generator=g)).repeat(n, 1)
    target = S.synthetic_target(256, 1).unsqueeze(0).repeat(n, 1, 1, 1)
    weight = S.synthetic_weight_mask(256).unsqueeze(0).repeat(n, 1, 1, 1)
    zd, cd = z.to(dev).requires_grad_(True), c.to(dev).requires_grad_(True)
    out = model(z=zd, c=cd)
    loss = loss_fn(out, target.to(dev), weight.to(dev))
    loss.mean().backward()
    torch.cuda.synchronize()
    import os
    from oracle.replay import native_decisions
    diag = [] if os.environ.get('P2L_DIAG') else None
    items = native_decisions(model, loss_fn, W, n, dev, out, target.to(dev), diag)
    for row in diag or []:
        if any(row[1]) or any(row[2]) or any(row[3]):
            print('ambiguous ReLU inputs', row)
    assert len(items) == 48 + 2 + 1 + 1 + 13 + 4

    def oracle(dtype, tape):
        Wd = {k: v.to(dtype) for k, v in W.items()}
        Wvd = {k: v.to(dtype) for k, v in Wv.items()}
        zr = z.to(dtype).clone().requires_grad_(True)
        cr = c.to(dtype).clone().requires_grad_(True)
        o = R.biggan_forward(Wd, zr, cr, tape=tape)
        l = L.projection_loss(Wvd, o, target.to(dtype), weight.to(dtype), tape=tape)
        l.mean().backward()
        assert tape is None or tape.pos == len(tape.items)
        return o.detach(), l.detach(), zr.grad.double(), cr.grad.double()
    o64, l64, dz64, dc64 = oracle(torch.float64, DecisionTape(replay=items))
    o32, l32, dz32, dc32 = oracle(torch.float32, DecisionTape(replay=items))
    _, _, dz64_free, _ = oracle(torch.float64, None)

    def rel(a, b):
        return ((a - b).norm(dim=1) / b.norm(dim=1))
    nat_z, nat_c = rel(zd.grad.cpu().double(), dz64), rel(cd.grad.cpu().double(), dc64)
    f32_z, f32_c = rel(dz32, dz64), rel(dc32, dc64)
    free = rel(zd.grad.cpu().double(), dz64_free)
    print('fixed decisions: dz native %s  fp32 oracle %s | dc native %s  fp32 oracle %s | free-running dz native %s'
          % (['%.2e' % v for v in nat_z.tolist()], ['%.2e' % v for v in f32_z.tolist()],
             ['%.2e' % v for v in nat_c.tolist()], ['%.2e' % v for v in f32_c.tolist()],
             ['%.2e' % v for v in free.tolist()]))
    # the image and the loss with the decisions replayed: the replay multiplies by the native masks,
    # so the oracle's pixels equal the native ones to rounding
    assert (out.detach().cpu().double() - o64).abs().max().item() < 1e-4
    assert (loss.detach().cpu().double() - l64).abs().max().item() < 1e-5
    # arithmetic alone: every candidate, no noise floor
    for nat, f32 in ((nat_z, f32_z), (nat_c, f32_c)):
        assert (nat <= 1.5 * f32 + 2e-5).all(), (nat, f32)
    assert nat_z.max().item() < 1e-4 and nat_c.max().item() < 1e-4
