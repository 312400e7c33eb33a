"""End-to-end parity of the native plans (p2l_biggan_*, p2l_projloss_*) against
the CPU oracle (oracle/biggan_ref.py, oracle/lpips_ref.py) on identical seeded
weights / latents / targets.  Bar (BASELINE.json north_star): per-pixel
|delta| < 1e-3, loss |delta| < 1e-3, identical ranking of the candidates."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PIX_TOL = 1e-3
LOSS_TOL = 1e-3
# Gradients of this network are discontinuous in the activations (ReLU masks,
# max-pool / soft-max arg-max): the CPU oracle ITSELF moves by relL2 ~3e-3 (dz, dc)
# between fp32 and fp64 (measured, DESIGN.md "numerics").  Gradient parity is
# therefore asserted as relative L2 error < 1e-2 plus cosine > 0.9999, not as a
# max-norm bound.
GRAD_REL_L2 = 1e-2
GRAD_COS = 0.9999


def grad_close(got, ref, name):
    got = got.detach().cpu().double().flatten()
    ref = ref.detach().cpu().double().flatten()
    rel = ((got - ref).norm() / ref.norm()).item()
    cos = (got @ ref / (got.norm() * ref.norm())).item()
    assert rel < GRAD_REL_L2 and cos > GRAD_COS, '%s: relL2 %g cos %.8f' % (name, rel, cos)


@pytest.fixture(scope='module')
def setup(dev):
    from pix2latent_amd.utils import synthetic as S
    from pix2latent_amd.model.biggan import BigGAN
    import pix2latent_amd.loss_functions as LF
    W = S.biggan_weights(0)
    Wv = S.lpips_vgg_weights(1)
    model = BigGAN(weights=W)
    loss = LF.ProjectionLoss(lpips_net='vgg', weights=Wv)
    g = torch.Generator().manual_seed(2)
    B = 3
    z = torch.fmod(torch.randn(B, 128, generator=g), 2.0)
    c = (0.05 * torch.randn(1, 128, generator=g)).repeat(B, 1) + 0.01 * torch.randn(B, 128, generator=g)
    target = S.synthetic_target(256, 1).unsqueeze(0).repeat(B, 1, 1, 1)
    weight = S.synthetic_weight_mask(256).unsqueeze(0).repeat(B, 1, 1, 1)
    return dict(W=W, Wv=Wv, model=model, loss=loss, z=z, c=c, target=target, weight=weight, B=B)


@pytest.fixture(scope='module')
def oracle_run(setup):
    """one CPU oracle forward+backward shared by the tests below."""
    from oracle import biggan_ref as R, lpips_ref as L
    s = setup
    z = s['z'].clone().requires_grad_(True)
    c = s['c'].clone().requires_grad_(True)
    out, inter = R.biggan_forward(s['W'], z, c, return_intermediates=True)
    rec = L.reconstruction_loss(out, s['target'], s['weight'])
    per = L.perceptual_loss(s['Wv'], out, s['target'], s['weight'])
    loss = rec + 10 * per
    out.retain_grad()
    loss.mean().backward()          # closure.py:58
    return dict(out=out.detach(), inter={k: v.detach() for k, v in inter.items()},
                rec=rec.detach(), per=per.detach(), loss=loss.detach(),
                dz=z.grad, dc=c.grad, dout=out.grad)


def test_generator_forward_pixels(setup, oracle_run, dev):
    s = setup
    with torch.no_grad():
        out = s['model'](z=s['z'].to(dev), c=s['c'].to(dev))
    torch.cuda.synchronize()
    # layer-by-layer first: localises a failure
    table_len = 13
    for li in range(table_len):
        act = s['model'].saved_activation(0, li).cpu().permute(0, 3, 1, 2)
        ref = oracle_run['inter']['layer%d' % li]
        err = (act - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        assert err < 1e-4, 'layer %d rel err %g' % (li, err)
    d = (out.cpu() - oracle_run['out']).abs().max().item()
    assert d < PIX_TOL, 'per-pixel |delta| = %g' % d


def test_loss_forward(setup, oracle_run, dev):
    s = setup
    out = oracle_run['out'].to(dev)
    loss = s['loss'](out, s['target'].to(dev), s['weight'].to(dev))
    torch.cuda.synchronize()
    eng = s['loss']._engine
    assert (eng.last_l1.cpu() - oracle_run['rec']).abs().max().item() < LOSS_TOL
    assert (eng.last_lpips.cpu() - oracle_run['per']).abs().max().item() < LOSS_TOL / 10
    assert (loss.cpu() - oracle_run['loss']).abs().max().item() < LOSS_TOL


def test_loss_backward(setup, oracle_run, dev):
    s = setup
    out = oracle_run['out'].to(dev).requires_grad_(True)
    loss = s['loss'](out, s['target'].to(dev), s['weight'].to(dev))
    loss.mean().backward()
    torch.cuda.synchronize()
    grad_close(out.grad, oracle_run['dout'], 'd loss / d out')


def test_full_step_gradients_and_ranking(setup, oracle_run, dev):
    s = setup
    z = s['z'].to(dev).requires_grad_(True)
    c = s['c'].to(dev).requires_grad_(True)
    out = s['model'](z=z, c=c)
    loss = s['loss'](out, s['target'].to(dev), s['weight'].to(dev))
    loss.mean().backward()
    torch.cuda.synchronize()
    assert (loss.detach().cpu() - oracle_run['loss']).abs().max().item() < LOSS_TOL
    assert np.array_equal(np.argsort(loss.detach().cpu().numpy()),
                          np.argsort(oracle_run['loss'].numpy())), 'CMA ranking differs'
    grad_close(z.grad, oracle_run['dz'], 'dz')
    grad_close(c.grad, oracle_run['dc'], 'dc')


def test_l1_only_config1(setup, oracle_run, dev):
    """BASELINE config 1: invert_biggan_adam, num_samples=1, L1 loss only."""
    import pix2latent_amd.loss_functions as LF
    from oracle import biggan_ref as R, lpips_ref as L
    s = setup
    z = s['z'][:1].clone().requires_grad_(True)
    c = s['c'][:1].clone().requires_grad_(True)
    out = R.biggan_forward(s['W'], z, c)
    ref = L.reconstruction_loss(out, s['target'][:1], s['weight'][:1])
    ref.mean().backward()
    zd = s['z'][:1].to(dev).requires_grad_(True)
    cd = s['c'][:1].to(dev).requires_grad_(True)
    rl = LF.ReconstructionLoss()
    got = rl(s['model'](z=zd, c=cd), s['target'][:1].to(dev), s['weight'][:1].to(dev))
    got.mean().backward()
    assert (got.detach().cpu() - ref.detach()).abs().max().item() < LOSS_TOL
    grad_close(zd.grad, z.grad, 'dz (L1 only)')
