"""BASELINE.json configurations C4 / C5 at FULL size (StyleGAN2-cars 512^2 with the
examples' rows-64:-64 loss mask, StyleGAN2-FFHQ 1024^2), where the CPU oracle is too
slow: size-independent properties of generator + ProjectionLoss + backward.

  * finite, bit-reproducible (fixed-order reductions: CMA ranks candidates on these)
  * batch-composition independence: a candidate's loss does not depend on which chunk it
    is evaluated in (ragged last chunk of 5 vs chunk of 9: reference chunking 9,9,9,5)
  * the analytic gradient matches a central finite difference of the loss along the
    gradient direction (fp32, piecewise-linear activations -> 5 % tolerance)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(dev, size, model_name):
    import warnings
    warnings.simplefilter('ignore')
    from pix2latent_amd.utils import synthetic as S
    from pix2latent_amd.model.stylegan2 import StyleGAN2
    import pix2latent_amd.loss_functions as LF
    model = StyleGAN2(model=model_name, search='z', device=dev)
    assert model.im_res == size
    loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=S.lpips_vgg_weights(1), device=dev)
    target = S.synthetic_target(size, 1).to(dev)
    weight = torch.ones(3, size, size, device=dev)
    loss_mask = torch.zeros(3, size, size, device=dev)
    loss_mask[:, size // 8:-size // 8, :] += 1.0           # examples/invert_stylegan2_cars_*.py:38-39
    g = torch.Generator().manual_seed(11)
    noises = [torch.randn(1, 1, s[2], s[3], generator=g).to(dev) for s in model.noise_shape]

    def loss_of(z):
        B = z.size(0)
        out = model.forward_z(z, noises=[n.expand(B, -1, -1, -1).contiguous() for n in noises])
        rep = lambda t: t.unsqueeze(0).expand(B, -1, -1, -1).contiguous()
        return loss_fn(out, rep(target), rep(weight), rep(loss_mask))
    return loss_of, g


def _check(loss_of, g, dev, B_big, B_small, fd_eps):
    z = torch.randn(B_big, 512, generator=g).to(dev)
    za = z.clone().requires_grad_(True)
    la = loss_of(za)
    la.sum().backward()
    zb = z.clone().requires_grad_(True)
    lb = loss_of(zb)
    lb.sum().backward()
    assert torch.isfinite(la).all() and torch.isfinite(za.grad).all()
    assert torch.equal(la, lb) and torch.equal(za.grad, zb.grad), 'must be bit-reproducible'
    # ragged chunk: the last B_small candidates alone
    zs = z[-B_small:].clone().requires_grad_(True)
    ls = loss_of(zs)
    ls.sum().backward()
    assert torch.equal(ls, la[-B_small:]), 'loss depends on chunk composition'
    assert torch.equal(zs.grad, za.grad[-B_small:]), 'gradient depends on chunk composition'
    # directional finite difference on the small chunk, along the (normalised) gradient:
    # the directional derivative is then |grad|, the best-conditioned direction there is
    v = zs.grad / zs.grad.norm(dim=1, keepdim=True)
    with torch.no_grad():
        lp = loss_of(zs.detach() + fd_eps * v).double()
        lm = loss_of(zs.detach() - fd_eps * v).double()
    fd = ((lp - lm) / (2 * fd_eps)).cpu().numpy()
    an = (zs.grad * v).sum(1).double().cpu().numpy()
    err = np.abs(fd - an) / (np.abs(an) + 1e-4)
    assert np.max(err) < 0.05, (fd, an)
    return la


def test_c4_stylegan2_cars_512_chunks_9_and_5(dev):
    loss_of, g = _problem(dev, 512, 'cars')
    la = _check(loss_of, g, dev, 9, 5, 2e-2)
    assert la.shape == (9,)


def test_c5_stylegan2_ffhq_1024_shard_of_3(dev):
    # pop 22 over 8 ranks -> 3 candidates per rank
    loss_of, g = _problem(dev, 1024, 'ffhq')
    _check(loss_of, g, dev, 3, 2, 2e-2)
