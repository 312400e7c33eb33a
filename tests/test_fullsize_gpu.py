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


# ---------------------------------------------------------------------------------------
# BASELINE config 5 AS SPECIFIED: StyleGAN2-FFHQ 1024^2 searched in W+ (latents [18,512] and
# the flat per-pixel noise vector both optimised), and the SpatialTransform search around it
# ---------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def ffhq_wplus(dev):
    import warnings
    warnings.simplefilter('ignore')
    from pix2latent_amd.utils import synthetic as S
    from pix2latent_amd.model.stylegan2 import StyleGAN2
    import pix2latent_amd.loss_functions as LF
    model = StyleGAN2(model='ffhq', search='w+', device=dev)
    assert model.im_res == 1024 and model._desc.n_latent == 18
    loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=S.lpips_vgg_weights(1), device=dev)
    n_noise = sum(s[-2] * s[-1] for s in model.noise_shape)
    assert n_noise == 2796176
    return model, loss_fn, n_noise


def test_c5_ffhq_1024_wplus_and_noise_gradients_shard_of_3(ffhq_wplus, dev):
    """model(z=w+ [B,18,512], noises=[B, 2 796 176]) (reference model/stylegan2.py:122-138)
    -> ProjectionLoss at 1024^2: gradients to w+ AND to the flat noises are finite,
    bit-reproducible, independent of the chunk a candidate sits in, and agree with a central
    finite difference along the gradient direction."""
    from pix2latent_amd.utils import synthetic as S
    model, loss_fn, n_noise = ffhq_wplus
    size, B = 1024, 3
    target = S.synthetic_target(size, 1).to(dev)
    weight = S.synthetic_weight_mask(size).to(dev)
    g = torch.Generator().manual_seed(21)
    w0 = (model.latent_mean.cpu().view(1, 1, 512) +
          0.3 * torch.randn(B, 18, 512, generator=g)).to(dev)
    n0 = torch.randn(B, n_noise, generator=g).to(dev)

    def rep(t, b):
        return t.unsqueeze(0).expand(b, -1, -1, -1).contiguous()

    def run(w, n):
        w = w.clone().requires_grad_(True)
        n = n.clone().requires_grad_(True)
        loss = loss_fn(model(w, n), rep(target, w.size(0)), rep(weight, w.size(0)))
        loss.sum().backward()
        return loss.detach(), w.grad, n.grad
    la, gwa, gna = run(w0, n0)
    lb, gwb, gnb = run(w0, n0)
    assert torch.isfinite(la).all() and torch.isfinite(gwa).all() and torch.isfinite(gna).all()
    assert torch.equal(la, lb) and torch.equal(gwa, gwb) and torch.equal(gna, gnb), \
        'must be bit-reproducible'
    ls, gws, gns = run(w0[1:], n0[1:])                       # a chunk of 2 of the same candidates
    assert torch.equal(ls, la[1:]) and torch.equal(gws, gwa[1:]) and torch.equal(gns, gna[1:]), \
        'loss / gradients depend on chunk composition'
    assert gwa.abs().max() > 0 and gna.abs().max() > 0
    # central finite differences along each gradient (w+ and noise separately)
    with torch.no_grad():
        for name, grad, eps in (('w+', gwa, 2e-2), ('noise', gna, 5e-1)):
            v = grad / grad.flatten(1).norm(dim=1).view(-1, *([1] * (grad.dim() - 1)))
            dw, dn = (eps * v, 0) if name == 'w+' else (0, eps * v)
            lp = loss_fn(model(w0 + dw, n0 + dn), rep(target, B), rep(weight, B)).double()
            lm = loss_fn(model(w0 - dw, n0 - dn), rep(target, B), rep(weight, B)).double()
            fd = ((lp - lm) / (2 * eps)).cpu().numpy()
            an = (grad * v).flatten(1).sum(1).double().cpu().numpy()
            err = np.abs(fd - an) / (np.abs(an) + 1e-5)
            assert np.max(err) < 0.05, (name, fd, an)


def test_c5_transform_basincma_around_ffhq_1024_wplus(ffhq_wplus, dev):
    """TransformBasinCMAOptimizer (reference transform/transform_optimizer.py:165-255) with a
    3-d transformation (-> population 7, executed as one chunk of 7, max_batch_size 8 as in
    examples/invert_biggan_with_transform.py:120-121) around the W+ generator at 1024^2:
    2 generations, targets / weights warped per candidate by p2l_affine_grid_sample, Adam on
    w+ and noises, un-warped re-score against the ORIGINAL target under the binarised weight
    for tell (base_cma_optimizer.py:120-138), propagation of w+."""
    from pix2latent_amd import VariableManager
    from pix2latent_amd.utils import synthetic as S
    from pix2latent_amd.utils.image import binarize
    from pix2latent_amd.transform import SpatialTransform, TransformBasinCMAOptimizer
    model, loss_fn, n_noise = ffhq_wplus
    size = 1024
    target = S.synthetic_target(size, 1)
    weight = S.synthetic_weight_mask(size)
    vm = VariableManager(device=dev)
    vm.register('z', (18, 512), 'input', learning_rate=0.05,
                default=model.latent_mean.cpu().view(1, 512).repeat(18, 1))
    vm.register('noises', (n_noise,), 'input', learning_rate=0.05,
                default=torch.randn(n_noise, generator=torch.Generator().manual_seed(22)))
    vm.register('target', (3, size, size), 'output', requires_grad=False, default=target)
    vm.register('weight', (3, size, size), 'output', requires_grad=False, default=weight)
    vm.register('t', (3,), 'transform', requires_grad=False, grad_free=True)
    opt = TransformBasinCMAOptimizer(model, vm, loss_fn, max_batch_size=8)
    opt.cma_seed = 3
    st = SpatialTransform()
    opt.register_transform(st, 't', 'target')
    opt.register_transform(st, 't', 'weight')
    opt.set_variable_propagation('z')
    torch.manual_seed(23)
    variables, (outs, targets, cand_target), losses = opt.optimize(meta_steps=2, grad_steps=2)
    assert opt.num_samples == 7
    final = np.asarray(losses)
    assert final.shape == (7,) and np.isfinite(final).all()
    # what was told after generation 0: losses of the un-warped outputs against the original
    # target -- recompute them from the optimizer's pieces for the LAST population and check
    # the candidate bookkeeping instead (generation 0's tensors are gone)
    t_rows = torch.stack(list(variables.transform.t.data))
    assert t_rows.shape == (7, 3) and len(opt.transform_tracked) == 2
    assert opt.get_candidate().shape == (3,) and np.isfinite(opt._best_loss)
    # the warped targets equal the fused kernel applied to the registered target
    with torch.no_grad():
        want = st(target.to(dev).unsqueeze(0).expand(7, -1, -1, -1).contiguous(), t_rows)
        got = torch.stack(list(variables.output.target.data))
        assert torch.equal(got, want)
        # ... and that kernel matches torch's affine_grid + grid_sample at this size
        ref = torch.nn.functional.grid_sample(
            target.unsqueeze(0).expand(2, -1, -1, -1),
            torch.nn.functional.affine_grid(st._theta((st._t + 0.1 * t_rows[:2].cpu())[:, 0],
                                                      (st._t + 0.1 * t_rows[:2].cpu())[:, 1:]),
                                            [2, 3, size, size], align_corners=False),
            align_corners=False)
        assert (got[:2].cpu() - ref).abs().max().item() < 1e-4   # fp32 grid coordinates at 1024 px
        # the un-warped scoring used for tell is reproducible and finite at this size
        a = opt.losses_for_tell(variables)
        b = opt.losses_for_tell(variables)
        assert a.shape == (7,) and np.isfinite(a).all() and np.array_equal(a, b)
        restored = st(opt.out, t_rows, invert=True)
        direct = loss_fn(restored, target.to(dev).unsqueeze(0),
                         binarize(weight.to(dev).unsqueeze(0))).cpu().numpy()
        assert np.allclose(a, direct, rtol=0, atol=1e-6)
    assert cand_target.shape == (3, size, size)
    assert 'z' in opt.vp_means and opt.vp_means['z'].shape == (18, 512)


def test_c4_hybrid_nevergrad_cars_512_num_samples_32(dev):
    """BASELINE config 4 run through the optimizer itself at full size
    (examples/invert_stylegan2_cars_hybrid_ng.py:41-44,70-73,103-114): StyleGAN2-cars 512^2,
    z in R^512, 32 samples in the reference's chunks 9,9,9,5, hook
    Compose(NormalPerturb(0.05), Clamp(2)), loss mask rows 64:-64, 1 ask/tell generation."""
    import warnings
    warnings.simplefilter('ignore')
    from pix2latent_amd import VariableManager, distribution
    from pix2latent_amd.utils import synthetic as S, function_hooks as hook
    from pix2latent_amd.model.stylegan2 import StyleGAN2
    from pix2latent_amd.optimizer import HybridNevergradOptimizer
    import pix2latent_amd.loss_functions as LF
    size = 512
    gen = StyleGAN2(model='cars', search='z', device=dev)
    g = torch.Generator().manual_seed(31)
    fixed = [torch.randn(1, 1, s[2], s[3], generator=g).to(dev) for s in gen.noise_shape]
    calls = []

    class Model(torch.nn.Module):           # fixed injected noise
        def forward(self, z=None):
            calls.append(int(z.size(0)))
            return gen.forward_z(z, noises=[n.expand(z.size(0), -1, -1, -1).contiguous() for n in fixed])
    loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=S.lpips_vgg_weights(1), device=dev)
    target = S.synthetic_target(size, 1)
    weight = torch.ones(3, size, size)
    loss_mask = torch.zeros(3, size, size)
    loss_mask[:, 64:-64, :] += 1.0
    vm = VariableManager(device=dev)
    vm.register('z', (512,), 'input', distribution=distribution.TruncatedNormalModulo(1.0, 2.0),
                learning_rate=0.05, grad_free=True,
                hook_fn=hook.Compose(hook.NormalPerturb(sigma=0.05), hook.Clamp(2.0)))
    for name, t in (('target', target), ('weight', weight), ('loss_mask', loss_mask)):
        vm.register(name, (3, size, size), 'output', requires_grad=False, default=t)
    opt = HybridNevergradOptimizer('CMA', Model(), vm, loss_fn, max_batch_size=9)
    opt.ng_seed = 0
    torch.manual_seed(32)
    variables, outs, losses = opt.optimize(num_samples=32, meta_steps=1, grad_steps=2,
                                           last_grad_steps=2)
    # 2 Adam steps + re-score, then 2 Adam steps of the last draw; every pass = 9,9,9,5
    assert calls == [9, 9, 9, 5] * 5
    ng = opt.sampler.opt
    assert ng.num_ask == 64 and ng.num_tell == 32 and ng.budget == 2
    final = np.array(losses[-1][1]['loss'])
    assert final.shape == (32,) and np.isfinite(final).all() and losses[-1][0] == 4
    z = torch.stack(list(variables.input.z.data))
    # Clamp(2) runs (after NormalPerturb) BEFORE each forward; the Adam update that follows
    # moves a latent by at most ~lr
    assert z.abs().max().item() <= 2.0 + 0.05 + 1e-3
    asked = torch.stack([torch.as_tensor(c.args[0], dtype=torch.float32) for c in opt.sampler._handle])
    assert asked.shape == (32, 512) and not torch.equal(asked.to(dev), z.detach())
