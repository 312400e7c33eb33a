"""BASELINE.json configurations 4 and 5 against the CPU ORACLE at their FULL sizes (VERDICT r5 "missing" #1):
StyleGAN2-cars 512^2 (z-space, injected noise, rows 64:-64 loss mask: examples/invert_stylegan2_cars_*.py:38-39,
reference model/stylegan2.py:116-119) and StyleGAN2-FFHQ 1024^2 (z, and W+ with the flat per-pixel noises:
model/stylegan2.py:122-138) pushed through generator + ProjectionLoss(VGG16) + backward on the device and
through oracle/stylegan2_ref.py + oracle/lpips_ref.py on the host.

Until round 6 every oracle comparison of this path ran a 64^2 network; the 128^2 ... 1024^2 layers, the 32- and
64-channel kernels and VGG at 512^2 / 1024^2 were only held to size-independent properties
(tests/test_fullsize_gpu.py).  The bars are north_star's: per-pixel |d| < 1e-3, loss |d| < 1e-3, identical
ranking; gradients are compared as everywhere else in this suite -- the native run's DISCRETE decisions
(leaky-ReLU signs, clamp, sign of the L1 term, VGG ReLUs and max-pool winners) replayed inside the oracle, and
every candidate within 1.5 x the fp32 oracle's own distance from the fp64 oracle + 2e-5.

Cost (32 host threads): one cars-512 candidate is ~0.14 TMAC forward in the oracle, one ffhq-1024 candidate
~0.4 TMAC; fp64 runs on ONE candidate at 1024^2.  PARITY UNPINNED as for every generator / LPIPS number
(oracle/stylegan2_ref.py header): the oracle restates rosinality's and lpips' published algorithms."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rep(t, b):
    return t.unsqueeze(0).expand(b, -1, -1, -1).contiguous()


def rel_rows(a, b):
    a = a.detach().cpu().double().reshape(a.shape[0], -1)
    b = b.detach().cpu().double().reshape(b.shape[0], -1)
    return (a - b).norm(dim=1) / b.norm(dim=1)


def _cast(W, dt):
    return {k: (v.to(dt) if torch.is_floating_point(v) else v) for k, v in W.items()}


class _Problem(object):
    def __init__(self, dev, size, name, masked):
        import warnings
        warnings.simplefilter('ignore')
        from pix2latent_amd.utils import synthetic as S
        from pix2latent_amd.model.stylegan2 import StyleGAN2
        import pix2latent_amd.loss_functions as LF
        from oracle import stylegan2_ref as R
        self.dev, self.size, self.R = dev, size, R
        self.W = S.stylegan2_weights(size, 0)
        self.Wv = S.lpips_vgg_weights(1)
        self.model = StyleGAN2(model=name, search='w+', weights=self.W, size=size, device=dev)
        assert self.model.im_res == size
        self.loss_fn = LF.ProjectionLoss(lpips_net='vgg', weights=self.Wv, device=dev)
        self.target = S.synthetic_target(size, 1)
        if masked:                                            # config 4: examples/invert_stylegan2_cars_*.py:38-39
            self.weight = torch.ones(3, size, size)
            self.loss_mask = torch.zeros(3, size, size)
            self.loss_mask[:, size // 8:-size // 8, :] += 1.0
        else:                                                 # config 5: weight mask, no loss mask
            self.weight = S.synthetic_weight_mask(size)
            self.loss_mask = None
        g = torch.Generator().manual_seed(41)
        self.noise1 = [torch.randn(1, 1, s[2], s[3], generator=g) for s in R.noise_shapes(size)]
        self.g = g

    def noises(self, B):
        return [n.expand(B, -1, -1, -1).contiguous() for n in self.noise1]

    def targets(self, B, dt=torch.float32, dev='cpu'):
        return tuple(None if t is None else _rep(t, B).to(dev, dt)
                     for t in (self.target, self.weight, self.loss_mask))

    # ---- native -------------------------------------------------------------------------------------
    def native_z(self, z, grad):
        B = z.size(0)
        zd = z.to(self.dev).requires_grad_(grad)
        t, w, m = self.targets(B, dev=self.dev)
        with torch.set_grad_enabled(grad):
            out = self.model.forward_z(zd, noises=[n.to(self.dev) for n in self.noises(B)])
            loss = self.loss_fn(out, t, w, m)
            if grad:
                loss.sum().backward()
        torch.cuda.synchronize()
        return out.detach(), loss.detach(), (zd.grad if grad else None), t

    # ---- oracle -------------------------------------------------------------------------------------
    def oracle_loss(self, out, B, dt, tape=None):
        from oracle import lpips_ref as L
        t, w, m = self.targets(B, dt)
        return L.projection_loss(_cast(self.Wv, dt), out, t, w, m, tape=tape)


@pytest.fixture(scope='module')
def cars(dev):
    return _Problem(dev, 512, 'cars', masked=True)


@pytest.fixture(scope='module')
def ffhq(dev):
    return _Problem(dev, 1024, 'ffhq', masked=False)


def test_c4_cars_512_pixels_loss_and_ranking_vs_oracle(cars):
    """4 candidates forward-only (the CMA re-score): pixels, ProjectionLoss-VGG and the ARGSORT of the four
    losses against the fp32 CPU oracle at 512^2 with the rows 64:-64 mask"""
    P, R, B = cars, cars.R, 4
    z = torch.randn(B, 512, generator=P.g)
    out, loss, _, _ = P.native_z(z, grad=False)
    t0 = time.time()
    with torch.no_grad():
        ref = R.forward_z(P.W, z, P.noises(B), P.size)
        lref = P.oracle_loss(ref, B, torch.float32)
    dpix = (out.cpu() - ref).abs().max().item()
    dloss = (loss.cpu() - lref).abs().max().item()
    print('cars-512: |dpix| %.3g |dloss| %.3g losses %s (oracle %.1f s)'
          % (dpix, dloss, [round(float(v), 5) for v in loss], time.time() - t0))
    assert out.shape == (B, 3, 512, 512)
    assert dpix < 1e-3 and dloss < 1e-3
    assert torch.equal(torch.argsort(loss.cpu()), torch.argsort(lref)), (loss, lref)
    # a chunk boundary must not show: candidates 2, 3 alone give the same bits
    out2, loss2, _, _ = P.native_z(z[2:], grad=False)
    assert torch.equal(out2, out[2:]) and torch.equal(loss2, loss[2:])


def _sg2_loss_tape(P, B, out, target_dev, with_mapping):
    from oracle import replay as RP
    sg2 = RP.sg2_decisions(P.model, B, out, with_mapping=with_mapping)
    return sg2, RP.loss_decisions(P.loss_fn, B, out, target_dev)


def test_c4_cars_512_replayed_gradients_vs_oracle(cars):
    """dz of 2 candidates through mapping + synthesis + masked L1 + 10 LPIPS-VGG at 512^2: the native run's
    decisions replayed in the fp32 and in the fp64 oracle; per candidate native <= 1.5 x fp32-oracle + 2e-5"""
    from oracle.masks import DecisionTape
    P, R, B = cars, cars.R, 2
    z = torch.randn(B, 512, generator=P.g)
    out, loss, gz, t_dev = P.native_z(z, grad=True)
    sg2, vgg = _sg2_loss_tape(P, B, out, t_dev, with_mapping=True)

    def oracle(dt, replayed):
        zr = z.detach().clone().to(dt).requires_grad_(True)
        nz = [n.to(dt) for n in P.noises(B)]
        if replayed:
            with R.replay(sg2):
                o = R.forward_z(_cast(P.W, dt), zr, nz, P.size)
            l = P.oracle_loss(o, B, dt, tape=DecisionTape(replay=vgg))
        else:
            o = R.forward_z(_cast(P.W, dt), zr, nz, P.size)
            l = P.oracle_loss(o, B, dt)
        l.sum().backward()
        return o.detach(), l.detach(), zr.grad
    t0 = time.time()
    _, l32, g32 = oracle(torch.float32, True)
    t1 = time.time()
    _, l64, g64 = oracle(torch.float64, True)
    t2 = time.time()
    _, _, gfree = oracle(torch.float32, False)
    dist, floor = rel_rows(gz, g64), rel_rows(g32, g64)
    print('cars-512 dz vs fp64 oracle (replayed decisions): native %s, fp32 oracle %s; free-running vs fp32 oracle '
          '%s  (oracle fp32 %.1f s, fp64 %.1f s)'
          % (['%.2e' % v for v in dist], ['%.2e' % v for v in floor],
             ['%.2e' % v for v in rel_rows(gz, gfree)], t1 - t0, t2 - t1))
    assert (loss.cpu().double() - l64).abs().max().item() < 1e-3
    assert bool((dist <= 1.5 * floor + 2e-5).all()), (dist, floor)
    assert rel_rows(gz, gfree).max().item() < 1e-2          # coarse free-running bound next to the tight one


def test_c5_ffhq_1024_z_pixels_and_loss_vs_oracle(ffhq):
    """2 candidates forward-only at 1024^2 (z-space, injected noise, weight mask): pixels and loss"""
    P, R, B = ffhq, ffhq.R, 2
    z = torch.randn(B, 512, generator=P.g)
    out, loss, _, _ = P.native_z(z, grad=False)
    t0 = time.time()
    with torch.no_grad():
        ref = R.forward_z(P.W, z, P.noises(B), P.size)
        lref = P.oracle_loss(ref, B, torch.float32)
    dpix = (out.cpu() - ref).abs().max().item()
    dloss = (loss.cpu() - lref).abs().max().item()
    print('ffhq-1024 z: |dpix| %.3g |dloss| %.3g (oracle %.1f s)' % (dpix, dloss, time.time() - t0))
    assert out.shape == (B, 3, 1024, 1024)
    assert dpix < 1e-3 and dloss < 1e-3
    assert torch.equal(torch.argsort(loss.cpu()), torch.argsort(lref))


def test_c5_ffhq_1024_wplus_and_noise_replayed_gradients_vs_oracle(ffhq):
    """BASELINE config 5 as specified: W+ latents [18,512] and the flat 2 796 176-element noise vector, both
    differentiated (model/stylegan2.py:122-138), ProjectionLoss-VGG at 1024^2.  Pixels and loss of 2
    candidates against the fp32 oracle; gradients with replayed decisions against the fp64 oracle -- fp64 on
    candidate 0 only (one 1024^2 candidate is ~1.6 TFLOP of fp64 forward + backward on the host)."""
    from oracle.masks import DecisionTape
    P, R, B = ffhq, ffhq.R, 2
    n_noise = sum(s[-2] * s[-1] for s in R.noise_shapes(P.size))
    assert n_noise == 2796176 and R.n_latent(P.size) == 18
    wplus = P.model.latent_mean.cpu().view(1, 1, 512) + 0.3 * torch.randn(B, 18, 512, generator=P.g)
    flat = torch.randn(B, n_noise, generator=P.g)
    wd, nd = wplus.to(P.dev).requires_grad_(True), flat.to(P.dev).requires_grad_(True)
    t, w, m = P.targets(B, dev=P.dev)
    out = P.model(wd, nd)
    loss = P.loss_fn(out, t, w, m)
    loss.sum().backward()
    torch.cuda.synchronize()
    sg2, vgg = _sg2_loss_tape(P, B, out.detach(), t, with_mapping=False)

    def oracle(dt, rows, replayed=True):
        n = rows.stop - rows.start
        wr = wplus[rows].detach().clone().to(dt).requires_grad_(True)
        nr = flat[rows].detach().clone().to(dt).requires_grad_(True)
        s_t = [mk[rows] for mk in sg2]
        v_t = [(a, k, mk[rows]) for a, k, mk in vgg]
        if replayed:
            with R.replay(s_t):
                o = R.forward_w(_cast(P.W, dt), wr, nr, P.size)
            l = P.oracle_loss(o, n, dt, tape=DecisionTape(replay=v_t))
        else:
            with torch.no_grad():
                o = R.forward_w(_cast(P.W, dt), wr, nr, P.size)
                return o, P.oracle_loss(o, n, dt), None, None
        l.sum().backward()
        return o.detach(), l.detach(), wr.grad, nr.grad
    t0 = time.time()
    ofree, lfree, _, _ = oracle(torch.float32, slice(0, B), replayed=False)
    dpix = (out.detach().cpu() - ofree).abs().max().item()
    dloss = (loss.detach().cpu() - lfree).abs().max().item()
    assert dpix < 1e-3 and dloss < 1e-3, (dpix, dloss)
    t1 = time.time()
    _, _, gw32, gn32 = oracle(torch.float32, slice(0, 1))
    t2 = time.time()
    _, l64, gw64, gn64 = oracle(torch.float64, slice(0, 1))
    t3 = time.time()
    res = {}
    for name, got, g32, g64 in (('dw+', wd.grad[:1], gw32, gw64), ('dnoise', nd.grad[:1], gn32, gn64)):
        dist, floor = rel_rows(got, g64), rel_rows(g32, g64)
        res[name] = (dist, floor)
    print('ffhq-1024 w+: |dpix| %.3g |dloss| %.3g; vs fp64 oracle (replayed): %s  (oracle fwd %.1f s, fp32 fwd+bwd '
          '%.1f s, fp64 %.1f s)' % (dpix, dloss, {k: ('%.2e' % v[0][0], '%.2e' % v[1][0]) for k, v in res.items()},
                                   t1 - t0, t2 - t1, t3 - t2))
    assert abs(float(loss[0]) - float(l64[0])) < 1e-3
    for name, (dist, floor) in res.items():
        assert bool((dist <= 1.5 * floor + 2e-5).all()), (name, dist, floor)
    # candidate 1 had no fp64 run: it must at least agree with what it gives alone (bits) -- the property the
    # 64^2 oracle tests extend to every batch position
    w1, n1 = wplus[1:].to(P.dev).requires_grad_(True), flat[1:].to(P.dev).requires_grad_(True)
    t1_, w1_, m1_ = P.targets(1, dev=P.dev)
    l1 = P.loss_fn(P.model(w1, n1), t1_, w1_, m1_)
    l1.sum().backward()
    assert torch.equal(l1.detach(), loss.detach()[1:]) and torch.equal(w1.grad, wd.grad[1:]) and \
        torch.equal(n1.grad, nd.grad[1:])
