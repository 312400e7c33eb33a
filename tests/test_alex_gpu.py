"""LPIPS-AlexNet path (the reference default `ProjectionLoss(lpips_net='alex')`,
pix2latent/loss_functions.py:87): kernel parity against plain PyTorch-CPU fp32 ops, then
loss + gradient parity against the CPU oracle (oracle/lpips_ref.py) on seeded weights."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(t, dev):
    return t.permute(0, 2, 3, 1).contiguous().to(dev)


def nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-20)).item()


@pytest.fixture(scope='module')
def O(dev):
    from pix2latent_amd import ops
    return ops


# (Cin, Cout, K, stride, pad, H): the five AlexNet convs at a 256^2 input + odd ragged sizes
GCONV_CASES = [(16, 64, 11, 4, 2, 256), (64, 192, 5, 1, 2, 31), (192, 384, 3, 1, 1, 15),
               (384, 256, 3, 1, 1, 15), (256, 256, 3, 1, 1, 15), (16, 64, 11, 4, 2, 67),
               (64, 64, 5, 1, 2, 9)]


@pytest.mark.parametrize('case', GCONV_CASES)
def test_gconv_forward_and_dgrad(dev, O, case):
    Cin, Cout, K, S, P, H = case
    g = torch.Generator().manual_seed(sum(case))
    B = 3
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    bias = torch.randn(Cout, generator=g)
    s = torch.rand(Cin, generator=g) + 0.5
    t = torch.randn(Cin, generator=g) * 0.1
    ref = F.relu(F.conv2d(x * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1), w, bias, stride=S, padding=P))
    wp = O.pack_conv_weight(w.to(dev), K * K, Cout, Cin, False)
    y = O.gconv(nhwc(x, dev), wp, B, H, H, Cin, Cout, K, S, P, bias=bias.to(dev), pro_s=s.to(dev),
                pro_t=t.to(dev), relu=True)
    assert tuple(y.shape[1:3]) == tuple(ref.shape[2:])
    assert relerr(nchw(y), ref) < 2e-5
    if S == 1:
        # input gradient = same kernel with the flipped / transposed packing, + residual, masked
        Ho = ref.shape[2]
        dy = torch.randn(B, Cout, Ho, Ho, generator=g)
        extra = torch.randn(B, Cin, H, H, generator=g)
        m = torch.randn(B, Cin, H, H, generator=g)
        xr = x.clone().requires_grad_(True)
        F.conv2d(xr, w, None, stride=1, padding=P).backward(dy)
        want = (xr.grad + extra) * (m > 0)
        wt = O.pack_conv_weight(w.to(dev), K * K, Cin, Cout, True)
        got = O.gconv(nhwc(dy, dev), wt, B, Ho, Ho, Cout, Cin, K, 1, K - 1 - P, res=nhwc(extra, dev),
                      mask=nhwc(m, dev))
        assert relerr(nchw(got), want) < 2e-5


@pytest.mark.parametrize('H,C', [(63, 64), (31, 192), (8, 4), (7, 8)])
def test_maxpool3s2(dev, O, H, C):
    g = torch.Generator().manual_seed(H * C)
    B = 2
    x = F.relu(torch.randn(B, C, H, H, generator=g)).requires_grad_(True)   # many exact-zero ties
    y = F.max_pool2d(x, 3, 2)
    gp = torch.randn(y.shape, generator=g)
    gt = torch.randn(x.shape, generator=g)
    y.backward(gp)
    want = (x.grad + gt) * (x.detach() > 0)
    yy = O.maxpool3s2_fwd(nhwc(x.detach(), dev))
    assert torch.equal(nchw(yy), y.detach())
    dx = O.maxpool3s2_bwd(nhwc(x.detach(), dev), nhwc(gp, dev), nhwc(gt, dev))
    assert relerr(nchw(dx), want) < 1e-6


def test_conv1_dgrad(dev, O):
    g = torch.Generator().manual_seed(5)
    B, H = 2, 256
    w = torch.randn(64, 3, 11, 11, generator=g) * 0.05
    x = torch.randn(B, 3, H, H, generator=g, requires_grad=True)
    y = F.conv2d(x, w, None, stride=4, padding=2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    w3 = w.permute(2, 3, 1, 0).reshape(121, 3, 64).contiguous()
    d = O.conv1_dgrad(nhwc(dy, dev), w3.to(dev), H, H, 11, 4, 2)
    assert relerr(nchw(d)[:, :3], x.grad) < 2e-5
    assert float(d[..., 3:].abs().max()) == 0.0


@pytest.mark.parametrize('C', [192, 384])
def test_lpips_tap_alex_widths(dev, O, C):
    g = torch.Generator().manual_seed(C)
    B, h = 2, 15
    f = F.relu(torch.randn(B, C, h, h, generator=g)).requires_grad_(True)
    ft = F.relu(torch.randn(B, C, h, h, generator=g))
    lin = torch.rand(C, generator=g) / C
    wt = torch.rand(B, h, h, generator=g)
    wsum = torch.tensor([3.0, 5.0])

    def norm(a):
        return a / (torch.sqrt((a ** 2).sum(1, keepdim=True)) + 1e-10)
    loss = (((norm(f) - norm(ft)) ** 2 * lin.view(1, C, 1, 1)).sum(1) * wt).sum((1, 2)) / wsum
    gl = torch.tensor([0.7, -1.3])
    (loss * gl).sum().backward()
    nft = O.lpips_normalize(nhwc(ft, dev))
    got = O.lpips_tap_fwd(nhwc(f.detach(), dev), nft, lin.to(dev), wt.to(dev), wsum.to(dev))
    assert relerr(got.cpu(), loss.detach()) < 1e-5
    df = O.lpips_tap_bwd(nhwc(f.detach(), dev), nft, lin.to(dev), wt.to(dev), (gl / wsum).to(dev))
    assert relerr(nchw(df), f.grad) < 1e-4


@pytest.mark.parametrize('h', [63, 31, 15, 127])
def test_bilinear_adjoint_non_integer_ratio(dev, O, h):
    g = torch.Generator().manual_seed(h)
    B, H = 2, 256
    wsrc = torch.rand(B, H, H, generator=g)
    m = torch.randn(B, 1, h, h, generator=g, requires_grad=True)
    up = F.interpolate(m, size=(H, H), mode='bilinear', align_corners=False)
    (up[:, 0] * wsrc).sum().backward()
    wt = O.bilinear_adjoint(wsrc.to(dev), h, h)
    assert relerr(wt.cpu(), m.grad[:, 0]) < 1e-5


@pytest.mark.parametrize('size,with_mask', [(256, False), (256, True), (64, True)])
def test_projection_loss_alex_vs_oracle(dev, size, with_mask):
    """ProjectionLoss() with the reference default network: loss |d| < 1e-3 (measured
    ~1e-6), identical candidate ranking, gradient to the image vs the fp64 oracle within
    3x the fp32 oracle's own distance."""
    import warnings
    warnings.simplefilter('ignore')
    import pix2latent_amd.loss_functions as LF
    from pix2latent_amd.utils import synthetic as S
    from oracle import lpips_ref as L
    Wa = S.lpips_alex_weights(2)
    g = torch.Generator().manual_seed(3)
    B = 5
    target = S.synthetic_target(size, 1)
    out = (target.unsqueeze(0) + 0.3 * torch.randn(B, 3, size, size, generator=g)).clamp(-1, 1)
    weight = S.synthetic_weight_mask(size)
    loss_mask = None
    if with_mask:
        loss_mask = torch.zeros(3, size, size)
        loss_mask[:, size // 8:-size // 8, :] += 1.0
    rep = lambda t: None if t is None else t.unsqueeze(0).repeat(B, 1, 1, 1)
    o32 = out.clone().requires_grad_(True)
    l32 = L.projection_loss(Wa, o32, rep(target), rep(weight), rep(loss_mask))
    l32.sum().backward()
    W64 = {k: v.double() for k, v in Wa.items()}
    d = lambda t: None if t is None else t.double()
    o64 = out.double().requires_grad_(True)
    L.projection_loss(W64, o64, d(rep(target)), d(rep(weight)), d(rep(loss_mask))).sum().backward()

    loss_fn = LF.ProjectionLoss(weights=Wa, device=dev)      # lpips_net default = 'alex'
    od = out.to(dev).requires_grad_(True)
    ld = loss_fn(od, target.to(dev), weight.to(dev), None if loss_mask is None else loss_mask.to(dev))
    ld.sum().backward()
    assert (ld.detach().cpu() - l32.detach()).abs().max().item() < 1e-3
    assert np.array_equal(np.argsort(ld.detach().cpu().numpy()), np.argsort(l32.detach().numpy()))

    def rel(a, b):
        a, b = a.double().flatten(), b.double().flatten()
        return ((a - b).norm() / b.norm()).item()
    floor = rel(o32.grad, o64.grad)
    got = rel(od.grad.cpu(), o64.grad)
    assert got < 3 * floor + 2e-4, (got, floor)
    # re-score is bit-reproducible
    with torch.no_grad():
        a = loss_fn(od.detach(), target.to(dev), weight.to(dev), None if loss_mask is None else loss_mask.to(dev))
        b = loss_fn(od.detach(), target.to(dev), weight.to(dev), None if loss_mask is None else loss_mask.to(dev))
    assert torch.equal(a, b)
