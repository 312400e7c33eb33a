"""ORACLE (test infrastructure, not product code): CPU restatement of the StyleGAN2
generator that `pix2latent.model.StyleGAN2` drives (reference
pix2latent/model/stylegan2.py:66-138).

PARITY UNPINNED: the arithmetic lives in `rosinality/stylegan2-pytorch`, git-cloned at
run time from an unpinned HEAD (stylegan2.py:12-28,73-74,83) together with two CUDA
extension kernels (fused_bias_act, upfirdn2d); neither the repository nor the
checkpoints exist in this environment and the reference has no test for them.  This
file restates the published architecture (`model.py`: PixelNorm, EqualLinear,
ModulatedConv2d, NoiseInjection, FusedLeakyReLU, Blur / Upsample via upfirdn2d,
StyledConv, ToRGB, Generator) from recall; SURVEY.md §2.1 / §8 a14.  What IS pinned is
the wrapper's call contract: `forward_z` = generator([z], truncation=1.0) then
`clamp_(-1, 1)`; `forward_w` = generator([w+], input_is_latent=True, noise=noises);
`reshape_noise` (stylegan2.py:116-138).

Plain PyTorch-CPU fp32 functional ops, NCHW.  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F

STYLE_DIM = 512
N_MLP = 8
LR_MLP = 0.01
BLUR = (1, 3, 3, 1)


def channels(channel_multiplier=2):
    cm = channel_multiplier
    return {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm,
            512: 32 * cm, 1024: 16 * cm}


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """upfirdn2d_native [3P-recall]: zero-insert upsample, pad, FIR with the flipped
    kernel, decimate.  x: [B,C,H,W]."""
    b, c, h, w = x.shape
    if up > 1:
        o = x.new_zeros(b, c, h, up, w, up)
        o[:, :, :, 0, :, 0] = x
        x = o.view(b, c, h * up, w * up)
    p0, p1 = pad
    x = F.pad(x, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    kh, kw = kernel.shape
    wk = torch.flip(kernel, [0, 1]).view(1, 1, kh, kw).to(x.dtype)
    x = F.conv2d(x.reshape(b * c, 1, x.shape[2], x.shape[3]), wk)
    x = x.view(b, c, x.shape[2], x.shape[3])
    return x[:, :, ::down, ::down]


# Decision replay (tests of the piecewise gradients): while a tape is set, every fused_leaky_relu takes the
# sign of its pre-activation -- and the final clamp its pass-through mask -- from the tape, in call order
# (mapping layers 1..8 when the mapping runs, styled convs 0..n-1, then the clamp), instead of deciding for
# itself.  oracle/replay.py builds the tape from a native run's saved activations.
_TAPE = None


class replay(object):
    """with replay(items): ... -- items = list of bool tensors (True = pre-activation > 0 / not clamped)"""

    def __init__(self, items):
        self.items = list(items)

    def __enter__(self):
        global _TAPE
        _TAPE = iter(self.items)
        return self

    def __exit__(self, *exc):
        global _TAPE
        left = sum(1 for _ in _TAPE)
        _TAPE = None
        assert exc[0] is not None or left == 0, '%d replayed decisions were not consumed' % left


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    shape = [1, -1] + [1] * (x.dim() - 2)
    if _TAPE is not None:
        pre = x + bias.view(*shape)
        mask = next(_TAPE).to(pre.device)
        assert mask.shape == pre.shape, (mask.shape, pre.shape)
        return torch.where(mask, pre, pre * negative_slope) * scale
    return F.leaky_relu(x + bias.view(*shape), negative_slope) * scale


def _clamp(img):
    if _TAPE is not None:
        keep = next(_TAPE).to(img.device)
        return torch.where(keep, img, img.detach().clamp(-1.0, 1.0))
    return img.clamp(-1.0, 1.0)


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    scale = (1 / math.sqrt(weight.shape[1])) * lr_mul
    if activation:
        return fused_leaky_relu(F.linear(x, weight * scale), bias * lr_mul)
    return F.linear(x, weight * scale, bias=bias * lr_mul)


def mapping(W, z):
    """PixelNorm + 8 x EqualLinear(512, 512, lr_mul=0.01, fused_lrelu)."""
    x = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(1, N_MLP + 1):
        x = equal_linear(x, W['style.%d.weight' % i], W['style.%d.bias' % i], LR_MLP, True)
    return x


def modulated_conv(W, p, x, style, demodulate=True, upsample=False):
    """ModulatedConv2d.forward [3P-recall]."""
    weight = W[p + '.weight']                         # [1, out, in, k, k]
    _, out_c, in_c, k, _ = weight.shape
    b, _, h, w = x.shape
    s = equal_linear(style, W[p + '.modulation.weight'], W[p + '.modulation.bias'])
    scale = 1 / math.sqrt(in_c * k * k)
    wgt = scale * weight * s.view(b, 1, in_c, 1, 1)
    if demodulate:
        demod = torch.rsqrt(wgt.pow(2).sum([2, 3, 4]) + 1e-8)
        wgt = wgt * demod.view(b, out_c, 1, 1, 1)
    if upsample:
        xi = x.reshape(1, b * in_c, h, w)
        wt = wgt.transpose(1, 2).reshape(b * in_c, out_c, k, k)
        out = F.conv_transpose2d(xi, wt, padding=0, stride=2, groups=b)
        out = out.view(b, out_c, out.shape[2], out.shape[3])
        kern = make_kernel(BLUR) * 4
        out = upfirdn2d(out, kern, pad=(1, 1))
    else:
        xi = x.reshape(1, b * in_c, h, w)
        out = F.conv2d(xi, wgt.view(b * out_c, in_c, k, k), padding=k // 2, groups=b)
        out = out.view(b, out_c, out.shape[2], out.shape[3])
    return out


def styled_conv(W, p, x, style, noise, upsample=False):
    """StyledConv.forward: modulated conv -> noise injection -> fused leaky ReLU."""
    out = modulated_conv(W, p + '.conv', x, style, True, upsample)
    out = out + W[p + '.noise.weight'] * noise
    return fused_leaky_relu(out, W[p + '.activate.bias'])


def to_rgb(W, p, x, style, skip=None):
    out = modulated_conv(W, p + '.conv', x, style, demodulate=False) + W[p + '.bias']
    if skip is not None:
        skip = upfirdn2d(skip, make_kernel(BLUR) * 4, up=2, pad=(2, 1))
        out = out + skip
    return out


def num_layers(size):
    return (int(math.log2(size)) - 2) * 2 + 1


def n_latent(size):
    return int(math.log2(size)) * 2 - 2


def noise_shapes(size):
    return [[1, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)] for i in range(num_layers(size))]


def synthesis(W, latent, noises, size):
    """Generator.forward after the mapping: latent [B, n_latent, 512], noises = list of
    [B,1,h,w] (explicit: the reference draws fresh normal noise per call in z-mode)."""
    b = latent.shape[0]
    out = W['input.input'].repeat(b, 1, 1, 1)
    out = styled_conv(W, 'conv1', out, latent[:, 0], noises[0])
    skip = to_rgb(W, 'to_rgb1', out, latent[:, 1])
    i = 1
    for j in range(int(math.log2(size)) - 2):
        out = styled_conv(W, 'convs.%d' % (2 * j), out, latent[:, i], noises[2 * j + 1], upsample=True)
        out = styled_conv(W, 'convs.%d' % (2 * j + 1), out, latent[:, i + 1], noises[2 * j + 2])
        skip = to_rgb(W, 'to_rgbs.%d' % j, out, latent[:, i + 2], skip)
        i += 2
    return skip


def forward_z(W, z, noises, size):
    """StyleGAN2.forward_z (reference stylegan2.py:116-119), noise made explicit."""
    w = mapping(W, z)
    latent = w.unsqueeze(1).repeat(1, n_latent(size), 1)
    return _clamp(synthesis(W, latent, noises, size))


def reshape_noise(z, size):
    """StyleGAN2.reshape_noise (reference stylegan2.py:128-138)."""
    st, out = 0, []
    for d in noise_shapes(size):
        en = st + d[-2] * d[-1]
        out.append(z[:, st:en].reshape(-1, 1, d[-2], d[-1]))
        st = en
    assert z.size(1) == en
    return out


def forward_w(W, wplus, noises_flat, size):
    """StyleGAN2.forward_w (reference stylegan2.py:122-125): w+ latents and flat noises."""
    latent = wplus if wplus.dim() == 3 else wplus.unsqueeze(1).repeat(1, n_latent(size), 1)
    return _clamp(synthesis(W, latent, reshape_noise(noises_flat, size), size))
