"""Initialisation distributions of the latent variables: callables `(num_samples, shape)`
-> CPU tensor [num_samples, *shape], drawn from torch's default generator (the streams are
pinned bit-for-bit against reference pix2latent/distribution.py by tests/golden/distribution.npz).
"""
import torch


class _Gaussian(object):
    """sigma * N(0, I) + mu followed by an optional bounding rule"""

    def __init__(self, mu=0., sigma=1.):
        self.mu = mu if isinstance(mu, (int, float)) else mu.detach().cpu()
        self.sigma = sigma

    def bound(self, x):
        return x

    @torch.no_grad()
    def __call__(self, num_samples, shape):
        return self.bound(self.sigma * torch.randn((num_samples,) + tuple(shape)) + self.mu)


class TruncatedNormalModulo(_Gaussian):
    """N(mu, I) wrapped into (-2, 2) by float modulo.

    The constructor accepts `sigma` and `trunc` for compatibility, but like the reference
    (distribution.py:27-28) it does NOT honour them: the deviation is always 1 and the modulus
    always 2, whatever is passed."""

    def __init__(self, mu=0., sigma=1., trunc=2.):
        _Gaussian.__init__(self, mu, 1.0)
        self.trunc = 2.0

    def bound(self, x):
        return torch.fmod(x, self.trunc)


class _Clamped(_Gaussian):
    def __init__(self, sigma, trunc):
        _Gaussian.__init__(self, 0., sigma)
        self.trunc = trunc

    def bound(self, x):
        return x.clamp_(-self.trunc, self.trunc)


def truncated_clamp_normal(sigma=1.0, trunc=2.0):
    """N(0, sigma^2) hard-clamped to [-trunc, trunc] (the reference's version,
    distribution.py:39-58, raises NameError when called)"""
    return _Clamped(sigma, trunc)


def normal(sigma=1.0):
    """N(0, sigma^2)"""
    return _Gaussian(0., sigma)
